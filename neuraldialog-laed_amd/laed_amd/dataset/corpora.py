"""Corpus readers (laed/dataset/corpora.py, Python 3): raw files -> token lists -> vocabulary -> id corpora.

Rules kept from the reference (file:line there):
  * tokeniser (utils.py:153-154): every maximal match of  \\w+ | <sil> | [^\\w\\s]+   -- nltk.RegexpTokenizer is
    re.findall with UNICODE|MULTILINE|DOTALL, restated here with `re` (nltk is not needed);
  * PTB (corpora.py:149-158): one sentence per line, the FIRST token of each line is dropped, <s> ... </s> added;
  * DailyDialog (:229-262): non-ASCII bytes dropped, turns split on __eou__ (the trailing piece discarded),
    lower-cased, speaker = turn index % 2, per-turn act / emotion in `meta`, every dialog opens with <s> <d> </s>
    and (unless config.include_eod is False) closes with <s> </d> </s>;
  * Stanford multi-domain (:53-80): JSON dialogues, speaker tag <sys>/<usr> after <s>, closing </d> turn only when
    config.include_eod is set and true;
  * vocabulary (:82-104,160-180,264-285): words of the TRAIN split by descending count, cut at max_vocab_cnt, after
    [<pad>, <unk>] (PTB) or [<pad>, <unk>, <sys>, <usr>] (dialog corpora); unknown words map to <unk>.
Parity: tests/golden/corpora.npz records what the reference's reader methods, run unmodified under Python 3
(tests/golden/make_goldens_corpora.py; count ties come out in first-occurrence order there as here), made of real
Stanford / DailyDialog text and PTB-format lines; tests/test_dataset_cpu.py holds these readers to it id for id.  The
tokeniser's regex engine is `re` on both sides (nltk absent): the pattern is the reference's, the matching is restated.
"""
import json
import os
import re
from collections import Counter

import numpy as np

from ..utils import Pack

PAD = '<pad>'
UNK = '<unk>'
BOS = '<s>'
EOS = '</s>'
BOD = "<d>"
EOD = "</d>"
BOT = "<t>"
EOT = "</t>"
ME = "<me>"
OT = "<ot>"
SYS = "<sys>"
USR = "<usr>"
KB = "<kb>"
SEP = "|"
REQ = "<requestable>"
INF = "<informable>"
WILD = "%s"

_CHAT_TOKEN = re.compile(r'\w+|<sil>|[^\w\s]+', re.UNICODE | re.MULTILINE | re.DOTALL)


def get_chat_tokenize():
    return _CHAT_TOKEN.findall


class _Corpus(object):
    """Shared vocabulary / id-mapping half of the three corpus classes."""
    specials = [PAD, UNK]

    def _turns(self, split):
        """Iterate every turn Pack of a split (flat corpora are lists of turns, dialog corpora lists of lists)."""
        for item in split:
            if isinstance(item, list):
                for turn in item:
                    yield turn
            else:
                yield item

    def _build_vocab(self, max_vocab_cnt):
        all_words = [w for turn in self._turns(self.train_corpus) for w in turn.utt]
        vocab_count = Counter(all_words).most_common()
        raw_vocab_size = len(vocab_count)
        discard_wc = sum(c for _, c in vocab_count[max_vocab_cnt:])
        vocab_count = vocab_count[:max_vocab_cnt]
        print("Load corpus with train size %d, valid size %d, test size %d raw vocab size %d vocab size %d at cut_off %d "
              "OOV rate %f" % (len(self.train_corpus), len(self.valid_corpus), len(self.test_corpus), raw_vocab_size,
                               len(vocab_count), vocab_count[-1][1], float(discard_wc) / max(len(all_words), 1)))
        self.vocab = list(self.specials) + [t for t, _ in vocab_count]
        self.rev_vocab = {t: idx for idx, t in enumerate(self.vocab)}
        self.unk_id = self.rev_vocab[UNK]

    def _sent2id(self, sent):
        return [self.rev_vocab.get(t, self.unk_id) for t in sent]

    def _id_turn(self, turn):
        return Pack(utt=self._sent2id(turn.utt), speaker=turn.speaker, meta=turn.get('meta'))

    def _to_id_corpus(self, data):
        return [[self._id_turn(t) for t in item] if isinstance(item, list) else self._id_turn(item) for item in data]

    def get_corpus(self):
        return Pack(train=self._to_id_corpus(self.train_corpus), valid=self._to_id_corpus(self.valid_corpus),
                    test=self._to_id_corpus(self.test_corpus))

    @staticmethod
    def _report(all_lens, all_dialog_lens=None):
        print("Max utt len %d, mean utt len %.2f" % (np.max(all_lens), float(np.mean(all_lens))))
        if all_dialog_lens is not None:
            print("Max dialog len %d, mean dialog len %.2f" % (np.max(all_dialog_lens), float(np.mean(all_dialog_lens))))


class PTBCorpus(_Corpus):
    specials = [PAD, UNK]

    def __init__(self, config):
        self.config = config
        self._path = config.data_dir[0]
        self.max_utt_len = config.max_utt_len
        self.tokenize = get_chat_tokenize()
        self.train_corpus = self._read_file(os.path.join(self._path, 'ptb.train.txt'))
        self.valid_corpus = self._read_file(os.path.join(self._path, 'ptb.valid.txt'))
        self.test_corpus = self._read_file(os.path.join(self._path, 'ptb.test.txt'))
        self._build_vocab(config.max_vocab_cnt)
        print("Done loading corpus")

    def _read_file(self, path):
        with open(path, 'r', encoding='utf-8', errors='ignore') as f:
            return self._process_data(f.readlines())

    def _process_data(self, data):
        all_text = [Pack(utt=[BOS] + self.tokenize(line)[1:] + [EOS], speaker=0) for line in data]
        self._report([len(t.utt) for t in all_text])
        return all_text


class _DialogCorpus(_Corpus):
    specials = [PAD, UNK, SYS, USR]
    bod_utt = [BOS, BOD, EOS]
    eod_utt = [BOS, EOD, EOS]

    def _finish(self, dialogs):
        self._report([len(t.utt) for d in dialogs for t in d[1:] if t.utt != self.eod_utt] or [0],
                     [len(d) for d in dialogs] or [0])
        return dialogs


class DailyDialogCorpus(_DialogCorpus):
    def __init__(self, config):
        self.config = config
        self._path = config.data_dir[0]
        self.max_utt_len = config.max_utt_len
        self.tokenize = get_chat_tokenize()
        self.train_corpus = self._read_file(os.path.join(self._path, 'train'))
        self.valid_corpus = self._read_file(os.path.join(self._path, 'validation'))
        self.test_corpus = self._read_file(os.path.join(self._path, 'test'))
        self._build_vocab(config.max_vocab_cnt)
        print("Done loading corpus")

    def _read_file(self, path):
        def lines(name):
            with open(os.path.join(path, name), 'rb') as f:
                return f.readlines()
        return self._process_dialog(zip(lines('dialogues.txt'), lines('dialogues_act.txt'),
                                        lines('dialogues_emotion.txt')))

    def _process_dialog(self, data):
        include_eod = not hasattr(self.config, 'include_eod') or self.config.include_eod
        dialogs = []
        for raw_dialog, raw_act, raw_emotion in data:
            text = raw_dialog.decode('ascii', 'ignore')
            acts, emotions = raw_act.split(), raw_emotion.split()
            dialog = [Pack(utt=list(self.bod_utt), speaker=0, meta=None)]
            for t_id, turn in enumerate(text.split('__eou__')[:-1]):
                dialog.append(Pack(utt=[BOS] + self.tokenize(turn.lower()) + [EOS], speaker=t_id % 2,
                                   meta={'emotion': emotions[t_id].decode(), 'act': acts[t_id].decode()}))
            if include_eod:
                dialog.append(Pack(utt=list(self.eod_utt), speaker=0))
            dialogs.append(dialog)
        return self._finish(dialogs)


class StanfordCorpus(_DialogCorpus):
    def __init__(self, config):
        self.config = config
        self._path = config.data_dir[0]
        self.max_utt_len = config.max_utt_len
        self.tokenize = get_chat_tokenize()
        self.train_corpus = self._read_file(os.path.join(self._path, 'kvret_train_public.json'))
        self.valid_corpus = self._read_file(os.path.join(self._path, 'kvret_dev_public.json'))
        self.test_corpus = self._read_file(os.path.join(self._path, 'kvret_test_public.json'))
        self._build_vocab(config.max_vocab_cnt)
        print("<d> index %d" % self.rev_vocab[BOD])
        print("Done loading corpus")

    def _read_file(self, path):
        with open(path, 'r', encoding='utf-8') as f:
            return self._process_dialog(json.load(f))

    def _process_dialog(self, data):
        include_eod = bool(getattr(self.config, 'include_eod', False))
        speaker_map = {'assistant': SYS, 'driver': USR}
        dialogs = []
        for raw_dialog in data:
            dialog = [Pack(utt=list(self.bod_utt), speaker=0, meta=None)]
            for turn in raw_dialog['dialogue']:
                words = self.tokenize(turn['data']['utterance'])
                dialog.append(Pack(utt=[BOS, speaker_map[turn['turn']]] + words + [EOS], speaker=turn['turn']))
            if include_eod:
                dialog.append(Pack(utt=list(self.eod_utt), speaker=0))
            dialogs.append(dialog)
        return self._finish(dialogs)
