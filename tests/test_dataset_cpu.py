"""Data layer (laed_amd/dataset): loaders against batches produced by RUNNING the reference loaders
(tests/golden/make_goldens_loaders.py), corpus readers against what the reference's reader methods made of the same
text (tests/golden/make_goldens_corpora.py) and against their documented rules on synthetic files, and a
real-text pass over the DailyDialog / Stanford splits that ship with the reference when that tree is present
(build container only; skipped elsewhere).  CPU only, no kernels."""
import json
import os
from argparse import Namespace

import numpy as np
import pytest

from helpers import GOLDEN, load_golden


def _packs(data):
    from laed_amd.utils import Pack
    if isinstance(data[0], dict):
        return [Pack(utt=list(d['utt']), speaker=d['speaker'], meta=d.get('meta')) for d in data]
    return [[Pack(utt=list(t['utt']), speaker=t['speaker'], meta=t.get('meta')) for t in dlg] for dlg in data]


def test_loaders_reproduce_reference_batches():
    from laed_amd.dataset import data_loaders
    G = load_golden("loaders")
    J = json.load(open(os.path.join(GOLDEN, "loaders_inputs.json")))
    for tag, case in J["cases"].items():
        config = Namespace(**case["config"])
        loader = getattr(data_loaders, case["cls"])(tag, _packs(J["corpora"][case["src"]]), config)
        assert loader.data_size == case["data_size"], tag
        loader.epoch_init(config, shuffle=False, verbose=False)
        assert loader.num_batch == case["num_batch"], tag
        for i in range(case["num_batch"]):
            b = loader.next_batch()
            keys = sorted(k.split("/")[2] for k in G.files if k.startswith("%s/%d/" % (tag, i)))
            assert sorted(k for k in b.keys() if k != "metas") == keys, (tag, i)
            for k in keys:
                ref = G["%s/%d/%s" % (tag, i, k)]
                assert np.array_equal(np.asarray(b[k]), ref), (tag, i, k)
                if ref.ndim > 1:
                    assert np.asarray(b[k]).dtype == np.int32, (tag, i, k)
            assert b.metas == case["metas"][i], (tag, i)
        assert loader.next_batch() is None


def test_loader_shuffle_keeps_examples_and_fix_batch_groups():
    from laed_amd.dataset import data_loaders
    J = json.load(open(os.path.join(GOLDEN, "loaders_inputs.json")))
    np.random.seed(0)
    for fix in (False, True):
        config = Namespace(max_utt_len=10, batch_size=4, backward_size=3, fix_batch=fix)
        loader = data_loaders.SMDDialogSkipLoader("t", _packs(J["corpora"]["dialogs"]), config)
        loader.epoch_init(config, shuffle=False, verbose=False)
        plain = [tuple(b) for b in loader.batch_indexes]
        loader.epoch_init(config, shuffle=True, verbose=False)          # the reference cannot do this under Python 3
        shuf = [tuple(b) for b in loader.batch_indexes]
        if fix:
            assert sorted(shuf) == sorted(plain) and shuf != plain      # batches intact, order shuffled
        else:
            assert len({i for b in shuf for i in b}) == 4 * len(shuf) and shuf != plain
        assert loader.pad_to(5, [2, 7, 8, 9, 10, 11, 3], do_pad=False) == [2, 7, 8, 9, 3]
        assert loader.pad_to(5, [2, 3]) == [2, 3, 0, 0, 0]


def test_corpus_readers_follow_the_reference_rules(tmp_path):
    from laed_amd.dataset import corpora
    # ---- PTB: first token dropped, <s> ... </s>, vocab = [<pad>, <unk>] + by count
    ptb = tmp_path / "ptb"
    ptb.mkdir()
    (ptb / "ptb.train.txt").write_text(" aa bb cc bb\n xx bb dd's <unk> N\n")
    (ptb / "ptb.valid.txt").write_text(" zz bb\n")
    (ptb / "ptb.test.txt").write_text(" aa\n")
    c = corpora.PTBCorpus(Namespace(data_dir=[str(ptb)], max_utt_len=40, max_vocab_cnt=6))
    assert [t.utt for t in c.train_corpus] == [['<s>', 'bb', 'cc', 'bb', '</s>'],
                                               ['<s>', 'bb', 'dd', "'", 's', '<', 'unk', '>', 'N', '</s>']]
    assert c.vocab[:2] == ['<pad>', '<unk>'] and c.vocab[2] == 'bb' and len(c.vocab) == 8
    ids = c.get_corpus()
    assert ids.valid[0].utt == [c.rev_vocab['<s>'], c.rev_vocab['bb'], c.rev_vocab['</s>']]      # 'zz' was dropped
    assert ids.test[0].utt == [c.rev_vocab['<s>'], c.rev_vocab['</s>']]
    assert c._sent2id(['never-seen']) == [1]

    # ---- DailyDialog: __eou__ split, lower-case, non-ASCII dropped, <d> / </d> turns, act / emotion meta
    dd = tmp_path / "dd"
    for split in ("train", "validation", "test"):
        d = dd / split
        d.mkdir(parents=True)
        (d / "dialogues.txt").write_bytes("Hello there , café ! __eou__ Fine . __eou__\nBye __eou__\n".encode("utf-8"))
        (d / "dialogues_act.txt").write_text("1 2\n3\n")
        (d / "dialogues_emotion.txt").write_text("0 4\n0\n")
    c = corpora.DailyDialogCorpus(Namespace(data_dir=[str(dd)], max_utt_len=40, max_vocab_cnt=100))
    d0 = c.train_corpus[0]
    assert [t.utt for t in d0] == [['<s>', '<d>', '</s>'], ['<s>', 'hello', 'there', ',', 'caf', '!', '</s>'],
                                   ['<s>', 'fine', '.', '</s>'], ['<s>', '</d>', '</s>']]
    assert [t.speaker for t in d0] == [0, 0, 1, 0] and d0[2].meta == {'emotion': '4', 'act': '2'}
    assert c.vocab[:4] == ['<pad>', '<unk>', '<sys>', '<usr>'] and c.vocab[4] in ('<s>', '</s>')
    c2 = corpora.DailyDialogCorpus(Namespace(data_dir=[str(dd)], max_utt_len=40, max_vocab_cnt=100, include_eod=False))
    assert len(c2.train_corpus[0]) == 3

    # ---- Stanford: speaker tags, </d> only when include_eod is set
    st = tmp_path / "st"
    st.mkdir()
    dlg = [{"dialogue": [{"turn": "driver", "data": {"utterance": "Where's gas?"}},
                         {"turn": "assistant", "data": {"utterance": "2 miles away."}}]}]
    for name in ("kvret_train_public.json", "kvret_dev_public.json", "kvret_test_public.json"):
        (st / name).write_text(json.dumps(dlg))
    c = corpora.StanfordCorpus(Namespace(data_dir=[str(st)], max_utt_len=40, max_vocab_cnt=100))
    assert [t.utt for t in c.train_corpus[0]] == [['<s>', '<d>', '</s>'],
                                                  ['<s>', '<usr>', 'Where', "'", 's', 'gas', '?', '</s>'],
                                                  ['<s>', '<sys>', '2', 'miles', 'away', '.', '</s>']]
    c = corpora.StanfordCorpus(Namespace(data_dir=[str(st)], max_utt_len=40, max_vocab_cnt=100, include_eod=True))
    assert c.train_corpus[0][-1].utt == ['<s>', '</d>', '</s>']
    ids = c.get_corpus().train[0]
    # quirk kept from corpora.py:101-102: <sys>/<usr> are listed as specials AND counted as words, so they appear twice in
    # `vocab` and rev_vocab points at the later (counted) slot; ids 2 and 3 are never produced
    assert c.vocab.count('<usr>') == 2 and c.rev_vocab['<usr>'] > 3
    assert ids[1].utt[:2] == [c.rev_vocab['<s>'], c.rev_vocab['<usr>']] and ids[1].speaker == 'driver'


REF_DATA = "/root/reference/data"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF_DATA, "daily_dialog", "validation")),
                    reason="the reference's data tree only exists in the build container")
def test_real_text_daily_dialog_and_stanford_through_the_loaders(tmp_path):
    """Real text end to end on the host side: DailyDialog validation/test (1000 dialogs each) and the Stanford
    dev/test JSONs ship with the reference (the train splits do not), so `train` points at another present split."""
    from laed_amd.dataset import corpora, data_loaders
    dd = tmp_path / "dd"
    dd.mkdir()
    for split, src in (("train", "validation"), ("validation", "validation"), ("test", "test")):
        os.symlink(os.path.join(REF_DATA, "daily_dialog", src), str(dd / split))
    cfg = Namespace(data_dir=[str(dd)], max_utt_len=40, max_vocab_cnt=10000, backward_size=10, fix_batch=False,
                    batch_size=60)
    c = corpora.DailyDialogCorpus(cfg)
    assert len(c.train_corpus) == 1000 and len(c.test_corpus) == 1000
    assert c.vocab[:4] == ['<pad>', '<unk>', '<sys>', '<usr>'] and 3000 < len(c.vocab) <= 10004
    ids = c.get_corpus()
    feed = data_loaders.DailyDialogSkipLoader("Train", ids.train, cfg)
    feed.epoch_init(cfg, shuffle=True, verbose=False)
    b = feed.next_batch()
    assert b.outputs.shape[0] == 60 and b.outputs.shape[1] <= 40 and b.contexts.shape[2] == 40
    assert b.contexts.shape[1] == b.context_lens.max() <= 10
    assert (b.outputs[:, 0] == c.rev_vocab['<s>']).all() and b.prevs.dtype == np.int32
    for r in range(60):
        assert b.outputs[r, b.output_lens[r] - 1] == c.rev_vocab['</s>'] and (b.outputs[r, b.output_lens[r]:] == 0).all()

    st = tmp_path / "st"
    st.mkdir()
    for name, src in (("kvret_train_public.json", "kvret_dev_public.json"), ("kvret_dev_public.json", "kvret_dev_public.json"),
                      ("kvret_test_public.json", "kvret_test_public.json")):
        os.symlink(os.path.join(REF_DATA, "stanford", src), str(st / name))
    cfg = Namespace(data_dir=[str(st)], max_utt_len=40, max_vocab_cnt=10000, backward_size=30, fix_batch=False,
                    batch_size=30, include_eod=True)
    c = corpora.StanfordCorpus(cfg)
    ids = c.get_corpus()
    for cls in (data_loaders.SMDDataLoader, data_loaders.SMDDialogSkipLoader):
        feed = cls("Valid", ids.valid, cfg)
        feed.epoch_init(cfg, shuffle=False, verbose=False)
        b = feed.next_batch()
        assert b.contexts.shape[0] == 30 and b.outputs[:, 1].tolist().count(0) == 0
        assert set(np.unique(b.outputs[:, 1])) <= {c.rev_vocab['<sys>'], c.rev_vocab['<usr>'], c.rev_vocab['<d>'],
                                                    c.rev_vocab['</d>']}


def _flatten(id_split, dialog):
    dialogs = id_split if dialog else [[t] for t in id_split]
    ids = [i for d in dialogs for t in d for i in t.utt]
    ulen = [len(t.utt) for d in dialogs for t in d]
    spk = [str(t.speaker) for d in dialogs for t in d]
    meta = ["" if t.get('meta') is None else "%s/%s" % (t.meta['act'], t.meta['emotion']) for d in dialogs for t in d]
    return ids, ulen, [len(d) for d in dialogs], spk, meta


def _check_against_record(G, name, c, dialog):
    assert c.vocab == G[name + "/vocab"].tolist(), name
    assert c.unk_id == int(G[name + "/unk_id"])
    assert [c.rev_vocab[t] for t in c.vocab[:4]] == G[name + "/rev_vocab_of_specials"].tolist()
    ids = c.get_corpus()
    for split in ("train", "valid", "test"):
        got = _flatten(ids[split], dialog)
        for k, v in zip(("ids", "utt_lens", "dialog_lens", "speakers", "metas"), got):
            assert list(v) == G["%s/%s/%s" % (name, split, k)].tolist(), (name, split, k)


def test_corpus_readers_reproduce_what_the_reference_readers_made():
    """tests/golden/corpora.npz holds what the REFERENCE's reader methods (corpora.py:53-126, :149-198, :229-307, run
    unmodified by tests/golden/make_goldens_corpora.py) made of real Stanford / DailyDialog text that ships with the
    reference and of PTB-format lines: token lists -> count-ordered, cut vocabulary (with the doubled <sys>/<usr> slots)
    -> id corpora, speakers and act / emotion metas.  This repo's readers must give the same, id for id.  The tokeniser's
    pattern was the reference's, its engine `re` standing in for nltk (absent): that one step is restated on both sides."""
    from laed_amd.dataset import corpora
    G = load_golden("corpora")
    for name in ("stanford", "stanford_noeod_fullvocab"):
        inputs = json.loads(str(G[name + "/inputs_json"]))
        c = corpora.StanfordCorpus.__new__(corpora.StanfordCorpus)
        c.config = Namespace(include_eod=True) if bool(G[name + "/include_eod"]) else Namespace()
        c.tokenize = corpora.get_chat_tokenize()
        c.train_corpus, c.valid_corpus, c.test_corpus = (c._process_dialog(inputs[s]) for s in ("train", "valid", "test"))
        c._build_vocab(int(G[name + "/max_vocab_cnt"]))
        _check_against_record(G, name, c, True)
    for name in ("daily_dialog", "daily_dialog_noeod_fullvocab"):
        assert int(G[name + "/non_ascii_bytes_in_inputs"]) > 0          # the ascii-ignore step has something to drop
        c = corpora.DailyDialogCorpus.__new__(corpora.DailyDialogCorpus)
        eod = int(G[name + "/include_eod"])
        c.config = Namespace() if eod < 0 else Namespace(include_eod=bool(eod))
        c.tokenize = corpora.get_chat_tokenize()
        for s in ("train", "valid", "test"):
            txt, act, emo = (G["%s/inputs/%s/%s" % (name, s, k)].tobytes().splitlines(True) for k in ("txt", "act", "emo"))
            setattr(c, s + "_corpus", c._process_dialog(zip(txt, act, emo)))
        c._build_vocab(int(G[name + "/max_vocab_cnt"]))
        _check_against_record(G, name, c, True)
    inputs = json.loads(str(G["ptb/inputs_json"]))
    c = corpora.PTBCorpus.__new__(corpora.PTBCorpus)
    c.config = Namespace()
    c.tokenize = corpora.get_chat_tokenize()
    c.train_corpus, c.valid_corpus, c.test_corpus = (c._process_data(inputs[s]) for s in ("train", "valid", "test"))
    c._build_vocab(int(G["ptb/max_vocab_cnt"]))
    _check_against_record(G, "ptb", c, False)
    tok = corpora.get_chat_tokenize()
    assert [tok(p) for p in G["tokenize/probes"].tolist()] == json.loads(str(G["tokenize/tokens_json"]))


def test_corpus_readers_from_files_equal_the_method_path(tmp_path):
    """The constructors (file reading: 'rb' lines for DailyDialog, utf-8 JSON for Stanford, text lines for PTB) give the
    same corpora as the method path the record above pins, on the same inputs written back to files."""
    from laed_amd.dataset import corpora
    G = load_golden("corpora")
    dd = tmp_path / "dd"
    for split, s in (("train", "train"), ("validation", "valid"), ("test", "test")):
        (dd / split).mkdir(parents=True)
        for fname, k in (("dialogues.txt", "txt"), ("dialogues_act.txt", "act"), ("dialogues_emotion.txt", "emo")):
            (dd / split / fname).write_bytes(G["daily_dialog/inputs/%s/%s" % (s, k)].tobytes())
    c = corpora.DailyDialogCorpus(Namespace(data_dir=[str(dd)], max_utt_len=40,
                                            max_vocab_cnt=int(G["daily_dialog/max_vocab_cnt"])))
    _check_against_record(G, "daily_dialog", c, True)
    st = tmp_path / "st"
    st.mkdir()
    inputs = json.loads(str(G["stanford/inputs_json"]))
    for fname, s in (("kvret_train_public.json", "train"), ("kvret_dev_public.json", "valid"),
                     ("kvret_test_public.json", "test")):
        (st / fname).write_text(json.dumps(inputs[s]), encoding="utf-8")
    c = corpora.StanfordCorpus(Namespace(data_dir=[str(st)], max_utt_len=40, include_eod=True,
                                         max_vocab_cnt=int(G["stanford/max_vocab_cnt"])))
    _check_against_record(G, "stanford", c, True)
    ptb = tmp_path / "ptb"
    ptb.mkdir()
    inputs = json.loads(str(G["ptb/inputs_json"]))
    for s in ("train", "valid", "test"):
        (ptb / ("ptb.%s.txt" % s)).write_text("".join(inputs[s]), encoding="utf-8")
    c = corpora.PTBCorpus(Namespace(data_dir=[str(ptb)], max_utt_len=40, max_vocab_cnt=int(G["ptb/max_vocab_cnt"])))
    _check_against_record(G, "ptb", c, False)
