"""Batch loaders (laed/dataset/data_loaders.py, Python 3): id corpora -> the numpy batch dicts the models consume.

  PTBDataLoader          lines            -> outputs [B,Tmax] int32, output_lens                      (DI-VAE)
  SMDDataLoader          dialogs          -> contexts [B,Cmax,max_utt_len], context_lens, outputs ...  (AE-ED)
  SMDDialogSkipLoader / DailyDialogSkipLoader
                         dialogs          -> the same + prevs / nexts (+ lens)                         (DI-VST, ST-ED)

The three dialog loaders of the reference are one class here with a `skip` switch: an example is (up to
backward_size previous turns, response[, previous turn, next turn]); skip loaders need a next turn, so the last turn
of a dialog is never a response there (data_loaders.py:92,185).  Every utterance is clipped to max_utt_len keeping
its last token (pad_to(do_pad=False)); batches are zero-padded to the longest row (contexts always to max_utt_len).
tests/test_dataset_cpu.py checks batches against ones produced by running the reference loaders.
"""
import numpy as np

from ..utils import Pack
from .dataloader_bases import DataLoader


def _rows_to_matrix(rows, width=None):
    """Ragged list of id lists -> zero-padded int32 [len(rows), width or longest], plus the lengths."""
    lens = np.array([len(r) for r in rows])
    out = np.zeros((len(rows), int(width if width is not None else lens.max())), dtype=np.int32)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out, lens


class _DialogLoader(DataLoader):
    skip = False

    def __init__(self, name, data, config):
        super(_DialogLoader, self).__init__(name, fix_batch=config.fix_batch)
        self.max_utt_size = config.max_utt_len
        self.data = self.flatten_dialog(data, config.backward_size)
        self.data_size = len(self.data)
        if config.fix_batch:          # longest contexts first, batches of similar context length
            self.indexes = list(np.argsort([len(d.context) for d in self.data]))[::-1]
        else:
            self.indexes = list(range(self.data_size))

    def _clip(self, turn):
        turn['utt'] = self.pad_to(self.max_utt_size, turn.utt, do_pad=False)
        return turn

    def flatten_dialog(self, data, backward_size):
        examples = []
        for dialog in data:
            last = len(dialog) - 1 if self.skip else len(dialog)
            for i in range(1, last):
                context = [self._clip(t) for t in dialog[max(0, i - backward_size):i]]
                if self.skip:
                    examples.append(Pack(context=context, response=self._clip(dialog[i]),
                                         prev_resp=self._clip(dialog[i - 1]), next_resp=self._clip(dialog[i + 1])))
                else:
                    examples.append(Pack(context=context, response=self._clip(dialog[i].copy())))
        return examples

    def _prepare_batch(self, selected_index):
        rows = [self.data[i] for i in selected_index]
        ctx_lens = np.array([len(r.context) for r in rows])
        contexts = np.zeros((self.batch_size, int(ctx_lens.max()), self.max_utt_size), dtype=np.int32)
        for b, r in enumerate(rows):
            for c, turn in enumerate(r.context):
                contexts[b, c, :len(turn.utt)] = turn.utt
        outs, out_lens = _rows_to_matrix([r.response.utt for r in rows])
        batch = Pack(contexts=contexts, context_lens=ctx_lens, outputs=outs, output_lens=out_lens,
                     metas=[r.response.meta for r in rows])
        if self.skip:
            batch['prevs'], batch['prev_lens'] = _rows_to_matrix([r.prev_resp.utt for r in rows])
            batch['nexts'], batch['next_lens'] = _rows_to_matrix([r.next_resp.utt for r in rows])
        return batch


class SMDDataLoader(_DialogLoader):
    skip = False


class SMDDialogSkipLoader(_DialogLoader):
    skip = True


class DailyDialogSkipLoader(_DialogLoader):
    skip = True


class PTBDataLoader(DataLoader):
    def __init__(self, name, data, config):
        super(PTBDataLoader, self).__init__(name, fix_batch=config.fix_batch)
        self.max_utt_size = config.max_utt_len
        self.data = self.pad_data(data)
        self.data_size = len(self.data)
        all_lens = [len(line.utt) for line in self.data]
        print("Max len %d and min len %d and avg len %f" % (np.max(all_lens), np.min(all_lens), float(np.mean(all_lens))))
        self.indexes = list(np.argsort(all_lens)) if config.fix_batch else list(range(self.data_size))

    def pad_data(self, data):
        for line in data:
            line['utt'] = self.pad_to(self.max_utt_size, line.utt, do_pad=False)
        return data

    def _prepare_batch(self, selected_index):
        rows = [self.data[i] for i in selected_index]
        inputs, lens = _rows_to_matrix([r.utt for r in rows])
        return Pack(outputs=inputs, output_lens=lens.astype(np.int32), metas=[None] * len(rows))
