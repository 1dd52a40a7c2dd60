#!/usr/bin/env python3
"""Golden batches from the reference's data loaders (laed/dataset/data_loaders.py), made by RUNNING them on CPU.

The loaders run under Python 3 as long as nothing asks them to shuffle (np.random.shuffle(range) fails), so they are
driven with epoch_init(shuffle=False) over small synthetic id corpora in the format corpus.get_corpus() returns
(corpora.py:109-126,185-198).  Inputs (the id corpora, as JSON) and the batches they produce are committed; the
corpus READERS do not run here (Python-2 bytes/str handling + nltk), see laed_amd/dataset/corpora.py.

usage:  python tests/golden/make_goldens_loaders.py
"""
import json
import os
from argparse import Namespace

import numpy as np

import make_goldens as mg          # noqa: F401  (reference import path + nltk stand-in)
from laed.dataset import data_loaders
from laed.utils import Pack

HERE = os.path.dirname(os.path.abspath(__file__))


def synth_dialogs(rs, n, V, max_turns, max_len):
    dialogs = []
    for _ in range(n):
        turns = []
        for t in range(int(rs.randint(3, max_turns + 1))):
            L = int(rs.randint(3, max_len + 1))
            utt = [2] + [int(x) for x in rs.randint(4, V, size=L - 2)] + [3]
            turns.append(dict(utt=utt, speaker=t % 2, meta={'act': str(int(rs.randint(1, 5)))} if t else None))
        dialogs.append(turns)
    return dialogs


def synth_lines(rs, n, V, max_len):
    return [dict(utt=[2] + [int(x) for x in rs.randint(4, V, size=int(rs.randint(1, max_len)))] + [3], speaker=0)
            for _ in range(n)]


def to_packs(data):
    if isinstance(data[0], dict):
        return [Pack(utt=list(d['utt']), speaker=d['speaker'], meta=d.get('meta')) for d in data]
    return [[Pack(utt=list(t['utt']), speaker=t['speaker'], meta=t.get('meta')) for t in dlg] for dlg in data]


def drain(loader, config):
    loader.epoch_init(config, shuffle=False, verbose=False)
    out = []
    while True:
        b = loader.next_batch()
        if b is None:
            return out
        out.append(b)


if __name__ == '__main__':
    rs = np.random.RandomState(99)
    G = {}
    corpora = {'dialogs': synth_dialogs(rs, 7, 40, 9, 14), 'lines': synth_lines(rs, 23, 40, 15)}
    cases = [('smd', data_loaders.SMDDataLoader, 'dialogs', dict(backward_size=3, fix_batch=False)),
             ('smd_fix', data_loaders.SMDDataLoader, 'dialogs', dict(backward_size=5, fix_batch=True)),
             ('smd_skip', data_loaders.SMDDialogSkipLoader, 'dialogs', dict(backward_size=2, fix_batch=False)),
             ('dd_skip', data_loaders.DailyDialogSkipLoader, 'dialogs', dict(backward_size=4, fix_batch=True)),
             ('ptb', data_loaders.PTBDataLoader, 'lines', dict(fix_batch=False)),
             ('ptb_fix', data_loaders.PTBDataLoader, 'lines', dict(fix_batch=True))]
    meta = {}
    for tag, cls, src, kw in cases:
        config = Namespace(max_utt_len=10, batch_size=4, **kw)
        loader = cls(tag, to_packs(corpora[src]), config)
        batches = drain(loader, config)
        meta[tag] = dict(cls=cls.__name__, src=src, config=vars(config), num_batch=len(batches),
                         data_size=loader.data_size, metas=[b.metas for b in batches])
        for i, b in enumerate(batches):
            for k, v in b.items():
                if k != 'metas':
                    G['%s/%d/%s' % (tag, i, k)] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, 'loaders.npz'), **G)
    with open(os.path.join(HERE, 'loaders_inputs.json'), 'w') as f:
        json.dump(dict(corpora=corpora, cases=meta), f)
    print('wrote loaders.npz (%d arrays) + loaders_inputs.json' % len(G), {k: v['num_batch'] for k, v in meta.items()})
