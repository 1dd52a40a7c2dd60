#!/usr/bin/env python3
"""Golden vectors for gen_type='sample' decoding (Gumbel-max, laed/enc2dec/base_modules.py:54-67 +
decoders.py:220-221), made by RUNNING THE REFERENCE DecoderRNN on CPU.

The reference draws its noise with torch.Tensor(size).uniform_(0, 1) and seeds nothing, so the harness swaps
torch.Tensor.uniform_ for a replay of a recorded uniform stream u[step] while the decoder runs (harness side only);
gumbel_max itself -- the -log(-log(u)) transform, the add to the log-probs and the argmax -- is the reference's.

usage:  python tests/golden/make_goldens_sample.py
"""
import os

import numpy as np

import make_goldens as mg          # noqa: F401  (reference import path + nltk stand-in)
import torch
from laed.enc2dec.decoders import GEN, DecoderRNN

HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == '__main__':
    rs = np.random.RandomState(31)
    torch.manual_seed(31)
    G = {'meta/torch_version': np.array(torch.__version__)}
    V, L, E, H, B = 29, 7, 6, 12, 5
    dec = DecoderRNN(V, L, E, H, 2, 3, n_layers=1, rnn_cell='gru', use_gpu=False)
    dec.eval()
    for n_, p_ in dec.state_dict().items():
        G['sd/' + n_] = mg.f32(p_)
    h0 = rs.randn(1, B, H).astype(np.float32)
    u = rs.uniform(size=(L, B, V)).astype(np.float32)
    G['h0'], G['u'] = h0, u
    calls = []
    orig = torch.Tensor.uniform_

    def replay(self, a=0, b=1):
        self.copy_(torch.from_numpy(u[len(calls)]))
        calls.append(1)
        return self
    torch.Tensor.uniform_ = replay
    try:
        with torch.no_grad():
            outs, hid, ret = dec(B, None, torch.from_numpy(h0), mode=GEN, gen_type='sample')
    finally:
        torch.Tensor.uniform_ = orig
    assert len(calls) == L
    seq = torch.cat(ret[DecoderRNN.KEY_SEQUENCE], dim=1)
    g = -np.log(-np.log(u.astype(np.float64)))
    score = outs.numpy().astype(np.float64).transpose(1, 0, 2) + g            # [L,B,V]
    top2 = np.sort(score, axis=2)[:, :, -2:]
    G['sequence'] = seq.numpy().astype(np.int64)
    G['length'] = np.array(ret[DecoderRNN.KEY_LENGTH], dtype=np.int64)
    G['margin'] = (top2[:, :, 1] - top2[:, :, 0]).astype(np.float32)
    G['hid'] = mg.f32(hid)
    np.savez_compressed(os.path.join(HERE, 'sample.npz'), **G)
    print('wrote sample.npz; min margin %.4f; seq[0] %s' % (G['margin'].min(), G['sequence'][0]))
