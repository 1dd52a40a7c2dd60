#!/usr/bin/env python3
"""Golden vectors for gen_type='beam' decoding, made by RUNNING THE REFERENCE DecoderRNN (laed/enc2dec/decoders.py:
194-202 repeated initial state, 222-235 candidate ranking, 258-268 free-running loop) on CPU for as far as it runs.

Under torch >= 0.4 the reference's search LOOP runs to the end -- every step's ranking of cum_sum + log-probs, the symbols
it feeds back, the hidden states it does NOT reorder by the back pointers -- and only the back-tracking after the loop
dies (decoders.py:289: torch.cat of 0-d tensors; `top_id.div` is a true division there, but the back pointers are not
used inside the loop).  So the harness wraps the instance's forward_step to record what the loop fed and what came back
at every step, lets the call fail in the back-tracking, and commits the recorded trajectory:

  fed   [L, B*beam]      ids fed at each step (step 0: <s>; step t > 0: the symbols the reference chose at step t-1)
  logp  [L, B*beam, V]   the log-probs the reference's decoder returned at each step
  gap   [L, B]           score gap between the beam-th and (beam+1)-th candidate of every ranking (a near-tie would make
                         the trajectory depend on summation order)

The back-tracked 1-best sequence itself cannot be produced by the reference; oracle.decoder_beam restates it with the
integer semantics of the torch the reference targets.

usage:  python tests/golden/make_goldens_beam.py
"""
import os

import numpy as np

import make_goldens as mg          # noqa: F401  (reference import path + nltk stand-in)
import torch
from laed.enc2dec.decoders import GEN, DecoderRNN

HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == '__main__':
    G = {'meta/torch_version': np.array(torch.__version__)}
    cases = [('a', 41, 31, 8, 6, 12, 4, 3), ('b', 42, 57, 10, 8, 16, 5, 4)]
    while cases:
        tag, seed, V, L, E, H, B, K = cases.pop(0)
        rs = np.random.RandomState(seed)
        torch.manual_seed(seed)
        dec = DecoderRNN(V, L, E, H, 2, 3, n_layers=1, rnn_cell='gru', use_gpu=False)
        dec.eval()
        for n_, p_ in dec.state_dict().items():
            G['%s/sd/%s' % (tag, n_)] = mg.f32(p_)
        h0 = (rs.randn(1, B, H) * 1.5).astype(np.float32)
        fed, outs = [], []
        inner = dec.forward_step

        def recording_step(input_var, hidden, encoder_outputs, _inner=inner):
            fed.append(input_var.detach().clone().view(-1).numpy().astype(np.int64))
            out, hid, attn = _inner(input_var, hidden, encoder_outputs)
            outs.append(out.detach().clone().squeeze(1).numpy().astype(np.float32))
            return out, hid, attn
        dec.forward_step = recording_step
        died = None
        try:
            with torch.no_grad():
                dec(B, None, torch.from_numpy(h0), mode=GEN, gen_type='beam', beam_size=K)
        except Exception as e:          # the back-tracking after the loop (decoders.py:270-291)
            died = "%s: %s" % (type(e).__name__, e)
        assert died is not None and len(fed) == L and len(outs) == L, (died, len(fed))
        fed, logp = np.stack(fed), np.stack(outs)
        # gap between the last kept and the first dropped candidate of every ranking (recomputed from the recorded log-probs)
        gap = np.zeros((L, B), dtype=np.float32)
        cum = None
        for t in range(L):
            score = logp[t].astype(np.float64).reshape(B, K * V)[:, :V] if t == 0 else \
                (cum + logp[t].astype(np.float64)).reshape(B, K * V)
            order = np.sort(score, axis=1)[:, ::-1]
            gap[t] = (order[:, K - 1] - order[:, K]).astype(np.float32)
            top = np.argsort(-score, axis=1, kind='stable')[:, :K]
            cum = np.take_along_axis(score, top, axis=1).reshape(-1, 1)
            if t + 1 < L:
                assert np.array_equal(top.reshape(-1) % V, fed[t + 1]), "harness ranking disagrees with the reference's"
        if gap.min() < 2e-3:            # a near-tie in some ranking: draw other weights (the seed that is kept is recorded)
            cases.insert(0, (tag, seed + 100, V, L, E, H, B, K))
            continue
        G['%s/seed' % tag] = np.array(seed)
        G['%s/h0' % tag], G['%s/fed' % tag], G['%s/logp' % tag], G['%s/gap' % tag] = h0, fed, logp, gap
        G['%s/dims' % tag] = np.array([V, L, E, H, B, K], dtype=np.int64)
        G['%s/died' % tag] = np.array(died)
        print(tag, 'reference loop ran', L, 'steps; back-tracking died with:', died[:70], '; min gap %.4f' % gap.min())
    np.savez_compressed(os.path.join(HERE, 'beam.npz'), **G)
    print('wrote beam.npz %.1f KB' % (os.path.getsize(os.path.join(HERE, 'beam.npz')) / 1024.0))
