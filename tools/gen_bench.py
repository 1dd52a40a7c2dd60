"""Greedy generation (DecoderRNN GEN/greedy, 40 dependent steps) at the scaled shape: ms per decode
and achieved TFLOP/s of the per-step projection.  usage: python tools/gen_bench.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "neuraldialog-laed_amd"))
import numpy as np
import torch
import bench
from laed_amd.enc2dec.decoders import GEN
from laed_amd.models.sent_models import DiVAE

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
V, E, H, y, k, T, _ = bench.SHAPES["S-fixed"]
cfg = bench.make_config(E, H, y, k, T, B)
torch.manual_seed(0)
model = DiVAE(bench.Corpus(V), cfg).cuda().eval()
feed = {"outputs": torch.from_numpy(bench.synth_batch(V, T, B).astype(np.int64)).cuda(), "output_lens": [T] * B}
with torch.no_grad():
    for _ in range(2):
        model(feed, mode=GEN, gen_type="greedy")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        model(feed, mode=GEN, gen_type="greedy")
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
steps = cfg.max_dec_len
fl = steps * 2.0 * B * (V * H + 3 * H * (E + H)) + 2.0 * B * T * 3 * H * (E + H)
print("greedy GEN B=%d: %.1f ms per batch (%d steps), %.0f sequences/s, %.1f TFLOP/s" % (B, ms, steps, B / ms * 1e3, fl / ms / 1e9))
