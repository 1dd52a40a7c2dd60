"""Validation pass (engine.validate: eval mode, no_grad, TEACH_FORCE; laed/main.py:165-181) at the scaled shape:
ms per batch and TFLOP/s of the forward FLOPs.  usage: python tools/valid_bench.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "neuraldialog-laed_amd"))
import numpy as np
import torch
import bench
from laed_amd.enc2dec.decoders import TEACH_FORCE
from laed_amd.models.sent_models import DiVAE

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
V, E, H, y, k, T, _ = bench.SHAPES["S-fixed"]
cfg = bench.make_config(E, H, y, k, T, B)
torch.manual_seed(0)
model = DiVAE(bench.Corpus(V), cfg).cuda().eval()
feed = {"outputs": torch.from_numpy(bench.synth_batch(V, T, B).astype(np.int64)).cuda(), "output_lens": [T] * B}
with torch.no_grad():
    for _ in range(2):
        loss = model(feed, mode=TEACH_FORCE)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        loss = model(feed, mode=TEACH_FORCE)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
fl = bench.flops_per_step(V, E, H, y, k, T, B) / 3.0
print("validation forward B=%d: %.1f ms per batch, %.0f tokens/s, %.1f TFLOP/s, nll %.5f" %
      (B, ms, B * (T - 1) / ms * 1e3, fl / ms / 1e9, float(loss.nll)))
