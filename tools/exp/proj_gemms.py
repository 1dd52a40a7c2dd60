"""The projection's two big backward products at the headline chunk shape, a few launches each (for rocprofv3 --pmc):
dW = dl^T h (TN, 50000 x 1024 x 40960, with workspace: whole-rounds launch + split-K tail) and dh = dl W (NN, 40960 x 1024 x
50000).  usage: python tools/exp/proj_gemms.py [launches]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "neuraldialog-laed_amd"))
import torch
from laed_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R, V, H = 40960, 50000, 1024
dl = torch.randn((R, V), device="cuda")
h = torch.randn((R, H), device="cuda")
W = torch.randn((V, H), device="cuda")
dW = torch.zeros((V, H), device="cuda")
db = torch.zeros(V, device="cuda")
dh = torch.empty((R, H), device="cuda")
for _ in range(n):
    ops.sgemm(dl, h, transA=True, out=dW, accumulate=True, colsum_out=db, colsum_accumulate=True, splitk_ws=False,
              ws_floats=8 * 8192 * 1024)
    ops.sgemm(dl, W, out=dh, splitk_ws=False)
torch.cuda.synchronize()
