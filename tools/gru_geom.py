import sys, os
sys.path.insert(0, "/root/repo/neuraldialog-laed_amd"); sys.path.insert(0, "/root/repo/tools")
import torch
from laed_amd import ops
from bench_kernels import timeit
H, E = 1024, 200
for B in (256, 512, 1024, 2048, 4096):
    T = 8
    x = torch.randn(T, B, E, device="cuda")
    w_ih, w_hh = torch.randn(3 * H, E, device="cuda") * 0.03, torch.randn(3 * H, H, device="cuda") * 0.03
    b = torch.zeros(3 * H, device="cuda")
    ms = timeit(lambda: ops.GRUFunction.apply(x, w_ih, w_hh, b, b, None), iters=10)
    print("WM=%s B=%d: %.1f us/step incl gi" % (os.environ.get("LAED_GRU_WM", "auto"), B, ms * 1e3 / T), flush=True)
