"""A/B of the fused GRU step kernel: us per forward recurrence step (input GEMM excluded: gi is computed once outside the
timed loop by running the whole forward at two lengths and differencing) for the library named by LAED_HIP_LIB."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "neuraldialog-laed_amd"))
import torch
from laed_amd import ops

H, E = 1024, 200


def run(B, T, train):
    x = torch.randn(T, B, E, device="cuda") * 0.5
    w_ih, w_hh = torch.randn(3 * H, E, device="cuda") * 0.03, torch.randn(3 * H, H, device="cuda") * 0.03
    b = torch.zeros(3 * H, device="cuda")
    if train:
        w_hh.requires_grad_(True)
    fn = lambda: ops.GRUFunction.apply(x, w_ih, w_hh, b, b, None)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5


for B in (512, 2048, 4096):
    for train in (True, False):
        t8, t40 = run(B, 8, train), run(B, 40, train)
        print("%s B=%d %s: %.1f us per step (T=40: %.3f ms)" % (os.path.basename(os.environ.get("LAED_HIP_LIB", "product")), B,
                                                             "train" if train else "eval ", (t40 - t8) * 1e3 / 32, t40), flush=True)
