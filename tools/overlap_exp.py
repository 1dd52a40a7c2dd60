"""Does the HBM-bound softmax-CE row pass hide behind an MFMA-bound GEMM when the two run on two streams?
(feasibility check for pipelining the projection chunks: rows(c) beside the logits GEMM of chunk c+1)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "neuraldialog-laed_amd"))
import torch
from laed_amd import ops

R, V, H = 16384, 50000, 1024
ldl = ops.proj_ldl(V)
h = torch.randn(R, H, device="cuda")
W = torch.randn(V, H, device="cuda") * 0.03
b = torch.zeros(V, device="cuda")
buf0 = torch.empty(R, ldl, device="cuda")
buf1 = torch.randn(R, ldl, device="cuda")
dl = torch.randn(R, ldl, device="cuda")
dh = torch.empty(R, H, device="cuda")
side = torch.cuda.Stream(priority=int(os.environ.get("SIDE_PRIO", "0")))


def gemm():
    ops.sgemm(h, W, transB=True, bias=b, out=buf0[:, :V])


def gemm2():
    ops.sgemm(dl[:, :V], W, out=dh)


def rows():
    ops.log_softmax_rows_(buf1, V)


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / n


def both(f_main, f_side):
    def run():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            f_side()
        f_main()
        torch.cuda.current_stream().wait_stream(side)
    return run


tg, tg2, tr = timed(gemm), timed(gemm2), timed(rows)
print("logits GEMM %.3f ms   dh GEMM %.3f ms   rows %.3f ms" % (tg, tg2, tr))
print("GEMM || rows      : %.3f ms (serial %.3f)" % (timed(both(gemm, rows)), tg + tr))
print("GEMM || dh GEMM   : %.3f ms (serial %.3f)" % (timed(both(gemm, gemm2)), tg + tg2))
