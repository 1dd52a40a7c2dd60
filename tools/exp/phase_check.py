"""Phase-locked GEMM launches give the same bits as unlocked ones (LAED_GEMM_PHASE on/off via two processes is not needed:
the lock only adds waits).  usage: LAED_HIP_LIB=.../liblaed_hip_hooks.so LAED_GEMM_PHASE=8 python tools/exp/phase_check.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "neuraldialog-laed_amd"))
import torch
from laed_amd import ops
torch.manual_seed(0)
for lay, M, N, K in (("TN", 16384, 1024, 2048), ("NN", 16384, 1024, 2048), ("NT", 4096, 4096, 1024)):
    tA, tB = lay[0] == "T", lay[1] == "T"
    A = torch.randn((K, M) if tA else (M, K), device="cuda")
    B = torch.randn((N, K) if tB else (K, N), device="cuda")
    C = ops.sgemm(A, B, transA=tA, transB=tB, splitk_ws=False)
    ref = ((A.t() if tA else A).double() @ (B.t() if tB else B).double())
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    C2 = ops.sgemm(A, B, transA=tA, transB=tB, splitk_ws=False)
    print(lay, M, N, K, "rel err %.2e" % err, "repeatable", bool(torch.equal(C, C2)), "sum %.10e" % float(C.double().sum()))
