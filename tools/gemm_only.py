"""One GEMM shape, a few launches (for rocprofv3 --pmc passes).  usage: M N K [NT|NN|TN] [fp32|bf16x3]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "neuraldialog-laed_amd"))
import torch
from laed_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
lay = sys.argv[4] if len(sys.argv) > 4 else "NT"
if len(sys.argv) > 5:
    ops.set_gemm_mode(sys.argv[5])
tA, tB = lay[0] == "T", lay[1] == "T"
A = torch.randn((K, M) if tA else (M, K), device="cuda")
B = torch.randn((N, K) if tB else (K, N), device="cuda")
C = torch.empty((M, N), device="cuda")
for _ in range(4):
    ops.sgemm(A, B, transA=tA, transB=tB, out=C)
torch.cuda.synchronize()
