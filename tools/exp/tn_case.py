"""One TN (A^T B) product for PMC passes.  usage: tn_case.py M N K lda [plain|full]   (lda >= M: row stride of A[K][lda])"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "neuraldialog-laed_amd"))
import torch
from laed_amd import ops
M, N, K, lda = (int(x) for x in sys.argv[1:5])
full = len(sys.argv) > 5 and sys.argv[5] == "full"
Abuf = torch.randn((K, lda), device="cuda")
A = Abuf[:, :M]
B = torch.randn((K, N), device="cuda")
C = torch.zeros((M, N), device="cuda")
cs = torch.zeros(M, device="cuda")
for _ in range(2):
    if full:
        ops.sgemm(A, B, transA=True, out=C, accumulate=True, colsum_out=cs, colsum_accumulate=True, splitk_ws=False)
    else:
        ops.sgemm(A, B, transA=True, out=C, splitk_ws=False)
torch.cuda.synchronize()
