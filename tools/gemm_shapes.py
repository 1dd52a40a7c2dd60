"""Time the fp32 GEMM on a list of shapes given on the command line: M,N,K,LAY[,lda] ...
(lda > the operand's width makes A a column slice of a wider matrix: stride experiments)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "neuraldialog-laed_amd"))
import torch
from laed_amd import ops

WS = "--ws" in sys.argv          # let the launcher split K (caller workspace), as the model code does
for spec in [a for a in sys.argv[1:] if a != "--ws"]:
    f = spec.split(",")
    M, N, K, lay = int(f[0]), int(f[1]), int(f[2]), f[3]
    tA, tB = lay[0] == "T", lay[1] == "T"
    wa = M if tA else K
    lda = int(f[4]) if len(f) > 4 else wa
    A = torch.randn((K if tA else M, lda), device="cuda")[:, :wa]
    B = torch.randn((N, K) if tB else (K, N), device="cuda")
    C = torch.empty((M, N), device="cuda")
    fn = lambda: ops.sgemm(A, B, transA=tA, transB=tB, out=C, splitk_ws=WS)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 5 if 2.0 * M * N * K > 1e12 else 50
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-28s %8.3f ms %6.1f TF" % (spec, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    del A, B, C
