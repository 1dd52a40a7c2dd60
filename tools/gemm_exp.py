"""Timing-only experiments on sgemm variants built with -DSGEMM_EXP=n (results are wrong by design for n != 0).
usage: python tools/gemm_exp.py lib1.so lib2.so ...   (each a full liblaed_hip build)"""
import ctypes, sys, torch

SHAPES = [("NT logits", 8192, 50000, 1024, 0, 1), ("NN dh", 8192, 1024, 50000, 0, 0), ("TN dW", 50000, 1024, 8192, 1, 0),
          ("NT square", 8192, 8192, 8192, 0, 1), ("TN dW K=40960", 50000, 1024, 40960, 1, 0),
          ("NT logits M=40960", 40960, 50000, 1024, 0, 1), ("NN dh M=40960", 40960, 1024, 50000, 0, 0)]


def run(path):
    lib = ctypes.CDLL(path)
    p, i = ctypes.c_void_p, ctypes.c_int
    lib.laed_sgemm.argtypes = [i, i, i, i, i, p, i, p, i, p, i, p, i, p, i, p]
    lib.laed_sgemm.restype = i
    out = []
    for name, M, N, K, tA, tB in SHAPES:
        A = torch.randn((K, M) if tA else (M, K), device="cuda")
        B = torch.randn((N, K) if tB else (K, N), device="cuda")
        C = torch.empty((M, N), device="cuda")
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        f = lambda: lib.laed_sgemm(tA, tB, M, N, K, A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], C.data_ptr(), N, None, 0, None, 0, st)
        for _ in range(2):
            assert f() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out.append("%s %.3f ms %.1f TF" % (name, ms, 2.0 * M * N * K / ms / 1e9))
        del A, B, C
    # the projection's dW product with workspace (tail split allowed)
    lib.laed_sgemm_ws.argtypes = [i, i, i, i, i, p, i, p, i, p, i, p, i, p, i, p, ctypes.c_int64, p]
    lib.laed_sgemm_ws.restype = i
    for K in (8192, 40960):
        M, N = 50000, 1024
        A = torch.randn((K, M), device="cuda"); B = torch.randn((K, N), device="cuda"); C = torch.zeros((M, N), device="cuda")
        cs = torch.zeros(M, device="cuda")
        ws = torch.empty(8 * 8192 * 1024, device="cuda")
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        f = lambda: lib.laed_sgemm_ws(1, 0, M, N, K, A.data_ptr(), M, B.data_ptr(), N, C.data_ptr(), N, None, 1, cs.data_ptr(), 1,
                                      ws.data_ptr(), ws.numel(), st)
        for _ in range(2):
            assert f() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out.append("TN dW+ws K=%d %.3f ms %.1f TF" % (K, ms, 2.0 * M * N * K / ms / 1e9))
    print(path.split("/")[-1], " | ".join(out), flush=True)


for path in sys.argv[1:]:
    run(path)
