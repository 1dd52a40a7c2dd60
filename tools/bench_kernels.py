"""Micro-benchmarks of the hot kernels at the scaled shape (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "neuraldialog-laed_amd"))
import torch
from laed_amd import ops


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


def gemm_case(name, M, N, K, tA, tB):
    A = torch.randn((K, M) if tA else (M, K), device="cuda")
    B = torch.randn((N, K) if tB else (K, N), device="cuda")
    C = torch.empty((M, N), device="cuda")
    ms = timeit(lambda: ops.sgemm(A, B, transA=tA, transB=tB, out=C))
    print("%-34s M=%6d N=%6d K=%6d  %8.3f ms  %7.1f TFLOP/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)


if __name__ == "__main__":
    H, V, E = 1024, 50000, 200
    R = 5120
    ops.set_gemm_mode("bf16x3")
    gemm_case("bf16x3 logits NT (h W^T)", R, V, H, False, True)
    gemm_case("bf16x3 square 4096 NT", 4096, 4096, 4096, False, True)
    gemm_case("bf16x3 square 8192 NT", 8192, 8192, 8192, False, True)
    gemm_case("bf16x3 gi NT B=4096", 4096 * 40, 3 * H, E, False, True)
    ops.set_gemm_mode("fp32")
    gemm_case("logits NT (h W^T)", R, V, H, False, True)
    gemm_case("dh NN (dl W)", R, H, V, False, False)
    gemm_case("dW TN (dl^T h)", V, H, R, True, False)
    gemm_case("gi NT (x W_ih^T) B=4096", 4096 * 40, 3 * H, E, False, True)
    gemm_case("square 4096 NT", 4096, 4096, 4096, False, True)
    gemm_case("square 4096 NN", 4096, 4096, 4096, False, False)
    gemm_case("square 4096 TN", 4096, 4096, 4096, True, False)
    gemm_case("dh step NN B=512", 512, H, 3 * H, False, False)
    gemm_case("dh step NN B=4096", 4096, H, 3 * H, False, False)
    gemm_case("dW_hh TN B=512", 3 * H, H, 512 * 40, True, False)
    for B in (512, 4096):
        T = 8
        x = torch.randn(T, B, E, device="cuda")
        w_ih, w_hh = torch.randn(3 * H, E, device="cuda") * 0.03, torch.randn(3 * H, H, device="cuda") * 0.03
        b = torch.zeros(3 * H, device="cuda")
        ms = timeit(lambda: ops.GRUFunction.apply(x, w_ih, w_hh, b, b, None))
        fl = 2.0 * T * B * 3 * H * (E + H)
        print("gru fwd T=%d B=%d: %.3f ms (%.1f us/step incl. gi)  %.1f TFLOP/s" % (T, B, ms, ms * 1e3 / T, fl / ms / 1e9), flush=True)
