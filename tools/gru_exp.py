"""Timing of the tiled GRU step kernel (us per forward step, input GEMM negligible: E = 8) for one or more builds of the library.

Ablation builds: compile csrc/gru.hip with -DGRU_EXP=n and link it with the other objects; the bits of n remove, from the
scheduled K loop of interior blocks, 1 = the barrier, 2 = the staging (LDS stores + global loads), 4 = the fragment reads,
and outside it 8 = the gate epilogue, 16 = the gate-operand loads (results are wrong by design for n != 0; only the clock
counts).  GRU_EXP_B="512,4096" and GRU_EXP_H=1024 choose the shapes; LAED_GRU_WM forces a block geometry (gru.hip gru_step).
usage: python tools/gru_exp.py lib.so ..."""
import ctypes, os, sys, torch
H = int(os.environ.get("GRU_EXP_H", "1024"))


def run(path):
    lib = ctypes.CDLL(path)
    out = []
    for B in [int(b) for b in os.environ.get("GRU_EXP_B", "512,4096").split(",")]:
        T = 8
        p = ctypes.c_void_p
        hbuf = torch.randn(T + 1, B, H, device="cuda") * 0.1
        gates = torch.randn(T, B, 3 * H, device="cuda")
        hn = torch.empty(T, B, H, device="cuda")
        x = torch.randn(T, B, 8, device="cuda")
        w_ih = torch.randn(3 * H, 8, device="cuda") * 0.03
        w_hh = torch.randn(3 * H, H, device="cuda") * 0.03
        b = torch.zeros(3 * H, device="cuda")
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib.laed_gru_fwd.argtypes = [p] * 6 + [ctypes.c_int] * 4 + [p] * 4 + [ctypes.c_int64, p]
        f = lambda: lib.laed_gru_fwd(x.data_ptr(), w_ih.data_ptr(), w_hh.data_ptr(), b.data_ptr(), b.data_ptr(), None, T, B, 8, H,
                                     hbuf.data_ptr(), gates.data_ptr(), hn.data_ptr(), None, 0, st)
        for _ in range(2):
            assert f() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        out.append("B=%d %.1f us/step" % (B, e0.elapsed_time(e1) / 20 / T * 1e3))
    print(path.split("/")[-1], " | ".join(out), flush=True)


for path in sys.argv[1:]:
    run(path)
