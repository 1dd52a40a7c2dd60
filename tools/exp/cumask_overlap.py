"""Would a CU-masked second stream let weight-gradient GEMMs run beside a persistent recurrence?

main stream : one GRU forward + backward at the P shape (T = 40, B = 30, E = 200, H = 512: persistent launches of 128 / 64
              workgroups, 79 dependent steps each, plus the post-loop products)
side stream : NG products of the projection's dW shape at P (10002 x 512 x 1170, A^T B), ~125 us each
Reported: each alone, both on two ordinary streams, both with the side stream restricted by hipExtStreamCreateWithCUMask to
the CUs the mask leaves (two mask layouts: the first 64 CU bits cleared / every fourth bit cleared)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "neuraldialog-laed_amd"))
import torch
from laed_amd import ops

hip = ctypes.CDLL("libamdhip64.so")
T, B, E, H, NG = 40, 30, 200, 512, int(os.environ.get("NG", "3"))
x = torch.randn(T, B, E, device="cuda") * 0.5
w_ih = (torch.randn(3 * H, E, device="cuda") * 0.03).requires_grad_(True)
w_hh = (torch.randn(3 * H, H, device="cuda") * 0.03).requires_grad_(True)
bias = torch.zeros(3 * H, device="cuda", requires_grad=True)
A = torch.randn(1170, 10002, device="cuda") * 0.1
Bm = torch.randn(1170, 512, device="cuda") * 0.1
C = torch.empty(10002, 512, device="cuda")


def chain():
    out, hT = ops.GRUFunction.apply(x, w_ih, w_hh, bias, bias, None)
    out.sum().backward()


def gemms():
    for _ in range(NG):
        ops.sgemm(A, Bm, transA=True, out=C, splitk_ws=True)


def masked_stream(mask_words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(mask_words))(*mask_words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(mask_words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3


def both(side):
    def run():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gemms()
        chain()
        torch.cuda.current_stream().wait_stream(side)
    return run


tc, tg = timed(chain), timed(gemms)
print("chain alone %.1f us   %d GEMMs alone %.1f us   serial %.1f us" % (tc, NG, tg, tc + tg), flush=True)
print("two ordinary streams        : %.1f us" % timed(both(torch.cuda.Stream())), flush=True)
full = [0xFFFFFFFF] * 8
m1 = list(full); m1[0] = 0; m1[1] = 0                       # CU bits 0..63 cleared
m2 = [0xEEEEEEEE] * 8                                       # every fourth bit cleared
for name, m in (("bits 0-63 cleared", m1), ("every 4th bit cleared", m2), ("all bits set", full)):
    s = masked_stream(m)
    with torch.cuda.stream(s):
        tgm = timed(gemms)
    print("side masked (%s): GEMMs alone on it %.1f us, both %.1f us" % (name, tgm, timed(both(s))), flush=True)
print("persistent launches (fwd, bwd):", ops.gru_persist_launches(), "timeouts", ops.gru_persist_timeouts())
