"""Which forced tile configuration of the GRU step kernel disagrees with the oracle, and where."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "neuraldialog-laed_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from laed_amd import ops
from oracle import laed_oracle as O

def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale

def case(T, B, E, H, seed):
    x = rnd(T, B, E, seed=seed); k = 1.0 / np.sqrt(H)
    w_ih, w_hh = rnd(3 * H, E, seed=seed + 1, scale=k), rnd(3 * H, H, seed=seed + 2, scale=k)
    b_ih, b_hh = rnd(3 * H, seed=seed + 3, scale=k), rnd(3 * H, seed=seed + 4, scale=k)
    h0 = rnd(B, H, seed=seed + 5)
    with torch.no_grad():
        out_c, _ = O.gru_seq(x.transpose(0, 1), h0, w_ih, w_hh, b_ih, b_hh)
    out_c = out_c.transpose(0, 1)
    for train in (True, False):
        gp = [t.cuda().requires_grad_(train) for t in (x, w_ih, w_hh, b_ih, b_hh)]
        out_g, _ = ops.GRUFunction.apply(gp[0], gp[1], gp[2], gp[3], gp[4], h0.cuda())
        d = (out_g.detach().cpu() - out_c).abs()
        bad = (d > 1e-5).nonzero()
        print("  T=%d B=%d H=%d train=%d: max err %.3e, bad %d" % (T, B, H, train, float(d.max()), bad.shape[0]), end="")
        if bad.shape[0]:
            print("  t in", sorted(set(bad[:, 0].tolist())), "rows", sorted(set(bad[:, 1].tolist()))[:12], "... cols", sorted(set(bad[:, 2].tolist()))[:12], end="")
        print(flush=True)

for wm in sys.argv[1:]:
    if wm != "auto":
        os.environ["LAED_GRU_WM"] = wm
    print("LAED_GRU_WM =", wm, flush=True)
    case(3, 260, 32, 96, 20)
    case(2, 5, 8, 40, 21)
    case(2, 300, 16, 512, 22)
