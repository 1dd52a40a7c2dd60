"""Where do the 44 us of embed_scatter_small_kernel at the P shape go?  Times laed_embed_scatter_add (through the C ABI) on the
bench's own id matrices (decoder inputs [30, 39], encoder inputs [30, 40]; time-major as the models pass them) with and
without the dropout mask, and on uniform ids.   usage (GPU box): python tools/exp/scatter_small_probe.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "neuraldialog-laed_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from laed_amd._lib import lib
from laed_amd.ops import ptr, check

V, E = 10002, 200
full = bench.synth_batch(V, 40, 30)
cases = {"decoder ids [39 x 30] time-major": np.ascontiguousarray(full[:, :-1].T), "encoder ids [40 x 30] time-major": np.ascontiguousarray(full.T),
         "decoder ids batch-major": np.ascontiguousarray(full[:, :-1]),
         "uniform ids 1170": np.random.RandomState(0).randint(4, V, size=(39, 30))}
for name, ids_np in cases.items():
    ids = torch.from_numpy(ids_np.astype(np.int64)).cuda().reshape(-1)
    n = ids.numel()
    dout = torch.randn(n, E, device="cuda")
    keep = (torch.rand(n, E, device="cuda") > 0.3).to(torch.uint8)
    table = torch.zeros(V, E, device="cuda")
    for mask in (False, True):
        def run():
            check(lib.laed_embed_scatter_add(ptr(ids, torch.int64), n, ptr(dout), V, E, ptr(keep, torch.uint8) if mask else None, 0.3 if mask else 0.0, 0,
                                             ptr(table), None, 0, None), "scatter")
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): run()
        e1.record(); torch.cuda.synchronize()
        uniq, cnt = np.unique(ids_np, return_counts=True)
        print("%-36s mask=%d  %6.1f us   rows %d  distinct ids %d  largest run %d" % (name, mask, e0.elapsed_time(e1) * 5, n, len(uniq), cnt.max()))
