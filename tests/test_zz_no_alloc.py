"""The allocation-free entry point `laed_latent_stats_noalloc` (file-scope __device__ scratch, no workspace argument)
against `laed_latent_stats` through the C ABI: bit-exact (same kernels, same fixed-order reduction), on two calls in a
row, the second with a smaller input -- stale partials of the first call must not leak into the second -- and the size
limit is refused with a clear error.  Collected last on purpose (see the file name)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "neuraldialog-laed_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _log_qy(B, y, k, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.log_softmax(torch.randn(B * y, k, generator=g) * 2.0, dim=1).cuda()


def _stats_ws(lq, B, y, k):
    from laed_amd._lib import check, lib, ptr, stream
    nblk = int(lib.laed_latent_stats_blocks())
    ws = torch.empty(nblk * (y * k + 2), device="cuda", dtype=torch.float32)
    out = torch.full((y * k + 2,), float("nan"), device="cuda", dtype=torch.float32)
    check(lib.laed_latent_stats(ptr(lq), B, y, k, ptr(ws), ptr(out), stream()), "laed_latent_stats")
    return out


def _stats_noalloc(lq, B, y, k, out=None):
    from laed_amd._lib import check, lib, ptr, stream
    if out is None:
        out = torch.full((y * k + 2,), float("nan"), device="cuda", dtype=torch.float32)
    check(lib.laed_latent_stats_noalloc(ptr(lq), B, y, k, ptr(out), stream()), "laed_latent_stats_noalloc")
    return out


def test_noalloc_latent_stats_matches_workspace_variant_twice():
    from laed_amd._lib import check, lib, ptr, stream
    from oracle import laed_oracle as O
    for B, y, k, seed in ((4096, 20, 10, 1), (37, 3, 5, 2)):          # second call: smaller input, fewer blocks
        lq = _log_qy(B, y, k, seed)
        a = _stats_noalloc(lq, B, y, k)
        b = _stats_ws(lq, B, y, k)
        torch.cuda.synchronize()
        assert torch.equal(a, b), (B, y, k, float((a - b).abs().max()))
        # the device's free memory does not move across a call whose output buffer already exists
        out = torch.full((y * k + 2,), float("nan"), device="cuda", dtype=torch.float32)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        _stats_noalloc(lq, B, y, k, out)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0
        assert torch.equal(out, b)
        # and both agree with the oracle's batch sums (sent_models.py:138: mean_b exp(log_qy) = stats / B)
        q = torch.exp(lq.cpu().double()).view(B, y * k).sum(0)
        assert float((a[:y * k].cpu().double() - q).abs().max()) <= 1e-5 * float(q.abs().max())
        reg_kl, bpr, mi = O.latent_losses(lq.cpu(), y, k, True)
        scal = torch.empty(4, device="cuda")
        check(lib.laed_latent_finalize(ptr(a), float(B), y, k, ptr(scal), None, stream()), "laed_latent_finalize")
        assert abs(float(scal[0]) - float(bpr)) <= 1e-7 + 1e-4 * abs(float(bpr))


def test_noalloc_refuses_sizes_beyond_its_scratch():
    from laed_amd._lib import LaedError
    lq = _log_qy(2, 300, 10, 3)                                       # y*k = 3000 > 2048
    with pytest.raises(LaedError, match="exceeds the built-in scratch"):
        _stats_noalloc(lq, 2, 300, 10)
