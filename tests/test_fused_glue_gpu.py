"""The "launch diet" entry points (round 3): kernels that fold the small torch compositions around the hot kernels into
the kernels themselves must give the SAME BITS as the compositions they replace (integer / selection work: exact; the
floating-point ones reuse the same arithmetic in the same order)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from laed_amd import ops as _ops
    return _ops


def test_embed_gather_bt_equals_transpose_mask_gather(ops):
    """laed_embed_gather_bt (ids transposition + dropout decisions + gather in one launch) against the three-step form:
    ids.t(), laed_dropout_keep_mask over the batch-first index space transposed to time-major, laed_embed_gather.
    ids are a strided view (the models pass out_utts[:, :-1])."""
    from laed_amd._lib import lib, ptr, stream, check
    torch.manual_seed(0)
    V, E, B, T, p = 977, 24, 7, 11, 0.3
    W = torch.randn(V, E, device="cuda")
    W[0] = 0
    full = torch.randint(0, V, (B, T + 1), device="cuda")
    full[2, 5:] = 0
    ids = full[:, :-1]                                          # row stride T+1
    seed, off = 123456789, 77 * B * T * E
    # reference composition
    keep_bf = ops.dropout_keep_mask(B * T * E, p, seed, off, "cuda").view(B, T, E)
    keep_tm = keep_bf.transpose(0, 1).contiguous()
    ids_tm = ids.t().contiguous().view(-1)
    ref = torch.empty(T * B, E, device="cuda")
    check(lib.laed_embed_gather(ptr(ids_tm, torch.int64), T * B, ptr(W), V, E, ptr(keep_tm.view(-1), torch.uint8), p, ptr(ref),
                                stream()), "gather")
    for mode in ("rng", "injected", "none"):
        w = W.clone().requires_grad_(True)
        out = ops.EmbedBTFunction.apply(w, ids, keep_tm if mode == "injected" else None,
                                        (seed, off) if mode == "rng" else None, p if mode != "none" else 0.0, 0)
        if mode == "none":
            assert torch.equal(out, W[ids_tm])
        else:
            assert torch.equal(out, ref), mode
        # backward: the saved transposed ids / mask drive the scatter-add exactly as the old EmbedFunction's did
        g = torch.randn_like(out)
        out.backward(g)
        w2 = W.clone().requires_grad_(True)
        old = ops.EmbedFunction.apply(w2, ids_tm, keep_tm.view(-1) if mode != "none" else None, p if mode != "none" else 0.0, 0)
        old.backward(g)
        assert torch.equal(w.grad, w2.grad), mode
        assert float(w.grad[0].abs().max()) == 0.0             # PAD row: no gradient


def test_gumbel_rng_in_kernel_equals_uniform_then_gumbel(ops):
    torch.manual_seed(1)
    for R, k, hard in ((600, 10, False), (90, 5, True), (37, 64, False), (5, 1, False)):
        logits = torch.randn(R, k, device="cuda")
        seed, off = 987654321, 1000 * k
        u = ops.uniform(R * k, seed, off, "cuda").view(R, k)
        y0, lq0, id0 = ops.GumbelLatentFunction.apply(logits, u, 0.7, hard)
        y1, lq1, id1 = ops.GumbelLatentFunction.apply(logits, None, 0.7, hard, (seed, off))
        assert torch.equal(y0, y1) and torch.equal(lq0, lq1) and torch.equal(id0, id1), (R, k, hard)


def test_connector_default_noise_is_drawn_in_the_kernel_and_keeps_its_stream(ops):
    """GumbelConnector without an injected sample_uniform: same sample as replaying sample_uniform's own uniforms, and
    the step counter advances once per call either way."""
    from laed_amd.nn_lib import GumbelConnector
    torch.manual_seed(2)
    logits = torch.randn(80, 10, device="cuda")
    a, b = GumbelConnector(), GumbelConnector()
    for _ in range(3):
        ya, ia = a(logits, 1.0, True, hard=False, return_max_id=True)
        u = b.sample_uniform(logits)                           # advances b.step like the in-kernel path advances a.step
        yb, _, ib = ops.GumbelLatentFunction.apply(logits, u, 1.0, False)
        assert torch.equal(ya, yb) and torch.equal(ia.view(-1), ib) and a.step == b.step


def test_latent_stats_finalize_equals_the_three_launch_form(ops):
    from laed_amd._lib import lib, ptr, stream, check
    torch.manual_seed(3)
    for B, y, k in ((30, 20, 10), (4096, 20, 10), (1, 3, 5), (300, 4, 6)):
        log_qy = torch.log_softmax(torch.randn(B * y, k, device="cuda"), dim=1)
        n = y * k + 2
        nblk = int(lib.laed_latent_stats_blocks())
        ws = torch.empty(nblk * n, device="cuda")
        st0, sc0, cf0 = torch.empty(n, device="cuda"), torch.empty(4, device="cuda"), torch.empty(y * k, device="cuda")
        check(lib.laed_latent_stats(ptr(log_qy), B, y, k, ptr(ws), ptr(st0), stream()), "stats")
        check(lib.laed_latent_finalize(ptr(st0), float(B), y, k, ptr(sc0), ptr(cf0), stream()), "finalize")
        st1, sc1, cf1 = torch.empty(n, device="cuda"), torch.empty(4, device="cuda"), torch.empty(y * k, device="cuda")
        check(lib.laed_latent_stats_finalize(ptr(log_qy), B, y, k, ptr(ws), ptr(st1), ptr(sc1), ptr(cf1), stream()), "fused")
        assert torch.equal(st0, st1) and torch.equal(sc0, sc1) and torch.equal(cf0, cf1), (B, y, k)


@pytest.mark.parametrize("avg_type", [None, "seq", "real_word", "word"])
def test_nll_prepare_equals_the_torch_composition(ops, avg_type):
    """laed_nll_prepare against NLLEntropy._row_scale + the transposing copies it replaces (criterions.py:43-72), on a
    strided label view with PAD tails, time-major and batch-first, with and without a pre-reduced global count."""
    from argparse import Namespace
    from laed_amd import criterions
    torch.manual_seed(4)
    B, T = 9, 13
    full = torch.randint(1, 50, (B, T + 1), device="cuda")
    for b in range(B):
        full[b, 3 + b:] = 0
    labels = full[:, 1:]
    crit = criterions.NLLEntropy(0, Namespace(avg_type=avg_type))
    ref = crit._row_scale(labels, B, T)
    lab_tm, sc_tm, cnt = ops.nll_prepare(labels, 0, avg_type, 1, time_major=True, want_local_count=True)
    assert torch.equal(lab_tm, labels.t().contiguous().view(-1))
    assert torch.equal(sc_tm, ref.t().contiguous().view(-1)), avg_type
    assert float(cnt) == float((labels != 0).sum())
    lab_bf, sc_bf, _ = ops.nll_prepare(labels, 0, avg_type, 1, time_major=False)
    assert torch.equal(lab_bf, labels.reshape(-1)) and torch.equal(sc_bf, ref.contiguous().view(-1))
    if avg_type == "word":                                    # a count the caller already all-reduced (DP)
        g = torch.tensor([123.0], device="cuda")
        _, sc_g, _ = ops.nll_prepare(labels, 0, avg_type, 4, global_count=g)
        assert torch.equal(sc_g, (1.0 / g).expand(B * T))
    if avg_type in ("seq", "real_word"):                      # world size enters the denominator
        _, sc_w, _ = ops.nll_prepare(labels, 0, avg_type, 4)
        if avg_type == "seq":
            assert torch.equal(sc_w, torch.full((B * T,), 1.0 / (B * 4), device="cuda"))
        else:
            wc = torch.sign(labels).sum(dim=1, keepdim=True).float()
            assert torch.equal(sc_w, (1.0 / (wc * (B * 4))).expand(B, T).t().contiguous().view(-1))


def test_scale_by_dev_multi(ops):
    torch.manual_seed(5)
    ts = [torch.randn(n, device="cuda") for n in (5, 70000, 1, 1024 * 1024 + 3)]
    ref = [t.clone() for t in ts]
    g = torch.tensor([0.37], device="cuda")
    ops.scale_by_dev_multi_(ts, g)
    for t, r in zip(ts, ref):
        assert torch.equal(t, r * g)
    one = torch.ones(1, device="cuda")
    before = [t.clone() for t in ts]
    ops.scale_by_dev_multi_(ts, one)
    assert all(torch.equal(a, b) for a, b in zip(ts, before))
