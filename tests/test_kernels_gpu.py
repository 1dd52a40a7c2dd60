"""GPU parity tests of every C-ABI kernel against the CPU oracle / plain torch-CPU fp32 math.

All calls go through liblaed_hip.so (ctypes).  Tolerances: fp32 MFMA is an exact fp32 fma chain, so results
differ from the CPU only by summation order: 1e-5 relative on GEMM-shaped outputs, 1e-4 after recurrences.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import rel_err  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from laed_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def O():
    from oracle import laed_oracle
    return laed_oracle


def dev(t):
    return t.cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize("M,N,K", [(70, 50, 33), (128, 128, 32), (300, 200, 129), (1170, 1000, 512), (2048, 2304, 200),
                                   (1, 7, 5), (33, 1, 64),
                                   # > 1024 tiles of 128x128: the 8-wave kernels (16-deep K tile for x W^T / x W, 32-deep
                                   # for A^T B) with edge tiles and a K tail that is a multiple of neither depth
                                   (4100, 4100, 1028)])
def test_sgemm_layouts(ops, M, N, K):
    for (tA, tB) in ((False, True), (False, False), (True, False)):
        A = rnd(K, M, seed=1) if tA else rnd(M, K, seed=1)
        B = rnd(N, K, seed=2) if tB else rnd(K, N, seed=2)
        bias = rnd(N, seed=3)
        ref = (A.t() if tA else A).double() @ (B.t() if tB else B).double() + bias.double()
        out = ops.sgemm(dev(A), dev(B), transA=tA, transB=tB, bias=dev(bias))
        assert rel_err(out.cpu(), ref) < 2e-6, (tA, tB)
        # accumulate + column sums of A (transposed-A layout only)
        C0 = rnd(M, N, seed=4)
        Cd = dev(C0.clone())
        cs = dev(torch.ones(M)) if tA else None
        ops.sgemm(dev(A), dev(B), transA=tA, transB=tB, out=Cd, accumulate=True, colsum_out=cs, colsum_accumulate=True)
        assert rel_err(Cd.cpu(), ref - bias.double() + C0.double()) < 2e-6
        if tA:
            assert rel_err(cs.cpu(), A.double().sum(0) + 1.0) < 2e-6


@pytest.mark.parametrize("M,N,K", [(96, 200, 4096), (512, 1024, 3072), (3072, 200, 5000), (64, 64, 300)])
def test_sgemm_splitk(ops, M, N, K):
    """Few output tiles + long K: the library splits K over blocks (caller workspace) and sums partials in order."""
    for (tA, tB) in ((False, True), (False, False), (True, False)):
        A = rnd(K, M, seed=1) if tA else rnd(M, K, seed=1)
        B = rnd(N, K, seed=2) if tB else rnd(K, N, seed=2)
        bias, C0 = rnd(N, seed=3), rnd(M, N, seed=4)
        ref = (A.t() if tA else A).double() @ (B.t() if tB else B).double() + bias.double() + C0.double()
        Cd = dev(C0.clone())
        cs = dev(torch.ones(M)) if tA else None
        ops.sgemm(dev(A), dev(B), transA=tA, transB=tB, bias=dev(bias), out=Cd, accumulate=True, colsum_out=cs,
                  colsum_accumulate=True)
        assert rel_err(Cd.cpu(), ref) < 2e-6, (tA, tB)
        if tA:
            assert rel_err(cs.cpu(), A.double().sum(0) + 1.0) < 2e-6
        # same result (bitwise) on a second run: fixed-order reduction, no atomics
        Cd2 = dev(C0.clone())
        ops.sgemm(dev(A), dev(B), transA=tA, transB=tB, bias=dev(bias), out=Cd2, accumulate=True)
        assert torch.equal(Cd, Cd2)
        # and equal (to rounding) to the unsplit kernel
        Cd3 = dev(C0.clone())
        ops.sgemm(dev(A), dev(B), transA=tA, transB=tB, bias=dev(bias), out=Cd3, accumulate=True, splitk_ws=False)
        assert rel_err(Cd3.cpu(), ref) < 1e-5      # one K-long fp32 fma chain: more rounding than the split sum


@pytest.mark.parametrize("M,N,K", [(4224, 2048, 512), (8500, 1024, 700), (4100, 2100, 1031)])
def test_sgemm_tail_split(ops, M, N, K):
    """More than one round of 128x128 tiles with a short last round: with workspace the library cuts the rows into a
    whole-rounds launch plus a split-K launch for the remaining row tiles.  Same results as the single launch."""
    for (tA, tB) in ((True, False), (False, True), (False, False)):
        A = rnd(K, M, seed=1) if tA else rnd(M, K, seed=1)
        B = rnd(N, K, seed=2) if tB else rnd(K, N, seed=2)
        bias, C0 = rnd(N, seed=3), rnd(M, N, seed=4)
        ref = (A.t() if tA else A).double() @ (B.t() if tB else B).double() + bias.double() + C0.double()
        Cd = dev(C0.clone())
        cs = dev(torch.ones(M)) if tA else None
        ops.sgemm(dev(A), dev(B), transA=tA, transB=tB, bias=dev(bias), out=Cd, accumulate=True, colsum_out=cs,
                  colsum_accumulate=True, ws_floats=16 * (256 * N + 256))
        assert rel_err(Cd.cpu(), ref) < 2e-6, (tA, tB)
        if tA:
            assert rel_err(cs.cpu(), A.double().sum(0) + 1.0) < 2e-6
        Cd2 = dev(C0.clone())
        ops.sgemm(dev(A), dev(B), transA=tA, transB=tB, bias=dev(bias), out=Cd2, accumulate=True, splitk_ws=False)
        assert rel_err(Cd2.cpu(), Cd.cpu().double()) < 2e-6


@pytest.mark.parametrize("M,N,K", [(2048, 2304, 200), (1536, 2100, 1024), (4000, 1111, 777), (2048, 2048, 4096)])
def test_xgemm_bf16x3(ops, M, N, K):
    """Opt-in split-bf16 math mode for x W^T: fp32-grade results (error within ~10x of the exact-fp32 kernel)."""
    A, B, bias, C0 = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = A.double() @ B.double().t() + bias.double() + C0.double()
    exact = dev(C0.clone())
    ops.sgemm(dev(A), dev(B), transB=True, bias=dev(bias), out=exact, accumulate=True)
    try:
        ops.set_gemm_mode("bf16x3")
        assert ops.get_gemm_mode() == "bf16x3"
        split = dev(C0.clone())
        ops.sgemm(dev(A), dev(B), transB=True, bias=dev(bias), out=split, accumulate=True)
        # unaligned leading dimensions -> scalar-load variant of the same kernel
        Ad, Bd = dev(torch.cat([A, A[:, :1]], 1)), dev(torch.cat([B, B[:, :1]], 1))
        split2 = ops.sgemm(Ad[:, :K], Bd[:, :K], transB=True, bias=dev(bias))
    finally:
        ops.set_gemm_mode("fp32")
    e_exact, e_split = rel_err(exact.cpu(), ref), rel_err(split.cpu(), ref)
    assert e_exact < 2e-6
    assert e_split < 2e-5 and e_split < 20 * max(e_exact, 1e-7), (e_exact, e_split)
    assert rel_err(split2.cpu(), ref - C0.double()) < 2e-5
    assert not torch.equal(exact, split)          # the mode switch really selects another kernel


def test_sgemm_unaligned_views(ops):
    # leading dimensions that are not multiples of 4 and 4-byte-aligned bases -> scalar-load kernels
    A = rnd(90, 71, seed=5)
    B = rnd(60, 71, seed=6)
    Ad, Bd = dev(A), dev(B)
    out = ops.sgemm(Ad[1:, 1:], Bd[:, 1:], transB=True)
    assert rel_err(out.cpu(), A[1:, 1:].double() @ B[:, 1:].double().t()) < 2e-6
    outbuf = dev(torch.zeros(89, 75))
    ops.sgemm(Ad[1:, 1:], Bd[:, 1:], transB=True, out=outbuf[:, 3:63])
    assert rel_err(outbuf[:, 3:63].cpu(), A[1:, 1:].double() @ B[:, 1:].double().t()) < 2e-6
    assert float(outbuf[:, :3].abs().sum()) == 0.0 and float(outbuf[:, 63:].abs().sum()) == 0.0


def test_sgemm_bad_args(ops):
    from laed_amd._lib import LaedError
    with pytest.raises(LaedError):
        ops.sgemm(dev(rnd(4, 5)), dev(rnd(6, 7)))
    with pytest.raises(LaedError):
        ops.sgemm(rnd(4, 5), rnd(5, 7))          # CPU tensors are refused: no CPU fallback


def _gru_case(ops, O, T, B, E, H, with_h0, seed, tol):
    x = rnd(T, B, E, seed=seed)
    k = 1.0 / np.sqrt(H)
    w_ih, w_hh = rnd(3 * H, E, seed=seed + 1, scale=k), rnd(3 * H, H, seed=seed + 2, scale=k)
    b_ih, b_hh = rnd(3 * H, seed=seed + 3, scale=k), rnd(3 * H, seed=seed + 4, scale=k)
    h0 = rnd(B, H, seed=seed + 5) if with_h0 else None
    wout, whT = rnd(T, B, H, seed=seed + 6), rnd(B, H, seed=seed + 7)
    # oracle (batch-first) with autograd
    cp = [t.clone().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
    ch0 = h0.clone().requires_grad_(True) if with_h0 else torch.zeros(B, H)
    out_c, hT_c = O.gru_seq(cp[0].transpose(0, 1), ch0, cp[1], cp[2], cp[3], cp[4])
    ((out_c.transpose(0, 1) * wout).sum() + (hT_c * whT).sum()).backward()
    # HIP
    gp = [dev(t).requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
    gh0 = dev(h0).requires_grad_(True) if with_h0 else None
    out_g, hT_g = ops.GRUFunction.apply(gp[0], gp[1], gp[2], gp[3], gp[4], gh0)
    assert rel_err(out_g.detach().cpu(), out_c.detach().transpose(0, 1)) < tol
    assert rel_err(hT_g.detach().cpu(), hT_c.detach()) < tol
    ((out_g * dev(wout)).sum() + (hT_g * dev(whT)).sum()).backward()
    for a, b, name in zip(gp, cp, ("dx", "dw_ih", "dw_hh", "db_ih", "db_hh")):
        assert rel_err(a.grad.cpu(), b.grad) < 10 * tol, name
    if with_h0:
        assert rel_err(gh0.grad.cpu(), ch0.grad) < 10 * tol


@pytest.mark.parametrize("T,B,E,H,with_h0", [(5, 6, 16, 32, False), (7, 40, 20, 48, True), (4, 200, 64, 96, True),
                                             (40, 30, 200, 512, True), (3, 130, 36, 160, False),
                                             (3, 100, 32, 256, True), (2, 512, 40, 512, False)])
def test_gru_fwd_bwd(ops, O, T, B, E, H, with_h0):
    _gru_case(ops, O, T, B, E, H, with_h0, seed=10, tol=1e-5)


@pytest.mark.parametrize("T,B,E,H,with_h0", [(4, 1, 8, 64, True), (6, 30, 24, 128, False), (5, 33, 16, 192, True),
                                             (3, 64, 40, 1024, True), (9, 60, 200, 512, False), (1, 16, 8, 16, True),
                                             (4, 45, 12, 80, False), (5, 17, 16, 48, True), (7, 30, 200, 512, True),
                                             (3, 8, 8, 1040, True), (2, 20, 8, 2064, False),
                                             # row groups of the persistent backward: 8-row groups in half-filled tiles (B <= 32),
                                             # a last group of one row, four full groups, shard counts that do not divide
                                             (3, 9, 8, 32, True), (3, 25, 8, 64, False), (2, 32, 16, 96, True), (3, 24, 8, 256, True),
                                             (2, 31, 8, 1000 // 16 * 16, True)])
def test_gru_small_batch_kernels(ops, O, T, B, E, H, with_h0):
    """B <= 64 and H % 16 == 0 take gru_small.hip: packed operands, 4 / 8 units per workgroup; one persistent launch
    per direction with LDS-resident W_hh while all workgroups fit the chip (H <= 1024 forward, <= 2048 backward),
    per-step launches beyond."""
    _gru_case(ops, O, T, B, E, H, with_h0, seed=30, tol=1e-5)


@pytest.mark.parametrize("T,B,E,H,with_h0", [(4, 65, 12, 32, True), (6, 300, 40, 256, False), (3, 128, 16, 64, True),
                                             (2, 511, 8, 48, True), (5, 129, 20, 512, False)])
def test_gru_mid_batch_row_groups(ops, O, T, B, E, H, with_h0):
    """64 < B <= 512 (the B x turns utterances of an AE-ED / ST-ED batch): the packed per-step kernels over row groups
    of 64, ragged last group included."""
    _gru_case(ops, O, T, B, E, H, with_h0, seed=40, tol=1e-5)


def test_gru_all_tile_configs(ops, O):
    # the block-tile choice is a pure launch-geometry decision: every forced config must agree
    try:
        for wm in ("4", "2", "1", "8", "42", "22"):
            os.environ["LAED_GRU_WM"] = wm
            _gru_case(ops, O, 3, 260, 32, 96, True, seed=20, tol=1e-5)
            _gru_case(ops, O, 2, 5, 8, 40, False, seed=21, tol=1e-5)
    finally:
        os.environ.pop("LAED_GRU_WM", None)


def test_gate_functions(ops):
    """The branch-free sigmoid / tanh every GRU kernel applies (laed_common.h), against float64 over the whole range a
    gate pre-activation can take, including the switch of the tanh form at |x| = 0.625, large arguments and zeros."""
    from laed_amd._lib import check, lib, ptr, stream
    g = torch.Generator().manual_seed(5)
    x = torch.cat([torch.linspace(-30, 30, 200001), torch.linspace(-0.7, 0.7, 100001), (torch.rand(100000, generator=g) - 0.5) * 1e-3,
                   torch.tensor([0.0, -0.0, 0.625, -0.625, 88.0, -88.0, 1e4, -1e4, 1e-30, float("inf"), float("-inf")])])
    xd = x.cuda()
    sig, tnh = torch.empty_like(xd), torch.empty_like(xd)
    check(lib.laed_gate_functions(ptr(xd), xd.numel(), ptr(sig), ptr(tnh), stream()), "gate_functions")
    x64 = x.double()
    rs, rt = torch.sigmoid(x64), torch.tanh(x64)
    ea = (sig.cpu().double() - rs).abs()
    assert float(ea.max()) < 1.2e-7, float(ea.max())                    # absolute: what a gate value needs
    es = ea / rs.clamp_min(1e-30)
    assert float(es[x.abs() <= 4].max()) < 4e-7, float(es[x.abs() <= 4].max())
    # far tails: the rounding of x * log2(e) is an absolute error of |x| * 4e-8 in the exponent, i.e. a RELATIVE error that
    # grows with |x| on values that are themselves below 2e-2 (sigmoid(-30) = 9e-14 is off by 2e-6 of itself)
    far = (x > -80) & (x.abs() > 4)
    assert float((es[far] / x64.abs()[far]).max()) < 1e-7
    nz = x64.abs() > 0
    et = (tnh.cpu().double() - rt).abs()[nz] / rt.abs()[nz]
    assert float(et.max()) < 4e-7, float(et.max())
    assert torch.equal(tnh.cpu()[x == 0], x[x == 0]) and bool((torch.signbit(tnh.cpu()) == torch.signbit(x)).all())
    assert float(tnh.cpu()[-2]) == 1.0 and float(tnh.cpu()[-1]) == -1.0 and float(sig.cpu()[-2]) == 1.0 and float(sig.cpu()[-1]) == 0.0
    nan = torch.full((4,), float("nan")).cuda()
    check(lib.laed_gate_functions(ptr(nan), 4, ptr(sig), ptr(tnh), stream()), "gate_functions")
    assert bool(torch.isnan(sig[:4]).all()) and bool(torch.isnan(tnh[:4]).all())


def test_gumbel_latent(ops, O):
    for (R, k, tau, seed) in ((12, 5, 1.0, 1), (600, 10, 1.0, 2), (77, 20, 0.5, 3), (9, 40, 2.0, 4)):
        logits = rnd(R, k, seed=seed, scale=2.0)
        u = torch.rand(R, k, generator=torch.Generator().manual_seed(seed))
        wy, wl = rnd(R, k, seed=seed + 50), rnd(R, k, seed=seed + 60)
        lc = logits.clone().requires_grad_(True)
        y_c, ids_c = O.gumbel_softmax(lc, u, tau, hard=False)
        lq_c = torch.log_softmax(lc, dim=1)
        ((y_c * wy).sum() + (lq_c * wl).sum()).backward()
        lg = dev(logits).requires_grad_(True)
        y_g, lq_g, ids_g = ops.GumbelLatentFunction.apply(lg, dev(u), tau, False)
        assert rel_err(y_g.detach().cpu(), y_c.detach()) < 2e-6
        assert rel_err(lq_g.detach().cpu(), lq_c.detach()) < 2e-6
        assert np.array_equal(ids_g.cpu().numpy(), ids_c.view(-1).numpy())
        ((y_g * dev(wy)).sum() + (lq_g * dev(wl)).sum()).backward()
        assert rel_err(lg.grad.cpu(), lc.grad) < 1e-5
        y_h, _, ids_h = ops.GumbelLatentFunction.apply(dev(logits), dev(u), tau, True)
        yh_c, idh_c = O.gumbel_softmax(logits, u, tau, hard=True)
        assert np.array_equal(y_h.cpu().numpy(), yh_c.numpy()) and np.array_equal(ids_h.cpu().numpy(), idh_c.view(-1).numpy())


def test_latent_losses(ops, O):
    for (B, y, k, seed) in ((6, 4, 5, 1), (30, 20, 10, 2), (300, 20, 10, 3), (5, 3, 40, 4)):
        log_qy = torch.log_softmax(rnd(B * y, k, seed=seed, scale=3.0), dim=1)
        for use_mutual in (True, False):
            lc = log_qy.clone().requires_grad_(True)
            reg_c, bpr_c, mi_c = O.latent_losses(lc, y, k, use_mutual)
            (reg_c * 1.7).backward()
            lg = dev(log_qy).requires_grad_(True)
            bpr_g, mi_g, rowkl_g = ops.LatentLossFunction.apply(lg, y, k, None)[:3]
            reg_g = bpr_g if use_mutual else rowkl_g
            assert abs(float(reg_g) - float(reg_c)) <= 1e-7 + 1e-5 * abs(float(reg_c))
            assert abs(float(bpr_g) - float(bpr_c)) <= 1e-7 + 1e-5 * abs(float(bpr_c))
            assert abs(float(mi_g) - float(mi_c)) <= 2e-6 + 1e-4 * abs(float(mi_c))   # difference of two entropies
            (reg_g * 1.7).backward()
            assert rel_err(lg.grad.cpu(), lc.grad) < 1e-5
        # mi is a differentiable member of the reference's Pack (sent_models.py:155-158): its gradient, alone and mixed
        lc = log_qy.clone().requires_grad_(True)
        reg_c, bpr_c, mi_c = O.latent_losses(lc, y, k, True)
        (mi_c * 0.9 + bpr_c * 0.3).backward()
        lg = dev(log_qy).requires_grad_(True)
        bpr_g, mi_g, _ = ops.LatentLossFunction.apply(lg, y, k, None)[:3]
        (mi_g * 0.9 + bpr_g * 0.3).backward()
        assert rel_err(lg.grad.cpu(), lc.grad) < 2e-5, rel_err(lg.grad.cpu(), lc.grad)


@pytest.mark.parametrize("momentum", [0.0, 0.9])
def test_sgd_and_rmsprop_vs_torch(ops, momentum):
    """The sgd / rmsprop branches of get_optimizer (model_bases.py:126-133) against torch.optim on the CPU, 4 steps."""
    from laed_amd import optim
    for name in ("sgd", "rmsprop"):
        shapes = [(37, 5), (1,), (1000,), (3, 4, 5)]
        pc = [torch.nn.Parameter(rnd(*sh, seed=i)) for i, sh in enumerate(shapes)]
        pg = [torch.nn.Parameter(dev(p.detach().clone())) for p in pc]
        if name == "sgd":
            oc, og = torch.optim.SGD(pc, lr=0.05, momentum=momentum), optim.FusedSGD(pg, lr=0.05, momentum=momentum)
        else:
            oc, og = torch.optim.RMSprop(pc, lr=0.01, momentum=momentum), optim.FusedRMSprop(pg, lr=0.01, momentum=momentum)
        for step in range(4):
            oc.zero_grad()
            og.zero_grad()
            for i, (a, b) in enumerate(zip(pc, pg)):
                if i == 1 and step == 0:
                    continue                                   # a parameter without a gradient is skipped, as torch does
                g = rnd(*a.shape, seed=100 * step + i, scale=0.3)
                a.grad, b.grad = g.clone(), dev(g)
            oc.step()
            og.step()
            for a, b in zip(pc, pg):
                assert float((b.detach().cpu() - a.detach()).abs().max()) < 2e-6 * max(1.0, float(a.abs().max())), (name, step)


def test_attn_pool_and_tanh(ops):
    """RnnUttEncoder's attention pooling (encoders.py:119-130) and the prior network's tanh on their kernels, values
    and every gradient vs the same op sequence in torch CPU fp32; batch-first and time-major encoder memory."""
    from laed_amd.nn_lib import Linear
    for (N, T, D, Hk, seed) in ((1, 1, 8, 4, 1), (6, 9, 64, 32, 2), (40, 37, 200, 100, 3), (3, 130, 20, 7, 4)):
        torch.manual_seed(seed)
        key_c, query_c = torch.nn.Linear(D, Hk), torch.nn.Linear(Hk, 1)
        key_g, query_g = Linear(D, Hk), Linear(Hk, 1)
        key_g.load_state_dict(key_c.state_dict())
        query_g.load_state_dict(query_c.state_dict())
        key_g.cuda()
        query_g.cuda()
        enc = rnd(N, T, D, seed=seed)
        w = rnd(N, D, seed=seed + 1)
        for time_major in (False, True):
            for m in (key_c, query_c, key_g, query_g):
                m.zero_grad()
            ec = enc.clone().requires_grad_(True)
            fc1 = torch.tanh(key_c(ec))
            a_c = torch.softmax(query_c(fc1).squeeze(2), 1).unsqueeze(2)
            out_c = torch.sum(a_c * ec, dim=1)
            ((out_c * w).sum() + 0.3 * (ec * ec).sum()).backward()
            if time_major:
                leaf = dev(enc.transpose(0, 1).contiguous()).requires_grad_(True)      # [T,N,D] memory
                eg = leaf.transpose(0, 1)
            else:
                leaf = dev(enc).requires_grad_(True)
                eg = leaf
            out_g, a_g = ops.attn_pool(eg, key_g, query_g)
            assert rel_err(out_g.detach().cpu(), out_c.detach()) < 1e-5, (N, T, D, Hk, time_major)
            assert rel_err(a_g.cpu(), a_c.detach().squeeze(2)) < 1e-5
            ((out_g * dev(w)).sum() + 0.3 * (eg * eg).sum()).backward()
            g = leaf.grad.transpose(0, 1) if time_major else leaf.grad
            assert rel_err(g.cpu(), ec.grad) < 2e-5, (N, T, D, Hk, time_major)
            for mc, mg in ((key_c, key_g), (query_c, query_g)):
                assert rel_err(mg.weight.grad.cpu(), mc.weight.grad) < 5e-5, (N, T, D, Hk, time_major)
                assert float((mg.bias.grad.cpu() - mc.bias.grad).abs().max()) < 1e-5 * max(1.0, float(mc.weight.grad.abs().max()))
    x = rnd(37, 53, seed=9, scale=3.0)
    xc, xg = x.clone().requires_grad_(True), dev(x).requires_grad_(True)
    wgt = rnd(37, 53, seed=10)
    (torch.tanh(xc) * wgt).sum().backward()
    y = ops.tanh(xg)
    assert rel_err(y.detach().cpu(), torch.tanh(x)) < 1e-6
    (y * dev(wgt)).sum().backward()
    assert rel_err(xg.grad.cpu(), xc.grad) < 1e-5


def test_nll_entropy_dense_input(ops, O):
    """NLLEntropy handed a materialised log-prob tensor (criterions.py:43-72): the reference's golden vectors for the
    four averaging modes, then values and the scatter gradient vs the oracle (ignore index 0 and -100)."""
    from argparse import Namespace
    from helpers import load_golden
    from laed_amd import criterions
    G = load_golden("modules")
    logp, labels = torch.from_numpy(G["nll/logp"]), torch.from_numpy(G["nll/labels"])
    for avg in (None, "seq", "real_word", "word"):
        crit = criterions.NLLEntropy(0, Namespace(avg_type=avg))
        assert abs(float(crit(dev(logp), dev(labels))) - float(G["nll/out_%s" % avg])) < 1e-5, avg
    gen = torch.Generator().manual_seed(11)
    for (B, T, V) in ((1, 1, 3), (5, 7, 33), (30, 39, 1003)):
        logp = torch.log_softmax(rnd(B, T, V, seed=B, scale=2.0), dim=2)
        labels = torch.randint(1, V, (B, T), generator=gen)
        labels[B // 2, T // 2:] = 0                                        # a PAD tail
        labels[:, 0] = torch.randint(1, V, (B,), generator=gen)            # every row keeps one word ('real_word')
        for avg in (None, "seq", "real_word", "word"):
            lc = logp.clone().requires_grad_(True)
            ref = O.nll_entropy(lc, labels, avg)
            (ref * 1.1).backward()
            lg = dev(logp).requires_grad_(True)
            out = criterions.NLLEntropy(0, Namespace(avg_type=avg))(lg, dev(labels))
            assert abs(float(out) - float(ref)) <= 1e-6 + 1e-5 * abs(float(ref)), (B, T, V, avg)
            (out * 1.1).backward()
            assert rel_err(lg.grad.cpu(), lc.grad) < 1e-5, (B, T, V, avg)
    # ignore index -100 (the attribute loss of dialog_models.py): every label counts, id 0 included
    logp = torch.log_softmax(rnd(4, 6, 9, seed=3), dim=2)
    labels = torch.randint(0, 9, (4, 6), generator=gen)
    ref = torch.nn.functional.nll_loss(logp.view(-1, 9), labels.view(-1), ignore_index=-100)
    out = criterions.NLLEntropy(-100, Namespace(avg_type="word"))(dev(logp), dev(labels))
    assert abs(float(out) - float(ref)) < 1e-5
    bad = labels.clone()
    bad[0, 0] = 9                                                          # out of range: NaN, never a silent value
    assert np.isnan(float(criterions.NLLEntropy(-100, Namespace(avg_type="word"))(dev(logp), dev(bad))))


def test_cat_kl_and_entropy_modules(ops, O):
    """criterions.CatKLLoss / Entropy (the general modules, laed/criterions.py:93-109, 132-147) on the row kernels:
    the reference's golden vectors, then every log_py broadcast the reference uses, values and gradients vs the oracle."""
    from helpers import load_golden
    from laed_amd import criterions
    G = load_golden("modules")
    kl, ent = criterions.CatKLLoss(), criterions.Entropy()
    lq, lp = torch.from_numpy(G["catkl/log_qy"]), torch.from_numpy(G["catkl/log_py"])
    assert abs(float(kl(dev(lq), dev(lp), 4, True)) - float(G["catkl/unit"])) < 1e-6
    assert abs(float(kl(dev(lq), dev(lp), 4, False)) - float(G["catkl/batch"])) < 1e-6
    assert abs(float(ent(dev(lq), None, True)) - float(G["entropy/unit"])) < 1e-6
    assert abs(float(ent(dev(lq), 4, False)) - float(G["entropy/batch"])) < 1e-6

    for (R, k, seed) in ((1, 3, 1), (20, 10, 2), (600, 10, 3), (70001, 10, 4), (37, 40, 5), (9, 64, 6)):
        log_qy = torch.log_softmax(rnd(R, k, seed=seed, scale=3.0), dim=1)
        priors = {"uniform1": torch.log(torch.ones(1) / k),                       # sent_models.py:81
                  "row": torch.log_softmax(rnd(k, seed=seed + 7), dim=0),
                  "row2d": torch.log_softmax(rnd(1, k, seed=seed + 8), dim=1),
                  "full": torch.log_softmax(rnd(R, k, seed=seed + 9, scale=2.0), dim=1)}
        for name, log_py in priors.items():
            for unit in (True, False):
                py_grad = name == "full"
                qc, pc = log_qy.clone().requires_grad_(True), log_py.clone().requires_grad_(py_grad)
                ref = O.cat_kl(qc, pc, 7, unit)
                (ref * 1.3).backward()
                qg, pg = dev(log_qy).requires_grad_(True), dev(log_py).requires_grad_(py_grad)
                out = kl(qg, pg, 7, unit)
                assert abs(float(out) - float(ref)) <= 1e-6 + 2e-5 * abs(float(ref)), (R, k, name, unit)
                (out * 1.3).backward()
                assert rel_err(qg.grad.cpu(), qc.grad) < 1e-5, (R, k, name, unit)
                if py_grad:
                    assert rel_err(pg.grad.cpu(), pc.grad) < 1e-5
        for unit in (True, False):
            qc = log_qy.clone().requires_grad_(True)
            ref = O.entropy(qc, 5, unit)
            (ref * 0.7).backward()
            qg = dev(log_qy).requires_grad_(True)
            out = ent(qg.view(R, 1, k) if R > 1 else qg, 5, unit)                 # the dim > 2 squeeze of criterions.py:140
            assert abs(float(out) - float(ref)) <= 1e-6 + 2e-5 * abs(float(ref)), (R, k, unit)
            (out * 0.7).backward()
            assert rel_err(qg.grad.cpu(), qc.grad) < 1e-5
    # detached posterior (dialog_models.py: pi_kl): only the prior receives a gradient
    qg, pg = dev(log_qy), dev(priors["full"]).requires_grad_(True)
    kl(qg, pg, None, True).backward()
    assert rel_err(pg.grad.cpu(), -torch.exp(log_qy) / log_qy.shape[0]) < 1e-5
    a = kl(dev(log_qy), dev(priors["full"]), None, True)
    assert float(a) == float(kl(dev(log_qy), dev(priors["full"]), None, True))    # fixed-order reduction
    with pytest.raises(Exception):
        kl(dev(log_qy), dev(torch.zeros(3, log_qy.shape[1])), None, True)        # does not broadcast
    with pytest.raises(TypeError):
        kl(dev(log_qy), dev(priors["full"]))                                      # batch_size missing, as sum / None raises


@pytest.mark.parametrize("N,H,V,chunk", [(48, 32, 50, None), (100, 64, 401, 32), (1170, 128, 1003, 512), (300, 96, 2050, 128)])
def test_proj_nll(ops, N, H, V, chunk):
    h, W, b = rnd(N, H, seed=1), rnd(V, H, seed=2, scale=0.3), rnd(V, seed=3, scale=0.1)
    labels = torch.randint(1, V, (N,), generator=torch.Generator().manual_seed(4))
    labels[::7] = 0                                       # PAD rows are ignored
    cnt = float((labels != 0).sum())
    hc, Wc, bc = (t.clone().requires_grad_(True) for t in (h, W, b))
    logp = torch.log_softmax(hc @ Wc.t() + bc, dim=1)
    loss_c = torch.nn.functional.nll_loss(logp, labels, ignore_index=0)
    (loss_c * 1.3).backward()
    tok_c = -logp.detach().gather(1, labels.view(-1, 1)).view(-1) * (labels != 0)
    hg, Wg, bg = (dev(t).requires_grad_(True) for t in (h, W, b))
    row_scale = dev(torch.full((N,), 1.0 / cnt))
    if chunk is not None:
        tok, lse, dh, dW, db = ops.proj_nll_raw(hg.detach(), Wg.detach(), bg.detach(), dev(labels), row_scale, 0, True, chunk)
        assert rel_err(tok.cpu(), tok_c) < 1e-5
        assert rel_err(dh.cpu() * 1.3, hc.grad) < 1e-5 and rel_err(dW.cpu() * 1.3, Wc.grad) < 1e-5
        assert rel_err(db.cpu() * 1.3, bc.grad) < 1e-5
        assert rel_err(lse.cpu()[labels != 0], torch.logsumexp(h @ W.t() + b, dim=1)[labels != 0]) < 1e-5
    loss_g, tok_g = ops.ProjNLLFunction.apply(hg, Wg, bg, dev(labels), row_scale, 0)
    assert abs(float(loss_g) - float(loss_c)) < 1e-5 * abs(float(loss_c))
    assert rel_err(tok_g.cpu(), tok_c) < 1e-5
    (loss_g * 1.3).backward()
    assert rel_err(hg.grad.cpu(), hc.grad) < 1e-5
    assert rel_err(Wg.grad.cpu(), Wc.grad) < 1e-5
    assert rel_err(bg.grad.cpu(), bc.grad) < 1e-5
    # forward-only path (validation): same token NLLs, no gradient buffers
    tok2, _, dh2, _, _ = ops.proj_nll_raw(dev(h), dev(W), dev(b), dev(labels), None, 0, False)
    assert dh2 is None and rel_err(tok2.cpu(), tok_c) < 1e-5
    red = ops.nll_reduce(tok2, dev(labels), 0).cpu()
    assert abs(float(red[0]) - float(tok_c.sum())) < 1e-4 * abs(float(tok_c.sum())) and float(red[1]) == cnt


def test_forward_only_projection_never_writes_logits(ops):
    """with_grad = 0 (validation / GEN-side callers): the logits GEMM's online-softmax epilogue keeps (max, sum exp) per
    row per 128-column tile + the label's logit, so the workspace is O(rows * V/128) floats -- not O(rows * V) -- and
    lse / token NLL still equal log_softmax + nll_loss (decoders.py:165-166, criterions.py:65-68).  Odd N / V / H,
    an unaligned h view (scalar-load kernel variant), PAD and out-of-range labels, bitwise run-to-run equality."""
    from laed_amd._lib import lib
    for (N, H, V, unaligned) in ((1170, 128, 1003, False), (300, 72, 50000, False), (257, 37, 8203, True), (1, 16, 5, False)):
        need = int(lib.laed_proj_nll_fwd_workspace(N, V))
        tiles = (V + 127) // 128
        assert need == N * (2 * tiles + 1)
        assert need * 20 < N * V or V < 2000                 # ~V/64 floats per row instead of V
        h, W, b = rnd(N, H, seed=1), rnd(V, H, seed=2, scale=0.3), rnd(V, seed=3, scale=0.1)
        labels = torch.randint(1, V, (N,), generator=torch.Generator().manual_seed(4))
        labels[::7] = 0
        if N > 20:
            labels[5], labels[11] = V + 3, -2                # outside [0,V): ignored like PAD (F.nll_loss would raise)
        ok = (labels > 0) & (labels < V)
        logits = h.double() @ W.double().t() + b.double()
        lse_c = torch.logsumexp(logits, dim=1)
        tok_c = (lse_c - logits.gather(1, labels.clamp(0, V - 1).view(-1, 1)).view(-1)) * ok
        hd = dev(h)
        if unaligned:                                         # H = 37: rows are not 16-byte aligned -> VEC = false variant
            assert (hd.data_ptr() + 4 * H) % 16 != 0
        ops._WS.clear()
        tok, lse, dh, dW, db = ops.proj_nll_raw(hd, dev(W), dev(b), dev(labels), None, 0, False)
        assert dh is None and dW is None and db is None
        held = [t.numel() for k, t in ops._WS.items() if k[0].startswith("proj")]
        assert held == [max(4, need)], (held, need)           # the only projection scratch this call made
        assert rel_err(lse.cpu(), lse_c) < 1e-6
        assert float((tok.cpu().double() - tok_c).abs().max()) <= 2e-5 * float(tok_c.abs().max() + 1.0)
        assert float(tok.cpu()[~ok].abs().max()) == 0.0
        tok2, lse2, _, _, _ = ops.proj_nll_raw(hd, dev(W), dev(b), dev(labels), None, 0, False)
        assert torch.equal(tok, tok2) and torch.equal(lse, lse2)


def test_proj_nll_bf16x3_mode(ops):
    """The projection+NLL pipeline under the opt-in split-bf16 GEMM mode (all three products as x W^T on transposed
    fp32 copies): same answers as the CPU reference to fp32-grade tolerance, two chunks, PAD rows, odd V."""
    N, H, V, chunk = 3000, 160, 8203, 2048
    h, W, b = rnd(N, H, seed=1), rnd(V, H, seed=2, scale=0.3), rnd(V, seed=3, scale=0.1)
    labels = torch.randint(1, V, (N,), generator=torch.Generator().manual_seed(4))
    labels[::7] = 0
    cnt = float((labels != 0).sum())
    hc, Wc, bc = (t.clone().requires_grad_(True) for t in (h, W, b))
    logp = torch.log_softmax(hc @ Wc.t() + bc, dim=1)
    torch.nn.functional.nll_loss(logp, labels, ignore_index=0).backward()
    tok_c = -logp.detach().gather(1, labels.view(-1, 1)).view(-1) * (labels != 0)
    try:
        ops.set_gemm_mode("bf16x3")
        tok, lse, dh, dW, db = ops.proj_nll_raw(dev(h), dev(W), dev(b), dev(labels), dev(torch.full((N,), 1.0 / cnt)), 0,
                                                True, chunk)
    finally:
        ops.set_gemm_mode("fp32")
    assert rel_err(tok.cpu(), tok_c) < 2e-5
    assert rel_err(dh.cpu(), hc.grad) < 5e-5 and rel_err(dW.cpu(), Wc.grad) < 5e-5 and rel_err(db.cpu(), bc.grad) < 1e-5


def test_embed_gather_scatter(ops):
    V, E, rows = 57, 24, 200
    table = rnd(V, E, seed=1)
    table[0] = 0
    ids = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(2))
    keep = (torch.rand(rows, E, generator=torch.Generator().manual_seed(3)) >= 0.3)
    w = rnd(rows, E, seed=4)
    for use_keep in (False, True):
        tc = table.clone().requires_grad_(True)
        x = torch.nn.functional.embedding(ids, tc, padding_idx=0)
        if use_keep:
            x = x * (keep.float() / (1.0 - 0.3))
        (x * w).sum().backward()
        tg = dev(table).requires_grad_(True)
        kd = dev(keep.to(torch.uint8)) if use_keep else None
        xg = ops.EmbedFunction.apply(tg, dev(ids), kd, 0.3 if use_keep else 0.0, 0)
        assert rel_err(xg.detach().cpu(), x.detach()) < 1e-6
        (xg * dev(w)).sum().backward()
        assert rel_err(tg.grad.cpu(), tc.grad) < 1e-5
        assert float(tg.grad[0].abs().sum()) == 0.0
    # E not a multiple of 4 -> scalar path
    t2 = rnd(9, 6, seed=7)
    i2 = torch.tensor([3, 0, 8, 8, 1])
    assert torch.equal(ops.EmbedFunction.apply(dev(t2), dev(i2), None, 0.0, 0).cpu(), t2[i2])


@pytest.mark.parametrize("rows,V,E,use_keep", [(1500, 40, 24, True), (4096, 3000, 8, False), (4097, 97, 20, True),
                                               (20000, 500, 200, True), (163840, 50000, 16, False), (5000, 3, 7, False)])
def test_embed_scatter_deterministic(ops, rows, V, E, use_keep):
    """Dense embedding gradient without float atomics: equals index_add in float64 and is bitwise repeatable, under
    Zipf-like id repetition, PAD / out-of-range ids and a dropout keep mask; every chunk / run / partial case occurs."""
    from laed_amd._lib import lib, ptr, check, stream
    g = torch.Generator().manual_seed(rows + V)
    w = 1.0 / torch.arange(1, V + 1, dtype=torch.float64)
    ids = torch.multinomial(w / w.sum(), rows, replacement=True, generator=g)
    ids[::97] = 0                                   # PAD rows
    ids[5::301] = V + 3                             # out-of-range ids are skipped too
    dout = rnd(rows, E, seed=11)
    keep = (torch.rand(rows, E, generator=g) >= 0.3) if use_keep else None
    p = 0.3 if use_keep else 0.0
    ref = torch.zeros(V, E, dtype=torch.float64)
    ok = (ids != 0) & (ids < V)
    src = dout.double() * (keep.double() / (1.0 - p) if use_keep else 1.0)
    ref.index_add_(0, ids[ok], src[ok])
    outs = []
    for _ in range(2):
        dw = dev(torch.ones(V, E))                  # accumulates into existing contents
        ws_b = int(lib.laed_embed_scatter_workspace(rows, V, E))
        assert (ws_b == 0) == (rows <= 4096)
        ws = torch.empty((ws_b + 3) // 4, device="cuda") if ws_b else None
        kd = dev(keep.to(torch.uint8)) if use_keep else None
        idd, dd = dev(ids), dev(dout)               # keep the device tensors alive across the launch
        check(lib.laed_embed_scatter_add(ptr(idd, torch.int64), rows, ptr(dd), V, E,
                                         ptr(kd, torch.uint8) if use_keep else None, p, 0, ptr(dw), ptr(ws), ws_b, stream()), "scatter")
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[0].double() - 1.0, ref) < 2e-6
    assert float((outs[0][0] - 1.0).abs().max()) == 0.0


def test_rng_fills(ops):
    u = ops.uniform(100000, 1234, 0, "cuda").cpu()
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 0.01
    # counter-based: a shard generated with an offset equals the slice of the full stream (GPU-count invariance)
    u2 = ops.uniform(1000, 1234, 5000, "cuda").cpu()
    assert torch.equal(u2, u[5000:6000])
    keep = ops.dropout_keep_mask(200000, 0.3, 99, 0, "cuda").cpu().float()
    assert abs(float(keep.mean()) - 0.7) < 0.01
    assert torch.equal(ops.dropout_keep_mask(100, 0.3, 99, 777, "cuda").cpu().float(), keep[777:877])


def test_adam(ops, O):
    n = 10007
    p, g = rnd(n, seed=1), rnd(n, seed=2, scale=0.01)
    m, v = torch.zeros(n), torch.zeros(n)
    pg, mg, vg = dev(p), dev(m), dev(v)
    for step in range(1, 6):
        gs = g * (1.0 + 0.1 * step)
        O.adam_step(p, gs, m, v, step)
        ops.adam_step(pg, dev(gs), mg, vg, step, 1e-3)
    assert rel_err(pg.cpu(), p) < 1e-6 and rel_err(mg.cpu(), m) < 1e-6 and rel_err(vg.cpu(), v) < 1e-6


def test_adam_multi_tensor_equals_single_tensor_launches(ops):
    """laed_adam_step_multi: 37 tensors (three launches of <= 16 descriptors), sizes from 1 to 300 001 elements, odd
    sizes, an empty tensor and 4-byte-aligned views (scalar path) -- bit-identical to one laed_adam_step per tensor."""
    sizes = [1, 3, 4, 5, 255, 256, 257, 1024, 4099, 300001, 0, 77] + [1000 + 13 * i for i in range(25)]
    base = [dev(rnd(n + 1, seed=10 + i)) for i, n in enumerate(sizes)]
    mk = lambda i, n, seed, scale=1.0: (dev(rnd(n + 1, seed=seed + i, scale=scale))[1:] if i % 5 == 2
                                        else dev(rnd(n, seed=seed + i, scale=scale)))       # every fifth one unaligned
    P = [mk(i, n, 100) for i, n in enumerate(sizes)]
    G = [mk(i, n, 200, 0.01) for i, n in enumerate(sizes)]
    M = [torch.zeros_like(t) for t in P]
    V_ = [torch.zeros_like(t) for t in P]
    P2, M2, V2 = [t.clone() for t in P], [t.clone() for t in M], [t.clone() for t in V_]
    for step in (1, 2, 3):
        ops.adam_step_multi([t for t in P], G, M, V_, step, 1e-3)
        for p, g, m, v in zip(P2, G, M2, V2):
            if p.numel():
                ops.adam_step(p, g, m, v, step, 1e-3)
    for a, b in zip(P + M + V_, P2 + M2 + V2):
        assert torch.equal(a, b)
    from laed_amd._lib import LaedError
    with pytest.raises(LaedError):
        ops.adam_step_multi([P[3]], [G[4]], [M[3]], [V_[3]], 1, 1e-3)        # size mismatch
    with pytest.raises(LaedError):
        ops.adam_step_multi([P[3].cpu()], [G[3]], [M[3]], [V_[3]], 1, 1e-3)  # no CPU path


def test_greedy_decode_argmax_from_the_gemm_epilogue(ops):
    """Greedy decode at >= 512 output tiles takes its symbols from the projection's arg-max epilogue (no [B][V] logits):
    every step must equal laed_proj_argmax (logits GEMM + laed_argmax_rows) on that step's hidden state, including
    exact ties (two identical vocabulary rows: the lower index wins) and the ragged last column tile (V % 128 != 0)."""
    B, E, H, V, L = 300, 24, 96, 33001, 6                     # 3 x 258 = 774 tiles of 128 x 128
    emb, w_ih, w_hh = dev(rnd(V, E, seed=1)), dev(rnd(3 * H, E, seed=2, scale=0.3)), dev(rnd(3 * H, H, seed=3, scale=0.3))
    b_ih, b_hh = dev(rnd(3 * H, seed=4, scale=0.1)), dev(rnd(3 * H, seed=5, scale=0.1))
    W, b = rnd(V, H, seed=6, scale=0.5), rnd(V, seed=7, scale=0.1)
    W[32999] = W[17]                                          # an exact tie ...
    b[32999] = b[17] = 0.0
    W[17] *= 3.0                                              # ... that wins many rows
    W[32999] *= 3.0
    Wd, bd = dev(W), dev(b)
    h0 = dev(rnd(B, H, seed=8))
    ids, hbuf = ops.decode_greedy(emb, w_ih, w_hh, b_ih, b_hh, Wd, bd, h0, L, 2)
    n_tie = 0
    for t in range(L):
        ref = ops.proj_argmax(hbuf[t + 1].contiguous(), Wd, bd)
        assert torch.equal(ids[t], ref), t
        n_tie += int((ref == 17).sum())
        assert int((ref == 32999).sum()) == 0
    assert n_tie > 0


def test_argmax_logsoftmax_greedy(ops, O):
    N, V = 37, 1003
    logits = rnd(N, V, seed=1, scale=4.0)
    logits[5, 100] = logits[5, 900] = 9.0                 # tie -> lowest index
    buf = dev(torch.zeros(N, ops.proj_ldl(V)))
    buf[:, :V] = dev(logits)
    from laed_amd._lib import check, lib, ptr, stream
    ids = torch.empty(N, device="cuda", dtype=torch.int64)
    check(lib.laed_argmax_rows(ptr(buf), N, V, buf.shape[1], ptr(ids, torch.int64), stream()), "argmax")
    assert np.array_equal(ids.cpu().numpy(), logits.argmax(dim=1).numpy()) and int(ids[5]) == 100
    ops.log_softmax_rows_(buf, V)
    assert rel_err(buf[:, :V].cpu(), torch.log_softmax(logits, dim=1)) < 2e-6
    # greedy decode loop vs the oracle
    B, E, H, V2, L = 5, 12, 32, 41, 9
    P = {"rnn.weight_ih_l0": rnd(3 * H, E, seed=2, scale=0.3), "rnn.weight_hh_l0": rnd(3 * H, H, seed=3, scale=0.3),
         "rnn.bias_ih_l0": rnd(3 * H, seed=4, scale=0.3), "rnn.bias_hh_l0": rnd(3 * H, seed=5, scale=0.3),
         "project.weight": rnd(V2, H, seed=6), "project.bias": rnd(V2, seed=7)}
    emb, h0 = rnd(V2, E, seed=8), rnd(B, H, seed=9)
    seq_c, _, logp_c = O.decoder_greedy(P, "", emb, h0, L, 2, 3)
    ids_g, hbuf = ops.decode_greedy(dev(emb), dev(P["rnn.weight_ih_l0"]), dev(P["rnn.weight_hh_l0"]),
                                    dev(P["rnn.bias_ih_l0"]), dev(P["rnn.bias_hh_l0"]), dev(P["project.weight"]),
                                    dev(P["project.bias"]), dev(h0), L, 2)
    top2 = logp_c.topk(2, dim=2)[0]
    assert float((top2[..., 0] - top2[..., 1]).min()) > 1e-4   # fixture has no near-ties
    assert np.array_equal(ids_g.t().cpu().numpy(), seq_c.numpy())


def test_single_step_entry_points_match_the_decode_loop(ops, O):
    """laed_gru_cell_fwd + laed_proj_argmax driven from the host, one step at a time, reproduce laed_gru_decode_greedy's
    ids (and so the oracle's greedy decode) exactly."""
    for (B, E, H, V, L) in ((5, 12, 32, 57, 7), (70, 20, 48, 301, 5)):
        emb, w_ih, w_hh = rnd(V, E, seed=1), rnd(3 * H, E, seed=2, scale=0.4), rnd(3 * H, H, seed=3, scale=0.4)
        b_ih, b_hh = rnd(3 * H, seed=4, scale=0.1), rnd(3 * H, seed=5, scale=0.1)
        pw, pb, h0 = rnd(V, H, seed=6), rnd(V, seed=7, scale=0.1), rnd(B, H, seed=8)
        d = [dev(t) for t in (emb, w_ih, w_hh, b_ih, b_hh, pw, pb, h0)]
        ids_loop, hbuf = ops.decode_greedy(*d, L, 2)
        h, ids, steps = d[7], torch.full((B,), 2, device="cuda", dtype=torch.int64), []
        for _ in range(L):
            x = ops.EmbedFunction.apply(d[0], ids, None, 0.0, -1)
            h = ops.gru_cell(x, d[1], d[2], d[3], d[4], h)
            ids = ops.proj_argmax(h, d[5], d[6])
            steps.append(ids)
        assert np.array_equal(torch.stack(steps).cpu().numpy(), ids_loop.cpu().numpy())
        assert torch.equal(h, hbuf[L])


def test_single_use_tapes_refuse_a_second_backward():
    """GRU / fused projection rewrite their saved buffers in backward; a retained graph must not be run twice."""
    from laed_amd import ops
    x = torch.randn(3, 4, 6, device="cuda", requires_grad=True)
    w_ih, w_hh = torch.randn(24, 6, device="cuda", requires_grad=True), torch.randn(24, 8, device="cuda", requires_grad=True)
    b = torch.zeros(24, device="cuda", requires_grad=True)
    out, hT = ops.GRUFunction.apply(x, w_ih, w_hh, b, b, None)
    loss = out.sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="single-use"):
        loss.backward()
    h = torch.randn(10, 8, device="cuda", requires_grad=True)
    W, bb = torch.randn(17, 8, device="cuda", requires_grad=True), torch.zeros(17, device="cuda", requires_grad=True)
    labels = torch.randint(1, 17, (10,), device="cuda")
    l2, _ = ops.ProjNLLFunction.apply(h, W, bb, labels, torch.full((10,), 0.1, device="cuda"), 0)
    l2.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="single-use"):
        l2.backward()
