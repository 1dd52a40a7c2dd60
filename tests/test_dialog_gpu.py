"""AE-ED / ST-ED on the HIP engine vs golden vectors produced by the reference (tests/golden/make_goldens_dialog.py):
phase A, the freeze flip, phase B (fresh optimizer), eval losses and greedy GEN -- identical weights, inputs and
injected Gumbel noise.  Bar: loss terms 1e-4 relative, greedy token ids exact.
"""
import numpy as np
import pytest
import torch

from helpers import golden_config, load_golden, rel_err

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-4
# d loss / d query.bias is exactly 0 in real arithmetic (softmax over T is shift invariant); autograd returns fp32
# round-off that Adam turns into +-lr moves of a parameter no output depends on (see the oracle test).
NOISE_ONLY = ("utt_encoder.query.bias",)


class Corpus(object):
    def __init__(self, V):
        self.vocab = ['<pad>', '<unk>', '<s>', '</s>'] + ['w%d' % i for i in range(4, V)]
        self.rev_vocab = {w: i for i, w in enumerate(self.vocab)}


def build(G, prefix="sd/"):
    from laed_amd.models.dialog_models import AeED, StED
    cfg = golden_config(G)
    cfg.use_gpu = True
    kind = str(G["meta/kind"])
    model = (AeED if kind == "aeed" else StED)(Corpus(int(G["config/vocab_size"])), cfg)
    assert list(model.state_dict().keys()) == [str(k) for k in G["meta/state_dict_keys"]]
    named = dict(model.named_parameters())
    sd = {}
    for k in model.state_dict().keys():
        src = prefix + k if prefix + k in G.files else "sd/" + k
        sd[k] = torch.from_numpy(np.array(G[src]))
    if prefix != "sd/":                       # aliases of shared tables follow the table they alias
        for k in sd:
            if k not in named:
                owner = "x_embedding.weight" if k.startswith(("x_decoder", "prev_decoder", "next_decoder")) \
                    else "embedding.weight"
                sd[k] = sd[owner]
    model.load_state_dict(sd)
    return model.cuda(), cfg, kind


def inject_u(model, u):
    ud = torch.from_numpy(np.array(u)).cuda()
    model.cat_connector.sample_uniform = lambda logits: ud


def close(a, b, rtol=LOSS_RTOL, atol=2e-6):
    return abs(float(a) - float(b)) <= atol + rtol * abs(float(b))


def check_losses(G, tag, loss, keys):
    for k in keys:
        if tag + k in G.files:
            tol = 5e-3 if k == "mi" else LOSS_RTOL     # mi: difference of two nearly equal entropies, logged only
            assert close(loss[k], G[tag + k], rtol=tol), (tag, k, float(loss[k]), float(G[tag + k]))


def check_grads(G, tag, model, tol=3e-4):
    for n, p in model.named_parameters():
        g = G[tag + n]
        if n in NOISE_ONLY:
            continue
        if g.size == 0:
            assert p.grad is None, (tag, n)
        else:
            assert p.grad is not None, (tag, n)
            assert rel_err(p.grad.cpu(), g) < tol, (tag, n, rel_err(p.grad.cpu(), g))


def check_params(G, tag, model, atol):
    for n, p in model.named_parameters():
        if n in NOISE_ONLY:
            continue
        d = float((p.detach().cpu() - torch.from_numpy(np.array(G[tag + n]))).abs().max())
        assert d < atol, (tag, n, d)


@pytest.mark.parametrize("name", ["aeed_tiny", "sted_tiny", "aeed_mid", "sted_mid"])
def test_dialog_model_two_phase_training_vs_reference(name):
    from laed_amd.enc2dec.decoders import TEACH_FORCE
    G = load_golden(name)
    model, cfg, kind = build(G)
    keys = [str(k) for k in G["meta/loss_keys"]]
    feed = {k[3:]: G[k] for k in G.files if k.startswith("in/") and not k.startswith("in/u_")}
    U = G["in/u_steps"]
    model.train()

    # ---- phase A: forward / backward
    inject_u(model, U[0])
    loss = model(feed, mode=TEACH_FORCE, return_latent=True)
    check_losses(G, "A/", loss, keys)
    for k in ("log_py", "log_qy", "dec_init_state"):
        assert rel_err(loss[k].detach().cpu(), G["A/" + k]) < 1e-5, k
    assert np.array_equal(loss["y_ids"].cpu().numpy(), G["A/y_ids"])
    model.zero_grad()
    model.backward(0, loss)
    check_grads(G, "A_grad/", model)

    # ---- 3 Adam steps (main.py:104-113 order)
    opt = model.get_optimizer(cfg)
    for s in range(3):
        inject_u(model, U[s])
        opt.zero_grad()
        loss = model(feed, mode=TEACH_FORCE)
        for j, k in enumerate(keys):
            assert close(loss[k], G["A/step_losses"][s, j], rtol=2e-4 if k != "mi" else 2e-2, atol=5e-6), (s, k)
        model.backward(s, loss)
        opt.step()
    check_params(G, "A_p3/", model, 1e-4)

    # ---- flip step: phase-A forward, backward past freeze_step freezes the recognition side first
    inject_u(model, U[3])
    opt.zero_grad()
    loss = model(feed, mode=TEACH_FORCE)
    check_losses(G, "flip/", loss, keys)
    model.backward(cfg.freeze_step + 1, loss)
    assert model.kl_w == 1.0 and model.flush_valid
    assert [int(p.requires_grad) for _, p in model.named_parameters()] == [int(r) for r in G["flip/requires_grad"]]
    check_grads(G, "flip_grad/", model)
    opt.step()
    check_params(G, "flip_p/", model, 1.5e-4)

    # ---- phase B: greedy recognition sample, optimizer re-created on flush_valid (main.py:106-110)
    batch_cnt = cfg.freeze_step + 2
    for s in range(3):
        opt.zero_grad()
        loss = model(feed, mode=TEACH_FORCE, return_latent=(s == 0))
        if model.flush_valid:
            model.flush_valid = False
            opt = model.get_optimizer(cfg)
        if s == 0:
            check_losses(G, "B/", loss, keys)
            assert np.array_equal(loss["y_ids"].cpu().numpy(), G["B/y_ids"])
            assert rel_err(loss["log_py"].detach().cpu(), G["B/log_py"]) < 1e-5
            assert rel_err(loss["dec_init_state"].detach().cpu(), G["B/dec_init_state"]) < 1e-5
        for j, k in enumerate(keys):
            assert close(loss[k], G["B/step_losses"][s, j], rtol=3e-4 if k != "mi" else 2e-2, atol=5e-6), (s, k)
        model.zero_grad()
        model.backward(batch_cnt + s, loss)
        if s == 0:
            check_grads(G, "B_grad/", model)
        opt.step()
    check_params(G, "B_p3/", model, 2e-4)


@pytest.mark.parametrize("name", ["aeed_tiny", "sted_tiny", "aeed_mid", "sted_mid"])
def test_dialog_model_eval_and_greedy_vs_reference(name):
    from laed_amd.enc2dec.decoders import GEN, TEACH_FORCE, DecoderRNN
    G = load_golden(name)
    model, cfg, kind = build(G, prefix="B_p3/")          # the weights the REFERENCE reached (exact-id requirement)
    keys = [str(k) for k in G["meta/loss_keys"]]
    feed = {k[3:]: G[k] for k in G.files if k.startswith("in/") and not k.startswith("in/u_")}
    model.kl_w = 1.0
    model.eval()
    inject_u(model, G["in/u_gen"])
    with torch.no_grad():
        loss = model(feed, mode=TEACH_FORCE)
        check_losses(G, "eval/", loss, keys)
        assert close(model.model_sel_loss(loss, cfg.freeze_step + 5), G["eval/model_sel_loss"])
        ret, labels = model(feed, mode=GEN, gen_type='greedy')
    assert float(G["gen/margin"].min()) > 1e-4
    seq = torch.cat(ret[DecoderRNN.KEY_SEQUENCE], dim=1).cpu().numpy()
    assert np.array_equal(seq, G["gen/sequence"])
    assert list(ret[DecoderRNN.KEY_LENGTH]) == list(G["gen/length"])
    assert np.array_equal(ret[DecoderRNN.KEY_LATENT].cpu().numpy(), G["gen/y_ids"])
    assert np.array_equal(ret[DecoderRNN.KEY_RECOG_LATENT].cpu().numpy(), G["gen/qy_ids"])
    assert rel_err(ret[DecoderRNN.KEY_POLICY].cpu(), G["gen/log_py"]) < 1e-5
    assert np.array_equal(labels.cpu().numpy(), G["gen/labels"])


@pytest.mark.parametrize("N,H,V,E,chunk,bare", [(300, 40, 1003, 24, 128, False),
                                                (1170, 256, 10004, 200, 1170, False),    # AE-ED / ST-ED at B = 30: K = V splits
                                                (1170, 256, 10004, 200, 600, True)])     # bare tiles: the unsplit products
def test_softmax_embed_kernel_vs_torch(N, H, V, E, chunk, bare):
    """laed_softmax_embed_{fwd,bwd} through the C ABI vs fp64 torch-CPU math: odd sizes with several row chunks, and the
    reference's batch shape with and without the split-K surplus in the workspace."""
    import ctypes
    from laed_amd import ops
    from laed_amd._lib import check, lib, ptr, stream
    rs = np.random.RandomState(5)
    h = rs.randn(N, H).astype(np.float32)
    W = (rs.randn(V, H) * 0.3).astype(np.float32)
    b = rs.randn(V).astype(np.float32)
    We = rs.randn(V, E).astype(np.float32)
    dout = rs.randn(N, E).astype(np.float32)
    t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in dict(h=h, W=W, b=b, We=We).items()}
    ref = torch.softmax(t["h"] @ t["W"].t() + t["b"], dim=1) @ t["We"]
    ref.backward(torch.tensor(dout, dtype=torch.float64))
    d = {k: torch.from_numpy(v).cuda() for k, v in dict(h=h, W=W, b=b, We=We, dout=dout).items()}
    ws_n = int(lib.laed_softmax_embed_workspace(V, chunk, 1))
    ldl = int(lib.laed_proj_nll_ldl(V))
    assert ws_n > 2 * chunk * ldl
    if bare:
        ws_n = 2 * chunk * ldl
    ws = torch.empty(ws_n, device="cuda")
    lse, out = torch.empty(N, device="cuda"), torch.empty((N, E), device="cuda")
    check(lib.laed_softmax_embed_fwd(ptr(d["h"]), ptr(d["W"]), ptr(d["b"]), ptr(d["We"]), N, H, V, E, ptr(ws), ws_n, chunk,
                                     ptr(lse), ptr(out), stream()), "fwd")
    assert rel_err(out.cpu(), ref.detach()) < 1e-5
    dh, dW, db, dWe = (torch.empty_like(d["h"]), torch.empty_like(d["W"]), torch.empty(V, device="cuda"),
                       torch.empty_like(d["We"]))
    check(lib.laed_softmax_embed_bwd(ptr(d["h"]), ptr(d["W"]), ptr(d["b"]), ptr(d["We"]), ptr(lse), ptr(d["dout"]), N, H, V, E,
                                     ptr(ws), ws_n, chunk, ptr(dh), ptr(dW), ptr(db), ptr(dWe), 0, stream()), "bwd")
    for got, key in ((dh, "h"), (dW, "W"), (db, "b"), (dWe, "We")):
        assert rel_err(got.cpu(), t[key].grad) < 2e-5, key
    # autograd wrapper gives the same numbers
    p = {k: d[k].clone().requires_grad_(True) for k in ("h", "W", "b", "We")}
    o2 = ops.SoftmaxEmbedFunction.apply(p["h"], p["W"], p["b"], p["We"])
    o2.backward(d["dout"])
    assert rel_err(o2.detach().cpu(), ref.detach()) < 1e-5
    for key in ("h", "W", "b", "We"):
        assert rel_err(p[key].grad.cpu(), t[key].grad) < 2e-5, key


@pytest.mark.parametrize("name", ["aeed_tiny", "sted_tiny"])
def test_backward_rooted_at_the_loss_terms_equals_backward_of_valid_loss(name):
    """AeED / StED.loss_terms: in the pretraining phase and in the encoder-decoder phase the backward pass rooted at the
    (weighted) terms gives bit-identical gradients to valid_loss(...).backward() and the terms add up to valid_loss; the
    step that changes phase returns None (valid_loss must run there: it freezes the recognition side)."""
    from laed_amd.enc2dec.decoders import TEACH_FORCE
    G = load_golden(name)
    feed = {k[3:]: G[k] for k in G.files if k.startswith("in/") and not k.startswith("in/u_")}

    def weighted_sum(terms):
        return float(sum((t[0] * t[1]) if isinstance(t, tuple) else t for t in terms).detach())

    for phase in ("A", "B"):
        grads = []
        for rooted in (True, False):
            model, cfg, kind = build(G)
            model.train()
            inject_u(model, G["in/u_steps"][0])
            cnt = 0
            if phase == "B":                                   # go through the flip the way the engine does
                assert model.loss_terms(model(feed, mode=TEACH_FORCE), cfg.freeze_step + 1) is None
                model.backward(cfg.freeze_step + 1, model(feed, mode=TEACH_FORCE))
                assert model.kl_w == 1.0
                model.zero_grad()
                cnt = cfg.freeze_step + 2
            loss = model(feed, mode=TEACH_FORCE)
            terms = model.loss_terms(loss, cnt)
            assert terms is not None and len(terms) >= 2
            assert abs(weighted_sum(terms) - float(model.valid_loss(loss, cnt).detach())) <= 1e-6 * abs(weighted_sum(terms))
            if rooted:
                model.backward(cnt, loss)
            else:
                model.valid_loss(loss, cnt).backward()
            torch.cuda.synchronize()
            grads.append({k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
        assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 5, phase
        for k in grads[0]:
            assert torch.equal(grads[0][k], grads[1][k]), (phase, k)
