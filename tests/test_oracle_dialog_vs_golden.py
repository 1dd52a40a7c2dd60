"""Pin oracle/dialog_oracle.py (AE-ED / ST-ED restatement) to golden vectors produced by running the reference
(tests/golden/make_goldens_dialog.py): phase A, the freeze flip, phase B, eval and greedy GEN.  CPU only."""
import numpy as np
import pytest
import torch

from helpers import golden_config, load_golden, rel_err
from oracle import dialog_oracle as D
from oracle import laed_oracle as O

ALIASES = ("x_decoder.embedding.weight", "prev_decoder.embedding.weight", "next_decoder.embedding.weight",
           "decoder.embedding.weight", "utt_encoder.embedding.weight")


def dialog_params(G, prefix="sd/"):
    P = {}
    for k in G.files:
        if k.startswith(prefix) and k[len(prefix):] not in ALIASES:
            P[k[len(prefix):]] = torch.from_numpy(np.array(G[k])).requires_grad_(True)
    return P


def close(a, b, rtol=1e-5, atol=1e-6):
    return abs(float(a) - float(b)) <= atol + rtol * abs(float(b))


def forward(kind, P, cfg, feed, u, kl_w, training=True):
    fn = D.aeed_forward if kind == "aeed" else D.sted_forward
    return fn(P, cfg, feed, u, kl_w=kl_w, training=training)


def check_losses(G, tag, res, keys):
    for k in keys:
        if tag + k in G.files:
            assert close(res[k], G[tag + k], 2e-5), (tag, k, float(res[k]), float(G[tag + k]))


# d loss / d query.bias is exactly 0 in real arithmetic (softmax over T is shift invariant, encoders.py:123-124); what
# autograd returns is fp32 round-off (~1e-10), which Adam's sign-like first steps turn into +-lr moves of a parameter
# that cannot change any output.  It is excluded from gradient / parameter comparisons.
NOISE_ONLY = ("utt_encoder.query.bias",)


def check_grads(G, tag, P, tol=5e-5):
    for k in P:
        if k in NOISE_ONLY:
            assert P[k].grad is None or float(P[k].grad.abs().max()) < 1e-7
            continue
        g = G[tag + k]
        if g.size == 0:
            assert P[k].grad is None, k
        else:
            assert P[k].grad is not None, k
            assert rel_err(P[k].grad, g) < tol, (tag, k)


class Adam(object):
    """torch.optim.Adam bookkeeping: per-parameter state created at the first step that sees a gradient."""

    def __init__(self, P, keys, lr):
        self.P, self.keys, self.lr, self.state = P, list(keys), lr, {}

    def step(self):
        with torch.no_grad():
            for k in self.keys:
                p = self.P[k]
                if p.grad is None:
                    continue
                st = self.state.setdefault(k, dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
                st["step"] += 1
                O.adam_step(p, p.grad, st["m"], st["v"], st["step"], lr=self.lr)


def zero_grad(P):
    for p in P.values():
        p.grad = None


@pytest.mark.parametrize("name", ["aeed_tiny", "sted_tiny", "aeed_mid", "sted_mid"])
def test_dialog_model_phases(name):
    G = load_golden(name)
    kind = str(G["meta/kind"])
    keys = [str(k) for k in G["meta/loss_keys"]]
    cfg = golden_config(G)
    P = dialog_params(G)
    feed = {k[3:]: G[k] for k in G.files if k.startswith("in/") and not k.startswith("in/u_")}
    U = G["in/u_steps"]
    lr = float(cfg.init_lr)
    checked = [k for k in P if k not in NOISE_ONLY]

    # ---- phase A forward / backward
    res = forward(kind, P, cfg, feed, U[0], 0.0)
    check_losses(G, "A/", res, keys)
    for k in ("log_py", "log_qy", "dec_init_state"):
        assert rel_err(res[k].detach(), G["A/" + k]) < 5e-6, k
    assert np.array_equal(res["y_ids"].numpy(), G["A/y_ids"])
    D.total_loss(kind, cfg, res, False).backward()
    check_grads(G, "A_grad/", P)

    # ---- 3 Adam steps in phase A
    opt = Adam(P, P.keys(), lr)
    for s in range(3):
        zero_grad(P)
        res = forward(kind, P, cfg, feed, U[s], 0.0)
        for j, k in enumerate(keys):
            assert close(res[k], G["A/step_losses"][s, j], 5e-5, 2e-6), (s, k)
        D.total_loss(kind, cfg, res, False).backward()
        opt.step()
    for k in checked:
        assert np.max(np.abs(P[k].detach().numpy() - G["A_p3/" + k])) < 1e-4, k     # 10 % of one lr-sized step

    # ---- flip step: phase-A forward, encoder-decoder loss, old optimizer
    zero_grad(P)
    res = forward(kind, P, cfg, feed, U[3], 0.0)
    check_losses(G, "flip/", res, keys)
    live = [k for k in P if not k.startswith(D.FROZEN_PREFIXES[kind])]
    for k in P:                       # valid_loss freezes BEFORE backward: AccumulateGrad then drops those grads
        P[k].requires_grad_(k in live)
    assert [int(k in live) for k in P] == [int(r) for r in G["flip/requires_grad"]]     # named_parameters() order
    D.total_loss(kind, cfg, res, True).backward()
    check_grads(G, "flip_grad/", P, tol=2e-4)
    opt.step()
    for k in checked:
        assert np.max(np.abs(P[k].detach().numpy() - G["flip_p/" + k])) < 1.5e-4, k

    # ---- phase B: frozen recognition side, greedy latent, fresh optimizer
    opt = Adam(P, live, lr)
    for s in range(3):
        zero_grad(P)
        res = forward(kind, P, cfg, feed, None, 1.0)
        if s == 0:
            check_losses(G, "B/", res, keys)
            assert np.array_equal(res["y_ids"].numpy(), G["B/y_ids"])
            assert rel_err(res["log_py"].detach(), G["B/log_py"]) < 5e-6
            assert rel_err(res["dec_init_state"].detach(), G["B/dec_init_state"]) < 5e-6
        for j, k in enumerate(keys):
            assert close(res[k], G["B/step_losses"][s, j], 1e-4, 5e-6), (s, k)
        D.total_loss(kind, cfg, res, True).backward()
        if s == 0:
            check_grads(G, "B_grad/", P, tol=2e-4)
        opt.step()
    for k in checked:
        assert np.max(np.abs(P[k].detach().numpy() - G["B_p3/" + k])) < 2e-4, k

    # ---- eval + greedy GEN (on the golden's own final parameters, so step round-off does not leak in)
    Pf = dialog_params(G, "B_p3/")
    with torch.no_grad():
        res = forward(kind, Pf, cfg, feed, G["in/u_gen"], 1.0, training=False)
        check_losses(G, "eval/", res, keys)
        assert close(res["pi_err"] + res["adv_err"], G["eval/model_sel_loss"])      # dialog_models.py:184-187
        seq, lengths, logp, y_id, log_py = D.dialog_greedy(Pf, cfg, feed, G["in/u_gen"])
    assert float(G["gen/margin"].min()) > 1e-4
    assert np.array_equal(seq.numpy(), G["gen/sequence"])
    assert list(lengths) == list(G["gen/length"])
    assert np.array_equal(y_id.numpy().reshape(-1), G["gen/y_ids"].reshape(-1))   # reference leaves it [B*y,1] in GEN
    assert rel_err(log_py, G["gen/log_py"]) < 5e-6
