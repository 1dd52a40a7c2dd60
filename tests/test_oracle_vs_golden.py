"""Pin the CPU oracle (oracle/laed_oracle.py) to golden vectors produced by running the
reference itself (tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest
import torch

from helpers import golden_config, golden_params, load_golden, rel_err
from oracle import laed_oracle as O

TOL = 2e-6   # same torch build, same arithmetic, different op grouping -> fp32 round-off only


@pytest.mark.parametrize("name", ["divae_tiny", "divae_mid", "divae_tiny_prior", "divae_tiny_rowkl"])
def test_divae_forward_backward(name):
    """divae_tiny_prior: train_prior=True from a non-uniform log_py (sent_models.py:144-146); divae_tiny_rowkl:
    use_mutual=False, the per-row KL of :150-151."""
    G = load_golden(name)
    cfg = golden_config(G)
    P = golden_params(G, requires_grad=True)
    res = O.divae_forward(P, cfg, G["in/outputs"], G["in/u_steps"][0], training=True)
    for k in ("nll", "reg_kl", "mi", "bpr"):
        assert abs(float(res[k]) - float(G["out/" + k])) <= 1e-6 + 1e-5 * abs(float(G["out/" + k])), k
    assert rel_err(res["log_qy"].detach(), G["out/log_qy"]) < TOL
    assert np.array_equal(res["y_ids"].numpy(), G["out/y_ids"])
    assert rel_err(res["dec_init_state"].detach(), G["out/dec_init_state"]) < TOL
    assert rel_err(res["x_last"].detach(), G["out/x_last"][0]) < TOL
    assert rel_err(res["tok_nll"].detach(), G["out/tok_nll"]) < 5e-6
    assert abs(float(res["total"]) - float(G["out/total_loss"])) < 1e-5
    res["total"].backward()
    for k in O.trainable_keys(P, bool(cfg.train_prior)):
        assert rel_err(P[k].grad, G["grad/" + k]) < 2e-5, k
    if cfg.train_prior:
        assert float(np.abs(G["grad/log_py"]).max()) > 1e-4 and float(res["reg_kl"]) != float(res["bpr"])
    else:
        assert G["grad/log_py"].size == 0 and P["log_py"].grad is None


@pytest.mark.parametrize("name", ["divst_tiny", "divst_mid"])
def test_divst_forward_backward(name):
    G = load_golden(name)
    cfg = golden_config(G)
    P = golden_params(G, requires_grad=True)
    res = O.divst_forward(P, cfg, G["in/outputs"], G["in/prevs"], G["in/nexts"], G["in/u_steps"][0])
    for k in ("prev_nll", "next_nll", "reg_kl", "mi", "b_pr"):
        assert abs(float(res[k]) - float(G["out/" + k])) <= 1e-6 + 1e-5 * abs(float(G["out/" + k])), k
    assert rel_err(res["prev_tok_nll"].detach(), G["out/prev_tok_nll"]) < 5e-6
    assert rel_err(res["next_tok_nll"].detach(), G["out/next_tok_nll"]) < 5e-6
    res["total"].backward()
    for k in O.trainable_keys(P):
        assert rel_err(P[k].grad, G["grad/" + k]) < 2e-5, k


def _train5(G, kind):
    cfg = golden_config(G)
    P = golden_params(G, requires_grad=True)
    keys = O.trainable_keys(P, bool(getattr(cfg, "train_prior", False)))
    m = {k: torch.zeros_like(P[k]) for k in keys}
    v = {k: torch.zeros_like(P[k]) for k in keys}
    losses = []
    out = {}
    for s in range(5):
        for k in keys:
            P[k].grad = None
        if kind == "divae":
            res = O.divae_forward(P, cfg, G["in/outputs"], G["in/u_steps"][s])
            losses.append([float(res[k]) for k in ("nll", "reg_kl", "mi", "bpr")])
        else:
            res = O.divst_forward(P, cfg, G["in/outputs"], G["in/prevs"], G["in/nexts"], G["in/u_steps"][s])
            losses.append([float(res[k]) for k in ("prev_nll", "next_nll", "reg_kl", "mi", "b_pr")])
        res["total"].backward()
        with torch.no_grad():
            for k in keys:
                O.adam_step(P[k], P[k].grad, m[k], v[k], s + 1, lr=float(cfg.init_lr))
        if s in (0, 4):
            out[s + 1] = {k: P[k].detach().clone() for k in keys}
    return cfg, P, np.array(losses), out


@pytest.mark.parametrize("name,kind", [("divae_tiny", "divae"), ("divst_tiny", "divst"), ("divae_mid", "divae"),
                                       ("divae_tiny_prior", "divae"), ("divae_tiny_rowkl", "divae")])
def test_adam_5_steps_and_greedy(name, kind):
    G = load_golden(name)
    cfg, P, losses, snaps = _train5(G, kind)
    # Adam's first steps are sign-like (|dp| ~ lr), so tiny grads amplify round-off: abs tol in units of lr
    for s in (1, 5):
        for k, t in snaps[s].items():
            d = np.max(np.abs(t.numpy() - G["p%d/%s" % (s, k)]))
            assert d < 2e-5, (s, k, d)
    assert np.allclose(losses, G["out/step_losses"], rtol=2e-4, atol=2e-6)
    # greedy decode on the 5-step weights written by the reference (exact ids)
    P5 = golden_params(G, prefix="p5/")
    for k in list(P):
        if k not in P5:
            P5[k] = P["embedding.weight"].detach() if k.endswith("embedding.weight") else P[k].detach()
    for k in list(P5):
        if k.endswith("decoder.embedding.weight"):
            P5[k] = P5["embedding.weight"]
    with torch.no_grad():
        if kind == "divae":
            seq, lens, logp = O.divae_greedy(P5, cfg, G["in/outputs"], G["in/u_gen"])
            assert np.array_equal(seq.numpy(), G["gen/sequence"])
            assert lens == G["gen/length"].tolist()
            ev = O.divae_forward(P5, cfg, G["in/outputs"], G["in/u_gen"], training=False)
            for k in ("nll", "reg_kl", "mi", "bpr"):
                assert abs(float(ev[k]) - float(G["eval/" + k])) <= 1e-6 + 2e-5 * abs(float(G["eval/" + k]))
            assert np.array_equal(ev["y_ids"].numpy(), G["eval/y_ids"])
        else:
            (ps, pl, _), (ns, nl, _) = O.divst_greedy(P5, cfg, G["in/outputs"], G["in/u_gen"])
            assert np.array_equal(ps.numpy(), G["gen/prev_sequence"])
            assert np.array_equal(ns.numpy(), G["gen/next_sequence"])
            assert pl == G["gen/prev_length"].tolist() and nl == G["gen/next_length"].tolist()


def test_dropout_mask_case():
    for name, kind in (("divae_tiny", "divae"), ("divst_tiny", "divst")):
        G = load_golden(name)
        cfg = golden_config(G)
        cfg.dropout = 0.3
        P = golden_params(G, requires_grad=True)
        if kind == "divae":
            res = O.divae_forward(P, cfg, G["in/outputs"], G["in/u_steps"][0], True, G["in/drop_mask0"])
            keys = ("nll", "reg_kl", "mi", "bpr")
        else:
            res = O.divst_forward(P, cfg, G["in/outputs"], G["in/prevs"], G["in/nexts"], G["in/u_steps"][0],
                                  True, (G["in/drop_mask0"], G["in/drop_mask1"]))
            keys = ("prev_nll", "next_nll", "reg_kl", "mi", "b_pr")
        for k in keys:
            assert abs(float(res[k]) - float(G["drop/" + k])) <= 1e-6 + 1e-5 * abs(float(G["drop/" + k])), k
        res["total"].backward()
        for k in G.files:
            if k.startswith("drop_grad/"):
                assert rel_err(P[k[len("drop_grad/"):]].grad, G[k]) < 2e-5, k


def test_modules():
    G = load_golden("modules")
    logits, u = torch.from_numpy(G["gumbel/logits"]), torch.from_numpy(G["gumbel/u"])
    for temp in (1.0, 0.5):
        lt = logits.clone().requires_grad_(True)
        y, ids = O.gumbel_softmax(lt, u, temp, hard=False)
        w = torch.from_numpy(np.arange(60, dtype=np.float32).reshape(12, 5) / 60.0)
        (y * w).sum().backward()
        assert rel_err(y.detach(), G["gumbel/y_t%g" % temp]) < TOL
        assert np.array_equal(ids.numpy(), G["gumbel/ids_t%g" % temp])
        assert rel_err(lt.grad, G["gumbel/dlogits_t%g" % temp]) < 1e-5
    yh, ids = O.gumbel_softmax(logits, u, 1.0, hard=True)
    assert np.array_equal(yh.numpy(), G["gumbel/y_hard"]) and np.array_equal(ids.numpy(), G["gumbel/ids_hard"])

    logp, labels = torch.from_numpy(G["nll/logp"]), torch.from_numpy(G["nll/labels"])
    for avg in (None, "seq", "real_word", "word"):
        assert abs(float(O.nll_entropy(logp, labels, avg)) - float(G["nll/out_%s" % avg])) < 1e-5, avg

    lq, lp = torch.from_numpy(G["catkl/log_qy"]), torch.from_numpy(G["catkl/log_py"])
    assert abs(float(O.cat_kl(lq, lp, 4, True)) - float(G["catkl/unit"])) < 1e-6
    assert abs(float(O.cat_kl(lq, lp, 4, False)) - float(G["catkl/batch"])) < 1e-6
    assert abs(float(O.entropy(lq, None, True)) - float(G["entropy/unit"])) < 1e-6
    assert abs(float(O.entropy(lq, 4, False)) - float(G["entropy/batch"])) < 1e-6

    for bi in (0, 1):
        tag = "enc_bi%d/" % bi
        sd = {k[len(tag + "sd/"):]: torch.from_numpy(G[k]) for k in G.files if k.startswith(tag + "sd/")}
        x, h0 = torch.from_numpy(G[tag + "x"]), torch.from_numpy(G[tag + "h0"])
        for use_h0, sfx in ((False, ""), (True, "_h0")):
            outs, hids = [], []
            for d, rev in enumerate(["", "_reverse"][: bi + 1]):
                hinit = h0[d] if use_h0 else torch.zeros_like(h0[d])
                o, h = O.gru_seq(x, hinit, sd["rnn.weight_ih_l0" + rev], sd["rnn.weight_hh_l0" + rev],
                                 sd["rnn.bias_ih_l0" + rev], sd["rnn.bias_hh_l0" + rev], reverse=bool(d))
                outs.append(o)
                hids.append(h)
            assert rel_err(torch.cat(outs, dim=2), G[tag + "out" + sfx]) < TOL
            assert rel_err(torch.stack(hids, dim=0), G[tag + "hid" + sfx]) < TOL

    P = {k[len("dec/sd/"):]: torch.from_numpy(G[k]) for k in G.files if k.startswith("dec/sd/")}
    logp, hT = O.decoder_teacher(P, "", P["embedding.weight"], torch.from_numpy(G["dec/ids"]),
                                 torch.from_numpy(G["dec/h0"])[0])
    assert rel_err(logp, G["dec/logp"]) < TOL and rel_err(hT, G["dec/hid"][0]) < TOL
    seq, lens, glp = O.decoder_greedy(P, "", P["embedding.weight"], torch.from_numpy(G["dec/h0"])[0], 6, 2, 3)
    assert np.array_equal(seq.numpy(), G["dec/gen_seq"]) and lens == G["dec/gen_len"].tolist()
    assert rel_err(glp, G["dec/gen_logp"]) < TOL

    for has_bias in (0, 1):
        tag = "linconn_b%d/" % has_bias
        b = torch.from_numpy(G[tag + "b"]) if has_bias else None
        out = torch.nn.functional.linear(torch.from_numpy(G[tag + "x"]), torch.from_numpy(G[tag + "w"]), b)
        assert rel_err(out.unsqueeze(0), G[tag + "out"]) < TOL


def test_module_form_matches_functional():
    """The nn.Module oracle (what bench.py times as the CPU baseline) == functional oracle."""
    G = load_golden("divae_tiny")
    cfg = golden_config(G)
    P = golden_params(G)
    model = O.OracleDiVAE(int(G["config/vocab_size"]), cfg)
    assert list(model.state_dict().keys()) == [str(k) for k in G["meta/state_dict_keys"]]
    model.load_state_dict(P)
    model.train()
    res = model(G["in/outputs"], G["in/u_steps"][0])
    for k in ("nll", "reg_kl", "mi", "bpr"):
        assert abs(float(res[k]) - float(G["out/" + k])) <= 1e-6 + 1e-5 * abs(float(G["out/" + k])), k


def test_sample_decoding_gumbel_max():
    """gen_type='sample' (tests/golden/make_goldens_sample.py: the reference decoder with a replayed uniform stream)."""
    G = load_golden("sample")
    P = {"dec." + k[3:]: torch.from_numpy(np.array(G[k])) for k in G.files if k.startswith("sd/")}
    assert float(G["margin"].min()) > 1e-3
    L = G["u"].shape[0]
    with torch.no_grad():
        seq, lengths, hT = O.decoder_sample(P, "dec.", P["dec.embedding.weight"], torch.from_numpy(G["h0"][0]), L, 2, 3,
                                            G["u"])
    assert np.array_equal(seq.numpy(), G["sequence"])
    assert list(lengths) == list(G["length"])
    assert rel_err(hT, G["hid"][0]) < 2e-6


def test_latent_analysis_entry_points_vs_reference():
    """DiVAE.sweep / enumerate / exp_enumerate (sent_models.py:167-274): greedy ids and code tables exact."""
    G = load_golden("divae_gen_tiny")
    cfg = golden_config(G)
    P = golden_params(G)
    (seq, _, _), all_y = O.divae_sweep(P, cfg, G["in/outputs"], G["in/u"])
    assert np.array_equal(all_y.numpy(), G["sweep/all_y_ids"])
    assert np.array_equal(seq.numpy(), G["sweep/sequence"])
    seq, _, _ = O.divae_enumerate(P, cfg, repeat=1)
    assert np.array_equal(seq.numpy(), G["enumerate1/sequence"])
    seq, _, _ = O.divae_enumerate(P, cfg, repeat=2)
    assert np.array_equal(seq.numpy(), G["enumerate2/sequence"])
    (seq, _, _), ids = O.divae_exp_enumerate(P, cfg)
    assert np.array_equal(ids.numpy(), G["exp_enumerate/all_y_ids"])
    assert np.array_equal(seq.numpy(), G["exp_enumerate/sequence"])


def test_oracle_beam_search_degenerates_to_greedy_and_tracks_back():
    """oracle.decoder_beam (decoders.py:194-291 restated; PARITY UNPINNED, the reference's back-tracking does not run
    under Python 3): beam_size = 1 is greedy decoding; every returned sequence is a path through its own back pointers
    (its first symbol is one of the first step's top-k symbols of that sequence)."""
    G = load_golden("modules")
    sd = {k[len("dec/sd/"):]: torch.from_numpy(np.array(G[k])) for k in G.files if k.startswith("dec/sd/")}
    P = {"d." + k: v for k, v in sd.items()}
    h0 = torch.from_numpy(np.random.RandomState(1).randn(5, 9).astype(np.float32))
    s1, l1 = O.decoder_beam(P, "d.", P["d.embedding.weight"], h0, 6, 2, 3, 1)
    g, lg, _ = O.decoder_greedy(P, "d.", P["d.embedding.weight"], h0, 6, 2, 3)
    assert torch.equal(s1, g) and l1 == lg
    s3, l3 = O.decoder_beam(P, "d.", P["d.embedding.weight"], h0, 6, 2, 3, 3)
    assert s3.shape == (5, 6) and len(l3) == 15
    lp0, _ = O.decoder_teacher(P, "d.", P["d.embedding.weight"], torch.full((5, 1), 2, dtype=torch.int64), h0)
    top3 = lp0[:, 0].topk(3)[1]
    assert all(int(s3[b, 0]) in top3[b].tolist() for b in range(5))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_beam_search_loop_vs_reference_trajectory(tag):
    """oracle.decoder_beam against the trajectory of the REFERENCE's own beam-search loop (tests/golden/beam.npz,
    make_goldens_beam.py: the loop of decoders.py:258-268 runs under torch 2, only the back-tracking after it dies): the
    symbols fed back at every step exact, the per-step log-probs -- which carry the reference's quirk of NOT reordering
    hidden states by the back pointers -- to 1e-5."""
    from helpers import load_golden
    G = load_golden("beam")
    V, L, E, H, B, K = (int(x) for x in G[tag + "/dims"])
    assert float(G[tag + "/gap"].min()) > 1e-3 and "cannot be concat" in str(G[tag + "/died"])
    P = {"d." + k[len(tag) + 4:]: torch.from_numpy(np.array(G[k])) for k in G.files if k.startswith(tag + "/sd/")}
    h0 = torch.from_numpy(np.array(G[tag + "/h0"]))[0]
    with torch.no_grad():
        seq, lengths, syms, lps = O.decoder_beam(P, "d.", P["d.embedding.weight"], h0, L, 2, 3, K, trace=True)
    fed = G[tag + "/fed"]
    assert (fed[0] == 2).all()                                              # step 0 feeds <s> to every slot
    assert np.array_equal(syms.numpy().reshape(L, B * K)[:L - 1], fed[1:])
    assert np.max(np.abs(lps.numpy() - G[tag + "/logp"])) < 1e-5
    assert seq.shape == (B, L) and len(lengths) == B * K
