"""End-to-end oracle parity at the reference's OWN script shapes (BASELINE.json configs 1 and 2):

  * P   -- DI-VAE at ptb-utt.py's defaults (/root/reference/ptb-utt.py:40-51,84): V=10002, E=200, H_enc=H_dec=512, y=20, k=10,
           B=30, T<=40;
  * VST -- DI-VST at dailydialog-utt-skip.py's defaults (:40-51,83-84): V=10004, B=60, three id arrays, two decoders.

A full train step (zero_grad -> forward(TEACH_FORCE) -> backward -> Adam, laed/main.py:104-113 order) with injected
Gumbel noise and dropout keep masks and ragged PAD tails, against `oracle.divae_forward` / `oracle.divst_forward` +
autograd: nll / reg_kl / bpr at 1e-4 relative, EVERY parameter gradient at 2e-4 (max-norm), the parameters after one
Adam step at 1e-4 absolute.  The oracle step costs ~0.1 s at these shapes.

Launch paths these shapes reach that no smaller golden does:
  * the persistent recurrences (`gru_small_persist_{fwd,bwd}_kernel`, W_hh LDS-resident across all timesteps) WITH ROW
    GROUPS -- P: B=30 -> 2 groups of 16 rows x 128 (fwd) / 64 (bwd) workgroups on the full CU budget; VST: B=60 on the
    half budget of the two-stream region (128 CUs: one group of 64 rows forward = 128 workgroups, 2 groups x 64
    workgroups backward) and on the full budget for the encoder (4 groups of 16 forward = 512 > 256 -> 2 groups of 32);
    `laed_gru_persist_launches` asserts the persistent path was the one taken, `laed_gru_persist_timeouts` that none of
    its bounded waits gave up;
  * `proj_rows_reg_kernel<MODE, 4>` (softmax-CE rows at ldl = 10016 / 10032, register-resident);
  * the cost-model `launch_layout` choices for mid-size products at V ~ 10 k: 64x64 tiles and split-K for the logits
    (1170 x 10002 x 512), dh (1170 x 512 x 10002) and dW (10002 x 512 x 1170) products, and for dx = dgi W_ih;
  * the per-row (<= 4096 rows) embedding scatter-add.
"""
import os
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "neuraldialog-laed_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

E, H, Y, K, T = 200, 512, 20, 10, 40


def _config(B):
    return Namespace(embed_size=E, max_utt_len=T, num_layer=1, dropout=0.3, enc_cell_size=H, dec_cell_size=H,
                     rnn_cell='gru', max_dec_len=T, use_attn=False, beam_size=10, utt_type='rnn', bi_enc_cell=False,
                     attn_type='cat', fix_batch=False, y_size=Y, k=K, use_gpu=True, avg_type='word',
                     use_mutual=True, use_reg_kl=True, train_prior=False, init_lr=0.001, op='adam', batch_size=B)


def _corpus(V):
    class Corpus(object):
        vocab = ['<pad>', '<unk>', '<s>', '</s>'] + ['w%d' % i for i in range(4, V)]
        rev_vocab = {w: i for i, w in enumerate(vocab)}
    return Corpus()


def _batch(V, B, Tmax, seed):
    """[B, Tmax] ids: <s> ... </s> then PAD; lengths ~ clip(round(lognormal(3.0, 0.45)), 3, Tmax) (SURVEY.md 8d 'P-varlen'),
    the first row at full length so the batch maximum is Tmax (data_loaders.py:283-291 pads to the batch maximum)."""
    rs = np.random.RandomState(seed)
    lens = np.clip(np.round(rs.lognormal(3.0, 0.45, size=B)), 3, Tmax).astype(np.int64)
    lens[0] = Tmax
    out = np.zeros((B, Tmax), dtype=np.int32)
    for b, L in enumerate(lens):
        out[b, 0] = 2
        out[b, 1:L - 1] = rs.randint(4, V, size=L - 2)
        out[b, L - 1] = 3
    return out


def _check(got, grads, after, ref, P, names, cfg, loss_keys, what):
    from helpers import rel_err
    from oracle import laed_oracle as O
    for k in loss_keys:
        b = float(ref[k].detach())
        assert abs(got[k] - b) <= 1e-7 + 1e-4 * abs(b), (what, k, got[k], b)
    assert set(grads) == {k for k in names if P[k].grad is not None}
    worst = {}
    for n, g in grads.items():
        worst[n] = rel_err(g, P[n].grad)
        assert worst[n] < 2e-4, (what, n, worst[n])
    for n in names:
        if P[n].grad is None:
            assert torch.equal(after[n], P[n].detach()), n
            continue
        p = P[n].detach().clone()
        O.adam_step(p, P[n].grad, torch.zeros_like(p), torch.zeros_like(p), 1, lr=cfg.init_lr)
        d = float((after[n] - p).abs().max())
        assert d < 1e-4, (what, n, d)
    print(what, "worst rel err per gradient:", {n: "%.1e" % v for n, v in worst.items()})


def test_divae_train_step_at_ptb_script_shape_matches_oracle():
    """Config 1 (ptb-utt.py defaults), B = 30 ragged sequences: see the module docstring for the kernels this reaches."""
    from laed_amd import ops
    from laed_amd.enc2dec.decoders import TEACH_FORCE
    from laed_amd.models.sent_models import DiVAE
    from oracle import laed_oracle as O
    V, B = 10002, 30
    cfg = _config(B)
    torch.manual_seed(0)
    model = DiVAE(_corpus(V), cfg).cuda().train()
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    batch = _batch(V, B, T, seed=77)
    rs = np.random.RandomState(21)
    u = rs.uniform(size=(B * Y, K)).astype(np.float32)
    keep = (rs.uniform(size=(B, T - 1, E)) >= 0.3).astype(np.uint8)
    ud = torch.from_numpy(u).cuda()
    model.cat_connector.sample_uniform = lambda logits: ud
    model.decoder.injected_keep = torch.from_numpy(keep).cuda()
    opt = model.get_optimizer(cfg)
    f0, b0 = ops.gru_persist_launches()
    opt.zero_grad()
    loss = model({"outputs": batch, "output_lens": [T] * B}, mode=TEACH_FORCE)
    model.backward(0, loss)
    torch.cuda.synchronize()
    f1, b1 = ops.gru_persist_launches()
    assert (f1 - f0, b1 - b0) == (2, 2)                    # encoder + decoder, forward and backward: all persistent
    assert ops.gru_persist_timeouts() == 0
    grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    opt.step()
    torch.cuda.synchronize()
    after = {n: p.detach().cpu() for n, p in model.named_parameters()}
    got = {k: float(loss[k].detach()) for k in ("nll", "reg_kl", "bpr")}

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    P["decoder.embedding.weight"] = P["embedding.weight"]
    names = [k for k in P if not k.endswith("decoder.embedding.weight")]
    for k in names:
        P[k].requires_grad_(True)
    ref = O.divae_forward(P, cfg, batch, u, training=True, drop_mask=keep.astype(np.float32))
    ref["total"].backward()
    _check(got, grads, after, ref, P, names, cfg, ("nll", "reg_kl", "bpr"), "DI-VAE @ P")
    # per-token NLL of the same step (PAD targets exactly 0)
    tok = model.nll_loss_tok()
    labels = batch[:, 1:]
    m = labels != 0
    a, b = tok[m], ref["tok_nll"].detach().numpy()[m]
    assert float(np.abs(tok[~m]).max()) == 0.0 and np.max(np.abs(a - b) / np.abs(b)) <= 1e-4


@pytest.mark.parametrize("streams", ["two", "one"])
def test_divst_train_step_at_dailydialog_script_shape_matches_oracle(monkeypatch, streams):
    """Config 2 (dailydialog-utt-skip.py defaults), B = 60, outputs / prevs / nexts each padded to its own maximum
    (data_loaders.py:236-237,255-258).  'two': the decoders run on two HIP streams, each persistent recurrence on half
    the CUs (backward: 2 row groups x 64 workgroups = exactly its 128-CU share -- the case VERDICT r2 found untested);
    'one': LAED_VST_STREAMS=0, every recurrence on the whole device."""
    from laed_amd import ops
    from laed_amd.enc2dec.decoders import TEACH_FORCE
    from laed_amd.models.sent_models import DiVST
    from oracle import laed_oracle as O
    if streams == "one":
        monkeypatch.setenv("LAED_VST_STREAMS", "0")
    V, B = 10004, 60
    cfg = _config(B)
    torch.manual_seed(0)
    model = DiVST(_corpus(V), cfg).cuda().train()
    assert model.two_streams == (streams == "two")
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    outs, prevs, nexts = _batch(V, B, T, 78), _batch(V, B, 37, 79), _batch(V, B, 33, 80)
    rs = np.random.RandomState(22)
    u = rs.uniform(size=(B * Y, K)).astype(np.float32)
    keeps = [(rs.uniform(size=(B, a.shape[1] - 1, E)) >= 0.3).astype(np.uint8) for a in (prevs, nexts)]
    ud = torch.from_numpy(u).cuda()
    model.cat_connector.sample_uniform = lambda logits: ud
    model.prev_decoder.injected_keep = torch.from_numpy(keeps[0]).cuda()
    model.next_decoder.injected_keep = torch.from_numpy(keeps[1]).cuda()
    opt = model.get_optimizer(cfg)
    f0, b0 = ops.gru_persist_launches()
    opt.zero_grad()
    loss = model({"outputs": outs, "prevs": prevs, "nexts": nexts, "output_lens": [T] * B}, mode=TEACH_FORCE)
    model.backward(0, loss)
    torch.cuda.synchronize()
    f1, b1 = ops.gru_persist_launches()
    assert (f1 - f0, b1 - b0) == (3, 3)                    # encoder + two decoders: all persistent, both directions
    assert ops.gru_persist_timeouts() == 0
    grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    opt.step()
    torch.cuda.synchronize()
    after = {n: p.detach().cpu() for n, p in model.named_parameters()}
    got = {k: float(loss[k].detach()) for k in ("prev_nll", "next_nll", "reg_kl", "b_pr")}

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for d in ("prev_decoder", "next_decoder"):
        P[d + ".embedding.weight"] = P["embedding.weight"]
    names = [k for k in P if not k.endswith("decoder.embedding.weight")]
    for k in names:
        P[k].requires_grad_(True)
    ref = O.divst_forward(P, cfg, outs, prevs, nexts, u, training=True,
                          drop_masks=tuple(k.astype(np.float32) for k in keeps))
    ref["total"].backward()
    _check(got, grads, after, ref, P, names, cfg, ("prev_nll", "next_nll", "reg_kl", "b_pr"), "DI-VST @ VST (%s)" % streams)


def test_divae_eval_and_greedy_at_ptb_script_shape_matches_oracle():
    """Validation-style forward (eval: hard one-hot latent, no dropout, no_grad -> the forward-only projection with the
    online-softmax GEMM epilogue at V = 10002) and 40 greedy steps (arg-max epilogue) at the P shape: losses 1e-4, token
    ids exact wherever the oracle's own top-1/top-2 log-prob margin exceeds 1e-3 (random weights give near-ties; a
    sequence is compared up to its first near-tie)."""
    from laed_amd.enc2dec.decoders import GEN, TEACH_FORCE
    from laed_amd.models.sent_models import DiVAE
    from oracle import laed_oracle as O
    V, B = 10002, 30
    cfg = _config(B)
    torch.manual_seed(1)
    model = DiVAE(_corpus(V), cfg).cuda().eval()
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    P["decoder.embedding.weight"] = P["embedding.weight"]
    batch = _batch(V, B, T, seed=81)
    u = np.random.RandomState(23).uniform(size=(B * Y, K)).astype(np.float32)
    ud = torch.from_numpy(u).cuda()
    model.cat_connector.sample_uniform = lambda logits: ud
    feed = {"outputs": batch, "output_lens": [T] * B}
    with torch.no_grad():
        res = model(feed, mode=TEACH_FORCE, return_latent=True)
        dec_ctx, _ = model(feed, mode=GEN, gen_type='greedy')
    ids = torch.stack([s.view(-1) for s in dec_ctx['sequence']], dim=1).cpu().numpy()      # [B, 40]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref = O.divae_forward(P, cfg, batch, u, training=False)
        for k in ("nll", "reg_kl", "bpr"):
            b = float(ref[k])
            assert abs(float(res[k]) - b) <= 1e-7 + 1e-4 * abs(b), (k, float(res[k]), b)
        assert np.array_equal(res.y_ids.cpu().numpy(), ref["y_ids"].numpy())
        ref_ids, _, logp = O.divae_greedy(P, cfg, batch, u)                   # [B,40] ids, [B,40,V] log-probs
        top2 = logp.topk(2, dim=2)[0]
    ref_ids, margins = ref_ids.numpy(), (top2[..., 0] - top2[..., 1]).numpy()
    compared = 0
    for b in range(B):
        close = np.nonzero(margins[b] <= 1e-3)[0]
        upto = int(close[0]) if close.size else ids.shape[1]
        assert np.array_equal(ids[b, :upto], ref_ids[b, :upto]), (b, upto)
        compared += upto
    assert compared >= B * 20, compared                    # the comparison is not vacuous


# ----------------------------------------------------------------------------------------------------------------------
# configs 3 and 4: AE-ED / ST-ED at the Stanford scripts' defaults
# ----------------------------------------------------------------------------------------------------------------------
DIALOG_ALIASES = ("x_decoder.embedding.weight", "prev_decoder.embedding.weight", "next_decoder.embedding.weight",
                  "decoder.embedding.weight", "utt_encoder.embedding.weight")
NOISE_ONLY = ("utt_encoder.query.bias",)        # true gradient 0 (softmax shift invariance): round-off only on both sides


def _dialog_config(skip, B):
    """stanford-ae.py:40-90 / stanford-skip.py:40-92 defaults (y=3, k=5, utt cell 256, ctx / dec cell 512, B=30) with
    dropout 0 (the dialog oracle restates the dropout-free arithmetic; the keep-mask path is pinned at config 1 / 2)."""
    return Namespace(embed_size=E, max_utt_len=T, num_layer=1, dropout=0.0, dec_cell_size=512, ctx_cell_size=512,
                     utt_cell_size=256, bi_ctx_cell=False, rnn_cell='gru', max_dec_len=T, use_attn=False, beam_size=10,
                     utt_type='rnn' if skip else 'attn_rnn', attn_type='cat', fix_batch=False, y_size=3, k=5, use_gpu=True,
                     avg_type='word', use_reg_kl=True, use_attribute=True, greedy_q=True, init_lr=0.001, op='adam',
                     momentum=0.0, batch_size=B, freeze_step=5000 if skip else 4000, backward_size=10 if skip else 30)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind", ["aeed", "sted"])
def test_dialog_models_at_stanford_script_shape_match_oracle(kind):
    """AE-ED (config 3) / ST-ED (config 4) at B = 30 dialogs x 10 context turns x 40 tokens, V = 10004: the recognition
    phase (total = VAE / VST loss) and the freeze-flip step (total = encoder-decoder loss with the recognition side
    frozen, dialog_models.py:278-307 / :549-585) against `oracle.dialog_oracle` + autograd: every loss term 1e-4, every
    live gradient 3e-4 (max-norm).  Reaches what the tiny / mid goldens do not: the 300-sequence bidirectional utterance
    encoder on the mid-size row-group kernels (H = 256, 5 groups of 64 rows), attention pooling over 40 positions, the
    attribute head's `softmax(dec_outs) @ x_embedding` with V = 10004 in chunks, and the two-stream branches."""
    from laed_amd import ops
    from laed_amd.enc2dec.decoders import TEACH_FORCE
    from laed_amd.models.dialog_models import AeED, StED
    from helpers import rel_err
    from oracle import dialog_oracle as D
    V, B, C = 10004, 30, 10
    skip = kind == "sted"
    cfg = _dialog_config(skip, B)
    torch.manual_seed(3)
    model = (StED if skip else AeED)(_corpus(V), cfg).cuda().train()
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items() if k not in DIALOG_ALIASES}
    feed = {"outputs": _batch(V, B, T, 90), "output_lens": [T] * B,
            "contexts": np.stack([_batch(V, B, T, 100 + c) for c in range(C)], axis=1), "context_lens": [C] * B}
    if skip:
        feed["prevs"], feed["nexts"] = _batch(V, B, 38, 91), _batch(V, B, T, 92)
    u = np.random.RandomState(24).uniform(size=(B * cfg.y_size, cfg.k)).astype(np.float32)
    ud = torch.from_numpy(u).cuda()
    model.cat_connector.sample_uniform = lambda logits: ud
    keys = ["nll", "pi_kl", "pi_nll", "attribute_nll", "reg_kl", "pi_err", "adv_err"] + \
        (["vst_prev_nll", "vst_next_nll"] if skip else ["vae_nll"])
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    fwd = D.sted_forward if skip else D.aeed_forward

    def compare(batch_cnt, phase_b):
        model.zero_grad()
        loss = model(feed, mode=TEACH_FORCE)
        got = {k: float(loss[k].detach()) for k in keys}
        model.backward(batch_cnt, loss)
        torch.cuda.synchronize()
        live = [k for k in P if not (phase_b and k.startswith(D.FROZEN_PREFIXES[kind]))]
        for k in P:
            P[k].grad = None
            P[k].requires_grad_(k in live)
        ref = fwd(P, cfg, feed, u, kl_w=0.0, training=True)       # both steps run a phase-A forward (main.py:104-113)
        for k in keys:
            b = float(ref[k].detach())
            assert abs(got[k] - b) <= 2e-6 + 1e-4 * abs(b), (kind, phase_b, k, got[k], b)
        D.total_loss(kind, cfg, ref, phase_b).backward()
        named = dict(model.named_parameters())
        worst = {}
        for k in P:
            if k in NOISE_ONLY:
                continue
            if P[k].grad is None:
                assert named[k].grad is None or float(named[k].grad.abs().max()) == 0.0, (kind, phase_b, k)
                continue
            worst[k] = rel_err(named[k].grad.cpu(), P[k].grad)
            assert worst[k] < 3e-4, (kind, phase_b, k, worst[k])
        print(kind, "phase B (flip)" if phase_b else "phase A", "worst:", max(worst.items(), key=lambda kv: kv[1]))

    compare(0, False)
    compare(cfg.freeze_step + 1, True)
    assert model.flush_valid and model.kl_w == 1.0
    assert ops.gru_persist_timeouts() == 0
