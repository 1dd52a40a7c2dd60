"""Full-size checks at the headline shape (BASELINE.json config 5 / SURVEY.md 8d "S-fixed": V=50000, E=200, H=1024,
y=20, k=10, T=40, B=4096).  The oracle cannot run 4096 sequences in seconds, so parity at this size is taken through
what the domain offers:
  * sequences are independent through encoder, latent, decoder and per-token NLL (given the injected Gumbel noise),
    so a SUBSET of the full batch must match the oracle run on that subset alone: log q(y|x), latent ids, decoder
    initial state and every per-token NLL, at the north-star tolerance (1e-4 relative; ids exact);
  * the 'word' NLL is the non-PAD mean of the per-token NLLs of the whole batch, and BPR / KL / MI follow from the
    full log q(y|x) tensor by the oracle's own formula (cheap on the CPU: [B*y, k]);
  * a training step is deterministic: two runs from the same state give bitwise-equal losses and gradients (every
    reduction has a fixed order; the embedding scatter-add is sorted, not atomic); every row of the softmax-CE gradient sums to zero, so the projection bias gradient sums to ~0.
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

V, E, H, Y, K, T, B = 50000, 200, 1024, 20, 10, 40, 4096


def _config():
    return Namespace(embed_size=E, max_utt_len=T, num_layer=1, dropout=0.3, enc_cell_size=H, dec_cell_size=H,
                     rnn_cell='gru', max_dec_len=T, use_attn=False, beam_size=10, utt_type='rnn', bi_enc_cell=False,
                     attn_type='cat', fix_batch=False, y_size=Y, k=K, use_gpu=True, avg_type='word',
                     use_mutual=True, use_reg_kl=True, train_prior=False, init_lr=0.001, op='adam', batch_size=B)


class _Corpus(object):
    vocab = ['<pad>', '<unk>', '<s>', '</s>'] + ['w%d' % i for i in range(4, V)]
    rev_vocab = {w: i for i, w in enumerate(vocab)}


def _batch(seed=1234):
    """Zipf interior ids as bench.py's S-fixed batch, with ragged PAD tails on a quarter of the rows."""
    rs = np.random.RandomState(seed)
    p = 1.0 / np.arange(1, V - 4 + 1, dtype=np.float64)
    p /= p.sum()
    out = np.empty((B, T), dtype=np.int32)
    out[:, 0] = 2
    out[:, 1:T - 1] = 4 + rs.choice(V - 4, size=(B, T - 2), p=p)
    out[:, T - 1] = 3
    for b in range(0, B, 4):                      # shorter sequences: </s> then PAD (ids 0)
        L = int(rs.randint(3, T))
        out[b, L - 1] = 3
        out[b, L:] = 0
    return out


@pytest.fixture(scope="module")
def model():
    from laed_amd.models.sent_models import DiVAE
    torch.manual_seed(0)
    m = DiVAE(_Corpus(), _config()).cuda()
    return m


def test_subset_of_full_batch_matches_oracle(model):
    from laed_amd.enc2dec.decoders import TEACH_FORCE
    from oracle import laed_oracle as O
    batch = _batch()
    rs = np.random.RandomState(5)
    u = rs.uniform(size=(B * Y, K)).astype(np.float32)
    ud = torch.from_numpy(u).cuda()
    model.cat_connector.sample_uniform = lambda logits: ud
    model.eval()                                   # no dropout; hard one-hot latent from logits + injected noise
    with torch.no_grad():
        res = model({"outputs": batch, "output_lens": [T] * B}, mode=TEACH_FORCE, return_latent=True)
    tok = model.nll_loss_tok()                     # [B, T-1]
    log_qy = res.log_qy.detach().cpu().view(B, Y, K)
    # (1) whole-batch reductions from the per-row outputs
    labels = batch[:, 1:]
    cnt = int((labels != 0).sum())
    assert abs(float(res.nll) - float(tok.sum(dtype=np.float64)) / cnt) <= 1e-5 * abs(float(res.nll))
    assert float(np.abs(tok[labels == 0]).max()) == 0.0
    reg_kl, bpr, mi = O.latent_losses(log_qy.view(-1, K), Y, K, True)
    assert abs(float(res.bpr) - float(bpr)) <= 1e-7 + 1e-4 * abs(float(bpr))
    assert abs(float(res.reg_kl) - float(reg_kl)) <= 1e-7 + 1e-4 * abs(float(reg_kl))
    assert abs(float(res.mi) - float(mi)) <= 1e-6 + 5e-3 * abs(float(mi))
    # (2) a subset of the sequences against the oracle run on that subset alone
    sel = np.concatenate([np.arange(0, 8), np.array([1023, 1024, 2048, 2051, 4088, 4095])])      # incl. ragged rows
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    P["decoder.embedding.weight"] = P["embedding.weight"]
    usel = u.reshape(B, Y, K)[sel].reshape(-1, K)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = O.divae_forward(P, _config(), batch[sel], usel, training=False)
    assert np.array_equal(res.y_ids.detach().cpu().numpy()[sel], ref["y_ids"].numpy())
    a, b = log_qy[sel].reshape(-1, K).numpy(), ref["log_qy"].detach().numpy()
    assert np.max(np.abs(a - b)) <= 1e-4 * np.max(np.abs(b))
    a, b = res.dec_init_state.detach().cpu().numpy()[0][sel], ref["dec_init_state"].detach().numpy()[0]
    assert np.max(np.abs(a - b)) <= 1e-4 * np.max(np.abs(b))
    mask = labels[sel] != 0
    a, b = tok[sel][mask], ref["tok_nll"].detach().numpy()[mask]
    assert np.max(np.abs(a - b) / np.abs(b)) <= 1e-4, float(np.max(np.abs(a - b) / np.abs(b)))


def test_training_step_is_deterministic_and_ce_gradient_rows_sum_to_zero(model):
    from laed_amd.enc2dec.decoders import TEACH_FORCE
    batch = _batch(seed=99)
    rs = np.random.RandomState(6)
    ud = torch.from_numpy(rs.uniform(size=(B * Y, K)).astype(np.float32)).cuda()
    model.cat_connector.sample_uniform = lambda logits: ud
    model.train()
    keep = torch.from_numpy((rs.uniform(size=(B, T - 1, E)) >= 0.3).astype(np.uint8)).cuda()
    runs = []
    for _ in range(2):
        model.decoder.injected_keep = keep
        for p in model.parameters():
            p.grad = None
        loss = model({"outputs": batch, "output_lens": [T] * B}, mode=TEACH_FORCE)
        model.backward(0, loss)
        torch.cuda.synchronize()
        runs.append((float(loss.nll.detach()), float(loss.reg_kl.detach()),
                     {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1]
    for n in runs[0][2]:
        assert torch.equal(runs[0][2][n], runs[1][2][n]), n     # the embedding scatter-add is sorted, not atomic
    g = runs[0][2]
    db = g["decoder.project.bias"].double()
    # each row of (softmax - onehot) * scale sums to 0, so sum_v db[v] = 0 up to fp32 rounding of 50000 column sums
    assert abs(float(db.sum())) <= 1e-4 * float(db.abs().sum())
    assert float(g["embedding.weight"][0].abs().max()) == 0.0          # PAD row gets no gradient
    for n, t in g.items():
        assert bool(torch.isfinite(t).all()), n


@pytest.mark.timeout(900)
def test_backward_at_headline_shape_matches_oracle_gradients(monkeypatch):
    """Train-mode step at the headline shape's V / E / H / y / k / T with B = 512 sequences (= the per-GPU share of the
    8-GPU point, 19 968 target rows) against `oracle.divae_forward` + autograd (sent_models.py:94-165 +
    model_bases.py:62-64): nll / reg_kl / bpr at 1e-4, EVERY parameter gradient at 2e-4 (max-norm), and the parameters
    after one Adam step (1e-4 absolute).  Ragged PAD tails, injected Gumbel noise and dropout keep mask.

    Launch paths this covers (B > 64, so none of the small-batch kernels run; V = 50000, H = 1024):
      * `proj_rows_reg_kernel<MODE_GRAD, 13>` (register-resident softmax-CE rows at ldl = 50016),
      * the projection's dW product `sgemm_kernel<128,128,TN>` with the tail split (391 x 8 = 3128 tiles > 512) and
        accumulated over 3 row chunks (LAED_PROJ_CHUNK_ROWS = 6656), dh = dl W split over K (`DH_SPLITS`),
      * `gru_step_fwd_kernel<2,2,...>` (64-row tiles: cdiv(512,64) * 32 unit tiles = 256 blocks),
      * `gru_bwd_gates_kernel` + the per-step split-K NN product dh_{t-1} += dgh_t W_hh (32 output tiles, partials folded
        into the next step's gate kernel), and the post-loop dW_hh / dW_ih / dx products,
      * the sorted (deterministic) embedding scatter-add over 512 x 79 rows, fused Adam over 69 M parameters.
    """
    from laed_amd.enc2dec.decoders import TEACH_FORCE
    from laed_amd.models.sent_models import DiVAE
    from oracle import laed_oracle as O
    Bs = 512
    monkeypatch.setenv("LAED_PROJ_CHUNK_ROWS", "6656")            # 19968 rows -> 3 chunks accumulate into dW / db
    cfg = _config()
    cfg.batch_size = Bs
    torch.manual_seed(0)
    model = DiVAE(_Corpus(), cfg).cuda().train()
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    batch = _batch(seed=4321)[:Bs]
    rs = np.random.RandomState(11)
    u = rs.uniform(size=(Bs * Y, K)).astype(np.float32)
    keep = (rs.uniform(size=(Bs, T - 1, E)) >= 0.3).astype(np.uint8)
    ud = torch.from_numpy(u).cuda()
    model.cat_connector.sample_uniform = lambda logits: ud
    model.decoder.injected_keep = torch.from_numpy(keep).cuda()
    opt = model.get_optimizer(cfg)
    opt.zero_grad()
    loss = model({"outputs": batch, "output_lens": [T] * Bs}, mode=TEACH_FORCE)
    model.backward(0, loss)
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    opt.step()
    torch.cuda.synchronize()
    after = {n: p.detach().cpu() for n, p in model.named_parameters()}
    got = {k: float(loss[k].detach()) for k in ("nll", "reg_kl", "bpr")}
    del model, opt, loss
    torch.cuda.empty_cache()

    # oracle: same state, same batch, same noise (CPU fp32; ~7 TFLOP, seconds on the box's 16 host threads)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    P["decoder.embedding.weight"] = P["embedding.weight"]
    names = [k for k in P if not k.endswith("decoder.embedding.weight")]
    for k in names:
        P[k].requires_grad_(True)
    ref = O.divae_forward(P, cfg, batch, u, training=True, drop_mask=keep.astype(np.float32))
    ref["total"].backward()
    for k in ("nll", "reg_kl", "bpr"):
        b = float(ref[k].detach())
        assert abs(got[k] - b) <= 1e-7 + 1e-4 * abs(b), (k, got[k], b)
    assert set(grads) == {k for k in names if P[k].grad is not None}
    from helpers import rel_err
    worst = {}
    for n, g in grads.items():
        worst[n] = rel_err(g, P[n].grad)
        assert worst[n] < 2e-4, (n, worst[n])
    # one Adam step (model_bases.py:122-125 defaults) on the oracle's own gradients
    for n in names:
        if P[n].grad is None:
            assert torch.equal(after[n], P[n].detach()), n                   # log_py: no gradient, untouched
            continue
        p = P[n].detach().clone()
        O.adam_step(p, P[n].grad, torch.zeros_like(p), torch.zeros_like(p), 1, lr=cfg.init_lr)
        d = float((after[n] - p).abs().max())
        assert d < 1e-4, (n, d)
    print("headline-shape gradient parity, worst rel err per tensor:", {n: "%.1e" % v for n, v in worst.items()})


@pytest.mark.timeout(900)
def test_divst_backward_at_headline_width_matches_oracle_gradients(monkeypatch):
    """DI-VST (config 2's model) at the headline V / E / H / y / k / T with B = 128 sequences: one encoder, two decoders
    that share the latent and the embedding and run -- forward and, replayed by autograd, backward -- on two HIP streams
    (sent_models.py:363-442).  Against `oracle.divst_forward` + autograd: prev_nll / next_nll / reg_kl / b_pr at 1e-4,
    every parameter gradient at 2e-4 (max-norm), parameters after one Adam step at 1e-4 absolute.

    What this adds to the DI-VAE case above: the two-stream region at B > 64 (tiled GRU step kernels and 128x128 GEMMs
    of two decoders in flight at once, each with its own projection workspace), the shared embedding's gradient summed
    from three scatter-adds, the connector bias, and dh0 of two decoders meeting in the latent's backward."""
    from laed_amd.enc2dec.decoders import TEACH_FORCE
    from laed_amd.models.sent_models import DiVST
    from oracle import laed_oracle as O
    Bs = 128
    monkeypatch.setenv("LAED_PROJ_CHUNK_ROWS", "2560")            # 4992 rows per decoder -> 2 chunks each
    cfg = _config()
    cfg.batch_size = Bs
    torch.manual_seed(0)
    model = DiVST(_Corpus(), cfg).cuda().train()
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    outs, prevs, nexts = (_batch(seed=s)[:Bs] for s in (4321, 4322, 4323))
    rs = np.random.RandomState(12)
    u = rs.uniform(size=(Bs * Y, K)).astype(np.float32)
    keeps = [(rs.uniform(size=(Bs, T - 1, E)) >= 0.3).astype(np.uint8) for _ in range(2)]
    ud = torch.from_numpy(u).cuda()
    model.cat_connector.sample_uniform = lambda logits: ud
    model.prev_decoder.injected_keep = torch.from_numpy(keeps[0]).cuda()
    model.next_decoder.injected_keep = torch.from_numpy(keeps[1]).cuda()
    opt = model.get_optimizer(cfg)
    opt.zero_grad()
    loss = model({"outputs": outs, "prevs": prevs, "nexts": nexts, "output_lens": [T] * Bs}, mode=TEACH_FORCE)
    model.backward(0, loss)
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    opt.step()
    torch.cuda.synchronize()
    after = {n: p.detach().cpu() for n, p in model.named_parameters()}
    got = {k: float(loss[k].detach()) for k in ("prev_nll", "next_nll", "reg_kl", "b_pr")}
    del model, opt, loss
    torch.cuda.empty_cache()

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for d in ("prev_decoder", "next_decoder"):
        P[d + ".embedding.weight"] = P["embedding.weight"]
    names = [k for k in P if not k.endswith("decoder.embedding.weight")]
    for k in names:
        P[k].requires_grad_(True)
    ref = O.divst_forward(P, cfg, outs, prevs, nexts, u, training=True,
                          drop_masks=tuple(k.astype(np.float32) for k in keeps))
    ref["total"].backward()
    for k in got:
        b = float(ref[k].detach())
        assert abs(got[k] - b) <= 1e-7 + 1e-4 * abs(b), (k, got[k], b)
    assert set(grads) == {k for k in names if P[k].grad is not None}
    from helpers import rel_err
    worst = {}
    for n, g in grads.items():
        worst[n] = rel_err(g, P[n].grad)
        assert worst[n] < 2e-4, (n, worst[n])
    for n in names:
        if P[n].grad is None:
            assert torch.equal(after[n], P[n].detach()), n
            continue
        p = P[n].detach().clone()
        O.adam_step(p, P[n].grad, torch.zeros_like(p), torch.zeros_like(p), 1, lr=cfg.init_lr)
        d = float((after[n] - p).abs().max())
        assert d < 1e-4, (n, d)
    print("DI-VST headline-width gradient parity, worst rel err per tensor:", {n: "%.1e" % v for n, v in worst.items()})
