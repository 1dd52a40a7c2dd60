"""Edge shapes of the DI-VAE / DI-VST step against the oracle (oracle/laed_oracle.py, golden-pinned to the reference):
the smallest batch and sequence, ragged rows down to a bare `<s> </s>`, a row that is all PAD, sizes that are no
multiple of any tile (odd V / E / k, H that leaves the packed small-batch kernels, B either side of their limit),
a vocabulary longer than the register-resident softmax-CE row (V > 65536), empty row sets at the C-ABI, and the
degenerate all-PAD batch (0/0: both sides NaN).  Same bar as the golden tests: losses 1e-4, latent ids exact.
"""
import os
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "neuraldialog-laed_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


class Corpus(object):
    def __init__(self, V):
        self.vocab = ['<pad>', '<unk>', '<s>', '</s>'] + ['w%d' % i for i in range(4, V)]
        self.rev_vocab = {w: i for i, w in enumerate(self.vocab)}


def _config(E, H, y, k, T, B, **kw):
    c = Namespace(embed_size=E, max_utt_len=T, num_layer=1, dropout=0.0, enc_cell_size=H, dec_cell_size=H,
                  rnn_cell='gru', max_dec_len=T, use_attn=False, beam_size=10, utt_type='rnn', bi_enc_cell=False,
                  attn_type='cat', fix_batch=False, y_size=y, k=k, use_gpu=True, avg_type='word',
                  use_mutual=True, use_reg_kl=True, train_prior=False, init_lr=0.001, op='adam', batch_size=B)
    for a, b in kw.items():
        setattr(c, a, b)
    return c


def _ragged(rs, B, T, V, lens=None):
    """[<s>, ids, </s>, PAD...]; lens[b] = number of non-PAD positions (0 = a row that is all PAD)."""
    out = np.zeros((B, T), dtype=np.int32)
    for b in range(B):
        L = int(rs.randint(2, T + 1)) if lens is None else int(lens[b])
        if L == 0:
            continue
        out[b, 0] = 2
        if L > 2:
            out[b, 1:L - 1] = rs.randint(4, V, size=L - 2)
        out[b, L - 1] = 3
    return out


def _run_case(kind, V, E, H, y, k, T, B, lens=None, seed=0, avg_type='word', use_mutual=True, grad_tol=3e-4):
    from laed_amd.enc2dec.decoders import TEACH_FORCE
    from laed_amd.models.sent_models import DiVAE, DiVST
    from oracle import laed_oracle as O
    rs = np.random.RandomState(seed)
    cfg = _config(E, H, y, k, T, B, avg_type=avg_type, use_mutual=use_mutual)
    torch.manual_seed(seed)
    model = (DiVAE if kind == "divae" else DiVST)(Corpus(V), cfg).cuda()
    model.train()
    u = rs.uniform(size=(B * y, k)).astype(np.float32)
    ud = torch.from_numpy(u).cuda()
    model.cat_connector.sample_uniform = lambda logits: ud
    feed = {"outputs": _ragged(rs, B, T, V, lens), "output_lens": [T] * B}
    if kind == "divst":
        feed["prevs"], feed["nexts"] = _ragged(rs, B, T, V), _ragged(rs, B, T, V)
    loss = model(feed, mode=TEACH_FORCE, return_latent=True)
    model.zero_grad()
    model.backward(0, loss)
    # oracle on the same weights, inputs and noise
    P = {n: t.detach().cpu().clone() for n, t in model.state_dict().items()}
    for n in list(P):
        if n.endswith("decoder.embedding.weight"):
            P[n] = P["embedding.weight"]
    for n, t in P.items():
        if not n.endswith("decoder.embedding.weight") and t.is_floating_point():
            t.requires_grad_(True)
    if kind == "divae":
        ref = O.divae_forward(P, cfg, feed["outputs"], u, training=True)
        terms = (("nll", "nll"), ("reg_kl", "reg_kl"), ("bpr", "bpr"))
    else:
        ref = O.divst_forward(P, cfg, feed["outputs"], feed["prevs"], feed["nexts"], u, training=True)
        terms = (("prev_nll", "prev_nll"), ("next_nll", "next_nll"), ("reg_kl", "reg_kl"), ("b_pr", "b_pr"))
    ref["total"].backward()
    for a, b in terms:
        x, r = float(loss[a].detach()), float(ref[b].detach())
        assert abs(x - r) <= 2e-6 + 1e-4 * abs(r), (a, x, r)
    assert np.array_equal(loss["y_ids"].cpu().numpy(), ref["y_ids"].numpy())
    assert rel_err(loss["log_qy"].detach().cpu(), ref["log_qy"].detach()) < 1e-5
    for n, p in model.named_parameters():
        g = P[n].grad
        if g is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        if float(g.abs().max()) < 1e-12:
            assert float(p.grad.abs().max()) < 1e-9, n
        else:
            assert rel_err(p.grad.cpu(), g) < grad_tol, (n, rel_err(p.grad.cpu(), g))
    return model, loss


@pytest.mark.parametrize("V,E,H,y,k,T,B", [
    (9, 3, 16, 1, 2, 2, 1),          # one sequence, one decoder position, one binary latent
    (37, 5, 24, 3, 7, 6, 5),         # H % 16 != 0: the tiled GRU kernels at a tiny batch; odd V, E, k
    (53, 16, 48, 2, 5, 9, 64),       # B = 64: last batch of the packed small-batch kernels
    (53, 16, 48, 2, 5, 9, 65),       # B = 65: first batch of the tiled kernels
    (131, 10, 32, 4, 10, 40, 7),     # full-length sequences (T = max_utt_len = 40)
])
def test_divae_edge_shapes(V, E, H, y, k, T, B):
    _run_case("divae", V, E, H, y, k, T, B, seed=B + T)


def test_divae_rows_down_to_bos_eos_and_one_all_pad_row():
    # lengths 2 (bare <s> </s>: one target), 3, full, and 0 (a row that is all PAD: contributes nothing to the NLL,
    # still goes through the encoder and the latent like any other row, as in the reference)
    _run_case("divae", 41, 8, 32, 2, 4, 7, 6, lens=[2, 3, 7, 0, 2, 5], seed=3)


def test_divst_edge_shapes():
    _run_case("divst", 29, 6, 16, 2, 3, 5, 3, seed=11)
    _run_case("divst", 61, 12, 40, 3, 4, 8, 70, seed=12)       # tiled kernels (H % 16 != 0), B > 64, two streams


@pytest.mark.parametrize("avg_type", ["seq", "real_word", None])
def test_other_nll_averaging_modes(avg_type):
    _run_case("divae", 33, 8, 32, 2, 4, 6, 4, lens=[6, 2, 4, 3], seed=21, avg_type=avg_type)


def test_non_mutual_kl():
    _run_case("divae", 33, 8, 32, 2, 4, 6, 4, seed=22, use_mutual=False)


def test_all_pad_batch_is_nan_like_the_reference():
    """'word' averaging divides by the non-PAD count (criterions.py:65-68): with no target at all the reference's
    nll_loss returns NaN (0/0), and so does the fused path."""
    from laed_amd import ops
    N, H, V = 6, 16, 21
    h = torch.randn(N, H, device="cuda")
    W, b = torch.randn(V, H, device="cuda"), torch.randn(V, device="cuda")
    labels = torch.zeros(N, dtype=torch.int64, device="cuda")
    scale = (1.0 / (labels != 0).sum().float()).expand(N).contiguous()
    loss, tok = ops.ProjNLLFunction.apply(h, W, b, labels, scale, 0)
    ref = torch.nn.functional.nll_loss(torch.log_softmax(h @ W.t() + b, dim=1), labels, ignore_index=0)
    assert bool(torch.isnan(ref)) and bool(torch.isnan(loss))


def test_vocabulary_longer_than_a_register_row():
    """V > 65536 takes the three-pass softmax-CE row kernel instead of the register-resident one."""
    from laed_amd import ops
    N, H, V = 40, 32, 70003
    g = torch.Generator().manual_seed(5)
    h, W, b = torch.randn(N, H, generator=g), torch.randn(V, H, generator=g) * 0.3, torch.randn(V, generator=g) * 0.1
    labels = torch.randint(1, V, (N,), generator=g)
    labels[::5] = 0
    labels[1] = V - 1
    cnt = float((labels != 0).sum())
    hc, Wc, bc = (t.clone().requires_grad_(True) for t in (h, W, b))
    logp = torch.log_softmax(hc @ Wc.t() + bc, dim=1)
    ref = torch.nn.functional.nll_loss(logp, labels, ignore_index=0)
    ref.backward()
    hg, Wg, bg = (t.cuda().requires_grad_(True) for t in (h, W, b))
    loss, tok = ops.ProjNLLFunction.apply(hg, Wg, bg, labels.cuda(), torch.full((N,), 1.0 / cnt, device="cuda"), 0)
    loss.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    assert rel_err(hg.grad.cpu(), hc.grad) < 1e-5 and rel_err(Wg.grad.cpu(), Wc.grad) < 1e-5
    assert rel_err(bg.grad.cpu(), bc.grad) < 1e-5


def test_empty_row_sets_at_the_c_abi():
    """Zero rows are a valid call (an empty shard under data parallelism): nothing is launched, outputs keep their
    shapes, gradients of the weights are zero."""
    from laed_amd import ops
    H, V, E = 16, 21, 8
    W, b = torch.randn(V, H, device="cuda"), torch.randn(V, device="cuda")
    h = torch.empty(0, H, device="cuda")
    labels = torch.empty(0, dtype=torch.int64, device="cuda")
    tok, lse, dh, dW, db = ops.proj_nll_raw(h, W, b, labels, torch.empty(0, device="cuda"), 0, True)
    torch.cuda.synchronize()
    assert tok.shape == (0,) and dh.shape == (0, H)
    assert float(dW.abs().max()) == 0.0 and float(db.abs().max()) == 0.0
    table = torch.randn(V, E, device="cuda")
    out = ops.EmbedFunction.apply(table, torch.empty(0, dtype=torch.int64, device="cuda"), None, 0.0, 0)
    assert out.shape == (0, E)
    c = ops.sgemm(torch.empty(0, H, device="cuda"), W, transB=True)
    assert c.shape == (0, V)


def test_divst_two_streams_wide_decoder_does_not_wait_for_cus():
    """DI-VST with dec_cell_size = 1024 at the script batch (60): each decoder's persistent recurrence would need
    1024/4 = 256 workgroups resident at once, one per CU -- two of them side by side on the model's two streams can never
    both fit the chip's 256 CUs.  Inside `ops.concurrent_streams(2)` each recurrence gets half the CUs as its budget and
    runs as per-step launches instead; nothing waits, nothing is poisoned with NaN, and losses and gradients equal the
    oracle's (sent_models.py:363-442)."""
    import numpy as np
    import torch
    from argparse import Namespace
    from laed_amd import ops
    from laed_amd.enc2dec.decoders import TEACH_FORCE
    from laed_amd.models.sent_models import DiVST
    from oracle import laed_oracle as O
    from helpers import rel_err
    V, E, He, Hd, Y, K, T, B = 300, 32, 64, 1024, 4, 5, 9, 60
    cfg = Namespace(embed_size=E, max_utt_len=T, num_layer=1, dropout=0.0, enc_cell_size=He, dec_cell_size=Hd,
                    rnn_cell='gru', max_dec_len=T, use_attn=False, beam_size=4, utt_type='rnn', bi_enc_cell=False,
                    attn_type='cat', fix_batch=False, y_size=Y, k=K, use_gpu=True, avg_type='word', use_mutual=True,
                    use_reg_kl=True, train_prior=False, init_lr=0.001, op='adam', batch_size=B)

    class Corpus(object):
        vocab = ['<pad>', '<unk>', '<s>', '</s>'] + ['w%d' % i for i in range(4, V)]
        rev_vocab = {w: i for i, w in enumerate(vocab)}

    torch.manual_seed(3)
    model = DiVST(Corpus(), cfg).cuda().train()
    assert model.two_streams
    rs = np.random.RandomState(8)

    def utts():
        a = rs.randint(4, V, size=(B, T)).astype(np.int32)
        a[:, 0], a[:, -1] = 2, 3
        a[::5, T - 3:] = 0
        a[::5, T - 4] = 3
        return a
    outs, prevs, nexts = utts(), utts(), utts()
    u = rs.uniform(size=(B * Y, K)).astype(np.float32)
    ud = torch.from_numpy(u).cuda()
    model.cat_connector.sample_uniform = lambda logits: ud
    before = ops.gru_persist_timeouts()
    loss = model({"outputs": outs, "output_lens": [T] * B, "prevs": prevs, "nexts": nexts}, mode=TEACH_FORCE)
    model.backward(0, loss)
    torch.cuda.synchronize()
    assert ops.gru_persist_timeouts() == before
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for k in list(P):
        if k.endswith("decoder.embedding.weight"):
            P[k] = P["embedding.weight"]
        else:
            P[k].requires_grad_(True)
    ref = O.divst_forward(P, cfg, outs, prevs, nexts, u, training=True)
    ref["total"].backward()
    for k in ("prev_nll", "next_nll", "reg_kl"):
        a, b = float(loss[k].detach()), float(ref[k].detach())
        assert np.isfinite(a) and abs(a - b) <= 1e-7 + 1e-4 * abs(b), (k, a, b)
    for n, p in model.named_parameters():
        if P[n].grad is not None:
            assert rel_err(p.grad.cpu(), P[n].grad) < 2e-4, (n, rel_err(p.grad.cpu(), P[n].grad))
