"""CPU oracle for the DI-VAE / DI-VST hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
file, and only as the checker / the timed CPU baseline.  The product package
(neuraldialog-laed_amd/laed_amd) never imports it and has no CPU fallback.

What it is: a fresh fp32 restatement, in plain torch-CPU tensor ops, of the arithmetic the
reference (snakeztc/NeuralDialog-LAED) performs on this path.  The reference delegates every
floating-point op to PyTorch (no pinned version beyond `pytorch >= 0.3.0.post4`,
/root/reference/README.md:19-22), so the oracle restates torch's published GRU / Linear /
log_softmax / nll_loss / Adam formulas explicitly and is PINNED by tests/golden/*.npz, which
were produced by importing and running the reference itself in the build container
(tests/golden/make_goldens.py, torch 2.10.0 CPU fp32; version recorded inside each file).
tests/test_oracle_vs_golden.py checks every function here against those files.

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

PAD_ID = 0          # laed/dataset/corpora.py:14  vocab = [PAD, UNK, ...] (corpora.py:178)
TEACH_FORCE = 'teacher_forcing'   # laed/enc2dec/decoders.py:12-14
TEACH_GEN = 'teacher_gen'
GEN = 'gen'


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def gru_seq(x, h0, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """1-layer GRU over x[B,T,E] from h0[B,H]; gate row blocks ordered (r,z,n).

    nn.GRU(batch_first=True) as built at laed/enc2dec/base_modules.py:43-45 and invoked at
    encoders.py:62-64 / decoders.py:158:
        r = sig(W_ir x + b_ir + W_hr h + b_hr);  z = sig(W_iz x + b_iz + W_hz h + b_hz)
        n = tanh(W_in x + b_in + r * (W_hn h + b_hn));  h' = (1-z)*n + z*h
    Returns (out[B,T,H], hT[B,H]).
    """
    B, T, _ = x.shape
    H = h0.shape[1]
    gi_all = F.linear(x, w_ih, b_ih)                       # hoisted input projection
    h = h0
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gi = gi_all[:, t]
        gh = F.linear(h, w_hh, b_hh)
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1.0 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, dim=1), h


def gumbel_noise(u, eps=1e-20):
    """laed/nn_lib.py:232-236  g = -log(-log(u + eps) + eps)."""
    return -torch.log(-torch.log(u + eps) + eps)


def gumbel_softmax(logits, u, temperature, hard):
    """laed/nn_lib.py:238-261.  logits,u [R,k] -> (y [R,k], ids [R,1] int64).

    y = softmax((logits + g)/tau); ids = argmax_k y (first max); hard -> one-hot(ids)
    (no straight-through: the hard sample carries no gradient, nn_lib.py:254-257).
    """
    y = F.softmax((logits + gumbel_noise(u)) / temperature, dim=1)
    ids = torch.max(y, dim=1, keepdim=True)[1]
    if hard:
        y = torch.zeros_like(y).scatter_(1, ids, 1.0)
    return y, ids


def nll_entropy(logp, labels, avg_type='word', pad_id=PAD_ID):
    """laed/criterions.py:43-72 (weight=None).  logp [B,T,V], labels [B,T] int64."""
    B, T, V = logp.shape
    tok = -logp.reshape(-1, V).gather(1, labels.reshape(-1, 1)).squeeze(1)
    mask = (labels.reshape(-1) != pad_id).to(logp.dtype)
    tok = tok * mask
    if avg_type is None:
        return tok.sum()
    if avg_type == 'seq':
        return tok.sum() / B
    if avg_type == 'real_word':
        word_cnt = torch.sign(labels).sum(dim=1).to(logp.dtype)      # criterions.py:62
        return (tok.view(B, T).sum(dim=1) / word_cnt).mean()
    if avg_type == 'word':
        return tok.sum() / mask.sum()
    raise ValueError("Unknown avg type")


def cat_kl(log_qy, log_py, batch_size=None, unit_average=False):
    """laed/criterions.py:97-109."""
    qy = torch.exp(log_qy)
    y_kl = torch.sum(qy * (log_qy - log_py), dim=1)
    return y_kl.mean() if unit_average else y_kl.sum() / batch_size


def entropy(log_qy, batch_size=None, unit_average=False):
    """laed/criterions.py:136-147."""
    h_q = torch.sum(-1.0 * log_qy * torch.exp(log_qy), dim=1)
    return h_q.mean() if unit_average else h_q.sum() / batch_size


def latent_losses(log_qy, y_size, k, use_mutual=True, log_py=None):
    """BPR / reg_kl / MI block, laed/models/sent_models.py:138-156.

    log_qy [B*y,k].  log_py [y,k] (the model's `log_py` parameter) selects the train_prior branch (:144-146):
    reg_kl = KL(avg q || softmax(log_py)).  Returns (reg_kl, bpr, mi).
    """
    log_uniform = math.log(1.0 / k)                                     # sent_models.py:81
    avg = torch.exp(log_qy.view(-1, y_size, k)).mean(dim=0)
    avg_log_qy = torch.log(avg + 1e-15)
    bpr = cat_kl(avg_log_qy, log_uniform, unit_average=True)
    if use_mutual:
        if log_py is not None:
            reg_kl = cat_kl(avg_log_qy, F.log_softmax(log_py, dim=log_py.dim() - 1), unit_average=True)
        else:
            reg_kl = bpr
    else:
        reg_kl = cat_kl(log_qy, log_uniform, unit_average=True)
    mi = entropy(avg_log_qy, unit_average=True) - entropy(log_qy, unit_average=True)
    return reg_kl, bpr, mi


def decoder_teacher(P, prefix, emb_w, dec_inputs, h0, drop_mask=None, drop_p=0.0):
    """DecoderRNN TEACH_FORCE branch, laed/enc2dec/decoders.py:152-167,251-256.

    dec_inputs [B,T] int64, h0 [B,H] -> (logp [B,T,V], hT [B,H]).
    drop_mask: keep mask [B,T,E] (1 keep / 0 drop) or None.
    """
    x = F.embedding(dec_inputs, emb_w, padding_idx=PAD_ID)   # padding_idx: PAD row gets no grad (sent_models.py:44-45)
    if drop_mask is not None:
        x = x * (drop_mask / (1.0 - drop_p))
    out, hT = gru_seq(x, h0, P[prefix + 'rnn.weight_ih_l0'], P[prefix + 'rnn.weight_hh_l0'],
                      P[prefix + 'rnn.bias_ih_l0'], P[prefix + 'rnn.bias_hh_l0'])
    logits = F.linear(out, P[prefix + 'project.weight'], P[prefix + 'project.bias'])
    return F.log_softmax(logits, dim=2), hT


def decoder_greedy(P, prefix, emb_w, h0, max_len, go_id, eos_id):
    """DecoderRNN GEN/greedy, laed/enc2dec/decoders.py:180-192,211-246,258-268.

    Exactly max_len steps, no early exit; EOS only updates `lengths`.
    Returns (sequence [B,max_len] int64, lengths list, logp [B,max_len,V]).
    """
    B = h0.shape[0]
    ids = torch.full((B, 1), go_id, dtype=torch.int64)
    h = h0
    seq, lps = [], []
    lengths = np.array([max_len] * B)
    for di in range(max_len):
        lp, h = decoder_teacher(P, prefix, emb_w, ids, h)
        ids = lp[:, 0].topk(1)[1]
        seq.append(ids)
        lps.append(lp)
        eos = (ids.view(-1) == eos_id).numpy()
        upd = (lengths > di) & eos
        lengths[upd] = len(seq)
    return torch.cat(seq, dim=1), lengths.tolist(), torch.cat(lps, dim=1)


def decoder_sample(P, prefix, emb_w, h0, max_len, go_id, eos_id, u):
    """DecoderRNN GEN with gen_type='sample': decoders.py:220-221 -> gumbel_max, base_modules.py:54-67
    (symbols = argmax(log_probs - log(-log(u)))), u [max_len,B,V] replayed uniforms.  Same bookkeeping as greedy."""
    B = h0.shape[0]
    ids = torch.full((B, 1), go_id, dtype=torch.int64)
    h = h0
    seq = []
    lengths = np.array([max_len] * B)
    for di in range(max_len):
        lp, h = decoder_teacher(P, prefix, emb_w, ids, h)
        g = -1.0 * torch.log(-1.0 * torch.log(torch.as_tensor(u[di])))
        ids = torch.max(lp[:, 0] + g, dim=-1, keepdim=True)[1]
        seq.append(ids)
        eos = (ids.view(-1) == eos_id).numpy()
        upd = (lengths > di) & eos
        lengths[upd] = len(seq)
    return torch.cat(seq, dim=1), lengths.tolist(), h


def decoder_beam(P, prefix, emb_w, h0, max_len, go_id, eos_id, beam_size, trace=False):
    """DecoderRNN GEN with gen_type='beam', laed/enc2dec/decoders.py:194-202,222-235,270-291, restated with the integer
    semantics of the torch the reference targets (top_id.div = floor division).  As in the reference the hidden states
    are NOT reordered by the back pointers.  The search LOOP is pinned by tests/golden/beam.npz (the reference's own loop
    runs under torch 2 and its per-step fed symbols and log-probs were recorded: make_goldens_beam.py); the
    back-tracking after it is PARITY UNPINNED -- the reference's dies at :289 (torch.cat of 0-d tensors).
    Returns (sequence [B,max_len] int64, lengths list of B*beam); with trace=True also (per-step symbols [L,B,K], per-step
    log-probs [L,B*K,V])."""
    B, K = h0.shape[0], beam_size
    h = h0.repeat(1, K).view(B * K, -1)                                  # repeat_state, base_modules.py:69-71
    ids = torch.full((B * K, 1), go_id, dtype=torch.int64)
    V = P[prefix + 'project.weight'].shape[0]
    cum_sum = None
    syms, bps, lps = [], [], []
    lengths = np.array([max_len] * (B * K))
    for di in range(max_len):
        lp, h = decoder_teacher(P, prefix, emb_w, ids, h)
        step = lp[:, 0]
        lps.append(step)
        if di == 0:
            seq_score = step.reshape(B, -1)[:, :V]
        else:
            seq_score = (cum_sum + step).reshape(B, -1)
        top_v, top_id = seq_score.topk(K)
        bps.append(top_id // V)
        ids = (top_id % V).reshape(-1, 1)
        cum_sum = top_v.reshape(-1, 1)
        syms.append(ids.view(B, K))
        eos = (ids.view(-1) == eos_id).numpy()
        lengths[(lengths > di) & eos] = di + 1
    best = cum_sum.view(B, K).topk(1)[1].view(-1)
    rows = torch.arange(B)
    out = torch.zeros(B, max_len, dtype=torch.int64)
    for di in range(max_len - 1, -1, -1):
        out[:, di] = syms[di][rows, best]
        best = bps[di][rows, best]
    if trace:
        return out, lengths.tolist(), torch.stack(syms), torch.stack(lps)
    return out, lengths.tolist()


# --------------------------------------------------------------------------------------
# models (functional; P = dict name -> tensor with the reference's state_dict keys)
# --------------------------------------------------------------------------------------
def _encode_latent(P, cfg, outputs, u, training, temperature, has_conn_bias):
    """Shared front half of DiVAE.forward / DiVST.forward (sent_models.py:96-119, 366-388)."""
    out_utts = torch.as_tensor(np.asarray(outputs)).long()              # model_bases.py:53-57
    B = out_utts.shape[0]
    emb = F.embedding(out_utts, P["embedding.weight"], padding_idx=PAD_ID)                  # PAD row is zero
    H = P['x_encoder.rnn.weight_hh_l0'].shape[1]
    _, x_last = gru_seq(emb, torch.zeros(B, H), P['x_encoder.rnn.weight_ih_l0'],
                        P['x_encoder.rnn.weight_hh_l0'], P['x_encoder.rnn.bias_ih_l0'],
                        P['x_encoder.rnn.bias_hh_l0'])                  # runs over PAD too
    qy_logits = F.linear(x_last, P['q_y.weight'], P['q_y.bias']).view(-1, cfg.k)
    log_qy = F.log_softmax(qy_logits, dim=1)
    sample_y, y_ids = gumbel_softmax(qy_logits, torch.as_tensor(u), temperature, hard=not training)
    sample_y = sample_y.view(-1, cfg.k * cfg.y_size)
    bias = P['dec_init_connector.linear.bias'] if has_conn_bias else None
    dec_init = F.linear(sample_y, P['dec_init_connector.linear.weight'], bias)   # [B,H]
    return out_utts, x_last, log_qy, y_ids.view(-1, cfg.y_size), dec_init


def divae_forward(P, cfg, outputs, u, training=True, drop_mask=None):
    """DiVAE.forward(mode=TEACH_FORCE), laed/models/sent_models.py:94-165.

    Returns dict(nll, reg_kl, mi, bpr, log_qy, y_ids, dec_init_state[1,B,H], x_last, tok_nll, total).
    total = nll + reg_kl (valid_loss, sent_models.py:87-92), computed out of place.
    """
    out_utts, x_last, log_qy, y_ids, dec_init = _encode_latent(P, cfg, outputs, u, training, 1.0, False)
    labels = out_utts[:, 1:].contiguous()
    dec_inputs = out_utts[:, :-1]
    dm = None if (drop_mask is None or not training) else torch.as_tensor(drop_mask)
    logp, _ = decoder_teacher(P, 'decoder.', P['embedding.weight'], dec_inputs, dec_init, dm, cfg.dropout)
    nll = nll_entropy(logp, labels, cfg.avg_type)
    reg_kl, bpr, mi = latent_losses(log_qy, cfg.y_size, cfg.k, cfg.use_mutual,
                                    P['log_py'] if getattr(cfg, 'train_prior', False) else None)
    total = nll + reg_kl if cfg.use_reg_kl else nll
    tok_nll = -logp.gather(2, labels.unsqueeze(2)).squeeze(2)
    return dict(nll=nll, reg_kl=reg_kl, mi=mi, bpr=bpr, log_qy=log_qy, y_ids=y_ids,
                dec_init_state=dec_init.unsqueeze(0), x_last=x_last, tok_nll=tok_nll, total=total,
                logp=logp)


def divst_forward(P, cfg, outputs, prevs, nexts, u, training=True, drop_masks=(None, None)):
    """DiVST.forward(mode=TEACH_FORCE), laed/models/sent_models.py:363-442 (kl_w = 1.0, :350)."""
    _, x_last, log_qy, y_ids, dec_init = _encode_latent(P, cfg, outputs, u, training, 1.0, True)
    res = {}
    for tag, utts, dm in (('prev', prevs, drop_masks[0]), ('next', nexts, drop_masks[1])):
        utts = torch.as_tensor(np.asarray(utts)).long()
        labels = utts[:, 1:].contiguous()
        dm = None if (dm is None or not training) else torch.as_tensor(dm)
        logp, _ = decoder_teacher(P, tag + '_decoder.', P['embedding.weight'], utts[:, :-1], dec_init,
                                  dm, cfg.dropout)
        res[tag + '_nll'] = nll_entropy(logp, labels, cfg.avg_type)
        res[tag + '_tok_nll'] = -logp.gather(2, labels.unsqueeze(2)).squeeze(2)
    reg_kl, bpr, mi = latent_losses(log_qy, cfg.y_size, cfg.k, cfg.use_mutual)
    total = res['prev_nll'] + res['next_nll']
    if cfg.use_reg_kl:
        total = total + reg_kl
    res.update(reg_kl=reg_kl, mi=mi, b_pr=bpr, log_qy=log_qy, y_ids=y_ids,
               dec_init_state=dec_init.unsqueeze(0), x_last=x_last, total=total)
    return res


def divae_greedy(P, cfg, outputs, u, go_id=2, eos_id=3):
    """DiVAE.forward(mode=GEN, gen_type='greedy') in eval(): hard one-hot latent, noise still drawn
    (sent_models.py:114 hard=not training; nn_lib.py:252-257), then 40-step free run."""
    _, _, _, _, dec_init = _encode_latent(P, cfg, outputs, u, False, 1.0, False)
    return decoder_greedy(P, 'decoder.', P['embedding.weight'], dec_init, cfg.max_dec_len, go_id, eos_id)


def _decode_codes(P, cfg, sample_y, go_id, eos_id):
    dec_init = F.linear(sample_y, P['dec_init_connector.linear.weight'], None)
    return decoder_greedy(P, 'decoder.', P['embedding.weight'], dec_init, cfg.max_dec_len, go_id, eos_id)


def divae_sweep(P, cfg, outputs, u, go_id=2, eos_id=3):
    """DiVAE.sweep (sent_models.py:167-218): codes from the first utterance's to the last one's, one variable at a time."""
    _, _, _, y_ids, _ = _encode_latent(P, cfg, outputs, u, False, 1.0, False)
    start, end = y_ids[0], y_ids[-1]
    rows = [start]
    for idx in range(cfg.y_size):
        m = torch.zeros(cfg.y_size, dtype=torch.long)
        m[:idx + 1] = 1
        rows.append((1 - m) * start + m * end)
    all_y = torch.stack(rows)
    sample_y = torch.zeros(all_y.numel(), cfg.k).scatter_(1, all_y.view(-1, 1), 1.0).view(-1, cfg.k * cfg.y_size)
    return _decode_codes(P, cfg, sample_y, go_id, eos_id), all_y


def divae_enumerate(P, cfg, repeat=1, go_id=2, eos_id=3):
    """DiVAE.enumerate (sent_models.py:220-248), row index y*k + k_id*repeat + r as in the reference."""
    y, k = cfg.y_size, cfg.k
    sample_y = torch.full((y * k * repeat, y, k), 1.0 / k)
    for y_id in range(y):
        for k_id in range(k):
            for r_id in range(repeat):
                idx = y_id * k + k_id * repeat + r_id
                sample_y[idx, y_id] = 0.0
                sample_y[idx, y_id, k_id] = 1.0
    return _decode_codes(P, cfg, sample_y.view(-1, k * y), go_id, eos_id)


def divae_exp_enumerate(P, cfg, go_id=2, eos_id=3):
    """DiVAE.exp_enumerate (sent_models.py:250-274): all k**y joint codes in itertools.product order."""
    import itertools
    y, k = cfg.y_size, cfg.k
    ids = torch.tensor([list(c) for c in itertools.product(*[range(k)] * y)], dtype=torch.long)
    sample_y = torch.zeros(ids.numel(), k).scatter_(1, ids.view(-1, 1), 1.0).view(-1, k * y)
    return _decode_codes(P, cfg, sample_y, go_id, eos_id), ids


def divst_greedy(P, cfg, outputs, u, go_id=2, eos_id=3):
    _, _, _, _, dec_init = _encode_latent(P, cfg, outputs, u, False, 1.0, True)
    return tuple(decoder_greedy(P, tag + '_decoder.', P['embedding.weight'], dec_init, cfg.max_dec_len,
                                go_id, eos_id) for tag in ('prev', 'next'))


def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults as used at laed/models/model_bases.py:122-125 (no weight decay,
    no amsgrad).  In-place on p, m, v; `step` is 1-based."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def trainable_keys(P, train_prior=False):
    """Parameters Adam updates: everything but the decoder.embedding alias; log_py gets no grad unless
    train_prior (sent_models.py:144-146) and Adam skips grad-less params."""
    return [k for k in P if not k.endswith('decoder.embedding.weight') and (train_prior or k != 'log_py')]


# --------------------------------------------------------------------------------------
# nn.Module form: the SAME torch modules the reference instantiates (nn.Embedding, nn.GRU,
# nn.Linear, F.log_softmax, F.nll_loss, torch.optim.Adam).  This is the "reference PyTorch CPU
# path" that bench.py times on the GPU box's host cores (cpu_baseline.kind = "port").
# --------------------------------------------------------------------------------------
class _Rnn(nn.Module):
    def __init__(self, E, H):
        super().__init__()
        self.rnn = nn.GRU(E, H, 1, batch_first=True)


class _Dec(nn.Module):
    def __init__(self, embedding, E, H, V):
        super().__init__()
        self.rnn = nn.GRU(E, H, 1, batch_first=True)
        self.embedding = embedding
        self.project = nn.Linear(H, V)


class _Conn(nn.Module):
    def __init__(self, i, o, bias):
        super().__init__()
        self.linear = nn.Linear(i, o, bias=bias)


class OracleDiVAE(nn.Module):
    """Module restatement of DiVAE (sent_models.py:20-165) with the reference's state_dict keys."""

    def __init__(self, vocab_size, cfg):
        super().__init__()
        self.cfg = cfg
        self.embedding = nn.Embedding(vocab_size, cfg.embed_size, padding_idx=PAD_ID)
        self.x_encoder = _Rnn(cfg.embed_size, cfg.enc_cell_size)
        self.q_y = nn.Linear(cfg.enc_cell_size, cfg.y_size * cfg.k)
        self.dec_init_connector = _Conn(cfg.y_size * cfg.k, cfg.dec_cell_size, False)
        self.decoder = _Dec(self.embedding, cfg.embed_size, cfg.dec_cell_size, vocab_size)
        self.log_py = nn.Parameter(torch.log(torch.ones(cfg.y_size, cfg.k) / cfg.k))

    def forward(self, outputs, u):
        cfg = self.cfg
        out_utts = torch.as_tensor(np.asarray(outputs)).long()
        _, x_last = self.x_encoder.rnn(self.embedding(out_utts))
        x_last = x_last.transpose(0, 1).contiguous().view(-1, cfg.enc_cell_size)
        qy_logits = self.q_y(x_last).view(-1, cfg.k)
        log_qy = F.log_softmax(qy_logits, dim=1)
        sample_y, _ = gumbel_softmax(qy_logits, torch.as_tensor(u), 1.0, hard=not self.training)
        h0 = self.dec_init_connector.linear(sample_y.view(-1, cfg.k * cfg.y_size)).unsqueeze(0)
        labels = out_utts[:, 1:].contiguous()
        x = F.dropout(self.embedding(out_utts[:, :-1]), cfg.dropout, self.training)
        out, _ = self.decoder.rnn(x, h0)
        logits = self.decoder.project(out.contiguous().view(-1, cfg.dec_cell_size))
        logp = F.log_softmax(logits, dim=1)
        nll = F.nll_loss(logp, labels.view(-1), ignore_index=PAD_ID)
        reg_kl, bpr, mi = latent_losses(log_qy, cfg.y_size, cfg.k, cfg.use_mutual,
                                        self.log_py if getattr(cfg, 'train_prior', False) else None)
        return dict(nll=nll, reg_kl=reg_kl, mi=mi, bpr=bpr, total=nll + reg_kl)
