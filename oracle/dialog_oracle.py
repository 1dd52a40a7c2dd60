"""CPU oracle for AE-ED / ST-ED -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Same rules as laed_oracle.py: imported only by tests/ (and smoke / bench baselines), never by the product.
A fresh fp32 torch-CPU restatement of laed/models/dialog_models.py (paths relative to /root/reference), PINNED by
tests/golden/{aeed_tiny,aeed_mid,sted_tiny,sted_mid}.npz which were produced by running the reference itself
(tests/golden/make_goldens_dialog.py).  tests/test_oracle_dialog_vs_golden.py checks it against those files.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .laed_oracle import (PAD_ID, cat_kl, decoder_greedy, decoder_teacher, entropy, gru_seq, gumbel_softmax,
                          nll_entropy)


def _gru(P, prefix, x, h0=None, reverse=False):
    sfx = '_reverse' if reverse else ''
    H = P[prefix + 'weight_hh_l0' + sfx].shape[1]
    if h0 is None:
        h0 = torch.zeros(x.shape[0], H)
    return gru_seq(x, h0, P[prefix + 'weight_ih_l0' + sfx], P[prefix + 'weight_hh_l0' + sfx],
                   P[prefix + 'bias_ih_l0' + sfx], P[prefix + 'bias_hh_l0' + sfx], reverse=reverse)


def utt_encoder(P, ctx_utts, use_attn):
    """RnnUttEncoder.forward, laed/enc2dec/encoders.py:101-137 (bidirectional GRU, feats=None, init_state=None).

    ctx_utts [B,C,T] int64 -> [B,C,2*utt_cell].  PAD positions and all-PAD context slots are consumed like any other.
    """
    B, C, T = ctx_utts.shape
    emb = F.embedding(ctx_utts.reshape(-1, T), P['embedding.weight'], padding_idx=PAD_ID)
    of, hf = _gru(P, 'utt_encoder.rnn.rnn.', emb)
    ob, hb = _gru(P, 'utt_encoder.rnn.rnn.', emb, reverse=True)
    enc_outs = torch.cat([of, ob], dim=2)
    if use_attn:
        fc1 = torch.tanh(F.linear(enc_outs, P['utt_encoder.key_w.weight'], P['utt_encoder.key_w.bias']))
        attn = F.linear(fc1, P['utt_encoder.query.weight'], P['utt_encoder.query.bias']).squeeze(2)
        attn = F.softmax(attn, dim=1).unsqueeze(2)
        utt = torch.sum(attn * enc_outs, dim=1)
    else:
        utt = torch.cat([hf, hb], dim=1)          # enc_last [2,N,H] -> transpose(0,1).view(-1, 2H)
    return utt.view(B, C, -1)


def recognition(P, cfg, x_emb, u, greedy, training):
    """LAED.qzx_forward, dialog_models.py:22-40, on already-embedded inputs x_emb [B,T,E].

    greedy (kl_w == 1.0 and greedy_q): GreedyConnector = argmax one-hot of the logits, nn_lib.py:264-282;
    else GumbelConnector(hard = not training).  Returns (qy_logits [B*y,k], sample_y [B,y*k], y_ids [B,y]).
    """
    _, x_last = _gru(P, 'x_encoder.rnn.', x_emb)
    qy_logits = F.linear(x_last, P['q_y.weight'], P['q_y.bias']).view(-1, cfg.k)
    if greedy:
        y_ids = torch.max(qy_logits, dim=1, keepdim=True)[1]
        sample_y = torch.zeros_like(qy_logits).scatter_(1, y_ids, 1.0)
    else:
        sample_y, y_ids = gumbel_softmax(qy_logits, torch.as_tensor(u), 1.0, hard=not training)
    return qy_logits, sample_y.view(-1, cfg.k * cfg.y_size), y_ids.view(-1, cfg.y_size)


def context_prior(P, cfg, contexts):
    """Context encoder + prior network, dialog_models.py:331-338.  -> (c_last [B,H], py_logits [B*y,k], log_py)."""
    ctx_utts = torch.as_tensor(np.asarray(contexts)).long()
    c_inputs = utt_encoder(P, ctx_utts, cfg.utt_type == 'attn_rnn')
    _, c_last = _gru(P, 'ctx_encoder.rnn.', c_inputs)
    fc1 = torch.tanh(F.linear(c_last, P['p_fc1.weight'], P['p_fc1.bias']))
    py_logits = F.linear(fc1, P['p_y.weight'], P['p_y.bias']).view(-1, cfg.k)
    return c_last, py_logits, F.log_softmax(py_logits, dim=1)


def attribute_head(P, cfg, logp, labels, y_id):
    """Attribute forcing, dialog_models.py:417-432 (AeED) / :721-741 (StED): re-encode the expected embedding of
    the decoder's output distribution with the recognition net and classify the latent it was conditioned on."""
    B, T, V = logp.shape
    x_emb_w = P['x_embedding.weight']
    pad = x_emb_w[PAD_ID].view(1, 1, -1).repeat(B, T, 1)
    mask = torch.sign(labels).float().unsqueeze(2)
    emb = torch.matmul(torch.exp(logp.reshape(-1, V)), x_emb_w).view(B, T, -1)
    valid = mask * emb + (1.0 - mask) * pad
    _, x_last = _gru(P, 'x_encoder.rnn.', valid)
    qy_logits = F.linear(x_last, P['q_y.weight'], P['q_y.bias']).view(-1, cfg.k)
    attribute_nll = F.cross_entropy(qy_logits, y_id.reshape(-1))
    adv_err = (torch.max(qy_logits, dim=1)[1] != y_id.reshape(-1)).float().mean()
    return attribute_nll, adv_err


def _latent_terms(cfg, qy_logits):
    log_qy = F.log_softmax(qy_logits, dim=1)
    avg = torch.log(torch.exp(log_qy.view(-1, cfg.y_size, cfg.k)).mean(dim=0) + 1e-15)
    log_uniform = math.log(1.0 / cfg.k)
    reg_kl = cat_kl(avg, log_uniform, unit_average=True)
    mi = entropy(avg, unit_average=True) - entropy(log_qy, unit_average=True)
    return log_qy, reg_kl, mi


def _encdec_terms(P, cfg, feed, sample_y, y_id, log_qy, res):
    out_utts = torch.as_tensor(np.asarray(feed['outputs'])).long()
    labels = out_utts[:, 1:].contiguous()
    c_last, py_logits, log_py = context_prior(P, cfg, feed['contexts'])
    dec_init = F.linear(sample_y.detach(), P['c_init_connector.linear.weight'],
                        P['c_init_connector.linear.bias']) + c_last
    logp, _ = decoder_teacher(P, 'decoder.', P['embedding.weight'], out_utts[:, :-1], dec_init)
    res['nll'] = nll_entropy(logp, labels, cfg.avg_type)
    res['pi_kl'] = cat_kl(log_qy.detach(), log_py, unit_average=True)
    res['pi_nll'] = F.cross_entropy(py_logits, y_id.reshape(-1))
    res['pi_err'] = (torch.max(py_logits, dim=1)[1] != y_id.reshape(-1)).float().mean()
    if cfg.use_attribute:
        res['attribute_nll'], res['adv_err'] = attribute_head(P, cfg, logp, labels, y_id)
    res.update(log_py=log_py, log_qy=log_qy, dec_init_state=dec_init.unsqueeze(0), y_ids=y_id)
    return res


def aeed_forward(P, cfg, feed, u, kl_w=0.0, training=True):
    """AeED.forward(mode=TEACH_FORCE), dialog_models.py:320-447 (dropout 0)."""
    out_utts = torch.as_tensor(np.asarray(feed['outputs'])).long()
    x_emb = F.embedding(out_utts[:, 1:], P['x_embedding.weight'])          # no padding_idx (dialog_models.py:203)
    qy_logits, sample_y, y_id = recognition(P, cfg, x_emb, u, kl_w == 1.0 and cfg.greedy_q, training)
    dec_init = F.linear(sample_y, P['x_init_connector.linear.weight'], P['x_init_connector.linear.bias'])
    vae_logp, _ = decoder_teacher_nopad(P, 'x_decoder.', P['x_embedding.weight'], out_utts[:, :-1], dec_init)
    log_qy, reg_kl, mi = _latent_terms(cfg, qy_logits)
    res = dict(vae_nll=nll_entropy(vae_logp, out_utts[:, 1:].contiguous(), cfg.avg_type), reg_kl=reg_kl, mi=mi)
    return _encdec_terms(P, cfg, feed, sample_y, y_id, log_qy, res)


def sted_forward(P, cfg, feed, u, kl_w=0.0, training=True):
    """StED.forward(mode=TEACH_FORCE), dialog_models.py:612-757 (dropout 0)."""
    out_utts = torch.as_tensor(np.asarray(feed['outputs'])).long()
    x_emb = F.embedding(out_utts[:, 1:], P['x_embedding.weight'])
    qy_logits, sample_y, y_id = recognition(P, cfg, x_emb, u, kl_w == 1.0 and cfg.greedy_q, training)
    dec_init = F.linear(sample_y, P['x_init_connector.linear.weight'], P['x_init_connector.linear.bias'])
    log_qy, reg_kl, mi = _latent_terms(cfg, qy_logits)
    res = dict(reg_kl=reg_kl, mi=mi)
    for tag in ('prev', 'next'):
        utts = torch.as_tensor(np.asarray(feed[tag + 's'])).long()
        logp, _ = decoder_teacher_nopad(P, tag + '_decoder.', P['x_embedding.weight'], utts[:, :-1], dec_init)
        res['vst_%s_nll' % tag] = nll_entropy(logp, utts[:, 1:].contiguous(), cfg.avg_type)
    return _encdec_terms(P, cfg, feed, sample_y, y_id, log_qy, res)


def decoder_teacher_nopad(P, prefix, emb_w, dec_inputs, h0):
    """decoder_teacher for decoders that share x_embedding, which has NO padding_idx: the PAD row is an ordinary
    trainable row (dialog_models.py:203,224)."""
    x = F.embedding(dec_inputs, emb_w)
    out, hT = gru_seq(x, h0, P[prefix + 'rnn.weight_ih_l0'], P[prefix + 'rnn.weight_hh_l0'],
                      P[prefix + 'rnn.bias_ih_l0'], P[prefix + 'rnn.bias_hh_l0'])
    logits = F.linear(out, P[prefix + 'project.weight'], P[prefix + 'project.bias'])
    return F.log_softmax(logits, dim=2), hT


def total_loss(kind, cfg, res, phase_b):
    """valid_loss, dialog_models.py:278-307 (AeED) / :549-585 (StED), out of place."""
    if not phase_b:
        if kind == 'aeed':
            return res['vae_nll'] + res['reg_kl']
        t = res['vst_prev_nll'] + res['vst_next_nll']
        return t + res['reg_kl'] if cfg.use_reg_kl else t
    t = res['nll'] + (res['pi_nll'] if cfg.greedy_q else res['pi_kl'])
    if cfg.use_attribute:
        t = t + (0.1 if kind == 'aeed' else 1.0) * res['attribute_nll']
    return t


FROZEN_PREFIXES = {'aeed': ('x_embedding.', 'x_encoder.', 'q_y.', 'x_init_connector.', 'x_decoder.'),
                   'sted': ('x_embedding.', 'x_encoder.', 'q_y.', 'x_init_connector.', 'prev_decoder.',
                            'next_decoder.')}


def dialog_greedy(P, cfg, feed, u_prior, go_id=2, eos_id=3):
    """forward(mode=GEN, gen_type='greedy', sample_n=1) in eval(): a hard Gumbel sample from the PRIOR network
    conditions the response decoder (dialog_models.py:374-391).  -> (sequence, lengths, logp, y_id, log_py)."""
    c_last, py_logits, log_py = context_prior(P, cfg, feed['contexts'])
    sample_y, y_id = gumbel_softmax(py_logits, torch.as_tensor(u_prior), 1.0, hard=True)
    dec_init = F.linear(sample_y.view(-1, cfg.k * cfg.y_size), P['c_init_connector.linear.weight'],
                        P['c_init_connector.linear.bias']) + c_last
    seq, lengths, logp = decoder_greedy(P, 'decoder.', P['embedding.weight'], dec_init, cfg.max_dec_len, go_id, eos_id)
    return seq, lengths, logp, y_id.view(-1, cfg.y_size), log_py
