#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE (read-only, /root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box).
Only the resulting .npz files (inputs + expected outputs) are committed; no
reference source is copied.  `nltk` is absent offline, and laed/utils.py:11
imports it at module scope for tokenisers that the model path never calls, so an
empty in-process stand-in is placed in sys.modules before the import (harness
side only; no reference file is touched) -- see SURVEY.md section 8c.

Noise injection (the reference seeds nothing):
  * Gumbel:  model.cat_connector.sample_gumbel is overridden on the instance
    with  -log(-log(u+eps)+eps)  for a recorded uniform u  (nn_lib.py:232-236).
  * decoder input dropout (decoders.py:156): the nn.Dropout instance is replaced
    by a module multiplying with a recorded keep mask / (1-p), which is the
    arithmetic torch's CPU dropout performs.

Loss terms are read BEFORE valid_loss (which aliases/mutates loss.nll in place,
sent_models.py:87-92).

usage:  python tests/golden/make_goldens.py   (writes next to this file)
"""
import os
import sys
import types
from argparse import Namespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    nltk = types.ModuleType("nltk")
    tok = types.ModuleType("nltk.tokenize")
    moses = types.ModuleType("nltk.tokenize.moses")
    moses.MosesDetokenizer = object
    nltk.RegexpTokenizer = object
    nltk.tokenize = tok
    tok.moses = moses
    sys.modules.update({"nltk": nltk, "nltk.tokenize": tok, "nltk.tokenize.moses": moses})
    sys.path.insert(0, "/root/reference")


_import_reference()
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from laed import criterions, nn_lib  # noqa: E402
from laed.enc2dec.decoders import GEN, TEACH_FORCE, DecoderRNN  # noqa: E402
from laed.enc2dec.encoders import EncoderRNN, RnnUttEncoder  # noqa: E402
from laed.models import sent_models  # noqa: E402

torch.set_num_threads(1)


class FakeCorpus(object):
    def __init__(self, vocab_size):
        self.vocab = ['<pad>', '<unk>', '<s>', '</s>'] + ['w%d' % i for i in range(4, vocab_size)]
        self.rev_vocab = {w: i for i, w in enumerate(self.vocab)}


def make_config(**kw):
    cfg = dict(embed_size=16, max_utt_len=8, num_layer=1, dropout=0.0, enc_cell_size=32,
               dec_cell_size=32, rnn_cell='gru', max_dec_len=10, use_attn=False, beam_size=4,
               utt_type='rnn', bi_enc_cell=False, attn_type='cat', fix_batch=False, y_size=4,
               k=5, use_gpu=False, avg_type='word', use_mutual=True, use_reg_kl=True,
               train_prior=False, init_lr=0.001, op='adam', momentum=0.0)
    cfg.update(kw)
    return Namespace(**cfg)


def make_batch(rs, B, T, V, min_len=3, ragged=True):
    """int32 [B,T]: <s>=2, ids in [4,V), </s>=3, zero-padded tails (data_loaders.py:283-291)."""
    out = np.zeros((B, T), dtype=np.int32)
    lens = []
    for b in range(B):
        L = T if (not ragged or b == 0) else int(rs.randint(min_len, T + 1))
        out[b, 0] = 2
        out[b, 1:L - 1] = rs.randint(4, V, size=L - 2)
        out[b, L - 1] = 3
        lens.append(L)
    return out, lens


class InjectedDropout(nn.Module):
    def __init__(self, p):
        super(InjectedDropout, self).__init__()
        self.p = p
        self.mask = None

    def forward(self, x):
        if self.mask is None or not self.training:
            return x
        return x * (self.mask / (1.0 - self.p))


def inject_gumbel(model, u):
    ut = torch.from_numpy(u)

    def sample_gumbel(logits, use_gpu, eps=1e-20):
        return -torch.log(-torch.log(ut + eps) + eps)
    model.cat_connector.sample_gumbel = sample_gumbel


def f32(t):
    return t.detach().cpu().numpy().astype(np.float32, copy=True)


def per_token_nll(logp, labels):
    return f32(-logp.gather(2, labels.unsqueeze(2)).squeeze(2))


def run_sent_model(kind, name, seed, V, B, T, cfg_kw, with_dropout_case):
    rs = np.random.RandomState(seed)
    torch.manual_seed(seed)
    config = make_config(max_utt_len=T, **cfg_kw)
    corpus = FakeCorpus(V)
    cls = sent_models.DiVAE if kind == 'divae' else sent_models.DiVST
    model = cls(corpus, config)
    if cfg_kw.get('train_prior'):       # start from a non-uniform learned prior so that its gradient path is exercised
        model.log_py.data.add_(torch.from_numpy(rs.normal(0.0, 0.5, size=tuple(model.log_py.shape)).astype(np.float32)))
    model.train()
    G = {}
    G['meta/torch_version'] = np.array(torch.__version__)
    G['meta/kind'] = np.array(kind)
    for k_, v_ in sorted(vars(config).items()):
        G['config/' + k_] = np.array(v_)
    G['config/vocab_size'] = np.array(V)
    sd = model.state_dict()
    G['meta/state_dict_keys'] = np.array(list(sd.keys()))
    for k_, v_ in sd.items():
        G['sd/' + k_] = f32(v_)

    outputs, lens = make_batch(rs, B, T, V)
    feed = {'outputs': outputs, 'output_lens': lens}
    G['in/outputs'] = outputs
    G['in/output_lens'] = np.array(lens, dtype=np.int32)
    if kind == 'divst':
        prevs, _ = make_batch(rs, B, T - 1, V)
        nexts, _ = make_batch(rs, B, T, V)
        feed['prevs'], feed['nexts'] = prevs, nexts
        G['in/prevs'], G['in/nexts'] = prevs, nexts
    R = B * config.y_size
    u_steps = rs.uniform(0.0, 1.0, size=(5, R, config.k)).astype(np.float32)
    G['in/u_steps'] = u_steps

    decoders = [model.decoder] if kind == 'divae' else [model.prev_decoder, model.next_decoder]
    cap = {}
    model.x_encoder.register_forward_hook(lambda m, i, o: cap.__setitem__('x_last', o[1]))
    for di, dec in enumerate(decoders):
        dec.register_forward_hook(lambda m, i, o, di=di: cap.__setitem__('dec%d' % di, o))

    # ---- step 0: forward (train mode, dropout 0) + backward
    inject_gumbel(model, u_steps[0])
    loss = model(feed, mode=TEACH_FORCE, return_latent=True)
    keys = ['nll', 'reg_kl', 'mi', 'bpr'] if kind == 'divae' else ['prev_nll', 'next_nll', 'reg_kl', 'mi', 'b_pr']
    for k_ in keys:
        G['out/' + k_] = f32(loss[k_])          # read BEFORE valid_loss
    G['out/log_qy'] = f32(loss['log_qy'])
    G['out/y_ids'] = loss['y_ids'].numpy().astype(np.int64)
    G['out/dec_init_state'] = f32(loss['dec_init_state'])
    G['out/x_last'] = f32(cap['x_last'])
    if kind == 'divae':
        labels = torch.from_numpy(outputs.astype(np.int64))[:, 1:]
        G['out/tok_nll'] = per_token_nll(cap['dec0'][0], labels)
        if V <= 64:
            G['out/dec_logp'] = f32(cap['dec0'][0])
    else:
        G['out/prev_tok_nll'] = per_token_nll(cap['dec0'][0], torch.from_numpy(feed['prevs'].astype(np.int64))[:, 1:])
        G['out/next_tok_nll'] = per_token_nll(cap['dec1'][0], torch.from_numpy(feed['nexts'].astype(np.int64))[:, 1:])
    model.zero_grad()
    model.backward(0, loss)
    G['out/total_loss'] = f32(model.valid_loss(model(feed, mode=TEACH_FORCE)))
    for n_, p_ in model.named_parameters():
        G['grad/' + n_] = f32(p_.grad) if p_.grad is not None else np.zeros(0, np.float32)

    # ---- 5 Adam steps in main.py:104-113 order; a fresh u per step
    opt = model.get_optimizer(config)
    losses = []
    for s in range(5):
        inject_gumbel(model, u_steps[s])
        opt.zero_grad()
        loss = model(feed, mode=TEACH_FORCE)
        losses.append([float(loss[k_]) for k_ in keys])
        model.backward(s, loss)
        opt.step()
        if s in (0, 4):
            for n_, p_ in model.named_parameters():
                G['p%d/' % (s + 1) + n_] = f32(p_)
    G['out/step_losses'] = np.array(losses, dtype=np.float32)
    G['meta/loss_keys'] = np.array(keys)

    # ---- eval-mode teacher-forced loss + greedy decode on the 5-step model (noise still drawn in eval)
    model.eval()
    u_gen = rs.uniform(0.0, 1.0, size=(R, config.k)).astype(np.float32)
    G['in/u_gen'] = u_gen
    inject_gumbel(model, u_gen)
    with torch.no_grad():
        loss = model(feed, mode=TEACH_FORCE, return_latent=True)
        for k_ in keys:
            G['eval/' + k_] = f32(loss[k_])
        G['eval/y_ids'] = loss['y_ids'].numpy().astype(np.int64)
        model(feed, mode=GEN, gen_type='greedy')
    for di in range(len(decoders)):
        dec_outs, _, ret = cap['dec%d' % di]
        seq = torch.cat(ret[DecoderRNN.KEY_SEQUENCE], dim=1)          # [B, max_dec_len]
        top2 = dec_outs.topk(2, dim=2)[0]                             # [B, L, 2]
        tag = '' if kind == 'divae' else ('prev_' if di == 0 else 'next_')
        G['gen/%ssequence' % tag] = seq.numpy().astype(np.int64)
        G['gen/%smargin' % tag] = f32(top2[:, :, 0] - top2[:, :, 1])
        G['gen/%slength' % tag] = np.array(ret[DecoderRNN.KEY_LENGTH], dtype=np.int64)

    # ---- one extra train-mode forward/backward with dropout p=0.3 and an injected keep mask
    if with_dropout_case:
        torch.manual_seed(seed + 1)
        config2 = make_config(max_utt_len=T, **dict(cfg_kw, dropout=0.3))
        model2 = cls(corpus, config2)
        model2.load_state_dict({k_: torch.from_numpy(G['sd/' + k_]) for k_ in sd.keys()})
        model2.train()
        decs2 = [model2.decoder] if kind == 'divae' else [model2.prev_decoder, model2.next_decoder]
        for di, dec in enumerate(decs2):
            Td = (T - 1) if (kind == 'divae' or di == 1) else (T - 2)
            mask = (rs.uniform(size=(B, Td, config.embed_size)) >= 0.3).astype(np.float32)
            G['in/drop_mask%d' % di] = mask
            inj = InjectedDropout(0.3)
            inj.mask = torch.from_numpy(mask)
            dec.input_dropout = inj
        inject_gumbel(model2, u_steps[0])
        loss = model2(feed, mode=TEACH_FORCE)
        for k_ in keys:
            G['drop/' + k_] = f32(loss[k_])
        model2.zero_grad()
        model2.backward(0, loss)
        for n_, p_ in model2.named_parameters():
            if p_.grad is not None and ('embedding' in n_ or 'x_encoder.rnn.weight_hh' in n_):
                G['drop_grad/' + n_] = f32(p_.grad)

    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **G)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.0), {k_: float(G['out/' + k_]) for k_ in keys})


def run_modules(seed=7):
    rs = np.random.RandomState(seed)
    torch.manual_seed(seed)
    G = {'meta/torch_version': np.array(torch.__version__)}

    # GumbelConnector soft / hard (nn_lib.py:228-261)
    logits = rs.randn(12, 5).astype(np.float32) * 2
    u = rs.uniform(size=(12, 5)).astype(np.float32)
    gc = nn_lib.GumbelConnector()
    ut = torch.from_numpy(u)
    gc.sample_gumbel = lambda lg, use_gpu, eps=1e-20: -torch.log(-torch.log(ut + eps) + eps)
    for temp in (1.0, 0.5):
        lt = torch.from_numpy(logits).requires_grad_(True)
        y, ids = gc(lt, temp, False, hard=False, return_max_id=True)
        w = torch.from_numpy(np.arange(60, dtype=np.float32).reshape(12, 5) / 60.0)
        (y * w).sum().backward()
        G['gumbel/y_t%g' % temp] = f32(y)
        G['gumbel/ids_t%g' % temp] = ids.numpy()
        G['gumbel/dlogits_t%g' % temp] = f32(lt.grad)
    yh, ids = gc(torch.from_numpy(logits), 1.0, False, hard=True, return_max_id=True)
    G['gumbel/logits'], G['gumbel/u'] = logits, u
    G['gumbel/y_hard'], G['gumbel/ids_hard'] = f32(yh), ids.numpy()

    # LinearConnector (nn_lib.py:138-158)
    for has_bias in (False, True):
        lc = nn_lib.LinearConnector(20, 8, False, has_bias=has_bias)
        x = rs.randn(6, 20).astype(np.float32)
        tag = 'linconn_b%d/' % int(has_bias)
        G[tag + 'x'] = x
        G[tag + 'w'] = f32(lc.linear.weight)
        if has_bias:
            G[tag + 'b'] = f32(lc.linear.bias)
        G[tag + 'out'] = f32(lc(torch.from_numpy(x)))

    # NLLEntropy, all 4 avg types (criterions.py:25-72)
    logp = torch.log_softmax(torch.from_numpy(rs.randn(4, 6, 11).astype(np.float32)), dim=2)
    labels = rs.randint(1, 11, size=(4, 6)).astype(np.int64)
    labels[1, 4:] = 0
    labels[3, 2:] = 0
    G['nll/logp'], G['nll/labels'] = f32(logp), labels
    for avg in (None, 'seq', 'real_word', 'word'):
        crit = criterions.NLLEntropy(0, Namespace(avg_type=avg, use_gpu=False))
        G['nll/out_%s' % avg] = f32(crit(logp, torch.from_numpy(labels)))

    # CatKLLoss / Entropy (criterions.py:93-147)
    lq = torch.log_softmax(torch.from_numpy(rs.randn(8, 5).astype(np.float32)), dim=1)
    lp = torch.log_softmax(torch.from_numpy(rs.randn(8, 5).astype(np.float32)), dim=1)
    G['catkl/log_qy'], G['catkl/log_py'] = f32(lq), f32(lp)
    G['catkl/unit'] = f32(criterions.CatKLLoss()(lq, lp, 4, unit_average=True))
    G['catkl/batch'] = f32(criterions.CatKLLoss()(lq, lp, 4, unit_average=False))
    G['entropy/unit'] = f32(criterions.Entropy()(lq, unit_average=True))
    G['entropy/batch'] = f32(criterions.Entropy()(lq, 4, unit_average=False))

    # EncoderRNN uni / bi with and without init_state (encoders.py:33-68)
    for bi in (False, True):
        enc = EncoderRNN(6, 10, bidirection=bi)
        x = rs.randn(3, 7, 6).astype(np.float32)
        h0 = rs.randn(2 if bi else 1, 3, 10).astype(np.float32)
        tag = 'enc_bi%d/' % int(bi)
        for n_, p_ in enc.state_dict().items():
            G[tag + 'sd/' + n_] = f32(p_)
        out, hid = enc(torch.from_numpy(x))
        out2, hid2 = enc(torch.from_numpy(x), init_state=torch.from_numpy(h0))
        G[tag + 'x'], G[tag + 'h0'] = x, h0
        G[tag + 'out'], G[tag + 'hid'] = f32(out), f32(hid)
        G[tag + 'out_h0'], G[tag + 'hid_h0'] = f32(out2), f32(hid2)

    # DecoderRNN teacher-forced log-probs + greedy (decoders.py:118-297)
    dec = DecoderRNN(13, 6, 5, 9, 2, 3, n_layers=1, rnn_cell='gru', use_gpu=False)
    dec.eval()
    ids = rs.randint(0, 13, size=(4, 5)).astype(np.int64)
    h0 = rs.randn(1, 4, 9).astype(np.float32)
    for n_, p_ in dec.state_dict().items():
        G['dec/sd/' + n_] = f32(p_)
    with torch.no_grad():
        lpb, hid, _ = dec(4, torch.from_numpy(ids), torch.from_numpy(h0), mode=TEACH_FORCE)
        gout, ghid, ret = dec(4, None, torch.from_numpy(h0), mode=GEN, gen_type='greedy')
    G['dec/ids'], G['dec/h0'] = ids, h0
    G['dec/logp'], G['dec/hid'] = f32(lpb), f32(hid)
    G['dec/gen_logp'], G['dec/gen_hid'] = f32(gout), f32(ghid)
    G['dec/gen_seq'] = torch.cat(ret[DecoderRNN.KEY_SEQUENCE], dim=1).numpy()
    G['dec/gen_len'] = np.array(ret[DecoderRNN.KEY_LENGTH], dtype=np.int64)

    # RnnUttEncoder bi-dir, with / without attention pooling (encoders.py:71-137)
    for attn in (False, True):
        ue = RnnUttEncoder(8, 0.0, bidirection=True, use_attn=attn, vocab_size=17, embed_dim=6)
        utts = rs.randint(0, 17, size=(2, 3, 5)).astype(np.int64)
        tag = 'uttenc_attn%d/' % int(attn)
        for n_, p_ in ue.state_dict().items():
            G[tag + 'sd/' + n_] = f32(p_)
        G[tag + 'utts'] = utts
        emb, outs, last, a = ue(torch.from_numpy(utts), return_all=True)
        G[tag + 'out'] = f32(emb)
        G[tag + 'enc_outs'] = f32(outs)
        if attn:
            G[tag + 'attn'] = f32(a)

    path = os.path.join(HERE, 'modules.npz')
    np.savez_compressed(path, **G)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.0))


if __name__ == '__main__':
    tiny = dict(embed_size=16, enc_cell_size=32, dec_cell_size=32, y_size=4, k=5, max_dec_len=10)
    mid = dict(embed_size=32, enc_cell_size=64, dec_cell_size=64, y_size=6, k=10, max_dec_len=12)
    # the two non-default regularisers of sent_models.py:143-151: a trained prior, and the per-row KL
    run_sent_model('divae', 'divae_tiny_prior', 15, V=50, B=6, T=8, cfg_kw=dict(tiny, train_prior=True),
                   with_dropout_case=False)
    run_sent_model('divae', 'divae_tiny_rowkl', 16, V=50, B=6, T=8, cfg_kw=dict(tiny, use_mutual=False),
                   with_dropout_case=False)
    if len(sys.argv) > 1 and sys.argv[1] == 'regularisers':
        sys.exit(0)
    run_sent_model('divae', 'divae_tiny', 11, V=50, B=6, T=8, cfg_kw=tiny, with_dropout_case=True)
    run_sent_model('divae', 'divae_mid', 12, V=400, B=12, T=14, cfg_kw=mid, with_dropout_case=False)
    run_sent_model('divst', 'divst_tiny', 13, V=50, B=6, T=8, cfg_kw=tiny, with_dropout_case=True)
    run_sent_model('divst', 'divst_mid', 14, V=400, B=12, T=14, cfg_kw=mid, with_dropout_case=False)
    run_modules()
