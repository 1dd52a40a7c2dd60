#!/usr/bin/env python3
"""Golden vectors for AE-ED / ST-ED (laed/models/dialog_models.py:190-757), made by RUNNING THE REFERENCE on CPU.

Same rules as make_goldens.py (whose import shim, FakeCorpus and helpers are reused): build container only, only
inputs + expected outputs are committed.  Both training phases are recorded:

  phase A (kl_w = 0, batch_cnt <= freeze_step): total = VAE/VST loss, Gumbel sample from the recognition net
  flip    (first backward with batch_cnt > freeze_step): total = encoder-decoder loss on a phase-A forward; the
          recognition side is frozen by valid_loss (dialog_models.py:289-303 / :565-581)
  phase B (kl_w = 1, greedy_q): argmax one-hot from the recognition net, fresh optimizer over the unfrozen params
          (main.py:106-110)

usage:  python tests/golden/make_goldens_dialog.py [aeed_tiny|sted_tiny|aeed_mid|sted_mid ...]
"""
import os
from argparse import Namespace

import numpy as np

import make_goldens as mg          # noqa: F401  (sets up the reference import + nltk stand-in)
import torch
from laed.enc2dec.decoders import GEN, TEACH_FORCE, DecoderRNN
from laed.models import dialog_models

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = mg.f32

LOSS_KEYS = {'aeed': ['nll', 'pi_kl', 'attribute_nll', 'vae_nll', 'reg_kl', 'mi', 'pi_nll', 'pi_err', 'adv_err'],
             'sted': ['nll', 'pi_kl', 'pi_nll', 'attribute_nll', 'vst_prev_nll', 'vst_next_nll', 'reg_kl', 'mi',
                      'pi_err', 'adv_err']}


def make_config(**kw):
    cfg = dict(embed_size=16, max_utt_len=8, num_layer=1, dropout=0.0, utt_cell_size=8, ctx_cell_size=32,
               dec_cell_size=32, rnn_cell='gru', max_dec_len=10, use_attn=False, beam_size=4, utt_type='attn_rnn',
               bi_ctx_cell=False, attn_type='cat', fix_batch=False, y_size=3, k=5, use_gpu=False, avg_type='word',
               use_attribute=True, use_reg_kl=True, greedy_q=True, freeze_step=100, init_lr=0.001, op='adam',
               momentum=0.0)
    cfg.update(kw)
    return Namespace(**cfg)


def make_contexts(rs, B, C, T, V):
    ctx = np.zeros((B, C, T), dtype=np.int32)
    lens = []
    for b in range(B):
        n = C if b == 0 else int(rs.randint(1, C + 1))
        for c in range(n):
            row, _ = mg.make_batch(rs, 1, T, V)
            if c % 2 == 1:                       # some short utterances
                L = int(rs.randint(3, T + 1))
                row[0, L - 1] = 3
                row[0, L:] = 0
            ctx[b, c] = row[0]
        lens.append(n)
    return ctx, np.array(lens)


def record_losses(G, tag, loss, keys):
    for k_ in keys:
        if loss[k_] is not None:
            G[tag + k_] = f32(loss[k_])


def record_grads(G, tag, model):
    for n_, p_ in model.named_parameters():
        G[tag + n_] = f32(p_.grad) if p_.grad is not None else np.zeros(0, np.float32)


def run_dialog_model(kind, name, seed, V, B, C, T, cfg_kw):
    rs = np.random.RandomState(seed)
    torch.manual_seed(seed)
    config = make_config(max_utt_len=T, **cfg_kw)
    corpus = mg.FakeCorpus(V)
    cls = dialog_models.AeED if kind == 'aeed' else dialog_models.StED
    model = cls(corpus, config)
    model.train()
    keys = LOSS_KEYS[kind]
    G = {'meta/torch_version': np.array(torch.__version__), 'meta/kind': np.array(kind),
         'meta/loss_keys': np.array(keys)}
    for k_, v_ in sorted(vars(config).items()):
        G['config/' + k_] = np.array(v_)
    G['config/vocab_size'] = np.array(V)
    sd = model.state_dict()
    G['meta/state_dict_keys'] = np.array(list(sd.keys()))
    for k_, v_ in sd.items():
        G['sd/' + k_] = f32(v_)

    outputs, lens = mg.make_batch(rs, B, T, V)
    contexts, ctx_lens = make_contexts(rs, B, C, T, V)
    feed = {'outputs': outputs, 'output_lens': np.array(lens), 'contexts': contexts, 'context_lens': ctx_lens}
    if kind == 'sted':
        feed['prevs'], _ = mg.make_batch(rs, B, T - 1, V)
        feed['nexts'], _ = mg.make_batch(rs, B, T, V)
    for k_, v_ in feed.items():
        G['in/' + k_] = np.asarray(v_).astype(np.int32)
    R = B * config.y_size
    u_steps = rs.uniform(0.0, 1.0, size=(4, R, config.k)).astype(np.float32)
    G['in/u_steps'] = u_steps

    # ---- phase A: forward + backward, then 3 Adam steps
    mg.inject_gumbel(model, u_steps[0])
    loss = model(feed, mode=TEACH_FORCE, return_latent=True)
    record_losses(G, 'A/', loss, keys)
    for k_ in ('log_py', 'log_qy', 'dec_init_state'):
        G['A/' + k_] = f32(loss[k_])
    G['A/y_ids'] = loss['y_ids'].numpy().astype(np.int64)
    model.zero_grad()
    model.backward(0, loss)
    record_grads(G, 'A_grad/', model)
    opt = model.get_optimizer(config)
    stepl = []
    for s in range(3):
        mg.inject_gumbel(model, u_steps[s])
        opt.zero_grad()
        loss = model(feed, mode=TEACH_FORCE)
        stepl.append([float(loss[k_]) for k_ in keys])
        model.backward(s, loss)
        opt.step()
    G['A/step_losses'] = np.array(stepl, dtype=np.float32)
    for n_, p_ in model.named_parameters():
        G['A_p3/' + n_] = f32(p_)

    # ---- flip step (main.py:104-113 order): forward still in phase A, backward past freeze_step
    mg.inject_gumbel(model, u_steps[3])
    opt.zero_grad()
    loss = model(feed, mode=TEACH_FORCE)
    record_losses(G, 'flip/', loss, keys)
    model.backward(config.freeze_step + 1, loss)
    assert model.kl_w == 1.0 and model.flush_valid
    record_grads(G, 'flip_grad/', model)
    opt.step()
    for n_, p_ in model.named_parameters():
        G['flip_p/' + n_] = f32(p_)
    G['flip/requires_grad'] = np.array([int(p_.requires_grad) for _, p_ in model.named_parameters()], dtype=np.int32)

    # ---- phase B: greedy recognition sample, fresh optimizer (flush_valid), 3 steps
    batch_cnt = config.freeze_step + 2
    stepl = []
    for s in range(3):
        opt.zero_grad()
        loss = model(feed, mode=TEACH_FORCE, return_latent=(s == 0))
        if model.flush_valid:
            model.flush_valid = False
            opt = model.get_optimizer(config)
        if s == 0:
            record_losses(G, 'B/', loss, keys)
            G['B/y_ids'] = loss['y_ids'].numpy().astype(np.int64)
            G['B/log_py'] = f32(loss['log_py'])
            G['B/dec_init_state'] = f32(loss['dec_init_state'])
        stepl.append([float(loss[k_]) for k_ in keys])
        model.zero_grad()
        model.backward(batch_cnt + s, loss)
        if s == 0:
            record_grads(G, 'B_grad/', model)
        opt.step()
    G['B/step_losses'] = np.array(stepl, dtype=np.float32)
    for n_, p_ in model.named_parameters():
        G['B_p3/' + n_] = f32(p_)

    # ---- eval: teacher-forced losses + model_sel_loss, greedy GEN with an injected prior-sampling u
    model.eval()
    cap = {}
    model.decoder.register_forward_hook(lambda m, i, o: cap.__setitem__('dec', o))
    u_gen = rs.uniform(0.0, 1.0, size=(R, config.k)).astype(np.float32)
    G['in/u_gen'] = u_gen
    mg.inject_gumbel(model, u_gen)
    with torch.no_grad():
        loss = model(feed, mode=TEACH_FORCE)
        record_losses(G, 'eval/', loss, keys)
        G['eval/model_sel_loss'] = f32(model.model_sel_loss(loss, batch_cnt))
        ret, labels = model(feed, mode=GEN, gen_type='greedy')
    dec_outs = cap['dec'][0]
    top2 = dec_outs.topk(2, dim=2)[0]
    G['gen/sequence'] = torch.cat(ret[DecoderRNN.KEY_SEQUENCE], dim=1).numpy().astype(np.int64)
    G['gen/margin'] = f32(top2[:, :, 0] - top2[:, :, 1])
    G['gen/length'] = np.array(ret[DecoderRNN.KEY_LENGTH], dtype=np.int64)
    G['gen/y_ids'] = ret[DecoderRNN.KEY_LATENT].numpy().astype(np.int64)
    G['gen/qy_ids'] = ret[DecoderRNN.KEY_RECOG_LATENT].numpy().astype(np.int64)
    G['gen/log_py'] = f32(ret[DecoderRNN.KEY_POLICY])
    G['gen/labels'] = labels.numpy().astype(np.int64)

    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **G)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.0),
          {k_: float(G['A/' + k_]) for k_ in keys if 'A/' + k_ in G})


JOBS = {
    'aeed_tiny': lambda: run_dialog_model('aeed', 'aeed_tiny', 21, V=50, B=5, C=4, T=8, cfg_kw={}),
    'sted_tiny': lambda: run_dialog_model('sted', 'sted_tiny', 22, V=50, B=5, C=3, T=8, cfg_kw=dict(utt_type='rnn')),
    'aeed_mid': lambda: run_dialog_model('aeed', 'aeed_mid', 23, V=300, B=8, C=5, T=12,
                                         cfg_kw=dict(embed_size=32, utt_cell_size=24, ctx_cell_size=64, dec_cell_size=64,
                                                     y_size=4, k=6, max_dec_len=12)),
    # config 4 (stanford-skip.py: utt_type='rnn', two skip-thought decoders) at the mid shape
    'sted_mid': lambda: run_dialog_model('sted', 'sted_mid', 24, V=300, B=8, C=5, T=12,
                                         cfg_kw=dict(utt_type='rnn', embed_size=32, utt_cell_size=24, ctx_cell_size=64,
                                                     dec_cell_size=64, y_size=4, k=6, max_dec_len=12)),
}

if __name__ == '__main__':
    import sys
    for job in (sys.argv[1:] or list(JOBS)):        # no argument: all of them
        JOBS[job]()
