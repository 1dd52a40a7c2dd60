"""AE-ED and ST-ED (mirror of laed/models/dialog_models.py:21-757) on the HIP engine.

Same constructor arguments, config attributes, state_dict keys, two-phase training (recognition side frozen and the
optimizer re-created once batch_cnt passes config.freeze_step, dialog_models.py:289-303 / :565-581 + main.py:106-110),
forward / valid_loss / model_sel_loss semantics and Pack keys as the reference.  Heavy arithmetic runs on the same
kernels as DI-VAE / DI-VST; the one new kernel is the attribute-forcing head, which needs
softmax(dec_outs) @ x_embedding.weight and gets it from laed_softmax_embed_{fwd,bwd} without forming [N,V].

Limits (raise NotImplementedError): decoder attention (use_attn), LSTM cells, fix_batch.  Data parallelism (round 3): batch
shards per rank as for DI-VAE / DI-VST; reg_kl / mi come from all-reduced batch statistics (BatchPriorKL), the 'word' NLLs
divide by the global target counts (NLLEntropy's own exchanges), and every per-row mean (pi_kl, pi_nll, attribute_nll, the
error rates) is taken over the global rows, so each rank holds its share of the global-batch value and the reducer's summed
gradients are the global-batch gradients (tests/test_y_dp_gpu.py).  valid_loss sums out of place (the reference's AeED mutates loss.nll in place,
dialog_models.py:280-284).
"""
import itertools
import os

import numpy as np
import torch

from .. import criterions, nn_lib, ops, parallel
from ..enc2dec.base_modules import BOS, EOS, PAD
from ..enc2dec.decoders import GEN, TEACH_FORCE, DecoderRNN, LazyLogProbs
from ..enc2dec.encoders import EncoderRNN, RnnUttEncoder
from ..utils import FLOAT, LONG, Pack
from .model_bases import BaseModel


def _mean_ce(log_p, ids):
    """F.cross_entropy(logits, ids) given log_p = log_softmax(logits): mean_r -log_p[r, ids[r]] (tiny [B*y,k]).
    Under data parallelism the mean runs over the GLOBAL rows (equal shards): every rank holds its share of the
    global-batch value, the shares add up to it, and the reducer's SUM of the gradients is the global-batch gradient."""
    R = log_p.shape[0] * parallel.world_size()
    scale = torch.full((log_p.shape[0],), 1.0 / R, device=log_p.device)
    return ops.NLLGatherFunction.apply(log_p, ids.reshape(-1).contiguous(), scale, -100)[0]


def _err_rate(logits, ids):
    """Share of the global-batch error rate (the local mean on one GPU); logged / model selection only."""
    err = torch.mean((torch.max(logits, dim=1)[1] != ids.reshape(-1)).float())
    return err if parallel.world_size() == 1 else err / parallel.world_size()


class LAED(BaseModel):
    """Shared pieces of AeED / StED (dialog_models.py:21-187)."""

    def _build_shared(self, corpus, config):
        self.vocab = corpus.vocab
        self.rev_vocab = corpus.rev_vocab
        self.vocab_size = len(self.vocab)
        self.go_id = self.rev_vocab[BOS]
        self.eos_id = self.rev_vocab[EOS]
        if not hasattr(config, "freeze_step"):
            config.freeze_step = 6000
        is_lstm = config.rnn_cell == 'lstm'
        # recognition side: x -> y   (x_embedding has NO padding_idx: its PAD row is an ordinary trainable row)
        self.x_embedding = nn_lib.Embedding(self.vocab_size, config.embed_size)
        self.x_encoder = EncoderRNN(config.embed_size, config.dec_cell_size, dropout_p=config.dropout,
                                    rnn_cell=config.rnn_cell, variable_lengths=False)
        self.q_y = nn_lib.Linear(config.dec_cell_size, config.y_size * config.k)
        self.x_init_connector = nn_lib.LinearConnector(config.y_size * config.k, config.dec_cell_size, is_lstm)

    def _x_decoder(self, config):
        return DecoderRNN(self.vocab_size, config.max_dec_len, config.embed_size, config.dec_cell_size, self.go_id,
                          self.eos_id, n_layers=1, rnn_cell=config.rnn_cell, input_dropout_p=config.dropout,
                          dropout_p=config.dropout, use_attention=False, use_gpu=config.use_gpu,
                          embedding=self.x_embedding)

    def _build_encdec(self, config):
        is_lstm = config.rnn_cell == 'lstm'
        self.embedding = nn_lib.Embedding(self.vocab_size, config.embed_size, padding_idx=self.rev_vocab[PAD])
        self.utt_encoder = RnnUttEncoder(config.utt_cell_size, config.dropout, use_attn=config.utt_type == 'attn_rnn',
                                         vocab_size=self.vocab_size, embedding=self.embedding)
        self.ctx_encoder = EncoderRNN(self.utt_encoder.output_size, config.ctx_cell_size, 0.0, config.dropout,
                                      config.num_layer, config.rnn_cell, variable_lengths=config.fix_batch)
        self.p_fc1 = nn_lib.Linear(config.ctx_cell_size, config.ctx_cell_size)
        self.p_y = nn_lib.Linear(config.ctx_cell_size, config.y_size * config.k)
        self.c_init_connector = nn_lib.LinearConnector(config.y_size * config.k, config.dec_cell_size, is_lstm)
        self.decoder = DecoderRNN(self.vocab_size, config.max_dec_len, config.embed_size, config.dec_cell_size,
                                  self.go_id, self.eos_id, n_layers=1, rnn_cell=config.rnn_cell,
                                  input_dropout_p=config.dropout, dropout_p=config.dropout,
                                  use_attention=config.use_attn, attn_size=config.dec_cell_size,
                                  attn_mode=config.attn_type, use_gpu=config.use_gpu, embedding=self.embedding)
        if config.use_attribute:
            self.attribute_loss = criterions.NLLEntropy(-100, config)
        self.cat_connector = nn_lib.GumbelConnector()
        self.greedy_cat_connector = nn_lib.GreedyConnector()
        self.nll_loss = criterions.NLLEntropy(self.rev_vocab[PAD], self.config)
        self.cat_kl_loss = criterions.CatKLLoss()
        self.entropy_loss = criterions.Entropy()
        self.bpr_loss = criterions.BatchPriorKL(config.y_size, config.k, True)
        self.kl_w = 0.0

    # ---- recognition network q(y|x)  (dialog_models.py:22-40)
    def qzx_forward(self, out_utts):
        output_embedding = self.x_embedding(out_utts)
        x_outs, x_last = self.x_encoder(output_embedding)
        x_last = x_last.transpose(0, 1).contiguous().view(-1, self.config.dec_cell_size)
        qy_logits = self.q_y(x_last).view(-1, self.config.k)
        if self.kl_w == 1.0 and self.config.greedy_q:
            sample_y, y_ids = self.greedy_cat_connector(qy_logits, self.use_gpu, return_max_id=True)
            log_qy = ops.LogSoftmaxKFunction.apply(qy_logits)
        else:   # posterior log-softmax comes out of the same kernel as the Gumbel-softmax sample
            sample_y, y_ids, log_qy = self.cat_connector(qy_logits, 1.0, self.use_gpu, hard=not self.training,
                                                         return_max_id=True, return_log_qy=True)
        sample_y = sample_y.view(-1, self.config.k * self.config.y_size)
        y_ids = y_ids.view(-1, self.config.y_size)
        return Pack(qy_logits=qy_logits, sample_y=sample_y, y_ids=y_ids, log_qy=log_qy)

    # ---- context encoder + prior network p(y|c)  (dialog_models.py:331-338)
    def _two_branches(self, main_fn, side_fn, *side_inputs):
        """(main_fn(), side_fn()) for two branches that share no intermediate result: the recognition / auto-encoder
        branch and the context branch of AE-ED / ST-ED (dialog_models.py:383-389 computes them one after the other).
        On the GPU the side branch runs on a second HIP stream beside the main one -- both are chains of launch-bound
        small-batch kernels -- and autograd replays each on its forward stream (LAED_BRANCH_STREAMS=0: one stream)."""
        dev = side_inputs[0].device if side_inputs else None
        if dev is None or dev.type != "cuda" or os.environ.get("LAED_BRANCH_STREAMS", "1") == "0":
            a = main_fn()
            return a, side_fn()
        cur = torch.cuda.current_stream(dev)
        if getattr(self, "_branch_stream", None) is None:
            self._branch_stream = torch.cuda.Stream(device=dev)
            if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
                torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        side = self._branch_stream
        side.wait_stream(cur)
        with ops.concurrent_streams(2):
            a = main_fn()                          # same host order as the one-stream path: same dropout / Gumbel draws
            with torch.cuda.stream(side):
                for t in side_inputs:
                    t.record_stream(side)
                b = side_fn()
        for t in (b if isinstance(b, (tuple, list)) else (b,)):
            if torch.is_tensor(t):
                t.record_stream(cur)
        cur.wait_stream(side)
        return a, b

    def _context_prior(self, ctx_utts, ctx_lens, with_fc1=True):
        c_inputs = self.utt_encoder(ctx_utts)
        c_outs, c_last = self.ctx_encoder(c_inputs, ctx_lens)
        c_last = c_last.squeeze(0)
        hid = ops.tanh(self.p_fc1(c_last)) if with_fc1 else c_last
        py_logits = self.p_y(hid).view(-1, self.config.k)
        log_py = ops.LogSoftmaxKFunction.apply(py_logits)
        return c_outs, c_last, py_logits, log_py

    def _sample_prior(self, py_logits, c_last, batch_size, sample_n, gen_type):
        """GEN-mode latent from the prior network (dialog_models.py:344-375 / :639-669)."""
        if sample_n > 1:
            if gen_type == 'greedy':
                ys, ids = [], []
                for _ in range(sample_n):
                    t_y, t_id = self.cat_connector(py_logits, 1.0, self.use_gpu, hard=True, return_max_id=True)
                    ids.append(t_id.view(-1, self.config.y_size))
                    ys.append(t_y.view(-1, self.config.k * self.config.y_size))
                sample_y, y_id = torch.cat(ys, dim=0), torch.cat(ids, dim=0)
            elif gen_type == 'sample':
                sample_y, y_id = self.greedy_cat_connector(py_logits, self.use_gpu, return_max_id=True)
                sample_y = sample_y.view(-1, self.config.k * self.config.y_size).repeat(sample_n, 1)
                y_id = y_id.view(-1, self.config.y_size).repeat(sample_n, 1)
            else:
                raise ValueError
            return sample_y, y_id, c_last.repeat(sample_n, 1), batch_size * sample_n
        sample_y, y_id = self.cat_connector(py_logits, 1.0, self.use_gpu, hard=True, return_max_id=True)
        return sample_y, y_id, c_last, batch_size

    def _attribute_head(self, dec_outs, labels, y_id):
        """Attribute forcing (dialog_models.py:417-432 / :721-741): expected embedding of the response decoder's
        output distribution -> recognition encoder -> classify the latent code the decoder was conditioned on."""
        pad_row = self.x_embedding.weight[self.rev_vocab[PAD]].view(1, 1, self.config.embed_size)
        mask_tm = torch.sign(labels).float().t().unsqueeze(2)                       # [T,B,1]
        if isinstance(dec_outs, LazyLogProbs):
            emb_tm = dec_outs.expected_embedding_tm(self.x_embedding.weight)        # [T,B,E], no [N,V] tensor
        else:
            emb_tm = torch.matmul(torch.exp(dec_outs), self.x_embedding.weight).transpose(0, 1)
        valid_tm = mask_tm * emb_tm + (1.0 - mask_tm) * pad_row
        _, x_last = self.x_encoder.rnn.forward_tm(valid_tm)
        qy_logits = self.q_y(x_last).view(-1, self.config.k)
        attribute_nll = _mean_ce(ops.LogSoftmaxKFunction.apply(qy_logits), y_id.detach())
        return attribute_nll, _err_rate(qy_logits, y_id)

    def _freeze(self, modules):
        for m in modules:
            for param in m.parameters():
                param.requires_grad = False

    def _latent_losses(self, log_qy, label_sets):
        """-> ((reg_kl, b_pr, mi), [global non-PAD count or None per label set]): under data parallelism with 'word'
        averaging the posterior's batch statistics and every decoder's target count travel in ONE all-reduce message
        (as in sent_models._SentBase._latent_losses)."""
        if parallel.group() is not None and self.nll_loss.avg_type == 'word':
            local = torch.cat([self.nll_loss.local_count(lab) for lab in label_sets])
            reg_kl, b_pr, mi, counts = self.bpr_loss(log_qy, None, extra=local)
            return (reg_kl, b_pr, mi), [counts[i:i + 1] for i in range(len(label_sets))]
        return self.bpr_loss(log_qy), [None] * len(label_sets)

    def _encdec_losses(self, dec_outs, labels, log_qy, log_py, py_logits, y_id, count=None):
        enc_dec_nll = self.nll_loss(dec_outs, labels, global_count=count)
        pi_kl = self.cat_kl_loss(log_qy.detach(), log_py, None, unit_average=True, dp_share=True)
        pi_nll = _mean_ce(log_py, y_id)
        pi_err = _err_rate(py_logits, y_id)
        if self.config.use_attribute:
            attribute_nll, adv_err = self._attribute_head(dec_outs, labels, y_id)
        else:
            attribute_nll, adv_err = None, None
        return enc_dec_nll, pi_kl, pi_nll, pi_err, attribute_nll, adv_err

    def model_sel_loss(self, loss, batch_cnt):
        if self.kl_w == 0.0:
            return self.valid_loss(loss, batch_cnt)
        return loss.pi_err + loss.adv_err

    # ---- analysis helpers (dialog_models.py:42-143)
    def greedy_forward(self, data_feed):
        ctx_lens = data_feed['context_lens']
        batch_size = len(ctx_lens)
        ctx_utts = self.np2var(data_feed['contexts'], LONG)
        out_utts = self.np2var(data_feed['outputs'], LONG)
        _, c_last, py_logits, _ = self._context_prior(ctx_utts, ctx_lens, with_fc1=False)
        sample_y, y_id = self.greedy_cat_connector(py_logits, self.use_gpu, return_max_id=True)
        sample_y = sample_y.view(-1, self.config.k * self.config.y_size)
        dec_init_state = self.c_init_connector(sample_y) + c_last.unsqueeze(0)
        dec_outs, _, _ = self.decoder(batch_size, out_utts[:, 0:-1], dec_init_state, attn_context=None,
                                      mode=TEACH_FORCE, gen_type="greedy", beam_size=self.config.beam_size)
        return Pack(nll=self.nll_loss(dec_outs, out_utts[:, 1:].contiguous()))

    def exp_forward(self, data_feed):
        """Exact marginal NLL over all k^y latent codes (dialog_models.py:42-110)."""
        ctx_lens = data_feed['context_lens']
        batch_size = len(ctx_lens)
        ctx_utts = self.np2var(data_feed['contexts'], LONG)
        out_utts = self.np2var(data_feed['outputs'], LONG)
        output_lens = self.np2var(np.asarray(data_feed['output_lens']), FLOAT)
        _, c_last, py_logits, log_py = self._context_prior(ctx_utts, ctx_lens, with_fc1=False)
        y, k = self.config.y_size, self.config.k
        labels = out_utts[:, 1:].contiguous()
        log_py = log_py.view(-1, y * k)
        nll_xcz = 0.0
        all_words = torch.sum(output_lens - 1)
        summed = criterions.NLLEntropy(self.nll_loss.padding_idx, Pack(avg_type=None))
        for combo in itertools.product(range(k), repeat=y):
            cur = torch.zeros(y, k, device=py_logits.device)
            cur[torch.arange(y), torch.tensor(combo)] = 1.0
            cur_sample_y = cur.view(1, y * k).expand(batch_size, y * k).contiguous()
            log_pyc = torch.sum(log_py * cur_sample_y, dim=1)
            dec_init_state = self.c_init_connector(cur_sample_y) + c_last.unsqueeze(0)
            dec_outs, _, _ = self.decoder(batch_size, out_utts[:, 0:-1], dec_init_state, attn_context=None,
                                          mode=TEACH_FORCE, gen_type="greedy", beam_size=self.config.beam_size)
            summed(dec_outs, labels)
            nll_xcz = nll_xcz + torch.exp(log_pyc) * dec_outs.tok_nll.sum(dim=1)
        return Pack(nll=torch.sum(nll_xcz) / all_words)

    def min_forward(self, data_feed, batch_size, sample_n):
        """sample_n prior samples per context, each decoded with gen_type='sample' (dialog_models.py:145-182)."""
        ctx_lens = data_feed['context_lens']
        ctx_utts = self.np2var(data_feed['contexts'], LONG)
        _, c_last, py_logits, log_py = self._context_prior(ctx_utts, ctx_lens)
        ys, ids = [], []
        for _ in range(sample_n):
            t_y, t_id = self.cat_connector(py_logits, 1.0, self.use_gpu, hard=True, return_max_id=True)
            ids.append(t_id.view(-1, self.config.y_size))
            ys.append(t_y.view(-1, self.config.k * self.config.y_size))
        sample_y, y_id = torch.cat(ys, dim=0), torch.cat(ids, dim=0)
        batch_size *= sample_n
        dec_init_state = self.c_init_connector(sample_y) + c_last.repeat(sample_n, 1).unsqueeze(0)
        _, _, dec_ctx = self.decoder(batch_size, None, dec_init_state, attn_context=None, mode=GEN, gen_type="sample",
                                     beam_size=self.config.beam_size)
        dec_ctx[DecoderRNN.KEY_LATENT] = y_id
        dec_ctx[DecoderRNN.KEY_POLICY] = log_py
        return dec_ctx


class AeED(LAED):
    def __init__(self, corpus, config):
        super(AeED, self).__init__(config)
        self._build_shared(corpus, config)
        self.x_decoder = self._x_decoder(config)
        self._build_encdec(config)

    def valid_loss(self, loss, batch_cnt=None):
        vae_loss = loss.vae_nll + loss.reg_kl
        enc_loss = loss.nll + (loss.pi_nll if self.config.greedy_q else loss.pi_kl)
        if self.config.use_attribute:
            enc_loss = enc_loss + 0.1 * loss.attribute_nll
        if batch_cnt is not None and batch_cnt > self.config.freeze_step:
            if self.kl_w == 0.0:
                self.kl_w = 1.0
                self.flush_valid = True
                self._freeze([self.x_embedding, self.x_encoder, self.q_y, self.x_init_connector, self.x_decoder])
            return enc_loss
        return vae_loss

    def loss_terms(self, loss, batch_cnt=None):
        """What valid_loss adds up on this step (BaseModel.backward roots the backward pass at the terms and skips the adds);
        the step that changes phase goes through valid_loss itself, which freezes the recognition side."""
        if batch_cnt is not None and batch_cnt > self.config.freeze_step:
            if self.kl_w == 0.0:
                return None
            terms = [loss.nll, loss.pi_nll if self.config.greedy_q else loss.pi_kl]
            return terms + ([(loss.attribute_nll, 0.1)] if self.config.use_attribute else [])
        return [loss.vae_nll, loss.reg_kl]

    def pxz_forward(self, batch_size, results, out_utts, mode, gen_type):
        dec_init_state = self.x_init_connector(results.sample_y)
        dec_outs, dec_last, dec_ctx = self.x_decoder(batch_size, out_utts[:, 0:-1], dec_init_state, mode=mode,
                                                     gen_type=gen_type, beam_size=self.config.beam_size)
        results['dec_outs'] = dec_outs
        results['dec_ctx'] = dec_ctx
        return results

    def forward(self, data_feed, mode, sample_n=1, gen_type='greedy', return_latent=False):
        ctx_lens = data_feed['context_lens']
        batch_size = len(ctx_lens)
        ctx_utts = self.np2var(data_feed['contexts'], LONG)
        out_utts = self.np2var(data_feed['outputs'], LONG)

        vae_resp, (c_outs, c_last, py_logits, log_py) = self._two_branches(
            lambda: self.pxz_forward(batch_size, self.qzx_forward(out_utts[:, 1:]), out_utts, mode, gen_type),
            lambda: self._context_prior(ctx_utts, ctx_lens), ctx_utts)

        qy_id = vae_resp.y_ids.detach().view(-1, self.config.y_size)
        if mode != GEN:
            sample_y, y_id = vae_resp.sample_y.detach(), qy_id
        else:
            sample_y, y_id, c_last, batch_size = self._sample_prior(py_logits, c_last, batch_size, sample_n, gen_type)

        sample_y = sample_y.view(-1, self.config.k * self.config.y_size)
        dec_init_state = self.c_init_connector(sample_y) + c_last.unsqueeze(0)
        dec_outs, dec_last, dec_ctx = self.decoder(batch_size, out_utts[:, 0:-1], dec_init_state,
                                                   attn_context=c_outs if self.config.use_attn else None,
                                                   mode=mode, gen_type=gen_type, beam_size=self.config.beam_size)
        labels = out_utts[:, 1:].contiguous()
        dec_ctx[DecoderRNN.KEY_RECOG_LATENT] = qy_id
        dec_ctx[DecoderRNN.KEY_LATENT] = y_id
        dec_ctx[DecoderRNN.KEY_POLICY] = log_py
        if mode == GEN:
            return dec_ctx, labels

        log_qy = vae_resp.log_qy
        (reg_kl, _, mi), (count,) = self._latent_losses(log_qy, [labels])      # both NLLs score the same targets
        vae_nll = self.nll_loss(vae_resp.dec_outs, labels, global_count=count)
        enc_dec_nll, pi_kl, pi_nll, pi_err, attribute_nll, adv_err = self._encdec_losses(
            dec_outs, labels, log_qy, log_py, py_logits, y_id, count)
        results = Pack(nll=enc_dec_nll, pi_kl=pi_kl, attribute_nll=attribute_nll, vae_nll=vae_nll, reg_kl=reg_kl,
                       mi=mi, pi_nll=pi_nll, pi_err=pi_err, adv_err=adv_err)
        if return_latent:
            results['log_py'] = log_py
            results['log_qy'] = log_qy
            results['dec_init_state'] = dec_init_state
            results['y_ids'] = y_id
        return results


class StED(LAED):
    def __init__(self, corpus, config):
        super(StED, self).__init__(config)
        self._build_shared(corpus, config)
        self.prev_decoder = self._x_decoder(config)
        self.next_decoder = self._x_decoder(config)
        self._build_encdec(config)

    def valid_loss(self, loss, batch_cnt=None):
        vst_loss = loss.vst_prev_nll + loss.vst_next_nll
        if self.config.use_reg_kl:
            vst_loss = vst_loss + loss.reg_kl
        enc_loss = loss.nll + (loss.pi_nll if self.config.greedy_q else loss.pi_kl)
        if self.config.use_attribute:
            enc_loss = enc_loss + loss.attribute_nll
        if batch_cnt is not None and batch_cnt > self.config.freeze_step:
            if self.kl_w == 0.0:
                self.kl_w = 1.0
                self.flush_valid = True
                self._freeze([self.x_embedding, self.x_encoder, self.q_y, self.x_init_connector, self.prev_decoder,
                              self.next_decoder])
            return enc_loss
        return vst_loss

    def loss_terms(self, loss, batch_cnt=None):
        """See AeED.loss_terms."""
        if batch_cnt is not None and batch_cnt > self.config.freeze_step:
            if self.kl_w == 0.0:
                return None
            terms = [loss.nll, loss.pi_nll if self.config.greedy_q else loss.pi_kl]
            return terms + ([loss.attribute_nll] if self.config.use_attribute else [])
        return [loss.vst_prev_nll, loss.vst_next_nll] + ([loss.reg_kl] if self.config.use_reg_kl else [])

    def pxz_forward(self, batch_size, results, prev_utts, next_utts, mode, gen_type):
        dec_init_state = self.x_init_connector(results.sample_y)
        prev_outs, _, prev_ctx = self.prev_decoder(batch_size, prev_utts[:, 0:-1], dec_init_state, mode=mode,
                                                   gen_type=gen_type, beam_size=self.config.beam_size)
        next_outs, _, next_ctx = self.next_decoder(batch_size, next_utts[:, 0:-1], dec_init_state, mode=mode,
                                                   gen_type=gen_type, beam_size=self.config.beam_size)
        results['prev_outs'], results['prev_ctx'] = prev_outs, prev_ctx
        results['next_outs'], results['next_ctx'] = next_outs, next_ctx
        return results

    def forward(self, data_feed, mode, sample_n=1, gen_type='greedy', return_latent=False):
        ctx_lens = data_feed['context_lens']
        batch_size = len(ctx_lens)
        ctx_utts = self.np2var(data_feed['contexts'], LONG)
        out_utts = self.np2var(data_feed['outputs'], LONG)
        prev_utts = self.np2var(data_feed['prevs'], LONG)
        next_utts = self.np2var(data_feed['nexts'], LONG)

        vst_resp, (c_outs, c_last, py_logits, log_py) = self._two_branches(
            lambda: self.pxz_forward(batch_size, self.qzx_forward(out_utts[:, 1:]), prev_utts, next_utts, mode, gen_type),
            lambda: self._context_prior(ctx_utts, ctx_lens), ctx_utts)

        qy_id = vst_resp.y_ids.detach().view(-1, self.config.y_size)
        if mode != GEN:
            sample_y, y_id = vst_resp.sample_y.detach(), qy_id
        else:
            sample_y, y_id, c_last, batch_size = self._sample_prior(py_logits, c_last, batch_size, sample_n, gen_type)

        sample_y = sample_y.view(-1, self.config.k * self.config.y_size)
        dec_init_state = self.c_init_connector(sample_y) + c_last.unsqueeze(0)
        dec_outs, dec_last, dec_ctx = self.decoder(batch_size, out_utts[:, 0:-1], dec_init_state, attn_context=None,
                                                   mode=mode, gen_type=gen_type, beam_size=self.config.beam_size)
        labels = out_utts[:, 1:].contiguous()
        dec_ctx[DecoderRNN.KEY_LATENT] = y_id
        dec_ctx[DecoderRNN.KEY_POLICY] = log_py
        dec_ctx[DecoderRNN.KEY_RECOG_LATENT] = qy_id
        if mode == GEN:
            return dec_ctx, labels

        log_qy = vst_resp.log_qy
        prev_labels, next_labels = prev_utts[:, 1:].contiguous(), next_utts[:, 1:].contiguous()
        (reg_kl, _, mi), (count, prev_count, next_count) = self._latent_losses(log_qy, [labels, prev_labels, next_labels])
        vst_prev_nll = self.nll_loss(vst_resp.prev_outs, prev_labels, global_count=prev_count)
        vst_next_nll = self.nll_loss(vst_resp.next_outs, next_labels, global_count=next_count)
        enc_dec_nll, pi_kl, pi_nll, pi_err, attribute_nll, adv_err = self._encdec_losses(
            dec_outs, labels, log_qy, log_py, py_logits, y_id, count)
        results = Pack(nll=enc_dec_nll, pi_kl=pi_kl, pi_nll=pi_nll, attribute_nll=attribute_nll,
                       vst_prev_nll=vst_prev_nll, vst_next_nll=vst_next_nll, reg_kl=reg_kl, mi=mi, pi_err=pi_err,
                       adv_err=adv_err)
        if return_latent:
            results['log_py'] = log_py
            results['log_qy'] = log_qy
            results['dec_init_state'] = dec_init_state
            results['y_ids'] = y_id
        return results
