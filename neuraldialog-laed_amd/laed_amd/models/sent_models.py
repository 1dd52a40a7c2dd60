"""DI-VAE and DI-VST (mirror of laed/models/sent_models.py:20-165, 278-442) on the HIP engine.

Same constructor arguments, config attributes, state_dict keys, forward/valid_loss/backward semantics and
returned Pack keys as the reference, including DiVAE's GEN-side latent analysis helpers (sweep / enumerate /
exp_enumerate).  DiVST runs its two independent decoders on two HIP streams (LAED_VST_STREAMS=0: one stream).
One deliberate difference, documented in DESIGN.md: valid_loss sums out of place, so loss.nll read after
backward() is still the clean NLL (the reference aliases and mutates it, sent_models.py:88-90).
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import criterions, nn_lib, ops
from ..enc2dec.base_modules import BOS, EOS, PAD
from ..enc2dec.decoders import GEN, DecoderRNN
from ..enc2dec.encoders import EncoderRNN
from ..utils import LONG, Pack
from .model_bases import BaseModel


class _SentBase(BaseModel):
    """Shared constructor part of DiVAE / DiVST (sent_models.py:21-58, 279-316)."""

    def _build_common(self, corpus, config, connector_bias):
        self.vocab = corpus.vocab
        self.rev_vocab = corpus.rev_vocab
        self.vocab_size = len(self.vocab)
        self.embed_size = config.embed_size
        self.max_utt_len = config.max_utt_len
        self.go_id = self.rev_vocab[BOS]
        self.eos_id = self.rev_vocab[EOS]
        self.num_layer = config.num_layer
        self.dropout = config.dropout
        self.enc_cell_size = config.enc_cell_size
        self.dec_cell_size = config.dec_cell_size
        self.rnn_cell = config.rnn_cell
        self.max_dec_len = config.max_dec_len
        self.use_attn = config.use_attn
        self.beam_size = config.beam_size
        self.utt_type = config.utt_type
        self.bi_enc_cell = config.bi_enc_cell
        self.attn_type = config.attn_type
        self.enc_out_size = self.enc_cell_size * 2 if self.bi_enc_cell else self.enc_cell_size

        self.embedding = nn_lib.Embedding(self.vocab_size, self.embed_size, padding_idx=self.rev_vocab[PAD])
        self.x_encoder = EncoderRNN(self.embed_size, self.enc_cell_size, dropout_p=self.dropout,
                                    rnn_cell=self.rnn_cell, variable_lengths=getattr(config, 'fix_batch', False),
                                    bidirection=self.bi_enc_cell)
        self.q_y = nn_lib.Linear(self.enc_out_size, config.y_size * config.k)
        self.cat_connector = nn_lib.GumbelConnector()
        self.dec_init_connector = nn_lib.LinearConnector(config.y_size * config.k, self.dec_cell_size,
                                                         self.rnn_cell == 'lstm', has_bias=connector_bias)

    def _make_decoder(self):
        return DecoderRNN(self.vocab_size, self.max_dec_len, self.embed_size, self.dec_cell_size, self.go_id,
                          self.eos_id, n_layers=1, rnn_cell=self.rnn_cell, input_dropout_p=self.dropout,
                          dropout_p=self.dropout, use_attention=self.use_attn, attn_size=self.enc_cell_size,
                          attn_mode=self.attn_type, use_gpu=self.use_gpu, embedding=self.embedding)

    def _build_losses(self, config):
        self.nll_loss = criterions.NLLEntropy(self.rev_vocab[PAD], self.config)
        self.cat_kl_loss = criterions.CatKLLoss()
        self.entropy_loss = criterions.Entropy()
        self.bpr_loss = criterions.BatchPriorKL(config.y_size, config.k, config.use_mutual)

    def _encode_latent(self, out_utts, temperature):
        """utterance -> (log_qy [B*y,k], y_ids [B,y], dec_init_state [1,B,H])   (sent_models.py:99-119)."""
        output_embedding = self.embedding(out_utts)
        x_outs, x_last = self.x_encoder(output_embedding)
        x_last = x_last.transpose(0, 1).contiguous().view(-1, self.enc_out_size)
        qy_logits = self.q_y(x_last).view(-1, self.config.k)
        # one kernel: posterior log-softmax + Gumbel-softmax sample (+ argmax ids); noise is drawn in eval too
        sample_y, y_ids, log_qy = self.cat_connector(qy_logits, temperature, self.use_gpu, hard=not self.training,
                                                     return_max_id=True, return_log_qy=True)
        sample_y = sample_y.view(-1, self.config.k * self.config.y_size)
        y_ids = y_ids.view(-1, self.config.y_size)
        dec_init_state = self.dec_init_connector(sample_y)
        return log_qy, y_ids, dec_init_state

    def _latent_losses(self, log_qy, label_sets, log_py=None):
        """-> ((reg_kl, b_pr, mi), [global non-PAD count or None per label set]).

        Under data parallelism with 'word' averaging the step needs the global batch statistics of the posterior AND
        the global non-PAD target count of every decoder; both are known once the encoder has run, so they travel in ONE
        all-reduce message (y*k+2 sums + one count per decoder) issued here, before the decoders start -- not one
        collective for the statistics and one per decoder for the counts."""
        from .. import parallel
        if parallel.group() is not None and self.nll_loss.avg_type == 'word':
            local = torch.cat([self.nll_loss.local_count(lab) for lab in label_sets])
            reg_kl, b_pr, mi, counts = self.bpr_loss(log_qy, log_py, extra=local)
            return (reg_kl, b_pr, mi), [counts[i:i + 1] for i in range(len(label_sets))]
        return self.bpr_loss(log_qy, log_py), [None] * len(label_sets)


class DiVAE(_SentBase):
    def __init__(self, corpus, config):
        super(DiVAE, self).__init__(config)
        self._build_common(corpus, config, connector_bias=False)
        self.decoder = self._make_decoder()
        self._build_losses(config)
        self.log_py = nn.Parameter(torch.log(torch.ones(self.config.y_size, self.config.k) / config.k),
                                   requires_grad=True)
        self.kl_w = 0.0

    def valid_loss(self, loss, batch_cnt=None):
        total_loss = loss.nll
        if self.config.use_reg_kl:
            total_loss = total_loss + loss.reg_kl
        return total_loss

    def loss_terms(self, loss, batch_cnt=None):
        return [loss.nll] + ([loss.reg_kl] if self.config.use_reg_kl else [])

    def forward(self, data_feed, mode, gen_type='greedy', sample_n=1, return_latent=False):
        batch_size = len(data_feed['output_lens'])
        out_utts = self.np2var(data_feed['outputs'], LONG)
        log_qy, y_ids, dec_init_state = self._encode_latent(out_utts, 1.0)

        labels = out_utts[:, 1:]            # a strided view: the loss kernels read it in place
        dec_inputs = out_utts[:, 0:-1]
        if mode != GEN:
            # the latent losses have no random draw and no consumer on the decoder side: computing them first lets their
            # batch statistics and the decoder's target count share one exchange under data parallelism
            (reg_kl, b_pr, mi), (count,) = self._latent_losses(
                log_qy, [labels], self.log_py if getattr(self.config, 'train_prior', False) else None)
        dec_outs, dec_last, dec_ctx = self.decoder(batch_size, dec_inputs, dec_init_state, mode=mode,
                                                   gen_type=gen_type, beam_size=self.beam_size)
        if mode == GEN:
            return dec_ctx, labels.contiguous()
        nll = self.nll_loss(dec_outs, labels, global_count=count)
        self._last_dec_outs = dec_outs
        results = Pack(nll=nll, reg_kl=reg_kl, mi=mi, bpr=b_pr)
        if return_latent:
            results['log_qy'] = log_qy
            results['dec_init_state'] = dec_init_state
            results['y_ids'] = y_ids
        return results


    # ---- GEN-side latent analysis (sent_models.py:167-274): host composition over the same kernels
    def _decode_codes(self, sample_y, gen_type):
        """one-hot / mixed latent rows [n, y*k] -> free-running decode from dec_init_connector(sample_y)."""
        n = sample_y.size(0)
        dec_init_state = self.dec_init_connector(sample_y)
        _, _, dec_ctx = self.decoder(n, None, dec_init_state, mode=GEN, gen_type=gen_type, beam_size=self.beam_size)
        return dec_ctx

    def sweep(self, data_feed, gen_type='greedy'):
        """Walk from the latent code of the first utterance of the batch to that of the last one, one latent variable
        at a time, decoding every intermediate code (sent_models.py:167-218).  Returns (dec_ctx, labels, all_y_ids)."""
        batch_size = len(data_feed['output_lens'])
        out_utts = self.np2var(data_feed['outputs'], LONG)
        x_outs, x_last = self.x_encoder(self.embedding(out_utts))
        x_last = x_last.transpose(0, 1).contiguous().view(-1, self.enc_out_size)
        qy_logits = self.q_y(x_last).view(-1, self.config.k)
        sample_y, y_id = self.cat_connector(qy_logits, 1.0, self.use_gpu, hard=True, return_max_id=True)
        y_id = y_id.view(-1, self.config.y_size)
        start_y_id, end_y_id = y_id[0], y_id[batch_size - 1]
        all_y_ids = [start_y_id]
        for idx in range(self.config.y_size):
            mask = torch.zeros(self.config.y_size, dtype=torch.long, device=y_id.device)
            mask[0:idx + 1] = 1
            all_y_ids.append((1 - mask) * start_y_id + mask * end_y_id)
        num_steps = len(all_y_ids)
        all_y_ids = torch.cat(all_y_ids, dim=0).view(num_steps, -1)
        sample_y = torch.zeros((num_steps * self.config.y_size, self.config.k), device=y_id.device)
        sample_y.scatter_(1, all_y_ids.view(-1, 1), 1.0)
        dec_ctx = self._decode_codes(sample_y.view(-1, self.config.k * self.config.y_size), gen_type)
        return dec_ctx, out_utts[:, 1:].contiguous(), all_y_ids

    def enumerate(self, repeat=1, gen_type='greedy'):
        """For each latent variable, each of its k values with the other variables at the uniform mixture
        (sent_models.py:220-248; the row index y*k + k_id*repeat + r is the reference's, kept as is)."""
        y, k = self.config.y_size, self.config.k
        sample_y = torch.full((y * k * repeat, y, k), 1.0 / k, device=self.embedding.weight.device)
        for y_id in range(y):
            for k_id in range(k):
                for r_id in range(repeat):
                    idx = y_id * k + k_id * repeat + r_id
                    sample_y[idx, y_id] = 0.0
                    sample_y[idx, y_id, k_id] = 1.0
        return self._decode_codes(sample_y.view(-1, k * y), gen_type)

    def exp_enumerate(self, repeat=1, gen_type='greedy'):
        """Every one of the k**y_size joint codes (sent_models.py:250-274).  Returns (dec_ctx, all_y_ids)."""
        import itertools
        y, k = self.config.y_size, self.config.k
        all_y_ids = [list(c) for c in itertools.product(*[range(k)] * y)]
        n = int(np.power(k, y)) * repeat
        ids = self.np2var(np.array(all_y_ids), LONG)
        sample_y = torch.zeros((n * y, k), device=ids.device)
        sample_y.scatter_(1, ids.view(-1, 1), 1.0)
        return self._decode_codes(sample_y.view(-1, k * y), gen_type), all_y_ids

    def nll_loss_tok(self):
        """Per-token NLL [B,T-1] of the last teacher-forced forward (0 at PAD targets), as a host array."""
        return self._last_dec_outs.tok_nll.detach().cpu().numpy()


class DiVST(_SentBase):
    def __init__(self, corpus, config):
        super(DiVST, self).__init__(config)
        self._build_common(corpus, config, connector_bias=True)
        self.prev_decoder = self._make_decoder()
        self.next_decoder = self._make_decoder()
        self._build_losses(config)
        self.kl_w = 1.0
        self.two_streams = os.environ.get("LAED_VST_STREAMS", "1") != "0"     # decoders side by side on two HIP streams
        self._side_stream = None
        if self.two_streams and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            # the next decoder's parameters are accumulated on the default stream and produced on the side stream: intended
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)

    def valid_loss(self, loss, batch_cnt=None):
        total_loss = loss.prev_nll + loss.next_nll
        if self.config.use_reg_kl:
            total_loss = total_loss + loss.reg_kl
        return total_loss

    def loss_terms(self, loss, batch_cnt=None):
        return [loss.prev_nll, loss.next_nll] + ([loss.reg_kl] if self.config.use_reg_kl else [])

    def forward(self, data_feed, mode, gen_type='greedy', sample_n=1, return_latent=False):
        batch_size = len(data_feed['output_lens'])
        out_utts = self.np2var(data_feed['outputs'], LONG)
        prev_utts = self.np2var(data_feed['prevs'], LONG)
        next_utts = self.np2var(data_feed['nexts'], LONG)
        log_qy, y_ids, dec_init_state = self._encode_latent(out_utts, self.kl_w)

        prev_labels = prev_utts[:, 1:] if mode != GEN else prev_utts[:, 1:].contiguous()
        next_labels = next_utts[:, 1:] if mode != GEN else next_utts[:, 1:].contiguous()
        prev_count = next_count = None
        if mode != GEN:                                  # one exchange for the statistics and both target counts
            (reg_kl, b_pr, mi), (prev_count, next_count) = self._latent_losses(log_qy, [prev_labels, next_labels])
        if mode != GEN and self.two_streams and dec_init_state.is_cuda:
            # The two decoders are independent given dec_init_state: the next-utterance branch (decoder + its fused
            # projection / NLL) runs on a side HIP stream beside the previous-utterance branch.  At the script batch
            # (60) each recurrence kernel occupies half the chip or less, so the branches overlap almost entirely;
            # autograd replays every node on its forward stream, so backward overlaps the same way.
            cur = torch.cuda.current_stream(dec_init_state.device)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=dec_init_state.device)
            side = self._side_stream
            side.wait_stream(cur)
            with ops.concurrent_streams(2):          # each decoder's persistent recurrence may count on half the CUs
                with torch.cuda.stream(side):
                    for t in (dec_init_state, next_utts, next_labels) + ((next_count,) if next_count is not None else ()):
                        t.record_stream(side)
                    next_dec_outs, _, next_dec_ctx = self.next_decoder(batch_size, next_utts[:, 0:-1], dec_init_state,
                                                                       mode=mode, gen_type=gen_type, beam_size=self.beam_size)
                    next_nll = self.nll_loss(next_dec_outs, next_labels, global_count=next_count)
                    next_nll.record_stream(cur)
                prev_dec_outs, _, prev_dec_ctx = self.prev_decoder(batch_size, prev_utts[:, 0:-1], dec_init_state, mode=mode,
                                                                   gen_type=gen_type, beam_size=self.beam_size)
                prev_nll = self.nll_loss(prev_dec_outs, prev_labels, global_count=prev_count)
            cur.wait_stream(side)
        else:
            prev_dec_outs, _, prev_dec_ctx = self.prev_decoder(batch_size, prev_utts[:, 0:-1], dec_init_state, mode=mode,
                                                               gen_type=gen_type, beam_size=self.beam_size)
            next_dec_outs, _, next_dec_ctx = self.next_decoder(batch_size, next_utts[:, 0:-1], dec_init_state, mode=mode,
                                                               gen_type=gen_type, beam_size=self.beam_size)
            if mode == GEN:
                return (prev_dec_ctx, next_dec_ctx), (prev_labels, next_labels)
            prev_nll = self.nll_loss(prev_dec_outs, prev_labels, global_count=prev_count)
            next_nll = self.nll_loss(next_dec_outs, next_labels, global_count=next_count)
        results = Pack(prev_nll=prev_nll, next_nll=next_nll, reg_kl=reg_kl, mi=mi, b_pr=b_pr)
        if return_latent:
            results['log_qy'] = log_qy
            results['dec_init_state'] = dec_init_state
            results['y_ids'] = y_ids
        return results
