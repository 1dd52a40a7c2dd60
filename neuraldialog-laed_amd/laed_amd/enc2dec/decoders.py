"""DecoderRNN with the reference's signature (laed/enc2dec/decoders.py:118-297), GRU / no-attention path.

TEACH_FORCE returns a LazyLogProbs handle instead of the [B,T,V] log-prob tensor: NLLEntropy recognises it and
runs projection + log-softmax + NLL (+ backward) fused; any other tensor-style use materialises the tensor, so the
public contract still holds.  GEN/greedy runs the whole free-running loop in one C call (laed_gru_decode_greedy).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops, parallel
from ..nn_lib import Embedding, Linear
from .base_modules import BaseRNN

TEACH_FORCE = "teacher_forcing"
TEACH_GEN = "teacher_gen"
GEN = "gen"


class LazyLogProbs(object):
    """log_softmax(project(h)) for h_tm [T,B,H] (time-major), not materialised until someone asks."""

    def __init__(self, h_tm, project, batch_size):
        self.h_tm = h_tm
        self.project = project
        self.B = batch_size
        self.T = h_tm.shape[0]
        self.V = project.weight.shape[0]
        self.tok_nll = None
        self._dense = None
        self._taped = torch.is_grad_enabled()     # a handle made under no_grad stays plain data, like the reference's tensor

    def materialize(self):
        """Dense [B,T,V] log-probs.  With the tape on and anything upstream requiring grad the tensor is differentiable
        (ops.DenseLogProbsFunction: a loss built on it reaches the decoder and the projection, as with the reference's
        tensor); under no_grad it is plain data.  The fused NLL path never materialises."""
        needs_tape = self._taped and torch.is_grad_enabled() and (self.h_tm.requires_grad or self.project.weight.requires_grad)
        if self._dense is not None and (self._dense.requires_grad or not needs_tape):
            return self._dense
        h_bf = self.h_tm.transpose(0, 1).reshape(self.B * self.T, -1)
        if needs_tape:
            dense = ops.DenseLogProbsFunction.apply(h_bf, self.project.weight, self.project.bias)
        else:
            with torch.no_grad():
                buf = ops.logits_rows(h_bf.detach(), self.project.weight.detach(), self.project.bias.detach())
                ops.log_softmax_rows_(buf, self.V)
                dense = buf[:, :self.V]
        self._dense = dense.view(self.B, self.T, self.V)
        return self._dense

    def expected_embedding_tm(self, emb_weight):
        """softmax(project(h)) @ emb_weight as a time-major [T,B,E] tensor, differentiable w.r.t. h, the projection and
        the embedding table, without forming the probabilities (= exp(log-probs) @ weight, dialog_models.py:421-423)."""
        T, B = self.T, self.B
        out = ops.SoftmaxEmbedFunction.apply(self.h_tm.reshape(T * B, -1), self.project.weight, self.project.bias,
                                             emb_weight)
        return out.view(T, B, -1)

    def size(self, *a):
        return torch.Size((self.B, self.T, self.V)) if not a else (self.B, self.T, self.V)[a[0]]

    @property
    def shape(self):
        return self.size()

    def __getattr__(self, name):            # .view / .exp / .topk / ... -> dense tensor semantics
        if name.startswith('__'):
            raise AttributeError(name)
        return getattr(self.materialize(), name)


class DecoderRNN(BaseRNN):
    def __init__(self, vocab_size, max_len, input_size, hidden_size, sos_id, eos_id, n_layers=1, rnn_cell='lstm',
                 input_dropout_p=0, dropout_p=0, use_attention=False, attn_mode='cat', attn_size=None, use_gpu=True,
                 embedding=None, output_size=None, tie_output_embed=False):
        super(DecoderRNN, self).__init__(vocab_size, input_size, hidden_size, input_dropout_p, dropout_p, n_layers,
                                         rnn_cell, False)
        if use_attention:
            raise NotImplementedError("decoder attention is off in every LAED script (use_attn=False) and not built")
        if tie_output_embed:
            raise NotImplementedError("tie_output_embed is never enabled by the reference's models")
        self.output_size = vocab_size if output_size is None else output_size
        self.max_length = max_len
        self.use_attention = use_attention
        self.eos_id = eos_id
        self.sos_id = sos_id
        self.init_input = None
        self.use_gpu = use_gpu
        self.embedding = Embedding(vocab_size, self.input_size) if embedding is None else embedding
        self.project = Linear(self.hidden_size, self.output_size)
        # noise control for the fused input dropout (decoders.py:156)
        self.drop_seed = 4321
        self.drop_step = 0
        self.injected_keep = None          # tests: keep mask [B,T,E] (1 keep / 0 drop), used instead of the RNG
        # noise control for gen_type='sample' (gumbel_max, base_modules.py:54-67)
        self.sample_seed = 9876
        self.sample_step = 0
        self.injected_u = None             # tests: uniforms [max_len,B,V] replayed instead of the RNG

    def _keep_mask_tm(self, B, T, device):
        """uint8 keep mask in time-major layout [T,B,E]; RNG keyed by the GLOBAL batch row."""
        E = self.input_size
        if self.injected_keep is not None:
            keep = torch.as_tensor(self.injected_keep).to(device=device)
            return (keep != 0).to(torch.uint8).view(B, T, E).transpose(0, 1).contiguous()
        row0 = parallel.rank() * B
        keep = ops.dropout_keep_mask(B * T * E, self.input_dropout_p,
                                     (self.drop_seed * 1000003 + 2 * self.drop_step + 1) & 0xFFFFFFFFFFFFFFFF,
                                     row0 * T * E, device)
        self.drop_step += 1
        return keep.view(B, T, E).transpose(0, 1).contiguous()

    def forward_step(self, input_var, hidden, encoder_outputs=None):
        """[B,T] ids, hidden [1,B,H] -> (lazy log-probs over [B,T,V], hidden [1,B,H], None)."""
        B, T = input_var.size(0), input_var.size(1)
        keep, p, rng = None, 0.0, None
        if self.training and self.input_dropout_p > 0:
            p = self.input_dropout_p
            if self.injected_keep is not None:
                keep = self._keep_mask_tm(B, T, input_var.device)        # tests: a recorded mask
            else:
                # the keep decisions are drawn inside the gather kernel (same counter RNG stream and element index as
                # _keep_mask_tm: the GLOBAL batch row keys it, so a DP job drops what the single-GPU job drops)
                rng = ((self.drop_seed * 1000003 + 2 * self.drop_step + 1) & 0xFFFFFFFFFFFFFFFF,
                       parallel.rank() * B * T * self.input_size)
                self.drop_step += 1
        embedded = self.embedding(input_var, keep=keep, drop_p=p, rng=rng)      # ids transpose + dropout + gather fused
        # hidden [1,B,H] -> [B,H] as a VIEW (hidden[0] is a select: its backward is a zero-fill plus a copy launch)
        h0 = None if hidden is None else hidden.reshape(hidden.shape[-2], hidden.shape[-1])
        out_tm, hT = self.rnn.forward_tm(embedded.transpose(0, 1), h0)
        return LazyLogProbs(out_tm, self.project, B), hT.unsqueeze(0), None

    def _beam_search(self, batch_size, init_state, beam_size):
        """gen_type='beam' (decoders.py:194-202, 222-235, 270-291), the reference's algorithm as written:
          * the initial state is repeated beam_size times per sequence (repeat_state: rows b*beam + j);
          * step 0 ranks only slot 0's V scores (all slots are identical); later steps rank cum_sum + log-probs over the
            beam*V candidates of a sequence; back pointer = candidate // V, symbol = candidate % V;
          * the chosen symbols are fed back slot by slot, but the HIDDEN STATES ARE NOT REORDERED by the back pointers
            (the reference never gathers decoder_hidden) -- reproduced, not fixed;
          * after max_length steps the best final slot is traced back through the back pointers to the 1-best sequence;
          * `length` has one entry per (sequence, slot): first step at which that slot emitted EOS.
        The search loop is pinned by tests/golden/beam.npz (the reference's loop runs under torch 2: its per-step fed
        symbols and log-probs were recorded; `ret_dict['beam_symbols']` [L,B,beam] exposes ours); the back-tracking is
        PARITY UNPINNED -- the reference's own dies there (`torch.cat` of 0-d tensors, true division in `top_id.div`) --
        and is checked against the oracle's restatement (oracle.decoder_beam).  Per step: embedding gather, GRU cell, projection and log-softmax on the HIP
        kernels; candidate ranking is torch.topk on the device (selection, GEN-side only)."""
        B, K, V, L = batch_size, int(beam_size), self.output_size, self.max_length
        with torch.no_grad():
            h = init_state[0].detach().contiguous().repeat_interleave(K, dim=0).contiguous()        # [B*K, H]
            dev = h.device
            symbols = torch.full((B * K,), int(self.sos_id), dtype=torch.int64, device=dev)
            W, bias = self.project.weight.detach(), self.project.bias.detach()
            rnn = self.rnn
            cum_sum = None
            steps, back_ptrs, hiddens = [], [], []
            for di in range(L):
                x = self.embedding(symbols.view(-1, 1)).reshape(B * K, -1).contiguous()
                h = ops.gru_cell(x, rnn.weight_ih_l0.detach(), rnn.weight_hh_l0.detach(), rnn.bias_ih_l0.detach(),
                                 rnn.bias_hh_l0.detach(), h)
                hiddens.append(h)
                buf = ops.logits_rows(h, W, bias)
                ops.log_softmax_rows_(buf, V)
                logp = buf[:, :V]
                if di == 0:
                    seq_score = logp.reshape(B, K * V)[:, :V]
                else:
                    seq_score = (cum_sum + logp).reshape(B, K * V)
                top_v, top_id = seq_score.topk(K)
                back_ptrs.append(torch.div(top_id, V, rounding_mode='floor'))        # [B, K]
                symbols = torch.remainder(top_id, V).reshape(-1)
                cum_sum = top_v.reshape(-1, 1)
                steps.append(symbols.view(B, K))
            all_sym = torch.stack(steps).cpu().numpy()                               # [L, B, K], one copy
            all_bp = torch.stack(back_ptrs).cpu().numpy()
            best = cum_sum.view(B, K).topk(1)[1].view(-1).cpu().numpy()
        lengths = np.array([L] * (B * K))
        for di in range(L):                                                          # decoders.py:241-245
            eos = all_sym[di].reshape(-1) == self.eos_id
            lengths[(lengths > di) & eos] = di + 1
        rows = np.arange(B)
        final = np.zeros((B, L), dtype=np.int64)
        for di in range(L - 1, -1, -1):                                              # back-tracking
            final[:, di] = all_sym[di][rows, best]
            best = all_bp[di][rows, best]
        seq = torch.from_numpy(final).to(dev)
        ret_dict = {DecoderRNN.KEY_SEQUENCE: [seq[:, t:t + 1] for t in range(L)], DecoderRNN.KEY_LENGTH: lengths.tolist(),
                    'beam_symbols': all_sym}
        decoder_outputs = LazyLogProbs(torch.stack(hiddens), self.project, B * K)
        return decoder_outputs, h.unsqueeze(0), ret_dict

    def forward(self, batch_size, inputs=None, init_state=None, attn_context=None, mode=TEACH_FORCE,
                gen_type='greedy', beam_size=4):
        ret_dict = dict()
        if mode == GEN:
            inputs = None
        if mode == TEACH_FORCE:
            if inputs is None:
                raise ValueError("teacher forcing needs decoder inputs")
            decoder_outputs, decoder_hidden, _ = self.forward_step(inputs, init_state, attn_context)
            ret_dict[DecoderRNN.KEY_SEQUENCE] = []
            ret_dict[DecoderRNN.KEY_LENGTH] = [self.max_length] * batch_size
            return decoder_outputs, decoder_hidden, ret_dict
        if gen_type == 'beam':
            return self._beam_search(batch_size, init_state, beam_size)
        if gen_type not in ('greedy', 'sample'):
            raise ValueError("Unsupported decoding mode")            # decoders.py:237
        with torch.no_grad():
            h0 = init_state[0].detach().contiguous()
            sample = gen_type == 'sample'
            u = None
            if sample and self.injected_u is not None:          # tests: replay recorded uniforms [max_len,B,V]
                u = torch.as_tensor(self.injected_u, dtype=torch.float32).to(h0.device).contiguous()
            ids, hbuf = ops.decode_greedy(self.embedding.weight.detach(), self.rnn.weight_ih_l0.detach(),
                                          self.rnn.weight_hh_l0.detach(), self.rnn.bias_ih_l0.detach(),
                                          self.rnn.bias_hh_l0.detach(), self.project.weight.detach(),
                                          self.project.bias.detach(), h0, self.max_length, self.sos_id,
                                          sample=sample, u=u, seed=(self.sample_seed * 1000003 + self.sample_step),
                                          row0=parallel.rank() * batch_size)
            if sample:
                self.sample_step += 1
        # ids [L,B]; EOS bookkeeping as decoders.py:241-245 (first EOS position + 1, else max_length)
        ids_bf = ids.t().contiguous()
        host = ids_bf.cpu().numpy()
        lengths = np.array([self.max_length] * batch_size)
        for b in range(batch_size):
            hit = np.nonzero(host[b] == self.eos_id)[0]
            if hit.size:
                lengths[b] = hit[0] + 1
        ret_dict[DecoderRNN.KEY_SEQUENCE] = [ids_bf[:, t:t + 1] for t in range(self.max_length)]
        ret_dict[DecoderRNN.KEY_LENGTH] = lengths.tolist()
        decoder_outputs = LazyLogProbs(hbuf[1:], self.project, batch_size)
        return decoder_outputs, hbuf[self.max_length].unsqueeze(0), ret_dict
