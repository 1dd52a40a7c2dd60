"""BaseRNN and the GRU module (mirror of laed/enc2dec/base_modules.py:11-74, GRU path).

`GRU` carries nn.GRU's parameter names, shapes, gate order (r,z,n) and initialisation (U(-1/sqrt(H), 1/sqrt(H)) in
nn.GRU's parameter order, so the same torch.manual_seed gives the same weights and reference checkpoints load),
but its forward/backward run on the HIP kernels (laed_gru_fwd / laed_gru_bwd), time-major inside.
"""
import math

import torch
import torch.nn as nn

import os

from .. import ops

_BIDIR_STREAMS = os.environ.get("LAED_BIDIR_STREAMS", "1") != "0"     # the two directions of a bidirectional GRU side by side

PAD = '<pad>'      # laed/dataset/corpora.py:14-17
UNK = '<unk>'
BOS = '<s>'
EOS = '</s>'


class GRU(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers=1, batch_first=True, dropout=0.0, bidirectional=False):
        super(GRU, self).__init__()
        if num_layers != 1:
            raise NotImplementedError("laed_amd.GRU: num_layers must be 1 (every LAED script uses 1)")
        if not batch_first:
            raise NotImplementedError("laed_amd.GRU: the reference builds batch_first=True RNNs only")
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.batch_first, self.dropout, self.bidirectional = 1, True, dropout, bidirectional
        for sfx in ([''] + (['_reverse'] if bidirectional else [])):
            self.register_parameter('weight_ih_l0' + sfx, nn.Parameter(torch.empty(3 * hidden_size, input_size)))
            self.register_parameter('weight_hh_l0' + sfx, nn.Parameter(torch.empty(3 * hidden_size, hidden_size)))
            self.register_parameter('bias_ih_l0' + sfx, nn.Parameter(torch.empty(3 * hidden_size)))
            self.register_parameter('bias_hh_l0' + sfx, nn.Parameter(torch.empty(3 * hidden_size)))
        stdv = 1.0 / math.sqrt(hidden_size)
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)
        self._side_stream = None
        if bidirectional and _BIDIR_STREAMS and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            # the reverse direction's parameters are produced on the side stream and accumulated on the default one: intended
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)

    def _params(self, sfx):
        return [getattr(self, n + sfx) for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]

    def forward_tm(self, x_tm, h0=None, reverse=False):
        """Time-major entry: x_tm [T,B,E], h0 [B,H] -> (out_tm [T,B,H], hT [B,H])."""
        if reverse:
            out, hT = ops.GRUFunction.apply(x_tm.flip(0), *self._params('_reverse'), h0, torch.is_grad_enabled())
            return out.flip(0), hT
        return ops.GRUFunction.apply(x_tm, *self._params(''), h0, torch.is_grad_enabled())

    def forward(self, x, hx=None):
        """nn.GRU(batch_first=True) contract: x [B,T,E], hx [dirs,B,H] -> (out [B,T,H*dirs], h_n [dirs,B,H]).
        `out` is a batch-first view of time-major memory; x given as such a view is consumed without a copy."""
        x_tm = x.transpose(0, 1)                  # [T,B,E]; contiguous already when x came from laed_amd.Embedding
        if not self.bidirectional:
            out_f, h_f = self.forward_tm(x_tm, None if hx is None else hx[0])
            return out_f.transpose(0, 1), h_f.unsqueeze(0)
        if x.is_cuda and _BIDIR_STREAMS:
            # the two directions are independent recurrences of launch-bound step kernels (the utterance encoder of
            # AE-ED / ST-ED: 300 sequences x 40 steps each way): the reverse one runs on a side HIP stream beside the
            # forward one; autograd replays every node on its forward stream, so backward overlaps the same way
            cur = torch.cuda.current_stream(x.device)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=x.device)
            side = self._side_stream
            side.wait_stream(cur)
            with ops.concurrent_streams(2):          # a persistent recurrence may only count on half the CUs
                with torch.cuda.stream(side):
                    out_b, h_b = self.forward_tm(x_tm, None if hx is None else hx[1], reverse=True)
                out_f, h_f = self.forward_tm(x_tm, None if hx is None else hx[0])
            x_tm.record_stream(side)
            for t in (out_b, h_b):
                t.record_stream(cur)
            cur.wait_stream(side)
        else:
            out_f, h_f = self.forward_tm(x_tm, None if hx is None else hx[0])
            out_b, h_b = self.forward_tm(x_tm, None if hx is None else hx[1], reverse=True)
        return torch.cat([out_f, out_b], dim=2).transpose(0, 1), torch.stack([h_f, h_b], dim=0)


class BaseRNN(nn.Module):
    SYM_MASK = PAD
    SYM_EOS = EOS

    KEY_ATTN_SCORE = 'attention_score'
    KEY_LENGTH = 'length'
    KEY_SEQUENCE = 'sequence'
    KEY_LATENT = 'latent'
    KEY_RECOG_LATENT = 'recog_latent'
    KEY_POLICY = "policy"
    KEY_G = 'g'
    KEY_PTR_SOFTMAX = 'ptr_softmax'
    KEY_PTR_CTX = "ptr_context"

    def __init__(self, vocab_size, input_size, hidden_size, input_dropout_p, dropout_p, n_layers, rnn_cell,
                 bidirectional):
        super(BaseRNN, self).__init__()
        self.vocab_size = vocab_size
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.n_layers = n_layers
        self.input_dropout_p = input_dropout_p
        self.input_dropout = nn.Dropout(p=input_dropout_p)
        if rnn_cell.lower() == 'gru':
            self.rnn_cell = GRU
        elif rnn_cell.lower() == 'lstm':
            raise NotImplementedError("laed_amd implements the GRU cell only (the default of all four scripts)")
        else:
            raise ValueError("Unsupported RNN Cell: {0}".format(rnn_cell))
        self.dropout_p = dropout_p
        # nn.GRU(dropout=p, num_layers=1) applies no dropout (base_modules.py:43-45; torch warns) -> same here
        self.rnn = self.rnn_cell(input_size, hidden_size, n_layers, batch_first=True, dropout=dropout_p,
                                 bidirectional=bidirectional)

    def repeat_state(self, state, batch_size, times):
        new_s = state.repeat(1, 1, times)
        return new_s.view(-1, batch_size * times, self.hidden_size)

    def forward(self, *args, **kwargs):
        raise NotImplementedError()
