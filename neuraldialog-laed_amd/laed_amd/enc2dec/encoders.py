"""EncoderRNN / RnnUttEncoder with the reference's signatures (laed/enc2dec/encoders.py:9-137) on the HIP GRU."""
import torch
import torch.nn as nn

from .. import ops
from ..nn_lib import Embedding, Linear
from .base_modules import BaseRNN


class EncoderRNN(BaseRNN):
    def __init__(self, input_size, hidden_size, input_dropout_p=0, dropout_p=0, n_layers=1, rnn_cell='gru',
                 variable_lengths=False, bidirection=False):
        super(EncoderRNN, self).__init__(-1, input_size, hidden_size, input_dropout_p, dropout_p, n_layers, rnn_cell,
                                         bidirection)
        if variable_lengths:
            raise NotImplementedError("packed variable-length sequences (fix_batch=True) are outside the hot path: "
                                      "all four scripts default to fix_batch=False, where PAD positions are consumed")
        self.variable_lengths = variable_lengths
        self.output_size = hidden_size * 2 if bidirection else hidden_size

    def forward(self, input_var, input_lengths=None, init_state=None):
        """input_var [B,T,E] -> (output [B,T,H*dirs], hidden [dirs,B,H]); runs over every position incl. PAD."""
        embedded = self.input_dropout(input_var)          # p = 0 in every LAED model (encoders.py:33-36 call sites)
        if init_state is not None:
            return self.rnn(embedded, init_state)
        return self.rnn(embedded)


class RnnUttEncoder(nn.Module):
    """[B,C,T] utterance ids -> [B,C,H*dirs] utterance vectors (encoders.py:71-137); used by AE-ED / ST-ED."""

    def __init__(self, utt_cell_size, dropout, rnn_cell='gru', bidirection=True, use_attn=False, embedding=None,
                 vocab_size=None, embed_dim=None, feat_size=0):
        super(RnnUttEncoder, self).__init__()
        self.bidirection = bidirection
        self.utt_cell_size = utt_cell_size
        if embedding is None:
            self.embed_size = embed_dim
            self.embedding = Embedding(vocab_size, embed_dim)
        else:
            self.embedding = embedding
            self.embed_size = embedding.embedding_dim
        self.rnn = EncoderRNN(self.embed_size + feat_size, utt_cell_size, 0.0, dropout, rnn_cell=rnn_cell,
                              variable_lengths=False, bidirection=bidirection)
        self.multipler = 2 if bidirection else 1
        self.output_size = self.utt_cell_size * self.multipler
        self.use_attn = use_attn
        self.feat_size = feat_size
        if use_attn:
            self.key_w = Linear(self.utt_cell_size * self.multipler, self.utt_cell_size)
            self.query = Linear(self.utt_cell_size, 1)

    def forward(self, utterances, feats=None, init_state=None, return_all=False):
        batch_size, max_ctx_lens, max_utt_len = (int(s) for s in utterances.size())
        if init_state is not None:
            init_state = init_state.repeat(1, max_ctx_lens, 1)
        flat_words = utterances.reshape(-1, max_utt_len)
        words_embeded = self.embedding(flat_words)
        if feats is not None:
            flat_feats = feats.view(-1, 1).unsqueeze(1).repeat(1, max_utt_len, 1)
            words_embeded = torch.cat([words_embeded, flat_feats], dim=2)
        enc_outs, enc_last = self.rnn(words_embeded, init_state=init_state)
        if self.use_attn:
            # tanh(key_w(.)) -> query -> softmax over positions -> weighted sum (encoders.py:119-130), HIP kernels
            utt_embedded, attn = ops.attn_pool(enc_outs, self.key_w, self.query)
            attn = attn.unsqueeze(2)
        else:
            attn = None
            utt_embedded = enc_last.transpose(0, 1).contiguous().view(-1, self.output_size)
        utt_embedded = utt_embedded.view(batch_size, max_ctx_lens, self.output_size)
        if return_all:
            return utt_embedded, enc_outs, enc_last, attn
        return utt_embedded
