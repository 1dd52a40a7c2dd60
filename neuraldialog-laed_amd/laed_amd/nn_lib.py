"""Connectors between encoder, latent and decoder (mirror of the reference's laed/nn_lib.py, hot-path subset).

GumbelConnector / LinearConnector / GreedyConnector keep the reference signatures (nn_lib.py:138-164,228-282);
Hidden2Discrete is the north star's name for the inline `q_y` Linear + view + log_softmax of
sent_models.py:53,110-111.  Arithmetic runs in the HIP kernels (ops.py); there is no CPU path.
"""
import torch
import torch.nn as nn

from . import ops, parallel


class Linear(nn.Linear):
    """nn.Linear parameters/initialisation, forward on the fp32 MFMA GEMM."""

    def forward(self, x):
        if x.dim() == 2:
            return ops.linear(x, self.weight, self.bias)
        lead = x.shape[:-1]
        return ops.linear(x.reshape(-1, x.shape[-1]), self.weight, self.bias).view(*lead, self.out_features)


class Embedding(nn.Embedding):
    """nn.Embedding parameters (PAD row zero, no gradient into it); forward = row-gather kernel.

    ids [B,T] -> [B,T,E] returned as a batch-first VIEW of time-major memory ([T,B,E] contiguous), which is the
    layout the GRU kernels consume, so no transpose copy is made downstream.  Optional fused input dropout:
    keep [T,B,E] uint8 (time-major), drop_p.
    """

    def forward(self, ids, keep=None, drop_p=0.0, rng=None):
        """keep: injected keep mask, TIME-major [T,B,E] uint8; rng = (seed, offset): the dropout decisions are drawn
        inside the gather kernel from the counter RNG (element (b,t,e) = draw offset + (b*T+t)*E + e)."""
        pad = -1 if self.padding_idx is None else self.padding_idx
        if ids.dim() == 1:
            return ops.EmbedFunction.apply(self.weight, ids.contiguous(), keep, drop_p, pad)
        B, T = ids.shape
        if ids.stride(1) != 1:
            ids = ids.contiguous()
        # one launch: ids transposed to time-major, dropout decisions, gather (the ids need not be contiguous rows)
        out = ops.EmbedBTFunction.apply(self.weight, ids, keep, rng, drop_p, pad)
        return out.view(T, B, self.embedding_dim).transpose(0, 1)


class LinearConnector(nn.Module):
    """latent sample -> decoder initial state [1,B,H] (nn_lib.py:138-164)."""

    def __init__(self, input_size, output_size, is_lstm, has_bias=True):
        super(LinearConnector, self).__init__()
        if is_lstm:
            raise NotImplementedError("laed_amd implements the GRU path only (rnn_cell='gru', the scripts' default)")
        self.linear = Linear(input_size, output_size, bias=has_bias)
        self.is_lstm = is_lstm

    def forward(self, inputs):
        return self.linear(inputs).unsqueeze(0)

    def get_w(self):
        return self.linear.weight


class Hidden2Discrete(nn.Module):
    """h [B,input_size] -> (logits [B*y,k], log_qy [B*y,k]) = Linear + view + log_softmax (sent_models.py:53,110-111)."""

    def __init__(self, input_size, y_size, k_size, is_lstm=False, has_bias=True):
        super(Hidden2Discrete, self).__init__()
        if is_lstm:
            raise NotImplementedError("laed_amd implements the GRU path only")
        self.y_size, self.k_size = y_size, k_size
        self.p_h = Linear(input_size, y_size * k_size, bias=has_bias)

    def forward(self, inputs):
        if inputs.dim() == 3:
            inputs = inputs.squeeze(0)
        logits = self.p_h(inputs).view(-1, self.k_size)
        return logits, ops.LogSoftmaxKFunction.apply(logits)


class GumbelConnector(nn.Module):
    """Gumbel-softmax sample over each k-way latent (nn_lib.py:228-261), one fused wavefront-reduction kernel.

    The uniform noise comes from `sample_uniform(logits)`: by default a counter-based stream keyed by
    (seed, step, GLOBAL row), so a data-parallel job draws the noise the single-GPU job would; tests override
    the method on the instance to inject recorded noise (the reference's tests would override sample_gumbel).
    """

    def __init__(self, seed=1234):
        super(GumbelConnector, self).__init__()
        self.seed = seed
        self.step = 0

    def sample_uniform(self, logits):
        R, k = logits.shape
        row0 = parallel.rank() * R                       # equal shards: global row of this rank's first row
        u = ops.uniform(R * k, (self.seed * 1000003 + 2 * self.step) & 0xFFFFFFFFFFFFFFFF, row0 * k, logits.device)
        self.step += 1
        return u.view(R, k)

    def _rng_for(self, logits):
        """(seed, offset) of the draws sample_uniform(logits) would make, advancing the step the same way."""
        R, k = logits.shape
        row0 = parallel.rank() * R
        rng = ((self.seed * 1000003 + 2 * self.step) & 0xFFFFFFFFFFFFFFFF, row0 * k)
        self.step += 1
        return rng

    def forward(self, logits, temperature, use_gpu=True, hard=False, return_max_id=False, return_log_qy=False):
        if 'sample_uniform' in self.__dict__ or type(self).sample_uniform is not GumbelConnector.sample_uniform:
            u, rng = self.sample_uniform(logits), None       # injected / overridden noise source: replay its uniforms
        else:
            u, rng = None, self._rng_for(logits)             # default: drawn inside the kernel, no [R,k] tensor of uniforms
        y, log_qy, ids = ops.GumbelLatentFunction.apply(logits, u, float(temperature), bool(hard), rng)
        ret = (y,)
        if return_max_id:
            ret += (ids.view(-1, 1),)
        if return_log_qy:
            ret += (log_qy,)
        return ret if len(ret) > 1 else y


class GreedyConnector(nn.Module):
    """argmax one-hot (nn_lib.py:264-282)."""

    def forward(self, logits, use_gpu=True, return_max_id=False):
        ids = torch.max(logits, dim=1, keepdim=True)[1]
        y = torch.zeros_like(logits).scatter_(1, ids, 1.0)
        return (y, ids) if return_max_id else y
