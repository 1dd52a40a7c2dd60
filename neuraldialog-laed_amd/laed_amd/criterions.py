"""Loss modules with the reference's signatures (laed/criterions.py), hot-path subset:
NLLEntropy (all four avg_type modes), CatKLLoss, Entropy, plus the fused BPR/MI block used by the models."""
import math

import torch
from torch.nn.modules.loss import _Loss

from . import ops, parallel


class NLLEntropy(_Loss):
    """Padding-masked NLL (criterions.py:25-72).

    `net_output` is normally the lazy log-prob handle returned by DecoderRNN (teacher forcing): the vocabulary
    projection, log-softmax, NLL and their backward then run fused, and the [B,T,V] tensor is never formed.
    A dense log-prob tensor is also accepted (label gather only; averaging on the same kernels).
    Under data parallelism 'word' divides by the GLOBAL non-PAD count and 'seq'/'real_word' by the global batch.
    """

    def __init__(self, padding_idx, config, rev_vocab=None, key_vocab=None):
        super(NLLEntropy, self).__init__()
        self.padding_idx = padding_idx
        self.avg_type = config.avg_type
        if rev_vocab is not None and key_vocab is not None:
            raise NotImplementedError("keyword-weighted NLL (key_vocab) is outside the DI-VAE/DI-VST hot path")
        self.weight = None

    def _row_scale(self, labels_bf, B, T):
        """Per-row loss weights [B,T] (batch-first) for the configured averaging mode."""
        dev = labels_bf.device
        ws = parallel.world_size()
        if self.avg_type is None:
            return torch.ones((B, T), device=dev)
        if self.avg_type == 'seq':
            return torch.full((B, T), 1.0 / (B * ws), device=dev)
        if self.avg_type == 'real_word':
            word_cnt = torch.sign(labels_bf).sum(dim=1, keepdim=True).float()     # criterions.py:62
            return (1.0 / (word_cnt * (B * ws))).expand(B, T)
        if self.avg_type == 'word':
            cnt = (labels_bf != self.padding_idx).sum().float().reshape(1)
            parallel.all_reduce_sum_(cnt)                                         # global non-PAD count
            return (1.0 / cnt).expand(B, T)
        raise ValueError("Unknown avg type")

    def forward(self, net_output, labels):
        from .enc2dec.decoders import LazyLogProbs
        B, T = labels.shape[0], labels.shape[1]
        scale_tm = self._row_scale(labels, B, T).t().contiguous().view(-1)
        labels_tm = labels.t().contiguous().view(-1)
        if isinstance(net_output, LazyLogProbs):
            loss, tok = ops.ProjNLLFunction.apply(net_output.h_tm.reshape(B * T, -1), net_output.project.weight,
                                                  net_output.project.bias, labels_tm, scale_tm, self.padding_idx)
            net_output.tok_nll = tok.view(T, B).t()
            return loss
        logp = net_output.reshape(B * T, -1)
        tok = -logp.gather(1, labels.reshape(-1, 1)).squeeze(1) * (labels.reshape(-1) != self.padding_idx).float()
        return torch.sum(tok * scale_tm.view(T, B).t().reshape(-1))


class CatKLLoss(_Loss):
    """sum_k q (log q - log p) per row (criterions.py:93-109); tiny, kept on stock tensor ops."""

    def forward(self, log_qy, log_py, batch_size=None, unit_average=False):
        if log_qy.dim() > 2:
            log_qy = log_qy.squeeze()
        qy = torch.exp(log_qy)
        y_kl = torch.sum(qy * (log_qy - log_py), dim=1)
        return torch.mean(y_kl) if unit_average else torch.sum(y_kl) / batch_size


class Entropy(_Loss):
    """-sum_k q log q per row (criterions.py:132-147)."""

    def forward(self, log_qy, batch_size=None, unit_average=False):
        if log_qy.dim() > 2:
            log_qy = log_qy.squeeze()
        qy = torch.exp(log_qy)
        h_q = torch.sum(-1 * log_qy * qy, dim=1)
        return torch.mean(h_q) if unit_average else torch.sum(h_q) / batch_size


class BatchPriorKL(_Loss):
    """The fused BPR / reg_kl / MI block of sent_models.py:138-156 (train_prior=False):
    (reg_kl, b_pr, mi) from log_qy [B*y,k]; batch statistics all-reduced across the DP group."""

    def __init__(self, y_size, k, use_mutual=True):
        super(BatchPriorKL, self).__init__()
        self.y_size, self.k, self.use_mutual = y_size, k, use_mutual
        self.log_uniform = math.log(1.0 / k)

    def forward(self, log_qy):
        b_pr, mi, row_kl = ops.LatentLossFunction.apply(log_qy, self.y_size, self.k, parallel.group())
        reg_kl = b_pr if self.use_mutual else row_kl
        return reg_kl, b_pr, mi.detach()
