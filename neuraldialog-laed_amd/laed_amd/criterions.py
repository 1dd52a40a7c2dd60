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
    A dense log-prob tensor is also accepted (`laed_nll_gather_fwd/bwd`: label gather and its scatter gradient).
    Under data parallelism 'word' divides by the GLOBAL non-PAD count and 'seq'/'real_word' by the global batch.
    """

    def __init__(self, padding_idx, config, rev_vocab=None, key_vocab=None):
        super(NLLEntropy, self).__init__()
        self.padding_idx = padding_idx
        self.avg_type = config.avg_type
        if rev_vocab is not None and key_vocab is not None:
            raise NotImplementedError("keyword-weighted NLL (key_vocab) is outside the DI-VAE/DI-VST hot path")
        self.weight = None

    def local_count(self, labels_bf):
        """This rank's non-PAD target count as a 1-element float tensor: the per-rank sum a model folds into its one
        forward exchange (BatchPriorKL(..., extra=...)) so that 'word' needs no collective of its own."""
        if labels_bf.is_cuda and labels_bf.dim() == 2:
            return ops.nll_prepare(labels_bf, self.padding_idx, None, 1, want_local_count=True)[2]
        return (labels_bf != self.padding_idx).sum().float().reshape(1)

    def _row_scale(self, labels_bf, B, T, global_count=None):
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
            if global_count is not None:                                         # already all-reduced by the caller
                cnt = global_count.reshape(1)
            else:
                cnt = self.local_count(labels_bf)
                parallel.all_reduce_sum_(cnt)                                     # global non-PAD count
            return (1.0 / cnt).expand(B, T)
        raise ValueError("Unknown avg type")

    def forward(self, net_output, labels, global_count=None):
        """`global_count` (optional, beyond the reference's signature): the GLOBAL non-PAD target count as a 1-element
        device tensor when the caller has already exchanged it (data parallelism, 'word' averaging)."""
        from .enc2dec.decoders import LazyLogProbs
        B, T = labels.shape[0], labels.shape[1]
        # one launch for everything derived from the labels (weights of the averaging mode, re-ordered labels); the torch
        # composition below it remains for 'word' under data parallelism when the caller has NOT exchanged the count
        fused = labels.is_cuda and (self.avg_type != 'word' or global_count is not None or parallel.group() is None)
        if isinstance(net_output, LazyLogProbs):
            if fused:
                labels_tm, scale_tm, _ = ops.nll_prepare(labels, self.padding_idx, self.avg_type, parallel.world_size(),
                                                         global_count, time_major=True)
            else:
                scale_tm = self._row_scale(labels, B, T, global_count).t().contiguous().view(-1)
                labels_tm = labels.t().contiguous().view(-1)
            loss, tok = ops.ProjNLLFunction.apply(net_output.h_tm.reshape(B * T, -1), net_output.project.weight,
                                                  net_output.project.bias, labels_tm, scale_tm, self.padding_idx,
                                                  torch.is_grad_enabled())
            net_output.tok_nll = tok.view(T, B).t()
            return loss
        # a materialised [B,T,V] log-prob tensor: label gather + weighted sum on the HIP kernels (batch-first rows)
        if fused:
            labels_bf, scale_bf, _ = ops.nll_prepare(labels, self.padding_idx, self.avg_type, parallel.world_size(),
                                                     global_count, time_major=False)
        else:
            scale_bf = self._row_scale(labels, B, T, global_count).contiguous().view(-1)
            labels_bf = labels.reshape(-1).contiguous()
        loss, _tok = ops.NLLGatherFunction.apply(net_output.reshape(B * T, -1), labels_bf, scale_bf, self.padding_idx)
        return loss


def _rows2d(log_qy):
    if log_qy.dim() > 2:
        log_qy = log_qy.squeeze()                                  # criterions.py:101-102, 140-141
    if log_qy.dim() != 2:
        raise ValueError("expected [rows, k] log-probabilities, got shape %s" % (tuple(log_qy.shape),))
    return log_qy


def _row_average(R, batch_size, unit_average):
    if unit_average:
        return 1.0 / R                                             # torch.mean over rows
    if batch_size is None:
        raise TypeError("batch_size is required when unit_average=False")
    return 1.0 / float(batch_size)


class CatKLLoss(_Loss):
    """sum_k q (log q - log p) per row, then mean over rows or sum / batch_size (criterions.py:93-109), on the
    wavefront row-reduction kernel `laed_cat_rows_fwd/bwd`.  log_py: 1-element, [k], [1,k], [R,1] or [R,k]."""

    def forward(self, log_qy, log_py, batch_size=None, unit_average=False, dp_share=False):
        """dp_share (no counterpart in the reference): the rows are this rank's shard of a data-parallel batch and the
        result is the rank's SHARE of the global-batch value (the row weight is divided by the world size)."""
        log_qy = _rows2d(log_qy)
        w = _row_average(log_qy.shape[0], batch_size, unit_average)
        if dp_share:
            w = w / parallel.world_size()
        return ops.CatRowsFunction.apply(log_qy, log_py, w)


class Entropy(_Loss):
    """-sum_k q log q per row (criterions.py:132-147), same kernel with no prior."""

    def forward(self, log_qy, batch_size=None, unit_average=False):
        log_qy = _rows2d(log_qy)
        return ops.CatRowsFunction.apply(log_qy, None, _row_average(log_qy.shape[0], batch_size, unit_average))


class BatchPriorKL(_Loss):
    """The fused BPR / reg_kl / MI block of sent_models.py:138-156: (reg_kl, b_pr, mi) from log_qy [B*y,k]; batch
    statistics all-reduced across the DP group.  With a learned prior `log_py` [y,k] (train_prior, :144-146) reg_kl is
    KL(avg q || softmax(log_py)) through the general CatKLLoss kernel; b_pr and mi still come from the fused block."""

    def __init__(self, y_size, k, use_mutual=True):
        super(BatchPriorKL, self).__init__()
        self.y_size, self.k, self.use_mutual = y_size, k, use_mutual
        self.log_uniform = math.log(1.0 / k)
        self.cat_kl_loss = CatKLLoss()

    def forward(self, log_qy, log_py=None, extra=None):
        """-> (reg_kl, b_pr, mi), or with `extra` (per-rank sums that ride in the same all-reduce message, e.g. the
        decoders' non-PAD target counts) -> (reg_kl, b_pr, mi, extra summed over the DP group)."""
        b_pr, mi, row_kl, stats, extra_g = ops.LatentLossFunction.apply(log_qy, self.y_size, self.k, parallel.group(), extra)
        if not self.use_mutual:
            reg_kl = row_kl
        elif log_py is not None:
            avg_log_qy = ops.AvgLogQyFunction.apply(log_qy, self.y_size, self.k, parallel.group(), stats)
            reg_kl = self.cat_kl_loss(avg_log_qy, ops.LogSoftmaxKFunction.apply(log_py), None, unit_average=True)
        else:
            reg_kl = b_pr
        if extra is not None:
            return reg_kl, b_pr, mi, extra_g
        return reg_kl, b_pr, mi
