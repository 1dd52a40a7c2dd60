"""Minibatch data parallelism: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference has no multi-GPU code; this layer is new.  Sequences are independent, so
the batch is sharded contiguously over ranks with the weights replicated.  Two forward couplings need a tiny
exchange so that every loss term equals its single-device global-batch value:
  * BPR / MI use the batch-mean posterior inside a log  -> all-reduce of y*k+2 floats (ops.LatentLossFunction)
  * 'word' NLL divides by the GLOBAL non-PAD token count   -> all-reduce of 1 float (criterions.NLLEntropy)
Each rank's loss is then its share of the global loss, so gradients are combined with a SUM all-reduce,
launched per parameter from a post-accumulate-grad hook on a side stream: decoder.project.weight (74 % of
the bytes) is produced first in backward and its reduce overlaps the GRU backward.
"""
import os

import torch
import torch.distributed as dist

_GROUP = None
_ENABLED = False


def init(group=None):
    """Enable DP over `group` (default: WORLD).  torch.distributed must already be initialised."""
    global _GROUP, _ENABLED
    if not dist.is_initialized():
        raise RuntimeError("laed_amd.parallel.init: torch.distributed is not initialised")
    _GROUP = group if group is not None else dist.group.WORLD
    # LAED_DP_FORCE=1 keeps the DP path (hooks, side stream, collectives) on even for a 1-rank group: rehearsal aid
    _ENABLED = dist.get_world_size(_GROUP) > 1 or os.environ.get("LAED_DP_FORCE", "0") == "1"


def shutdown():
    global _GROUP, _ENABLED
    _GROUP, _ENABLED = None, False


def group():
    """Process group for the forward statistics exchanges, or None when running on one GPU."""
    return _GROUP if _ENABLED else None


def world_size():
    return dist.get_world_size(_GROUP) if _ENABLED else 1


def rank():
    return dist.get_rank(_GROUP) if _ENABLED else 0


def all_reduce_sum_(t):
    if _ENABLED:
        dist.all_reduce(t, group=_GROUP)
    return t


def shard_rows(n):
    """Contiguous [start, stop) of this rank's share of n rows (n must divide evenly)."""
    ws = world_size()
    if n % ws != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (n, ws))
    per = n // ws
    return rank() * per, (rank() + 1) * per


class GradReducer(object):
    """Overlapped gradient all-reduce (SUM) for a list of parameters.

    A hook on each parameter fires when its .grad has been accumulated in backward; the reduce is enqueued on
    a dedicated stream that waits on the compute stream's current position, so collectives for early-ready
    gradients (the vocabulary projection) run while the rest of backward computes.  `wait()` joins the streams.
    """

    def __init__(self, params, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.enabled = _ENABLED
        self.handles = []
        self.bucket_bytes = bucket_bytes
        self._pending = []
        self._pending_bytes = 0
        if not self.enabled:
            return
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)

    def _flush(self):
        if not self._pending:
            return
        pending, self._pending, self._pending_bytes = self._pending, [], 0
        if self.stream is not None:
            # each gradient was finished on the stream its hook ran on (DI-VST's two decoders run, and are replayed by
            # autograd, on two streams): wait for every producer, not only for the stream that triggered the flush
            for _, ev in pending:
                if ev is not None:
                    self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                for g, _ in pending:
                    g.record_stream(self.stream)
                    dist.all_reduce(g, group=_GROUP)
        else:                                   # CPU / gloo test path
            for g, _ in pending:
                dist.all_reduce(g, group=_GROUP)

    def _hook(self, p):
        g = p.grad
        ev = None
        if self.stream is not None and g.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        self._pending.append((g, ev))
        self._pending_bytes += g.numel() * g.element_size()
        if self._pending_bytes >= self.bucket_bytes:
            self._flush()

    def wait(self):
        """Call after backward(), before the optimizer step."""
        if not self.enabled:
            return
        self._flush()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
