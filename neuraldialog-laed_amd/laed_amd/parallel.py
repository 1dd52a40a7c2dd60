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
    _GLOO.clear()
    COLLECTIVES["forward"] = COLLECTIVES["grad"] = COLLECTIVES["report"] = 0


def group():
    """Process group for the forward statistics exchanges, or None when running on one GPU."""
    return _GROUP if _ENABLED else None


def world_size():
    return dist.get_world_size(_GROUP) if _ENABLED else 1


def rank():
    return dist.get_rank(_GROUP) if _ENABLED else 0


_GLOO = {}
# collectives issued since init(): "forward" = loss-statistics exchanges (ops.LatentLossFunction and friends, NLLEntropy),
# "grad" = the GradReducer's, "report" = the engine's loss reports (global_terms).  Tests assert one forward exchange per
# DI-VAE / DI-VST step.
COLLECTIVES = {"forward": 0, "grad": 0, "report": 0}


def _is_gloo(group):
    if group not in _GLOO:
        _GLOO[group] = str(dist.get_backend(group)).lower() == "gloo"
    return _GLOO[group]


def all_reduce_(t, group=None, kind="forward"):
    """In-place SUM all-reduce of `t` over `group` (default: the DP group) on the current stream.

    Backend "nccl" (RCCL) is stream-ordered and needs nothing else.  Backend "gloo" is only the transport of the tests
    and of the rehearsal modes (several ranks sharing one GPU): torch's gloo path for CUDA tensors can fail to complete
    on this ROCm stack when another process shares the GPU -- always when the collective is issued while the stream still
    has kernels in flight (tools/exp/gloo_shared_gpu.py reproduces it with plain torch matmuls; round 1's "4-rank hang"),
    and, seen once in round 3 with both ranks inside the same all_reduce of a 1-element tensor, now and then even after the
    stream was drained.  A CUDA tensor therefore never reaches gloo: it is staged through a host copy (which also drains
    the current stream), reduced there, and copied back."""
    group = _GROUP if group is None else group
    COLLECTIVES[kind] += 1
    if t.is_cuda and _is_gloo(group):
        host = t.detach().cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
        return t
    dist.all_reduce(t, group=group)
    return t


def all_reduce_sum_(t):
    if _ENABLED:
        all_reduce_(t)
    return t


def shard_rows(n):
    """Contiguous [start, stop) of this rank's share of n rows (n must divide evenly)."""
    ws = world_size()
    if n % ws != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (n, ws))
    per = n // ws
    return rank() * per, (rank() + 1) * per


def shard_batch(batch):
    """This rank's contiguous rows of a loader batch (every entry of a loader's Pack is per-row: id matrices padded to the
    GLOBAL batch's widths, length vectors, metas).  Identity on one GPU."""
    if not _ENABLED or batch is None or not hasattr(batch.get("outputs"), "__len__"):
        return batch
    n = len(batch["outputs"])
    b0, b1 = shard_rows(n)
    out = type(batch)()
    for key, val in batch.items():
        if val is not None and hasattr(val, "__len__") and len(val) == n:
            out[key] = val[b0:b1]
        else:
            out[key] = val
    return out


def global_terms(loss, replicated=()):
    """The loss terms of one step as GLOBAL-batch values on every rank: under data parallelism a model's per-row terms (NLLs,
    per-row KLs, error rates) are this rank's SHARE of the global-batch value and the batch statistics (`replicated`: reg_kl,
    mi, bpr) are already global.  One small all-reduce carries every share.  Reporting / model selection only -- the training
    step itself never needs it.  Identity on one GPU."""
    if not _ENABLED:
        return loss
    keys = [k for k, v in loss.items() if torch.is_tensor(v) and v.numel() == 1 and v.is_floating_point() and k not in replicated]
    if not keys:
        return loss
    vec = torch.stack([loss[k].detach().reshape(()).float() for k in keys])
    all_reduce_(vec, kind="report")
    out = type(loss)(loss)
    for i, k in enumerate(keys):
        out[k] = vec[i]
    return out


class GradReducer(object):
    """Overlapped, bucketed gradient all-reduce (SUM) for a list of parameters.

    A hook on each parameter fires when its .grad has been accumulated in backward.  Gradients are reduced in the
    order backward produces them (decoder.project.weight -- 74 % of the bytes at the scaled shape -- comes first and
    its reduce overlaps the decoder / encoder GRU backward):
      * a gradient of at least `big_bytes` is all-reduced in place, on its own (no copy);
      * smaller ones (biases, q_y, the connector, ...) are coalesced: copied into one flat fp32 buffer, reduced with ONE
        collective once `bucket_bytes` are pending (or at wait()), and the parameters' .grad are then re-pointed at their
        slices of the reduced buffer (no copy back).
    Collectives are enqueued on a dedicated stream that waits for each producer stream, so they run beside the rest of
    backward; `wait()` joins the streams.  `close()` removes the hooks (a model that builds a second optimizer must
    not leave the first reducer's hooks live: every gradient would be summed twice).
    """

    def __init__(self, params, bucket_bytes=32 << 20, big_bytes=8 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.enabled = _ENABLED
        self.bucket_bytes = bucket_bytes
        self.big_bytes = big_bytes
        self._pending = []                      # (param, producer event) of small gradients not yet flushed
        self._flats = []                        # coalesced buffers made since the last wait()
        self._pending_bytes = 0
        self._hooks = []
        self.collectives = 0                    # issued since construction (tests / rehearsal bookkeeping)
        self.stream = None
        if not self.enabled:
            return
        self.stream = torch.cuda.Stream() if (torch.cuda.is_available() and any(p.is_cuda for p in self.params)) else None
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._hook))

    def close(self):
        """Flush what is pending, join the side stream and remove the hooks."""
        if self.enabled:
            self.wait()
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.enabled = False

    def _reduce(self, t, events):
        self.collectives += 1
        if self.stream is not None and t.is_cuda:
            # each gradient was finished on the stream its hook ran on (DI-VST's two decoders run, and are replayed by
            # autograd, on two streams): wait for every producer, not only for the stream that triggered the flush
            for ev in events:
                if ev is not None:
                    self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                t.record_stream(self.stream)
                all_reduce_(t, kind="grad")
        else:                                   # CPU / gloo test path
            all_reduce_(t, kind="grad")

    def _flush(self):
        if not self._pending:
            return
        pending, self._pending, self._pending_bytes = self._pending, [], 0
        if len(pending) == 1:
            p, ev = pending[0]
            self._reduce(p.grad, [ev])
            return
        g0 = pending[0][0].grad
        n = sum(p.grad.numel() for p, _ in pending)
        side = self.stream is not None and g0.is_cuda
        if side:
            for _, ev in pending:
                if ev is not None:
                    self.stream.wait_event(ev)
        ctx = torch.cuda.stream(self.stream) if side else _Null()
        with ctx:
            flat = torch.empty(n, device=g0.device, dtype=g0.dtype)
            off = 0
            views = []
            for p, _ in pending:
                g = p.grad
                k = g.numel()
                v = flat[off:off + k].view(g.shape)
                v.copy_(g)
                if side:
                    g.record_stream(self.stream)
                views.append((p, v))
                off += k
            self.collectives += 1
            all_reduce_(flat, kind="grad")
        if side:
            # `flat` belongs to the reducer stream's pool; its readers are the optimizer (and whoever reads .grad) on the
            # stream that calls wait() -- not necessarily the stream this hook fired on (DI-VST's side stream, a branch
            # stream).  wait() records that consumer stream on every buffer made since the last wait, so the caching
            # allocator cannot hand the block back to the reducer stream while an Adam launch still reads it.
            self._flats.append(flat)
        for p, v in views:
            p.grad = v                          # consumers (the optimizer) join the side stream in wait() first

    def _hook(self, p):
        g = p.grad
        if g is None:
            # frozen between forward and backward (AE-ED / ST-ED freeze the recognition side inside valid_loss, just before
            # the flip step's backward): autograd dropped the gradient, on every rank alike -- nothing to reduce
            return
        ev = None
        if self.stream is not None and g.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        nbytes = g.numel() * g.element_size()
        if nbytes >= self.big_bytes or g.dtype != torch.float32 or not g.is_contiguous():
            self._reduce(g, [ev])
            return
        self._pending.append((p, ev))
        self._pending_bytes += nbytes
        if self._pending_bytes >= self.bucket_bytes:
            self._flush()

    def wait(self):
        """Call after backward(), before the optimizer step."""
        if not self.enabled:
            return
        self._flush()
        if self.stream is not None:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self.stream)
            for flat in self._flats:
                flat.record_stream(cur)
        self._flats = []


class _Null(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
