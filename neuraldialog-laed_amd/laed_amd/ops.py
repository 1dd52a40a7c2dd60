"""Host-side operators over the C ABI: thin checked wrappers + torch.autograd.Function glue.

Everything here runs on the HIP kernels of liblaed_hip.so; torch is used for device memory, streams and
the autograd tape only.  Sequence tensors are time-major ([T,B,*]) at this level.
"""
import os

import torch

from . import _lib
from ._lib import check, lib, ptr, stream

I64 = torch.int64
U8 = torch.uint8


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------------
# raw wrappers
# ----------------------------------------------------------------------------------------------
def sgemm(A, B, transA=False, transB=False, bias=None, out=None, accumulate=False, colsum_out=None,
          colsum_accumulate=False, splitk_ws=True, ws_floats=0):
    """C = op(A) op(B) (+bias) (+C).  A, B 2-D fp32 (row stride arbitrary, unit column stride)."""
    assert A.dim() == 2 and B.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1
    M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    N = B.shape[0] if transB else B.shape[1]
    Kb = B.shape[1] if transB else B.shape[0]
    if K != Kb:
        raise _lib.LaedError("sgemm: inner dimensions differ (%d vs %d)" % (K, Kb))
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    assert out.shape == (M, N) and out.stride(1) == 1
    import ctypes
    vp = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    for t in (A, B, out, bias, colsum_out):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32):
            raise _lib.LaedError("sgemm: operands must be fp32 CUDA tensors")
    ws, ws_n = None, 0
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    if splitk_ws and t128 < 1024 and K >= 512:              # under two rounds of output tiles: the library may split K
        ws_n = (8 if M * N <= (1 << 22) else 4) * (M * N + M)
        ws = workspace("sgemm_ws", ws_n, A.device)
    elif ws_floats > 0:                                     # caller-sized scratch (tail split of many-tile problems)
        ws_n = int(ws_floats)
        ws = workspace("sgemm_ws", ws_n, A.device)
    check(lib.laed_sgemm_ws(int(transA), int(transB), M, N, K, vp(A), A.stride(0) if A.shape[0] > 1 else max(A.shape[1], 1),
                            vp(B), B.stride(0) if B.shape[0] > 1 else max(B.shape[1], 1),
                            vp(out), out.stride(0) if out.shape[0] > 1 else max(N, 1), vp(bias), int(accumulate),
                            vp(colsum_out), int(colsum_accumulate), vp(ws), ws_n, stream()), "laed_sgemm_ws")
    return out


def set_gemm_mode(mode):
    """'fp32' (default; exact fp32 MFMA) or 'bf16x3' (split-bf16 products, fp32 accumulate) for x W^T GEMMs."""
    m = {"fp32": 0, "bf16x3": 1, 0: 0, 1: 1}[mode]
    check(lib.laed_set_gemm_mode(m), "laed_set_gemm_mode")


def get_gemm_mode():
    return "bf16x3" if lib.laed_get_gemm_mode() == 1 else "fp32"


def uniform(n, seed, offset, device):
    u = torch.empty(n, device=device, dtype=torch.float32)
    check(lib.laed_uniform(ptr(u), n, seed, offset, stream()), "laed_uniform")
    return u


def dropout_keep_mask(n, p, seed, offset, device):
    keep = torch.empty(n, device=device, dtype=U8)
    check(lib.laed_dropout_keep_mask(ptr(keep, U8), n, float(p), seed, offset, stream()), "laed_dropout_keep_mask")
    return keep


def adam_step_multi(ps, gs, ms, vs, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """torch.optim.Adam's update for several parameter tensors that share the step count, in one launch per 16 tensors."""
    import ctypes
    n = len(ps)
    if n == 0:
        return
    for t in list(ps) + list(gs) + list(ms) + list(vs):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.LaedError("adam_step_multi: fp32 contiguous tensors expected")
    arr = lambda ts: (ctypes.c_void_p * n)(*[ptr(t).value for t in ts])        # ptr() refuses CPU tensors
    sizes = (ctypes.c_int64 * n)(*[t.numel() for t in ps])
    for a, b, c, d in zip(ps, gs, ms, vs):
        if not (a.numel() == b.numel() == c.numel() == d.numel()):
            raise _lib.LaedError("adam_step_multi: size mismatch")
    check(lib.laed_adam_step_multi(n, arr(ps), arr(gs), arr(ms), arr(vs), sizes, int(step), lr, beta1, beta2, eps,
                                   float(grad_scale), stream()), "laed_adam_step_multi")


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    check(lib.laed_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), int(step), lr, beta1, beta2, eps,
                             float(grad_scale), stream()), "laed_adam_step")


def nll_reduce(tok_nll, labels, pad_id):
    out = torch.empty(2, device=tok_nll.device, dtype=torch.float32)
    check(lib.laed_nll_reduce(ptr(tok_nll), ptr(labels, I64), tok_nll.numel(), int(pad_id), ptr(out), stream()),
          "laed_nll_reduce")
    return out


_AVG_MODES = {None: 0, 'seq': 1, 'real_word': 2, 'word': 3}


def nll_prepare(labels, pad_id, avg_type, world, global_count=None, time_major=True, want_local_count=False):
    """labels [B,T] int64 (batch-first view, unit column stride) -> (labels_out [B*T], scale_out [B*T], local_count|None):
    NLLEntropy's per-row weights, re-ordered labels and non-PAD count in one launch (laed_nll_prepare)."""
    if avg_type not in _AVG_MODES:
        raise ValueError("Unknown avg type")
    if labels.dim() != 2 or labels.stride(1) != 1 or labels.dtype != I64:
        labels = labels.long().contiguous()
    B, T = labels.shape
    dev = labels.device
    import ctypes
    if not labels.is_cuda:
        raise _lib.LaedError("laed_amd kernels need CUDA (ROCm) tensors; got a CPU tensor")
    labels_out = torch.empty(B * T, device=dev, dtype=I64)
    scale_out = torch.empty(B * T, device=dev, dtype=torch.float32)
    local = torch.empty(1, device=dev, dtype=torch.float32) if want_local_count else None
    check(lib.laed_nll_prepare(ctypes.c_void_p(labels.data_ptr()), labels.stride(0) if B > 1 else max(T, 1), B, T, int(pad_id),
                               _AVG_MODES[avg_type], float(B * world), ptr(global_count), int(bool(time_major)),
                               ptr(labels_out, I64), ptr(scale_out), ptr(local), stream()), "laed_nll_prepare")
    return labels_out, scale_out, local


def scale_by_dev_multi_(tensors, g):
    """t *= g[0] for every tensor of the list, one launch (nothing moves when g[0] == 1)."""
    import ctypes
    ts = [t for t in tensors if t is not None and t.numel() > 0]
    for i in range(0, len(ts), 8):
        grp = ts[i:i + 8]
        n = len(grp)
        arr = (ctypes.c_void_p * n)(*[ptr(t).value for t in grp])
        sizes = (ctypes.c_int64 * n)(*[t.numel() for t in grp])
        check(lib.laed_scale_by_dev_multi(n, arr, sizes, ptr(g), stream()), "laed_scale_by_dev_multi")


def proj_ldl(V):
    return int(lib.laed_proj_nll_ldl(int(V)))


_WS = {}


def workspace(key, numel, device):
    """Reusable fp32 scratch (grown on demand, one per key / device / STREAM: two streams of one model -- DiVST's two
    decoders -- never share scratch); contents are undefined between calls."""
    k = (key, device, torch.cuda.current_stream(device).cuda_stream)
    t = _WS.get(k)
    if t is None or t.numel() < numel:
        t = torch.empty(numel, device=device, dtype=torch.float32)
        _WS[k] = t
    return t


def proj_chunk_rows(N, V, H=1024):
    """Rows per logits chunk.  The dh GEMM of a chunk has (rows/128) x (H/128) block tiles and the chip runs
    2 blocks on each of its 256 CUs, so rows are chosen to make that a multiple of 512 tiles (a multiple of 8192 rows
    at H=1024) within ~8 GiB of workspace; problems that fit use one chunk."""
    ldl = proj_ldl(V)
    forced = int(os.environ.get("LAED_PROJ_CHUNK_ROWS", "0"))      # experiment hook
    if forced > 0:
        return int(min(max(N, 1), forced))
    # rows that fit 8 GiB of fp32 workspace: fewer, longer chunks mean fewer read-modify-write passes over dW
    # (205 MB each at the scaled shape) and fewer launch tails; 40960-row chunks measured -0.3 % step time against
    # 8192-row ones at B = 4096, -0.4 % at B = 512 (one chunk), on a GPU with 288 GB
    # (the split-bf16 mode, whose transposed operand copies grow with the chunk, is 2.6 % faster with 2 GiB chunks)
    cap = max(1024, ((1 << 29) if get_gemm_mode() == "bf16x3" else (1 << 31)) // ldl)
    # ... but never more than half of the [N, V] tensor once that tensor would be big (> 512 MiB): a rank that holds 512
    # sequences of the scaled shape (N = 19968) must not materialise its whole 4 GB logits tensor just because it fits
    # the cap.  Two 9984-row chunks cost +0.4 % step time there (59.68 -> 59.91 ms, tools/chunk_sweep.sh); chunks sized
    # to the 256 MB Infinity Cache (1280 rows) cost +6.5 % (B = 512) / +8.6 % (B = 4096): the dW read-modify-write and
    # the launch tails per chunk outweigh the cache hits, so chunks stay long.
    if N * ldl * 4 > (512 << 20):
        half = ((N + 1) // 2 + 127) // 128 * 128
        if half <= cap:
            return int(half)                                       # two balanced chunks
    if N <= cap:
        return int(max(N, 1))
    tiles_n = max(1, (H + 127) // 128)
    unit = max(128, (512 // tiles_n) * 128) if tiles_n <= 512 else 128
    rows = (cap // unit) * unit if cap >= unit else (cap // 128) * 128
    return int(max(rows, 128))


def proj_nll_raw(h, W, b, labels, row_scale, pad_id, with_grad, chunk_rows=None):
    N, H = h.shape
    V = W.shape[0]
    dev = h.device
    if not with_grad and get_gemm_mode() == "fp32":
        # forward only: softmax partials per 128-column tile, O(rows * V / 128) floats; no logits tile at all
        chunk_rows = max(1, N)
        ws_n = max(4, int(lib.laed_proj_nll_fwd_workspace(int(N), int(V))))
    else:
        if chunk_rows is None:
            chunk_rows = proj_chunk_rows(N, V, H)
        ws_n = int(lib.laed_proj_nll_workspace(int(H), int(V), int(chunk_rows)))
    ws = workspace("proj_fwd" if not with_grad else "proj", ws_n, dev)
    tok = torch.empty(N, device=dev, dtype=torch.float32)
    lse = torch.empty(N, device=dev, dtype=torch.float32)
    dh = dW = db = None
    if with_grad:
        dh = torch.empty_like(h)
        dW = torch.empty_like(W)
        db = torch.empty(V, device=dev, dtype=torch.float32)
    check(lib.laed_proj_nll(ptr(h), ptr(W), ptr(b), ptr(labels, I64), ptr(row_scale), N, H, V, int(pad_id),
                            int(with_grad), 0, ptr(ws), ws_n, int(chunk_rows), ptr(tok), ptr(lse),
                            ptr(dh), ptr(dW), ptr(db), stream()), "laed_proj_nll")
    return tok, lse, dh, dW, db


def logits_rows(h, W, b):
    """[N, ldl] logits buffer (fresh allocation, only for callers that want the full tensor)."""
    N = h.shape[0]
    V = W.shape[0]
    ldl = proj_ldl(V)
    buf = torch.empty((N, ldl), device=h.device, dtype=torch.float32)
    sgemm(h, W, transB=True, bias=b, out=buf[:, :V])
    return buf


def log_softmax_rows_(buf, V):
    N, ldl = buf.shape
    check(lib.laed_log_softmax_rows(ptr(buf), N, V, ldl, None, stream()), "laed_log_softmax_rows")
    return buf


class DenseLogProbsFunction(torch.autograd.Function):
    """h[N,H], W[V,H], b[V] -> log_softmax(h W^T + b) as a dense [N,V] tensor (a view of an [N,ldl] buffer), WITH a
    backward: the reference's DecoderRNN returns this tensor (decoders.py:165-166) and a caller may build any loss on
    it.  The fused NLL path never goes through here; this is the contract-keeping slow road (the [N,V] tensor and its
    gradient are materialised)."""

    @staticmethod
    def forward(ctx, h, W, b):
        h = _f32c(h)
        V = W.shape[0]
        buf = logits_rows(h, W, b)
        log_softmax_rows_(buf, V)
        ctx.save_for_backward(h, W, buf)
        ctx.has_bias = b is not None
        return buf[:, :V]

    @staticmethod
    def backward(ctx, dlogp):
        h, W, buf = ctx.saved_tensors
        N, V = h.shape[0], W.shape[0]
        if dlogp.dtype != torch.float32 or dlogp.stride(1) != 1:
            dlogp = dlogp.float().contiguous()
        dl = torch.empty((N, V), device=h.device, dtype=torch.float32)
        check(lib.laed_log_softmax_rows_bwd(ctypes_ptr(buf), buf.stride(0), ctypes_ptr(dlogp), dlogp.stride(0) if N > 1 else V,
                                            N, V, ptr(dl), V, stream()), "laed_log_softmax_rows_bwd")
        dh = sgemm(dl, W) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            db = torch.empty(V, device=h.device, dtype=torch.float32) if ctx.has_bias else None
            dW = sgemm(dl, h, transA=True, colsum_out=db)
        return dh, dW, db


def decode_greedy(emb, w_ih, w_hh, b_ih, b_hh, proj_w, proj_b, h0, max_len, go_id, sample=False, u=None, seed=0,
                  row0=0):
    """Free-running decode from h0 [B,H]: greedy, or (sample=True) Gumbel-max sampling with replayed uniforms
    u [max_len,B,V] or the counter RNG keyed by (seed, step, row0 + row)."""
    B, H = h0.shape
    V, E = emb.shape
    dev = h0.device
    hbuf = torch.empty((max_len + 1, B, H), device=dev, dtype=torch.float32)
    xbuf = torch.empty((B, E), device=dev, dtype=torch.float32)
    gates = torch.empty((B, 3 * H), device=dev, dtype=torch.float32)
    logits = torch.empty((B, proj_ldl(V)), device=dev, dtype=torch.float32)
    ids = torch.empty((max_len + 1, B), device=dev, dtype=I64)
    if sample:
        if u is not None and tuple(u.shape) != (max_len, B, V):
            raise ValueError("decode: u must be [max_len, B, V] = %s, got %s" % ((max_len, B, V), tuple(u.shape)))
        check(lib.laed_gru_decode_sample(ptr(emb), ptr(w_ih), ptr(w_hh), ptr(b_ih), ptr(b_hh), ptr(proj_w), ptr(proj_b),
                                         ptr(h0), B, E, H, V, int(max_len), int(go_id), ptr(u), int(seed), int(row0),
                                         ptr(hbuf), ptr(xbuf), ptr(gates), ptr(logits), ptr(ids, I64), stream()),
              "laed_gru_decode_sample")
        return ids[1:], hbuf
    check(lib.laed_gru_decode_greedy(ptr(emb), ptr(w_ih), ptr(w_hh), ptr(b_ih), ptr(b_hh), ptr(proj_w), ptr(proj_b),
                                     ptr(h0), B, E, H, V, int(max_len), int(go_id), ptr(hbuf), ptr(xbuf), ptr(gates),
                                     ptr(logits), ptr(ids, I64), stream()), "laed_gru_decode_greedy")
    return ids[1:], hbuf


def gru_cell(x, w_ih, w_hh, b_ih, b_hh, hprev):
    """One GRU step (no tape): x [B,E], hprev [B,H] -> h [B,H]."""
    B, E = x.shape
    H = hprev.shape[1]
    gates = torch.empty((B, 3 * H), device=x.device, dtype=torch.float32)
    hout = torch.empty_like(hprev)
    check(lib.laed_gru_cell_fwd(ptr(_f32c(x)), ptr(w_ih), ptr(w_hh), ptr(b_ih), ptr(b_hh), ptr(_f32c(hprev)), B, E, H,
                                ptr(gates), ptr(hout), stream()), "laed_gru_cell_fwd")
    return hout


def proj_argmax(h, W, b):
    """argmax_v (h W^T + b) per row, first maximum: h [B,H] -> ids [B] int64 (the greedy symbol of decoders.py:219)."""
    B, H = h.shape
    V = W.shape[0]
    logits = torch.empty((B, proj_ldl(V)), device=h.device, dtype=torch.float32)
    ids = torch.empty(B, device=h.device, dtype=I64)
    check(lib.laed_proj_argmax(ptr(_f32c(h)), ptr(W), ptr(b), B, H, V, ptr(logits), ptr(ids, I64), stream()), "laed_proj_argmax")
    return ids


# ----------------------------------------------------------------------------------------------
# autograd glue
# ----------------------------------------------------------------------------------------------
def _single_use(ctx, name):
    """GRUFunction / ProjNLLFunction rewrite their saved buffers in place during backward (reserve-space style), so a
    second backward over the same graph (retain_graph=True) would silently produce wrong gradients: refuse it."""
    if getattr(ctx, "_laed_consumed", False):
        raise RuntimeError("%s: backward was already run on this graph; its saved buffers are single-use "
                           "(re-run the forward pass instead of retain_graph=True)" % name)
    ctx._laed_consumed = True


class EmbedFunction(torch.autograd.Function):
    """out[rows,E] = weight[ids] (* keep/(1-p)); PAD rows get no gradient."""

    @staticmethod
    def forward(ctx, weight, ids, keep, drop_p, pad_id):
        rows = ids.numel()
        V, E = weight.shape
        out = torch.empty((rows, E), device=weight.device, dtype=torch.float32)
        check(lib.laed_embed_gather(ptr(ids, I64), rows, ptr(weight), V, E, ptr(keep, U8) if keep is not None else None,
                                    float(drop_p), ptr(out), stream()), "laed_embed_gather")
        ctx.save_for_backward(ids, keep)
        ctx.meta = (V, E, float(drop_p), int(pad_id))
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, keep = ctx.saved_tensors
        V, E, p, pad_id = ctx.meta
        dout = _f32c(dout)
        dw = torch.zeros((V, E), device=dout.device, dtype=torch.float32)
        ws_b = int(lib.laed_embed_scatter_workspace(ids.numel(), V, E))      # deterministic (sorted) scatter-add
        ws = workspace("scatter", (ws_b + 3) // 4, dout.device) if ws_b > 0 else None
        check(lib.laed_embed_scatter_add(ptr(ids, I64), ids.numel(), ptr(dout), V, E,
                                         ptr(keep, U8) if keep is not None else None, p, pad_id, ptr(dw),
                                         ptr(ws), ws_b, stream()),
              "laed_embed_scatter_add")
        return dw, None, None, None, None


class EmbedBTFunction(torch.autograd.Function):
    """ids [B,T] (batch-first, any row stride) -> out [T*B,E] TIME-major rows, in one launch: the ids' transposition, the
    input-dropout decisions (an injected time-major mask `keep_in`, or `rng` = (seed, offset) of the counter RNG keyed by
    the batch-first element index, or none) and the gather.  The transposed ids and the mask are kept for backward."""

    @staticmethod
    def forward(ctx, weight, ids, keep_in, rng, drop_p, pad_id):
        if ids.dim() != 2 or ids.stride(1) != 1 or ids.dtype != I64:
            raise _lib.LaedError("embed: ids must be an int64 [B,T] tensor with unit column stride")
        B, T = ids.shape
        V, E = weight.shape
        dev = weight.device
        out = torch.empty((T * B, E), device=dev, dtype=torch.float32)
        ids_tm = torch.empty(T * B, device=dev, dtype=I64)
        dropping = drop_p > 0.0 and (keep_in is not None or rng is not None)
        keep_out = torch.empty(T * B * E, device=dev, dtype=U8) if (dropping and keep_in is None) else None
        seed, off = rng if (rng is not None and keep_in is None) else (0, 0)
        import ctypes
        check(lib.laed_embed_gather_bt(ctypes.c_void_p(ids.data_ptr()), ids.stride(0) if B > 1 else max(T, 1), B, T, ptr(weight),
                                       V, E, ptr(keep_in, U8) if dropping else None, int(dropping and keep_in is None),
                                       float(drop_p) if dropping else 0.0, seed, off, ptr(out), ptr(ids_tm, I64),
                                       ptr(keep_out, U8), stream()), "laed_embed_gather_bt")
        keep = (keep_in if keep_in is not None else keep_out) if dropping else None
        ctx.save_for_backward(ids_tm, keep)
        ctx.meta = (V, E, float(drop_p) if dropping else 0.0, int(pad_id))
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, keep = ctx.saved_tensors
        V, E, p, pad_id = ctx.meta
        dout = _f32c(dout)
        dw = torch.zeros((V, E), device=dout.device, dtype=torch.float32)
        ws_b = int(lib.laed_embed_scatter_workspace(ids.numel(), V, E))
        ws = workspace("scatter", (ws_b + 3) // 4, dout.device) if ws_b > 0 else None
        check(lib.laed_embed_scatter_add(ptr(ids, I64), ids.numel(), ptr(dout), V, E,
                                         ptr(keep, U8) if keep is not None else None, p, pad_id, ptr(dw),
                                         ptr(ws), ws_b, stream()),
              "laed_embed_scatter_add")
        return dw, None, None, None, None, None


class LinearFunction(torch.autograd.Function):
    """y = x W^T + b on the fp32 MFMA GEMM; backward = two more GEMMs (+ bias grad as a column sum)."""

    @staticmethod
    def forward(ctx, x, W, b):
        x = _f32c(x)
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return sgemm(x, W, transB=True, bias=b)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = _f32c(dy)
        dx = sgemm(dy, W) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1] or ctx.has_bias:
            db = torch.empty(W.shape[0], device=W.device, dtype=torch.float32) if ctx.has_bias else None
            dW = sgemm(dy, x, transA=True, colsum_out=db)
        return dx, dW, db


def linear(x, W, b=None):
    return LinearFunction.apply(x, W, b)


_CU_SHARE = {"streams": 1, "processes": 1}


def set_device_sharers(n):
    """Tell the engine that `n` processes run on THIS GPU (a rehearsal of the multi-rank path on fewer GPUs than
    ranks; production is one process per GPU, n = 1).  The persistent small-batch GRU kernels need their whole grid
    resident at once, which nothing guarantees beside another process's work: with n > 1 every recurrence runs as
    per-step launches (see _cu_budget)."""
    _CU_SHARE["processes"] = max(1, int(n))


class concurrent_streams(object):
    """`with ops.concurrent_streams(2):` -- recurrences issued inside may run side by side on that many HIP streams
    (DI-VST's two decoders).  The persistent small-batch GRU kernels need their whole grid resident at once, so each
    gets 1/n of the device's CUs as its budget (forward and, recorded on the tape, backward); a grid that does not fit
    its share runs as per-step launches instead of waiting for CUs the other stream's grid holds."""

    def __init__(self, n):
        self.n = max(1, int(n))

    def __enter__(self):
        self.prev = _CU_SHARE["streams"]
        _CU_SHARE["streams"] = self.prev * self.n          # nested regions multiply (a bidirectional GRU inside a branch)
        return self

    def __exit__(self, *exc):
        _CU_SHARE["streams"] = self.prev
        return False


def _cu_budget(device):
    if _CU_SHARE["processes"] > 1:
        # Another PROCESS on this GPU: co-residency of a persistent grid cannot be counted on at all.  The hardware
        # scheduler time-slices the processes' queues (waves are saved and restored), so a grid that spins on its own
        # counters keeps being switched out under the spin: measured with two ranks on one MI355X (bench.py --backend
        # gloo, B = 64 per rank, H = 1024: the 128-workgroup backward grid fits half the CUs) single 64x64 GEMM launches
        # took up to 1 s and steps 0.1 .. 18 s -- results correct, no poll timed out.  A budget of one CU sends every
        # recurrence to the per-step launches (82 ms per step in the same rehearsal).
        return 1
    n = _CU_SHARE["streams"]
    from . import parallel
    dp = parallel.group() is not None
    if n <= 1 and not dp:
        return 0                                            # the whole device
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    if dp:
        # Data parallelism: the gradient reducer's RCCL kernels (one workgroup per channel, up to a few dozen) run on
        # their own stream beside the backward recurrences.  A persistent grid sized to every CU would have to wait for
        # them to leave (bounded, but a wait of seconds that ends in poisoned outputs); a quarter of the CUs is left free.
        cus = (cus * 3) // 4
    return max(1, cus // max(n, 1))


def gru_persist_timeouts():
    """Timed-out in-kernel waits of the persistent GRU kernels on the current device (synchronises the device)."""
    import ctypes
    out = ctypes.c_uint(0)
    check(lib.laed_gru_persist_timeouts(ctypes.byref(out)), "laed_gru_persist_timeouts")
    return int(out.value)


def gru_persist_launches():
    """(forward, backward) persistent-recurrence launches this process has issued so far (host counters)."""
    import ctypes
    f, b = ctypes.c_uint(0), ctypes.c_uint(0)
    check(lib.laed_gru_persist_launches(ctypes.byref(f), ctypes.byref(b)), "laed_gru_persist_launches")
    return int(f.value), int(b.value)


class GRUFunction(torch.autograd.Function):
    """x[T,B,E] (time-major) -> out[T,B,H], hT[B,H]."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h0, taped=True):
        x = _f32c(x)
        T, B, E = x.shape
        H = w_hh.shape[1]
        dev = x.device
        hbuf = torch.empty((T + 1, B, H), device=dev, dtype=torch.float32)
        gates = torch.empty((T, B, 3 * H), device=dev, dtype=torch.float32)
        # `taped` = the caller's torch.is_grad_enabled() (see ProjNLLFunction.forward): without a tape the step kernels
        # skip the reserve space (r, z, n and W_hn h + b_hn: four of the nine floats they move per output)
        keep = bool(taped) and any(ctx.needs_input_grad[:6])
        hn = torch.empty((T, B, H), device=dev, dtype=torch.float32) if keep else None
        h0c = _f32c(h0) if h0 is not None else None
        ws_n = int(lib.laed_gru_fwd_workspace(T, B, E, H))         # packed operands of the small-batch step kernels
        ws = workspace("gru_ws", ws_n, dev) if ws_n > 0 else None
        ctx.cu_budget = _cu_budget(dev)
        check(lib.laed_gru_fwd_cus(ptr(x), ptr(w_ih), ptr(w_hh), ptr(b_ih), ptr(b_hh), ptr(h0c), T, B, E, H,
                                   ptr(hbuf), ptr(gates), ptr(hn), ptr(ws), ws_n, ctx.cu_budget, stream()), "laed_gru_fwd")
        if keep:
            ctx.save_for_backward(x, w_ih, w_hh, hbuf, gates, hn)
        ctx.has_h0 = h0 is not None
        ctx.set_materialize_grads(False)          # an unused output (encoder: out; decoder: hT) arrives as None, not zeros
        out = hbuf[1:]
        hT = hbuf[T]
        return out, hT

    @staticmethod
    def backward(ctx, dout, dhT):
        _single_use(ctx, "GRUFunction")
        x, w_ih, w_hh, hbuf, gates, hn = ctx.saved_tensors
        T, B, E = x.shape
        H = w_hh.shape[1]
        dev = x.device
        dout = _f32c(dout) if dout is not None else None
        dhT = _f32c(dhT) if dhT is not None else None
        dgh = workspace("gru_dgh", T * B * 3 * H, dev)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw_ih = torch.empty_like(w_ih)
        dw_hh = torch.empty_like(w_hh)
        db_ih = torch.empty(3 * H, device=dev, dtype=torch.float32)
        db_hh = torch.empty(3 * H, device=dev, dtype=torch.float32)
        dh0 = torch.empty((B, H), device=dev, dtype=torch.float32)
        ws_n = int(lib.laed_gru_bwd_workspace(T, B, E, H))         # split-K scratch (+ transposed operands in bf16x3 mode)
        ws = workspace("gru_ws", ws_n, dev)
        # NOTE: overwrites the saved `gates` (r,z,n -> d gi): the tape is single-use, like cuDNN's reserve space
        check(lib.laed_gru_bwd_cus(ptr(x), ptr(w_ih), ptr(w_hh), ptr(hbuf), ptr(gates), ptr(hn), ptr(dout), ptr(dhT),
                                   T, B, E, H, ptr(dgh), ptr(dx), ptr(dw_ih), ptr(dw_hh), ptr(db_ih), ptr(db_hh), ptr(dh0), 0,
                                   ptr(ws), ws_n, ctx.cu_budget, stream()), "laed_gru_bwd")
        return dx, dw_ih, dw_hh, db_ih, db_hh, (dh0 if ctx.has_h0 else None), None


class GumbelLatentFunction(torch.autograd.Function):
    """logits[R,k], u[R,k] -> y (Gumbel-softmax sample or one-hot), log_qy = log_softmax(logits), ids[R]."""

    @staticmethod
    def forward(ctx, logits, u, tau, hard, rng=None):
        """u: uniforms [R,k] to replay, or None with rng = (seed, offset): drawn inside the kernel (same stream, same bits)."""
        logits = _f32c(logits)
        R, k = logits.shape
        y = torch.empty_like(logits)
        log_qy = torch.empty_like(logits)
        ids = torch.empty(R, device=logits.device, dtype=I64)
        if u is None:
            check(lib.laed_gumbel_latent_fwd_rng(ptr(logits), R, k, float(tau), int(bool(hard)), int(rng[0]), int(rng[1]),
                                                 ptr(y), ptr(ids, I64), ptr(log_qy), stream()), "laed_gumbel_latent_fwd_rng")
        else:
            check(lib.laed_gumbel_latent_fwd(ptr(logits), ptr(_f32c(u)), R, k, float(tau), int(bool(hard)), ptr(y),
                                             ptr(ids, I64), ptr(log_qy), stream()), "laed_gumbel_latent_fwd")
        ctx.save_for_backward(y, log_qy)
        ctx.meta = (float(tau), bool(hard))
        ctx.mark_non_differentiable(ids)
        ctx.set_materialize_grads(False)          # an unused output's gradient arrives as None, not as a zero-filled tensor
        return y, log_qy, ids

    @staticmethod
    def backward(ctx, dy, dlog_qy, _dids):
        y, log_qy = ctx.saved_tensors
        tau, hard = ctx.meta
        R, k = y.shape
        dy = None if (hard or dy is None) else _f32c(dy)          # the hard one-hot carries no gradient
        dlq = _f32c(dlog_qy) if dlog_qy is not None else None
        dlogits = torch.empty_like(y)
        check(lib.laed_gumbel_latent_bwd(ptr(y), ptr(dy), ptr(log_qy), ptr(dlq), R, k, tau, ptr(dlogits), stream()),
              "laed_gumbel_latent_bwd")
        return dlogits, None, None, None, None


class LatentLossFunction(torch.autograd.Function):
    """log_qy[B*y,k] -> (bpr, mi, row_kl) 0-dim tensors (+ the reduced statistics and the reduced `extra`).

    Batch statistics are all-reduced over `group` in ONE message: the y*k+2 latent sums and, behind them, whatever
    per-rank sums the caller hands in as `extra` (a 1-D float tensor: the non-PAD target counts of the step's
    decoders, whose global values the 'word' NLL divides by -- SURVEY.md 8e "fold into the same message").  Outputs 4
    and 5 are the reduced statistics [y*k+2] and the reduced extra (None without one); neither carries a gradient."""

    @staticmethod
    def forward(ctx, log_qy, y_size, k, group, extra=None):
        log_qy = _f32c(log_qy)
        dev = log_qy.device
        B = log_qy.shape[0] // y_size
        n = y_size * k + 2
        nblk = int(lib.laed_latent_stats_blocks())
        ws = workspace("latent", nblk * n, dev)
        ne = 0 if extra is None else int(extra.numel())
        msg = torch.empty(n + ne, device=dev, dtype=torch.float32)
        stats = msg[:n]
        scal = torch.empty(4, device=dev, dtype=torch.float32)
        coef = torch.empty(y_size * k, device=dev, dtype=torch.float32)
        Bg = float(B)
        if group is None:                                    # one GPU: nothing to exchange between statistics and losses
            check(lib.laed_latent_stats_finalize(ptr(log_qy), B, y_size, k, ptr(ws), ptr(stats), ptr(scal), ptr(coef), stream()),
                  "laed_latent_stats_finalize")
            if ne:
                msg[n:].copy_(extra.reshape(-1))
        else:
            check(lib.laed_latent_stats(ptr(log_qy), B, y_size, k, ptr(ws), ptr(stats), stream()), "laed_latent_stats")
            if ne:
                msg[n:].copy_(extra.reshape(-1))
            import torch.distributed as dist
            from . import parallel
            parallel.all_reduce_(msg, group)
            Bg = float(B * dist.get_world_size(group))       # equal shards per rank
            check(lib.laed_latent_finalize(ptr(stats), Bg, y_size, k, ptr(scal), ptr(coef), stream()), "laed_latent_finalize")
        ctx.save_for_backward(log_qy, coef, stats)
        ctx.meta = (B, Bg, y_size, k)
        ctx.set_materialize_grads(False)          # unused outputs (mi, row-KL in the default model) arrive as None
        extra_out = msg[n:] if ne else None
        ctx.mark_non_differentiable(stats, *([extra_out] if ne else []))
        return scal[0], scal[1], scal[2], stats, extra_out

    @staticmethod
    def backward(ctx, d_bpr, d_mi, d_rowkl, _dstats=None, _dextra=None):
        # mi is only logged by the reference's models (sent_models.py:155-158), but it is an ordinary differentiable
        # member of the returned Pack there, so its gradient is provided too
        log_qy, coef, stats = ctx.saved_tensors
        B, Bg, y_size, k = ctx.meta
        dlq = None
        for g, mode in ((d_bpr, 1), (d_rowkl, 0), (d_mi, 2)):
            if g is None:
                continue
            g = _f32c(g).reshape(1)
            first = dlq is None
            if first:
                dlq = torch.empty_like(log_qy)
            c = coef
            if mode == 2:
                c = torch.empty_like(coef)
                check(lib.laed_latent_mi_coef(ptr(stats), Bg, y_size, k, ptr(c), stream()), "laed_latent_mi_coef")
            check(lib.laed_latent_bwd(ptr(log_qy), ptr(c), ptr(g), mode, B, Bg, y_size, k, ptr(dlq), int(not first),
                                      stream()), "laed_latent_bwd")
        return dlq, None, None, None, None


class AvgLogQyFunction(torch.autograd.Function):
    """log_qy[B*y,k] -> avg_log_qy[y,k] = log(mean_b exp(log_qy) + 1e-15) (sent_models.py:138-139).  `stats` = the
    ALREADY all-reduced batch sums LatentLossFunction returned for the same log_qy (so every rank holds the
    global-batch value without a second exchange); None: computed and reduced over `group` here."""

    @staticmethod
    def forward(ctx, log_qy, y_size, k, group, stats=None):
        log_qy = _f32c(log_qy)
        dev = log_qy.device
        B = log_qy.shape[0] // y_size
        n = y_size * k
        Bg = float(B)
        if group is not None:
            import torch.distributed as dist
            Bg = float(B * dist.get_world_size(group))
        if stats is None:
            nblk = int(lib.laed_latent_stats_blocks())
            ws = workspace("latent", nblk * (n + 2), dev)
            stats = torch.empty(n + 2, device=dev, dtype=torch.float32)
            check(lib.laed_latent_stats(ptr(log_qy), B, y_size, k, ptr(ws), ptr(stats), stream()), "laed_latent_stats")
            if group is not None:
                from . import parallel
                parallel.all_reduce_(stats, group)
        avg = torch.empty((y_size, k), device=dev, dtype=torch.float32)
        check(lib.laed_avg_log_fwd(ptr(stats), Bg, n, ptr(avg), stream()), "laed_avg_log_fwd")
        ctx.save_for_backward(log_qy, stats)
        ctx.meta = (B, Bg, y_size, k)
        return avg

    @staticmethod
    def backward(ctx, davg):
        log_qy, stats = ctx.saved_tensors
        B, Bg, y_size, k = ctx.meta
        coef = torch.empty(y_size * k, device=log_qy.device, dtype=torch.float32)
        check(lib.laed_avg_log_coef(ptr(stats), ptr(_f32c(davg)), Bg, y_size * k, ptr(coef), stream()), "laed_avg_log_coef")
        dlq = torch.empty_like(log_qy)
        check(lib.laed_latent_bwd(ptr(log_qy), ptr(coef), None, 1, B, Bg, y_size, k, ptr(dlq), 0, stream()),
              "laed_latent_bwd")
        return dlq, None, None, None, None


class CatRowsFunction(torch.autograd.Function):
    """CatKLLoss / Entropy on arbitrary rows: log_qy[R,k] (, log_py 1-element | [k] | [1,k] | [R,1] | [R,k]) ->
    scale * sum_r row_r, row_r = sum_c q (lq - lp)  (log_py None: sum_c -lq q).  criterions.py:97-109, 136-147."""

    @staticmethod
    def forward(ctx, log_qy, log_py, scale):
        log_qy = _f32c(log_qy)
        ptr(log_qy)                                           # refuses CPU tensors before any device call
        R, k = log_qy.shape
        dev = log_qy.device
        py_rows, py_cols = 1, 1
        if log_py is not None:
            log_py = _f32c(log_py)
            shp = tuple(log_py.shape)
            if log_py.numel() == 1:
                py_rows, py_cols = 1, 1
            elif shp in ((k,), (1, k)):
                py_rows, py_cols = 1, k
            elif shp == (R, 1):
                py_rows, py_cols = R, 1
            elif shp == (R, k):
                py_rows, py_cols = R, k
            else:
                raise _lib.LaedError("CatKLLoss: log_py of shape %s does not broadcast to log_qy %s" % (shp, (R, k)))
        if ctx.needs_input_grad[1] and (py_rows, py_cols) != (R, k):
            raise NotImplementedError("CatKLLoss: a broadcast log_py that requires grad is not supported "
                                      "(the reference only trains full-shape priors)")
        part = workspace("cat_rows", int(lib.laed_cat_rows_blocks()), dev)
        out = torch.empty(1, device=dev, dtype=torch.float32)
        check(lib.laed_cat_rows_fwd(ptr(log_qy), ptr(log_py), py_rows, py_cols, R, k, float(scale), None, ptr(part),
                                    ptr(out), stream()), "laed_cat_rows_fwd")
        ctx.save_for_backward(log_qy, log_py)
        ctx.meta = (py_rows, py_cols, float(scale))
        return out[0]

    @staticmethod
    def backward(ctx, dout):
        log_qy, log_py = ctx.saved_tensors
        py_rows, py_cols, scale = ctx.meta
        R, k = log_qy.shape
        g = _f32c(dout).reshape(1)
        dlq = torch.empty_like(log_qy) if ctx.needs_input_grad[0] else None
        dlp = torch.empty_like(log_qy) if ctx.needs_input_grad[1] else None
        if dlq is not None or dlp is not None:
            check(lib.laed_cat_rows_bwd(ptr(log_qy), ptr(log_py), py_rows, py_cols, R, k, ptr(g), scale, ptr(dlq), ptr(dlp),
                                        stream()), "laed_cat_rows_bwd")
        if dlp is not None:
            dlp = dlp.view(log_py.shape)
        return dlq, dlp, None


class NLLGatherFunction(torch.autograd.Function):
    """Dense log-probs logp[N,V], labels[N], row_scale[N] -> (sum_r row_scale[r]*tok_nll[r], tok_nll[N]) with
    tok_nll[r] = -logp[r, labels[r]] on non-PAD rows (criterions.py:43-72 on a materialised tensor)."""

    @staticmethod
    def forward(ctx, logp, labels, row_scale, pad_id):
        logp = _f32c(logp)
        N, V = logp.shape
        tok = torch.empty(N, device=logp.device, dtype=torch.float32)
        check(lib.laed_nll_gather_fwd(ptr(logp), V, ptr(labels, I64), N, V, int(pad_id), ptr(tok), stream()),
              "laed_nll_gather_fwd")
        loss = torch.empty(1, device=logp.device, dtype=torch.float32)
        check(lib.laed_dot(ptr(tok), ptr(row_scale), N, ptr(loss), stream()), "laed_dot")
        ctx.save_for_backward(labels, row_scale)
        ctx.meta = (N, V, int(pad_id))
        ctx.mark_non_differentiable(tok)
        return loss[0], tok

    @staticmethod
    def backward(ctx, dloss, _dtok):
        labels, row_scale = ctx.saved_tensors
        N, V, pad_id = ctx.meta
        g = _f32c(dloss).reshape(1)
        dlogp = torch.empty((N, V), device=labels.device, dtype=torch.float32)
        check(lib.laed_nll_gather_bwd(ptr(labels, I64), ptr(row_scale), ptr(g), N, V, pad_id, ptr(dlogp), V, stream()),
              "laed_nll_gather_bwd")
        return dlogp, None, None, None


class TanhFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        y = torch.empty_like(x)
        check(lib.laed_tanh_fwd(ptr(x), ptr(y), x.numel(), stream()), "laed_tanh_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        check(lib.laed_tanh_bwd(ptr(y), ptr(_f32c(dy)), ptr(dx), y.numel(), stream()), "laed_tanh_bwd")
        return dx


def tanh(x):
    return TanhFunction.apply(x)


def _nt_strides(t3):
    """(utterance stride, position stride) of a [N,T,D] tensor whose last dimension is dense; the tensor itself when
    it is batch-first contiguous or a batch-first view of time-major contiguous memory, else a contiguous copy."""
    N, T, D = t3.shape
    if t3.stride(2) == 1 or D == 1:
        if t3.is_contiguous():
            return t3, T * D, D
        if t3.transpose(0, 1).is_contiguous():
            return t3, D, N * D
    t3 = t3.contiguous()
    return t3, T * D, D


class AttnPoolFunction(torch.autograd.Function):
    """enc[N,T,D], pre = key_w(enc) as rows [R,Hk] in the SAME (utterance, position) memory order as enc, query weight
    wq[1,Hk] and bias bq[1] -> (pooled[N,D], attn[N,T]).  encoders.py:119-130 minus the key_w GEMM, which stays a
    LinearFunction node in front of this one (its backward turns d pre into d enc and d key_w)."""

    @staticmethod
    def forward(ctx, enc, pre, wq, bq, time_major):
        N, T, D = enc.shape
        Hk = pre.shape[1]
        dev = enc.device
        fc1 = _f32c(pre).clone()                          # tanh in place; `pre` belongs to the GEMM node
        score = torch.empty(N * T, device=dev, dtype=torch.float32)
        check(lib.laed_attn_score_fwd(ptr(fc1), ptr(wq), ptr(bq), N * T, Hk, ptr(score), stream()), "laed_attn_score_fwd")
        s_sn, s_st = (1, N) if time_major else (T, 1)
        e_sn, e_st = (D, N * D) if time_major else (T * D, D)
        attn = torch.empty((N, T), device=dev, dtype=torch.float32)
        out = torch.empty((N, D), device=dev, dtype=torch.float32)
        check(lib.laed_attn_pool_fwd(ptr(score), s_sn, s_st, ctypes_ptr(enc), e_sn, e_st, N, T, D, ptr(attn), ptr(out), stream()),
              "laed_attn_pool_fwd")
        ctx.save_for_backward(enc, fc1, wq, attn)
        ctx.meta = (bool(time_major), bq is not None)
        ctx.mark_non_differentiable(attn)
        return out, attn

    @staticmethod
    def backward(ctx, dout, _dattn):
        enc, fc1, wq, attn = ctx.saved_tensors
        time_major, has_bq = ctx.meta
        N, T, D = enc.shape
        R, Hk = fc1.shape
        dev = enc.device
        s_sn, s_st = (1, N) if time_major else (T, 1)
        e_sn, e_st = (D, N * D) if time_major else (T * D, D)
        dscore = torch.empty(R, device=dev, dtype=torch.float32)
        denc = None
        if ctx.needs_input_grad[0]:                       # same memory order as enc
            denc = torch.empty((T, N, D) if time_major else (N, T, D), device=dev, dtype=torch.float32)
        check(lib.laed_attn_pool_bwd(ctypes_ptr(enc), e_sn, e_st, ptr(attn), ptr(_f32c(dout)), N, T, D, ptr(dscore), s_sn, s_st,
                                     ptr(denc), e_sn, e_st, stream()), "laed_attn_pool_bwd")
        dpre = torch.empty_like(fc1)
        check(lib.laed_attn_score_bwd(ptr(fc1), ptr(dscore), ptr(wq), R, Hk, ptr(dpre), stream()), "laed_attn_score_bwd")
        dbq = torch.empty(1, device=dev, dtype=torch.float32) if has_bq else None
        dwq = sgemm(dscore.view(R, 1), fc1, transA=True, colsum_out=dbq)      # [1,Hk] = dscore^T fc1; db = sum dscore
        if denc is not None and time_major:
            denc = denc.transpose(0, 1)
        return denc, dpre, dwq, dbq, None


def ctypes_ptr(t):
    """Device pointer of a dense-last-dim CUDA fp32 tensor addressed with explicit strides by the callee."""
    import ctypes
    if not t.is_cuda or t.dtype != torch.float32:
        raise _lib.LaedError("laed_amd kernels need fp32 CUDA (ROCm) tensors")
    return ctypes.c_void_p(t.data_ptr())


def attn_pool(enc_outs, key_w, query):
    """RnnUttEncoder's attention pooling: enc_outs [N,T,D] -> (utt_embedded [N,D], attn [N,T])."""
    if not enc_outs.is_cuda:
        raise _lib.LaedError("laed_amd kernels need CUDA (ROCm) tensors; got a CPU tensor")
    enc, e_sn, e_st = _nt_strides(_f32c(enc_outs) if enc_outs.dtype != torch.float32 else enc_outs)
    N, T, D = enc.shape
    time_major = (e_sn, e_st) == (D, N * D) and N > 1
    rows = enc.transpose(0, 1).reshape(T * N, D) if time_major else enc.reshape(N * T, D)      # views, no copy
    pre = linear(rows, key_w.weight, key_w.bias)
    return AttnPoolFunction.apply(enc, pre, query.weight, query.bias, time_major)


class ProjNLLFunction(torch.autograd.Function):
    """h[N,H], W[V,H], b[V], labels[N], row_scale[N] -> (loss = sum_r row_scale[r]*nll[r], tok_nll[N]).

    Forward and backward GEMMs run in ONE fused pass per row chunk (the gradients for d loss = 1 are produced
    while the logits chunk is still in the workspace); backward only rescales them by the incoming gradient.
    """

    @staticmethod
    def forward(ctx, h, W, b, labels, row_scale, pad_id, taped=True):
        h = _f32c(h)
        # `taped` = the caller's torch.is_grad_enabled(): inside forward() grad mode is always off, and needs_input_grad
        # only mirrors requires_grad of the inputs -- under no_grad (validation, engine.validate) W still "needs" a
        # gradient and the whole backward half (softmax-CE rows, dh and dW products: 2/3 of the FLOPs) would run for
        # nothing (measured: 384 ms per validation batch at the scaled shape instead of ~150)
        with_grad = bool(taped) and any(ctx.needs_input_grad[:3])
        tok, lse, dh, dW, db = proj_nll_raw(h, W, b, labels, row_scale, pad_id, with_grad)
        loss = torch.empty(1, device=h.device, dtype=torch.float32)
        check(lib.laed_dot(ptr(tok), ptr(row_scale), tok.numel(), ptr(loss), stream()), "laed_dot")
        loss = loss[0]
        if with_grad:
            ctx.save_for_backward(dh, dW, db)
        ctx.mark_non_differentiable(tok)
        ctx.set_materialize_grads(False)          # tok's (never defined) gradient arrives as None, not as an [N] zero fill
        return loss, tok

    @staticmethod
    def backward(ctx, dloss, _dtok):
        if dloss is None:
            return None, None, None, None, None, None, None
        _single_use(ctx, "ProjNLLFunction")
        dh, dW, db = ctx.saved_tensors
        g = _f32c(dloss).reshape(1)
        scale_by_dev_multi_((dh, dW, db), g)
        return dh, dW, db, None, None, None, None


class SoftmaxEmbedFunction(torch.autograd.Function):
    """h[N,H], W[V,H], b[V], Wemb[V,E] -> softmax(h W^T + b) Wemb  [N,E]; the [N,V] probabilities only ever exist
    as workspace tiles (dialog_models.py:421-422 'attribute forcing').  Backward recomputes the logits per chunk."""

    @staticmethod
    def forward(ctx, h, W, b, Wemb):
        h = _f32c(h)
        N, H = h.shape
        V, E = Wemb.shape
        dev = h.device
        chunk = proj_chunk_rows(N, V, H)
        ws_n = int(lib.laed_softmax_embed_workspace(int(V), int(chunk), 0))
        ws = workspace("proj", ws_n, dev)
        lse = torch.empty(N, device=dev, dtype=torch.float32)
        out = torch.empty((N, E), device=dev, dtype=torch.float32)
        check(lib.laed_softmax_embed_fwd(ptr(h), ptr(W), ptr(b), ptr(Wemb), N, H, V, E, ptr(ws), ws_n, int(chunk),
                                         ptr(lse), ptr(out), stream()), "laed_softmax_embed_fwd")
        ctx.save_for_backward(h, W, b, Wemb, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, W, b, Wemb, lse = ctx.saved_tensors
        N, H = h.shape
        V, E = Wemb.shape
        dev = h.device
        chunk = max(128, proj_chunk_rows(N, V, H) // 2) if N > 128 else proj_chunk_rows(N, V, H)   # two tiles
        ws_n = int(lib.laed_softmax_embed_workspace(int(V), int(chunk), 1))
        ws = workspace("proj", ws_n, dev)
        dh = torch.empty_like(h)
        dW = torch.empty_like(W)
        db = torch.empty(V, device=dev, dtype=torch.float32)
        dWemb = torch.empty_like(Wemb)
        check(lib.laed_softmax_embed_bwd(ptr(h), ptr(W), ptr(b), ptr(Wemb), ptr(lse), ptr(_f32c(dout)), N, H, V, E,
                                         ptr(ws), ws_n, int(chunk), ptr(dh), ptr(dW), ptr(db), ptr(dWemb), 0, stream()),
              "laed_softmax_embed_bwd")
        return dh, dW, db, dWemb


class LogSoftmaxKFunction(torch.autograd.Function):
    """log_softmax over the last (k <= 64) dimension of a [R,k] matrix: the posterior log q(y|x)."""

    @staticmethod
    def forward(ctx, logits):
        logits = _f32c(logits)
        R, k = logits.shape
        log_qy = torch.empty_like(logits)
        check(lib.laed_gumbel_latent_fwd(ptr(logits), None, R, k, 1.0, 0, None, None, ptr(log_qy), stream()),
              "laed_gumbel_latent_fwd")
        ctx.save_for_backward(log_qy)
        return log_qy

    @staticmethod
    def backward(ctx, dlq):
        (log_qy,) = ctx.saved_tensors
        R, k = log_qy.shape
        dlogits = torch.empty_like(log_qy)
        check(lib.laed_gumbel_latent_bwd(None, None, ptr(log_qy), ptr(_f32c(dlq)), R, k, 1.0, ptr(dlogits), stream()),
              "laed_gumbel_latent_bwd")
        return dlogits
