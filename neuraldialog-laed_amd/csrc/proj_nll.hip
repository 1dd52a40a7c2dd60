// Vocabulary projection fused with log-softmax + NLL and its backward.
//
// The [N][V] logits / log-prob tensor of the reference (decoders.py:165-166; 31.9 GB at the scaled shape)
// is never materialised.  Rows are processed in chunks of `chunk_rows`; per chunk, on one stream:
//   1. L = h_chunk W^T + b                         fp32 MFMA GEMM into the [chunk][ldl] workspace
//   2. row kernel: lse, tok_nll = lse - L[label];  L <- (softmax(L) - onehot(label)) * row_scale   in place
//   3. dh_chunk = L W                              fp32 MFMA GEMM
//   4. dW += L^T h_chunk,  db += colsum(L)         fp32 MFMA GEMM (bias gradient from its A-operand loads)
// i.e. exactly the 3 x 2NHV algorithmic FLOPs of forward+backward (no recomputation), and the workspace
// round trips are ~1 ps/logit of HBM/MALL traffic against ~13 ps/logit of fp32 MFMA time.
// A forward-only call (with_grad = 0: validation) stops after step 2 without the in-place rewrite.
#include "laed_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

enum { MODE_STATS = 0, MODE_GRAD = 1, MODE_LOGP = 2 };

struct MS { float m, s; };
__device__ __forceinline__ MS ms_combine(MS a, MS b) {
    const float m = fmaxf(a.m, b.m);
    MS r;
    r.m = m;
    r.s = (a.m == -INFINITY ? 0.f : a.s * expf(a.m - m)) + (b.m == -INFINITY ? 0.f : b.s * expf(b.m - m));
    return r;
}

// one block per row (grid-stride); 256 threads; float4 accesses (ldl % 4 == 0, workspace 16-byte aligned)
template <int MODE>
__global__ __launch_bounds__(256) void proj_rows_kernel(float* __restrict__ L, int64_t rows, int V, int64_t ldl,
                                                        const int64_t* __restrict__ labels, const float* __restrict__ row_scale,
                                                        int64_t pad_id, float* __restrict__ tok_nll, float* __restrict__ lse_out) {
    __shared__ float sm[4], ss[4];
    __shared__ float s_lab;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V4 = V >> 2;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        float* Lr = L + row * ldl;
        const float4* Lr4 = reinterpret_cast<const float4*>(Lr);
        int64_t lab = (labels != nullptr) ? labels[row] : -1;
        const bool ignored = (labels != nullptr) && (lab == pad_id || lab < 0 || lab >= V);
        if (MODE == MODE_GRAD && ignored) {   // ignored rows: zero loss, zero gradient
            for (int i = tid; i < V4; i += 256) reinterpret_cast<float4*>(Lr)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = V4 * 4 + tid; i < V; i += 256) Lr[i] = 0.f;
            if (tid == 0) { tok_nll[row] = 0.f; if (lse_out) lse_out[row] = 0.f; }
            continue;
        }
        // pass 1: online max / sum-exp
        MS t;
        t.m = -INFINITY;
        t.s = 0.f;
        for (int i = tid; i < V4; i += 256) {
            const float4 x = Lr4[i];
            const float cm = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
            const float m = fmaxf(t.m, cm);
            t.s = t.s * expf(t.m - m) + (expf(x.x - m) + expf(x.y - m)) + (expf(x.z - m) + expf(x.w - m));
            t.m = m;
        }
        for (int i = V4 * 4 + tid; i < V; i += 256) {
            const float x = Lr[i];
            const float m = fmaxf(t.m, x);
            t.s = t.s * expf(t.m - m) + expf(x - m);
            t.m = m;
        }
        if (t.m == -INFINITY) t.s = 0.f;   // thread saw no element (exp(-inf - -inf) = nan guard)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            MS ot;
            ot.m = __shfl_xor(t.m, o, 64);
            ot.s = __shfl_xor(t.s, o, 64);
            t = ms_combine(t, ot);
        }
        if (lane == 0) { sm[wave] = t.m; ss[wave] = t.s; }
        if (tid == 0 && labels != nullptr && !ignored) s_lab = Lr[lab];
        __syncthreads();
        MS a, b, c, d;
        a.m = sm[0]; a.s = ss[0]; b.m = sm[1]; b.s = ss[1]; c.m = sm[2]; c.s = ss[2]; d.m = sm[3]; d.s = ss[3];
        const MS tot = ms_combine(ms_combine(a, b), ms_combine(c, d));
        const float lse = tot.m + logf(tot.s);
        if (tid == 0) {
            if (lse_out) lse_out[row] = lse;
            if (tok_nll) tok_nll[row] = ignored ? 0.f : (labels != nullptr ? lse - s_lab : 0.f);
        }
        if (MODE == MODE_GRAD) {
            const float g = row_scale ? row_scale[row] : 1.0f;
            const int lab_i = (int)lab;
            for (int i = tid; i < V4; i += 256) {
                float4 x = Lr4[i];
                x.x = expf(x.x - lse); x.y = expf(x.y - lse); x.z = expf(x.z - lse); x.w = expf(x.w - lse);
                const int c0 = i * 4;
                if (lab_i >= c0 && lab_i < c0 + 4) {
                    if (lab_i == c0) x.x -= 1.0f; else if (lab_i == c0 + 1) x.y -= 1.0f;
                    else if (lab_i == c0 + 2) x.z -= 1.0f; else x.w -= 1.0f;
                }
                x.x *= g; x.y *= g; x.z *= g; x.w *= g;
                reinterpret_cast<float4*>(Lr)[i] = x;
            }
            for (int i = V4 * 4 + tid; i < V; i += 256) {
                float x = expf(Lr[i] - lse);
                if (i == lab_i) x -= 1.0f;
                Lr[i] = x * g;
            }
        } else if (MODE == MODE_LOGP) {
            for (int i = tid; i < V4; i += 256) {
                float4 x = Lr4[i];
                x.x -= lse; x.y -= lse; x.z -= lse; x.w -= lse;
                reinterpret_cast<float4*>(Lr)[i] = x;
            }
            for (int i = V4 * 4 + tid; i < V; i += 256) Lr[i] -= lse;
        }
        __syncthreads();   // sm/ss/s_lab reused by the next row
    }
}

// Register-resident variant: one 1024-thread block holds a whole row (NV float4 per thread, V <= 4096 NV), so the
// row is read ONCE and written once (2 passes over the [rows][ldl] workspace instead of 3): exact two-step softmax
// (row max, then sum of exp(x - max)), the exponentials are kept and reused for the gradient.
template <int MODE, int NV>
__global__ __launch_bounds__(1024) void proj_rows_reg_kernel(float* __restrict__ L, int64_t rows, int V, int64_t ldl,
                                                             const int64_t* __restrict__ labels, const float* __restrict__ row_scale,
                                                             int64_t pad_id, float* __restrict__ tok_nll, float* __restrict__ lse_out) {
    __shared__ float sred[2][16];
    __shared__ float s_lab;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n4 = (int)(ldl >> 2);
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        float4* Lr4 = reinterpret_cast<float4*>(L + row * ldl);
        const int64_t lab = (labels != nullptr) ? labels[row] : -1;
        const bool ignored = (labels != nullptr) && (lab == pad_id || lab < 0 || lab >= V);
        if (MODE == MODE_GRAD && ignored) {   // ignored rows: zero loss, zero gradient
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int i = tid + 1024 * c;
                if (i < n4) Lr4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (tid == 0) { tok_nll[row] = 0.f; if (lse_out) lse_out[row] = 0.f; }
            continue;
        }
        float4 x[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int i = tid + 1024 * c;
            x[c] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            if (i < n4) {
                x[c] = load_stream(&Lr4[i]);           // read once, rewritten below: no reuse for any cache to keep
                const int c0 = 4 * i;                  // columns >= V are workspace padding
                if (c0 + 3 >= V) {
                    if (c0 >= V) x[c].x = -INFINITY;
                    if (c0 + 1 >= V) x[c].y = -INFINITY;
                    if (c0 + 2 >= V) x[c].z = -INFINITY;
                    x[c].w = -INFINITY;
                }
            }
        }
        const int lab_i = ignored ? -1 : (int)lab;
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            m = fmaxf(m, fmaxf(fmaxf(x[c].x, x[c].y), fmaxf(x[c].z, x[c].w)));
            const int c0 = 4 * (tid + 1024 * c);
            if (lab_i >= c0 && lab_i < c0 + 4)
                s_lab = lab_i == c0 ? x[c].x : (lab_i == c0 + 1 ? x[c].y : (lab_i == c0 + 2 ? x[c].z : x[c].w));
        }
        m = wave_max(m);
        if (lane == 0) sred[0][wave] = m;
        __syncthreads();
        m = sred[0][0];
#pragma unroll
        for (int w = 1; w < 16; ++w) m = fmaxf(m, sred[0][w]);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NV; ++c) {          // exp(-inf - m) = 0 for padding
            x[c].x = expf(x[c].x - m); x[c].y = expf(x[c].y - m); x[c].z = expf(x[c].z - m); x[c].w = expf(x[c].w - m);
            sum += (x[c].x + x[c].y) + (x[c].z + x[c].w);
        }
        sum = wave_sum(sum);
        if (lane == 0) sred[1][wave] = sum;
        __syncthreads();
        sum = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) sum += sred[1][w];     // fixed order
        const float lg = logf(sum);
        const float lse = m + lg;
        if (tid == 0) {
            if (lse_out) lse_out[row] = lse;
            if (tok_nll) tok_nll[row] = ignored ? 0.f : (labels != nullptr ? lse - s_lab : 0.f);
        }
        if (MODE == MODE_GRAD) {
            const float g = row_scale ? row_scale[row] : 1.0f;
            const float sc = g / sum;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int i = tid + 1024 * c;
                if (i < n4) {
                    float4 v = x[c];
                    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
                    const int c0 = 4 * i;
                    if (lab_i >= c0 && lab_i < c0 + 4) {
                        if (lab_i == c0) v.x -= g; else if (lab_i == c0 + 1) v.y -= g;
                        else if (lab_i == c0 + 2) v.z -= g; else v.w -= g;
                    }
                    store_stream(&Lr4[i], v);
                }
            }
        } else if (MODE == MODE_LOGP) {
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int i = tid + 1024 * c;
                if (i < n4) {
                    float4 v = x[c];        // log(exp(x - m)) would lose precision: recompute from the stored logits
                    const float4 o = Lr4[i];
                    v.x = o.x - lse; v.y = o.y - lse; v.z = o.z - lse; v.w = o.w - lse;
                    Lr4[i] = v;
                }
            }
        }
        __syncthreads();   // sred / s_lab reused by the next row
    }
}

// Forward-only path: combine the per-tile (max, sum exp) partials the logits GEMM's online-softmax epilogue left in
// part[tiles][rows][2] (tile order: fixed, reproducible) into lse and the token NLL.  One thread per row: consecutive
// threads read consecutive rows of a tile's slab (coalesced float2).
__global__ __launch_bounds__(256) void softmax_stats_combine_kernel(const float* __restrict__ part, const float* __restrict__ lab_logit,
                                                                    int64_t rows, int tiles, int V, const int64_t* __restrict__ labels,
                                                                    int64_t pad_id, float* __restrict__ tok_nll, float* __restrict__ lse_out) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        float m = -INFINITY, s = 0.f;
        for (int t = 0; t < tiles; ++t) {
            const float2 ps = *reinterpret_cast<const float2*>(part + ((int64_t)t * rows + r) * 2);
            const float mm = fmaxf(m, ps.x);
            s = (m == -INFINITY ? 0.f : s * expf(m - mm)) + (ps.x == -INFINITY ? 0.f : ps.y * expf(ps.x - mm));
            m = mm;
        }
        const float lse = m + logf(s);
        const int64_t lab = labels[r];
        const bool ignored = lab == pad_id || lab < 0 || lab >= V;
        if (lse_out) lse_out[r] = lse;
        tok_nll[r] = ignored ? 0.f : lse - lab_logit[r];
    }
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ L, int64_t rows, int V, int64_t ldl,
                                                          int64_t* __restrict__ ids) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* Lr = L + row * ldl;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < V; i += 256) {
            const float x = Lr[i];
            if (x > bv) { bv = x; bi = i; }   // ascending i per thread: first max kept
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
            ids[row] = bi;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void nll_reduce_kernel(const float* __restrict__ tok, const int64_t* __restrict__ labels,
                                                          int64_t N, int64_t pad_id, float* __restrict__ out) {
    __shared__ float ss[16], sc[16];
    float s = 0.f, c = 0.f;
    for (int64_t i = threadIdx.x; i < N; i += 1024) {
        if (labels[i] != pad_id) { s += tok[i]; c += 1.0f; }
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) { ss[threadIdx.x >> 6] = s; sc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 16; ++w) { a += ss[w]; b += sc[w]; }
        out[0] = a;
        out[1] = b;
    }
}

__global__ __launch_bounds__(1024) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                   float* __restrict__ out) {
    __shared__ float ss[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += a[i] * b[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) ss[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += ss[w];
        out[0] = t;
    }
}

// Dense log-prob input (NLLEntropy handed a materialised [N,V] tensor, criterions.py:43-72): label gather and its
// scatter gradient.  A label outside [0,V) that is not the ignore index poisons its row with NaN (F.nll_loss raises).
// Everything NLLEntropy needs from the labels in ONE single-block launch (criterions.py:43-72): the per-row loss weights of
// the averaging mode, the labels in the order the consumer wants (time-major for the fused projection, batch-first for a
// dense log-prob tensor) and this rank's non-PAD count.  mode: 0 = sum (weights 1), 1 = 'seq' (1/denom), 2 = 'real_word'
// (1 / (sign(labels).sum(row) * denom), criterions.py:62), 3 = 'word' (1 / count; count = gcount[0] when the caller has
// already all-reduced it, else this batch's own non-PAD count).  denom = B * world size.
__global__ __launch_bounds__(1024) void nll_prepare_kernel(const int64_t* __restrict__ labels, int64_t ld, int B, int T,
                                                           int64_t pad, int mode, float denom, const float* __restrict__ gcount,
                                                           int time_major, int64_t* __restrict__ labels_out,
                                                           float* __restrict__ scale_out, float* __restrict__ local_count) {
    __shared__ int s_cnt;
    extern __shared__ float s_row[];               // mode 2: per-sequence weights
    const int64_t n = (int64_t)B * T;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    if ((mode == 3 && gcount == nullptr) || local_count != nullptr) {
        int c = 0;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) c += labels[(i / T) * ld + (i % T)] != pad ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((threadIdx.x & 63) == 0 && c != 0) atomicAdd(&s_cnt, c);       // integer: exact, order-free
    }
    if (mode == 2) {
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            int c = 0;
            for (int t = 0; t < T; ++t) { const int64_t l = labels[(int64_t)b * ld + t]; c += (l > 0) - (l < 0); }   // torch.sign
            s_row[b] = 1.0f / ((float)c * denom);
        }
    }
    __syncthreads();
    float w = 1.0f;
    if (mode == 1) w = 1.0f / denom;
    else if (mode == 3) w = 1.0f / (gcount != nullptr ? gcount[0] : (float)s_cnt);
    if (threadIdx.x == 0 && local_count != nullptr) local_count[0] = (float)s_cnt;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int b = (int)(i / T), t = (int)(i - (int64_t)b * T);
        const int64_t o = time_major ? (int64_t)t * B + b : i;
        if (labels_out != nullptr) labels_out[o] = labels[(int64_t)b * ld + t];
        scale_out[o] = mode == 2 ? s_row[b] : w;
    }
}

__global__ __launch_bounds__(256) void nll_gather_fwd_kernel(const float* __restrict__ logp, int64_t ld, const int64_t* __restrict__ labels,
                                                             int64_t N, int V, int64_t pad_id, float* __restrict__ tok) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lab = labels[r];
        float v = 0.f;
        if (lab != pad_id) v = (lab >= 0 && lab < V) ? -logp[r * ld + lab] : NAN;
        tok[r] = v;
    }
}

__global__ __launch_bounds__(256) void nll_gather_bwd_kernel(const int64_t* __restrict__ labels, const float* __restrict__ row_scale,
                                                             const float* __restrict__ gscale, int64_t N, int V, int64_t pad_id,
                                                             float* __restrict__ dlogp, int64_t ld) {
    const float g = gscale ? gscale[0] : 1.0f;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lab = labels[r];
        if (lab != pad_id && lab >= 0 && lab < V) dlogp[r * ld + lab] = -g * row_scale[r];
    }
}

// Backward of log-softmax over rows, for callers that differentiate the DENSE log-prob tensor (DecoderRNN's public
// contract, decoders.py:166): dlogits[r][v] = dlogp[r][v] - exp(logp[r][v]) * sum_v dlogp[r][v].  One block per row
// (grid-stride), two passes; scalar accesses so that any row stride works.  dlogits may alias dlogp.
__global__ __launch_bounds__(256) void log_softmax_rows_bwd_kernel(const float* __restrict__ logp, int64_t ldp, const float* dlogp,
                                                                   int64_t ldd, int64_t rows, int V, float* dlogits, int64_t ldo) {
    __shared__ float ss[4];
    const int tid = threadIdx.x;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* d = dlogp + row * ldd;
        const float* lp = logp + row * ldp;
        float s = 0.f;
        for (int i = tid; i < V; i += 256) s += d[i];
        s = wave_sum(s);
        if ((tid & 63) == 0) ss[tid >> 6] = s;
        __syncthreads();
        const float tot = (ss[0] + ss[1]) + (ss[2] + ss[3]);
        float* o = dlogits + row * ldo;
        for (int i = tid; i < V; i += 256) o[i] = d[i] - expf(lp[i]) * tot;
        __syncthreads();
    }
}

// dst[c][r] = src[r][c] for a [R][C] fp32 matrix (row stride ld), dst row stride ldd.  64x64 tiles through a
// padded LDS tile, float4 (16-byte) accesses on both sides when alignment allows: HBM-bound.
template <bool VEC>
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int64_t ld, int R, int C,
                                                        float* __restrict__ dst, int64_t ldd) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tq = threadIdx.x & 15, tr = threadIdx.x >> 4;   // 16 float4 columns x 16 rows per pass
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + tr + 16 * i, c = c0 + 4 * tq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R) {
            const float* sp = src + (int64_t)r * ld + c;
            if (VEC && c + 3 < C) v = *reinterpret_cast<const float4*>(sp);
            else { if (c < C) v.x = sp[0]; if (c + 1 < C) v.y = sp[1]; if (c + 2 < C) v.z = sp[2]; if (c + 3 < C) v.w = sp[3]; }
        }
        float* t = &tile[tr + 16 * i][4 * tq];
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + tr + 16 * i, r = r0 + 4 * tq;   // dst row c, 4 consecutive r
        if (c < C && r < R) {
            const int lc = tr + 16 * i, lr = 4 * tq;
            float4 v = make_float4(tile[lr][lc], tile[lr + 1][lc], tile[lr + 2][lc], tile[lr + 3][lc]);
            float* dp = dst + (int64_t)c * ldd + r;
            if (VEC && r + 3 < R) *reinterpret_cast<float4*>(dp) = v;
            else { dp[0] = v.x; if (r + 1 < R) dp[1] = v.y; if (r + 2 < R) dp[2] = v.z; if (r + 3 < R) dp[3] = v.w; }
        }
    }
}

}   // namespace

int laed_transpose_impl(const float* src, int64_t ld, int R, int C, float* dst, int64_t ldd, hipStream_t stream) {
    if (R <= 0 || C <= 0) return 0;
    const bool vec = aligned16(src) && aligned16(dst) && (ld % 4 == 0) && (ldd % 4 == 0);
    dim3 grid((C + 63) / 64, (R + 63) / 64);
    if (grid.y > 65535) return laed_fail(LAED_E_SIZE, "transpose: too many rows (%d)", R);
    if (vec) hipLaunchKernelGGL(transpose_kernel<true>, grid, dim3(256), 0, stream, src, ld, R, C, dst, ldd);
    else hipLaunchKernelGGL(transpose_kernel<false>, grid, dim3(256), 0, stream, src, ld, R, C, dst, ldd);
    LAED_LAUNCH_CHECK("transpose");
    return 0;
}

namespace {

inline int rows_grid(int64_t rows) { return (int)(rows < 4096 ? (rows < 1 ? 1 : rows) : 4096); }

template <int MODE>
int launch_rows(float* L, int64_t rows, int V, int64_t ldl, const int64_t* labels, const float* row_scale, int64_t pad_id,
                float* tok_nll, float* lse, hipStream_t stream) {
    const int64_t need = cdiv64(ldl / 4, 1024);     // float4 per thread for a register-resident row
    static const bool reg_on = []() { const char* e = laed_exp_hook("LAED_PROJ_ROWS_REG"); return !(e != nullptr && atoi(e) == 0); }();
    const dim3 grid(rows_grid(rows));
#define LAED_ROWS_REG(NV) hipLaunchKernelGGL((proj_rows_reg_kernel<MODE, NV>), grid, dim3(1024), 0, stream, L, rows, V, ldl, labels, \
                                             row_scale, pad_id, tok_nll, lse)
    if (reg_on && ldl % 4 == 0 && need <= 16) {
        if (need <= 1) LAED_ROWS_REG(1);
        else if (need <= 2) LAED_ROWS_REG(2);
        else if (need <= 4) LAED_ROWS_REG(4);
        else if (need <= 8) LAED_ROWS_REG(8);
        else if (need <= 13) LAED_ROWS_REG(13);
        else LAED_ROWS_REG(16);
    } else {
        hipLaunchKernelGGL((proj_rows_kernel<MODE>), grid, dim3(256), 0, stream, L, rows, V, ldl, labels, row_scale, pad_id, tok_nll, lse);
    }
#undef LAED_ROWS_REG
    LAED_LAUNCH_CHECK("proj_rows");
    return 0;
}

}   // namespace

// dh = dl W has only (rows/128) x (H/128) output tiles but K = V is long: chunks that do not fill whole rounds of the
// chip's 512 resident blocks (the last chunk, or every chunk of a small per-GPU batch) split K up to this many ways.
constexpr int DH_SPLITS = 8;
// dW = dl^T h has (V/128) x (H/128) output tiles: when that is under one round of the 512 block slots (the reference's
// own vocabulary and hidden sizes) its K = rows may be split up to this many ways too
constexpr int DW_SPLITS = 4;

static int64_t split_ws_floats(int H, int V, int chunk_rows) {
    int64_t n = (int64_t)DH_SPLITS * chunk_rows * H;
    const int64_t t128 = (int64_t)((V + 127) / 128) * ((H + 127) / 128);
    const int64_t dw = t128 < 512 ? (int64_t)DW_SPLITS * ((int64_t)V * H + V) : 0;
    return n > dw ? n : dw;
}

extern "C" int laed_proj_nll_ldl(int V) { return (V + 31) / 32 * 32; }

// floats of workspace laed_proj_nll needs for the current GEMM math mode
extern "C" int64_t laed_proj_nll_workspace(int H, int V, int chunk_rows) {
    const int64_t ldl = laed_proj_nll_ldl(V);
    int64_t n = (int64_t)chunk_rows * ldl;
    n += split_ws_floats(H, V, chunk_rows);            // split-K partials of the dh (long K = V) / small dW GEMMs
    if (laed_get_gemm_mode() == 1) {
        const int64_t ldr = (chunk_rows + 31) / 32 * 32;
        n += (int64_t)V * ldr + (int64_t)H * ldr + (int64_t)H * ldl;   // dl^T, h^T, W^T (K-contiguous operands)
    }
    return n;
}

// Workspace of a forward-only call (with_grad = 0) in the exact-fp32 mode: per-tile softmax partials + label logits,
// O(rows * V/128) floats -- the logits are never written anywhere.  (The split-bf16 mode has no such epilogue and
// keeps the chunked [chunk_rows][ldl] path: laed_proj_nll_workspace.)
extern "C" int64_t laed_proj_nll_fwd_workspace(int64_t N, int V) {
    const int64_t tiles = (V + 127) / 128;
    return N * tiles * 2 + N;
}

extern "C" int laed_proj_nll(const float* h, const float* W, const float* b, const int64_t* labels, const float* row_scale,
                             int64_t N, int H, int V, int64_t pad_id, int with_grad, int accumulate_w,
                             float* workspace, int64_t ws_floats, int chunk_rows,
                             float* tok_nll, float* lse, float* dh, float* dW, float* db, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAED_CHECK_ARG(N >= 0 && H > 0 && V > 0 && chunk_rows > 0, "proj_nll: bad size");
    if (N == 0) {      // an empty row set contributes nothing: dW / db are still outputs and must not stay undefined
        if (with_grad && !accumulate_w) {
            if (dW) LAED_HIP_TRY(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)V * H, stream));
            if (db) LAED_HIP_TRY(hipMemsetAsync(db, 0, sizeof(float) * (size_t)V, stream));
        }
        return 0;
    }
    LAED_CHECK_ARG(h && W && labels && workspace && tok_nll, "proj_nll: null pointer");
    LAED_CHECK_ARG(!with_grad || (dh && dW && db), "proj_nll: with_grad needs dh, dW, db");
    LAED_CHECK_ARG(aligned16(workspace), "proj_nll: workspace must be 16-byte aligned");
    if (!with_grad && laed_get_gemm_mode() == 0 && N < 0x7fffffff) {
        // forward only (validation, every GEN-side caller): ONE logits GEMM over all rows whose epilogue keeps, per
        // 128-column tile, each row's (max, sum exp) and the label's logit; then a combine pass.  No [rows][V] tile
        // exists at any point; chunk_rows is ignored.
        const int64_t need = laed_proj_nll_fwd_workspace(N, V);
        LAED_CHECK_ARG(ws_floats >= need, "proj_nll: forward-only workspace too small (%lld < %lld floats)",
                       (long long)ws_floats, (long long)need);
        const int tiles = (V + 127) / 128;
        float* part = workspace;
        float* lab_logit = workspace + (int64_t)N * tiles * 2;
        LAED_TRY(laed_sgemm_softmax_stats_impl((int)N, V, H, h, H, W, H, b, labels, part, lab_logit, stream));
        const int nblk = (int)(cdiv64(N, 256) < 4096 ? cdiv64(N, 256) : 4096);
        hipLaunchKernelGGL(softmax_stats_combine_kernel, dim3(nblk), dim3(256), 0, stream, part, lab_logit, N, tiles, V, labels, pad_id,
                           tok_nll, lse);
        LAED_LAUNCH_CHECK("softmax_stats_combine");
        return 0;
    }
    const int64_t ldl = laed_proj_nll_ldl(V);
    LAED_CHECK_ARG(ws_floats >= laed_proj_nll_workspace(H, V, chunk_rows), "proj_nll: workspace too small (%lld < %lld floats)",
                   (long long)ws_floats, (long long)laed_proj_nll_workspace(H, V, chunk_rows));
    bool acc = accumulate_w != 0;
    // split-bf16 mode: every product is brought to the K-contiguous (x W^T) form via fp32 transposed copies
    const bool xmode = laed_get_gemm_mode() == 1 && with_grad;
    const int64_t ldr = (chunk_rows + 31) / 32 * 32;
    float* dh_ws = workspace + (int64_t)chunk_rows * ldl;          // [DH_SPLITS][chunk_rows][H] split-K partials
    const int64_t dh_ws_floats = split_ws_floats(H, V, chunk_rows);
    float* dlT = dh_ws + dh_ws_floats;                             // [V][ldr]
    float* hT = dlT + (int64_t)V * ldr;                    // [H][ldr]
    float* WT = hT + (int64_t)H * ldr;                     // [H][ldl]
    if (xmode) LAED_TRY(laed_transpose_impl(W, H, V, H, WT, ldl, stream));
    for (int64_t r0 = 0; r0 < N; r0 += chunk_rows) {
        const int rows = (int)((N - r0) < chunk_rows ? (N - r0) : chunk_rows);
        const float* hc = h + r0 * H;
        LAED_TRY(laed_sgemm_impl(false, true, rows, V, H, hc, H, W, H, workspace, ldl, b, false, nullptr, false, stream));
        if (with_grad)
            LAED_TRY(launch_rows<MODE_GRAD>(workspace, (int64_t)rows, V, ldl, labels + r0, row_scale ? row_scale + r0 : nullptr, pad_id,
                                            tok_nll + r0, lse ? lse + r0 : nullptr, stream));
        else
            LAED_TRY(launch_rows<MODE_STATS>(workspace, (int64_t)rows, V, ldl, labels + r0, nullptr, pad_id, tok_nll + r0,
                                             lse ? lse + r0 : nullptr, stream));
        if (with_grad && xmode) {
            // dh = dl W  as  dl [rows][V] x (W^T [H][V])^T ;  dW += dl^T h  as  dl^T [V][rows] x (h^T [H][rows])^T
            LAED_TRY(laed_sgemm_impl(false, true, rows, H, V, workspace, ldl, WT, ldl, dh + r0 * H, H, nullptr, false, nullptr, false, stream));
            LAED_TRY(laed_transpose_impl(workspace, ldl, rows, V, dlT, ldr, stream));
            LAED_TRY(laed_transpose_impl(hc, H, rows, H, hT, ldr, stream));
            //   db = row sums of dl^T, taken from the A-operand registers of the same GEMM
            LAED_TRY(laed_sgemm_rowsum_impl(V, H, rows, dlT, ldr, hT, ldr, dW, H, acc, db, acc, nullptr, 0, stream));
            acc = true;
        } else if (with_grad) {
            LAED_TRY(laed_sgemm_splitk_impl(false, false, rows, H, V, workspace, ldl, W, H, dh + r0 * H, H, nullptr, false, nullptr,
                                            false, dh_ws, dh_ws_floats, false, nullptr, stream));
            //   (dh_ws is free again: the dh product above has consumed its partials in stream order)
            LAED_TRY(laed_sgemm_splitk_impl(true, false, V, H, rows, workspace, ldl, hc, H, dW, H, nullptr, acc, db, acc,
                                            dh_ws, dh_ws_floats, false, nullptr, stream));
            acc = true;
        }
    }
    return 0;
}

extern "C" int laed_log_softmax_rows(float* logits, int64_t N, int V, int ldl, float* lse, void* stream) {
    LAED_CHECK_ARG(N >= 0 && V > 0 && ldl >= V && ldl % 4 == 0, "log_softmax_rows: bad size (ldl must be a multiple of 4)");
    if (N == 0) return 0;
    LAED_CHECK_ARG(logits && aligned16(logits), "log_softmax_rows: null or unaligned pointer");
    return launch_rows<MODE_LOGP>(logits, N, V, (int64_t)ldl, nullptr, nullptr, (int64_t)-1, nullptr, lse, (hipStream_t)stream);
}

extern "C" int laed_log_softmax_rows_bwd(const float* logp, int64_t ld_logp, const float* dlogp, int64_t ld_dlogp, int64_t N, int V,
                                         float* dlogits, int64_t ld_dlogits, void* stream) {
    LAED_CHECK_ARG(N >= 0 && V > 0 && ld_logp >= V && ld_dlogp >= V && ld_dlogits >= V, "log_softmax_rows_bwd: bad size");
    if (N == 0) return 0;
    LAED_CHECK_ARG(logp && dlogp && dlogits, "log_softmax_rows_bwd: null pointer");
    hipLaunchKernelGGL(log_softmax_rows_bwd_kernel, dim3(rows_grid(N)), dim3(256), 0, (hipStream_t)stream, logp, ld_logp, dlogp,
                       ld_dlogp, N, V, dlogits, ld_dlogits);
    LAED_LAUNCH_CHECK("log_softmax_rows_bwd");
    return 0;
}

extern "C" int laed_argmax_rows(const float* logits, int64_t N, int V, int ldl, int64_t* ids, void* stream) {
    LAED_CHECK_ARG(N >= 0 && V > 0 && ldl >= V, "argmax_rows: bad size");
    if (N == 0) return 0;
    LAED_CHECK_ARG(logits && ids, "argmax_rows: null pointer");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows_grid(N)), dim3(256), 0, (hipStream_t)stream, logits, N, V, (int64_t)ldl, ids);
    LAED_LAUNCH_CHECK("argmax_rows");
    return 0;
}

extern "C" int laed_nll_reduce(const float* tok_nll, const int64_t* labels, int64_t N, int64_t pad_id, float* out, void* stream) {
    LAED_CHECK_ARG(N >= 0, "nll_reduce: bad size");
    LAED_CHECK_ARG(tok_nll && labels && out, "nll_reduce: null pointer");
    hipLaunchKernelGGL(nll_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, tok_nll, labels, N, pad_id, out);
    LAED_LAUNCH_CHECK("nll_reduce");
    return 0;
}

extern "C" int laed_nll_prepare(const int64_t* labels, int64_t ld, int B, int T, int64_t pad_id, int avg_mode, double denom,
                                const float* global_count, int time_major, int64_t* labels_out, float* scale_out,
                                float* local_count, void* stream) {
    LAED_CHECK_ARG(B > 0 && T > 0 && ld >= T && avg_mode >= 0 && avg_mode <= 3 && denom > 0, "nll_prepare: bad argument");
    LAED_CHECK_ARG(labels && scale_out, "nll_prepare: null pointer");
    LAED_CHECK_ARG(avg_mode != 2 || B <= 8192, "nll_prepare: 'real_word' supports at most 8192 sequences per call");
    hipLaunchKernelGGL(nll_prepare_kernel, dim3(1), dim3(1024), avg_mode == 2 ? (size_t)B * sizeof(float) : 0, (hipStream_t)stream,
                       labels, ld, B, T, pad_id, avg_mode, (float)denom, global_count, time_major, labels_out, scale_out, local_count);
    LAED_LAUNCH_CHECK("nll_prepare");
    return 0;
}

extern "C" int laed_dot(const float* a, const float* b, int64_t n, float* out, void* stream) {
    LAED_CHECK_ARG(n >= 0, "dot: bad size");
    LAED_CHECK_ARG(a && b && out, "dot: null pointer");
    hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, n, out);
    LAED_LAUNCH_CHECK("dot");
    return 0;
}

extern "C" int laed_nll_gather_fwd(const float* logp, int64_t ld, const int64_t* labels, int64_t N, int V, int64_t pad_id,
                                   float* tok_nll, void* stream) {
    LAED_CHECK_ARG(N >= 0 && V > 0 && ld >= V, "nll_gather_fwd: need V > 0 and ld >= V (got V=%d ld=%lld)", V, (long long)ld);
    if (N == 0) return 0;
    LAED_CHECK_ARG(logp && labels && tok_nll, "nll_gather_fwd: null pointer");
    const int nblk = (int)(cdiv64(N, 256) < 1024 ? cdiv64(N, 256) : 1024);
    hipLaunchKernelGGL(nll_gather_fwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, logp, ld, labels, N, V, pad_id, tok_nll);
    LAED_LAUNCH_CHECK("nll_gather_fwd");
    return 0;
}

extern "C" int laed_nll_gather_bwd(const int64_t* labels, const float* row_scale, const float* gscale, int64_t N, int V,
                                   int64_t pad_id, float* dlogp, int64_t ld, void* stream) {
    LAED_CHECK_ARG(N >= 0 && V > 0 && ld >= V, "nll_gather_bwd: need V > 0 and ld >= V (got V=%d ld=%lld)", V, (long long)ld);
    if (N == 0) return 0;
    LAED_CHECK_ARG(labels && row_scale && dlogp, "nll_gather_bwd: null pointer");
    LAED_HIP_TRY(hipMemsetAsync(dlogp, 0, (size_t)N * (size_t)ld * sizeof(float), (hipStream_t)stream));
    const int nblk = (int)(cdiv64(N, 256) < 1024 ? cdiv64(N, 256) : 1024);
    hipLaunchKernelGGL(nll_gather_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, labels, row_scale, gscale, N, V, pad_id,
                       dlogp, ld);
    LAED_LAUNCH_CHECK("nll_gather_bwd");
    return 0;
}
