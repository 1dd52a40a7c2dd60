// Discrete-latent kernels: posterior log-softmax + Gumbel-softmax sample (fwd/bwd) and the
// batch-prior-regularised KL / mutual-information statistics.  All are tiny HBM-bound row reductions:
// a row of k <= 64 classes lives on a W-lane sub-group of a wavefront (W = 16/32/64) and is reduced with
// __shfl_xor, so one 64-lane wave handles 64/W rows per instruction.
#include "laed_common.h"
#include <mutex>
#include <math.h>

namespace {

template <int W>
__device__ __forceinline__ float grp_sum(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, W);
    return v;
}
template <int W>
__device__ __forceinline__ float grp_max(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, W));
    return v;
}

template <int W>
__global__ __launch_bounds__(256) void gumbel_latent_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ u,
                                                                int64_t R, int k, float tau, int hard,
                                                                float* __restrict__ y, int64_t* __restrict__ ids,
                                                                float* __restrict__ log_qy, int use_rng, uint64_t seed,
                                                                uint64_t rng_offset) {
    constexpr int RPB = 256 / W;
    const int c = threadIdx.x % W;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / W;
    const bool valid = (row < R) && (c < k);
    const int64_t off = valid ? row * k + c : 0;
    const float l = valid ? logits[off] : -INFINITY;

    // log q(y|x) = log_softmax(logits)                         (sent_models.py:111)
    const float m = grp_max<W>(l);
    const float e = valid ? expf(l - m) : 0.f;
    const float s = grp_sum<W>(e);
    if (valid && log_qy != nullptr) log_qy[off] = (l - m) - logf(s);

    if (u == nullptr && !use_rng) return;   // posterior only ("Hidden2Discrete"); block-uniform: no shuffles follow for anyone
    // y = softmax((logits + g)/tau), g = -log(-log(u+eps)+eps)  (nn_lib.py:232-242)
    float a = -INFINITY;
    if (valid) {
        // the uniform is replayed from the caller's buffer or drawn here from the counter RNG: element (row, c) is draw
        // number rng_offset + row*k + c of stream `seed`, exactly what laed_uniform would have written at u[off]
        const float uu = use_rng ? uniform01(seed, rng_offset + (uint64_t)off) : u[off];
        const float g = -logf(-logf(uu + 1e-20f) + 1e-20f);
        a = (l + g) / tau;
    }
    const float m2 = grp_max<W>(a);
    const float e2 = valid ? expf(a - m2) : 0.f;
    const float s2 = grp_sum<W>(e2);
    float yv = e2 / s2;

    // ids = argmax_k y, first maximum                            (nn_lib.py:253)
    float bv = valid ? yv : -1.f;
    int bi = c;
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, W);
        const int oi = __shfl_xor(bi, o, W);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (hard) yv = (c == bi) ? 1.f : 0.f;                        // nn_lib.py:254-257
    if (valid) {
        y[off] = yv;
        if (c == 0 && ids != nullptr) ids[row] = bi;
    }
}

template <int W>
__global__ __launch_bounds__(256) void gumbel_latent_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                                const float* __restrict__ log_qy, const float* __restrict__ dlog_qy,
                                                                int64_t R, int k, float tau, float* __restrict__ dlogits) {
    constexpr int RPB = 256 / W;
    const int c = threadIdx.x % W;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / W;
    const bool valid = (row < R) && (c < k);
    const int64_t off = valid ? row * k + c : 0;
    float out = 0.f;
    if (dy != nullptr) {   // y = softmax(a), a = (logits+g)/tau : dlogits = y*(dy - sum(dy*y))/tau
        const float yv = valid ? y[off] : 0.f;
        const float dv = valid ? dy[off] : 0.f;
        const float dot = grp_sum<W>(yv * dv);
        out += yv * (dv - dot) / tau;
    }
    if (dlog_qy != nullptr) {   // log_softmax: dlogits = dlq - q*sum(dlq)
        const float dv = valid ? dlog_qy[off] : 0.f;
        const float q = valid ? expf(log_qy[off]) : 0.f;
        const float sm = grp_sum<W>(dv);
        out += dv - q * sm;
    }
    if (valid) dlogits[off] = out;
}

// ---- BPR / MI statistics -------------------------------------------------------------------
constexpr int STAT_BLOCKS = 256;
constexpr int MAXE = 8;   // y*k <= 8*256

__global__ __launch_bounds__(256) void latent_stats_kernel(const float* __restrict__ log_qy, int64_t B, int yk, int k,
                                                           float* __restrict__ ws) {
    __shared__ float red[2][4];
    float sq[MAXE];
#pragma unroll
    for (int i = 0; i < MAXE; ++i) sq[i] = 0.f;
    float ent = 0.f, kl = 0.f;
    const float logu = logf(1.0f / (float)k);
    // four rows in flight per pass (independent loads), accumulated in row order: the sums do not depend on the unroll
    constexpr int RU = 4;
    for (int64_t b0 = blockIdx.x; b0 < B; b0 += (int64_t)gridDim.x * RU) {
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
            const int e = threadIdx.x + i * 256;
            if (e < yk) {
                float lq[RU];
#pragma unroll
                for (int r = 0; r < RU; ++r) {
                    const int64_t b = b0 + (int64_t)r * gridDim.x;
                    lq[r] = b < B ? log_qy[b * yk + e] : -INFINITY;      // exp(-inf) = 0: contributes nothing
                }
#pragma unroll
                for (int r = 0; r < RU; ++r) {
                    if (lq[r] != -INFINITY) {
                        const float q = expf(lq[r]);
                        sq[i] += q;
                        ent += -1.0f * lq[r] * q;
                        kl += q * (lq[r] - logu);
                    }
                }
            }
        }
    }
    float* out = ws + (int64_t)blockIdx.x * (yk + 2);
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int e = threadIdx.x + i * 256;
        if (e < yk) out[e] = sq[i];
    }
    ent = wave_sum(ent);
    kl = wave_sum(kl);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = ent; red[1][wave] = kl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[yk] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        out[yk + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(256) void latent_stats_combine_kernel(const float* __restrict__ ws, int nblk, int n,
                                                                   float* __restrict__ stats) {
    // one wave per statistic: lanes stride over the block partials, then a fixed shuffle tree (reproducible)
    const int lane = threadIdx.x & 63;
    for (int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); e < n; e += gridDim.x * (blockDim.x >> 6)) {
        float s = 0.f;
        for (int b = lane; b < nblk; b += 64) s += ws[(int64_t)b * n + e];
        s = wave_sum(s);
        if (lane == 0) stats[e] = s;
    }
}

__device__ __forceinline__ void latent_finalize_block(const float* stats, float Bg, int y, int k, float* __restrict__ scal,
                                                      float* __restrict__ coef, float* s_kl, float* s_h) {
    // one block; thread e handles class element e of [y][k]; per-y sums through LDS
    const int yk = y * k;
    const float logu = logf(1.0f / (float)k);
    for (int e = threadIdx.x; e < yk; e += blockDim.x) {
        const float qbar = stats[e] / Bg;                       // torch.mean(dim=0)
        const float l = logf(qbar + 1e-15f);                    // sent_models.py:139
        const float el = expf(l);                               // CatKLLoss recomputes exp(log) (criterions.py:104)
        s_kl[e] = el * (l - logu);
        s_h[e] = -1.0f * l * el;
        if (coef != nullptr) coef[e] = el * ((l - logu) + 1.0f) / (qbar + 1e-15f) / ((float)y * Bg);
    }
    __syncthreads();
    // row sums (over the k classes, in class order) by one thread per latent variable, in place at the row's first slot;
    // thread 0 then adds the rows in row order: the same association as one thread walking everything, y times shorter
    for (int yy = threadIdx.x; yy < y; yy += blockDim.x) {
        float a = 0.f, h = 0.f;
        for (int c = 0; c < k; ++c) { a += s_kl[yy * k + c]; h += s_h[yy * k + c]; }
        s_kl[yy * k] = a;
        s_h[yy * k] = h;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bpr = 0.f, hq = 0.f;
        for (int yy = 0; yy < y; ++yy) { bpr += s_kl[yy * k]; hq += s_h[yy * k]; }
        bpr /= (float)y;                                        // unit_average=True -> mean over the y rows
        hq /= (float)y;
        const float rows = (float)y * Bg;
        scal[0] = bpr;
        scal[1] = hq - stats[yk] / rows;                        // mi = H(avg) - mean_rows H(q)   (sent_models.py:155-156)
        scal[2] = stats[yk + 1] / rows;                         // non-mutual reg_kl               (sent_models.py:150-151)
        scal[3] = hq;
    }
}

__global__ __launch_bounds__(256) void latent_finalize_kernel(const float* __restrict__ stats, float Bg, int y, int k,
                                                              float* __restrict__ scal, float* __restrict__ coef) {
    __shared__ float s_kl[MAXE * 256], s_h[MAXE * 256];
    latent_finalize_block(stats, Bg, y, k, scal, coef, s_kl, s_h);
}

// combine + finalize in ONE block (single-GPU path: no all-reduce sits between them): each statistic is summed over the
// block partials by one wave exactly as latent_stats_combine_kernel does it (same lanes, same shuffle tree: same bits),
// written to stats[] for the backward pass, and after a barrier the block runs the finalize arithmetic on them.
__global__ __launch_bounds__(1024) void latent_combine_finalize_kernel(const float* __restrict__ ws, int nblk, float* stats,
                                                                       float Bg, int y, int k, float* __restrict__ scal,
                                                                       float* __restrict__ coef) {
    __shared__ float s_kl[MAXE * 256], s_h[MAXE * 256];
    const int n = y * k + 2, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    // four statistics per wave and pass, so that their (dependent) load -> shuffle-tree chains overlap
    for (int e0 = (threadIdx.x >> 6) * 4; e0 < n; e0 += nw * 4) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int b = lane; b < nblk; b += 64) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (e0 + j < n) s[j] += ws[(int64_t)b * n + e0 + j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = wave_sum(s[j]);
            if (lane == 0 && e0 + j < n) stats[e0 + j] = t;
        }
    }
    __threadfence_block();
    __syncthreads();
    latent_finalize_block(stats, Bg, y, k, scal, coef, s_kl, s_h);
}

__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* __restrict__ log_qy, const float* __restrict__ coef,
                                                         const float* __restrict__ gscale, int use_mutual, int64_t n,
                                                         int yk, int k, float rows_g, float* __restrict__ dlog_qy,
                                                         int accumulate) {
    const float g = gscale ? gscale[0] : 1.0f;
    const float logu = logf(1.0f / (float)k);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float lq = log_qy[i];
        const float q = expf(lq);
        float d;
        if (use_mutual == 2) d = g * q * (coef[i % yk] + (lq + 1.0f) / rows_g);      // d mi / d log_qy (coef from mi_coef)
        else if (use_mutual) d = g * coef[i % yk] * q;          // d bpr / d log_qy = coef * q
        else d = g * q * ((lq - logu) + 1.0f) / rows_g;
        dlog_qy[i] = accumulate ? dlog_qy[i] + d : d;
    }
}

// ---- generic CatKLLoss / Entropy rows (criterions.py:93-109, 132-147) ------------------------
// row value  KL: sum_c q (lq - lp)   /  entropy (lp == nullptr): sum_c -1 * lq * q,   q = exp(lq)
// log_py is addressed as lp[(row % py_rows) * py_cols + (c % py_cols)] with py_rows in {1, R}, py_cols in {1, k}:
// the broadcasts torch allows between a [R,k] log_qy and a 1-element / [k] / [1,k] / [R,1] / [R,k] log_py.
constexpr int CATROW_BLOCKS = 256;

template <int W>
__global__ __launch_bounds__(256) void cat_rows_fwd_kernel(const float* __restrict__ log_qy, const float* __restrict__ log_py,
                                                           int64_t py_rows, int py_cols, int64_t R, int k,
                                                           float* __restrict__ rows_out, float* __restrict__ part) {
    constexpr int RPB = 256 / W;
    __shared__ float red[4];
    const int c = threadIdx.x % W, g = threadIdx.x / W;
    float acc = 0.f;      // lane 0 of each sub-group: this sub-group's rows, in row order
    for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < R; r0 += (int64_t)gridDim.x * RPB) {
        const int64_t row = r0 + g;
        const bool valid = (row < R) && (c < k);
        float v = 0.f;
        if (valid) {
            const float lq = log_qy[row * k + c];
            const float q = expf(lq);
            if (log_py != nullptr) v = q * (lq - log_py[(row % py_rows) * py_cols + (c % py_cols)]);
            else v = -1.0f * lq * q;
        }
        v = grp_sum<W>(v);
        if (valid && c == 0) {
            if (rows_out != nullptr) rows_out[row] = v;
            acc += v;
        }
    }
    acc = wave_sum(acc);                                 // fixed tree over the wave's sub-groups
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void cat_rows_combine_kernel(const float* __restrict__ part, int nblk, float scale,
                                                              float* __restrict__ out) {
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) s += part[b];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s * scale;
}

__global__ __launch_bounds__(256) void cat_rows_bwd_kernel(const float* __restrict__ log_qy, const float* __restrict__ log_py,
                                                           int64_t py_rows, int py_cols, int64_t R, int k,
                                                           const float* __restrict__ gscale, float scale,
                                                           float* __restrict__ dlog_qy, float* __restrict__ dlog_py) {
    const float g = (gscale ? gscale[0] : 1.0f) * scale;
    const int64_t n = R * k;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float lq = log_qy[i];
        const float q = expf(lq);
        if (log_py != nullptr) {
            const int64_t row = i / k;
            const int c = (int)(i - row * k);
            const float lp = log_py[(row % py_rows) * py_cols + (c % py_cols)];
            if (dlog_qy != nullptr) dlog_qy[i] = g * q * ((lq - lp) + 1.0f);     // d/dlq  q (lq - lp)
            if (dlog_py != nullptr) dlog_py[i] = -g * q;                          // full-shape log_py only
        } else if (dlog_qy != nullptr) {
            dlog_qy[i] = -g * q * (lq + 1.0f);                                    // d/dlq  -lq q
        }
    }
}

// MI = mean_y H(exp l) - mean_rows H(q_row), l = log(qbar + 1e-15) (sent_models.py:155-156, criterions.py:136-147):
// coef[e] = d [mean_y H(exp l)] / d q[b][e] = -exp(l) (l + 1) / (qbar + 1e-15) / (y Bg); the per-row entropy term is added
// by latent_bwd_kernel (mode 2).
__global__ __launch_bounds__(256) void mi_coef_kernel(const float* __restrict__ stats, float Bg, int y, int n, float* __restrict__ coef) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const float qbar = stats[e] / Bg;
        const float l = logf(qbar + 1e-15f);
        coef[e] = -1.0f * expf(l) * (l + 1.0f) / (qbar + 1e-15f) / ((float)y * Bg);
    }
}

// avg_log_qy = log(mean_b q + 1e-15) as a tensor of its own (sent_models.py:138-139), for the train_prior branch
// (:144-146) where it meets a learned prior in CatKLLoss.  stats = sum_b q (laed_latent_stats, all-reduced).
__global__ __launch_bounds__(256) void avg_log_fwd_kernel(const float* __restrict__ stats, float Bg, int n, float* __restrict__ out) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) out[e] = logf(stats[e] / Bg + 1e-15f);
}
// coef[e] = d out[e] / d q[b][e] * dav[e] = dav[e] / (qbar + 1e-15) / Bg;  d log_qy = coef .* q  (laed_latent_bwd)
__global__ __launch_bounds__(256) void avg_log_coef_kernel(const float* __restrict__ stats, const float* __restrict__ dav, float Bg,
                                                           int n, float* __restrict__ coef) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x)
        coef[e] = dav[e] / (stats[e] / Bg + 1e-15f) / Bg;
}

}   // namespace

#define DISPATCH_W(k, CALL)                         \
    do {                                            \
        if ((k) <= 16) { CALL(16); }                \
        else if ((k) <= 32) { CALL(32); }           \
        else { CALL(64); }                          \
    } while (0)

extern "C" int laed_gumbel_latent_fwd(const float* logits, const float* u, int64_t R, int k, float tau, int hard,
                                      float* y, int64_t* ids, float* log_qy, void* stream) {
    LAED_CHECK_ARG(R >= 0 && k > 0 && k <= 64, "gumbel_latent_fwd: need 0 < k <= 64 (got %d)", k);
    LAED_CHECK_ARG(tau > 0.f, "gumbel_latent_fwd: temperature must be positive");
    if (R == 0) return 0;
    LAED_CHECK_ARG(logits && ((u && y) || log_qy), "gumbel_latent_fwd: null pointer");
#define CALL(W) hipLaunchKernelGGL((gumbel_latent_fwd_kernel<W>), dim3((unsigned)cdiv64(R, 256 / W)), dim3(256), 0, \
                                   (hipStream_t)stream, logits, u, R, k, tau, hard, y, ids, log_qy, 0, (uint64_t)0, (uint64_t)0)
    DISPATCH_W(k, CALL);
#undef CALL
    LAED_LAUNCH_CHECK("gumbel_latent_fwd");
    return 0;
}

// The same kernel drawing its own uniforms (one launch instead of laed_uniform + laed_gumbel_latent_fwd): element (r, c)
// uses draw `offset + r*k + c` of the counter RNG stream `seed`, so the sample equals the two-call form bit for bit.
extern "C" int laed_gumbel_latent_fwd_rng(const float* logits, int64_t R, int k, float tau, int hard, uint64_t seed,
                                          uint64_t offset, float* y, int64_t* ids, float* log_qy, void* stream) {
    LAED_CHECK_ARG(R >= 0 && k > 0 && k <= 64, "gumbel_latent_fwd_rng: need 0 < k <= 64 (got %d)", k);
    LAED_CHECK_ARG(tau > 0.f, "gumbel_latent_fwd_rng: temperature must be positive");
    if (R == 0) return 0;
    LAED_CHECK_ARG(logits && y, "gumbel_latent_fwd_rng: null pointer");
#define CALL(W) hipLaunchKernelGGL((gumbel_latent_fwd_kernel<W>), dim3((unsigned)cdiv64(R, 256 / W)), dim3(256), 0, \
                                   (hipStream_t)stream, logits, (const float*)nullptr, R, k, tau, hard, y, ids, log_qy, 1, seed, offset)
    DISPATCH_W(k, CALL);
#undef CALL
    LAED_LAUNCH_CHECK("gumbel_latent_fwd_rng");
    return 0;
}

extern "C" int laed_gumbel_latent_bwd(const float* y, const float* dy, const float* log_qy, const float* dlog_qy,
                                      int64_t R, int k, float tau, float* dlogits, void* stream) {
    LAED_CHECK_ARG(R >= 0 && k > 0 && k <= 64, "gumbel_latent_bwd: need 0 < k <= 64 (got %d)", k);
    if (R == 0) return 0;
    LAED_CHECK_ARG(dlogits && (dy == nullptr || y) && (dlog_qy == nullptr || log_qy), "gumbel_latent_bwd: null pointer");
#define CALL(W) hipLaunchKernelGGL((gumbel_latent_bwd_kernel<W>), dim3((unsigned)cdiv64(R, 256 / W)), dim3(256), 0, \
                                   (hipStream_t)stream, y, dy, log_qy, dlog_qy, R, k, tau, dlogits)
    DISPATCH_W(k, CALL);
#undef CALL
    LAED_LAUNCH_CHECK("gumbel_latent_bwd");
    return 0;
}

extern "C" int laed_latent_stats_blocks(void) { return STAT_BLOCKS; }

extern "C" int laed_latent_stats(const float* log_qy, int64_t B, int y, int k, float* workspace, float* stats, void* stream) {
    LAED_CHECK_ARG(B > 0 && y > 0 && k > 0 && y * k <= MAXE * 256, "latent_stats: need y*k <= %d", MAXE * 256);
    LAED_CHECK_ARG(log_qy && workspace && stats, "latent_stats: null pointer");
    const int yk = y * k;
    const int nblk = (int)(B < STAT_BLOCKS ? B : STAT_BLOCKS);
    hipLaunchKernelGGL(latent_stats_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, log_qy, B, yk, k, workspace);
    LAED_LAUNCH_CHECK("latent_stats");
    hipLaunchKernelGGL(latent_stats_combine_kernel, dim3((yk + 2 + 3) / 4), dim3(256), 0, (hipStream_t)stream, workspace, nblk, yk + 2, stats);
    LAED_LAUNCH_CHECK("latent_stats_combine");
    return 0;
}

// ---- allocation-free variant of laed_latent_stats (no counterpart in the reference; see include/laed_hip.h) ----------
// The block partials ([STAT_BLOCKS][y*k + 2] floats) live in a file-scope device array instead of a caller workspace, so
// callers that cannot allocate (HIP graph capture, tight serving loops) need to bring inputs and outputs only.  The
// array's NAME is a host-side stand-in, not a device address: its address is resolved once per device with
// hipGetSymbolAddress (with that device current) and only that pointer is ever handed to kernels.  Every GPU holds its
// own copy (STAT_BLOCKS * (MAXE*256 + 2) floats = 2.1 MB) until the process ends.  Each call overwrites the partials it
// later reads (nblk rows written, the same nblk rows read), so nothing has to start at zero; calls that overlap on one
// GPU (two streams) share the array and race.
static __device__ __align__(256) float scratch[STAT_BLOCKS * (MAXE * 256 + 2)];

namespace {
std::mutex g_scratch_mu;
float* g_scratch_ptr[64] = {nullptr};     // per device
}

extern "C" int laed_latent_stats_noalloc(const float* log_qy, int64_t B, int y, int k, float* stats, void* stream) {
    LAED_CHECK_ARG(B > 0 && y > 0 && k > 0, "latent_stats_noalloc: bad size");
    if ((int64_t)y * k > (int64_t)MAXE * 256)
        return laed_fail(LAED_E_SIZE, "latent_stats_noalloc: y*k = %lld exceeds the built-in scratch (max y*k = %d)",
                         (long long)y * k, MAXE * 256);
    LAED_CHECK_ARG(log_qy && stats, "latent_stats_noalloc: null pointer");
    int dev = 0;
    LAED_HIP_TRY(hipGetDevice(&dev));
    LAED_CHECK_ARG(dev >= 0 && dev < 64, "latent_stats_noalloc: device index %d out of range", dev);
    float* ws;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        if (g_scratch_ptr[dev] == nullptr) {      // set-up, once per device, with that device current
            void* sym = nullptr;
            LAED_HIP_TRY(hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)));
            if (sym == nullptr) return laed_fail(LAED_E_BADARG, "latent_stats_noalloc: hipGetSymbolAddress returned NULL");
            g_scratch_ptr[dev] = (float*)sym;
        }
        ws = g_scratch_ptr[dev];
    }
    const int yk = y * k;
    const int nblk = (int)(B < STAT_BLOCKS ? B : STAT_BLOCKS);
    hipLaunchKernelGGL(latent_stats_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, log_qy, B, yk, k, ws);
    LAED_LAUNCH_CHECK("latent_stats (noalloc)");
    hipLaunchKernelGGL(latent_stats_combine_kernel, dim3((yk + 2 + 3) / 4), dim3(256), 0, (hipStream_t)stream, ws, nblk, yk + 2, stats);
    LAED_LAUNCH_CHECK("latent_stats_combine (noalloc)");
    return 0;
}

// laed_latent_stats + laed_latent_finalize for a job with no exchange between them (one GPU): two launches instead of
// three, same bits (the combine step keeps its per-statistic wave reduction).
extern "C" int laed_latent_stats_finalize(const float* log_qy, int64_t B, int y, int k, float* workspace, float* stats,
                                          float* scal, float* coef, void* stream) {
    LAED_CHECK_ARG(B > 0 && y > 0 && k > 0 && y * k <= MAXE * 256, "latent_stats_finalize: need y*k <= %d", MAXE * 256);
    LAED_CHECK_ARG(log_qy && workspace && stats && scal, "latent_stats_finalize: null pointer");
    const int yk = y * k;
    const int nblk = (int)(B < STAT_BLOCKS ? B : STAT_BLOCKS);
    hipLaunchKernelGGL(latent_stats_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, log_qy, B, yk, k, workspace);
    LAED_LAUNCH_CHECK("latent_stats");
    hipLaunchKernelGGL(latent_combine_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace, nblk,
                       stats, (float)B, y, k, scal, coef);
    LAED_LAUNCH_CHECK("latent_combine_finalize");
    return 0;
}

extern "C" int laed_latent_finalize(const float* stats, double B_global, int y, int k, float* scal, float* coef, void* stream) {
    LAED_CHECK_ARG(B_global > 0 && y > 0 && k > 0 && y * k <= MAXE * 256, "latent_finalize: bad size");
    LAED_CHECK_ARG(stats && scal, "latent_finalize: null pointer");
    hipLaunchKernelGGL(latent_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, (float)B_global, y, k, scal, coef);
    LAED_LAUNCH_CHECK("latent_finalize");
    return 0;
}

extern "C" int laed_latent_bwd(const float* log_qy, const float* coef, const float* gscale, int use_mutual,
                               int64_t B, double B_global, int y, int k, float* dlog_qy, int accumulate, void* stream) {
    LAED_CHECK_ARG(B > 0 && y > 0 && k > 0, "latent_bwd: bad size");
    LAED_CHECK_ARG(log_qy && dlog_qy && (!use_mutual || coef), "latent_bwd: null pointer");
    const int64_t n = B * y * k;
    const int nblk = (int)(cdiv64(n, 256) < 1024 ? cdiv64(n, 256) : 1024);
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, log_qy, coef, gscale, use_mutual, n,
                       y * k, k, (float)(B_global * y), dlog_qy, accumulate);
    LAED_LAUNCH_CHECK("latent_bwd");
    return 0;
}

extern "C" int laed_cat_rows_blocks(void) { return CATROW_BLOCKS; }

static bool cat_rows_py_ok(const float* log_py, int64_t py_rows, int py_cols, int64_t R, int k) {
    return log_py == nullptr || ((py_rows == 1 || py_rows == R) && (py_cols == 1 || py_cols == k));
}

extern "C" int laed_cat_rows_fwd(const float* log_qy, const float* log_py, int64_t py_rows, int py_cols, int64_t R, int k,
                                 float scale, float* rows_out, float* partials, float* out, void* stream) {
    LAED_CHECK_ARG(R > 0 && k > 0 && k <= 64, "cat_rows_fwd: need R > 0 and 0 < k <= 64 (got R=%lld k=%d)", (long long)R, k);
    LAED_CHECK_ARG(log_qy && partials && out, "cat_rows_fwd: null pointer");
    LAED_CHECK_ARG(cat_rows_py_ok(log_py, py_rows, py_cols, R, k), "cat_rows_fwd: log_py [%lld,%d] does not broadcast to [%lld,%d]",
                   (long long)py_rows, py_cols, (long long)R, k);
    int nblk = 0;
#define CALL(W) do { nblk = (int)(cdiv64(R, 256 / W) < CATROW_BLOCKS ? cdiv64(R, 256 / W) : CATROW_BLOCKS); \
                     hipLaunchKernelGGL((cat_rows_fwd_kernel<W>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, log_qy, log_py, \
                                        py_rows, py_cols, R, k, rows_out, partials); } while (0)
    DISPATCH_W(k, CALL);
#undef CALL
    LAED_LAUNCH_CHECK("cat_rows_fwd");
    hipLaunchKernelGGL(cat_rows_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, nblk, scale, out);
    LAED_LAUNCH_CHECK("cat_rows_combine");
    return 0;
}

extern "C" int laed_cat_rows_bwd(const float* log_qy, const float* log_py, int64_t py_rows, int py_cols, int64_t R, int k,
                                 const float* gscale, float scale, float* dlog_qy, float* dlog_py, void* stream) {
    LAED_CHECK_ARG(R > 0 && k > 0 && k <= 64, "cat_rows_bwd: need R > 0 and 0 < k <= 64 (got R=%lld k=%d)", (long long)R, k);
    LAED_CHECK_ARG(log_qy && (dlog_qy || dlog_py), "cat_rows_bwd: null pointer");
    LAED_CHECK_ARG(cat_rows_py_ok(log_py, py_rows, py_cols, R, k), "cat_rows_bwd: log_py [%lld,%d] does not broadcast to [%lld,%d]",
                   (long long)py_rows, py_cols, (long long)R, k);
    LAED_CHECK_ARG(dlog_py == nullptr || (log_py && py_rows == R && py_cols == k),
                   "cat_rows_bwd: a gradient for log_py needs a full-shape [R,k] log_py");
    const int64_t n = R * k;
    const int nblk = (int)(cdiv64(n, 256) < 1024 ? cdiv64(n, 256) : 1024);
    hipLaunchKernelGGL(cat_rows_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, log_qy, log_py, py_rows, py_cols, R, k,
                       gscale, scale, dlog_qy, dlog_py);
    LAED_LAUNCH_CHECK("cat_rows_bwd");
    return 0;
}

extern "C" int laed_latent_mi_coef(const float* stats, double B_global, int y, int k, float* coef, void* stream) {
    LAED_CHECK_ARG(B_global > 0 && y > 0 && k > 0, "latent_mi_coef: bad size");
    LAED_CHECK_ARG(stats && coef, "latent_mi_coef: null pointer");
    const int n = y * k;
    hipLaunchKernelGGL(mi_coef_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, (float)B_global, y, n, coef);
    LAED_LAUNCH_CHECK("latent_mi_coef");
    return 0;
}

extern "C" int laed_avg_log_fwd(const float* stats, double B_global, int n, float* avg_log, void* stream) {
    LAED_CHECK_ARG(B_global > 0 && n > 0, "avg_log_fwd: bad size");
    LAED_CHECK_ARG(stats && avg_log, "avg_log_fwd: null pointer");
    hipLaunchKernelGGL(avg_log_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, (float)B_global, n, avg_log);
    LAED_LAUNCH_CHECK("avg_log_fwd");
    return 0;
}

extern "C" int laed_avg_log_coef(const float* stats, const float* davg_log, double B_global, int n, float* coef, void* stream) {
    LAED_CHECK_ARG(B_global > 0 && n > 0, "avg_log_coef: bad size");
    LAED_CHECK_ARG(stats && davg_log && coef, "avg_log_coef: null pointer");
    hipLaunchKernelGGL(avg_log_coef_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, davg_log, (float)B_global, n, coef);
    LAED_LAUNCH_CHECK("avg_log_coef");
    return 0;
}
