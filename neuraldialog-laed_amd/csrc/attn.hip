// Attention pooling of RnnUttEncoder (encoders.py:119-130 of the reference) and the tanh of the prior network
// (dialog_models.py p_fc1): small HBM / latency-bound kernels around the two Linear layers, which stay on the GEMM.
//
//   fc1   = tanh(key_w(enc))            [R][Hk]   R = N*T rows            (pre-activation comes from the GEMM)
//   score = fc1 . wq + bq               [R]       one wavefront per row, __shfl_xor reduction
//   attn  = softmax_t(score[n][:])      [N][T]
//   out   = sum_t attn[n][t] enc[n][t]  [N][D]    one block per utterance, lanes over D (coalesced), t in order
//
// enc / score / gradients are addressed with explicit (utterance, position) strides, so the time-major memory the
// GRU produces is consumed without a transposing copy.
#include "laed_common.h"
#include <math.h>

namespace {

constexpr int ATTN_MAXT = 1024;

__global__ __launch_bounds__(256) void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = tanhf(x[i]);
}

__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                       float* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = y[i];
        dx[i] = dy[i] * (1.0f - v * v);
    }
}

// fc1 (in/out): pre-activation -> tanh, score[row] = fc1[row] . wq + bq
__global__ __launch_bounds__(256) void attn_score_fwd_kernel(float* __restrict__ fc1, const float* __restrict__ wq,
                                                             const float* __restrict__ bq, int64_t R, int Hk,
                                                             float* __restrict__ score) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;                              // wave-uniform
    float* p = fc1 + row * Hk;
    float acc = 0.f;
    for (int j = lane; j < Hk; j += 64) {
        const float v = tanhf(p[j]);
        p[j] = v;
        acc += v * wq[j];
    }
    acc = wave_sum(acc);
    if (lane == 0) score[row] = acc + (bq ? bq[0] : 0.f);
}

__global__ __launch_bounds__(256) void attn_score_bwd_kernel(const float* __restrict__ fc1, const float* __restrict__ dscore,
                                                             const float* __restrict__ wq, int64_t R, int Hk,
                                                             float* __restrict__ dpre) {
    const int64_t n = R * Hk;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / Hk;
        const int j = (int)(i - row * Hk);
        const float v = fc1[i];
        dpre[i] = dscore[row] * wq[j] * (1.0f - v * v);
    }
}

// one block per utterance n: softmax over its T scores, then the weighted sum of its T encoder states
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(const float* __restrict__ score, int64_t s_sn, int64_t s_st,
                                                            const float* __restrict__ enc, int64_t e_sn, int64_t e_st,
                                                            int T, int D, float* __restrict__ attn, float* __restrict__ out) {
    __shared__ float a[ATTN_MAXT];
    const int64_t n = blockIdx.x;
    for (int t = threadIdx.x; t < T; t += 256) a[t] = score[n * s_sn + t * s_st];
    __syncthreads();
    float m = -INFINITY;
    for (int t = 0; t < T; ++t) m = fmaxf(m, a[t]);    // every thread: T is a sentence length
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += expf(a[t] - m);
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
        const float w = expf(a[t] - m) / s;
        a[t] = w;
        attn[n * T + t] = w;
    }
    __syncthreads();
    const float* e = enc + n * e_sn;
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc += a[t] * e[t * e_st + d];
        out[n * D + d] = acc;
    }
}

// dattn[t] = dout[n] . enc[n][t];  dscore[n][t] = attn (dattn - sum_t attn dattn);  denc[n][t] = attn[t] dout[n]
__global__ __launch_bounds__(256) void attn_pool_bwd_kernel(const float* __restrict__ enc, int64_t e_sn, int64_t e_st,
                                                            const float* __restrict__ attn, const float* __restrict__ dout,
                                                            int T, int D, float* __restrict__ dscore, int64_t s_sn, int64_t s_st,
                                                            float* __restrict__ denc, int64_t g_sn, int64_t g_st) {
    __shared__ float a[ATTN_MAXT], da[ATTN_MAXT];
    const int64_t n = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* e = enc + n * e_sn;
    const float* g = dout + n * D;
    for (int t = threadIdx.x; t < T; t += 256) a[t] = attn[n * T + t];
    for (int t = wave; t < T; t += 4) {
        float acc = 0.f;
        for (int d = lane; d < D; d += 64) acc += g[d] * e[t * e_st + d];
        acc = wave_sum(acc);
        if (lane == 0) da[t] = acc;
    }
    __syncthreads();
    float dot = 0.f;
    for (int t = 0; t < T; ++t) dot += a[t] * da[t];
    for (int t = threadIdx.x; t < T; t += 256) dscore[n * s_sn + t * s_st] = a[t] * (da[t] - dot);
    if (denc != nullptr) {
        float* ge = denc + n * g_sn;
        for (int d = threadIdx.x; d < D; d += 256) {
            const float gv = g[d];
            for (int t = 0; t < T; ++t) ge[t * g_st + d] = a[t] * gv;
        }
    }
}

inline int grid_for(int64_t n) {
    int64_t b = cdiv64(n, 256);
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}   // namespace

extern "C" int laed_tanh_fwd(const float* x, float* y, int64_t n, void* stream) {
    LAED_CHECK_ARG(n >= 0, "tanh_fwd: bad size");
    if (n == 0) return 0;
    LAED_CHECK_ARG(x && y, "tanh_fwd: null pointer");
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    LAED_LAUNCH_CHECK("tanh_fwd");
    return 0;
}

extern "C" int laed_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream) {
    LAED_CHECK_ARG(n >= 0, "tanh_bwd: bad size");
    if (n == 0) return 0;
    LAED_CHECK_ARG(y && dy && dx, "tanh_bwd: null pointer");
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, n);
    LAED_LAUNCH_CHECK("tanh_bwd");
    return 0;
}

extern "C" int laed_attn_score_fwd(float* fc1, const float* wq, const float* bq, int64_t R, int Hk, float* score, void* stream) {
    LAED_CHECK_ARG(R >= 0 && Hk > 0, "attn_score_fwd: bad size");
    if (R == 0) return 0;
    LAED_CHECK_ARG(fc1 && wq && score, "attn_score_fwd: null pointer");
    hipLaunchKernelGGL(attn_score_fwd_kernel, dim3((unsigned)cdiv64(R, 4)), dim3(256), 0, (hipStream_t)stream, fc1, wq, bq, R, Hk, score);
    LAED_LAUNCH_CHECK("attn_score_fwd");
    return 0;
}

extern "C" int laed_attn_score_bwd(const float* fc1, const float* dscore, const float* wq, int64_t R, int Hk, float* dpre,
                                   void* stream) {
    LAED_CHECK_ARG(R >= 0 && Hk > 0, "attn_score_bwd: bad size");
    if (R == 0) return 0;
    LAED_CHECK_ARG(fc1 && dscore && wq && dpre, "attn_score_bwd: null pointer");
    hipLaunchKernelGGL(attn_score_bwd_kernel, dim3(grid_for(R * Hk)), dim3(256), 0, (hipStream_t)stream, fc1, dscore, wq, R, Hk, dpre);
    LAED_LAUNCH_CHECK("attn_score_bwd");
    return 0;
}

extern "C" int laed_attn_pool_fwd(const float* score, int64_t s_sn, int64_t s_st, const float* enc, int64_t e_sn, int64_t e_st,
                                  int64_t N, int T, int D, float* attn, float* out, void* stream) {
    LAED_CHECK_ARG(N >= 0 && T > 0 && T <= ATTN_MAXT && D > 0, "attn_pool_fwd: need 0 < T <= %d and D > 0 (got T=%d D=%d)",
                   ATTN_MAXT, T, D);
    if (N == 0) return 0;
    LAED_CHECK_ARG(score && enc && attn && out, "attn_pool_fwd: null pointer");
    hipLaunchKernelGGL(attn_pool_fwd_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, score, s_sn, s_st, enc, e_sn, e_st,
                       T, D, attn, out);
    LAED_LAUNCH_CHECK("attn_pool_fwd");
    return 0;
}

extern "C" int laed_attn_pool_bwd(const float* enc, int64_t e_sn, int64_t e_st, const float* attn, const float* dout,
                                  int64_t N, int T, int D, float* dscore, int64_t s_sn, int64_t s_st,
                                  float* denc, int64_t g_sn, int64_t g_st, void* stream) {
    LAED_CHECK_ARG(N >= 0 && T > 0 && T <= ATTN_MAXT && D > 0, "attn_pool_bwd: need 0 < T <= %d and D > 0 (got T=%d D=%d)",
                   ATTN_MAXT, T, D);
    if (N == 0) return 0;
    LAED_CHECK_ARG(enc && attn && dout && dscore, "attn_pool_bwd: null pointer");
    hipLaunchKernelGGL(attn_pool_bwd_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, enc, e_sn, e_st, attn, dout,
                       T, D, dscore, s_sn, s_st, denc, g_sn, g_st);
    LAED_LAUNCH_CHECK("attn_pool_bwd");
    return 0;
}
