// Expected embedding of the decoder's output distribution ("attribute forcing", AE-ED / ST-ED):
//
//   out[N][E] = softmax(h W^T + b) Wemb            h [N][H], W [V][H], b [V], Wemb [V][E]
//
// replaces  torch.exp(dec_outs.view(-1, V)) @ x_embedding.weight  (laed/models/dialog_models.py:421-422, :726-727),
// which needs the [N][V] probability tensor the fused projection+NLL path never forms.  Here too it exists only as
// a [chunk_rows][ldl] workspace tile:
//   forward, per row chunk:   L = h_c W^T + b (GEMM) -> rows: lse, P = exp(L - lse) in place -> out_c = P Wemb (GEMM)
//   backward, per row chunk:  L recomputed (GEMM) -> P = exp(L - lse)             (lse saved by forward)
//                             dWemb += P^T dout_c (GEMM);  dP = dout_c Wemb^T (GEMM, second tile)
//                             dL = P * (dP - <P, dP>_row) in place -> dh_c = dL W;  dW += dL^T h_c;  db += colsum(dL)
// The gradient of this head only arrives after the backward of the recognition network it feeds, so unlike
// laed_proj_nll the logits are recomputed in backward (one extra 2NHV GEMM) instead of being kept.
#include "laed_common.h"
#include <math.h>

namespace {

// P = exp(L - lse) in place; lse computed here (HAVE_LSE = false) or given.  One block per row, float4 accesses.
template <bool HAVE_LSE>
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ L, int64_t rows, int V, int64_t ldl,
                                                           float* __restrict__ lse_io) {
    __shared__ float sm[4], ss[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V4 = V >> 2;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        float* Lr = L + row * ldl;
        float4* Lr4 = reinterpret_cast<float4*>(Lr);
        float lse;
        if (HAVE_LSE) {
            lse = lse_io[row];
        } else {
            float m = -INFINITY;
            for (int i = tid; i < V4; i += 256) {
                const float4 x = Lr4[i];
                m = fmaxf(m, fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w)));
            }
            for (int i = V4 * 4 + tid; i < V; i += 256) m = fmaxf(m, Lr[i]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            if (lane == 0) sm[wave] = m;
            __syncthreads();
            m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
            float s = 0.f;
            for (int i = tid; i < V4; i += 256) {
                const float4 x = Lr4[i];
                s += (expf(x.x - m) + expf(x.y - m)) + (expf(x.z - m) + expf(x.w - m));
            }
            for (int i = V4 * 4 + tid; i < V; i += 256) s += expf(Lr[i] - m);
            s = wave_sum(s);
            if (lane == 0) ss[wave] = s;
            __syncthreads();
            lse = m + logf((ss[0] + ss[1]) + (ss[2] + ss[3]));
            if (tid == 0) lse_io[row] = lse;
        }
        for (int i = tid; i < V4; i += 256) {
            float4 x = Lr4[i];
            x.x = expf(x.x - lse); x.y = expf(x.y - lse); x.z = expf(x.z - lse); x.w = expf(x.w - lse);
            Lr4[i] = x;
        }
        for (int i = V4 * 4 + tid; i < V; i += 256) Lr[i] = expf(Lr[i] - lse);
        __syncthreads();   // sm/ss reused by the next row
    }
}

// P <- P * (dP - sum_v P dP)   (softmax backward), row by row
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(float* __restrict__ P, const float* __restrict__ dP, int64_t rows,
                                                               int V, int64_t ldl) {
    __shared__ float ss[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V4 = V >> 2;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        float* Pr = P + row * ldl;
        const float* Dr = dP + row * ldl;
        float4* Pr4 = reinterpret_cast<float4*>(Pr);
        const float4* Dr4 = reinterpret_cast<const float4*>(Dr);
        float s = 0.f;
        for (int i = tid; i < V4; i += 256) {
            const float4 p = Pr4[i], d = Dr4[i];
            s += (p.x * d.x + p.y * d.y) + (p.z * d.z + p.w * d.w);
        }
        for (int i = V4 * 4 + tid; i < V; i += 256) s += Pr[i] * Dr[i];
        s = wave_sum(s);
        if (lane == 0) ss[wave] = s;
        __syncthreads();
        const float dot = (ss[0] + ss[1]) + (ss[2] + ss[3]);
        for (int i = tid; i < V4; i += 256) {
            float4 p = Pr4[i];
            const float4 d = Dr4[i];
            p.x *= d.x - dot; p.y *= d.y - dot; p.z *= d.z - dot; p.w *= d.w - dot;
            Pr4[i] = p;
        }
        for (int i = V4 * 4 + tid; i < V; i += 256) Pr[i] *= Dr[i] - dot;
        __syncthreads();
    }
}

inline int rows_grid(int64_t rows) { return (int)(rows < 4096 ? (rows < 1 ? 1 : rows) : 4096); }

}   // namespace

// floats of workspace: forward needs one [chunk_rows][ldl] tile, backward two; whatever the caller passes beyond the
// tiles holds split-K partials of the products whose output is small and whose K is long (P Wemb and dh = dP W: 76-80
// blocks of 64x64 with K = V at the reference's batch of 30 -- 340 / 166 us unsplit).  This function asks for up to one
// more tile (capped at 256 MB) for them; a call that passes the bare tiles still works, unsplit.
extern "C" int64_t laed_softmax_embed_workspace(int V, int chunk_rows, int backward) {
    const int64_t ldl = laed_proj_nll_ldl(V);
    const int64_t tile = (int64_t)chunk_rows * ldl, cap = (int64_t)1 << 26;
    return tile * (backward ? 2 : 1) + (tile < cap ? tile : cap);
}

extern "C" int laed_softmax_embed_fwd(const float* h, const float* W, const float* b, const float* Wemb,
                                      int64_t N, int H, int V, int E, float* workspace, int64_t ws_floats, int chunk_rows,
                                      float* lse, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAED_CHECK_ARG(N >= 0 && H > 0 && V > 0 && E > 0 && chunk_rows > 0, "softmax_embed_fwd: bad size");
    if (N == 0) return 0;
    LAED_CHECK_ARG(h && W && Wemb && workspace && lse && out, "softmax_embed_fwd: null pointer");
    LAED_CHECK_ARG(aligned16(workspace), "softmax_embed_fwd: workspace must be 16-byte aligned");
    const int64_t ldl = laed_proj_nll_ldl(V);
    LAED_CHECK_ARG(ws_floats >= (int64_t)chunk_rows * ldl, "softmax_embed_fwd: workspace too small");
    float* sk = workspace + (int64_t)chunk_rows * ldl;               // split-K partials (may be empty)
    const int64_t sk_floats = ws_floats - (int64_t)chunk_rows * ldl;
    for (int64_t r0 = 0; r0 < N; r0 += chunk_rows) {
        const int rows = (int)((N - r0) < chunk_rows ? (N - r0) : chunk_rows);
        LAED_TRY(laed_sgemm_impl(false, true, rows, V, H, h + r0 * H, H, W, H, workspace, ldl, b, false, nullptr, false, stream));
        hipLaunchKernelGGL((softmax_rows_kernel<false>), dim3(rows_grid(rows)), dim3(256), 0, stream, workspace, (int64_t)rows, V,
                           ldl, lse + r0);
        LAED_LAUNCH_CHECK("softmax_rows");
        LAED_TRY(laed_sgemm_splitk_impl(false, false, rows, E, V, workspace, ldl, Wemb, E, out + r0 * E, E, nullptr, false, nullptr,
                                        false, sk_floats > 0 ? sk : nullptr, sk_floats, false, nullptr, stream));
    }
    return 0;
}

extern "C" int laed_softmax_embed_bwd(const float* h, const float* W, const float* b, const float* Wemb, const float* lse,
                                      const float* dout, int64_t N, int H, int V, int E,
                                      float* workspace, int64_t ws_floats, int chunk_rows,
                                      float* dh, float* dW, float* db, float* dWemb, int accumulate_w, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAED_CHECK_ARG(N >= 0 && H > 0 && V > 0 && E > 0 && chunk_rows > 0, "softmax_embed_bwd: bad size");
    if (N == 0) return 0;
    LAED_CHECK_ARG(h && W && Wemb && lse && dout && workspace && dh && dW && db && dWemb, "softmax_embed_bwd: null pointer");
    LAED_CHECK_ARG(aligned16(workspace), "softmax_embed_bwd: workspace must be 16-byte aligned");
    const int64_t ldl = laed_proj_nll_ldl(V);
    LAED_CHECK_ARG(ws_floats >= 2 * (int64_t)chunk_rows * ldl, "softmax_embed_bwd: workspace too small");
    float* P = workspace;
    float* dP = workspace + (int64_t)chunk_rows * ldl;
    const int64_t sk_floats = ws_floats - 2 * (int64_t)chunk_rows * ldl;
    float* sk = sk_floats > 0 ? workspace + 2 * (int64_t)chunk_rows * ldl : nullptr;    // split-K partials, consumed in stream order
    bool acc = accumulate_w != 0;
    for (int64_t r0 = 0; r0 < N; r0 += chunk_rows) {
        const int rows = (int)((N - r0) < chunk_rows ? (N - r0) : chunk_rows);
        const float* hc = h + r0 * H;
        const float* dc = dout + r0 * E;
        LAED_TRY(laed_sgemm_impl(false, true, rows, V, H, hc, H, W, H, P, ldl, b, false, nullptr, false, stream));
        hipLaunchKernelGGL((softmax_rows_kernel<true>), dim3(rows_grid(rows)), dim3(256), 0, stream, P, (int64_t)rows, V, ldl,
                           const_cast<float*>(lse) + r0);
        LAED_LAUNCH_CHECK("softmax_rows");
        // dWemb[V][E] (+)= P^T[V][rows] dout_c[rows][E]
        LAED_TRY(laed_sgemm_splitk_impl(true, false, V, E, rows, P, ldl, dc, E, dWemb, E, nullptr, acc, nullptr, false, sk, sk_floats,
                                        false, nullptr, stream));
        // dP[rows][V] = dout_c[rows][E] Wemb^T   (Wemb [V][E] is the K-contiguous "W" of an x W^T product)
        LAED_TRY(laed_sgemm_impl(false, true, rows, V, E, dc, E, Wemb, E, dP, ldl, nullptr, false, nullptr, false, stream));
        hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3(rows_grid(rows)), dim3(256), 0, stream, P, dP, (int64_t)rows, V, ldl);
        LAED_LAUNCH_CHECK("softmax_bwd_rows");
        LAED_TRY(laed_sgemm_splitk_impl(false, false, rows, H, V, P, ldl, W, H, dh + r0 * H, H, nullptr, false, nullptr, false, sk,
                                        sk_floats, false, nullptr, stream));
        LAED_TRY(laed_sgemm_splitk_impl(true, false, V, H, rows, P, ldl, hc, H, dW, H, nullptr, acc, db, acc, sk, sk_floats, false,
                                        nullptr, stream));
        acc = true;
    }
    return 0;
}
