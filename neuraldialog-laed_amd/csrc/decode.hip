// Greedy free-running decode (DecoderRNN GEN/greedy, decoders.py:180-192,258-268): max_len steps of
//   embed(ids) -> x W_ih^T + b_ih -> fused GRU step -> h W_proj^T + b -> argmax (first max)
// all enqueued from one host call; log_softmax is monotone, so topk(1) of the log-probs (decoders.py:219)
// equals the argmax of the logits and the [B][V] log-prob tensor is never formed.
// gen_type='sample' (decoders.py:220-221 -> gumbel_max, base_modules.py:54-67) replaces the argmax by
//   argmax_v (log_softmax(logits)[v] + g[v]),  g = -log(-log(u)),  u ~ U[0,1)
// with u either replayed from a caller buffer (parity tests) or drawn from the library's counter RNG.
#include "laed_common.h"
#include <math.h>

namespace {
__global__ void fill_ids_kernel(int64_t* ids, int n, int64_t v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ids[i] = v;
}

// one block per row: lse of the row, then first-max of (x - lse) + gumbel(u)
__global__ __launch_bounds__(256) void gumbel_argmax_rows_kernel(const float* __restrict__ L, int rows, int V, int64_t ldl,
                                                                 const float* __restrict__ u, uint64_t seed, uint64_t idx0,
                                                                 int64_t* __restrict__ ids) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* Lr = L + (int64_t)row * ldl;
        float m = -INFINITY;
        for (int i = tid; i < V; i += 256) m = fmaxf(m, Lr[i]);
        m = wave_max(m);
        if (lane == 0) sv[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        __syncthreads();
        float s = 0.f;
        for (int i = tid; i < V; i += 256) s += expf(Lr[i] - m);
        s = wave_sum(s);
        if (lane == 0) sv[wave] = s;
        __syncthreads();
        const float lse = m + logf((sv[0] + sv[1]) + (sv[2] + sv[3]));
        __syncthreads();
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < V; i += 256) {
            const float ui = u ? u[(int64_t)row * V + i] : uniform01(seed, idx0 + (uint64_t)row * V + i);
            const float x = (Lr[i] - lse) + (-logf(-logf(ui)));
            if (x > bv) { bv = x; bi = i; }
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
            ids[row] = bi == 0x7fffffff ? 0 : bi;
        }
        __syncthreads();
    }
}

int decode_loop(const float* emb, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                const float* proj_w, const float* proj_b, const float* h0, int B, int E, int H, int V, int max_len,
                int64_t go_id, bool sample, const float* u, uint64_t seed, uint64_t row0,
                float* hbuf, float* xbuf, float* gates, float* logits, int64_t* ids_out, hipStream_t stream);
}   // namespace

namespace {
int decode_loop(const float* emb, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                const float* proj_w, const float* proj_b, const float* h0, int B, int E, int H, int V, int max_len,
                int64_t go_id, bool sample, const float* u, uint64_t seed, uint64_t row0,
                float* hbuf, float* xbuf, float* gates, float* logits, int64_t* ids_out, hipStream_t stream) {
    void* stream_ = (void*)stream;
    LAED_CHECK_ARG(B > 0 && E > 0 && H > 0 && V > 0 && max_len > 0, "gru_decode: bad size");
    LAED_CHECK_ARG(emb && w_ih && w_hh && b_ih && b_hh && proj_w && h0 && hbuf && xbuf && gates && logits && ids_out,
                   "gru_decode: null pointer");
    LAED_CHECK_ARG(go_id >= 0 && go_id < V, "gru_decode: go_id out of range");
    const int64_t BH = (int64_t)B * H;
    const int ldl = laed_proj_nll_ldl(V);
    if (h0 != hbuf) {
        hipError_t e = hipMemcpyAsync(hbuf, h0, BH * sizeof(float), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return laed_fail((int)e, "gru_decode: h0 copy failed");
    }
    hipLaunchKernelGGL(fill_ids_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, ids_out, B, go_id);
    LAED_LAUNCH_CHECK("fill_ids");
    // exact-fp32 mode with at least a round of 128x128 tiles: the arg-max comes out of the GEMM (2 * ceil(V/128) <= ldl,
    // so the partials fit the scratch); smaller problems and the split-bf16 mode keep the two-kernel path
    const bool fused_argmax = laed_get_gemm_mode() == 0 && (int64_t)((B + 127) / 128) * ((V + 127) / 128) >= 512 &&
                              2 * ((V + 127) / 128) <= ldl;
    for (int t = 0; t < max_len; ++t) {
        const int64_t* ids = ids_out + (int64_t)t * B;
        LAED_TRY(laed_embed_gather(ids, B, emb, V, E, nullptr, 0.f, xbuf, stream_));
        LAED_TRY(laed_sgemm_impl(false, true, B, 3 * H, E, xbuf, E, w_ih, E, gates, 3 * H, b_ih, false, nullptr, false, stream));
        LAED_TRY(laed_gru_step_impl(hbuf + (int64_t)t * BH, w_hh, b_hh, gates, hbuf + (int64_t)(t + 1) * BH, nullptr, B, H, stream));
        if (!sample && fused_argmax) {
            // greedy: per-tile (max, column) from the projection's epilogue, reduced over the tiles -- the [B][V] logits
            // are never written; the scratch holds the [tiles][B][2] partials instead
            LAED_TRY(laed_sgemm_argmax_impl(B, V, H, hbuf + (int64_t)(t + 1) * BH, H, proj_w, H, proj_b, logits,
                                            ids_out + (int64_t)(t + 1) * B, stream));
            continue;
        }
        LAED_TRY(laed_sgemm_impl(false, true, B, V, H, hbuf + (int64_t)(t + 1) * BH, H, proj_w, H, logits, ldl, proj_b, false,
                                 nullptr, false, stream));
        if (!sample) {
            LAED_TRY(laed_argmax_rows(logits, B, V, ldl, ids_out + (int64_t)(t + 1) * B, stream_));
        } else {
            // RNG stream index of element (step t, global row r, v): ((t * 2^32) + row0 + r) * V + v
            const uint64_t idx0 = (((uint64_t)t << 32) + row0) * (uint64_t)V;
            hipLaunchKernelGGL(gumbel_argmax_rows_kernel, dim3(B < 4096 ? B : 4096), dim3(256), 0, stream, logits, B, V,
                               (int64_t)ldl, u ? u + (int64_t)t * B * V : nullptr, seed, idx0, ids_out + (int64_t)(t + 1) * B);
            LAED_LAUNCH_CHECK("gumbel_argmax_rows");
        }
    }
    return 0;
}
}   // namespace

extern "C" int laed_gru_decode_greedy(const float* emb, const float* w_ih, const float* w_hh, const float* b_ih,
                                      const float* b_hh, const float* proj_w, const float* proj_b, const float* h0,
                                      int B, int E, int H, int V, int max_len, int64_t go_id,
                                      float* hbuf, float* xbuf, float* gates, float* logits, int64_t* ids_out,
                                      void* stream) {
    return decode_loop(emb, w_ih, w_hh, b_ih, b_hh, proj_w, proj_b, h0, B, E, H, V, max_len, go_id, false, nullptr, 0, 0,
                       hbuf, xbuf, gates, logits, ids_out, (hipStream_t)stream);
}

extern "C" int laed_gru_decode_sample(const float* emb, const float* w_ih, const float* w_hh, const float* b_ih,
                                      const float* b_hh, const float* proj_w, const float* proj_b, const float* h0,
                                      int B, int E, int H, int V, int max_len, int64_t go_id,
                                      const float* u, uint64_t seed, uint64_t row0,
                                      float* hbuf, float* xbuf, float* gates, float* logits, int64_t* ids_out,
                                      void* stream) {
    return decode_loop(emb, w_ih, w_hh, b_ih, b_hh, proj_w, proj_b, h0, B, E, H, V, max_len, go_id, true, u, seed, row0,
                       hbuf, xbuf, gates, logits, ids_out, (hipStream_t)stream);
}

// ---- one decode step at a time (laed_gru_cell_fwd, laed_proj_argmax): the two halves of the loop above for
// callers that drive their own loop (serving, beam-style search on the host)
extern "C" int laed_gru_cell_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                 const float* hprev, int B, int E, int H, float* gates, float* hout, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAED_CHECK_ARG(B > 0 && E > 0 && H > 0, "gru_cell_fwd: bad size");
    LAED_CHECK_ARG(x && w_ih && w_hh && b_ih && b_hh && hprev && gates && hout, "gru_cell_fwd: null pointer");
    LAED_CHECK_ARG(hout != hprev, "gru_cell_fwd: hout must not alias hprev (blocks read other rows' h while writing theirs)");
    LAED_TRY(laed_sgemm_impl(false, true, B, 3 * H, E, x, E, w_ih, E, gates, 3 * H, b_ih, false, nullptr, false, stream));
    return laed_gru_step_impl(hprev, w_hh, b_hh, gates, hout, nullptr, B, H, stream);
}

extern "C" int laed_proj_argmax(const float* h, const float* proj_w, const float* proj_b, int B, int H, int V,
                                float* logits, int64_t* ids, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAED_CHECK_ARG(B > 0 && H > 0 && V > 0, "proj_argmax: bad size");
    LAED_CHECK_ARG(h && proj_w && logits && ids, "proj_argmax: null pointer");
    const int ldl = laed_proj_nll_ldl(V);
    LAED_TRY(laed_sgemm_impl(false, true, B, V, H, h, H, proj_w, H, logits, ldl, proj_b, false, nullptr, false, stream));
    return laed_argmax_rows(logits, B, V, ldl, ids, stream_);
}
