// Greedy free-running decode (DecoderRNN GEN/greedy, decoders.py:180-192,258-268): max_len steps of
//   embed(ids) -> x W_ih^T + b_ih -> fused GRU step -> h W_proj^T + b -> argmax (first max)
// all enqueued from one host call; log_softmax is monotone, so topk(1) of the log-probs (decoders.py:219)
// equals the argmax of the logits and the [B][V] log-prob tensor is never formed.
#include "laed_common.h"

namespace {
__global__ void fill_ids_kernel(int64_t* ids, int n, int64_t v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ids[i] = v;
}
}   // namespace

extern "C" int laed_gru_decode_greedy(const float* emb, const float* w_ih, const float* w_hh, const float* b_ih,
                                      const float* b_hh, const float* proj_w, const float* proj_b, const float* h0,
                                      int B, int E, int H, int V, int max_len, int64_t go_id,
                                      float* hbuf, float* xbuf, float* gates, float* logits, int64_t* ids_out,
                                      void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAED_CHECK_ARG(B > 0 && E > 0 && H > 0 && V > 0 && max_len > 0, "gru_decode_greedy: bad size");
    LAED_CHECK_ARG(emb && w_ih && w_hh && b_ih && b_hh && proj_w && h0 && hbuf && xbuf && gates && logits && ids_out,
                   "gru_decode_greedy: null pointer");
    LAED_CHECK_ARG(go_id >= 0 && go_id < V, "gru_decode_greedy: go_id out of range");
    const int64_t BH = (int64_t)B * H;
    const int ldl = laed_proj_nll_ldl(V);
    if (h0 != hbuf) {
        hipError_t e = hipMemcpyAsync(hbuf, h0, BH * sizeof(float), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return laed_fail((int)e, "gru_decode_greedy: h0 copy failed");
    }
    hipLaunchKernelGGL(fill_ids_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, ids_out, B, go_id);
    LAED_LAUNCH_CHECK("fill_ids");
    for (int t = 0; t < max_len; ++t) {
        const int64_t* ids = ids_out + (int64_t)t * B;
        LAED_TRY(laed_embed_gather(ids, B, emb, V, E, nullptr, 0.f, xbuf, stream_));
        LAED_TRY(laed_sgemm_impl(false, true, B, 3 * H, E, xbuf, E, w_ih, E, gates, 3 * H, b_ih, false, nullptr, false, stream));
        LAED_TRY(laed_gru_step_impl(hbuf + (int64_t)t * BH, w_hh, b_hh, gates, hbuf + (int64_t)(t + 1) * BH, nullptr, B, H, stream));
        LAED_TRY(laed_sgemm_impl(false, true, B, V, H, hbuf + (int64_t)(t + 1) * BH, H, proj_w, H, logits, ldl, proj_b, false,
                                 nullptr, false, stream));
        LAED_TRY(laed_argmax_rows(logits, B, V, ldl, ids_out + (int64_t)(t + 1) * B, stream_));
    }
    return 0;
}
