/*
 * laed_hip.h -- C ABI of liblaed_hip.so: the MI355X (gfx950) kernels behind the DI-VAE / DI-VST
 * train step of snakeztc/NeuralDialog-LAED.
 *
 * The reference has NO native boundary on this path: every op below is a stock torch.nn call made
 * from Python.  Each entry point therefore cites the reference Python call site whose arithmetic it
 * replaces (paths relative to the reference root).  A maintainer binds these with ctypes (see
 * INTEGRATION.md); the in-tree binding is neuraldialog-laed_amd/laed_amd/_lib.py.
 *
 * Conventions
 *   - all matrices are dense row-major fp32; token ids / labels are int64 (model_bases.py:53-57).
 *   - sequence tensors are TIME-MAJOR inside the library: [T][B][*] (flat row = t*B + b).
 *   - every pointer is a DEVICE pointer owned by the caller (the PyTorch caching allocator);
 *     the library allocates nothing and never synchronises the device.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *   - return value: 0 = ok, <0 = bad argument (LAED_E_*), >0 = hipError_t of a failed launch.
 *     laed_last_error() returns a thread-local description of the last non-zero status.
 *   - reentrant across streams and threads.  Process-wide state is limited to: the GEMM math mode
 *     (laed_set_gemm_mode, an atomic; default exact fp32), the opt-in GEMM profiler (laed_prof_*), per-device cached
 *     device attributes, the allocation-free entry point's scratch array and the sticky timeout counter of the
 *     persistent GRU kernels.  The product library reads four environment switches, once each:
 *     LAED_GEMM_MODE, LAED_GRU_SMALL, LAED_GRU_PERSIST, LAED_GRU_WM (test hook); every other tuning override exists
 *     only in the hooks build (make HOOKS=1).
 *   - third-party device code on the path: rocPRIM's radix sort inside laed_embed_scatter_add (the deterministic
 *     scatter-add sorts (id, position) pairs with it); everything else is this library's own kernels.
 */
#ifndef LAED_HIP_H
#define LAED_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAED_E_BADARG (-1)
#define LAED_E_ALIGN  (-2)
#define LAED_E_SIZE   (-3)

int         laed_version(void);
const char* laed_last_error(void);

/* ------------------------------------------------------------------------------------------
 * fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32):  C[M,N] = op(A)[M,K] * op(B)[K,N] (+ bias[N]) (+ C if accumulate)
 *   transA = 0: A stored [M][K] (lda >= K);  transA = 1: A stored [K][M] (lda >= M)
 *   transB = 0: B stored [K][N] (ldb >= N);  transB = 1: B stored [N][K] (ldb >= K)
 *   colsum_a (optional, needs transA = 1): colsum_a[m] (+)= sum_k A[k][m]   (bias gradients)
 * Replaces: nn.Linear forward  x W^T + b  (sent_models.py:53,110; nn_lib.py:145,158; decoders.py:149,165)
 * and the two autograd GEMMs of its backward (dx = dy W, dW = dy^T x, db = colsum dy).
 * ------------------------------------------------------------------------------------------ */
int laed_sgemm(int transA, int transB, int M, int N, int K,
               const float* A, int lda, const float* B, int ldb,
               float* C, int ldc, const float* bias, int accumulate,
               float* colsum_a, int colsum_accumulate, void* stream);

/* Same product with a caller workspace (ws_floats floats, may be NULL): problems with few output tiles and a long K
 * are split over K so that all 256 CUs work; partial tiles are summed in a fixed order (bitwise reproducible). */
int laed_sgemm_ws(int transA, int transB, int M, int N, int K,
                  const float* A, int lda, const float* B, int ldb,
                  float* C, int ldc, const float* bias, int accumulate,
                  float* colsum_a, int colsum_accumulate, float* ws, int64_t ws_floats, void* stream);

/* ------------------------------------------------------------------------------------------
 * Embedding gather / scatter with optional input dropout.
 *   out[r][:] = table[ids[r]][:] * (keep[r][:] ? 1/(1-p) : 0)            (keep == NULL: no dropout)
 *   scatter: dtable[ids[r]][:] += dout[r][:] * (same factor), rows with ids[r] == pad_id skipped
 * Replaces: nn.Embedding(padding_idx=PAD) (sent_models.py:44-45,99; decoders.py:155) and the
 * decoder's nn.Dropout on the embedded inputs (base_modules.py:34; decoders.py:156).
 * laed_dropout_keep_mask fills keep[n] (uint8 0/1) from a counter-based RNG keyed by
 * (seed, offset + element index): results do not depend on how rows are split over GPUs.
 * ------------------------------------------------------------------------------------------ */
int laed_embed_gather(const int64_t* ids, int64_t rows, const float* table, int V, int E,
                      const uint8_t* keep, float drop_p, float* out, void* stream);
/* scatter-add with `ws` (laed_embed_scatter_workspace bytes, 256-byte aligned; not needed for rows <= 4096) is
 * DETERMINISTIC: stable sort of (id, position) + fixed-order segment sums, no float atomics.  ws == NULL for more
 * than 4096 rows falls back to fp32 atomics (order-dependent rounding). */
/* Batch-first ids in, time-major everything out, one launch: out_tm row t*B + b = table[ids[b*ld_ids + t]] (the
 * transposing copy of the ids that Embedding.forward made before the gather, sent_models.py:99 / decoders.py:155, is
 * folded in); ids_tm (may be NULL) receives the transposed ids the backward scatter-add walks.  Input dropout
 * (decoders.py:156): keep_in = an injected time-major mask [T][B][E], or use_rng != 0 = draw (b, t, c) as number
 * rng_offset + (b*T + t)*E + c of stream `seed` (the index laed_dropout_keep_mask gives a batch-first mask); either way
 * keep_out (may be NULL) receives the time-major mask for the backward pass. */
int laed_embed_gather_bt(const int64_t* ids, int64_t ld_ids, int B, int T, const float* table, int V, int E,
                         const uint8_t* keep_in, int use_rng, float drop_p, uint64_t seed, uint64_t rng_offset,
                         float* out_tm, int64_t* ids_tm, uint8_t* keep_out, void* stream);
int64_t laed_embed_scatter_workspace(int64_t rows, int V, int E);
int laed_embed_scatter_add(const int64_t* ids, int64_t rows, const float* dout, int V, int E,
                           const uint8_t* keep, float drop_p, int64_t pad_id, float* dtable,
                           void* ws, int64_t ws_bytes, void* stream);
int laed_dropout_keep_mask(uint8_t* keep, int64_t n, float drop_p, uint64_t seed, uint64_t offset, void* stream);
int laed_uniform(float* u, int64_t n, uint64_t seed, uint64_t offset, void* stream);

/* ------------------------------------------------------------------------------------------
 * GRU, 1 layer, uni-directional, gate order (r,z,n), time-major.
 * Replaces nn.GRU(E,H,1,batch_first=True) built at base_modules.py:43-45 and called at
 * encoders.py:62-64 / decoders.py:158, and its autograd backward.
 *
 *  fwd:  x[T][B][E]  ->  gates = x W_ih^T + b_ih (one MFMA GEMM for all T), then T launches of the fused
 *        recurrent step kernel (h W_hh^T MFMA + gate epilogue).
 *        hbuf [(T+1)][B][H]: hbuf[0] = h0 (copied in; h0 == NULL -> zeros), hbuf[t+1] = h_t.
 *        gates[T][B][3H]   : overwritten with (r,z,n) (reserve for bwd; may be NULL with hn when save=0,
 *                            then a [B][3H] scratch `gates` of one step is still required: pass T=1 per call)
 *        hn   [T][B][H]    : W_hn h + b_hn (reserve for bwd)
 *  bwd:  dout[T][B][H] (may be NULL), dhT[B][H] (may be NULL) ->
 *        dx[T][B][E] (may be NULL), dw_ih[3H][E], dw_hh[3H][H], db_ih[3H], db_hh[3H], dh0[B][H]
 *        gates is overwritten with d(gi); dgh[T][B][3H] is caller-provided workspace.
 *        accumulate_w != 0: weight/bias grads are added to the existing contents.
 *        ws (ws_floats floats, may be NULL): split-K scratch for small batches (see laed_sgemm_ws).
 *  Small batches (B <= 64, H % 16 == 0) with a 16-byte aligned ws of at least laed_gru_*_workspace floats run the
 *  packed-operand step kernels of gru_small.hip (W_hh re-laid once per call in MFMA lane order, activations handed
 *  from step to step in the same order; backward = one fused launch per step); ws == NULL keeps the tiled kernels.
 * ------------------------------------------------------------------------------------------ */
int64_t laed_gru_fwd_workspace(int T, int B, int E, int H);   /* ws_floats laed_gru_fwd can use (0: none) */
int64_t laed_gru_bwd_workspace(int T, int B, int E, int H);   /* recommended ws_floats for laed_gru_bwd */
int laed_gru_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                 const float* h0, int T, int B, int E, int H,
                 float* hbuf, float* gates, float* hn, float* ws, int64_t ws_floats, void* stream);
int laed_gru_bwd(const float* x, const float* w_ih, const float* w_hh,
                 const float* hbuf, float* gates, const float* hn,
                 const float* dout, const float* dhT, int T, int B, int E, int H,
                 float* dgh, float* dx, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh,
                 float* dh0, int accumulate_w, float* ws, int64_t ws_floats, void* stream);

/* The same two calls with a CU budget for the persistent small-batch kernels (B <= 64): those launches need ALL their
 * workgroups (H/4 forward, H/8 backward, one per CU) resident at once, so they are taken only when the grid fits
 * min(cu_budget, the device's CU count); otherwise the recurrence runs as per-step launches of the same core.
 * cu_budget = 0 means the whole device (what laed_gru_fwd / laed_gru_bwd pass).  A caller that runs two recurrences
 * side by side on two streams (DI-VST's two decoders, sent_models.py:397-405) passes half the CU count to each.
 * laed_gru_persist_timeouts: number of bounded in-kernel waits that timed out on the current device since the process
 * started (a grid that could not become co-resident poisons its outputs with NaN and bumps this counter); synchronous
 * 4-byte device read -- call it where the host synchronises anyway. */
int laed_gru_fwd_cus(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                     const float* h0, int T, int B, int E, int H,
                     float* hbuf, float* gates, float* hn, float* ws, int64_t ws_floats, int cu_budget, void* stream);
int laed_gru_bwd_cus(const float* x, const float* w_ih, const float* w_hh,
                     const float* hbuf, float* gates, const float* hn,
                     const float* dout, const float* dhT, int T, int B, int E, int H,
                     float* dgh, float* dx, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh,
                     float* dh0, int accumulate_w, float* ws, int64_t ws_floats, int cu_budget, void* stream);
int laed_gru_persist_timeouts(unsigned* out_host);
/* Persistent launches this process has issued so far (host-side counters, no device access): which path a call took
 * is otherwise invisible -- a CU budget, a sharer count or a grid that does not fit silently selects the per-step
 * launches.  No counterpart in the reference (nn.GRU hides its algorithm choice the same way). */
int laed_gru_persist_launches(unsigned* fwd_host, unsigned* bwd_host);

/* ------------------------------------------------------------------------------------------
 * Posterior / Gumbel-softmax latent (rows R = B*y_size, k classes, k <= 64).
 *   log_qy = log_softmax(logits)                         sent_models.py:110-111 ("Hidden2Discrete")
 *   g = -log(-log(u+1e-20)+1e-20); y = softmax((logits+g)/tau); ids = argmax (first max);
 *   hard: y = one_hot(ids)                               nn_lib.py:232-261 (GumbelConnector)
 * bwd: dlogits = softmax-Jacobian(y, dy)/tau  (+)  log-softmax-Jacobian(log_qy, dlog_qy)
 *   (dy == NULL or hard sample: first term skipped; dlog_qy == NULL: second term skipped)
 * ------------------------------------------------------------------------------------------ */
int laed_gumbel_latent_fwd(const float* logits, const float* u, int64_t R, int k, float tau, int hard,
                           float* y, int64_t* ids, float* log_qy, void* stream);
/* the same with the uniforms drawn inside the kernel (draw offset + r*k + c of stream `seed`: bit-identical to
 * laed_uniform followed by laed_gumbel_latent_fwd; replaces the CPU torch.rand + copy of nn_lib.py:233-235) */
int laed_gumbel_latent_fwd_rng(const float* logits, int64_t R, int k, float tau, int hard, uint64_t seed, uint64_t offset,
                               float* y, int64_t* ids, float* log_qy, void* stream);
int laed_gumbel_latent_bwd(const float* y, const float* dy, const float* log_qy, const float* dlog_qy,
                           int64_t R, int k, float tau, float* dlogits, void* stream);

/* ------------------------------------------------------------------------------------------
 * Batch-prior-regularised KL, MI (sent_models.py:138-156; criterions.py:97-109,136-147).
 * Three phases so that a data-parallel job can all-reduce `stats` (y*k+2 floats) between 1 and 2:
 *  1. laed_latent_stats:    stats[0..y*k) = sum_b q(y|x_b);  stats[y*k] = sum_rows H(q_row);
 *                           stats[y*k+1] = sum_rows KL(q_row || uniform)          (local batch)
 *                           workspace: nblk*(y*k+2) floats, nblk = laed_latent_stats_blocks()
 *  2. laed_latent_finalize: scal[0]=bpr, scal[1]=mi, scal[2]=row-KL mean (use_mutual=False reg_kl),
 *                           coef[y][k] = d bpr / d q[b][y][k]  (global batch size B_global)
 *  3. laed_latent_bwd:      dlog_qy[r][:] (+)= gscale[0] * (use_mutual == 1 ? coef.*q : use_mutual == 0 ? row-KL term
 *                                                          : d mi / d log_qy with coef from laed_latent_mi_coef)
 *     laed_latent_mi_coef:  coef[y][k] for the MI term (sent_models.py:155-156: mi is part of the returned Pack, so a
 *                           caller may differentiate it although no valid_loss of the reference does)
 * ------------------------------------------------------------------------------------------ */
int laed_latent_stats_blocks(void);
int laed_latent_stats(const float* log_qy, int64_t B, int y, int k, float* workspace, float* stats, void* stream);
int laed_latent_finalize(const float* stats, double B_global, int y, int k, float* scal, float* coef, void* stream);
/* laed_latent_stats + laed_latent_finalize without an exchange between them (one GPU: B_global = B), two launches
 * instead of three, same bits */
int laed_latent_stats_finalize(const float* log_qy, int64_t B, int y, int k, float* workspace, float* stats,
                               float* scal, float* coef, void* stream);
/* Allocation-free variant of laed_latent_stats.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate or bring a workspace -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * Same outputs as laed_latent_stats (bit for bit: same kernels, same fixed-order reduction); the block partials live
 * in a file-scope __device__ array inside the library (2.1 MB per GPU, held until the process ends) whose per-device
 * address is resolved once with hipGetSymbolAddress.  Limit: y*k <= 2048 (LAED_E_SIZE beyond it).  Calls nothing that
 * allocates (no hipMalloc / hipFree / torch).  Call it with the stream's device current.  All calls on one GPU share
 * the array: calls that overlap on one GPU (two streams) race -- serialise them. */
int laed_latent_stats_noalloc(const float* log_qy, int64_t B, int y, int k, float* stats, void* stream);
int laed_latent_mi_coef(const float* stats, double B_global, int y, int k, float* coef, void* stream);
int laed_latent_bwd(const float* log_qy, const float* coef, const float* gscale, int use_mutual,
                    int64_t B, double B_global, int y, int k, float* dlog_qy, int accumulate, void* stream);

/* The batch-mean posterior as a tensor of its own, for the train_prior branch (sent_models.py:138-139, 144-146):
 *   laed_avg_log_fwd:  avg_log[e] = log(stats[e] / B_global + 1e-15), e < n = y*k, stats from laed_latent_stats
 *                      (all-reduced under data parallelism)
 *   laed_avg_log_coef: coef[e] = davg_log[e] / (stats[e]/B_global + 1e-15) / B_global; then
 *                      laed_latent_bwd(use_mutual = 1, gscale = NULL) gives d log_qy = coef .* q */
int laed_avg_log_fwd(const float* stats, double B_global, int n, float* avg_log, void* stream);
int laed_avg_log_coef(const float* stats, const float* davg_log, double B_global, int n, float* coef, void* stream);

/* ------------------------------------------------------------------------------------------
 * CatKLLoss / Entropy on arbitrary rows (the general criterions, outside the fused block above).
 * Replaces criterions.py:97-109 (CatKLLoss.forward: sum_k exp(lq) (lq - lp) per row, then mean over rows or
 * sum / batch_size) and criterions.py:136-147 (Entropy.forward: sum_k -lq exp(lq); pass log_py = NULL).
 *  log_qy [R][k], k <= 64.  log_py is [py_rows][py_cols] with py_rows in {1, R} and py_cols in {1, k} (the
 *  broadcasts torch allows: the 1-element log_uniform_y of sent_models.py:81, a [k] prior, a full [R,k] prior).
 *  fwd: out[0] = scale * sum_r row_r  (scale = 1/R for unit_average, 1/batch_size otherwise); rows_out[R]
 *       optional; partials: laed_cat_rows_blocks() floats of scratch; fixed-order reduction (reproducible).
 *  bwd: dlog_qy = gscale[0]*scale * q (lq - lp + 1)   (entropy: -q (lq + 1));  dlog_py = -gscale[0]*scale * q,
 *       only for a full-shape log_py.  gscale == NULL means 1.
 * ------------------------------------------------------------------------------------------ */
int laed_cat_rows_blocks(void);
int laed_cat_rows_fwd(const float* log_qy, const float* log_py, int64_t py_rows, int py_cols, int64_t R, int k,
                      float scale, float* rows_out, float* partials, float* out, void* stream);
int laed_cat_rows_bwd(const float* log_qy, const float* log_py, int64_t py_rows, int py_cols, int64_t R, int k,
                      const float* gscale, float scale, float* dlog_qy, float* dlog_py, void* stream);

/* ------------------------------------------------------------------------------------------
 * Vocabulary projection fused with log-softmax + NLL (+ its backward), logits never materialised
 * beyond a [chunk_rows][ldl] workspace.
 * Replaces decoders.py:165-166 (project + log_softmax) and criterions.py:43-72 (F.nll_loss,
 * ignore_index=PAD) plus their autograd backward.
 *   h[N][H], W[V][H], b[V], labels[N] (int64; label == pad_id -> row ignored)
 *   row_scale[N] (device, may be NULL = 1): per-row weight of the NLL in the final loss (e.g.
 *       1/count_global for 'word' averaging) -- multiplies the gradients only.
 *   outputs: tok_nll[N] = lse - logit[label] (0 on ignored rows), lse[N]
 *   if with_grad: dh[N][H] = dL/dh, dW[V][H] (+)=, db[V] (+)= for  L = sum_r row_scale[r]*tok_nll[r]
 *   workspace: laed_proj_nll_workspace(H, V, chunk_rows) floats (chunk_rows*ldl, ldl = laed_proj_nll_ldl(V), plus
 *       fp32 transposed operand copies when the split-bf16 GEMM mode is on)
 *   with_grad = 0 (validation, GEN-side callers; exact-fp32 mode): the logits are not written at all -- the logits
 *       GEMM's epilogue keeps a (max, sum exp) pair per row per 128-column tile plus the label's logit, a second pass
 *       combines them in tile order.  workspace: laed_proj_nll_fwd_workspace(N, V) = N*(2*ceil(V/128) + 1) floats
 *       (62 MB at N = 19968, V = 50000, against 4.0 GB for the [N][V] tensor); chunk_rows is ignored.
 */
int     laed_proj_nll_ldl(int V);
int64_t laed_proj_nll_fwd_workspace(int64_t N, int V);
int64_t laed_proj_nll_workspace(int H, int V, int chunk_rows);   /* floats needed (depends on the GEMM math mode) */
int laed_proj_nll(const float* h, const float* W, const float* b, const int64_t* labels, const float* row_scale,
                  int64_t N, int H, int V, int64_t pad_id, int with_grad, int accumulate_w,
                  float* workspace, int64_t ws_floats, int chunk_rows,
                  float* tok_nll, float* lse, float* dh, float* dW, float* db, void* stream);

/* ------------------------------------------------------------------------------------------
 * Expected embedding of the decoder's output distribution ("attribute forcing" of AE-ED / ST-ED):
 *   out[N][E] = softmax(h W^T + b) Wemb
 * Replaces torch.exp(dec_outs.view(-1, V)) @ x_embedding.weight, laed/models/dialog_models.py:421-422
 * (AeED) and :726-727 (StED), and its autograd backward; the [N][V] probabilities live only in a
 * [chunk_rows][ldl] workspace tile (two tiles in backward, which recomputes the logits from h).
 *   fwd: lse[N] (saved for backward), out[N][E]
 *   bwd: dout[N][E] -> dh[N][H], dW[V][H] (+)=, db[V] (+)=, dWemb[V][E] (+)=   ((+)= if accumulate_w)
 *   workspace: laed_softmax_embed_workspace(V, chunk_rows, backward) floats = the tiles plus up to one more tile (at most
 *   2^26 floats) of split-K scratch for the long-K products with few output tiles (P Wemb, dP W); the calls accept any
 *   size from the bare tiles up and split K only as far as the surplus allows
 * ------------------------------------------------------------------------------------------ */
int64_t laed_softmax_embed_workspace(int V, int chunk_rows, int backward);
int laed_softmax_embed_fwd(const float* h, const float* W, const float* b, const float* Wemb,
                           int64_t N, int H, int V, int E, float* workspace, int64_t ws_floats, int chunk_rows,
                           float* lse, float* out, void* stream);
int laed_softmax_embed_bwd(const float* h, const float* W, const float* b, const float* Wemb, const float* lse,
                           const float* dout, int64_t N, int H, int V, int E,
                           float* workspace, int64_t ws_floats, int chunk_rows,
                           float* dh, float* dW, float* db, float* dWemb, int accumulate_w, void* stream);

/* log-probabilities for callers that really want the [N][V] tensor (DecoderRNN's public contract,
 * decoders.py:166): logp = logits - lse, in place on a caller buffer of [N][ldl]. */
int laed_log_softmax_rows(float* logits, int64_t N, int V, int ldl, float* lse, void* stream);
/* its backward, for a caller that builds a loss on the dense tensor: dlogits = dlogp - exp(logp) * rowsum(dlogp)
 * (row strides in floats; dlogits may alias dlogp); dh / dW / db then follow from laed_sgemm as for any nn.Linear. */
int laed_log_softmax_rows_bwd(const float* logp, int64_t ld_logp, const float* dlogp, int64_t ld_dlogp, int64_t N, int V,
                              float* dlogits, int64_t ld_dlogits, void* stream);

/* greedy symbol selection  ids[r] = argmax_v logits[r][v]  (lowest index on ties), decoders.py:218-219
 * (topk(1) of log_softmax == argmax of logits). */
int laed_argmax_rows(const float* logits, int64_t N, int V, int ldl, int64_t* ids, void* stream);

/* Greedy free-running decode, max_len steps from h0 (DecoderRNN GEN/greedy, decoders.py:180-192,258-268):
 * ids_out[(max_len+1)][B]: row 0 = go_id, row t+1 = symbols chosen at step t; hbuf[(max_len+1)][B][H] as in
 * laed_gru_fwd; xbuf[B][E], gates[B][3H], logits[B][laed_proj_nll_ldl(V)] are caller-provided scratch.  In the exact-fp32
 * mode and from 512 output tiles of 128 x 128 upwards the symbols come out of the projection GEMM's arg-max epilogue and
 * the scratch only ever holds [ceil(V/128)][B][2] per-tile partials: the [B][V] logits are not written. */
int laed_gru_decode_greedy(const float* emb, const float* w_ih, const float* w_hh, const float* b_ih,
                           const float* b_hh, const float* proj_w, const float* proj_b, const float* h0,
                           int B, int E, int H, int V, int max_len, int64_t go_id,
                           float* hbuf, float* xbuf, float* gates, float* logits, int64_t* ids_out, void* stream);

/* Sampled free-running decode, gen_type='sample' (decoders.py:220-221 -> gumbel_max, base_modules.py:54-67):
 * as laed_gru_decode_greedy but step t picks argmax_v (log_softmax(logits)[v] - log(-log(u[t][r][v]))).
 * u: [max_len][B][V] uniforms to replay, or NULL = the library's counter RNG keyed by (seed, t, row0 + r, v)
 * (row0 = global row of this rank's first row, so a sharded job draws the single-GPU job's noise). */
int laed_gru_decode_sample(const float* emb, const float* w_ih, const float* w_hh, const float* b_ih,
                           const float* b_hh, const float* proj_w, const float* proj_b, const float* h0,
                           int B, int E, int H, int V, int max_len, int64_t go_id,
                           const float* u, uint64_t seed, uint64_t row0,
                           float* hbuf, float* xbuf, float* gates, float* logits, int64_t* ids_out, void* stream);

/* One decode step at a time, for callers that drive their own loop (decoders.py:152-167 forward_step + :219 topk(1)):
 *   laed_gru_cell_fwd: gates[B][3H] (scratch) = x W_ih^T + b_ih, then the fused recurrent step -> hout[B][H]
 *                      (hout must not alias hprev);
 *   laed_proj_argmax:  logits[B][ldl] (scratch, ldl = laed_proj_nll_ldl(V)) = h W_proj^T + b, ids[b] = first maximum
 *                      (= topk(1) of the log-softmax). */
int laed_gru_cell_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                      const float* hprev, int B, int E, int H, float* gates, float* hout, void* stream);
int laed_proj_argmax(const float* h, const float* proj_w, const float* proj_b, int B, int H, int V,
                     float* logits, int64_t* ids, void* stream);

/* masked reductions used by NLLEntropy's averaging modes (criterions.py:47-68):
 * out[0] = sum_r tok_nll[r], out[1] = #(labels != pad). */
int laed_nll_reduce(const float* tok_nll, const int64_t* labels, int64_t N, int64_t pad_id, float* out, void* stream);
/* What NLLEntropy.forward derives from the labels (criterions.py:43-72), one single-block launch: the per-row loss weights
 * of the averaging mode (avg_mode 0 = None: 1; 1 = 'seq': 1/denom; 2 = 'real_word': 1 / (sign(labels).sum(row) * denom);
 * 3 = 'word': 1 / count, count = global_count[0] if given (already all-reduced) else this batch's own non-PAD count;
 * denom = B * world size), the labels re-ordered for the consumer (labels[b*ld + t] -> time-major t*B + b, or batch-first;
 * labels_out may be NULL) and, if local_count != NULL, this batch's non-PAD count. */
int laed_nll_prepare(const int64_t* labels, int64_t ld, int B, int T, int64_t pad_id, int avg_mode, double denom,
                     const float* global_count, int time_major, int64_t* labels_out, float* scale_out,
                     float* local_count, void* stream);

/* NLLEntropy on a dense log-prob tensor (criterions.py:43-72 handed a materialised [N,V] input instead of the lazy
 * decoder handle): tok_nll[r] = labels[r] == pad_id ? 0 : -logp[r][labels[r]]  (row stride ld >= V);
 * bwd: dlogp = 0 except dlogp[r][labels[r]] = -gscale[0]*row_scale[r] on non-pad rows.  A label outside [0,V)
 * that is not pad_id gives NaN (F.nll_loss raises there). */
int laed_nll_gather_fwd(const float* logp, int64_t ld, const int64_t* labels, int64_t N, int V, int64_t pad_id,
                        float* tok_nll, void* stream);
int laed_nll_gather_bwd(const int64_t* labels, const float* row_scale, const float* gscale, int64_t N, int V,
                        int64_t pad_id, float* dlogp, int64_t ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * Attention pooling of RnnUttEncoder (encoders.py:119-130: fc1 = tanh(key_w(enc_outs)); attn = softmax_t(query(fc1));
 * utt_embedded = sum_t attn * enc_outs) around the two Linear layers, which run on laed_sgemm.
 *   laed_attn_score_fwd: fc1[R][Hk] holds key_w's output on entry and tanh of it on return (kept for backward);
 *                        score[r] = fc1[r] . wq + bq[0]   (query.weight [1][Hk], query.bias [1]; bq may be NULL)
 *   laed_attn_pool_fwd:  attn[N][T] = softmax over t of score[n*s_sn + t*s_st];  out[N][D] = sum_t attn * enc,
 *                        enc element (n,t,d) at enc[n*e_sn + t*e_st + d]  (T <= 1024): the GRU's time-major output is
 *                        read in place (e_sn = D, e_st = N*D)
 *   laed_attn_pool_bwd:  dscore (same strides as score) and, if denc != NULL, denc[n*g_sn + t*g_st + d] = attn*dout
 *   laed_attn_score_bwd: dpre[r][j] = dscore[r] wq[j] (1 - fc1^2)  (d wq = dscore^T fc1 and d bq = sum dscore are a
 *                        laed_sgemm_ws call with transA and its column sum)
 * laed_tanh_fwd/bwd: the tanh between p_fc1 and p_y in the prior network (dialog_models.py:243, :513).
 * ------------------------------------------------------------------------------------------ */
int laed_attn_score_fwd(float* fc1, const float* wq, const float* bq, int64_t R, int Hk, float* score, void* stream);
int laed_attn_score_bwd(const float* fc1, const float* dscore, const float* wq, int64_t R, int Hk, float* dpre, void* stream);
int laed_attn_pool_fwd(const float* score, int64_t s_sn, int64_t s_st, const float* enc, int64_t e_sn, int64_t e_st,
                       int64_t N, int T, int D, float* attn, float* out, void* stream);
int laed_attn_pool_bwd(const float* enc, int64_t e_sn, int64_t e_st, const float* attn, const float* dout,
                       int64_t N, int T, int D, float* dscore, int64_t s_sn, int64_t s_st,
                       float* denc, int64_t g_sn, int64_t g_st, void* stream);
int laed_tanh_fwd(const float* x, float* y, int64_t n, void* stream);
int laed_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Adam (torch.optim.Adam defaults; model_bases.py:122-125): in place on p, m, v.  `step` is 1-based.
 * grad_scale multiplies g first (1/world_size after a sum all-reduce).
 * ------------------------------------------------------------------------------------------ */
int laed_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step,
                   double lr, double beta1, double beta2, double eps, float grad_scale, void* stream);
/* The same update for `ntensors` tensors in as few launches as their descriptors fit into kernel arguments (16 per
 * launch); p, g, m, v, n are HOST arrays of device pointers / element counts.  All tensors share `step` (torch's Adam
 * keeps one step count per parameter: the caller groups parameters by it).  Element-wise identical to laed_adam_step. */
int laed_adam_step_multi(int ntensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                         const int64_t* n, int step, double lr, double beta1, double beta2, double eps,
                         float grad_scale, void* stream);

/* The other two optimizers get_optimizer can build (model_bases.py:126-133): torch.optim.SGD(lr, momentum) and
 * torch.optim.RMSprop(lr, momentum) with torch's defaults (alpha 0.99, eps 1e-8, not centred); in place, state
 * buffers start at zero; momentum_buf may be NULL when momentum == 0. */
int laed_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, double lr, double momentum,
                  float grad_scale, void* stream);
int laed_rmsprop_step(float* p, const float* g, float* square_avg, float* momentum_buf, int64_t n, double lr,
                      double alpha, double eps, double momentum, float grad_scale, void* stream);

/* small utilities */
int laed_dot(const float* a, const float* b, int64_t n, float* out, void* stream);      /* out[0] = sum a*b (one block, fixed order) */
int laed_axpby(float* y, const float* x, int64_t n, float a, float b, void* stream);   /* y = a*x + b*y */
int laed_scale_by_dev(float* y, const float* x, const float* s, int64_t n, void* stream); /* y = s[0]*x */
int laed_scale_by_dev_multi(int ntensors, float* const* y, const int64_t* n, const float* s, void* stream); /* y_t *= s[0], <= 8 tensors, one launch */
/* sig[i] = sigmoid(x[i]), tnh[i] = tanh(x[i]) with the branch-free gate functions every GRU kernel of the library applies
 * (torch.sigmoid / torch.tanh inside nn.GRU, base_modules.py:43-45): exposed so that their accuracy can be checked on its own */
int laed_gate_functions(const float* x, int64_t n, float* sig, float* tnh, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement aid (no counterpart in the reference): when enabled, every GEMM launch is bracketed by HIP events
 * on its own stream.  laed_prof_collect waits for them and fills a HOST array of 21 doubles:
 * out[v*3+0..2] = {launches, total ms, total algorithmic flops} for variant v = layout*2 + tile,
 * layout 0 = NT (x W^T), 1 = NN (dy W), 2 = TN (dy^T x); tile 0 = 128x128, 1 = 64x64.
 * ------------------------------------------------------------------------------------------ */
int laed_prof_enable(int on);
int laed_prof_reserve(int n_events);         /* pre-create events (2 per GEMM launch) outside the timed region */
int laed_prof_collect(double* out_host21);   /* variant 6 = split-bf16 NT 128x128 (see laed_set_gemm_mode) */

/* GEMM math mode (process-wide; default 0, or LAED_GEMM_MODE=bf16x3 in the environment):
 *   0 = exact fp32 on v_mfma_f32_32x32x2_f32 everywhere (bit-for-bit an fp32 fma chain)
 *   1 = K-contiguous (x W^T) products with >= 192 128x128 tiles run as split-bf16 "bf16x3": operands are split
 *       hi+lo into bf16 on the fly and accumulated in fp32 (a_lo b_hi + a_hi b_lo + a_hi b_hi); ~7x the rounding
 *       error of fp32 (still ~1e-6 relative on this model's losses), ~2.5x the throughput. */
int laed_set_gemm_mode(int mode);
int laed_set_xgemm_min_tiles(int n);   /* test hook: minimum 128x128 tile count for mode 1 to apply (default 192) */
int laed_get_gemm_mode(void);

#ifdef __cplusplus
}
#endif
#endif /* LAED_HIP_H */
