/* hmmc_hip.h — C-ABI of libhmmc_hip.so, the MI355X (gfx950) kernels behind the HMMC training
 * hot path.  The reference (cheetah003/HMMC) has no FFI boundary of its own: its hot path sits
 * behind torch nn.Module / Optimizer objects.  Each entry point below therefore names the
 * reference op sequence (file:line under the reference checkout) whose device work it replaces;
 * the Python classes in hmmc_amd/ that mirror the reference's module API call these through
 * ctypes (hmmc_amd/_lib.py), and INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions: plain device pointers and sizes; the caller owns every buffer (outputs and
 * workspaces are pre-allocated; hmmc_*_workspace() returns the bytes an op needs); kernels are
 * enqueued on `stream` (a hipStream_t) and never synchronise; no device memory is ever allocated.
 * Re-entrant across host threads and devices (device properties and LDS opt-ins are kept per
 * device).  Process-wide state, all of it listed here: (1) hmmc_gemm_reserve_cus(n) and the A/B switches
 * of hmmc_set_option, settings; (2) the benchmark timing switch hmmc_gemm_profile_start/stop
 * (mutex-protected; creates HIP events while on, and stop() synchronises the device); (3) seven HIP
 * events per (stream, weight-gradient stream) pair, created by that pair's first hmmc_tower_bwd /
 * hmmc_tower_bwd_fold call, reused by every later one and given back by hmmc_tower_release.
 * The library reads NO environment variable: results and speed depend on the arguments and on the
 * settings above only.
 * Return value: 0 on success, HMMC_ERR_* (< 0) otherwise — nothing is launched on error.
 * fp16 buffers are IEEE binary16; "tokens" are rows of a row-major [tokens, D] matrix with each
 * sequence's L tokens contiguous.
 */
#ifndef HMMC_HIP_H
#define HMMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hmmc_stream_t; /* hipStream_t */

#define HMMC_OK 0
#define HMMC_ERR_ARG (-1)
#define HMMC_ERR_UNSUPPORTED (-2)
#define HMMC_ERR_WORKSPACE (-3)
#define HMMC_ERR_LAUNCH (-4)

/* epilogue flags of hmmc_gemm_f16 */
#define HMMC_EPI_BIAS 1  /* + bias[n]                                   */
#define HMMC_EPI_RESID 2 /* out = fp16(resid + fp16(acc + bias))        */
#define HMMC_EPI_QGELU 4 /* out = QuickGELU(h), aux_out = h = fp16(acc + bias) */
#define HMMC_EPI_DGELU 8 /* out = acc * QuickGELU'(aux_in)              */
#define HMMC_EPI_SAVE_DGELU 64 /* with QGELU: aux_out = QuickGELU'(h) instead of h (what the backward multiplies by) */
#define HMMC_EPI_MULAUX 128 /* out = acc * aux_in (backward of QuickGELU with the saved derivative)        */
#define HMMC_EPI_COLSUM 32 /* + fp32 partial column sums of C into `workspace` (see hmmc_gemm_f16_colsum_rows) */
#define HMMC_EPI_LNFOLD 256  /* hmmc_gemm_f16_fold: acc -> rowstat[m][0] * acc + rowstat[m][1] * colterms[n] + colterms[N + n] */
#define HMMC_EPI_ROWSTAT 512 /* hmmc_gemm_f16_fold: + (sum, sum of squares) of every output row per 64-column block into stat_part */
#define HMMC_EPI_ROWSCALE 1024 /* hmmc_gemm_f16_fold: out *= rowstat[m][0] as the last step (column sums stay those of the unscaled values) */

/* fp16 MFMA GEMM, fp32 accumulate: C[M,N] = epilogue(sum_k Aop[m][k] * Bop[n][k]).
 * a_kmajor: Aop[m][k] = A[m*lda + k], else A[k*lda + m]; likewise b_kmajor for B (rows n).
 * Replaces F.linear / nn.MultiheadAttention in/out projections and the MLP of
 * ResidualAttentionBlock (modules/module_clip.py:231-257), the patch conv1 (:278,307), the
 * ln_post/ln_final projections (modules/module_cross.py:228,296) and all their backward GEMMs.
 * Without an epilogue, small-output/long-K problems (weight gradients) are split over K into
 * fp32 slabs in `workspace` (hmmc_gemm_f16_workspace bytes; may be NULL to disable). */
size_t hmmc_gemm_f16_workspace(int M, int N, int K);
/* With HMMC_EPI_COLSUM (32) in `epilogue` (any epilogue, no split-K) `workspace` instead receives fp32 partial column
 * sums of the fp16 values written to C: hmmc_gemm_f16_colsum_rows(M, N, K) rows of N floats, one per 128 (or 64) output
 * rows; hmmc_colsum over them gives the bias gradient of the layer that produced C's pre-image (c_fc at
 * module_clip.py:240) without re-reading C. */
size_t hmmc_gemm_f16_colsum_rows(int M, int N, int K);
/* Benchmark-only live timing of every hmmc_gemm_f16 / hmmc_gemm_f32 launch with HIP events on the launch stream; stop()
 * synchronises and returns in arrays of FOUR - slots 0 forward, 1 dgrad, 2 wgrad (hmmc_gemm_f16 by operand layout), 3 =
 * hmmc_gemm_f32 - the summed 2MNK flops, algorithmic operand bytes, seconds and launch counts. */
int hmmc_gemm_profile_start(void);
int hmmc_gemm_profile_stop(double* flops, double* bytes, double* seconds, long* launches);
/* The weight gradients of one layer as ONE launch: dW_j[Np_j, Kp_j] = dY_j[T, Np_j]^T X_j[T, Kp_j], j < nprob <= 4, all over the
 * same T tokens (F.linear's weight gradients of in_proj / out_proj / c_fc / c_proj, modules/module_clip.py:231-257).  Work
 * items of all problems share one persistent grid and one K split (chosen so that their tiles together fill the chip), fp32
 * partial slabs in `workspace`, one reduce.  dY / X / dW / Np / Kp: HOST arrays of nprob device pointers / sizes.
 * hmmc_gemm_f16_wgrad_group_workspace returns the workspace bytes, or 0 when the shapes should take one hmmc_gemm_f16 call per
 * gradient instead (dimensions that are not multiples of 256, fewer than 512 tokens, operands of 2 GiB and more). */
size_t hmmc_gemm_f16_wgrad_group_workspace(const int* Np, const int* Kp, int nprob, int T);
/* dW32 (may be NULL; HOST array of nprob device pointers, entries may be NULL): problem j with dW32[j] != NULL leaves as dense
 * fp32 sums [Np_j][Kp_j] there instead of fp16 in dW[j] - the folded weight gradients that hmmc_fold_grad_finish completes. */
int hmmc_gemm_f16_wgrad_group(const void* const* dY, const void* const* X, void* const* dW, float* const* dW32, const int* Np,
                              const int* Kp, int nprob, int T, void* workspace, size_t ws_bytes, hmmc_stream_t stream);
/* Leave `cus` (0..128, default 0) compute units out of every later hmmc_gemm_f16 grid.  The host sets this once when
 * gradients are all-reduced while the backward pass runs (DistributedDataParallel at main_task_retrieval.py:207, main_pretrain.py:204), so that
 * RCCL's workgroups find free CUs instead of waiting for a persistent GEMM grid to drain.  Process-wide. */
int hmmc_gemm_reserve_cus(int cus);
int hmmc_gemm_f16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int a_kmajor,
                  int b_kmajor, const void* bias, const void* resid, void* aux_out, const void* aux_in, int epilogue,
                  void* workspace, size_t ws_bytes, hmmc_stream_t stream);

/* LayerNorm folded into the linear layer behind it (fp16 towers; ln_1 -> in_proj and ln_2 -> c_fc of ResidualAttentionBlock,
 * modules/module_clip.py:252-256 with the LayerNorm of :217-223):
 *   LN(x) W^T + b = rstd_r (x (gamma o W)^T)[r][n] - rstd_r mean_r c_n + d_n,  c_n = sum_k (gamma o W)[n][k],  d_n = sum_k beta_k W[n][k] + b_n
 * so the GEMM reads the raw residual stream and LN(x) never exists in memory.  hmmc_ln_fold_prep writes gamma o W (fp16, one
 * rounding), c (sums of the rounded values) and d for `count` <= 32 weight matrices [N_e][K] in one launch (W / gamma / beta /
 * bias / Wf / cd / N are HOST arrays; cd_e is [2][N_e] fp32: c, then d; bias entries may be NULL).  hmmc_rowstat gives the row
 * pairs (rstd_r, -rstd_r mean_r) [rows][2] of fp16 rows; hmmc_rowstat_finalize gives the same from the partial sums a
 * HMMC_EPI_ROWSTAT launch wrote ([nparts = D / 64][rows][2]).  hmmc_gemm_f16_fold is hmmc_gemm_f16 for a k-major A (b_kmajor as
 * there) with those operands: HMMC_EPI_LNFOLD (needs rowstat, colterms; bias is inside d) optionally with HMMC_EPI_QGELU
 * [| HMMC_EPI_SAVE_DGELU, aux_out]; HMMC_EPI_ROWSTAT (needs stat_part, N % 64 == 0) with any of BIAS / RESID; HMMC_EPI_ROWSCALE
 * (needs rowstat) with MULAUX | COLSUM: the data gradient in front of a folded LayerNorm (below).  `workspace` as in
 * hmmc_gemm_f16 (COLSUM partials).  The rounding points differ from LayerNorm-then-GEMM (gamma o W is rounded instead of
 * LN(x)): hmmc_tower_fwd_fused uses this for forward passes that keep no activations and, opt-in, for training.
 *
 * Backward of a folded layer.  The gradient reaching the LayerNorm is du = dy W'; with dy~ = rstd_r dy[r][:] handed over by
 * the producer (HMMC_EPI_ROWSCALE, hmmc_attention_f16_bwd_scaled) the data-gradient GEMM gives du~ = rstd_r du and
 * hmmc_layernorm_bwd_fold computes dx = du~ - mean(du~) - u mean(du~ o u) (+ dres), u = stat[r][0] x + stat[r][1]; its optional
 * partial [hmmc_layernorm_bwd_fold_rows(rows)][D] holds column sums of dx for hmmc_multi_colreduce.  The weight gradient is
 * taken against the RAW rows, S = dy~^T x in fp32 (hmmc_gemm_f16_wgrad_group's dW32), and hmmc_fold_grad_finish turns up to
 * 32 such sums into dW = gamma_k (S - rowmean(S)) + beta_k db_n (fp16), dgamma_k = sum_n W[n][k] (S - rowmean(S))[n][k] and
 * dbeta_k = sum_n W[n][k] db_n (fp32) - sum_k (x[r][k] - mean_r) = 0 makes each row's mean correction that row's own mean.
 * HOST arrays of `count` entries; db_e: the layer's fp16 bias gradient [N_e]; vmean_e: fp32 scratch of
 * hmmc_fold_grad_scratch_floats(N_e, K) floats. */
int hmmc_ln_fold_prep(const void* const* W, const float* const* gamma, const float* const* beta, const void* const* bias,
                      void* const* Wf, float* const* cd, const int* N, int K, int count, hmmc_stream_t stream);
int hmmc_rowstat(const void* x, float* stat, int rows, int D, long stride, float eps, hmmc_stream_t stream);
int hmmc_rowstat_finalize(const float* part, float* stat, int nparts, int rows, int D, float eps, hmmc_stream_t stream);
int hmmc_gemm_f16_fold(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int b_kmajor,
                       const void* bias, const void* resid, void* aux_out, const void* aux_in, int epilogue, const float* rowstat,
                       const float* colterms, float* stat_part, void* workspace, size_t ws_bytes, hmmc_stream_t stream);
int hmmc_layernorm_bwd_fold_rows(int rows);
int hmmc_layernorm_bwd_fold(const void* dut, const void* x, const float* stat, const void* dres, void* dx, float* partial,
                            int want_dx_colsum, int rows, int D, long stride, hmmc_stream_t stream);
size_t hmmc_fold_grad_scratch_floats(int N, int K);   /* floats of each vmean_e */
int hmmc_fold_grad_finish(const float* const* S, const void* const* W, const float* const* gamma, const float* const* beta,
                          const void* const* db, void* const* dW, float* const* dgamma, float* const* dbeta, float* const* vmean,
                          const int* N, int K, int count, hmmc_stream_t stream);

/* LayerNorm over the last dim (fp32 statistics).  dtype 0: fp16 in/out (CLIP LayerNorm,
 * modules/module_clip.py:217-223, eps 1e-5); dtype 1: fp32 (TF-style LN of the temporal blocks and
 * MLM head, modules/until_module.py:54-67, eps 1e-12).  Output row r reads input row
 * (row_index ? row_index[r] : r) at stride in_stride — used to normalise only the CLS / EOT rows
 * (modules/module_cross.py:228-230,296-300).  mean/rstd [rows] are saved for the backward. */
int hmmc_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                       const int* row_index, int rows, int D, long in_stride, float eps, int dtype, hmmc_stream_t stream);
size_t hmmc_layernorm_bwd_workspace(int rows, int D);
/* dx[row] = LN'(dy)[row] + (dres ? dres[row] : 0), written at the rows the forward read.  dx_colsum (optional, [D],
 * dtype of dx) receives the column sums of the dx rows written: the bias gradient of the linear layer whose output feeds
 * this LayerNorm's input (out_proj / c_proj bias at module_clip.py:235-246), without a second pass over dx. */
int hmmc_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                       const void* dres, void* dx, float* dgamma, float* dbeta, void* dx_colsum, const int* row_index,
                       int rows, int D, long in_stride, int dtype, void* workspace, size_t ws_bytes, hmmc_stream_t stream);

/* The two stages of hmmc_layernorm_bwd separately, for callers that batch the second one (hmmc_tower_bwd):
 * hmmc_layernorm_bwd_partial writes dx and the fp32 partial matrix [hmmc_layernorm_bwd_rows(rows)][np * D], np = 2
 * (dgamma | dbeta) or 3 (... | column sums of dx); hmmc_multi_colreduce sums the rows of up to many such matrices in ONE
 * launch: task t writes out[c / seg][c % seg] = sum_r partial[r][c] with the given dtype (0 fp16, 1 fp32; out[] entries may
 * be NULL, N <= 3 seg).  `tasks` is a HOST array. */
typedef struct HmmcReduceTask { const float* partial; int R, N, seg; void* out[3]; int dtype[3]; } HmmcReduceTask;
int hmmc_layernorm_bwd_rows(int rows);
int hmmc_layernorm_bwd_partial(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                               const void* dres, void* dx, float* partial, int want_dx_colsum, const int* row_index, int rows,
                               int D, long in_stride, int dtype, hmmc_stream_t stream);
int hmmc_multi_colreduce(const void* tasks, int ntasks, hmmc_stream_t stream);

/* out[n] = sum_m X[m][n] (bias, class/positional-embedding gradients). dtypes: 0 fp16, 1 fp32. */
size_t hmmc_colsum_workspace(int M, int N);
int hmmc_colsum(const void* X, void* out, int M, int N, long ld, int in_dtype, int out_dtype, int round_f16,
                void* workspace, size_t ws_bytes, hmmc_stream_t stream);

/* Patch extraction for conv1 (kernel = stride = patch, modules/module_clip.py:278,307-310):
 * fp32 NCHW frames -> [nframes*(g*g+1), 3*patch*patch]; row 0 of each frame is zero (class slot).
 * out_dtype / dtype of the entry points below: 0 = fp16 (the towers as written, convert_weights, module_clip.py:506-527,
 * image.type(fp16) at module_cross.py:224), 1 = fp32 (the reference after model.float(): no value is rounded). */
int hmmc_patchify(const float* img, void* out, int nframes, int H, int W, int patch, int out_dtype, hmmc_stream_t stream);
/* The same from uint8 NCHW frames with the loader's normalisation fused in (dataloaders/dataloader_msrvtt_retrieval.py:
 * 242-247: x/255, (x - mean[c]) / std[c] in fp32, then fp16): a quarter of the input bytes.  mean3 / std3 are HOST arrays.
 * frame_index (device int32 [nframes], may be NULL): output frame n is read from stored frame frame_index[n] of img - the
 * loader's frame sampling (:296-312 sample_slice) applied in place, without a gathered copy of the chosen frames. */
int hmmc_patchify_u8(const void* img, const int* frame_index, void* out, int nframes, int H, int W, int patch,
                     const float* mean3, const float* std3, int out_dtype, hmmc_stream_t stream);
/* In place on the patch-GEMM output: class_embedding into row 0, + positional_embedding
 * (modules/module_clip.py:311-312), with the reference's fp16 rounding points (dtype 0) or in fp32 (dtype 1). */
int hmmc_vit_embed(void* x, const float* cls, const float* pos, long rows, int L, int D, int dtype, hmmc_stream_t stream);
/* hmmc_vit_embed and ln_pre (modules/module_clip.py:311-313) in one pass, fp16 tower: x0 [rows, D] is the patch-GEMM output and
 * is rewritten with the embedded rows when write_x0 (the backward of ln_pre reads them); y = ln_pre(embedded rows); mean / rstd
 * [rows] as hmmc_layernorm_fwd saves them; stat (may be NULL) [rows][2] = (rstd, -rstd mean) of the rows of y for
 * hmmc_tower_fwd_fused.  Results are bit-identical to hmmc_vit_embed followed by hmmc_layernorm_fwd. */
int hmmc_vit_embed_ln(void* x0, const float* cls, const float* pos, const float* gamma, const float* beta, void* y, float* mean,
                      float* rstd, float* stat, int rows, int L, int D, float eps, int write_x0, hmmc_stream_t stream);
/* token_embedding(ids).half() + positional_embedding[:L].half() (modules/module_cross.py:288-291).  An id outside
 * [0, vocab) never indexes the table (the reference's nn.Embedding raises): its row is the position embedding alone and
 * *err_flag (device int, may be NULL) is set to 1 for the host to check. */
int hmmc_text_embed(const long* ids, const float* table, const float* pos, void* x, long rows, int L, int D, long vocab,
                    int* err_flag, int out_dtype, hmmc_stream_t stream);
/* index[i] = base + i * stride + argmax_l ids[i][l] (the first position of the largest id, torch.argmax's tie rule), i < b: the row
 * of caption i's EOT token - CLIP's largest id - in a token-major activation buffer whose captions sit `stride` rows apart
 * (modules/module_cross.py:300-303: x[arange(b), text.argmax(-1)]); ONE launch for the reference's arange / argmax / add. */
int hmmc_eot_index(const long* ids, int* index, int b, int L, long base, long stride, hmmc_stream_t stream);
/* dense fp32 embedding gradient: dtable[id] = sum of dx[r] over the rows with ids[r] == id (dtable zeroed by the caller;
 * ids outside [0, vocab) contribute nothing).  No atomics: bit-identical from run to run.  dx_dtype: 0 fp16, 1 fp32. */
int hmmc_text_embed_bwd(const long* ids, const void* dx, float* dtable, long rows, int D, long vocab, int dx_dtype,
                        hmmc_stream_t stream);
/* kind 0: fp16 -> fp32, 1: fp32 -> fp16 */
int hmmc_cast(const void* in, void* out, long n, int kind, hmmc_stream_t stream);

/* Fused softmax(QK^T/8 + mask)V per (sequence, head), head dim 64, L <= 64.  qkv: [nseq*L, 3*64*H]
 * packed in-projection output; out: [nseq*L, 64*H]; lse: [nseq, H, L] log-sum-exp rows (saved for
 * backward).  causal != 0 applies CLIP's text mask (modules/module_clip.py:441-447).
 * Replaces the attention core of nn.MultiheadAttention at modules/module_clip.py:251. */
int hmmc_attention_f16_fwd(const void* qkv, void* out, float* lse, int nseq, int L, int H, int causal,
                           hmmc_stream_t stream);
/* dbias_partial (optional, fp32 [nseq][3*64*H]): per-sequence column sums of dqkv, i.e. partial sums of
 * the in-projection bias gradient; the caller finishes with hmmc_colsum over the nseq rows instead of re-reading dqkv. */
int hmmc_attention_f16_bwd(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv,
                           float* dbias_partial, int nseq, int L, int H, int causal, hmmc_stream_t stream);
/* The same with row r of dqkv multiplied by rowstat[r][0] on its way out - the rstd of a folded ln_1, see
 * hmmc_gemm_f16_fold; dbias_partial stays the column sums of the unscaled gradient. */
int hmmc_attention_f16_bwd_scaled(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv,
                                  float* dbias_partial, const float* rowstat, int nseq, int L, int H, int causal,
                                  hmmc_stream_t stream);
/* Query 0 of every sequence only - the last block of a tower whose caller reads the class token alone (modules/module_cross.py:228-230
 * takes hidden[:, 0, :]; hmmc_tower_fwd / hmmc_tower_bwd with lead_only).  Forward: out row n*L and lse[(n, h, 0)], bit-identical to
 * hmmc_attention_f16_fwd's, from K and V of all L tokens and Q of token 0; no other row of out / lse is written, no other Q is
 * read.  Backward: dout is read at row n*L only; dK and dV of every token and dQ of token 0 are written (the Q columns of the
 * other rows of dqkv are left untouched: their gradient is exactly zero); rowstat optional, as hmmc_attention_f16_bwd_scaled;
 * out (the forward's result, read at row n*L only) is needed above 64 tokens.  L <= 256. */
int hmmc_attention_f16_fwd_lead(const void* qkv, void* out, float* lse, int nseq, int L, int H, int causal, hmmc_stream_t stream);
int hmmc_attention_f16_bwd_lead(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv,
                                float* dbias_partial, const float* rowstat, int nseq, int L, int H, int causal, hmmc_stream_t stream);

/* fp32 MFMA GEMM (exact f32 FMA chain) with general strides: C[m][n] = epi(alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]);
 * one stride of each operand must be 1.  Epilogue flags as hmmc_gemm_f16 plus HMMC_EPI_RELU; QuickGELU is evaluated in fp32.
 * Temporal transformer (modules/module_cross.py:114-149), similarity matrices (modules/modeling.py:207-229,286-313),
 * projector MLPs (:788-807), MLM head (modules/module_cross.py:308-357).
 * Three kernels behind one entry point, picked by an estimate of their times: 64x64x16 tiles staged through LDS (many tiles),
 * 16x32 tiles whose four waves split K with operands straight from global memory (few tiles), and the same scheme on
 * (32..96)x64 tiles for problems of at most one tile per CU with K >= ~1 536.  Every kernel sums in a fixed order: the
 * result of a call depends on its arguments only (bit-identical from launch to launch), not on which kernel another shape took. */
#define HMMC_EPI_RELU 16
int hmmc_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak, long sbk, long sbn,
                  int ldc, float alpha, const float* bias, const float* resid, float* aux_out, const float* aux_in,
                  int epilogue, hmmc_stream_t stream);

/* y = x / max(||x||, eps) per row (eps 0: loose_similarity, modules/modeling.py:210-214; eps 1e-12: F.normalize, :289-292). */
int hmmc_l2norm_fwd(const float* x, float* y, float* norm, int rows, int D, float eps, hmmc_stream_t stream);
int hmmc_l2norm_bwd(const float* dy, const float* y, const float* norm, float* dx, int rows, int D, hmmc_stream_t stream);

/* Hierarchical InfoNCE of BirdModel.forward (modules/modeling.py:665-672,702-709; CrossEn until_module.py:196-205) on
 * S_all [B, B*(1+F)] = 100 * Qn [Vn ; Un]^T (column c < B: video c; column B + b*F + f: frame f of video b):
 * loss = w_video * (CE(S) + CE(S^T)) + w_frame * sum_f (CE(S_f) + CE(S_f^T)), w_frame = weight_FTM / F.
 * lse_row [B][1+F] and lse_col [B*(1+F)] are saved for the backward; grad_out is a device scalar. */
int hmmc_infonce_fwd(const float* S, float* lse_row, float* lse_col, float* loss, int B, int F, float w_video, float w_frame,
                     hmmc_stream_t stream);
int hmmc_infonce_bwd(const float* S, const float* lse_row, const float* lse_col, const float* grad_out, float* dS, int B,
                     int F, float w_video, float w_frame, hmmc_stream_t stream);

/* Eval scorer (main_task_retrieval.py:332-336,512-513): out[i][b] = base[i][b] + mean(top-k over f of S_frame[i][b*F + f]). */
int hmmc_topk_mean(const float* S_frame, const float* base, float* out, int bq, int bv, int F, int k, long lds, long ldb,
                   hmmc_stream_t stream);

/* Rank metrics (metrics.py:12-39 compute_metrics): rank[q] = #{j : S(q, j) > S(q, target[q])}, i.e. the position of the
 * ground-truth item in the descending sort of query q's scores (first position on ties); target NULL = the diagonal.
 * S(q, j) = S[q*ld + j], or S[j*ld + q] with transposed != 0 (video -> text on the same matrix).  R@K, median and mean
 * rank follow from the Q integers on the host, as in the reference. */
int hmmc_retrieval_rank(const float* S, const int* target, int* rank, int Q, int V, long ld, int transposed,
                        hmmc_stream_t stream);
/* Video-to-text matrix of multi-sentence retrieval (metrics.py:79-86 tensor_video_to_text_sim): out[g][v] = max over the
 * sentences offsets[g] <= s < offsets[g+1] of S[s][v] (NaN counts as -inf); offsets is a device int32 [groups + 1].
 * Rank it with hmmc_retrieval_rank(transposed = 1). */
int hmmc_segment_max(const float* S, const int* offsets, float* out, int groups, int V, long ld, hmmc_stream_t stream);
/* Eval scorer (main_task_retrieval.py:321-357 _run_on_single_gpu; modules/modeling.py:207-229 loose_similarity):
 * hmmc_eval_slots(F) = rows per video of the packed candidate matrix (16 or 32; 0: F + 1 > 32, use hmmc_gemm_f32 +
 * hmmc_topk_mean).  hmmc_eval_pack writes packed [nv * slots][E]: slot 0 = visual[v] / |visual[v]|, slots 1..F =
 * frames[v][f] / |frames[v][f]|, the rest zero.  hmmc_eval_score computes, for every (query, video), the video logit
 * scale * q . v and the mean of the k largest frame logits scale * q . u_f in ONE pass (exact-f32 MFMA, top-k by wave
 * shuffles on the accumulators): the [queries, videos, F] tensor of the reference is never written.  Outputs are
 * [nq][nv] fp32; any of the three may be NULL (out_score = out_video + out_frame).  queries_unit rows have unit norm. */
int hmmc_eval_slots(int F);
int hmmc_eval_pack(const float* visual, const float* frames, float* packed, int nv, int F, int E, hmmc_stream_t stream);
int hmmc_eval_score(const float* queries_unit, const float* packed, float* out_video, float* out_frame, float* out_score,
                    int nq, int nv, int F, int E, int k, float scale, hmmc_stream_t stream);

/* video_emb[b] = mean_f (h + u)/||h + u||  (modules/module_cross.py:207-212); u may be NULL (use_temp False). */
int hmmc_temporal_pool_fwd(const float* h, const float* u, float* out, float* norms, int b, int F, int D,
                           hmmc_stream_t stream);
int hmmc_temporal_pool_bwd(const float* h, const float* u, const float* norms, const float* dout, float* dvf, int b, int F,
                           int D, hmmc_stream_t stream);
/* out[r] = x[r] + table[r % period]  (frame_position_embeddings, modules/module_cross.py:195-199). */
int hmmc_add_rowbias(const float* x, const float* table, float* out, long rows, int period, int D, hmmc_stream_t stream);
/* fp32 attention, head dim 64, probabilities saved [b, H, F, F]: the temporal blocks (modules/module_cross.py:127-131; F <= 64:
 * one wave per (video, head)) and, for 65 <= F <= 256, the CLIP towers in the reference's fp32-upcast regime (model.float():
 * ViT-B/16's 197 tokens, a 77-token text; a parity path: one workgroup per (sequence, head), exact fp32, not tuned). */
int hmmc_temporal_attention_fwd(const float* qkv, float* out, float* probs, int b, int F, int H, int causal,
                                hmmc_stream_t stream);
int hmmc_temporal_attention_bwd(const float* qkv, const float* probs, const float* dout, float* dqkv, int b, int F, int H,
                                hmmc_stream_t stream);

/* Multi-tensor kernels.  tab: int64 [T][8] = {p, g, m, v device pointers, numel, dtype (0 fp16, 1 fp32), group,
 * index of the tensor's first entry in `chunk`} (the chunk list is grouped by tensor, chunks ascending);
 * groups_host: HOST float [ngroups <= 32][8] = {scheduled lr, weight_decay, b1, b2, eps, max_grad_norm, 1-b1, 1-b2}, passed to the
 * kernel by value (tab[t][6] = group of tensor t): the per-step scalars need no device copy;
 * chunk: int32 [nchunks][2] = {tensor index, chunk index}, hmmc_mt_chunk_elems() elements per chunk;
 * sumsq: float [T + nchunks] scratch: per-chunk partial sums behind the T squared norms, which are formed by adding a tensor's
 * chunks in list order - no atomics, so norms, clip coefficient and updated weights are bit-identical from run to run. */
int hmmc_mt_chunk_elems(void);
int hmmc_mt_sumsq(const long* tab, const int* chunk, int nchunks, float* sumsq, int T, hmmc_stream_t stream);
/* torch.nn.utils.clip_grad_norm_(params, max_norm) (main_task_retrieval.py:291): out[0] = coefficient, out[1] = total norm. */
int hmmc_mt_clip_grad_norm(const long* tab, const int* chunk, int nchunks, float* sumsq, int T, float max_norm, float* out,
                           hmmc_stream_t stream);
/* BertAdam.step (modules/optimization.py:103-168) for every tensor, including its per-parameter clip. */
int hmmc_mt_bertadam(const long* tab, const float* groups_host, int ngroups, const int* chunk, int nchunks, float* sumsq,
                     int T, hmmc_stream_t stream);
/* The pair the training loop's `clip_grad_norm_(...); optimizer.step()` (main_task_retrieval.py:291-296) maps to when both see
 * the same gradients: the clip also leaves the squared norm of every gradient as it stands after scaling in
 * sumsq_after[0 .. T) (float [T + nchunks], formed inside the scaling pass), and the optimizer takes tensor t's norm for its
 * per-parameter clip from norms[index[t]] instead of reading every gradient once more. */
int hmmc_mt_clip_grad_norm_keep(const long* tab, const int* chunk, int nchunks, float* sumsq, int T, float max_norm, float* out,
                                float* sumsq_after, hmmc_stream_t stream);
int hmmc_mt_bertadam_ext(const long* tab, const float* groups_host, int ngroups, const int* chunk, int nchunks, int T,
                         const float* norms, const int* index, hmmc_stream_t stream);
/* _momentum_update (modules/modeling.py:238-242); tab rows = {p_k, p, 0, 0, numel, dtype}. */
int hmmc_mt_ema(const long* tab, const int* chunk, int nchunks, float momentum, float one_minus_momentum,
                hmmc_stream_t stream);
/* _dequeue_and_enqueue (modules/modeling.py:262-278): queue[:, col0:col0+R] = normalize(keys).T; queue is [E][W]. */
int hmmc_enqueue(const float* keys, float* queue, int R, int E, long W, long col0, hmmc_stream_t stream);

/* BatchNorm1d (train mode) of the MoCo projector / predictor MLPs (modules/modeling.py:788-807, SyncBN via :115-129).
 * hmmc_bn_stats: sums[0][n] = sum_m h, sums[1][n] = sum_m h^2 over THIS rank's rows (the caller all-reduces over ranks);
 * hmmc_bn_apply_relu: y = relu((h - mean) * rstd * gamma + beta);
 * hmmc_bn_bwd_reduce: sums[0] = sum d, sums[1] = sum d * xhat with d = dy * (y > 0);
 * hmmc_bn_bwd_apply: dh = gamma * rstd * (d - sums[0]*inv_n - xhat * sums[1]*inv_n), inv_n = 1 / global row count. */
/* hmmc_bn_finalize (one launch for the [N]-sized vector arithmetic between hmmc_bn_stats and hmmc_bn_apply_relu): mean = sums[0] / n,
 * var = max(sums[1] / n - mean^2, 0) (biased), rstd = 1 / sqrt(var + eps), n = *n_dev (device scalar: the row count summed over ranks,
 * SyncBatchNorm) or n_host when n_dev is NULL; with running_mean / running_var (both or neither) the train-mode update of
 * nn.BatchNorm1d: running = (1 - momentum) running + momentum x (mean | var n / (n - 1)), *num_batches_tracked += 1 (may be NULL). */
int hmmc_bn_finalize(const float* sums, const float* n_dev, float n_host, float eps, float momentum, float* mean, float* var,
                     float* rstd, float* running_mean, float* running_var, long* num_batches_tracked, int N, hmmc_stream_t stream);
size_t hmmc_bn_workspace(int M, int N);
int hmmc_bn_stats(const float* h, float* sums, int M, int N, void* workspace, size_t ws_bytes, hmmc_stream_t stream);
int hmmc_bn_apply_relu(const float* h, const float* mean, const float* rstd, const float* gamma, const float* beta, float* y,
                       long M, int N, hmmc_stream_t stream);
int hmmc_bn_bwd_reduce(const float* dy, const float* y, const float* h, const float* mean, const float* rstd, float* sums,
                       int M, int N, void* workspace, size_t ws_bytes, hmmc_stream_t stream);
int hmmc_bn_bwd_apply(const float* dy, const float* y, const float* h, const float* mean, const float* rstd,
                      const float* gamma, const float* sums, float* dh, long M, int N, float inv_n, hmmc_stream_t stream);

/* contrastive_loss against a negative queue (modules/modeling.py:286-313), for R stacked (q, k) rows sharing one queue:
 * lpos[r] = <qn[r], kn[r]> (hmmc_rowdot), S = qn . queue [R, Kq] (hmmc_gemm_f32),
 * loss = sum_r w * (logsumexp([lpos[r], S[r][:]] / T) - lpos[r] / T).  The backward overwrites S with dS. */
int hmmc_rowdot(const float* a, const float* b, float* out, int rows, int D, hmmc_stream_t stream);
int hmmc_moco_loss_fwd(const float* S, const float* lpos, float* lse, float* rowloss, float* loss, int R, long Kq,
                       float temperature, float w, hmmc_stream_t stream);
int hmmc_moco_loss_bwd(float* S, const float* lpos, const float* lse, const float* grad_out, float* dlpos, int R, long Kq,
                       float temperature, float w, hmmc_stream_t stream);
/* y[r][:] += s[r] * x[r][:] */
int hmmc_row_axpy(float* y, const float* s, const float* x, long rows, int D, hmmc_stream_t stream);

/* MLM head pieces: erf-GELU (modules/module_cross.py:33-39) and F.cross_entropy(ignore_index < 0)
 * (modules/modeling.py:171-179): loss = SUM over valid rows of (lse - logit[label]), count[0] = #valid rows;
 * the backward overwrites logits with (softmax - onehot) * grad_out / count on valid rows, 0 elsewhere. */
int hmmc_gelu_erf_fwd(const float* x, float* y, long n, hmmc_stream_t stream);
int hmmc_gelu_erf_bwd(const float* x, const float* dy, float* dx, long n, hmmc_stream_t stream);
int hmmc_ce_fwd(const float* logits, const long* labels, float* lse, float* rowloss, float* count, float* loss, int R, long V,
                hmmc_stream_t stream);
int hmmc_ce_bwd(float* logits, const long* labels, const float* lse, const float* grad_out, const float* count, int R, long V,
                hmmc_stream_t stream);

/* Native layer runtime: all ResidualAttentionBlocks of a tower in one call (modules/module_clip.py:231-268 fp16 CLIP
 * towers, fp32 = 0, eps 1e-5; modules/module_cross.py:114-149 fp32 temporal transformer, fp32 = 1, eps 1e-12).
 * params / grads: nlayers x 12 pointers in the order ln_1.{w,b}, attn.in_proj_{weight,bias}, attn.out_proj.{weight,bias},
 * ln_2.{w,b}, mlp.c_fc.{weight,bias}, mlp.c_proj.{weight,bias}.  acts: hmmc_tower_act_bytes() per layer when keep_acts
 * (training), one slab otherwise.  The backward needs hmmc_tower_bwd_scratch_bytes() of scratch; both need
 * hmmc_tower_workspace_bytes().  Only sequences this library's kernels on `stream`. */
size_t hmmc_tower_act_bytes(long tokens, int D, int nseq, int L, int heads, int fp32);
size_t hmmc_tower_bwd_scratch_bytes(long tokens, int D, int fp32);
size_t hmmc_tower_workspace_bytes(long tokens, int D, int nseq, int fp32, int bwd_layers);   /* bwd_layers: 0 for hmmc_tower_fwd, nlayers for hmmc_tower_bwd */
/* lead_only (fp16 towers, 0 / 1): the caller consumes only token 0 of every sequence of y - the ViT class token, the only
 * row VisualEncoder.encode_image keeps (modules/module_cross.py:228-230).  The last block's out_proj, ln_2 and MLP are
 * per-token, so they then run on the nseq leading rows alone (addressed in place at stride L*D); the other rows of y are
 * undefined, and hmmc_tower_bwd with the same flag reads only the leading rows of dy.  Loss and gradients are unchanged:
 * no gradient reaches the rows that are skipped. */
int hmmc_tower_fwd(const void* x, void* y, const void* const* params, void* acts, int keep_acts, int nseq, int L, int heads,
                   int D, int nlayers, int causal, float eps, int fp32, int lead_only, void* workspace, size_t ws_bytes,
                   hmmc_stream_t stream);
/* hmmc_tower_fwd for an fp16 tower with ln_1 / ln_2 folded into in_proj / c_fc (hmmc_gemm_f16_fold above): no LayerNorm pass over
 * the residual stream, the row statistics come out of the out_proj / c_proj epilogues.  keep_acts = 0 (eval, the momentum
 * encoders of modules/modeling.py:347-357): acts is ONE slab of hmmc_tower_act_bytes(), fold_ws hmmc_tower_fold_bytes(.., 0).
 * keep_acts = 1 (training): acts as hmmc_tower_act_bytes_fold describes, fold_ws hmmc_tower_fold_bytes(.., 1), BOTH go to hmmc_tower_bwd_fold;
 * needs D % 256 == 0, >= 2048 tokens, and last_exact when lead_only.  last_exact = 1: the LAST layer runs on the
 * unfolded kernels.  x_stat (may be NULL): the row pairs of x as hmmc_rowstat / hmmc_vit_embed_ln give them.  Returns
 * HMMC_ERR_UNSUPPORTED for operands of 2 GiB and more or shapes outside the above (use hmmc_tower_fwd). */
size_t hmmc_tower_fold_bytes(long tokens, int D, int nlayers, int train);
/* slab of a layer that runs folded with keep_acts = 1 (no ln_1 / ln_2 outputs are kept): acts of such a call = nfold of these
 * followed by (nlayers - nfold) slabs of hmmc_tower_act_bytes(), nfold = nlayers - (last_exact ? 1 : 0) */
size_t hmmc_tower_act_bytes_fold(long tokens, int D, int nseq, int L, int heads);
int hmmc_tower_fwd_fused(const void* x, const float* x_stat, void* y, const void* const* params, void* acts, int keep_acts, int nseq,
                         int L, int heads, int D, int nlayers, int causal, float eps, int lead_only, int last_exact, void* fold_ws,
                         size_t fold_bytes, hmmc_stream_t stream);
/* backward of hmmc_tower_fwd_fused(keep_acts = 1): hmmc_tower_bwd's arguments plus the forward's fold_ws and last_exact */
int hmmc_tower_bwd_fold(const void* dy, void* dx, const void* x0, const void* const* params, void* const* grads, const void* acts,
                        void* fold_ws, size_t fold_bytes, void* scratch, int nseq, int L, int heads, int D, int nlayers, int causal,
                        int lead_only, int last_exact, void* workspace, size_t ws_bytes, hmmc_stream_t wgrad_stream,
                        hmmc_stream_t stream);
/* wgrad_stream (optional, NULL = `stream`): a second stream for the weight-gradient GEMMs, which are leaves of the backward
 * pass; they then run beside the dgrad / LayerNorm / attention chain.  `stream` waits for it before the call's work is
 * complete in stream order, so callers keep single-stream semantics. */
int hmmc_tower_bwd(const void* dy, void* dx, const void* x0, const void* const* params, void* const* grads, const void* acts,
                   void* scratch, int nseq, int L, int heads, int D, int nlayers, int causal, int fp32, int lead_only,
                   void* workspace, size_t ws_bytes, hmmc_stream_t wgrad_stream, hmmc_stream_t stream);

/* Give back the seven events of the (stream, wgrad_stream) pair (process-wide state (3) of the header comment) - for callers
 * that create and destroy streams.  No hmmc_tower_bwd work of the pair may still be in flight.  Both NULL: every pair.
 * Returns the number of pairs released; the pair's next hmmc_tower_bwd call creates a fresh set. */
int hmmc_tower_release(hmmc_stream_t stream, hmmc_stream_t wgrad_stream);

/* A/B switches (process-wide, 0 / 1, all default 0), the only knobs besides hmmc_gemm_reserve_cus; the library itself reads
 * no environment variable (hmmc_amd/_lib.py maps HMMC_NO_WGRAD_GROUP etc. onto these when it loads the library):
 *   "no_wgrad_group"  one hmmc_gemm_f16 launch per weight gradient instead of hmmc_gemm_f16_wgrad_group (hmmc_tower_bwd;
 *                     hmmc_gemm_f16_wgrad_group_workspace then returns 0 and hmmc_tower_fwd_fused(keep_acts = 1) is unsupported)
 *   "no_f32_wavek"    hmmc_gemm_f32 never takes the wave-split-K kernel
 *   "no_f32_dma"      hmmc_gemm_f32 / hmmc_eval_score never take the LDS-DMA kernel
 *   "no_lead_attn"    a lead_only tower's last block runs its attention for every query
 * Results are bit-identical either way for no_lead_attn (class-token rows) and equal up to the order of the fp32 partial sums
 * (the K split of a weight gradient, the tile kernel of an fp32 GEMM) for the other three.  hmmc_set_option: HMMC_ERR_ARG for an unknown key.  hmmc_get_option: the value, or HMMC_ERR_ARG. */
int hmmc_set_option(const char* key, int value);
int hmmc_get_option(const char* key);

#ifdef __cplusplus
}
#endif
#endif /* HMMC_HIP_H */
