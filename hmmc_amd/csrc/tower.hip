// Native layer runtime: one C call runs every ResidualAttentionBlock of a tower, forward or backward
// (reference modules/module_clip.py:231-268 for the fp16 CLIP towers, modules/module_cross.py:114-149 for the
// fp32 temporal transformer).  It only SEQUENCES the kernels of this library on the caller's stream — ~20
// launches per layer forward, ~30 backward — so the Python host issues 4 calls per step instead of ~1000 and
// the launch stream stays ahead of the GPU even at 32 videos per GPU.  All memory is caller-owned:
//   params  : nlayers x 12 pointers  (ln_1.w, ln_1.b, in_proj.w, in_proj.b, out_proj.w, out_proj.b,
//                                      ln_2.w, ln_2.b, c_fc.w, c_fc.b, c_proj.w, c_proj.b)
//   acts    : one slab of hmmc_tower_act_bytes() per layer (saved for backward); layout below
//   scratch : hmmc_tower_bwd_scratch_bytes() for the backward's transient gradients
//   grads   : nlayers x 12 pointers, written (not accumulated)
#include "common.h"
#include "options.h"
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

extern "C" {
int hmmc_gemm_f16(const void*, const void*, void*, int, int, int, int, int, int, int, int, const void*, const void*, void*,
                  const void*, int, void*, size_t, hipStream_t);
size_t hmmc_gemm_f16_workspace(int, int, int);
int hmmc_gemm_f32(const float*, const float*, float*, int, int, int, long, long, long, long, int, float, const float*,
                  const float*, float*, const float*, int, hipStream_t);
int hmmc_layernorm_fwd(const void*, const float*, const float*, void*, float*, float*, const int*, int, int, long, float, int,
                       hipStream_t);
int hmmc_layernorm_bwd(const void*, const void*, const float*, const float*, const float*, const void*, void*, float*, float*,
                       void*, const int*, int, int, long, int, void*, size_t, hipStream_t);
size_t hmmc_layernorm_bwd_workspace(int, int);
int hmmc_layernorm_bwd_rows(int);
int hmmc_layernorm_bwd_partial(const void*, const void*, const float*, const float*, const float*, const void*, void*, float*, int,
                               const int*, int, int, long, int, hipStream_t);
int hmmc_multi_colreduce(const void*, int, hipStream_t);
int hmmc_colsum(const void*, void*, int, int, long, int, int, int, void*, size_t, hipStream_t);
size_t hmmc_colsum_workspace(int, int);
int hmmc_attention_f16_fwd(const void*, void*, float*, int, int, int, int, hipStream_t);
int hmmc_attention_f16_bwd(const void*, const void*, const float*, const void*, void*, float*, int, int, int, int, hipStream_t);
size_t hmmc_gemm_f16_colsum_rows(int, int, int);
size_t hmmc_gemm_f16_wgrad_group_workspace(const int*, const int*, int, int);
int hmmc_gemm_f16_wgrad_group(const void* const*, const void* const*, void* const*, float* const*, const int*, const int*, int, int,
                              void*, size_t, hipStream_t);
int hmmc_gemm_f16_fold(const void*, const void*, void*, int, int, int, int, int, int, int, const void*, const void*, void*, const void*,
                       int, const float*, const float*, float*, void*, size_t, hipStream_t);
int hmmc_ln_fold_prep(const void* const*, const float* const*, const float* const*, const void* const*, void* const*, float* const*,
                      const int*, int, int, hipStream_t);
int hmmc_rowstat(const void*, float*, int, int, long, float, hipStream_t);
int hmmc_rowstat_finalize(const float*, float*, int, int, int, float, hipStream_t);
int hmmc_attention_f16_fwd_lead(const void* qkv, void* out, float* lse, int nseq, int L, int H, int causal, hipStream_t stream);
int hmmc_attention_f16_bwd_lead(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv, float* dbias_partial,
                                const float* rowstat, int nseq, int L, int H, int causal, hipStream_t stream);
int hmmc_attention_f16_bwd_scaled(const void*, const void*, const float*, const void*, void*, float*, const float*, int, int, int, int,
                                  hipStream_t);
int hmmc_layernorm_bwd_fold_rows(int);
int hmmc_layernorm_bwd_fold(const void*, const void*, const float*, const void*, void*, float*, int, int, int, long, hipStream_t);
int hmmc_fold_grad_finish(const float* const*, const void* const*, const float* const*, const float* const*, const void* const*,
                          void* const*, float* const*, float* const*, float* const*, const int*, int, int, hipStream_t);
int hmmc_temporal_attention_fwd(const float*, float*, float*, int, int, int, int, hipStream_t);
int hmmc_temporal_attention_bwd(const float*, const float*, const float*, float*, int, int, int, hipStream_t);
}

namespace {

enum { EPI_BIAS = 1, EPI_RESID = 2, EPI_QGELU = 4, EPI_DGELU = 8, EPI_COLSUM = 32, EPI_SAVE_DGELU = 64, EPI_MULAUX = 128,
       EPI_LNFOLD = 256, EPI_ROWSTAT = 512, EPI_ROWSCALE = 1024 };

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

// activation slab of one layer (es = element size: 2 fp16 tower, 4 fp32 tower)
struct Acts {
  char *x, *ln1, *qkv, *att, *x1, *ln2, *h, *g;
  float *m1, *r1, *m2, *r2, *stat;       // stat: lse [nseq,H,L] (fp16 tower) or probs [nseq,H,L,L] (fp32 tower)
  size_t bytes;
};

// skip_ln: a layer that runs with its LayerNorms folded keeps no normalised activations (ln1 / ln2 are null)
Acts carve(char* base, long T, int D, int nseq, int L, int H, int es, bool f32, bool skip_ln = false) {
  Acts a;
  size_t td = al((size_t)T * D * es);
  char* p = base;
  a.x = p; p += td;
  a.ln1 = skip_ln ? nullptr : p; p += skip_ln ? 0 : td;
  a.qkv = p; p += al((size_t)T * 3 * D * es);
  a.att = p; p += td;
  a.x1 = p; p += td;
  a.ln2 = skip_ln ? nullptr : p; p += skip_ln ? 0 : td;
  a.h = p; p += al((size_t)T * 4 * D * es);
  a.g = p; p += al((size_t)T * 4 * D * es);
  a.m1 = (float*)p; p += al((size_t)T * 4);
  a.r1 = (float*)p; p += al((size_t)T * 4);
  a.m2 = (float*)p; p += al((size_t)T * 4);
  a.r2 = (float*)p; p += al((size_t)T * 4);
  a.stat = (float*)p; p += al((size_t)nseq * H * L * (f32 ? L : 1) * 4);
  a.bytes = (size_t)(p - base);
  return a;
}

inline int linear(bool f32, const void* x, const void* w, void* y, int M, int N, int K, const void* bias, const void* resid,
                  void* aux_out, int epi, void* ws, size_t wsb, hipStream_t s) {
  if (f32)
    return hmmc_gemm_f32((const float*)x, (const float*)w, (float*)y, M, N, K, K, 1, 1, K, N, 1.0f, (const float*)bias,
                         (const float*)resid, (float*)aux_out, nullptr, epi | (bias ? EPI_BIAS : 0) | (resid ? EPI_RESID : 0), s);
  return hmmc_gemm_f16(x, w, y, M, N, K, K, K, N, 1, 1, bias, resid, aux_out, nullptr,
                       epi | (bias ? EPI_BIAS : 0) | (resid ? EPI_RESID : 0), nullptr, 0, s);
}
// dx[M,K'] = dy[M,N'] w[N',K']
// csum (fp16 tower only): fp32 partial column sums of dx, hmmc_gemm_f16_colsum_rows(M, K', N') rows of K' floats
inline int dgrad(bool f32, const void* dy, const void* w, void* dx, int M, int Np, int Kp, const void* aux_in, int epi,
                 hipStream_t s, float* csum = nullptr, size_t csum_bytes = 0) {
  if (f32)
    return hmmc_gemm_f32((const float*)dy, (const float*)w, (float*)dx, M, Kp, Np, Np, 1, Kp, 1, Kp, 1.0f, nullptr, nullptr,
                         nullptr, (const float*)aux_in, epi, s);
  return hmmc_gemm_f16(dy, w, dx, M, Kp, Np, Np, Kp, Kp, 1, 0, nullptr, nullptr, nullptr, aux_in, epi | (csum ? EPI_COLSUM : 0),
                       csum, csum_bytes, s);
}
// dW[N',K'] = dy[T,N']^T x[T,K'];  ldy / ldx: row strides of dy and x (0 = dense)
inline int wgrad(bool f32, const void* dy, const void* x, void* dw, int T, int Np, int Kp, void* ws, size_t wsb, hipStream_t s,
                 int ldy = 0, int ldx = 0) {
  if (f32)
    return hmmc_gemm_f32((const float*)dy, (const float*)x, (float*)dw, Np, Kp, T, 1, Np, Kp, 1, Kp, 1.0f, nullptr, nullptr,
                         nullptr, nullptr, 0, s);
  size_t need = hmmc_gemm_f16_workspace(Np, Kp, T);
  return hmmc_gemm_f16(dy, x, dw, Np, Kp, T, ldy ? ldy : Np, ldx ? ldx : Kp, Kp, 0, 0, nullptr, nullptr, nullptr, nullptr, 0,
                       need <= wsb ? ws : nullptr, need <= wsb ? wsb : 0, s);
}

#define CK(call) do { int rc_ = (call); if (rc_ != 0) return rc_; } while (0)

// The attention half of a LAST block whose caller reads the class token alone (lead_only), sequences of at most 256 tokens: only
// query 0 of every sequence is wanted, so K | V are projected for every token, Q for the nseq class tokens only (their rows of
// ln1 / qkv addressed in place at stride L), and the attention runs for that query (hmmc_attention_f16_fwd_lead: bit-identical
// class rows).  The Q columns of the other rows of qkv, the other rows of att and the other entries of stat are NOT written.
inline bool lead_attention(bool lead, bool f32, int L) {
  return lead && !f32 && L <= 256 && !hmmc_option(HMMC_OPT_NO_LEAD_ATTN);      // A/B runs: all queries in the last block, as before round 4
}
inline int lead_inproj_attention(const void* ln1, const void* w_in, const void* b_in, void* qkv, void* att, float* stat, long T, int nseq,
                                 int L, int heads, int D, int causal, hipStream_t s) {
  const half_t* W = (const half_t*)w_in;
  const half_t* B = (const half_t*)b_in;
  CK(hmmc_gemm_f16(ln1, W + (size_t)D * D, (half_t*)qkv + D, (int)T, 2 * D, D, D, D, 3 * D, 1, 1, B + D, nullptr, nullptr, nullptr, EPI_BIAS,
                   nullptr, 0, s));
  CK(hmmc_gemm_f16(ln1, W, qkv, nseq, D, D, L * D, D, L * 3 * D, 1, 1, B, nullptr, nullptr, nullptr, EPI_BIAS, nullptr, 0, s));
  return hmmc_attention_f16_fwd_lead(qkv, att, stat, nseq, L, heads, causal, s);
}

}  // namespace

extern "C" size_t hmmc_tower_act_bytes(long tokens, int D, int nseq, int L, int heads, int fp32) {
  return carve(nullptr, tokens, D, nseq, L, heads, fp32 ? 4 : 2, fp32 != 0).bytes;
}
// slab of a layer hmmc_tower_fwd_fused(keep_acts = 1) runs folded (no ln_1 / ln_2 outputs): the acts buffer of such a call is
// nfold of these followed by (nlayers - nfold) of hmmc_tower_act_bytes(), nfold = nlayers - (last_exact ? 1 : 0)
extern "C" size_t hmmc_tower_act_bytes_fold(long tokens, int D, int nseq, int L, int heads) {
  return carve(nullptr, tokens, D, nseq, L, heads, 2, false, true).bytes;
}

// transient gradients of the backward: dh [T,4D], dqkv [T,3D], dln [T,D], dx1 [T,D], ping/pong dx [T,D] x2; the fp16 tower holds
// dh / dqkv / dx1 twice (layers alternate): a layer's grouped weight-gradient launch reads them while the next layer's chain
// already writes its own
static size_t group_ws_bytes(long tokens, int D);
// the second set exists only where a grouped launch can read one set while the next layer's chain writes the other
static bool two_sets(long tokens, int D, int fp32) { return !fp32 && group_ws_bytes(tokens, D) > 0; }
extern "C" size_t hmmc_tower_bwd_scratch_bytes(long tokens, int D, int fp32) {
  size_t es = fp32 ? 4 : 2;
  const size_t set = al((size_t)tokens * 4 * D * es) + al((size_t)tokens * 3 * D * es) + al((size_t)tokens * D * es);
  return (two_sets(tokens, D, fp32) ? 2 : 1) * set + 3 * al((size_t)tokens * D * es);
}

// fp32 partial sums of the fp16 tower's deferred reductions, one slot per layer (hmmc_tower_bwd reduces all of them in one
// launch at its end): LayerNorm 2 and LayerNorm 1 ([ln rows][3D]: dgamma | dbeta | column sums of dx), the c_fc bias from the
// QuickGELU' dgrad epilogue [colsum_rows][4D], the in_proj bias from the attention backward [nseq][3D]
struct ReduceTask { const float* partial; int R, N, seg; void* out[3]; int dtype[3]; };     // = HmmcReduceTask
struct PartSlot { size_t ln2, ln1, fc, attn, bytes; };
static PartSlot part_slot(long tokens, int D, int nseq) {
  PartSlot p;
  const size_t ln = al((size_t)hmmc_layernorm_bwd_rows((int)tokens) * 3 * D * sizeof(float));
  p.ln2 = 0; p.ln1 = ln;
  p.fc = 2 * ln;
  p.attn = p.fc + al(hmmc_gemm_f16_colsum_rows((int)tokens, 4 * D, D) * (size_t)4 * D * sizeof(float));
  p.bytes = p.attn + al((size_t)nseq * 3 * D * sizeof(float));
  return p;
}
static size_t partial_bytes(long tokens, int D, int nseq, int layers) { return part_slot(tokens, D, nseq).bytes * (size_t)(layers > 0 ? layers : 0); }

// the four weight gradients of a layer as one grouped launch: (c_proj, c_fc, out_proj, in_proj)
static void group_dims(int D, int (&Np)[4], int (&Kp)[4]) {
  Np[0] = D; Kp[0] = 4 * D; Np[1] = 4 * D; Kp[1] = D; Np[2] = D; Kp[2] = D; Np[3] = 3 * D; Kp[3] = D;
}
static size_t group_ws_bytes(long tokens, int D) {
  int Np[4], Kp[4];
  group_dims(D, Np, Kp);
  return hmmc_gemm_f16_wgrad_group_workspace(Np, Kp, 4, (int)tokens);
}

static size_t general_bytes(long tokens, int D, int fp32) {
  size_t w = hmmc_layernorm_bwd_workspace((int)tokens, D);
  size_t c = hmmc_colsum_workspace((int)tokens, 4 * D);
  if (c > w) w = c;
  if (!fp32) {
    size_t g = hmmc_gemm_f16_workspace(3 * D, D, (int)tokens);
    size_t g2 = hmmc_gemm_f16_workspace(4 * D, D, (int)tokens);
    size_t g3 = hmmc_gemm_f16_workspace(D, 4 * D, (int)tokens);
    size_t g4 = hmmc_gemm_f16_workspace(D, D, (int)tokens);
    size_t g5 = hmmc_gemm_f16_workspace(2 * D, D, (int)tokens);      // K | V rows of in_proj: a lead-only last block
    if (g2 > g) g = g2;
    if (g3 > g) g = g3;
    if (g4 > g) g = g4;
    if (g5 > g) g = g5;
    if (g > w) w = g;
    size_t grp = group_ws_bytes(tokens, D);          // the grouped launch takes this region when there is no second stream
    if (grp > w) w = grp;
  }
  return al(w);
}

static size_t wgrad_ws_bytes(long tokens, int D, int fp32) {
  if (fp32) return 0;
  const size_t grp = group_ws_bytes(tokens, D);
  size_t g = hmmc_gemm_f16_workspace(3 * D, D, (int)tokens);
  if (grp > g) g = grp;
  size_t g2 = hmmc_gemm_f16_workspace(4 * D, D, (int)tokens);
  size_t g3 = hmmc_gemm_f16_workspace(D, 4 * D, (int)tokens);
  size_t g4 = hmmc_gemm_f16_workspace(D, D, (int)tokens);
  size_t g5 = hmmc_gemm_f16_workspace(2 * D, D, (int)tokens);
  if (g2 > g) g = g2;
  if (g3 > g) g = g3;
  if (g4 > g) g = g4;
  if (g5 > g) g = g5;
  return al(g);
}

// [general: LayerNorm / column-sum / split-K slabs][fp16 tower backward: per-layer slots of deferred-reduction partials]
// [split-K slabs of the weight-gradient stream]
extern "C" size_t hmmc_tower_workspace_bytes(long tokens, int D, int nseq, int fp32, int bwd_layers) {
  return general_bytes(tokens, D, fp32) + (fp32 ? 0 : partial_bytes(tokens, D, nseq, bwd_layers)) + wgrad_ws_bytes(tokens, D, fp32);
}

// y = tower(x).  keep_acts: acts holds nlayers slabs (training); otherwise one slab is reused (key encoders, eval).
//
// lead_only (fp16 towers): the caller consumes only token 0 of every sequence of y (the ViT's class token: ln_post and proj
// are applied to that row alone, modules/module_cross.py:228-230).  Everything of the LAST block after its attention is
// per-token, so out_proj, ln_2 and the MLP run on the nseq leading rows only (read and written in place at stride L*D);
// the other rows of y are left undefined.  hmmc_tower_bwd with the same flag reads only those rows of dy.
extern "C" int hmmc_tower_fwd(const void* x, void* y, const void* const* params, void* acts, int keep_acts, int nseq, int L,
                              int heads, int D, int nlayers, int causal, float eps, int fp32, int lead_only, void* workspace,
                              size_t ws_bytes, hipStream_t s) {
  if (!x || !y || !params || !acts || nseq <= 0 || L <= 0 || heads <= 0 || nlayers <= 0 || D != heads * 64) return HMMC_ERR_ARG;
  if (lead_only && fp32) return HMMC_ERR_UNSUPPORTED;
  const bool f32 = fp32 != 0;
  const int es = f32 ? 4 : 2, dt = f32 ? 1 : 0;
  const long T = (long)nseq * L;
  const size_t slab = hmmc_tower_act_bytes(T, D, nseq, L, heads, fp32);
  const void* cur = x;
  for (int i = 0; i < nlayers; ++i) {
    const void* const* P = params + (size_t)i * 12;
    Acts a = carve((char*)acts + (keep_acts ? (size_t)i * slab : 0), T, D, nseq, L, heads, es, f32);
    const void* xin = cur;
    CK(hmmc_layernorm_fwd(xin, (const float*)P[0], (const float*)P[1], a.ln1, a.m1, a.r1, nullptr, (int)T, D, D, eps, dt, s));
    if (lead_attention(lead_only && i + 1 == nlayers, f32, L)) {
      CK(lead_inproj_attention(a.ln1, P[2], P[3], a.qkv, a.att, a.stat, T, nseq, L, heads, D, causal, s));
    } else {
      CK(linear(f32, a.ln1, P[2], a.qkv, (int)T, 3 * D, D, P[3], nullptr, nullptr, 0, workspace, ws_bytes, s));
      if (f32) CK(hmmc_temporal_attention_fwd((const float*)a.qkv, (float*)a.att, a.stat, nseq, L, heads, causal, s));
      else CK(hmmc_attention_f16_fwd(a.qkv, a.att, a.stat, nseq, L, heads, causal, s));
    }
    // output of this layer: next layer's saved input slot, or y for the last layer.  Without saved activations one
    // slab is reused: the output alternates between the x and h slots (h is dead once c_proj has read g, and the
    // next layer's input is dead before its own c_fc rewrites h).
    void* out = y;
    if (i + 1 < nlayers) out = keep_acts ? carve((char*)acts + (size_t)(i + 1) * slab, T, D, nseq, L, heads, es, f32).x
                                         : (void*)(((i & 1) == 0) ? a.x : a.h);
    const int save_epi = EPI_QGELU | ((!f32 && keep_acts) ? EPI_SAVE_DGELU : 0);
    if (lead_only && i + 1 == nlayers) {
      // leading rows only: x1 and y are addressed in place (row n*L of the [T, D] buffers), ln2 / g / h are compact [nseq, .]
      const int ldl = L * D;
      CK(hmmc_gemm_f16(a.att, P[4], a.x1, nseq, D, D, ldl, D, ldl, 1, 1, P[5], xin, nullptr, nullptr, EPI_BIAS | EPI_RESID,
                       nullptr, 0, s));
      CK(hmmc_layernorm_fwd(a.x1, (const float*)P[6], (const float*)P[7], a.ln2, a.m2, a.r2, nullptr, nseq, D, ldl, eps, dt, s));
      CK(linear(f32, a.ln2, P[8], a.g, nseq, 4 * D, D, P[9], nullptr, keep_acts ? a.h : nullptr, save_epi, workspace, ws_bytes, s));
      CK(hmmc_gemm_f16(a.g, P[10], out, nseq, D, 4 * D, 4 * D, 4 * D, ldl, 1, 1, P[11], a.x1, nullptr, nullptr,
                       EPI_BIAS | EPI_RESID, nullptr, 0, s));
      cur = out;
      continue;
    }
    CK(linear(f32, a.att, P[4], a.x1, (int)T, D, D, P[5], xin, nullptr, 0, workspace, ws_bytes, s));
    CK(hmmc_layernorm_fwd(a.x1, (const float*)P[6], (const float*)P[7], a.ln2, a.m2, a.r2, nullptr, (int)T, D, D, eps, dt, s));
    // fp16 tower: the `h` slot receives QuickGELU'(pre-activation), which is all the backward needs of it
    CK(linear(f32, a.ln2, P[8], a.g, (int)T, 4 * D, D, P[9], nullptr, keep_acts ? a.h : nullptr, save_epi, workspace, ws_bytes, s));
    CK(linear(f32, a.g, P[10], out, (int)T, D, 4 * D, P[11], a.x1, nullptr, 0, workspace, ws_bytes, s));
    cur = out;
  }
  return HMMC_OK;
}

// ---- forward with the LayerNorms folded into in_proj / c_fc (fp16 towers, no saved activations) ----------------------------
// Per layer: [3D + 4D][D] fp16 folded weights + [2][3D + 4D] fp32 column terms; shared: the row pairs [T][2] and the
// per-64-column partial statistics [D / 64][T][2] of the residual stream (see ln_fold.hip).
namespace {
// s1 / s2 / vm (training only): fp32 weight-gradient sums of in_proj / c_fc against the raw rows and the row-mean scratch of
// hmmc_fold_grad_finish
struct FoldLayer { char *w1, *w2; float *cd1, *cd2, *s1, *s2, *vm; };
struct FoldWs { float *stat, *part; size_t per_layer, bytes; };
FoldWs fold_carve(char* base, long T, int D, int nlayers, int train, FoldLayer* out) {
  FoldWs f;
  const size_t w1 = al((size_t)3 * D * D * 2), w2 = al((size_t)4 * D * D * 2), c1 = al((size_t)2 * 3 * D * 4), c2 = al((size_t)2 * 4 * D * 4);
  const size_t g1 = train ? al((size_t)3 * D * D * 4) : 0, g2 = train ? al((size_t)4 * D * D * 4) : 0, vm = train ? al((size_t)(7 * D + 2 * 2 * 8 * D) * 4) : 0;      // 2 x hmmc_fold_grad_scratch_floats
  f.per_layer = w1 + w2 + c1 + c2 + g1 + g2 + vm;
  char* p = base;
  for (int i = 0; i < nlayers; ++i) {
    if (out) {
      out[i].w1 = p; out[i].w2 = p + w1; out[i].cd1 = (float*)(p + w1 + w2); out[i].cd2 = (float*)(p + w1 + w2 + c1);
      out[i].s1 = (float*)(p + w1 + w2 + c1 + c2); out[i].s2 = (float*)(p + w1 + w2 + c1 + c2 + g1);
      out[i].vm = (float*)(p + w1 + w2 + c1 + c2 + g1 + g2);
    }
    p += f.per_layer;
  }
  f.stat = (float*)p; p += al((size_t)T * 2 * 4);
  f.part = (float*)p; p += al((size_t)(D / 64) * T * 2 * 4);
  f.bytes = (size_t)(p - base);
  return f;
}
// layers of a training call that run folded: all but the last when the caller keeps that one on the unfolded kernels
inline int nfold_of(int nlayers, int last_exact) { return last_exact ? nlayers - 1 : nlayers; }
}  // namespace

// train = 1: the workspace also holds what hmmc_tower_bwd_fold needs (fp32 weight-gradient sums); it must then live from the
// forward call to the backward call (the folded weights in it are operands of the backward's data gradients)
extern "C" size_t hmmc_tower_fold_bytes(long tokens, int D, int nlayers, int train) {
  return fold_carve(nullptr, tokens, D, nlayers, train, nullptr).bytes;
}

// y = tower(x) for an fp16 tower with ln_1 / ln_2 folded into in_proj / c_fc: the same blocks as hmmc_tower_fwd
// (modules/module_clip.py:231-257) without the four LayerNorm passes over the residual stream per layer.
// keep_acts = 0 (eval, momentum encoders): acts is ONE slab of hmmc_tower_act_bytes(), fold_ws = hmmc_tower_fold_bytes(.., 0).
// keep_acts = 1 (training; hmmc_tower_bwd_fold takes the same acts and fold_ws): acts holds nlayers slabs, the folded layers'
// row statistics are kept in their slabs' mean / rstd slots, c_fc also writes QuickGELU'(h).  last_exact = 1 runs the LAST
// layer on the unfolded kernels (with lead_only that layer works on 1 / L of the rows: nothing to fold away, and the class-token
// rows stay bit-identical between a lead_only and an all-token call).  x_stat (optional): the row pairs (rstd, -rstd mean)
// [tokens][2] of x when its producer already has them (hmmc_vit_embed_ln).
extern "C" int hmmc_tower_fwd_fused(const void* x, const float* x_stat, void* y, const void* const* params, void* acts, int keep_acts,
                                    int nseq, int L, int heads, int D, int nlayers, int causal, float eps, int lead_only, int last_exact,
                                    void* fold_ws, size_t fold_bytes, hipStream_t s) {
  if (!x || !y || !params || !acts || !fold_ws || nseq <= 0 || L <= 0 || heads <= 0 || nlayers <= 0 || D != heads * 64) return HMMC_ERR_ARG;
  const long T = (long)nseq * L;
  if ((uint64_t)(T + 256) * 4 * D * 2 >= (1ull << 31) - (1ull << 24)) return HMMC_ERR_UNSUPPORTED;     // operands of 2 GiB: hmmc_tower_fwd
  // training: the folded weight gradients exist for the grouped launch, and a lead-only last layer must be the unfolded one
  if (keep_acts && (L > 256 || group_ws_bytes(T, D) == 0 || (lead_only && !last_exact))) return HMMC_ERR_UNSUPPORTED;
  const int nfold = nfold_of(nlayers, last_exact);
  std::vector<FoldLayer> fl(nlayers);
  const FoldWs fw = fold_carve((char*)fold_ws, T, D, nlayers, keep_acts, fl.data());
  if (fold_bytes < fw.bytes) return HMMC_ERR_WORKSPACE;
  const size_t slab = hmmc_tower_act_bytes(T, D, nseq, L, heads, 0), slab_f = hmmc_tower_act_bytes_fold(T, D, nseq, L, heads);
  // layer i's saved activations (training): folded layers first, in the compact slab
  auto acts_of = [&](int i) {
    if (!keep_acts) return carve((char*)acts, T, D, nseq, L, heads, 2, false);
    const size_t off = i < nfold ? (size_t)i * slab_f : (size_t)nfold * slab_f + (size_t)(i - nfold) * slab;
    return carve((char*)acts + off, T, D, nseq, L, heads, 2, false, i < nfold);
  };
  // folded weights and column terms of every folded layer (16 layers = 32 matrices per launch)
  for (int l0 = 0; l0 < nfold; l0 += 16) {
    const void* W[32]; const float* gm[32]; const float* bt[32]; const void* bs[32]; void* Wf[32]; float* cd[32]; int N[32];
    int n = 0;
    for (int i = l0; i < nfold && i < l0 + 16; ++i) {
      const void* const* P = params + (size_t)i * 12;
      W[n] = P[2]; gm[n] = (const float*)P[0]; bt[n] = (const float*)P[1]; bs[n] = P[3]; Wf[n] = fl[i].w1; cd[n] = fl[i].cd1; N[n] = 3 * D; ++n;
      W[n] = P[8]; gm[n] = (const float*)P[6]; bt[n] = (const float*)P[7]; bs[n] = P[9]; Wf[n] = fl[i].w2; cd[n] = fl[i].cd2; N[n] = 4 * D; ++n;
    }
    if (n) CK(hmmc_ln_fold_prep(W, gm, bt, bs, Wf, cd, N, D, n, s));
  }
  const int nparts = D / 64;
  const void* cur = x;
  for (int i = 0; i < nlayers; ++i) {
    const void* const* P = params + (size_t)i * 12;
    Acts a = acts_of(i);
    const void* xin = cur;
    void* out = y;
    if (i + 1 < nlayers) out = keep_acts ? (void*)acts_of(i + 1).x : (void*)(((i & 1) == 0) ? a.x : a.h);
    const bool lead = lead_only && i + 1 == nlayers;
    const int ldl = L * D;
    if (i >= nfold) {
      // the last layer on the unfolded kernels (as hmmc_tower_fwd)
      const int save_epi = EPI_QGELU | (keep_acts ? EPI_SAVE_DGELU : 0);
      const int R = lead ? nseq : (int)T;                 // rows of the per-token half
      const int ldr = lead ? ldl : D;                     // their stride in the [T, D] buffers
      CK(hmmc_layernorm_fwd(xin, (const float*)P[0], (const float*)P[1], a.ln1, a.m1, a.r1, nullptr, (int)T, D, D, eps, 0, s));
      if (lead_attention(lead, false, L)) {
        CK(lead_inproj_attention(a.ln1, P[2], P[3], a.qkv, a.att, a.stat, T, nseq, L, heads, D, causal, s));
      } else {
        CK(hmmc_gemm_f16(a.ln1, P[2], a.qkv, (int)T, 3 * D, D, D, D, 3 * D, 1, 1, P[3], nullptr, nullptr, nullptr, EPI_BIAS, nullptr, 0, s));
        CK(hmmc_attention_f16_fwd(a.qkv, a.att, a.stat, nseq, L, heads, causal, s));
      }
      CK(hmmc_gemm_f16(a.att, P[4], a.x1, R, D, D, ldr, D, ldr, 1, 1, P[5], xin, nullptr, nullptr, EPI_BIAS | EPI_RESID, nullptr, 0, s));
      CK(hmmc_layernorm_fwd(a.x1, (const float*)P[6], (const float*)P[7], a.ln2, a.m2, a.r2, nullptr, R, D, ldr, eps, 0, s));
      CK(hmmc_gemm_f16(a.ln2, P[8], a.g, R, 4 * D, D, D, D, 4 * D, 1, 1, P[9], nullptr, keep_acts ? a.h : nullptr, nullptr,
                       EPI_BIAS | save_epi, nullptr, 0, s));
      CK(hmmc_gemm_f16(a.g, P[10], out, R, D, 4 * D, 4 * D, 4 * D, ldr, 1, 1, P[11], a.x1, nullptr, nullptr, EPI_BIAS | EPI_RESID, nullptr, 0, s));
      cur = out;
      continue;
    }
    // row pairs of this layer's input and of x1: in the slab (training: the backward reads them) or in the shared buffer
    float* const stat1 = keep_acts ? a.m1 : fw.stat;
    float* const stat2 = keep_acts ? a.m2 : fw.stat;
    const float* st_in = stat1;
    if (i == 0) {
      if (x_stat && !keep_acts) st_in = x_stat;
      else if (x_stat) { if (hipMemcpyAsync(stat1, x_stat, (size_t)T * 8, hipMemcpyDeviceToDevice, s) != hipSuccess) return HMMC_ERR_LAUNCH; }
      else CK(hmmc_rowstat(x, stat1, (int)T, D, D, eps, s));
    }
    if (!lead_attention(lead, false, L)) {
      CK(hmmc_gemm_f16_fold(xin, fl[i].w1, a.qkv, (int)T, 3 * D, D, D, D, 3 * D, 1, nullptr, nullptr, nullptr, nullptr, EPI_LNFOLD, st_in, fl[i].cd1,
                            nullptr, nullptr, 0, s));
      CK(hmmc_attention_f16_fwd(a.qkv, a.att, a.stat, nseq, L, heads, causal, s));
    } else {
      // A folded lead-only last block (passes without saved activations): K | V for every token, Q for the class rows, as
      // lead_inproj_attention.  The folded GEMM takes its column terms as one [2][N] array and its row pairs per GEMM row, so the
      // two sub-problems get copies: (c | d) of columns D..3D and of columns 0..D, and the class rows' pairs gathered from
      // stride L - into the partial-statistics buffer, which is idle until this block's out_proj writes it.
      float* const cd_kv = fw.part;                       // [2][2D]
      float* const cd_q = cd_kv + 4 * D;                  // [2][D]
      float* const st_q = cd_q + 2 * D;                   // [nseq][2]
      const size_t fb = sizeof(float);
      if ((size_t)(6 * D + 2 * nseq) > (size_t)nparts * T * 2) return HMMC_ERR_WORKSPACE;
      if (hipMemcpyAsync(cd_kv, fl[i].cd1 + D, 2 * D * fb, hipMemcpyDeviceToDevice, s) != hipSuccess ||
          hipMemcpyAsync(cd_kv + 2 * D, fl[i].cd1 + 3 * D + D, 2 * D * fb, hipMemcpyDeviceToDevice, s) != hipSuccess ||
          hipMemcpyAsync(cd_q, fl[i].cd1, D * fb, hipMemcpyDeviceToDevice, s) != hipSuccess ||
          hipMemcpyAsync(cd_q + D, fl[i].cd1 + 3 * D, D * fb, hipMemcpyDeviceToDevice, s) != hipSuccess ||
          hipMemcpy2DAsync(st_q, 2 * fb, st_in, (size_t)L * 2 * fb, 2 * fb, nseq, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return HMMC_ERR_LAUNCH;
      const half_t* const w1 = (const half_t*)fl[i].w1;
      CK(hmmc_gemm_f16_fold(xin, w1 + (size_t)D * D, (half_t*)a.qkv + D, (int)T, 2 * D, D, D, D, 3 * D, 1, nullptr, nullptr, nullptr, nullptr,
                            EPI_LNFOLD, st_in, cd_kv, nullptr, nullptr, 0, s));
      CK(hmmc_gemm_f16_fold(xin, w1, a.qkv, nseq, D, D, ldl, D, L * 3 * D, 1, nullptr, nullptr, nullptr, nullptr, EPI_LNFOLD, st_q, cd_q, nullptr,
                            nullptr, 0, s));
      CK(hmmc_attention_f16_fwd_lead(a.qkv, a.att, a.stat, nseq, L, heads, causal, s));
    }
    const int fc_epi = EPI_LNFOLD | EPI_QGELU | (keep_acts ? EPI_SAVE_DGELU : 0);
    void* const fc_aux = keep_acts ? a.h : nullptr;
    if (lead) {
      // the last block's per-token half on the leading rows only, as hmmc_tower_fwd: the same folded arithmetic on rows
      // addressed in place at stride L*D (so a class-token row comes out bit-identical to the all-token pass)
      CK(hmmc_gemm_f16_fold(a.att, P[4], a.x1, nseq, D, D, ldl, D, ldl, 1, P[5], xin, nullptr, nullptr, EPI_BIAS | EPI_RESID | EPI_ROWSTAT, nullptr,
                            nullptr, fw.part, nullptr, 0, s));
      CK(hmmc_rowstat_finalize(fw.part, stat2, nparts, nseq, D, eps, s));
      CK(hmmc_gemm_f16_fold(a.x1, fl[i].w2, a.g, nseq, 4 * D, D, ldl, D, 4 * D, 1, nullptr, nullptr, fc_aux, nullptr, fc_epi, stat2, fl[i].cd2, nullptr,
                            nullptr, 0, s));
      CK(hmmc_gemm_f16(a.g, P[10], out, nseq, D, 4 * D, 4 * D, 4 * D, ldl, 1, 1, P[11], a.x1, nullptr, nullptr, EPI_BIAS | EPI_RESID, nullptr, 0, s));
      cur = out;
      continue;
    }
    CK(hmmc_gemm_f16_fold(a.att, P[4], a.x1, (int)T, D, D, D, D, D, 1, P[5], xin, nullptr, nullptr, EPI_BIAS | EPI_RESID | EPI_ROWSTAT, nullptr, nullptr,
                          fw.part, nullptr, 0, s));
    CK(hmmc_rowstat_finalize(fw.part, stat2, nparts, (int)T, D, eps, s));
    CK(hmmc_gemm_f16_fold(a.x1, fl[i].w2, a.g, (int)T, 4 * D, D, D, D, 4 * D, 1, nullptr, nullptr, fc_aux, nullptr, fc_epi, stat2, fl[i].cd2, nullptr,
                          nullptr, 0, s));
    const bool more = i + 1 < nfold;                      // the next layer is folded too: it needs the row pairs of this output
    CK(hmmc_gemm_f16_fold(a.g, P[10], out, (int)T, D, 4 * D, 4 * D, 4 * D, D, 1, P[11], a.x1, nullptr, nullptr,
                          EPI_BIAS | EPI_RESID | (more ? EPI_ROWSTAT : 0), nullptr, nullptr, more ? fw.part : nullptr, nullptr, 0, s));
    if (more) {
      float* const next = keep_acts ? acts_of(i + 1).m1 : fw.stat;
      CK(hmmc_rowstat_finalize(fw.part, next, nparts, (int)T, D, eps, s));
    }
    cur = out;
  }
  return HMMC_OK;
}

// dx = d tower / d x (dy given), grads[nlayers*12] written.  x0 is the tower input given to hmmc_tower_fwd.
namespace {
// Events between the main and the weight-gradient stream ("operand ready" and back, "operand no longer read"): one set per
// (main stream, weight-gradient stream) pair (seven: operand ready, four "operand no longer read", two for the grouped launches
// of even / odd layers), created on the pair's FIRST hmmc_tower_bwd call and kept for the life of the
// process (hipEventCreate is the only allocation the library ever makes besides that of hmmc_gemm_profile_start; a
// recorded event may be recorded again - a wait already enqueued keeps the record it was enqueued against).
struct WgradSync {
  hipEvent_t ready = nullptr, done[4] = {nullptr, nullptr, nullptr, nullptr}, gdone[2] = {nullptr, nullptr};
  bool ok = false;
  WgradSync() {
    ok = hipEventCreateWithFlags(&ready, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < 4; ++k) ok = ok && hipEventCreateWithFlags(&done[k], hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < 2; ++k) ok = ok && hipEventCreateWithFlags(&gdone[k], hipEventDisableTiming) == hipSuccess;
  }
};
std::mutex g_sync_mu;
std::map<std::pair<hipStream_t, hipStream_t>, WgradSync*> g_sync;
WgradSync* wgrad_sync_for(hipStream_t s, hipStream_t sw) {
  std::lock_guard<std::mutex> lk(g_sync_mu);
  WgradSync*& e = g_sync[{s, sw}];
  if (!e) e = new WgradSync();
  return e;
}
}  // namespace

// Give back the seven events of a (stream, weight-gradient stream) pair - for callers that create and destroy streams (a
// handle value may be reused by a later stream; the pair's next hmmc_tower_bwd call simply creates a fresh set).  The caller
// guarantees that no hmmc_tower_bwd work of the pair is still in flight (hipEventDestroy of a recorded, not yet completed
// event is legal - the runtime defers the release - but a call enqueued concurrently from another thread would race).
// stream == wgrad_stream == NULL releases every pair.  Returns the number of pairs released.
extern "C" int hmmc_tower_release(hipStream_t stream, hipStream_t wgrad_stream) {
  std::lock_guard<std::mutex> lk(g_sync_mu);
  int n = 0;
  for (auto it = g_sync.begin(); it != g_sync.end();) {
    const bool all = !stream && !wgrad_stream;
    if (all || (it->first.first == stream && it->first.second == wgrad_stream)) {
      WgradSync* e = it->second;
      if (e) {
        if (e->ready) (void)hipEventDestroy(e->ready);
        for (int k = 0; k < 4; ++k) if (e->done[k]) (void)hipEventDestroy(e->done[k]);
        for (int k = 0; k < 2; ++k) if (e->gdone[k]) (void)hipEventDestroy(e->gdone[k]);
        delete e;
      }
      it = g_sync.erase(it);
      ++n;
    } else {
      ++it;
    }
  }
  return n;
}

// The library's A/B switches (options.h).  key: "no_wgrad_group", "no_f32_wavek", "no_f32_dma", "no_lead_attn"; value 0 / 1.
// hmmc_set_option returns HMMC_ERR_ARG for an unknown key; hmmc_get_option returns the value or HMMC_ERR_ARG.
std::atomic<int> g_hmmc_options[HMMC_OPT_COUNT];
namespace {
int option_index(const char* key) {
  static const char* const names[HMMC_OPT_COUNT] = {"no_wgrad_group", "no_f32_wavek", "no_f32_dma", "no_lead_attn"};
  if (!key) return -1;
  for (int i = 0; i < HMMC_OPT_COUNT; ++i)
    if (std::strcmp(key, names[i]) == 0) return i;
  return -1;
}
}  // namespace
extern "C" int hmmc_set_option(const char* key, int value) {
  const int i = option_index(key);
  if (i < 0) return HMMC_ERR_ARG;
  g_hmmc_options[i].store(value != 0, std::memory_order_relaxed);
  return HMMC_OK;
}
extern "C" int hmmc_get_option(const char* key) {
  const int i = option_index(key);
  return i < 0 ? HMMC_ERR_ARG : g_hmmc_options[i].load(std::memory_order_relaxed);
}

// wgrad_stream (optional): the four weight-gradient GEMMs of every layer are leaves of the backward pass; given a second
// stream they run there, beside the dgrad / LayerNorm / attention chain on `s`, and fill the CUs the persistent GEMMs of
// that chain leave idle in their partial last rounds.  `s` waits for the stream before this call returns control of the
// gradients (stream order), so the caller sees ordinary single-stream semantics.
namespace {
struct FoldBwd { const FoldLayer* fl; int nfold; };     // layers [0, nfold) were run folded by hmmc_tower_fwd_fused(keep_acts = 1)
}
static int tower_bwd_impl(const void* dy, void* dx, const void* x0, const void* const* params, void* const* grads,
                          const void* acts, void* scratch, int nseq, int L, int heads, int D, int nlayers, int causal,
                          int fp32, int lead_only, void* workspace, size_t ws_bytes, hipStream_t wgrad_stream, hipStream_t s,
                          const FoldBwd* fb);

extern "C" int hmmc_tower_bwd(const void* dy, void* dx, const void* x0, const void* const* params, void* const* grads,
                              const void* acts, void* scratch, int nseq, int L, int heads, int D, int nlayers, int causal,
                              int fp32, int lead_only, void* workspace, size_t ws_bytes, hipStream_t wgrad_stream, hipStream_t s) {
  return tower_bwd_impl(dy, dx, x0, params, grads, acts, scratch, nseq, L, heads, D, nlayers, causal, fp32, lead_only, workspace, ws_bytes,
                        wgrad_stream, s, nullptr);
}

// The backward of hmmc_tower_fwd_fused(keep_acts = 1): same arguments as hmmc_tower_bwd plus the forward's fold_ws and
// last_exact.  In a folded layer the gradients that reach a LayerNorm arrive scaled by the row's rstd (QuickGELU' dgrad
// epilogue HMMC_EPI_ROWSCALE, hmmc_attention_f16_bwd_scaled), the data gradients go through the folded weights, the
// LayerNorm backward needs no gamma (hmmc_layernorm_bwd_fold), the in_proj / c_fc weight gradients are taken against the raw
// residual stream into fp32 and finished - together with dgamma / dbeta of ln_1 / ln_2 - by ONE hmmc_fold_grad_finish at the
// end (ln_fold.hip has the algebra).  Needs the grouped weight-gradient path (D % 256 == 0, >= 2048 tokens).
extern "C" int hmmc_tower_bwd_fold(const void* dy, void* dx, const void* x0, const void* const* params, void* const* grads,
                                   const void* acts, void* fold_ws, size_t fold_bytes, void* scratch, int nseq, int L, int heads, int D,
                                   int nlayers, int causal, int lead_only, int last_exact, void* workspace, size_t ws_bytes,
                                   hipStream_t wgrad_stream, hipStream_t s) {
  if (!fold_ws || nseq <= 0 || L <= 0 || nlayers <= 0) return HMMC_ERR_ARG;
  const long T = (long)nseq * L;
  if (L > 256 || group_ws_bytes(T, D) == 0 || (lead_only && !last_exact)) return HMMC_ERR_UNSUPPORTED;
  std::vector<FoldLayer> fl(nlayers);
  const FoldWs fw = fold_carve((char*)fold_ws, T, D, nlayers, 1, fl.data());
  if (fold_bytes < fw.bytes) return HMMC_ERR_WORKSPACE;
  FoldBwd fb{fl.data(), nfold_of(nlayers, last_exact)};
  return tower_bwd_impl(dy, dx, x0, params, grads, acts, scratch, nseq, L, heads, D, nlayers, causal, 0, lead_only, workspace, ws_bytes,
                        wgrad_stream, s, &fb);
}

static int tower_bwd_impl(const void* dy, void* dx, const void* x0, const void* const* params, void* const* grads,
                          const void* acts, void* scratch, int nseq, int L, int heads, int D, int nlayers, int causal,
                          int fp32, int lead_only, void* workspace, size_t ws_bytes, hipStream_t wgrad_stream, hipStream_t s,
                          const FoldBwd* fb) {
  if (!dy || !dx || !x0 || !params || !grads || !acts || !scratch || nseq <= 0 || L <= 0 || nlayers <= 0 || D != heads * 64)
    return HMMC_ERR_ARG;
  if (lead_only && fp32) return HMMC_ERR_UNSUPPORTED;
  const bool f32 = fp32 != 0;
  const int es = f32 ? 4 : 2, dt = f32 ? 1 : 0;
  const long T = (long)nseq * L;
  const size_t slab = hmmc_tower_act_bytes(T, D, nseq, L, heads, fp32);
  char* sp = (char*)scratch;
  void* dh_set[2]; void* dqkv_set[2]; void* dx1_set[2];
  const bool sets2 = two_sets(T, D, fp32);
  for (int k = 0; k < (sets2 ? 2 : 1); ++k) {
    dh_set[k] = sp; sp += al((size_t)T * 4 * D * es);
    dqkv_set[k] = sp; sp += al((size_t)T * 3 * D * es);
    dx1_set[k] = sp; sp += al((size_t)T * D * es);
  }
  if (!sets2) { dh_set[1] = dh_set[0]; dqkv_set[1] = dqkv_set[0]; dx1_set[1] = dx1_set[0]; }
  void* dln = sp; sp += al((size_t)T * D * es);
  void* ping[2];
  ping[0] = sp; sp += al((size_t)T * D * es);
  ping[1] = sp;
  // grouped weight gradients (fp16 tower, shapes the 256x256 tile takes): one launch per layer once the layer's last
  // gradient operand (dqkv) exists, on the weight-gradient stream; gdone[parity] orders the reuse of the operand set
  const bool grouped = !f32 && group_ws_bytes(T, D) > 0;
  // workspace = [general][partials]; the fp16 tower takes three of its four bias gradients from the kernels that
  // produce the tensors (LayerNorm backward: out_proj / c_proj; QuickGELU' dgrad epilogue: c_fc; attention backward:
  // in_proj) instead of re-reading them.  Only the c_proj bias of the LAST layer still needs its own pass over dy.
  const size_t gen = general_bytes(T, D, fp32);
  const size_t part_bytes = f32 ? 0 : partial_bytes(T, D, nseq, nlayers);
  const size_t wws_bytes = wgrad_ws_bytes(T, D, fp32);
  if (ws_bytes < gen + part_bytes + wws_bytes) return HMMC_ERR_WORKSPACE;
  const PartSlot slot = f32 ? PartSlot{} : part_slot(T, D, nseq);
  char* const part_base = (char*)workspace + gen;
  void* wws = (char*)workspace + gen + part_bytes;
  // second-stage reductions of the fp16 tower, all in one launch at the end (none of their results is read before)
  std::vector<ReduceTask> tasks;
  auto defer = [&](const float* partial, int R, int N, int seg, void* o0, int d0, void* o1 = nullptr, int d1 = 1, void* o2 = nullptr,
                   int d2 = 1) { tasks.push_back(ReduceTask{partial, R, N, seg, {o0, o1, o2}, {d0, d1, d2}}); };
  // LayerNorm backward: fp32 towers reduce at once, fp16 towers leave the partial matrix in the layer's slot
  auto ln_bwd = [&](const void* dyp, const void* xp, const float* gm, const float* mean, const float* rstd, const void* dres,
                    void* dxp, float* dgamma, float* dbeta, void* dx_colsum, int rows, long stride, float* slotp) -> int {
    if (f32)
      return hmmc_layernorm_bwd(dyp, xp, gm, mean, rstd, dres, dxp, dgamma, dbeta, dx_colsum, nullptr, rows, D, stride, dt, workspace,
                                gen, s);
    int rc = hmmc_layernorm_bwd_partial(dyp, xp, gm, mean, rstd, dres, dxp, slotp, dx_colsum != nullptr, nullptr, rows, D, stride, dt, s);
    if (rc == 0) defer(slotp, hmmc_layernorm_bwd_rows(rows), (dx_colsum ? 3 : 2) * D, D, dgamma, 1, dbeta, 1, dx_colsum, dt);
    return rc;
  };
  // folded layers: dx = du~ - mean(du~) - u mean(du~ o u) + dres; only the column sums of dx (a bias gradient) are left to reduce
  auto ln_bwd_folded = [&](const void* dut, const void* xp, const float* stat, const void* dres, void* dxp, void* dx_colsum, int rows,
                           long stride, float* slotp) -> int {
    int rc = hmmc_layernorm_bwd_fold(dut, xp, stat, dres, dxp, slotp, dx_colsum != nullptr, rows, D, stride, s);
    if (rc == 0 && dx_colsum) defer(slotp, hmmc_layernorm_bwd_fold_rows(rows), D, D, dx_colsum, dt);
    return rc;
  };
  hipStream_t sw = wgrad_stream ? wgrad_stream : s;
  const bool two = sw != s;
  WgradSync* const syncp = two ? wgrad_sync_for(s, sw) : nullptr;
  if (two && !syncp->ok) return HMMC_ERR_LAUNCH;
  bool done_set[4] = {false, false, false, false};
  bool gdone_set[2] = {false, false};
  // weight gradient k of a layer (0: c_proj <- g_in, 1: c_fc <- dh, 2: out_proj <- dx1, 3: in_proj <- dqkv)
  auto side_wgrad = [&](int k, const void* dyk, const void* xk, void* dW, int Np, int Kp, int rows = 0, int ldy = 0,
                        int ldx = 0) -> int {
    if (two) {
      if (hipEventRecord(syncp->ready, s) != hipSuccess || hipStreamWaitEvent(sw, syncp->ready, 0) != hipSuccess) return HMMC_ERR_LAUNCH;
    }
    int rc = wgrad(f32, dyk, xk, dW, rows ? rows : (int)T, Np, Kp, two ? wws : workspace, two ? wws_bytes : gen, sw, ldy, ldx);
    if (rc == 0 && two) {
      if (hipEventRecord(syncp->done[k], sw) != hipSuccess) return HMMC_ERR_LAUNCH;
      done_set[k] = true;
    }
    return rc;
  };
  // the chain on `s` is about to overwrite the operand weight gradient k read (scratch buffers are reused every layer)
  auto before_overwrite = [&](int k) -> int {
    if (two && done_set[k] && hipStreamWaitEvent(s, syncp->done[k], 0) != hipSuccess) return HMMC_ERR_LAUNCH;
    return 0;
  };
  const void* g_in = dy;
  const int ldl = L * D;                         // lead_only: row stride between the leading tokens of consecutive sequences
  if (lead_only) CK(hmmc_colsum(dy, grads[(size_t)(nlayers - 1) * 12 + 11], nseq, D, ldl, dt, dt, 0, workspace, gen, s));
  else CK(hmmc_colsum(dy, grads[(size_t)(nlayers - 1) * 12 + 11], (int)T, D, D, dt, dt, 0, workspace, gen, s));
  for (int i = nlayers - 1; i >= 0; --i) {
    const void* const* P = params + (size_t)i * 12;
    void* const* G = grads + (size_t)i * 12;
    Acts a;
    if (fb) {
      const size_t slab_f = hmmc_tower_act_bytes_fold(T, D, nseq, L, heads);
      const size_t off = i < fb->nfold ? (size_t)i * slab_f : (size_t)fb->nfold * slab_f + (size_t)(i - fb->nfold) * slab;
      a = carve((char*)acts + off, T, D, nseq, L, heads, es, f32, i < fb->nfold);
    } else {
      a = carve((char*)acts + (size_t)i * slab, T, D, nseq, L, heads, es, f32);
    }
    const void* xin = i == 0 ? x0 : (const void*)a.x;
    void* g_out = i == 0 ? dx : ping[i & 1];
    char* const sl = part_base + (size_t)i * slot.bytes;           // this layer's slot of partial matrices (fp16 tower)
    float* const p_ln2 = (float*)(sl + slot.ln2);
    float* const p_ln1 = (float*)(sl + slot.ln1);
    float* const p_fc = (float*)(sl + slot.fc);
    float* const p_attn = (float*)(sl + slot.attn);
    const size_t fc_bytes = slot.attn - slot.fc, attn_bytes = slot.bytes - slot.attn;
    // this layer's set of transient gradients; a set is written again two layers on, after the grouped launch that read it
    const int par = i & 1;
    void* const dh = dh_set[par];
    void* const dqkv = dqkv_set[par];
    void* const dx1 = dx1_set[par];
    const bool group_layer = grouped && !(lead_only && i + 1 == nlayers);
    if (two && gdone_set[par]) {
      if (hipStreamWaitEvent(s, syncp->gdone[par], 0) != hipSuccess) return HMMC_ERR_LAUNCH;
      gdone_set[par] = false;
    }
    if (fb && i < fb->nfold) {
      // ---- a layer hmmc_tower_fwd_fused ran folded (never a lead-only one: hmmc_tower_bwd_fold checks)
      const FoldLayer& f = fb->fl[i];
      const float* const stat1 = a.m1;            // row pairs of the layer's input x
      const float* const stat2 = a.m2;            // ... and of x1
      CK(before_overwrite(1));
      {
        // dh~ = rstd2_r x [(g_in W_proj) o QuickGELU'(h)], bias partials of the unscaled product
        const int rows = (int)hmmc_gemm_f16_colsum_rows((int)T, 4 * D, D);
        CK(hmmc_gemm_f16_fold(g_in, P[10], dh, (int)T, 4 * D, D, D, 4 * D, 4 * D, 0, nullptr, nullptr, nullptr, a.h,
                              EPI_MULAUX | EPI_COLSUM | EPI_ROWSCALE, stat2, nullptr, nullptr, p_fc, fc_bytes, s));
        defer(p_fc, rows, 4 * D, 4 * D, G[9], dt);
      }
      CK(hmmc_gemm_f16(dh, f.w2, dln, (int)T, D, 4 * D, 4 * D, D, D, 1, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, s));    // du~2
      CK(before_overwrite(2));
      CK(ln_bwd_folded(dln, a.x1, stat2, g_in, dx1, G[5], (int)T, D, p_ln2));                   // G[5]: out_proj bias = colsum(dx1)
      CK(dgrad(f32, dx1, P[4], dln, (int)T, D, D, nullptr, 0, s));                             // datt
      CK(before_overwrite(3));
      CK(hmmc_attention_f16_bwd_scaled(a.qkv, a.att, a.stat, dln, dqkv, p_attn, stat1, nseq, L, heads, causal, s));       // rstd1_r x dqkv
      defer(p_attn, nseq, 3 * D, 3 * D, G[3], dt);
      {
        const void* gy[4] = {g_in, dh, dx1, dqkv};
        const void* gx[4] = {a.g, a.x1, a.att, xin};
        void* gw[4] = {G[10], nullptr, G[4], nullptr};
        float* gw32[4] = {nullptr, f.s2, nullptr, f.s1};
        int Np[4], Kp[4];
        group_dims(D, Np, Kp);
        if (two && (hipEventRecord(syncp->ready, s) != hipSuccess || hipStreamWaitEvent(sw, syncp->ready, 0) != hipSuccess)) return HMMC_ERR_LAUNCH;
        CK(hmmc_gemm_f16_wgrad_group(gy, gx, gw, gw32, Np, Kp, 4, (int)T, two ? wws : workspace, two ? wws_bytes : gen, sw));
        if (two) {
          if (hipEventRecord(syncp->gdone[par], sw) != hipSuccess) return HMMC_ERR_LAUNCH;
          gdone_set[par] = true;
        }
      }
      CK(hmmc_gemm_f16(dqkv, f.w1, dln, (int)T, D, 3 * D, 3 * D, D, D, 1, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, s));   // du~1
      CK(before_overwrite(0));
      if (two && gdone_set[par ^ 1]) {
        if (hipStreamWaitEvent(s, syncp->gdone[par ^ 1], 0) != hipSuccess) return HMMC_ERR_LAUNCH;
        gdone_set[par ^ 1] = false;
      }
      CK(ln_bwd_folded(dln, xin, stat1, dx1, g_out, i > 0 ? grads[(size_t)(i - 1) * 12 + 11] : nullptr, (int)T, D, p_ln1));
      g_in = g_out;
      continue;
    }
    if (lead_only && i + 1 == nlayers) {
      // Last block, leading rows only (see hmmc_tower_fwd): g_in = dy and x1 / att are addressed at stride L*D, dh / dln / ln2 /
      // g / h are compact [nseq, .].  dx1 and the attention-output gradient are full [T, D] buffers that the attention and
      // ln_1 backward read whole: zero everywhere except the leading rows.
      if (hipMemsetAsync(dx1, 0, (size_t)T * D * es, s) != hipSuccess) return HMMC_ERR_LAUNCH;
      CK(side_wgrad(0, g_in, a.g, G[10], D, 4 * D, nseq, ldl, 0));
      const int rows = (int)hmmc_gemm_f16_colsum_rows(nseq, 4 * D, D);
      CK(hmmc_gemm_f16(g_in, P[10], dh, nseq, 4 * D, D, ldl, 4 * D, 4 * D, 1, 0, nullptr, nullptr, nullptr, a.h,
                       EPI_MULAUX | EPI_COLSUM, p_fc, fc_bytes, s));
      defer(p_fc, rows, 4 * D, 4 * D, G[9], dt);
      CK(side_wgrad(1, dh, a.ln2, G[8], 4 * D, D, nseq));
      CK(dgrad(f32, dh, P[8], dln, nseq, 4 * D, D, nullptr, 0, s));
      CK(ln_bwd(dln, a.x1, (const float*)P[6], a.m2, a.r2, g_in, dx1, (float*)G[6], (float*)G[7], G[5], nseq, ldl,
                p_ln2));                                                       // G[5]: out_proj bias = colsum(dx1 rows)
      CK(side_wgrad(2, dx1, a.att, G[4], D, D, nseq, ldl, ldl));
      if (lead_attention(true, f32, L)) {
        // The forward ran this block's attention for query 0 only (lead_inproj_attention).  datt exists on the class rows alone
        // (the rest of dln is not initialised and not read); the attention backward writes dK | dV of every token and dQ of the
        // class tokens; no gradient reaches the other tokens' Q, so the in_proj weight gradient is (dK | dV)^T ln1 over every
        // token plus dQ^T ln1 over the class rows, and the data gradient is (dK | dV) W_kv with the class rows recomputed over
        // the whole of K = 3D (one rounding, as in the all-token pass).
        CK(hmmc_gemm_f16(dx1, P[4], dln, nseq, D, D, ldl, D, ldl, 1, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, s));
        CK(before_overwrite(3));
        CK(hmmc_attention_f16_bwd_lead(a.qkv, a.att, a.stat, dln, dqkv, p_attn, nullptr, nseq, L, heads, causal, s));
        defer(p_attn, nseq, 3 * D, 3 * D, G[3], dt);
        half_t* const dqkv_h = (half_t*)dqkv;
        half_t* const gw = (half_t*)G[2];
        const half_t* const w_in = (const half_t*)P[2];
        CK(side_wgrad(3, dqkv_h + D, a.ln1, gw + (size_t)D * D, 2 * D, D, (int)T, 3 * D, 0));
        CK(side_wgrad(3, dqkv_h, a.ln1, gw, D, D, nseq, L * 3 * D, ldl));
        CK(hmmc_gemm_f16(dqkv_h + D, w_in + (size_t)D * D, dln, (int)T, D, 2 * D, 3 * D, D, D, 1, 0, nullptr, nullptr, nullptr, nullptr, 0,
                         nullptr, 0, s));
        CK(hmmc_gemm_f16(dqkv_h, w_in, dln, nseq, D, 3 * D, L * 3 * D, D, ldl, 1, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, s));
        CK(before_overwrite(0));
        if (two && gdone_set[par ^ 1]) {
          if (hipStreamWaitEvent(s, syncp->gdone[par ^ 1], 0) != hipSuccess) return HMMC_ERR_LAUNCH;
          gdone_set[par ^ 1] = false;
        }
        CK(ln_bwd(dln, xin, (const float*)P[0], a.m1, a.r1, dx1, g_out, (float*)G[0], (float*)G[1],
                  i > 0 ? grads[(size_t)(i - 1) * 12 + 11] : nullptr, (int)T, D, p_ln1));
        g_in = g_out;
        continue;
      }
      if (hipMemsetAsync(dln, 0, (size_t)T * D * es, s) != hipSuccess) return HMMC_ERR_LAUNCH;
      CK(hmmc_gemm_f16(dx1, P[4], dln, nseq, D, D, ldl, D, ldl, 1, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, s));
    } else {
    // MLP: x2 = x1 + c_proj(QuickGELU(c_fc(ln2)))
    if (!group_layer) CK(side_wgrad(0, g_in, a.g, G[10], D, 4 * D));
    CK(before_overwrite(1));
    if (f32) {
      CK(dgrad(f32, g_in, P[10], dh, (int)T, D, 4 * D, a.h, EPI_DGELU, s));
      CK(hmmc_colsum(dh, G[9], (int)T, 4 * D, 4 * D, dt, dt, 0, workspace, gen, s));
    } else {
      const int rows = (int)hmmc_gemm_f16_colsum_rows((int)T, 4 * D, D);
      CK(dgrad(f32, g_in, P[10], dh, (int)T, D, 4 * D, a.h, EPI_MULAUX, s, p_fc, fc_bytes));
      defer(p_fc, rows, 4 * D, 4 * D, G[9], dt);
    }
    if (!group_layer) CK(side_wgrad(1, dh, a.ln2, G[8], 4 * D, D));
    CK(dgrad(f32, dh, P[8], dln, (int)T, 4 * D, D, nullptr, 0, s));
    CK(before_overwrite(2));
    CK(ln_bwd(dln, a.x1, (const float*)P[6], a.m2, a.r2, g_in, dx1, (float*)G[6], (float*)G[7], G[5], (int)T, D,
              p_ln2));                                                         // G[5]: out_proj bias = colsum(dx1)
    // attention: x1 = x + out_proj(attn(in_proj(ln1)))
    if (!group_layer) CK(side_wgrad(2, dx1, a.att, G[4], D, D));
    CK(dgrad(f32, dx1, P[4], dln, (int)T, D, D, nullptr, 0, s));                 // datt (reuses dln)
    }
    CK(before_overwrite(3));
    if (f32) {
      CK(hmmc_temporal_attention_bwd((const float*)a.qkv, a.stat, (const float*)dln, (float*)dqkv, nseq, L, heads, s));
      CK(hmmc_colsum(dqkv, G[3], (int)T, 3 * D, 3 * D, dt, dt, 0, workspace, gen, s));
    } else {
      CK(hmmc_attention_f16_bwd(a.qkv, a.att, a.stat, dln, dqkv, p_attn, nseq, L, heads, causal, s));
      defer(p_attn, nseq, 3 * D, 3 * D, G[3], dt);
      (void)attn_bytes;
    }
    if (group_layer) {
      // all four weight gradients of the layer, now that the last operand (dqkv) exists
      const void* gy[4] = {g_in, dh, dx1, dqkv};
      const void* gx[4] = {a.g, a.ln2, a.att, a.ln1};
      void* gw[4] = {G[10], G[8], G[4], G[2]};
      int Np[4], Kp[4];
      group_dims(D, Np, Kp);
      if (two && (hipEventRecord(syncp->ready, s) != hipSuccess || hipStreamWaitEvent(sw, syncp->ready, 0) != hipSuccess)) return HMMC_ERR_LAUNCH;
      CK(hmmc_gemm_f16_wgrad_group(gy, gx, gw, nullptr, Np, Kp, 4, (int)T, two ? wws : workspace, two ? wws_bytes : gen, sw));
      if (two) {
        if (hipEventRecord(syncp->gdone[par], sw) != hipSuccess) return HMMC_ERR_LAUNCH;
        gdone_set[par] = true;
      }
    } else {
      CK(side_wgrad(3, dqkv, a.ln1, G[2], 3 * D, D));
    }
    CK(dgrad(f32, dqkv, P[2], dln, (int)T, 3 * D, D, nullptr, 0, s));
    // the c_proj bias gradient of the layer below is the column sum of the dx this call writes (into the buffer weight
    // gradient 0 of the layer above read as its g_in)
    CK(before_overwrite(0));
    if (two && gdone_set[par ^ 1]) {                // g_out is the g_in the grouped launch of the layer above reads
      if (hipStreamWaitEvent(s, syncp->gdone[par ^ 1], 0) != hipSuccess) return HMMC_ERR_LAUNCH;
      gdone_set[par ^ 1] = false;
    }
    CK(ln_bwd(dln, xin, (const float*)P[0], a.m1, a.r1, dx1, g_out, (float*)G[0], (float*)G[1],
              i > 0 ? grads[(size_t)(i - 1) * 12 + 11] : nullptr, (int)T, D, p_ln1));
    g_in = g_out;
  }
  if (!tasks.empty()) CK(hmmc_multi_colreduce(tasks.data(), (int)tasks.size(), s));
  if (two) {                                     // hand the weight gradients back in `s` order
    if (hipEventRecord(syncp->ready, sw) != hipSuccess || hipStreamWaitEvent(s, syncp->ready, 0) != hipSuccess) return HMMC_ERR_LAUNCH;
  }
  if (fb && fb->nfold > 0) {
    // the folded layers' in_proj / c_fc weight gradients and ln_1 / ln_2 gradients from the fp32 sums against the raw rows: they
    // need the bias gradients the reduce above just wrote and the sums the weight-gradient stream just handed back
    for (int l0 = 0; l0 < fb->nfold; l0 += 16) {
      const float* S[32]; const void* W[32]; const float* gm[32]; const float* bt[32]; const void* db[32]; void* dW[32]; float* dg[32];
      float* dbt[32]; float* vm[32]; int N[32];
      int n = 0;
      for (int i = l0; i < fb->nfold && i < l0 + 16; ++i) {
        const void* const* P = params + (size_t)i * 12;
        void* const* G = grads + (size_t)i * 12;
        const FoldLayer& f = fb->fl[i];
        S[n] = f.s1; W[n] = P[2]; gm[n] = (const float*)P[0]; bt[n] = (const float*)P[1]; db[n] = G[3]; dW[n] = G[2]; dg[n] = (float*)G[0];
        dbt[n] = (float*)G[1]; vm[n] = f.vm; N[n] = 3 * D; ++n;
        S[n] = f.s2; W[n] = P[8]; gm[n] = (const float*)P[6]; bt[n] = (const float*)P[7]; db[n] = G[9]; dW[n] = G[8]; dg[n] = (float*)G[6];
        dbt[n] = (float*)G[7]; vm[n] = f.vm + 3 * D + 2 * 8 * D; N[n] = 4 * D; ++n;
      }
      CK(hmmc_fold_grad_finish(S, W, gm, bt, db, dW, dg, dbt, vm, N, D, n, s));
    }
  }
  return HMMC_OK;
}
