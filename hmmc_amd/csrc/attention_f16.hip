// Fused multi-head self-attention for the CLIP towers (K4 of SURVEY.md section 2.3), forward and
// backward, fp16 in / fp32 softmax / fp16 out.  Replaces nn.MultiheadAttention's core at
// reference modules/module_clip.py:251 (ViT: no mask; text: causal mask of :441-447).
//
// Layout: qkv is the packed in-projection output [tokens, 3*D] (Q | K | V, head h = columns
// 64h..64h+63), tokens = sequences * L with a sequence's L tokens contiguous (batch-first; the
// reference's LND layout is only a permutation).  One wave owns one (sequence, head): the whole
// L x L problem (L <= 64: 50 ViT-B/32 tokens, <= 45 text tokens) lives in its registers and a
// private LDS slice, so there is no workgroup barrier anywhere.
//
// MFMA orientation ("key on the lane's row, query on the lane's column"): S^T = K Q^T is computed
// so that a lane holds, for ONE query column, 4 consecutive keys per 16x16 tile.  The softmax
// reduction over keys is then in-register plus two 16-lane shuffles, and the accumulator tile is
// directly the B operand of O^T = V^T P^T (k-order permuted identically on the V^T side, which is
// fetched with ds_read_b64_tr_b16 from the row-major V tile).  Outputs are written 8 bytes per lane.
#include "attn_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;


// Global -> registers in LANE ORDER: lane l takes row (l >> 3) + 8 i, bytes 16 (l & 7) .. +15 of the [L][64] operand, so
// every wave-instruction reads 8 full 128-byte rows with consecutive lanes on consecutive bytes.  (Loading the MFMA
// fragments directly - lane (c, g) on row c, 16 bytes at 64 ks + 16 g - puts 16 consecutive lanes on 16 different rows:
// the memory pipeline then handles 64 separate 16-byte pieces per instruction; scratch/ubench/burst_store.hip measures
// 3.4x between the two orders.)  Rows past L are clamped, never masked.
template <int KT>
__device__ __forceinline__ void load_rows(const half_t* src, long ld, int L, u4v (&raw)[2 * KT], int lane) {
#pragma unroll
  for (int i = 0; i < 2 * KT; ++i) {
    const int row = min((lane >> 3) + 8 * i, L - 1);
    raw[i] = *reinterpret_cast<const u4v*>(src + (long)row * ld + 8 * (lane & 7));
  }
}
// ... -> the wave's LDS tile (row-major, LDS_STRIDE) -> MFMA row fragments f[t][ks] = rows 16 t + (lane & 15), halves
// 32 ks + 8 (lane >> 4) .. +7.  The tile keeps the operand afterwards (the transposed reads use it).
template <int KT>
__device__ __forceinline__ void rows_to_frags(half_t* tile, const u4v (&raw)[2 * KT], h8 (&f)[KT][2], int lane) {
#pragma unroll
  for (int i = 0; i < 2 * KT; ++i)
    *reinterpret_cast<u4v*>(tile + ((lane >> 3) + 8 * i) * LDS_STRIDE + 8 * (lane & 7)) = raw[i];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      f[t][ks] = *reinterpret_cast<const h8*>(tile + (t * 16 + (lane & 15)) * LDS_STRIDE + ks * 32 + 8 * (lane >> 4));
}

// store the wave's [LP][64] operand (row fragments in registers) into its LDS tile, row-major
template <int KT>
__device__ __forceinline__ void frags_to_tile(half_t* tile, const h8 (&f)[KT][2], int lane) {
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      *reinterpret_cast<h8*>(tile + (t * 16 + (lane & 15)) * LDS_STRIDE + ks * 32 + 8 * (lane >> 4)) = f[t][ks];
}

// Row 0 of a [L][64] operand in the lane order of load_rows (lanes 0-7: its 128 bytes), every other row of the first 16 as zeros.
__device__ __forceinline__ void load_lead_row(const half_t* src, u4v (&raw)[2], int lane) {
  const u4v v = *reinterpret_cast<const u4v*>(src + 8 * (lane & 7));
  const u4v z = {0u, 0u, 0u, 0u};
  raw[0] = lane < 8 ? v : z;
  raw[1] = z;
}
// ... -> rows 0..15 of the wave's tile -> the row fragments of tile 0
__device__ __forceinline__ void lead_to_frags(half_t* tile, const u4v (&raw)[2], h8 (&f)[2], int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
    *reinterpret_cast<u4v*>(tile + ((lane >> 3) + 8 * i) * LDS_STRIDE + 8 * (lane & 7)) = raw[i];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) f[ks] = *reinterpret_cast<const h8*>(tile + (lane & 15) * LDS_STRIDE + ks * 32 + 8 * (lane >> 4));
}

// LEAD: only query 0 of every sequence is wanted (the last block of a tower whose caller reads the class token alone,
// hmmc_tower_fwd with lead_only): the SAME instruction sequence on query tile 0 with the tile's other fifteen rows as zeros - a
// query's column of every MFMA is independent of the other columns, so row 0 of the output and its log-sum-exp are bit-identical
// to the all-query kernel's - and only that row is stored.  The Q columns of the other tokens are never read.
#if defined(HMMC_SCRATCH) && defined(HMMC_ATTN_FWD_WPB)
constexpr int FWD_WPB = HMMC_ATTN_FWD_WPB;      // scratch experiments: waves per workgroup of the forward
#else
constexpr int FWD_WPB = 4;
#endif
template <int KT, bool LEAD = false>
__global__ __launch_bounds__(64 * FWD_WPB, 3) void attn_fwd_kernel(AttnArgs p) {
  constexpr int LP = 16 * KT;
  constexpr int QT = LEAD ? 1 : KT;              // query tiles computed
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long pair = (long)blockIdx.x * FWD_WPB + wid;
  if (pair >= (long)p.nseq * p.H) return;
  const int n = (int)(pair / p.H), h = (int)(pair % p.H);
  const int D = p.H * DH, L = p.L;
  const long ld = 3L * D;
  const half_t* q = p.qkv + (long)n * L * ld + h * DH;
  const half_t* k = q + D;
  const half_t* v = q + 2 * D;
  half_t* vtile = reinterpret_cast<half_t*>(smem) + wid * (LP * LDS_STRIDE + 16 * LDS_STRIDE);
  half_t* scr = vtile + LP * LDS_STRIDE;
  const int g = lane >> 4, c = lane & 15;

  // every operand is requested before the first MFMA (rows past L clamped; masked below where it matters); V goes last
  // through the tile, which then holds it for the transposed reads
  h8 kf[KT][2], qf[QT][2], vf[KT][2];
  {
    u4v rk[2 * KT], rq[2 * QT], rv[2 * KT];
    load_rows<KT>(k, ld, L, rk, lane);
    if constexpr (LEAD) load_lead_row(q, rq, lane); else load_rows<KT>(q, ld, L, rq, lane);
    load_rows<KT>(v, ld, L, rv, lane);
    rows_to_frags<KT>(vtile, rk, kf, lane);
    if constexpr (LEAD) lead_to_frags(vtile, rq, qf[0], lane); else rows_to_frags<KT>(vtile, rq, qf, lane);
    rows_to_frags<KT>(vtile, rv, vf, lane);
  }
  h8 vT[4][KT / 2];                              // V^T fragments, k-order permuted like the P^T accumulators
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int ks = 0; ks < KT / 2; ++ks) vT[dt][ks] = tr_frag(vtile, ks * 32, ks * 32 + 16, dt * 16, lane);

  half_t* o = p.out + (long)n * L * D + h * DH;
  const int LQ = LEAD ? 1 : L;                   // queries whose results are stored
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qi = qt * 16 + c;
    // S^T[key][q] = sum_d K[key][d] Q[q][d]; lane holds keys kt*16 + 4g + r of query column qi
    f4 s[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      f4 z = {0.f, 0.f, 0.f, 0.f};
      s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][0], qf[qt][0], z, 0, 0, 0);
      s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][1], qf[qt][1], s[kt], 0, 0, 0);
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        float val = s[kt][r] * 0.125f;
        if ((kt + 1) * 16 > L || p.causal) val = (key < L && (!p.causal || key <= qi || qi >= L)) ? val : -INFINITY;
        s[kt][r] = val;
        m = fmaxf(m, val);
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = __expf(s[kt][r] - m);
        s[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (p.lse && g == 0 && qi < LQ) p.lse[((long)n * p.H + h) * L + qi] = m + __logf(sum);
    h4 pt[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pt[kt][r] = (half_t)(s[kt][r] * inv);
    // O^T[d][q] = sum_key V[key][d] P[q][key]; k-step ks covers key tiles 2ks, 2ks+1 in permuted order
    f4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      acc[dt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KT / 2; ++ks)
        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vT[dt][ks], cat4(pt[2 * ks], pt[2 * ks + 1]), acc[dt], 0, 0, 0);
    }
    store_rows(o, D, acc, qt * 16, LQ, scr, lane);
  }
}

// Backward.  One wave per (sequence, head), no workgroup barrier.
//  * Every global operand (Q, K, V, dO row fragments: 8 x KT loads of 1 KiB per wave) is requested before the first
//    MFMA, so a wave has its whole input in flight at once and the two waves of a SIMD cover each other's latency.
//    Rows past L are clamped, never masked: their probabilities are forced to zero instead (lse = +inf for queries,
//    an explicit key mask in the last key tile), so whatever they hold cannot reach an output.
//  * softmax backward exactly as autograd writes it: dS = P o (dP - rowsum(P o dP)); the attention output O is not read.
//  * S and dP are formed in BOTH orientations from the same register fragments (MFMA operands swapped): keys on the
//    lane's rows for dQ (the accumulator tile is directly the B operand of dQ^T = K^T dS^T), queries on the lane's rows
//    for dV^T = dO^T P and dK^T = Q^T dS.  That costs 2 x the (cheap) QK^T / dO V^T MFMAs and exponentials and
//    removes every LDS transpose of P / dS.
//  * One 9 KiB LDS tile per wave, refilled from registers (K, then dO, then Q), serves the three transposed
//    operands (ds_read_b64_tr_b16); per-query lse / delta are redistributed through 512 B of scratch.
//  * Outputs leave as 16 B per lane: v_permlane16_swap pairs the d-tiles (2q, 2q+1) so a lane owns 8 consecutive d.
//  * LEAD: the gradient of the output is non-zero for query 0 of every sequence only (hmmc_tower_bwd with lead_only: dout is read
//    at that row alone, the rest of the buffer may hold anything).  Q and dO enter as tile 0 with fifteen zero rows, every other
//    query has lse = +inf (probability exactly 0, like the queries past L), phase 1 runs for query tile 0 and phase 2 over the
//    first pair of query tiles; dK and dV are written for every token, dQ for query 0 only - the Q columns of the other rows
//    of dqkv are NOT written (their gradient is exactly zero; the tower's data and weight gradients do not read them).
// Workgroups of TWO waves (round 5; four until then): a workgroup's LDS and wave slots come free when its last wave ends, so
// smaller groups refill a CU's eight slots sooner (scratch/attn_abl.sh, 3 072 x 50 x 12: 430 us with 4 waves per workgroup, 415 with
// 2, 412 with 1, 468 with 8).  The same ablations say where the time goes: 232 us with the arithmetic removed (loads, fragment
// shuffles and stores only), -82 / -100 us with the dV / dK pass of phase 2 removed - load time and compute time ADD.  A
// persistent, software-pipelined form (next pair's operands by LDS-DMA under the current pair's phases; scratch/attn_bwd_pipe.patch,
// bit-identical on its first run) needs 40 KiB of LDS per wave = one wave per SIMD, and one wave's dependent MFMA / exp / LDS
// chain takes 16.5 us per pair alone (594 us); two waves per SIMD sharing the pipes are what hides it here (23 us per pair and wave).
constexpr int BWD_WPB = 2;
template <int KT, bool LEAD = false>
__global__ __launch_bounds__(64 * BWD_WPB, 2) void attn_bwd_kernel(AttnArgs p) {
  constexpr int LP = 16 * KT;
  constexpr int QT = LEAD ? 1 : KT;              // query tiles with a non-zero output gradient
  constexpr int WAVE_LDS = LP * LDS_STRIDE * 2 + 16 * LDS_STRIDE * 2 + 2 * LP * 4;     // tile + store scratch + lse[LP] + delta[LP]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long pair = (long)blockIdx.x * BWD_WPB + wid;
  if (pair >= (long)p.nseq * p.H) return;
  const int n = (int)(pair / p.H), h = (int)(pair % p.H);
  const int D = p.H * DH, L = p.L;
  const long ld = 3L * D;
  const half_t* q = p.qkv + (long)n * L * ld + h * DH;
  const half_t* k = q + D;
  const half_t* v = q + 2 * D;
  const half_t* dO = p.dout + (long)n * L * D + h * DH;
  half_t* dq = p.dqkv + (long)n * L * ld + h * DH;
  half_t* dk = dq + D;
  half_t* dv = dq + 2 * D;
  half_t* xt = reinterpret_cast<half_t*>(smem + wid * WAVE_LDS);
  half_t* scr = xt + LP * LDS_STRIDE;
  float* lse_s = reinterpret_cast<float*>(smem + wid * WAVE_LDS + LP * LDS_STRIDE * 2 + 16 * LDS_STRIDE * 2);
  float* del_s = lse_s + LP;
  const int g = lane >> 4, c = lane & 15;
  const float* lse_g = p.lse + ((long)n * p.H + h) * L;

  h8 qf[KT][2], kf[KT][2], vf[KT][2], df[KT][2];
  if constexpr (LEAD) {
    u4v rq[2], rk[2 * KT], rv[2 * KT], rd[2];
    load_lead_row(q, rq, lane);
    load_rows<KT>(v, ld, L, rv, lane);
    load_lead_row(dO, rd, lane);
    load_rows<KT>(k, ld, L, rk, lane);
    const h8 zf = {};
#pragma unroll
    for (int t = 1; t < KT; ++t) { qf[t][0] = zf; qf[t][1] = zf; df[t][0] = zf; df[t][1] = zf; }
    lead_to_frags(xt, rq, qf[0], lane);
    rows_to_frags<KT>(xt, rv, vf, lane);
    lead_to_frags(xt, rd, df[0], lane);
    rows_to_frags<KT>(xt, rk, kf, lane);       // K last: phase 1 reads K^T from the tile
  } else {
    u4v rq[2 * KT], rk[2 * KT], rv[2 * KT], rd[2 * KT];
    load_rows<KT>(q, ld, L, rq, lane);
    load_rows<KT>(v, ld, L, rv, lane);
    load_rows<KT>(dO, D, L, rd, lane);
    load_rows<KT>(k, ld, L, rk, lane);
    rows_to_frags<KT>(xt, rq, qf, lane);
    rows_to_frags<KT>(xt, rv, vf, lane);
    rows_to_frags<KT>(xt, rd, df, lane);
    rows_to_frags<KT>(xt, rk, kf, lane);       // K last: phase 1 reads K^T from the tile
  }
  const int LQ = LEAD ? 1 : L;                   // queries with a gradient
  float lse_c[KT];                               // lse of query qt*16 + c; +inf past LQ: its probabilities vanish
#pragma unroll
  for (int t = 0; t < KT; ++t) lse_c[t] = (t * 16 + c < LQ) ? lse_g[t * 16 + c] : INFINITY;
  if (g == 0) {                                  // exp(x - lse) = exp2(x log2(e) - lse log2(e)): one fma + v_exp_f32 per probability
#pragma unroll
    for (int t = 0; t < KT; ++t) lse_s[t * 16 + c] = LOG2E * lse_c[t];
  }

  float* dbias = p.dbias ? p.dbias + (long)n * 3 * D + h * DH : nullptr;   // + type * D
  float rs[KT];                                  // row factor of token t*16 + c (1 without p.rowstat)
#pragma unroll
  for (int t = 0; t < KT; ++t) rs[t] = p.rowstat ? p.rowstat[2 * ((long)n * L + min(t * 16 + c, L - 1))] : 1.0f;
  f4 csum[4];
  // ---- phase 1: keys on the lane's rows -> delta and dQ (the tile holds K)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) csum[dt] = f4{0.f, 0.f, 0.f, 0.f};
  if constexpr (LEAD) {                          // phase 2 reads delta of the first PAIR of query tiles
    if (g == 0) del_s[16 + c] = 0.f;
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qi = qt * 16 + c;
    f4 s[KT], dp[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      f4 z = {0.f, 0.f, 0.f, 0.f};
      s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][0], qf[qt][0], z, 0, 0, 0);
      s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][1], qf[qt][1], s[kt], 0, 0, 0);
      dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[kt][0], df[qt][0], z, 0, 0, 0);
      dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[kt][1], df[qt][1], dp[kt], 0, 0, 0);
    }
    float dl = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], 0.125f * LOG2E, -LOG2E * lse_c[qt]));
        if ((kt + 1) * 16 > L || p.causal) pv = (key < L && (!p.causal || key <= qi)) ? pv : 0.f;
        s[kt][r] = pv;
        dl += pv * dp[kt][r];
      }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    if (g == 0) del_s[qi] = dl;
    h4 ds16[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) ds16[kt][r] = (half_t)(s[kt][r] * (dp[kt][r] - dl) * 0.125f);
    f4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      acc[dt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KT / 2; ++ks)          // K^T fragments re-read from the tile: keeps 32 VGPRs free
        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag(xt, ks * 32, ks * 32 + 16, dt * 16, lane),
                                                         cat4(ds16[2 * ks], ds16[2 * ks + 1]), acc[dt], 0, 0, 0);
    }
    store_rows(dq, ld, acc, qt * 16, LQ, scr, lane, rs[qt]);
    if (dbias) add_rounded(csum, acc);           // rows past LQ are exact zeros (their dS is)
  }
  if (dbias) store_colsum(dbias, csum, lane);

  // ---- phase 2: queries on the lane's rows.  dV^T[d][key] = sum_q dO[q][d] P[q][key], then dK^T[d][key] = sum_q Q[q][d]
  // dS[q][key].  One key tile at a time and the probabilities formed again in each pass (32 MFMAs and 64 exponentials
  // more): P or dS of ONE key tile (8 VGPRs) is all that lives beside the four operand fragments.  Holding P and dS of all
  // 16 tile pairs (64 VGPRs) as the B operands of both products took the kernel to 286 VGPRs: 30 spilled to scratch
  // memory and reloaded behind full vmcnt waits in the middle of the MFMA stream.
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#if defined(HMMC_SCRATCH) && defined(HMMC_ATTN_ABL)
    if (pass == HMMC_ATTN_ABL - 1) continue;     // timing experiments only (wrong results): 1 skips the dV pass, 2 the dK pass
#endif
    if (pass == 0) frags_to_tile<KT>(xt, df, lane); else frags_to_tile<KT>(xt, qf, lane);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) csum[dt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int key = kt * 16 + c;
      constexpr int QP = LEAD ? 2 : KT;          // query tiles that enter the products (a k-step is a pair of them)
      h4 b16[QP];                                // P (pass 0) or dS (pass 1) of key tile kt, [qt]
#pragma unroll
      for (int qt = 0; qt < QP; ++qt) {
        f4 z = {0.f, 0.f, 0.f, 0.f};
        f4 sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[qt][0], kf[kt][0], z, 0, 0, 0);
        sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[qt][1], kf[kt][1], sc, 0, 0, 0);
        f4 dp = z;
        if (pass == 1) {
          dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(df[qt][0], vf[kt][0], z, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(df[qt][1], vf[kt][1], dp, 0, 0, 0);
        }
        const f4 lr = *reinterpret_cast<const f4*>(lse_s + qt * 16 + 4 * g);
        const f4 dr = *reinterpret_cast<const f4*>(del_s + qt * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = qt * 16 + 4 * g + r;
          float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], 0.125f * LOG2E, -lr[r]));      // lse_s holds lse * log2(e)
          if ((kt + 1) * 16 > L || p.causal) pv = (key < L && (!p.causal || key <= qi)) ? pv : 0.f;
          b16[qt][r] = pass == 0 ? (half_t)pv : (half_t)(pv * (dp[r] - dr[r]) * 0.125f);
        }
      }
      f4 acc[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        acc[dt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < QP / 2; ++ks)        // dO^T / Q^T fragments re-read from the tile
          acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag(xt, ks * 32, ks * 32 + 16, dt * 16, lane),
                                                           cat4(b16[2 * ks], b16[2 * ks + 1]), acc[dt], 0, 0, 0);
      }
      store_rows(pass == 0 ? dv : dk, ld, acc, kt * 16, L, scr, lane, rs[kt]);
      if (dbias) add_rounded(csum, acc);         // keys past L are exact zeros (P and dS are)
      __builtin_amdgcn_sched_barrier(0);         // one key tile at a time: the scheduler would otherwise hoist every tile's MFMAs
    }
    if (dbias) store_colsum(dbias + (pass == 0 ? 2 : 1) * D, csum, lane);
  }
}

}  // namespace

int hmmc_attention_long_fwd(const AttnArgs& p, hipStream_t stream, bool lead = false);
int hmmc_attention_long_bwd(const AttnArgs& p, hipStream_t stream, bool lead = false);

extern "C" int hmmc_attention_f16_fwd(const void* qkv, void* out, float* lse, int nseq, int L, int H, int causal,
                                      hipStream_t stream) {
  if (!qkv || !out || !lse || nseq <= 0 || L <= 0 || H <= 0) return HMMC_ERR_ARG;
  if (L > 256) return HMMC_ERR_UNSUPPORTED;
  if (L > 64) {
    AttnArgs pl{};
    pl.qkv = (const half_t*)qkv; pl.out = (half_t*)out; pl.lse = lse; pl.nseq = nseq; pl.L = L; pl.H = H; pl.causal = causal;
    return hmmc_attention_long_fwd(pl, stream);
  }
  AttnArgs p{};
  p.qkv = (const half_t*)qkv; p.out = (half_t*)out; p.lse = lse; p.nseq = nseq; p.L = L; p.H = H; p.causal = causal;
  long pairs = (long)nseq * H;
  dim3 grid((unsigned)((pairs + FWD_WPB - 1) / FWD_WPB)), block(64 * FWD_WPB);
  if (L <= 32) hipLaunchKernelGGL(attn_fwd_kernel<2>, grid, block, FWD_WPB * (32 + 16) * LDS_STRIDE * 2, stream, p);
  else hipLaunchKernelGGL(attn_fwd_kernel<4>, grid, block, FWD_WPB * (64 + 16) * LDS_STRIDE * 2, stream, p);
  return hmmc_launch_status();
}

static int attention_bwd(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv, float* dbias_partial,
                         const float* rowstat, int nseq, int L, int H, int causal, hipStream_t stream);

extern "C" int hmmc_attention_f16_bwd(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv,
                                      float* dbias_partial, int nseq, int L, int H, int causal, hipStream_t stream) {
  return attention_bwd(qkv, out, lse, dout, dqkv, dbias_partial, nullptr, nseq, L, H, causal, stream);
}

// the same with every row of dqkv multiplied by rowstat[token][0] on its way out: the
// gradient a folded ln_1 -> in_proj consumes (ln_fold.hip); dbias_partial stays the column sums of the unscaled gradient
extern "C" int hmmc_attention_f16_bwd_scaled(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv,
                                             float* dbias_partial, const float* rowstat, int nseq, int L, int H, int causal,
                                             hipStream_t stream) {
  if (!rowstat) return HMMC_ERR_ARG;
  return attention_bwd(qkv, out, lse, dout, dqkv, dbias_partial, rowstat, nseq, L, H, causal, stream);
}

static int attention_bwd(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv, float* dbias_partial,
                         const float* rowstat, int nseq, int L, int H, int causal, hipStream_t stream) {
  if (!qkv || !out || !lse || !dout || !dqkv || nseq <= 0 || L <= 0 || H <= 0) return HMMC_ERR_ARG;
  if (L > 256) return HMMC_ERR_UNSUPPORTED;
  if (L > 64) {
    AttnArgs pl{};
    pl.qkv = (const half_t*)qkv; pl.out = (half_t*)out; pl.lse = (float*)lse; pl.dout = (const half_t*)dout;
    pl.dqkv = (half_t*)dqkv; pl.dbias = dbias_partial; pl.rowstat = rowstat; pl.nseq = nseq; pl.L = L; pl.H = H; pl.causal = causal;
    return hmmc_attention_long_bwd(pl, stream);
  }
  AttnArgs p{};
  p.qkv = (const half_t*)qkv; p.out = (half_t*)out; p.lse = (float*)lse; p.dout = (const half_t*)dout;
  p.dqkv = (half_t*)dqkv; p.dbias = dbias_partial; p.rowstat = rowstat; p.nseq = nseq; p.L = L; p.H = H; p.causal = causal;
  long pairs = (long)nseq * H;
  dim3 grid((unsigned)((pairs + BWD_WPB - 1) / BWD_WPB)), block(64 * BWD_WPB);
  if (L <= 32) hipLaunchKernelGGL(attn_bwd_kernel<2>, grid, block, BWD_WPB * ((32 + 16) * LDS_STRIDE * 2 + 2 * 32 * 4), stream, p);
  else hipLaunchKernelGGL(attn_bwd_kernel<4>, grid, block, BWD_WPB * ((64 + 16) * LDS_STRIDE * 2 + 2 * 64 * 4), stream, p);
  return hmmc_launch_status();
}

// ---- query 0 only (the last block of a tower read at its class token: hmmc_tower_fwd / hmmc_tower_bwd with lead_only) --------
// out row n*L and lse[(n, h, 0)] of every (sequence, head) - bit-identical to hmmc_attention_f16_fwd's - from the K and V columns of
// all L tokens and the Q columns of token 0 alone; nothing else of `out` or `lse` is written, no other Q is read.  L <= 64.
extern "C" int hmmc_attention_f16_fwd_lead(const void* qkv, void* out, float* lse, int nseq, int L, int H, int causal,
                                           hipStream_t stream) {
  if (!qkv || !out || !lse || nseq <= 0 || L <= 0 || H <= 0) return HMMC_ERR_ARG;
  if (L > 256) return HMMC_ERR_UNSUPPORTED;
  AttnArgs p{};
  p.qkv = (const half_t*)qkv; p.out = (half_t*)out; p.lse = lse; p.nseq = nseq; p.L = L; p.H = H; p.causal = causal;
  if (L > 64) return hmmc_attention_long_fwd(p, stream, true);
  long pairs = (long)nseq * H;
  dim3 grid((unsigned)((pairs + FWD_WPB - 1) / FWD_WPB)), block(64 * FWD_WPB);
  if (L <= 32) hipLaunchKernelGGL((attn_fwd_kernel<2, true>), grid, block, FWD_WPB * (32 + 16) * LDS_STRIDE * 2, stream, p);
  else hipLaunchKernelGGL((attn_fwd_kernel<4, true>), grid, block, FWD_WPB * (64 + 16) * LDS_STRIDE * 2, stream, p);
  return hmmc_launch_status();
}

// The backward of that: dout (and, for sequences above 64 tokens, out: row n*L of the forward's result) is read at row n*L of
// every sequence only; dqkv receives dK and dV of every token and dQ of token 0
// (the Q columns of the other rows are left untouched: their gradient is exactly zero); dbias_partial as hmmc_attention_f16_bwd;
// rowstat (optional) as hmmc_attention_f16_bwd_scaled.
extern "C" int hmmc_attention_f16_bwd_lead(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv,
                                           float* dbias_partial, const float* rowstat, int nseq, int L, int H, int causal,
                                           hipStream_t stream) {
  if (!qkv || !lse || !dout || !dqkv || nseq <= 0 || L <= 0 || H <= 0) return HMMC_ERR_ARG;
  if (L > 256 || (L > 64 && !out)) return HMMC_ERR_UNSUPPORTED;         // the long-sequence kernel forms delta from O (row 0 here)
  AttnArgs p{};
  p.qkv = (const half_t*)qkv; p.out = (half_t*)out; p.lse = (float*)lse; p.dout = (const half_t*)dout;
  p.dqkv = (half_t*)dqkv; p.dbias = dbias_partial; p.rowstat = rowstat; p.nseq = nseq; p.L = L; p.H = H; p.causal = causal;
  if (L > 64) return hmmc_attention_long_bwd(p, stream, true);
  long pairs = (long)nseq * H;
  dim3 grid((unsigned)((pairs + BWD_WPB - 1) / BWD_WPB)), block(64 * BWD_WPB);
  if (L <= 32) hipLaunchKernelGGL((attn_bwd_kernel<2, true>), grid, block, BWD_WPB * ((32 + 16) * LDS_STRIDE * 2 + 2 * 32 * 4), stream, p);
  else hipLaunchKernelGGL((attn_bwd_kernel<4, true>), grid, block, BWD_WPB * ((64 + 16) * LDS_STRIDE * 2 + 2 * 64 * 4), stream, p);
  return hmmc_launch_status();
}
