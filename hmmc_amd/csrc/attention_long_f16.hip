// Fused attention for sequences of 65..256 tokens (ViT-B/16: 197 tokens per frame; CLIP text up to 77): the same
// "key on the row, query on the column" MFMA orientation and register-fragment layout as attention_f16.hip, in blocks of
// 64 x 64 with an online softmax over key blocks (forward) and block-wise recomputation from the saved log-sum-exp
// (backward).  One wave owns one (sequence, head, 64-row block[, role]); all LDS is wave-private, no workgroup barriers.
// Reference: nn.MultiheadAttention core at modules/module_clip.py:251 with 197 x 197 heads.
#include "attn_common.h"

namespace {

// ---- register-fragment helpers (the layout of attention_f16.hip: operand rows straight from global memory into MFMA
// fragments, one 9 KiB LDS tile per wave for the transposed operand) -----------------------------------------------------
__device__ __forceinline__ h8 gfrag_clamped(const half_t* src, int row0, int ks, int L, long ld, int lane) {
  int row = min(row0 + (lane & 15), L - 1);
  return *reinterpret_cast<const h8*>(src + (long)row * ld + ks * 32 + 8 * (lane >> 4));
}
__device__ __forceinline__ void frags_to_tile(half_t* tile, const h8 (&f)[4][2], int lane) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      *reinterpret_cast<h8*>(tile + (t * 16 + (lane & 15)) * LDS_STRIDE + ks * 32 + 8 * (lane >> 4)) = f[t][ks];
}
// X^T accumulators of one 16-row tile (4 d-tiles) -> global row `row`, 16 B per lane (v_permlane16_swap pairs the d-tiles)
__device__ __forceinline__ void store_row16(half_t* dst, long ld, const f4 (&acc)[4], int row, bool ok, int lane) {
  const int g = lane >> 4;
  unsigned d[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    h4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (half_t)acc[dt][r];
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    u2v u = __builtin_bit_cast(u2v, v);
    d[dt][0] = u[0]; d[dt][1] = u[1];
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      auto r = __builtin_amdgcn_permlane16_swap(d[2 * q][e], d[2 * q + 1][e], false, false);
      d[2 * q][e] = r[0]; d[2 * q + 1][e] = r[1];
    }
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    u4v o = {d[2 * q][0], d[2 * q][1], d[2 * q + 1][0], d[2 * q + 1][1]};
    if (ok) *reinterpret_cast<u4v*>(dst + (long)row * ld + 32 * q + 16 * (g & 1) + 8 * (g >> 1)) = o;
  }
}

// Forward: one WORKGROUP per (sequence, head).  Its four waves first bring the head's K and V (all L <= 256 keys: 2 x 36 KiB
// at most) into LDS with full-row, lane-ordered loads, once; then each wave takes 16-query tiles (tile w, w + 4, ...).  A
// whole row of scores - KTL key tiles x 4 values per lane - fits the registers, so the softmax is the one-block kernel's
// (attention_f16.hip: in-register maximum and sum, two 16-lane shuffles, no online rescaling), K fragments come from LDS with
// ds_read_b128 and V^T fragments with ds_read_b64_tr_b16.  Against the previous wave-private 64 x 64 blocks (each wave
// fetching every K / V block itself, in 16-byte pieces, with nothing in flight while it computed) this reads K and V once
// per head instead of four times and has no global load between the first MFMA and the last.
constexpr float LOG2E = 1.4426950408889634f;

template <int KTL, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_long_fwd_kernel(AttnArgs p) {
  constexpr int ROWS = 16 * KTL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int D = p.H * DH, L = p.L;
  const long ld = 3L * D;
  const half_t* q = p.qkv + (long)n * L * ld + h * DH;
  const half_t* k = q + D;
  const half_t* v = q + 2 * D;
  half_t* ktile = reinterpret_cast<half_t*>(smem);
  half_t* vtile = ktile + ROWS * LDS_STRIDE;
  half_t* scr = vtile + ROWS * LDS_STRIDE + wid * (16 * LDS_STRIDE);
  const int g = lane >> 4, c = lane & 15;
  {                                              // K, V -> LDS: thread -> row (tid >> 3) + 32 i, bytes 16 (tid & 7) .. +15; rows past L zero
    u4v rk[ROWS / 32], rv[ROWS / 32];
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const long off = (long)min(row, L - 1) * ld + 8 * (tid & 7);
      rk[i] = *reinterpret_cast<const u4v*>(k + off);
      rv[i] = *reinterpret_cast<const u4v*>(v + off);
    }
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const u4v z = {0u, 0u, 0u, 0u};
      *reinterpret_cast<u4v*>(ktile + row * LDS_STRIDE + 8 * (tid & 7)) = row < L ? rk[i] : z;
      *reinterpret_cast<u4v*>(vtile + row * LDS_STRIDE + 8 * (tid & 7)) = row < L ? rv[i] : z;
    }
  }
  __syncthreads();
  half_t* o = p.out + (long)n * L * D + h * DH;
  const int nqt = (L + 15) / 16;
  // a tile's 16 query rows: lane-ordered loads (requested one tile ahead) -> scratch -> row fragments
  auto load_q = [&](int qt, u4v (&raw)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      raw[t] = *reinterpret_cast<const u4v*>(q + (long)min(qt * 16 + (lane >> 3) + 8 * t, L - 1) * ld + 8 * (lane & 7));
  };
  u4v qraw[2], qnext[2];
  load_q(min(wid, nqt - 1), qraw);
  for (int qt = wid; qt < nqt; qt += 4) {
    const int q0 = qt * 16, qi = q0 + c;
    load_q(min(qt + 4, nqt - 1), qnext);
#pragma unroll
    for (int t = 0; t < 2; ++t)
      *reinterpret_cast<u4v*>(scr + ((lane >> 3) + 8 * t) * LDS_STRIDE + 8 * (lane & 7)) = qraw[t];
    h8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const h8*>(scr + c * LDS_STRIDE + ks * 32 + 8 * g);
    // S^T[key][q]: lane holds keys kt*16 + 4g + r of query column qi, for every key tile
    f4 s[KTL];
#pragma unroll
    for (int half = 0; half < 2; ++half) {         // the K fragments of half the key tiles are requested together, then multiplied
      h8 kf[KTL / 2][2];
#pragma unroll
      for (int i = 0; i < KTL / 2; ++i) {
        const half_t* kr = ktile + ((half * (KTL / 2) + i) * 16 + c) * LDS_STRIDE + 8 * g;
        kf[i][0] = *reinterpret_cast<const h8*>(kr);
        kf[i][1] = *reinterpret_cast<const h8*>(kr + 32);
      }
#pragma unroll
      for (int i = 0; i < KTL / 2; ++i) {
        f4 z = {0.f, 0.f, 0.f, 0.f};
        z = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i][0], qf[0], z, 0, 0, 0);
        s[half * (KTL / 2) + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i][1], qf[1], z, 0, 0, 0);
      }
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KTL; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        float val = s[kt][r] * (0.125f * LOG2E);                  // base-2 exponent: exp(x) = exp2(x log2 e)
        // keys past L can only sit in the last two tiles (KTL = key tiles rounded up to an even count): no test elsewhere
        if (CAUSAL) val = (key < L && (key <= qi || qi >= L)) ? val : -INFINITY;
        else if (kt >= KTL - 2) val = key < L ? val : -INFINITY;
        s[kt][r] = val;
        m = fmaxf(m, val);
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KTL; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[kt][r] - m);
        s[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (g == 0 && qi < L) p.lse[((long)n * p.H + h) * L + qi] = (m + __log2f(sum)) * (1.0f / LOG2E);
    h4 pt[KTL];
#pragma unroll
    for (int kt = 0; kt < KTL; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pt[kt][r] = (half_t)(s[kt][r] * inv);
    // O^T[d][q] = sum_key V[key][d] P[q][key]; k-step ks covers key tiles 2ks, 2ks+1 in permuted order
    f4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      acc[dt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KTL / 2; ++ks)
        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag(vtile, ks * 32, ks * 32 + 16, dt * 16, lane),
                                                         cat4(pt[2 * ks], pt[2 * ks + 1]), acc[dt], 0, 0, 0);
    }
    store_rows(o, D, acc, q0, L, scr, lane);
    qraw[0] = qnext[0]; qraw[1] = qnext[1];
  }
}

// Backward: one wave per (sequence, head, 64-row block, role).  role 0: dQ of a query block (loops over key blocks);
// role 1: dV of a key block, role 2: dK of a key block (loop over query blocks).  Every role recomputes the probabilities it
// needs from the saved log-sum-exp (8 MFMA units per block pair instead of the minimal 5) and in exchange holds only its
// own 64 accumulator registers, keeps all operand rows as register fragments and needs one 9 KiB LDS tile for the
// transposed operand - the layout of attention_f16.hip's one-block kernel - so 8 waves per CU are resident where the
// previous one-wave-per-head kernel had 3.  delta[q] = <dO[q], O[q]> is recomputed per query block from the fragments.
// Rows past L are clamped, never masked: their probabilities are forced to zero (lse = +inf for queries, key mask).
__device__ __forceinline__ void delta_of_block(const h8 (&df)[4][2], const half_t* o, int q0, int L, int D, int lane,
                                               float (&delta_c)[4]) {
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const h8 o0 = gfrag_clamped(o, q0 + qt * 16, 0, L, D, lane), o1 = gfrag_clamped(o, q0 + qt * 16, 1, L, D, lane);
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) d += (float)df[qt][0][j] * (float)o0[j] + (float)df[qt][1][j] * (float)o1[j];
    d += __shfl_xor(d, 16, 64);
    d += __shfl_xor(d, 32, 64);
    delta_c[qt] = d;                              // query q0 + qt*16 + (lane & 15)
  }
}

template <int ROLE>
__global__ __launch_bounds__(256, 2) void attn_long_bwd_kernel(AttnArgs p) {
  constexpr int WAVE_LDS = 64 * LDS_STRIDE * 2 + 2 * 64 * 4;      // tile + lse[64] + delta[64]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nb = (p.L + 63) / 64;
  const long idx = (long)blockIdx.x * 4 + wid;
  if (idx >= (long)p.nseq * p.H * nb) return;
  constexpr int role = ROLE;
  const int blk = (int)(idx % nb);
  const long pair = idx / nb;
  const int n = (int)(pair / p.H), h = (int)(pair % p.H);
  const int D = p.H * DH, L = p.L;
  const long ld = 3L * D;
  const half_t* q = p.qkv + (long)n * L * ld + h * DH;
  const half_t* k = q + D;
  const half_t* v = q + 2 * D;
  const half_t* o = p.out + (long)n * L * D + h * DH;
  const half_t* dO = p.dout + (long)n * L * D + h * DH;
  half_t* dq = p.dqkv + (long)n * L * ld + h * DH;
  half_t* dk = dq + D;
  half_t* dv = dq + 2 * D;
  half_t* xt = reinterpret_cast<half_t*>(smem + wid * WAVE_LDS);
  float* lse_s = reinterpret_cast<float*>(smem + wid * WAVE_LDS + 64 * LDS_STRIDE * 2);
  float* del_s = lse_s + 64;
  const int g = lane >> 4, c = lane & 15;
  const float* lse_g = p.lse + ((long)n * p.H + h) * L;

  if constexpr (role == 0) {
    // ---- dQ[q0 .. q0+63]: keys on the lane's rows, this block's queries on its columns
    const int q0 = blk * 64;
    h8 qf[4][2], df[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qf[t][ks] = gfrag_clamped(q, q0 + t * 16, ks, L, ld, lane);
        df[t][ks] = gfrag_clamped(dO, q0 + t * 16, ks, L, D, lane);
      }
    float lse_c[4], delta_c[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) lse_c[t] = (q0 + t * 16 + c < L) ? lse_g[q0 + t * 16 + c] : INFINITY;
    delta_of_block(df, o, q0, L, D, lane, delta_c);
    f4 acc[4][4];                                // [query tile][d-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const int nkb = p.causal ? blk + 1 : nb;
    for (int kb = 0; kb < nkb; ++kb) {
      const int k0 = kb * 64;
      h8 kf[4][2], vf[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          kf[t][ks] = gfrag_clamped(k, k0 + t * 16, ks, L, ld, lane);
          vf[t][ks] = gfrag_clamped(v, k0 + t * 16, ks, L, ld, lane);
        }
      frags_to_tile(xt, kf, lane);
      h8 kT[4][2];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kT[dt][ks] = tr_frag(xt, ks * 32, ks * 32 + 16, dt * 16, lane);
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        const int qi = q0 + qt * 16 + c;
        h4 ds16[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          f4 z = {0.f, 0.f, 0.f, 0.f};
          f4 sv = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][0], qf[qt][0], z, 0, 0, 0);
          sv = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][1], qf[qt][1], sv, 0, 0, 0);
          f4 dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[kt][0], df[qt][0], z, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[kt][1], df[qt][1], dp, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = k0 + kt * 16 + 4 * g + r;
            float pv = __expf(sv[r] * 0.125f - lse_c[qt]);
            pv = (key < L && (!p.causal || key <= qi)) ? pv : 0.f;
            ds16[kt][r] = (half_t)(pv * (dp[r] - delta_c[qt]) * 0.125f);
          }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kT[dt][ks], cat4(ds16[2 * ks], ds16[2 * ks + 1]), acc[qt][dt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const int qi = q0 + qt * 16 + c;
      store_row16(dq, ld, acc[qt], qi, qi < L, lane);
    }
  } else {
  // ---- dV (role 1) / dK (role 2) of keys k0 .. k0+63: queries on the lane's rows, this block's keys on its columns
  const int k0 = blk * 64;
  constexpr bool want_dk = role == 2;
  h8 kf[4][2], vf[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[t][ks] = gfrag_clamped(k, k0 + t * 16, ks, L, ld, lane);
      if constexpr (want_dk) vf[t][ks] = gfrag_clamped(v, k0 + t * 16, ks, L, ld, lane);
    }
  f4 acc[4][4];                                  // [key tile][d-tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  for (int qb = p.causal ? blk : 0; qb < nb; ++qb) {
    const int q0 = qb * 64;
    h8 qf[4][2], df[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qf[t][ks] = gfrag_clamped(q, q0 + t * 16, ks, L, ld, lane);
        df[t][ks] = gfrag_clamped(dO, q0 + t * 16, ks, L, D, lane);
      }
    // row statistics of this query block, redistributed through LDS from "query on the column" to "query on the row"
    if (g == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) lse_s[t * 16 + c] = (q0 + t * 16 + c < L) ? lse_g[q0 + t * 16 + c] : INFINITY;
    }
    if constexpr (want_dk) {
      float delta_c[4];
      delta_of_block(df, o, q0, L, D, lane, delta_c);
      if (g == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) del_s[t * 16 + c] = delta_c[t];
      }
    }
    // transposed operand of the output product: dO^T for dV, Q^T for dK
    frags_to_tile(xt, want_dk ? qf : df, lane);
    h8 xT[4][2];                                 // dV holds the transposed fragments; dK (more live operands) re-reads them per key tile
    if constexpr (!want_dk) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xT[dt][ks] = tr_frag(xt, ks * 32, ks * 32 + 16, dt * 16, lane);
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int key = k0 + kt * 16 + c;
      if constexpr (want_dk) asm volatile("" ::: "memory");      // re-read the row statistics per key tile instead of holding 32 VGPRs
      h4 y16[4];                                 // P^T (dV) or dS^T (dK) of this key tile, per query tile
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        f4 z = {0.f, 0.f, 0.f, 0.f};
        f4 sv = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[qt][0], kf[kt][0], z, 0, 0, 0);
        sv = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[qt][1], kf[kt][1], sv, 0, 0, 0);
        f4 dp = z;
        if constexpr (want_dk) {
          dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(df[qt][0], vf[kt][0], z, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(df[qt][1], vf[kt][1], dp, 0, 0, 0);
        }
        const f4 lr = *reinterpret_cast<const f4*>(lse_s + qt * 16 + 4 * g);
        f4 dr = z;
        if constexpr (want_dk) dr = *reinterpret_cast<const f4*>(del_s + qt * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = q0 + qt * 16 + 4 * g + r;
          float pv = __expf(sv[r] * 0.125f - lr[r]);
          pv = (key < L && (!p.causal || key <= qi)) ? pv : 0.f;
          y16[qt][r] = want_dk ? (half_t)(pv * (dp[r] - dr[r]) * 0.125f) : (half_t)pv;
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          acc[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(want_dk ? tr_frag(xt, ks * 32, ks * 32 + 16, dt * 16, lane) : xT[dt][ks],
                                                               cat4(y16[2 * ks], y16[2 * ks + 1]), acc[kt][dt], 0, 0, 0);
    }
  }
  half_t* dst = want_dk ? dk : dv;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int key = k0 + kt * 16 + c;
    store_row16(dst, ld, acc[kt], key, key < L, lane);
  }
  }
}

}  // namespace

template <int KTL, bool CAUSAL>
static void launch_long_fwd2(const AttnArgs& p, hipStream_t stream) {
  constexpr int LDS = (2 * 16 * KTL + 4 * 16) * LDS_STRIDE * 2;            // K, V tiles + one 16-row staging tile per wave
  if (LDS > 64 * 1024) {
    static bool done[HMMC_MAX_DEVICES] = {false};
    hmmc_allow_lds((const void*)attn_long_fwd_kernel<KTL, CAUSAL>, LDS, done);
  }
  hipLaunchKernelGGL((attn_long_fwd_kernel<KTL, CAUSAL>), dim3((unsigned)(p.nseq * p.H)), dim3(256), LDS, stream, p);
}
template <int KTL>
static void launch_long_fwd(const AttnArgs& p, hipStream_t stream) {
  if (p.causal) launch_long_fwd2<KTL, true>(p, stream); else launch_long_fwd2<KTL, false>(p, stream);
}

int hmmc_attention_long_fwd(const AttnArgs& p, hipStream_t stream) {
  if ((long)p.nseq * p.H >= (1l << 31)) return HMMC_ERR_UNSUPPORTED;
  const int ktl = ((p.L + 31) / 32) * 2;                                    // key tiles, rounded up to an even count
  switch (ktl) {
    case 6: launch_long_fwd<6>(p, stream); break;
    case 8: launch_long_fwd<8>(p, stream); break;
    case 10: launch_long_fwd<10>(p, stream); break;
    case 12: launch_long_fwd<12>(p, stream); break;
    case 14: launch_long_fwd<14>(p, stream); break;
    case 16: launch_long_fwd<16>(p, stream); break;
    default: return HMMC_ERR_UNSUPPORTED;
  }
  return hmmc_launch_status();
}

int hmmc_attention_long_bwd(const AttnArgs& p, hipStream_t stream) {
  const int nb = (p.L + 63) / 64;
  const long waves = (long)p.nseq * p.H * nb;
  const int lds = 4 * (64 * LDS_STRIDE * 2 + 2 * 64 * 4);
  const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  hipLaunchKernelGGL(attn_long_bwd_kernel<2>, grid, block, lds, stream, p);     // dK (the longest role) first
  hipLaunchKernelGGL(attn_long_bwd_kernel<0>, grid, block, lds, stream, p);     // dQ
  hipLaunchKernelGGL(attn_long_bwd_kernel<1>, grid, block, lds, stream, p);     // dV
  return hmmc_launch_status();
}
