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

// Forward: one wave per (sequence, head, 64-query block), online softmax over 64-key blocks.  Q, K and V rows go from
// global memory straight into MFMA fragments; only V passes through the wave's LDS tile (transposed reads for O^T = V^T P^T),
// so a workgroup of four waves needs 36 KiB and two waves per SIMD stay resident.
__global__ __launch_bounds__(256, 2) void attn_long_fwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nqb = (p.L + 63) / 64;
  const long idx = (long)blockIdx.x * 4 + wid;
  if (idx >= (long)p.nseq * p.H * nqb) return;
  const int qb = (int)(idx % nqb);
  const long pair = idx / nqb;
  const int n = (int)(pair / p.H), h = (int)(pair % p.H);
  const int D = p.H * DH, L = p.L;
  const long ld = 3L * D;
  const half_t* q = p.qkv + (long)n * L * ld + h * DH;
  const half_t* k = q + D;
  const half_t* v = q + 2 * D;
  half_t* vtile = reinterpret_cast<half_t*>(smem) + wid * (64 * LDS_STRIDE);
  const int g = lane >> 4, c = lane & 15;
  const int q0 = qb * 64;
  h8 qf[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[t][ks] = gfrag_clamped(q, q0 + t * 16, ks, L, ld, lane);
  float m[4], l[4];
  f4 acc[4][4];                                  // [d-tile][query tile]
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    m[qt] = -INFINITY; l[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt][qt] = f4{0.f, 0.f, 0.f, 0.f};
  }
  const int nkb = p.causal ? qb + 1 : nqb;
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
    frags_to_tile(vtile, vf, lane);
    h8 vT[4][2];                                 // V^T fragments, k-order permuted like the P^T accumulators
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) vT[dt][ks] = tr_frag(vtile, ks * 32, ks * 32 + 16, dt * 16, lane);
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const int qi = q0 + qt * 16 + c;
      f4 s[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f4 z = {0.f, 0.f, 0.f, 0.f};
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][0], qf[qt][0], z, 0, 0, 0);
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][1], qf[qt][1], s[kt], 0, 0, 0);
      }
      float bm = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = k0 + kt * 16 + 4 * g + r;
          const bool ok = key < L && (!p.causal || key <= qi || qi >= L);
          const float val = ok ? s[kt][r] * 0.125f : -INFINITY;
          s[kt][r] = val;
          bm = fmaxf(bm, val);
        }
      bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
      bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
      const float mn = fmaxf(m[qt], bm);
      const float alpha = (mn == -INFINITY) ? 1.f : __expf(m[qt] - mn);
      float sum = 0.f;
      h4 pt[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = (mn == -INFINITY) ? 0.f : __expf(s[kt][r] - mn);
          pt[kt][r] = (half_t)e;
          sum += e;
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      l[qt] = l[qt] * alpha + sum;
      m[qt] = mn;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f4 a = acc[dt][qt] * alpha;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(vT[dt][ks], cat4(pt[2 * ks], pt[2 * ks + 1]), a, 0, 0, 0);
        acc[dt][qt] = a;
      }
    }
  }
  half_t* o = p.out + (long)n * L * D + h * DH;
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const int qi = q0 + qt * 16 + c;
    const float inv = 1.0f / l[qt];
    if (g == 0 && qi < L) p.lse[((long)n * p.H + h) * L + qi] = m[qt] + __logf(l[qt]);
    f4 t[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) t[dt] = acc[dt][qt] * inv;
    store_row16(o, D, t, qi, qi < L, lane);
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

int hmmc_attention_long_fwd(const AttnArgs& p, hipStream_t stream) {
  const int nqb = (p.L + 63) / 64;
  long waves = (long)p.nseq * p.H * nqb;
  const int lds = 4 * 64 * LDS_STRIDE * 2;
  hipLaunchKernelGGL(attn_long_fwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), lds, stream, p);
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
