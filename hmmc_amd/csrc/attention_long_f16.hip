// Fused attention for sequences of 65..256 tokens (ViT-B/16: 197 tokens per frame; CLIP text up to 77):
// the same "key on the row, query on the column" MFMA orientation as attention_f16.hip, tiled 64 x 64 with an
// online softmax over key blocks (forward) and block-wise recomputation from the saved log-sum-exp (backward).
// One wave owns one (sequence, head[, query block]); all LDS is wave-private, no workgroup barriers.
// Reference: nn.MultiheadAttention core at modules/module_clip.py:251 with 197 x 197 heads.
#include "attn_common.h"

namespace {

constexpr int TILE = 64 * LDS_STRIDE;      // halves per 64-row LDS tile

// S^T block [4 key tiles][4 query tiles] = K_blk Q_blk^T with K and Q rows read from LDS tiles
__device__ __forceinline__ void st_block(const half_t* ktile, const half_t* qtile, f4 (&s)[4][4], int lane) {
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const half_t* qr = qtile + (qt * 16 + c) * LDS_STRIDE + 8 * g;
    h8 q0 = *reinterpret_cast<const h8*>(qr), q1 = *reinterpret_cast<const h8*>(qr + 32);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const half_t* kr = ktile + (kt * 16 + c) * LDS_STRIDE + 8 * g;
      h8 k0 = *reinterpret_cast<const h8*>(kr), k1 = *reinterpret_cast<const h8*>(kr + 32);
      f4 a = {0.f, 0.f, 0.f, 0.f};
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, q0, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, q1, a, 0, 0, 0);
      s[kt][qt] = a;
    }
  }
}

// acc[dt][col tile] += X^T[d][k] * Y[k][col]: X from an LDS tile via transposed reads, Y from registers
// (k order: tiles 2s, 2s+1 interleaved as produced by the accumulator layout)
__device__ __forceinline__ void acc_tr_regs(const half_t* xtile, const h4 (&y)[4][4], f4 (&acc)[4][4], int lane) {
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    h8 x0 = tr_frag(xtile, 0, 16, dt * 16, lane), x1 = tr_frag(xtile, 32, 48, dt * 16, lane);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      acc[dt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x0, cat4(y[0][ct], y[1][ct]), acc[dt][ct], 0, 0, 0);
      acc[dt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x1, cat4(y[2][ct], y[3][ct]), acc[dt][ct], 0, 0, 0);
    }
  }
}

// acc[dt][kt] += X^T[d][q] * Y[q][key]: X via transposed reads, Y staged in LDS as [key][q]
__device__ __forceinline__ void acc_tr_lds(const half_t* xtile, const half_t* ytile, f4 (&acc)[4][4], int lane) {
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    h8 x0 = tr_frag(xtile, 0, 16, dt * 16, lane), x1 = tr_frag(xtile, 32, 48, dt * 16, lane);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const half_t* yr = ytile + (kt * 16 + c) * LDS_STRIDE + 4 * g;
      h4 a0 = *reinterpret_cast<const h4*>(yr), a1 = *reinterpret_cast<const h4*>(yr + 16);
      h4 b0 = *reinterpret_cast<const h4*>(yr + 32), b1 = *reinterpret_cast<const h4*>(yr + 48);
      acc[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x0, cat4(a0, a1), acc[dt][kt], 0, 0, 0);
      acc[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x1, cat4(b0, b1), acc[dt][kt], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void stage_t(half_t* ytile, const h4 (&v)[4][4], int lane) {
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) ytile[(kt * 16 + 4 * g + r) * LDS_STRIDE + qt * 16 + c] = v[kt][qt][r];
}

__device__ __forceinline__ void store_t(half_t* dst, long ld, int row0, int L, const f4 (&acc)[4][4], int lane) {
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      int row = row0 + ct * 16 + c;
      if (row < L) {
        h4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)acc[dt][ct][r];
        *reinterpret_cast<h4*>(dst + (long)row * ld + dt * 16 + 4 * g) = o;
      }
    }
}

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

// recompute P^T and dS^T of one (query block, key block) from LDS tiles and the saved row statistics
__device__ __forceinline__ void pds_block(const half_t* ktile, const half_t* vtile, const half_t* qtile, const half_t* dotile,
                                          const float* lse, const float* delta, int q0, int k0, int L, int causal,
                                          h4 (&pt)[4][4], h4 (&dst)[4][4], int lane) {
  const int g = lane >> 4, c = lane & 15;
  f4 s[4][4], dp[4][4];
  st_block(ktile, qtile, s, lane);
  st_block(vtile, dotile, dp, lane);
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const int ql = qt * 16 + c, qi = q0 + ql;
    const float ls = lse[q0 + ql], de = delta[q0 + ql];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = k0 + kt * 16 + 4 * g + r;
        bool ok = key < L && qi < L && (!causal || key <= qi);
        float pv = ok ? __expf(s[kt][qt][r] * 0.125f - ls) : 0.f;
        pt[kt][qt][r] = (half_t)pv;
        dst[kt][qt][r] = (half_t)(pv * (dp[kt][qt][r] - de) * 0.125f);
      }
  }
}

// one wave per (sequence, head): phase 0 row statistics, phase 1 dK/dV per key block, phase 2 dQ per query block
__global__ __launch_bounds__(64) void attn_long_bwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const long pair = blockIdx.x;
  const int n = (int)(pair / p.H), h = (int)(pair % p.H);
  const int D = p.H * DH, L = p.L;
  const long ld = 3L * D;
  const int nb = (L + 63) / 64;
  const half_t* q = p.qkv + (long)n * L * ld + h * DH;
  const half_t* k = q + D;
  const half_t* v = q + 2 * D;
  const half_t* o = p.out + (long)n * L * D + h * DH;
  const half_t* dO = p.dout + (long)n * L * D + h * DH;
  half_t* dq = p.dqkv + (long)n * L * ld + h * DH;
  half_t* dk = dq + D;
  half_t* dv = dq + 2 * D;
  half_t* base = reinterpret_cast<half_t*>(smem);
  half_t* ktile = base; half_t* vtile = base + TILE; half_t* qtile = base + 2 * TILE; half_t* dotile = base + 3 * TILE;
  half_t* ytile = base + 4 * TILE;
  float* lse = reinterpret_cast<float*>(base + 5 * TILE);
  float* delta = lse + 256;
  // phase 0: delta[q] = <dO[q], O[q]>, lse[q]
  for (int r = lane; r < 256; r += 64) {
    float d = 0.f;
    if (r < L) {
      const half_t* a = dO + (long)r * D;
      const half_t* b = o + (long)r * D;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        h8 x = *reinterpret_cast<const h8*>(a + ch * 8), y = *reinterpret_cast<const h8*>(b + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) d += (float)x[e] * (float)y[e];
      }
    }
    delta[r] = d;
    lse[r] = r < L ? p.lse[((long)n * p.H + h) * L + r] : 0.f;
  }
  // phase 1: dK, dV
  for (int kb = 0; kb < nb; ++kb) {
    const int k0 = kb * 64;
    load_tile<64>(ktile, k + (long)k0 * ld, L - k0, ld, lane);
    load_tile<64>(vtile, v + (long)k0 * ld, L - k0, ld, lane);
    f4 dka[4][4], dva[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { dka[i][j] = f4{0.f, 0.f, 0.f, 0.f}; dva[i][j] = f4{0.f, 0.f, 0.f, 0.f}; }
    for (int qb = p.causal ? kb : 0; qb < nb; ++qb) {
      const int q0 = qb * 64;
      load_tile<64>(qtile, q + (long)q0 * ld, L - q0, ld, lane);
      load_tile<64>(dotile, dO + (long)q0 * D, L - q0, D, lane);
      h4 pt[4][4], dst[4][4];
      pds_block(ktile, vtile, qtile, dotile, lse, delta, q0, k0, L, p.causal, pt, dst, lane);
      stage_t(ytile, pt, lane);
      acc_tr_lds(dotile, ytile, dva, lane);
      stage_t(ytile, dst, lane);
      acc_tr_lds(qtile, ytile, dka, lane);
    }
    store_t(dv, ld, k0, L, dva, lane);
    store_t(dk, ld, k0, L, dka, lane);
  }
  // phase 2: dQ
  for (int qb = 0; qb < nb; ++qb) {
    const int q0 = qb * 64;
    load_tile<64>(qtile, q + (long)q0 * ld, L - q0, ld, lane);
    load_tile<64>(dotile, dO + (long)q0 * D, L - q0, D, lane);
    f4 dqa[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) dqa[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const int nkb = p.causal ? qb + 1 : nb;
    for (int kb = 0; kb < nkb; ++kb) {
      const int k0 = kb * 64;
      load_tile<64>(ktile, k + (long)k0 * ld, L - k0, ld, lane);
      load_tile<64>(vtile, v + (long)k0 * ld, L - k0, ld, lane);
      h4 pt[4][4], dst[4][4];
      pds_block(ktile, vtile, qtile, dotile, lse, delta, q0, k0, L, p.causal, pt, dst, lane);
      acc_tr_regs(ktile, dst, dqa, lane);
    }
    store_t(dq, ld, q0, L, dqa, lane);
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
  const int lds = 5 * TILE * 2 + 2 * 256 * 4;
  static bool once = (hmmc_allow_lds((const void*)attn_long_bwd_kernel, lds), true);
  (void)once;
  hipLaunchKernelGGL(attn_long_bwd_kernel, dim3((unsigned)((long)p.nseq * p.H)), dim3(64), lds, stream, p);
  return hmmc_launch_status();
}
