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


template <int KT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
  constexpr int LP = 16 * KT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long pair = (long)blockIdx.x * 4 + wid;
  if (pair >= (long)p.nseq * p.H) return;
  const int n = (int)(pair / p.H), h = (int)(pair % p.H);
  const int D = p.H * DH, L = p.L;
  const long ld = 3L * D;
  const half_t* q = p.qkv + (long)n * L * ld + h * DH;
  const half_t* k = q + D;
  const half_t* v = q + 2 * D;
  half_t* ktile = reinterpret_cast<half_t*>(smem) + wid * (2 * LP * LDS_STRIDE);
  half_t* vtile = ktile + LP * LDS_STRIDE;
  load_tile<LP>(ktile, k, L, ld, lane);
  load_tile<LP>(vtile, v, L, ld, lane);

  const int g = lane >> 4, c = lane & 15;
  // S^T[key][q] = sum_d K[key][d] Q[q][d]
  f4 s[KT][KT];
#pragma unroll
  for (int qt = 0; qt < KT; ++qt) {
    h8 qf0 = gfrag(q, qt * 16, 0, L, ld, lane), qf1 = gfrag(q, qt * 16, 1, L, ld, lane);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const half_t* kr = ktile + (kt * 16 + c) * LDS_STRIDE + 8 * g;
      h8 kf0 = *reinterpret_cast<const h8*>(kr), kf1 = *reinterpret_cast<const h8*>(kr + 32);
      f4 a = {0.f, 0.f, 0.f, 0.f};
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf0, qf0, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf1, qf1, a, 0, 0, 0);
      s[kt][qt] = a;
    }
  }
  // softmax over keys for query column (qt*16 + c); lane holds keys kt*16 + 4g + r
  h4 pt[KT][KT];
#pragma unroll
  for (int qt = 0; qt < KT; ++qt) {
    const int qi = qt * 16 + c;
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = kt * 16 + 4 * g + r;
        bool ok = key < L && (!p.causal || key <= qi || qi >= L);
        float val = ok ? s[kt][qt][r] * 0.125f : -INFINITY;
        s[kt][qt][r] = val;
        m = fmaxf(m, val);
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = __expf(s[kt][qt][r] - m);
        s[kt][qt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (p.lse && g == 0 && qi < L) p.lse[((long)n * p.H + h) * L + qi] = m + __logf(sum);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pt[kt][qt][r] = (half_t)(s[kt][qt][r] * inv);
  }
  // O^T[d][q] = sum_key V[key][d] P[q][key]; k-step s covers key tiles 2s, 2s+1 in permuted order
  half_t* o = p.out + (long)n * L * D + h * DH;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    h8 vf[KT / 2];
#pragma unroll
    for (int ks = 0; ks < KT / 2; ++ks) vf[ks] = tr_frag(vtile, ks * 32, ks * 32 + 16, dt * 16, lane);
#pragma unroll
    for (int qt = 0; qt < KT; ++qt) {
      f4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KT / 2; ++ks)
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[ks], cat4(pt[2 * ks][qt], pt[2 * ks + 1][qt]), a, 0, 0, 0);
      int qi = qt * 16 + c;
      if (qi < L) {
        h4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = (half_t)a[r];
        *reinterpret_cast<h4*>(o + (long)qi * D + dt * 16 + 4 * g) = ov;
      }
    }
  }
}

// Backward.  One wave per (sequence, head); two LDS buffers per wave (tile X: K, then dO, then Q;
// tile Y: P^T then dS^T, both stored [key][q]).
template <int KT>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs p) {
  constexpr int LP = 16 * KT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long pair = (long)blockIdx.x * 4 + wid;
  if (pair >= (long)p.nseq * p.H) return;
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
  half_t* xt = reinterpret_cast<half_t*>(smem) + wid * (2 * LP * LDS_STRIDE);
  half_t* yt = xt + LP * LDS_STRIDE;
  const int g = lane >> 4, c = lane & 15;

  load_tile<LP>(xt, k, L, ld, lane);   // X = K (needed transposed for dQ)

  // phase A: P^T and dS^T in registers
  h4 pt[KT][KT], dst[KT][KT];
#pragma unroll
  for (int qt = 0; qt < KT; ++qt) {
    const int qi = qt * 16 + c;
    h8 qf0 = gfrag(q, qt * 16, 0, L, ld, lane), qf1 = gfrag(q, qt * 16, 1, L, ld, lane);
    h8 df0 = gfrag(dO, qt * 16, 0, L, D, lane), df1 = gfrag(dO, qt * 16, 1, L, D, lane);
    h8 of0 = gfrag(o, qt * 16, 0, L, D, lane), of1 = gfrag(o, qt * 16, 1, L, D, lane);
    float delta = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) delta += (float)df0[j] * (float)of0[j] + (float)df1[j] * (float)of1[j];
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    const float lse = qi < L ? p.lse[((long)n * p.H + h) * L + qi] : 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const half_t* kr = xt + (kt * 16 + c) * LDS_STRIDE + 8 * g;
      h8 kf0 = *reinterpret_cast<const h8*>(kr), kf1 = *reinterpret_cast<const h8*>(kr + 32);
      h8 vf0 = gfrag(v, kt * 16, 0, L, ld, lane), vf1 = gfrag(v, kt * 16, 1, L, ld, lane);
      f4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf0, qf0, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf1, qf1, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf0, df0, dp, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf1, df1, dp, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = kt * 16 + 4 * g + r;
        bool ok = key < L && qi < L && (!p.causal || key <= qi);
        float pv = ok ? __expf(s[r] * 0.125f - lse) : 0.f;
        pt[kt][qt][r] = (half_t)pv;
        dst[kt][qt][r] = (half_t)(pv * (dp[r] - delta) * 0.125f);
      }
    }
  }
  // dQ^T[d][q] = sum_key K[key][d] dS[q][key]   (K^T via transposed reads of X)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    h8 kf[KT / 2];
#pragma unroll
    for (int ks = 0; ks < KT / 2; ++ks) kf[ks] = tr_frag(xt, ks * 32, ks * 32 + 16, dt * 16, lane);
#pragma unroll
    for (int qt = 0; qt < KT; ++qt) {
      f4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KT / 2; ++ks)
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[ks], cat4(dst[2 * ks][qt], dst[2 * ks + 1][qt]), a, 0, 0, 0);
      int qi = qt * 16 + c;
      if (qi < L) {
        h4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = (half_t)a[r];
        *reinterpret_cast<h4*>(dq + (long)qi * ld + dt * 16 + 4 * g) = ov;
      }
    }
  }
  // dV^T[d][key] = sum_q dO[q][d] P[q][key]: X = dO (transposed reads), Y = P^T as [key][q]
  // dK^T[d][key] = sum_q Q[q][d] dS[q][key]: X = Q,                      Y = dS^T
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    load_tile<LP>(xt, pass == 0 ? dO : q, L, pass == 0 ? (long)D : ld, lane);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int qt = 0; qt < KT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          yt[(kt * 16 + 4 * g + r) * LDS_STRIDE + qt * 16 + c] = pass == 0 ? pt[kt][qt][r] : dst[kt][qt][r];
    half_t* dst_ptr = pass == 0 ? dv : dk;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      h8 xf[KT / 2];
#pragma unroll
      for (int ks = 0; ks < KT / 2; ++ks) xf[ks] = tr_frag(xt, ks * 32 + 4 * 0, ks * 32 + 16, dt * 16, lane);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        f4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KT / 2; ++ks) {
          // B fragment: Y[key = kt*16 + c][q in the same permuted order as tr_frag's rows]
          const half_t* yr = yt + (kt * 16 + c) * LDS_STRIDE + ks * 32 + 4 * g;
          h4 lo = *reinterpret_cast<const h4*>(yr), hi = *reinterpret_cast<const h4*>(yr + 16);
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[ks], cat4(lo, hi), a, 0, 0, 0);
        }
        int key = kt * 16 + c;
        if (key < L) {
          h4 ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (half_t)a[r];
          *reinterpret_cast<h4*>(dst_ptr + (long)key * ld + dt * 16 + 4 * g) = ov;
        }
      }
    }
  }
}

}  // namespace

int hmmc_attention_long_fwd(const AttnArgs& p, hipStream_t stream);
int hmmc_attention_long_bwd(const AttnArgs& p, hipStream_t stream);

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
  dim3 grid((unsigned)((pairs + 3) / 4)), block(256);
  static bool once = (hmmc_allow_lds((const void*)attn_fwd_kernel<4>, 4 * 2 * 64 * LDS_STRIDE * 2), true);
  (void)once;
  if (L <= 32) hipLaunchKernelGGL(attn_fwd_kernel<2>, grid, block, 4 * 2 * 32 * LDS_STRIDE * 2, stream, p);
  else hipLaunchKernelGGL(attn_fwd_kernel<4>, grid, block, 4 * 2 * 64 * LDS_STRIDE * 2, stream, p);
  return hmmc_launch_status();
}

extern "C" int hmmc_attention_f16_bwd(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv,
                                      int nseq, int L, int H, int causal, hipStream_t stream) {
  if (!qkv || !out || !lse || !dout || !dqkv || nseq <= 0 || L <= 0 || H <= 0) return HMMC_ERR_ARG;
  if (L > 256) return HMMC_ERR_UNSUPPORTED;
  if (L > 64) {
    AttnArgs pl{};
    pl.qkv = (const half_t*)qkv; pl.out = (half_t*)out; pl.lse = (float*)lse; pl.dout = (const half_t*)dout;
    pl.dqkv = (half_t*)dqkv; pl.nseq = nseq; pl.L = L; pl.H = H; pl.causal = causal;
    return hmmc_attention_long_bwd(pl, stream);
  }
  AttnArgs p{};
  p.qkv = (const half_t*)qkv; p.out = (half_t*)out; p.lse = (float*)lse; p.dout = (const half_t*)dout;
  p.dqkv = (half_t*)dqkv; p.nseq = nseq; p.L = L; p.H = H; p.causal = causal;
  long pairs = (long)nseq * H;
  dim3 grid((unsigned)((pairs + 3) / 4)), block(256);
  static bool once = (hmmc_allow_lds((const void*)attn_bwd_kernel<4>, 4 * 2 * 64 * LDS_STRIDE * 2), true);
  (void)once;
  if (L <= 32) hipLaunchKernelGGL(attn_bwd_kernel<2>, grid, block, 4 * 2 * 32 * LDS_STRIDE * 2, stream, p);
  else hipLaunchKernelGGL(attn_bwd_kernel<4>, grid, block, 4 * 2 * 64 * LDS_STRIDE * 2, stream, p);
  return hmmc_launch_status();
}
