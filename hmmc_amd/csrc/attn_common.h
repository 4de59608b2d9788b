// Shared device helpers of the fused attention kernels (attention_f16.hip, attention_long_f16.hip).
#pragma once
#include "common.h"

// plain struct at global scope: shared by the two translation units and their host launchers
struct HmmcAttnArgs {
  const half_t* qkv; half_t* out; float* lse;
  const half_t* dout; half_t* dqkv;
  float* dbias;                 // optional [nseq][3D]: per-sequence column sums of dqkv (in_proj bias gradient partials)
  const float* rowstat;         // optional [nseq * L][2]: the backward writes rowstat[token][0] x dqkv[token][:] (the rstd of a folded
                                // ln_1, ln_fold.hip); the bias partials stay those of the unscaled gradient
  int nseq, L, H, causal;
};
typedef HmmcAttnArgs AttnArgs;

namespace {

constexpr int DH = 64;          // head dim of every CLIP tower
constexpr int LDS_STRIDE = 72;  // halves per LDS row (64 + 8 pad): 144 B, keeps 16-B alignment

typedef __attribute__((address_space(3))) fp16x4* lds_tr_ptr;

__device__ __forceinline__ h4 tr_read(const half_t* p) {
  fp16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_tr_ptr)LDS_PTR(p));
  h4 r;
  r[0] = (half_t)t[0]; r[1] = (half_t)t[1]; r[2] = (half_t)t[2]; r[3] = (half_t)t[3];
  return r;
}
__device__ __forceinline__ h8 cat4(h4 a, h4 b) {
  h8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

// transposed fragment from an LDS tile [r][c] (stride LDS_STRIDE): 8 halves T[kperm][c0 + (lane&15)]
// with rows r = rA + 4*(lane>>4) + j (j<4) and rB + 4*(lane>>4) + (j-4) (j>=4)
__device__ __forceinline__ h8 tr_frag(const half_t* tile, int rA, int rB, int c0, int lane) {
  int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
  h4 lo = tr_read(tile + (rA + 4 * g + qq) * LDS_STRIDE + c0 + 4 * pp);
  h4 hi = tr_read(tile + (rB + 4 * g + qq) * LDS_STRIDE + c0 + 4 * pp);
  return cat4(lo, hi);
}


typedef unsigned u4v __attribute__((ext_vector_type(4)));

// X^T[d = dt*16 + 4g + r][row = c] accumulators (4 d-tiles) of the 16 rows row0 .. row0 + 15 -> global [row][64].  The
// strip goes through a 16 x 144 B LDS scratch (8-byte writes from the MFMA layout, 16-byte reads in lane order) so that
// every store instruction writes 8 full 128-byte rows with consecutive lanes on consecutive bytes; a wave's LDS
// operations execute in order, so strips follow each other without a wait.
// scale: a factor for this lane's row (row0 + c), applied to the value stored only.
__device__ __forceinline__ void store_rows(half_t* dst, long ld, const f4 (&acc)[4], int row0, int L, half_t* scr, int lane,
                                           float scale = 1.0f) {
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    h4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (half_t)(acc[dt][r] * scale);
    *reinterpret_cast<h4*>(scr + c * LDS_STRIDE + dt * 16 + 4 * g) = v;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = (lane >> 3) + 8 * t;
    const u4v o = *reinterpret_cast<const u4v*>(scr + r * LDS_STRIDE + 8 * (lane & 7));
    if (row0 + r < L) *reinterpret_cast<u4v*>(dst + (long)(row0 + r) * ld + 8 * (lane & 7)) = o;
  }
}

// column sums over the 16 lanes c of a 16-lane row (every lane ends with the total)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
// csum[dt][r] += the fp16-rounded value of acc[dt][r] (what the stored tensor holds)
__device__ __forceinline__ void add_rounded(f4 (&csum)[4], const f4 (&acc)[4]) {
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) csum[dt][r] += (float)(half_t)acc[dt][r];
}
// dst[dt*16 + 4g + r] = sum over the 16 lanes c of csum[dt][r]
__device__ __forceinline__ void store_colsum(float* dst, f4 (&csum)[4], int lane) {
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    f4 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = row16_sum(csum[dt][r]);
    if ((lane & 15) == 0) *reinterpret_cast<f4*>(dst + dt * 16 + 4 * (lane >> 4)) = t;
  }
}

}  // namespace
