// Fused attention for sequences of 65..256 tokens (ViT-B/16: 197 tokens per frame; CLIP text up to 77): the same
// "key on the row, query on the column" MFMA orientation as attention_f16.hip, one workgroup per (sequence, head) with the
// head's operands resident in LDS (one barrier after the load, none afterwards) and 16-row tiles dealt to the waves.
// Reference: nn.MultiheadAttention core at modules/module_clip.py:251 with 197 x 197 heads.
#include "attn_common.h"

namespace {

// Forward: one WORKGROUP per (sequence, head).  Its four waves first bring the head's K and V (all L <= 256 keys: 2 x 36 KiB
// at most) into LDS with full-row, lane-ordered loads, once; then each wave takes 16-query tiles (tile w, w + 4, ...).  A
// whole row of scores - KTL key tiles x 4 values per lane - fits the registers, so the softmax is the one-block kernel's
// (attention_f16.hip: in-register maximum and sum, two 16-lane shuffles, no online rescaling), K fragments come from LDS with
// ds_read_b128 and V^T fragments with ds_read_b64_tr_b16.  Against the previous wave-private 64 x 64 blocks (each wave
// fetching every K / V block itself, in 16-byte pieces, with nothing in flight while it computed) this reads K and V once
// per head instead of four times and has no global load between the first MFMA and the last.
constexpr float LOG2E = 1.4426950408889634f;

// LDS images of the head's operands: [rows][64 halves] = 128-byte rows WITHOUT padding (57 KiB for K and V of 224 rows, so two
// workgroups fit a CU), the 16-byte chunk index XORed with a key of the row so that both access patterns are conflict-free:
//   "row" image (read with ds_read_b128: 16 lanes on 16 consecutive rows, one chunk each): key = (row >> 1) & 7
//   "tr" image (read with ds_read_b64_tr_b16: 8 consecutive rows x 32 contiguous bytes per 32 lanes): key = ((row >> 1) & 3) << 1
__device__ __forceinline__ int swz_row(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_tr(int row) { return ((row >> 1) & 3) << 1; }
// 8 halves op[row][8 chunk .. +7] of a "row" image
__device__ __forceinline__ h8 row_frag(const half_t* tile, int row, int chunk) {
  return *reinterpret_cast<const h8*>(tile + row * DH + ((chunk ^ swz_row(row)) << 3));
}
// transposed fragment of a "tr" image: T[k][c0 + (lane & 15)], k = rows rA + 4g + j (j < 4) and rB + 4g + j - 4
__device__ __forceinline__ h8 tr_frag_swz(const half_t* tile, int rA, int rB, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
  const int ra = rA + 4 * g + qq, rb = rB + 4 * g + qq;
  const int chunk = (c0 >> 3) + (pp >> 1), sub = 4 * (pp & 1);
  h4 lo = tr_read(tile + ra * DH + ((chunk ^ swz_tr(ra)) << 3) + sub);
  h4 hi = tr_read(tile + rb * DH + ((chunk ^ swz_tr(rb)) << 3) + sub);
  return cat4(lo, hi);
}

constexpr int fwd_lds_bytes(int KTL, int NW) { return 2 * 16 * KTL * DH * 2 + NW * 16 * LDS_STRIDE * 2; }   // K, V images + a staging tile per wave
// waves per SIMD the register allocation must leave room for: the workgroups the LDS lets a CU hold x NW / 4
constexpr int fwd_min_waves(int KTL, int NW) { return (160 * 1024 / fwd_lds_bytes(KTL, NW) >= 2 ? 2 : 1) * NW / 4; }

// LEAD (query 0 of every sequence only, attention_f16.hip): wave 0 runs query tile 0 with the tile's other fifteen rows as zeros -
// the same instruction sequence as the all-query kernel on that tile, so row 0 and its log-sum-exp are bit-identical - and stores
// that row alone; the other waves only help bring K and V in.
template <int KTL, int NW, bool CAUSAL, bool LEAD = false>
__global__ __launch_bounds__(64 * NW, fwd_min_waves(KTL, NW)) void attn_long_fwd_kernel(AttnArgs p) {
  constexpr int ROWS = 16 * KTL, NT = 64 * NW, RPP = NT / 8, PASSES = (ROWS + RPP - 1) / RPP;
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
  half_t* vtile = ktile + ROWS * DH;
  half_t* scr = vtile + ROWS * DH + wid * (16 * LDS_STRIDE);
  const int g = lane >> 4, c = lane & 15;
  {                                              // K, V -> LDS: thread -> row (tid >> 3) + RPP i, chunk tid & 7; rows past L zero
    u4v rk[PASSES], rv[PASSES];
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int row = (tid >> 3) + RPP * i;
      const long off = (long)min(row, L - 1) * ld + 8 * (tid & 7);
      rk[i] = *reinterpret_cast<const u4v*>(k + off);
      rv[i] = *reinterpret_cast<const u4v*>(v + off);
    }
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int row = (tid >> 3) + RPP * i, ch = tid & 7;
      const u4v z = {0u, 0u, 0u, 0u};
      if (ROWS % RPP == 0 || row < ROWS) {
        *reinterpret_cast<u4v*>(ktile + row * DH + ((ch ^ swz_row(row)) << 3)) = row < L ? rk[i] : z;
        *reinterpret_cast<u4v*>(vtile + row * DH + ((ch ^ swz_tr(row)) << 3)) = row < L ? rv[i] : z;
      }
    }
  }
  __syncthreads();
  half_t* o = p.out + (long)n * L * D + h * DH;
  const int nqt = LEAD ? 1 : (L + 15) / 16;
  const int LQ = LEAD ? 1 : L;                   // queries whose results are stored
  // a tile's 16 query rows: lane-ordered loads (requested one tile ahead) -> scratch -> row fragments
  auto load_q = [&](int qt, u4v (&raw)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if constexpr (LEAD) {                      // row 0 alone is read; the tile's other rows are zeros
        const u4v v0 = *reinterpret_cast<const u4v*>(q + 8 * (lane & 7)), z = {0u, 0u, 0u, 0u};
        raw[t] = (t == 0 && lane < 8) ? v0 : z;
      } else {
        raw[t] = *reinterpret_cast<const u4v*>(q + (long)min(qt * 16 + (lane >> 3) + 8 * t, L - 1) * ld + 8 * (lane & 7));
      }
    }
  };
  constexpr float C1 = 0.125f * LOG2E;
  u4v qraw[2], qnext[2];
  load_q(min(wid, nqt - 1), qraw);
  for (int qt = wid; qt < nqt; qt += NW) {
    const int q0 = qt * 16, qi = q0 + c;
    load_q(min(qt + NW, nqt - 1), qnext);
#pragma unroll
    for (int t = 0; t < 2; ++t)
      *reinterpret_cast<u4v*>(scr + ((lane >> 3) + 8 * t) * LDS_STRIDE + 8 * (lane & 7)) = qraw[t];
    h8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const h8*>(scr + c * LDS_STRIDE + ks * 32 + 8 * g);
    // S^T[key][q]: lane holds keys kt*16 + 4g + r of query column qi, for every key tile; two key tiles per round
    f4 s[KTL];
    float m = -INFINITY;
#pragma unroll
    for (int k0 = 0; k0 < KTL; k0 += 2) {
      h8 kf[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kf[i][ks] = row_frag(ktile, (k0 + i) * 16 + c, ks * 4 + g);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kt = k0 + i;
        f4 z = {0.f, 0.f, 0.f, 0.f};
        z = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i][0], qf[0], z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i][1], qf[1], z, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * g + r;
          // keys past L can only sit in the last two tiles (KTL = key tiles rounded up to an even count): no test elsewhere
          if (CAUSAL) z[r] = (key < L && (key <= qi || qi >= L)) ? z[r] : -INFINITY;
          else if (kt >= KTL - 2) z[r] = key < L ? z[r] : -INFINITY;
          m = fmaxf(m, z[r]);
        }
        s[kt] = z;
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    // exp(x / 8 - max / 8) = exp2(x C1 - max C1): one fma + v_exp_f32 per score; the row is normalised after P V (16 values per
    // lane instead of 4 KTL), P enters the MFMA as the fp16 value of the un-normalised exponential (<= 1)
    const float m2 = m * C1;
    float sum = 0.f;
    h4 pt[KTL];
#pragma unroll
    for (int kt = 0; kt < KTL; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], C1, -m2));
        sum += e;
        pt[kt][r] = (half_t)e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (g == 0 && qi < LQ) p.lse[((long)n * p.H + h) * L + qi] = (m2 + __log2f(sum)) * (1.0f / LOG2E);
    // O^T[d][q] = sum_key V[key][d] P[q][key]; k-step ks covers key tiles 2ks, 2ks+1 in permuted order
    f4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KTL / 2; ++ks) {
      const h8 pf = cat4(pt[2 * ks], pt[2 * ks + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag_swz(vtile, ks * 32, ks * 32 + 16, dt * 16, lane), pf, acc[dt], 0, 0, 0);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] *= inv;
    store_rows(o, D, acc, q0, LQ, scr, lane);
    qraw[0] = qnext[0]; qraw[1] = qnext[1];
  }
}

// Backward: persistent workgroups of NW = 8 waves, one per CU (the four operand images of a head fill the LDS), each walking
// (sequence, head) pairs.  Per head two phases, and every full-size operand image arrives by LDS-DMA (buffer_load ... lds: no
// registers, asynchronous) UNDER the phase before the one that needs it:
//   phase 1 (dQ), a wave per 16-query tile: K and V images resident (they landed during the previous head's phase 2); the tile's
//     own Q / dO / O rows come from global memory through the wave's staging tile, one tile ahead (delta[q] = <dO[q], O[q]> and the
//     log-sum-exp go to LDS for phase 2); S^T = K Q^T and dP^T = V dO^T by ds_read_b128 fragments, dS of the whole key range in
//     registers, dQ^T = K^T dS^T by transposed reads.  Meanwhile the Q and dO images of THIS head are in flight.
//   phase 2 (dK, dV), a wave per 16-key tile, two query tiles at a time: Q and dO images resident; the wave takes the K / V
//     fragments of its own key tiles into registers first, and after a barrier the K and V images are dead: the NEXT head's K
//     and V are requested into them and land while dV^T += dO^T P and dK^T += Q^T dS run.
// The images are unpadded [rows][64] with the chunk index XORed by swz_tr(row) on the SOURCE side of the DMA (the LDS side of a
// DMA is lane-linear); rows past L belong to the next sequence (or read as zero past the tensor: the descriptor ends there) and
// only ever meet probabilities that are exactly zero.  Round 2's kernel loaded all five operands synchronously in front of
// the phases: 35 % of its time (skip-phase builds: 476 us = 114 phase 1 + 195 phase 2 + 167 load / store / barriers).
// 8 halves op[row][8 chunk .. +7] of an image swizzled with swz_tr (conflict-free for ds_read_b128 as well: rows c and c + 8 of a
// 16-lane group never share a chunk because the two groups of g differ in bit 0 of the chunk)
__device__ __forceinline__ h8 frag_t(const half_t* tile, int row, int chunk) {
  return *reinterpret_cast<const h8*>(tile + row * DH + ((chunk ^ swz_tr(row)) << 3));
}

template <int ROWS, int NW>
__device__ __forceinline__ void dma_image(__amdgpu_buffer_rsrc_t rsrc, half_t* image, unsigned col_bytes, unsigned ld_bytes,
                                          int wid, int lane) {
  for (int rb = wid; rb < ROWS / 8; rb += NW) {                  // 1 KiB per wave-instruction: image rows 8 rb .. 8 rb + 7
    const int row = rb * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ swz_tr(row);
    const unsigned voff = (unsigned)row * ld_bytes + col_bytes + (unsigned)lc * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(reinterpret_cast<char*>(image) + rb * 1024), 16, voff, 0, 0, 0);
  }
}

// LEAD (the output gradient exists for query 0 of every sequence only; dout is read at that row alone): phase 1 runs for query
// tile 0 on wave 0 with the tile's other rows as zeros and every other query at lse = +inf (probability exactly 0); the Q and dO
// images of phase 2 are not fetched - their first 32 rows are written from that tile (row 0 and zeros) - and phase 2 walks the
// first pair of query tiles only; dK | dV of every token and dQ of token 0 are written, the Q columns of the other rows of dqkv
// are left untouched.
template <int KTL, int NW, bool CAUSAL, bool LEAD = false>
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void attn_long_bwd_kernel(AttnArgs p) {
  constexpr int ROWS = 16 * KTL;
  constexpr int QPAIRS = LEAD ? 1 : KTL / 2;     // pairs of query tiles phase 2 walks
  static_assert(NW * 2 >= KTL, "a wave holds the K / V fragments of at most two key tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = p.H * DH, L = p.L;
  const long ld = 3L * D;
  half_t* ktile = reinterpret_cast<half_t*>(smem);
  half_t* vtile = ktile + ROWS * DH;
  half_t* qtile = vtile + ROWS * DH;
  half_t* dtile = qtile + ROWS * DH;
  half_t* scr = dtile + ROWS * DH + wid * (16 * LDS_STRIDE);
  float* lse_s = reinterpret_cast<float*>(dtile + ROWS * DH + NW * 16 * LDS_STRIDE);
  float* del_s = lse_s + ROWS;
  float* red = del_s + ROWS;                     // [3][NW][64]: every wave's column sums of its dQ / dK / dV tiles
  float* rsc_s = red + 3 * NW * 64;              // [ROWS]: the row factors of the head's tokens (p.rowstat; 1 without)
  const bool want_dbias = p.dbias != nullptr;
  const int nt = (L + 15) / 16;                  // 16-row tiles that hold a real token
  const int ntq = LEAD ? 1 : nt;                 // ... query tiles with a gradient
  const int LQ = LEAD ? 1 : L;
  constexpr float C1 = 0.125f * LOG2E;
  const int total = p.nseq * p.H;
  const unsigned ldb = (unsigned)(ld * 2), ldo = (unsigned)(D * 2);
  // descriptors start at the sequence's first row and end with the tensor (at most 1 GiB ahead: a head touches ROWS rows)
  auto rsrc_qkv = [&](int n) {
    const long left = (long)(p.nseq - n) * L * ld * 2;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p.qkv + (long)n * L * ld), 0, (int)(left < (1l << 30) ? left : (1l << 30)), 0x00020000);
  };
  auto rsrc_dout = [&](int n) {
    const long left = (long)(p.nseq - n) * L * D * 2;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p.dout + (long)n * L * D), 0, (int)(left < (1l << 30) ? left : (1l << 30)), 0x00020000);
  };
  {                                              // the first head's K and V
    const int n = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const __amdgpu_buffer_rsrc_t r = rsrc_qkv(n);
    dma_image<ROWS, NW>(r, ktile, (unsigned)((D + h * DH) * 2), ldb, wid, lane0);
    dma_image<ROWS, NW>(r, vtile, (unsigned)((2 * D + h * DH) * 2), ldb, wid, lane0);
  }
  for (int head = blockIdx.x; head < total; head += gridDim.x) {
    // the lane index is made opaque per head: every LDS address below is a function of it, and with a loop-invariant lane
    // index the compiler hoists ~150 address registers out of this loop and spills them
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4, c = lane & 15;
    const int n = head / p.H, h = head % p.H;
    const half_t* q = p.qkv + (long)n * L * ld + h * DH;
    const half_t* dO = p.dout + (long)n * L * D + h * DH;
    const half_t* o = p.out + (long)n * L * D + h * DH;
    half_t* dq = p.dqkv + (long)n * L * ld + h * DH;
    half_t* dk = dq + D;
    half_t* dv = dq + 2 * D;
    const float* lse_g = p.lse + ((long)n * p.H + h) * L;
    // a query tile's Q / dO / O rows, lane-ordered (row (lane >> 3) + 8 t, bytes 16 (lane & 7) .. +15), and its log-sum-exp
    u4v rq[2], rd[2], ro[2];
    float rl;
    // INVARIANT (the s_waitcnt vmcnt(7) below counts on it): load_tile is exactly TILE_LOADS = 8 vector-memory loads - six 16-byte
    // loads (16-byte aligned: one instruction each, nothing to merge or split) and two dwords - every result is consumed, and the
    // "memory" clobber of the wait keeps all eight in front of it.  Changing the number of loads here means changing that count.
    constexpr int TILE_LOADS = 8;
    const float* const rs_g = p.rowstat ? p.rowstat + 2 * (long)n * L : lse_g;    // (lse_g: a valid address; the value is not used)
    float rsv;
    auto load_tile = [&](int qt) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // LEAD: row 0 is the only row of Q / dO / O that may be read (the same six loads: the counted wait below stands)
        const long row = LEAD ? 0 : min(qt * 16 + (lane >> 3) + 8 * t, L - 1);
        rq[t] = *reinterpret_cast<const u4v*>(q + row * ld + 8 * (lane & 7));
        rd[t] = *reinterpret_cast<const u4v*>(dO + row * D + 8 * (lane & 7));
        ro[t] = *reinterpret_cast<const u4v*>(o + row * D + 8 * (lane & 7));
      }
      rl = lse_g[LEAD ? 0 : min(qt * 16 + c, L - 1)];
      rsv = rs_g[p.rowstat ? 2 * min(qt * 16 + c, L - 1) : 0];
    };
    load_tile(min(wid, nt - 1));                 // requested BEFORE the DMAs below: the in-order vmcnt then does not make the
                                                 // first tile wait for 56 KiB of images
    static_assert(TILE_LOADS == 8, "the wait below leaves TILE_LOADS - 1 loads in flight");
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");   // all but 7 of those 8 loads: the K / V images of this head (older) have landed
    __syncthreads();
    if constexpr (!LEAD) {                       // this head's Q and dO images: needed in phase 2, in flight under phase 1
      dma_image<ROWS, NW>(rsrc_qkv(n), qtile, (unsigned)(h * DH * 2), ldb, wid, lane);
      dma_image<ROWS, NW>(rsrc_dout(n), dtile, (unsigned)(h * DH * 2), ldo, wid, lane);
    }
    // ---- phase 1: dQ
    f4 csum[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) csum[dt] = f4{0.f, 0.f, 0.f, 0.f};
#ifndef HMMC_ATTN_PIN
#define HMMC_ATTN_PIN 1
#endif
#ifndef HMMC_ATTN_SKIP
#define HMMC_ATTN_SKIP 0          // timing / register experiments (scratch/): 1 skips phase 1, 2 skips phase 2 (wrong results)
#endif
#if HMMC_ATTN_SKIP != 0 && !defined(HMMC_SCRATCH)
#error "HMMC_ATTN_SKIP builds compute wrong results: scratch experiments only (-DHMMC_SCRATCH)"
#endif
    for (int qt = wid; qt < ntq && HMMC_ATTN_SKIP != 1; qt += NW) {
      const int q0 = qt * 16, qi = q0 + c;
      const u4v z = {0u, 0u, 0u, 0u};
      if constexpr (LEAD) {                      // rows 1..15 of the tile as zeros (every lane loaded row 0)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bool keep = t == 0 && lane < 8;
          rq[t] = keep ? rq[t] : z; rd[t] = keep ? rd[t] : z; ro[t] = keep ? ro[t] : z;
        }
        // the first pair of query tiles of the Q and dO images, from this tile: row 0 and 31 rows of zeros (32 rows x 128 B = 4 KiB
        // per image; chunk index XORed with swz_tr(row) as the DMA would have placed it)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = (lane >> 3) + 8 * t, ch = (lane & 7) ^ swz_tr(row);
          *reinterpret_cast<u4v*>(qtile + row * DH + (ch << 3)) = t == 0 ? rq[0] : z;
          *reinterpret_cast<u4v*>(dtile + row * DH + (ch << 3)) = t == 0 ? rd[0] : z;
        }
      }
      // delta and lse of the tile -> LDS (phase 2 reads them for every query), this wave's own values straight from registers
      float dlv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const h8 hd = __builtin_bit_cast(h8, rd[t]), ho = __builtin_bit_cast(h8, ro[t]);
        float dl = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += (float)hd[j] * (float)ho[j];
        dl += __shfl_xor(dl, 1, 64);               // the 8 lanes of a row are consecutive
        dl += __shfl_xor(dl, 2, 64);
        dl += __shfl_xor(dl, 4, 64);
        const int row = q0 + (lane >> 3) + 8 * t;
        dlv[t] = row < L ? dl : 0.f;
        if ((lane & 7) == 0) del_s[row] = dlv[t];
      }
      const float lq = qi < LQ ? LOG2E * rl : INFINITY;         // +inf past LQ: those queries' probabilities vanish
      const float sq = p.rowstat ? rsv : 1.0f;                  // row factor of token qi (hmmc_attention_f16_bwd_scaled)
      if (g == 0) { lse_s[qi] = lq; rsc_s[qi] = sq; }
      // rows -> staging tile -> fragments (rows past L as zeros), Q then dO through the same tile (a wave's LDS operations are in order)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<u4v*>(scr + ((lane >> 3) + 8 * t) * LDS_STRIDE + 8 * (lane & 7)) = q0 + (lane >> 3) + 8 * t < L ? rq[t] : z;
      h8 qf[2], df[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const h8*>(scr + c * LDS_STRIDE + ks * 32 + 8 * g);
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<u4v*>(scr + ((lane >> 3) + 8 * t) * LDS_STRIDE + 8 * (lane & 7)) = q0 + (lane >> 3) + 8 * t < L ? rd[t] : z;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) df[ks] = *reinterpret_cast<const h8*>(scr + c * LDS_STRIDE + ks * 32 + 8 * g);
      const float dlq = del_s[qi];
      h4 ds16[KTL];
#pragma unroll
      for (int k0 = 0; k0 < KTL; k0 += 2) {          // two key tiles per round: their 8 fragments are requested together
        h8 kf[2][2], vf[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = (k0 + i) * 16 + c;
          kf[i][0] = frag_t(ktile, row, g); kf[i][1] = frag_t(ktile, row, 4 + g);
          vf[i][0] = frag_t(vtile, row, g); vf[i][1] = frag_t(vtile, row, 4 + g);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int kt = k0 + i;
          f4 zz = {0.f, 0.f, 0.f, 0.f};
          f4 sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i][0], qf[0], zz, 0, 0, 0);
          sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i][1], qf[1], sc, 0, 0, 0);
          f4 dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[i][0], df[0], zz, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[i][1], df[1], dp, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * g + r;
            float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], C1, -lq));
            if (CAUSAL) pv = (key < L && key <= qi) ? pv : 0.f;
            else if (kt >= KTL - 2) pv = key < L ? pv : 0.f;
            ds16[kt][r] = (half_t)(pv * (dp[r] - dlq));            // the 1/8 of dS is applied to the 16 results below
          }
        }
      }
      f4 acc[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        acc[dt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KTL / 2; ++ks)
          acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag_swz(ktile, ks * 32, ks * 32 + 16, dt * 16, lane),
                                                           cat4(ds16[2 * ks], ds16[2 * ks + 1]), acc[dt], 0, 0, 0);
        acc[dt] *= 0.125f;
      }
      if (qt + NW < ntq) load_tile(qt + NW);         // the wave's second tile: requested under the dQ stores of the first
      store_rows(dq, ld, acc, qt * 16, LQ, scr, lane, sq);
      if (want_dbias) add_rounded(csum, acc);        // queries past L are exact zeros (their dS is)
    }
    if (want_dbias) store_colsum(red + wid * 64, csum, lane);
    for (int r = ntq * 16 + tid; r < ROWS; r += 64 * NW) {    // whole tiles without a query: phase 2 walks every query of the image
      lse_s[r] = INFINITY;
      del_s[r] = 0.f;
    }
    if constexpr (LEAD) {                        // the row factors of the tokens phase 1 did not visit (phase 2 scales dK / dV by them)
      for (int r = 16 + tid; r < nt * 16; r += 64 * NW) rsc_s[r] = p.rowstat ? rs_g[2 * min(r, L - 1)] : 1.0f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's share of the Q / dO images has landed
    __syncthreads();                                          // ... everyone's, and every delta / lse is in LDS
    // ---- phase 2: dV and dK.  The K / V fragments of this wave's (at most two) key tiles first; then K and V are dead
    h8 kfr[2][2], vfr[2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int kt = min(wid + s2 * NW, ROWS / 16 - 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kfr[s2][ks] = frag_t(ktile, kt * 16 + c, ks * 4 + g);
        vfr[s2][ks] = frag_t(vtile, kt * 16 + c, ks * 4 + g);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (head + (int)gridDim.x < total) {           // the next head's K and V, under this phase
      const int hn = head + gridDim.x, n2 = hn / p.H, h2 = hn % p.H;
      const __amdgpu_buffer_rsrc_t r = rsrc_qkv(n2);
      dma_image<ROWS, NW>(r, ktile, (unsigned)((D + h2 * DH) * 2), ldb, wid, lane);
      dma_image<ROWS, NW>(r, vtile, (unsigned)((2 * D + h2 * DH) * 2), ldb, wid, lane);
    }
    f4 csk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { csum[dt] = f4{0.f, 0.f, 0.f, 0.f}; csk[dt] = f4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int kt = wid + s2 * NW;
      if (kt >= nt || HMMC_ATTN_SKIP == 2) break;
      const int key = kt * 16 + c;
      const bool key_ok = key < L;
      const h8 (&kf)[2] = kfr[s2];
      const h8 (&vf)[2] = vfr[s2];
      f4 av[4], ak[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { av[dt] = f4{0.f, 0.f, 0.f, 0.f}; ak[dt] = f4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int qp = 0; qp < QPAIRS; ++qp) {
        h4 p16[2], ds16[2];
        h8 qf[2][2], df[2][2];
        f4 lr[2], dl[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {                // both query tiles of the pair: fragments, lse and delta requested together
          const int qt = 2 * qp + e;
          qf[e][0] = frag_t(qtile, qt * 16 + c, g); qf[e][1] = frag_t(qtile, qt * 16 + c, 4 + g);
          df[e][0] = frag_t(dtile, qt * 16 + c, g); df[e][1] = frag_t(dtile, qt * 16 + c, 4 + g);
          lr[e] = *reinterpret_cast<const f4*>(lse_s + qt * 16 + 4 * g);
          dl[e] = *reinterpret_cast<const f4*>(del_s + qt * 16 + 4 * g);
        }
        h8 dT[4], qT[4];                             // dO^T / Q^T of the pair, used after the exponentials
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dT[dt] = tr_frag_swz(dtile, qp * 32, qp * 32 + 16, dt * 16, lane);
          qT[dt] = tr_frag_swz(qtile, qp * 32, qp * 32 + 16, dt * 16, lane);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int qt = 2 * qp + e;
          f4 z = {0.f, 0.f, 0.f, 0.f};
          f4 sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[e][0], kf[0], z, 0, 0, 0);
          sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[e][1], kf[1], sc, 0, 0, 0);
          f4 dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(df[e][0], vf[0], z, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(df[e][1], vf[1], dp, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = qt * 16 + 4 * g + r;
            float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], C1, -lr[e][r]));   // 0 for queries past L (lse = +inf)
            pv = (key_ok && (!CAUSAL || key <= qi)) ? pv : 0.f;
            p16[e][r] = (half_t)pv;
            ds16[e][r] = (half_t)(pv * (dp[r] - dl[e][r]));         // the 1/8 of dS is applied to dK below
          }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(dT[dt], cat4(p16[0], p16[1]), av[dt], 0, 0, 0);
          ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qT[dt], cat4(ds16[0], ds16[1]), ak[dt], 0, 0, 0);
        }
#if HMMC_ATTN_PIN
        __builtin_amdgcn_sched_barrier(0);           // keep the scheduler from hoisting the next pairs' 28 fragment reads over this one
#endif
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) ak[dt] *= 0.125f;
      const float sk = kt * 16 + c < nt * 16 ? rsc_s[kt * 16 + c] : 1.0f;      // phase 1 wrote the factors of every real tile
      store_rows(dv, ld, av, kt * 16, L, scr, lane, sk);
      store_rows(dk, ld, ak, kt * 16, L, scr, lane, sk);
      if (want_dbias) { add_rounded(csum, av); add_rounded(csk, ak); }   // keys past L are exact zeros (P and dS are)
    }
    // in_proj bias-gradient partials of this (sequence, head): the waves' sums in wave order (fixed: bit-stable)
    if (want_dbias) {
      store_colsum(red + (1 * NW + wid) * 64, csk, lane);
      store_colsum(red + (2 * NW + wid) * 64, csum, lane);
      __syncthreads();
      if (tid < 192) {
        const int type = tid >> 6, d = tid & 63;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[(type * NW + w) * 64 + d];
        p.dbias[(long)n * 3 * D + type * D + h * DH + d] = t;
      }
    }
    // the loop top waits for the K / V images of the next head and passes a barrier before anything of this head's LDS state
    // (Q / dO images, delta, lse, red) is written again
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// waves per workgroup: as many as share the head's query tiles evenly in two rounds (197 tokens = 13 tiles: 7 waves), at
// least 4; two workgroups per CU (unpadded K / V images), i.e. 3 - 4 waves per SIMD to overlap one wave's softmax with
// another's MFMAs
template <int KTL, int NW, bool CAUSAL, bool LEAD>
static void launch_long_fwd2(const AttnArgs& p, hipStream_t stream) {
  constexpr int LDS = fwd_lds_bytes(KTL, NW);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  if (LDS > 64 * 1024) {
    static bool done[HMMC_MAX_DEVICES] = {false};
    hmmc_allow_lds((const void*)attn_long_fwd_kernel<KTL, NW, CAUSAL, LEAD>, LDS, done);
  }
  hipLaunchKernelGGL((attn_long_fwd_kernel<KTL, NW, CAUSAL, LEAD>), dim3((unsigned)(p.nseq * p.H)), dim3(64 * NW), LDS, stream, p);
}
template <int KTL, int NW>
static void launch_long_fwd(const AttnArgs& p, hipStream_t stream, bool lead) {
  if (lead) { if (p.causal) launch_long_fwd2<KTL, NW, true, true>(p, stream); else launch_long_fwd2<KTL, NW, false, true>(p, stream); }
  else { if (p.causal) launch_long_fwd2<KTL, NW, true, false>(p, stream); else launch_long_fwd2<KTL, NW, false, false>(p, stream); }
}

int hmmc_attention_long_fwd(const AttnArgs& p, hipStream_t stream, bool lead) {
  if ((long)p.nseq * p.H >= (1l << 31)) return HMMC_ERR_UNSUPPORTED;
  const int ktl = ((p.L + 31) / 32) * 2;                                    // key tiles, rounded up to an even count
  switch (ktl) {
    case 6: launch_long_fwd<6, 4>(p, stream, lead); break;       // <= 96 tokens: 5-6 query tiles
    case 8: launch_long_fwd<8, 4>(p, stream, lead); break;
    case 10: launch_long_fwd<10, 5>(p, stream, lead); break;
    case 12: launch_long_fwd<12, 6>(p, stream, lead); break;
    case 14: launch_long_fwd<14, 7>(p, stream, lead); break;     // 197 tokens (ViT-B/16): 13 tiles on 7 waves
    case 16: launch_long_fwd<16, 8>(p, stream, lead); break;
    default: return HMMC_ERR_UNSUPPORTED;
  }
  return hmmc_launch_status();
}

template <int KTL, int NW, bool CAUSAL, bool LEAD>
static void launch_long_bwd2(const AttnArgs& p, hipStream_t stream) {
  constexpr int LDS = 4 * 16 * KTL * DH * 2 + NW * 16 * LDS_STRIDE * 2 + 2 * 16 * KTL * 4   // K, V, Q, dO images + staging per wave + lse, delta
                      + 3 * NW * 64 * 4 + 16 * KTL * 4;                                  // + the waves' column sums + the row factors
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static bool done[HMMC_MAX_DEVICES] = {false};
  hmmc_allow_lds((const void*)attn_long_bwd_kernel<KTL, NW, CAUSAL, LEAD>, LDS, done);
  const long total = (long)p.nseq * p.H;
  const long wgs = hmmc_num_cus();                                         // persistent: one workgroup per CU
  hipLaunchKernelGGL((attn_long_bwd_kernel<KTL, NW, CAUSAL, LEAD>), dim3((unsigned)(total < wgs ? total : wgs)), dim3(64 * NW), LDS, stream, p);
}
template <int KTL, int NW>
static void launch_long_bwd(const AttnArgs& p, hipStream_t stream, bool lead) {
  if (lead) { if (p.causal) launch_long_bwd2<KTL, NW, true, true>(p, stream); else launch_long_bwd2<KTL, NW, false, true>(p, stream); }
  else { if (p.causal) launch_long_bwd2<KTL, NW, true, false>(p, stream); else launch_long_bwd2<KTL, NW, false, false>(p, stream); }
}

int hmmc_attention_long_bwd(const AttnArgs& p, hipStream_t stream, bool lead) {
  if ((long)p.nseq * p.H >= (1l << 31) || !p.out) return HMMC_ERR_UNSUPPORTED;
  const int ktl = ((p.L + 31) / 32) * 2;
  switch (ktl) {
    // 8 waves (two per SIMD, up to 256 registers each: phase 2 holds 64 accumulator + 64 transposed-operand registers).  One wave
    // per tile - 13 waves for ViT-B/16's 13 tiles - would need 128 registers per wave and spills ~100 of them; 7 waves (13 tiles
    // in two even rounds) measured the same as 8 (480 vs 487 us at 384 x 197 x 12).
    case 6: launch_long_bwd<6, 8>(p, stream, lead); break;
    case 8: launch_long_bwd<8, 8>(p, stream, lead); break;
    case 10: launch_long_bwd<10, 8>(p, stream, lead); break;
    case 12: launch_long_bwd<12, 8>(p, stream, lead); break;
    case 14: launch_long_bwd<14, 8>(p, stream, lead); break;
    case 16: launch_long_bwd<16, 8>(p, stream, lead); break;
    default: return HMMC_ERR_UNSUPPORTED;
  }
  return hmmc_launch_status();
}
