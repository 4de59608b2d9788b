// fp16 MFMA GEMM for the CLIP towers (K3, K5, K6, K1, K7 of SURVEY.md section 2.3 and their
// backward passes).  Replaces the F.linear / nn.MultiheadAttention projections at
// reference modules/module_clip.py:235-257 and the patch conv at :278,307.
//
//   C[m][n] = epilogue( sum_k Aop[m][k] * Bop[n][k] ),  fp16 in, fp32 accumulate, fp16 out
//
// Operand layouts (no transposed copies are ever made in HBM):
//   k-major  : op[r][k] = P[r*ld + k]   (activations X[M,K], weights W[N,K])
//   m-major  : op[r][k] = P[k*ld + r]   (dY / X seen by wgrad, W seen by dgrad)
// forward y = x W^T        : A k-major, B k-major
// dgrad   dx = dy W        : A k-major (dy), B m-major (W[N',K'] indexed [k=n'][r=k'])
// wgrad   dW = dy^T x      : A m-major (dy), B m-major (x), split-K over tokens
//
// Structure: two tile configurations of one template — 256x256x64 (8 waves as 2x4, 128x64 per wave,
// one workgroup per CU, 128 KiB of LDS) for the large tower GEMMs, where it halves the L2->LDS
// bytes per flop, and 128x128x64 (4 waves as 2x2, 64x64 each, two workgroups per CU) for small or
// ragged problems.  16x16x32 f16 MFMA with the operands swapped (MFMA-A = weight rows, MFMA-B =
// activation rows) so each lane owns 4 consecutive n of one output row (8-byte stores).  Tiles are staged by LDS-DMA
// (buffer_load ... lds, 16 B/lane) through a bounds-checked buffer descriptor: rows past the
// end of a matrix read as zero, so ragged M/N/K need no branches and cannot fault.
// LDS images are lane-linear; bank conflicts are removed by XOR-swizzling the per-lane SOURCE
// chunk and applying the same XOR on the read (k-major: ds_read_b128; m-major:
// ds_read_b64_tr_b16 hardware transpose).  Two LDS stages; the prefetch of tile t+1 stays in
// flight across the compute of tile t (counted vmcnt + raw s_barrier).
#include "common.h"
#include "options.h"
#include <cstdlib>

#ifndef HMMC_DBG
#define HMMC_DBG 0   // diagnostics (scratch/): 1 no global stores, 2 no epilogue (wrong results; timing only);
                     // 3 / 4 / 5: s_memrealtime stamps of the third work item into p.ws (scratch/gemm_stamps.py), 4 = 1 + stamps, 5 = 2 + stamps
#endif

#ifndef HMMC_PHASES
#define HMMC_PHASES 2  // segments of the 256x256 K-loop per K-tile and group: 2 x 32 MFMAs (product); 4 x 16 MFMAs = rounds 1-2 (scratch)
#endif
#ifndef HMMC_PF
#define HMMC_PF 6     // prefetch distance of the 256x256 K-loop in half-tiles (scratch experiments build 4)
#endif
#if (HMMC_DBG != 0 || HMMC_PHASES != 2 || HMMC_PF != 6) && !defined(HMMC_SCRATCH)
#error "HMMC_DBG / HMMC_PHASES / HMMC_PF other than the product values (0 / 2 / 6) are scratch experiments: build with -DHMMC_SCRATCH"
#endif

namespace {

constexpr int BKT = 64;

enum { EPI_BIAS = 1, EPI_RESID = 2, EPI_QGELU = 4, EPI_DGELU = 8, EPI_COLSUM = 32, EPI_SAVE_DGELU = 64, EPI_MULAUX = 128,
       EPI_LNFOLD = 256,      // acc -> a_r * acc + (b_r * c_n + d_n): LayerNorm folded into the GEMM (rowstat, colterms), see below
       EPI_ROWSTAT = 512,     // + per-row (sum, sum of squares) of the fp16 values written, one pair per 64-column block (stat_part)
       EPI_ROWSCALE = 1024 }; // out *= rowstat[m][0] (the row's rstd) as the last step: the backward of a folded LayerNorm (ln_fold.hip)

struct GemmArgs {
  const half_t* A; const half_t* B; half_t* C;
  const half_t* bias; const half_t* resid; half_t* aux_out; const half_t* aux_in;
  float* ws;
  float* csum;                  // EPI_COLSUM: fp32 [row blocks of 128 (256x256 tile) or 64 rows][N] partial column sums of C
  const float* rowstat;         // EPI_LNFOLD: [M][2] = (rstd_r, -rstd_r * mean_r) of the rows of A
  const float* colterms;        // EPI_LNFOLD: [2][N] = c_n = sum_k B[n][k] (B = gamma o W), d_n = sum_k beta_k W[n][k] + bias_n
  float* stat_part;             // EPI_ROWSTAT: [N / 64][M][2] = (sum, sum of squares) over the 64-column block of the row
  int M, N, K, lda, ldb, ldc;
  int flags, splitk, ktps;
  int slab;                     // 1: write the fp32 partial slab even when splitk == 1 (pieces along K, reduced by the host's launch)
  unsigned a_bytes, b_bytes;
};

// Grouped weight gradients: up to four problems dW_j[M_j, N_j] = dY_j^T X_j over the SAME token range (K = tokens) in one
// persistent launch.  An item is (K split, problem, tile); all problems share nkt and the split, so the four weight
// gradients of a layer fill the chip together (108 tiles at ViT-B/32: split 7 = 756 items = 2.95 rounds) instead of one
// after the other with a partial last round and a split chosen for 9-36 tiles each.  Slabs: [problem][split][M_j][N_j] fp32.
constexpr int GROUP_MAX = 4;
struct GroupProb { const half_t* A; const half_t* B; int lda, ldb, M, N; unsigned a_bytes, b_bytes; int ntn; int pad_; long slab_off; };
struct GemmGroup { int n; int tile0[GROUP_MAX + 1]; GroupProb pr[GROUP_MAX]; };

// ---- LDS-DMA staging -------------------------------------------------------------------------
// k-major tile image: [128 rows][8 chunks of 16 B]; phys chunk = logical ^ ((row >> 1) & 7)
// m-major tile image: [64 k-rows][16 chunks of 16 B]; phys chunk = logical ^ f(krow),
//                     f(krow) = ((krow & 3) << 2) | ((krow >> 2) & 3)
// R = tile rows (128 or 256), NTH = threads per workgroup; R*8 chunks of 16 B, one per thread per pass.
template <bool KMAJ, int R, int NTH>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int wid, int tid,
                                           int r0, int k0, int ld) {
  constexpr int PASSES = R * 8 / NTH;
  constexpr int CPR = R / 8;                    // chunks per k-row of an m-major image
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    unsigned goff;
    if (KMAJ) {
      int row = ps * (NTH / 8) + (tid >> 3);
      int logical = (tid & 7) ^ ((row >> 1) & 7);
      goff = ((unsigned)(r0 + row) * (unsigned)ld + (unsigned)(k0 + logical * 8)) * 2u;
    } else {
      int krow = ps * (NTH / CPR) + tid / CPR;
      int pc = tid % CPR;
      int f = ((krow & 3) << 2) | ((krow >> 2) & 3);
      int logical = (pc & ~15) | ((pc & 15) ^ f);
      goff = ((unsigned)(k0 + krow) * (unsigned)ld + (unsigned)(r0 + logical * 8)) * 2u;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + ps * (NTH * 16) + wid * 1024), 16, goff, 0, 0, 0);
  }
}

// ---- half-tile staging for the ping-pong schedule (256x256 tile) ------------------------------------
// A 256-row operand tile is held as two 16 KiB half images of 128 rows.  Half h holds, for every wave, the
// rows that wave needs in its half-h quadrants, so that a half image is dead (and can be restaged) as soon
// as one phase of the K-tile has read it:
//   A: LDS row r of half h  <->  tile row (r >> 6) * 128 + h * 64 + (r & 63)    (wave wm reads r = wm*64 ..+63)
//   B: LDS row r of half h  <->  tile row (r >> 5) *  64 + h * 32 + (r & 31)    (wave wn reads r = wn*32 ..+31)
// so each wave still owns 128 x 64 CONTIGUOUS outputs.  k-major half image: [128 rows][8 chunks];
// m-major: [64 k-rows][16 chunks]; same XOR swizzles as above.  The per-thread part of the source offset is
// loop-invariant (half_vbase), the per-half part is wave-uniform.
template <bool KMAJ, bool IS_A>
__device__ __forceinline__ unsigned half_vbase(int tid, int ld) {
  if (KMAJ) {
    int r = tid >> 3;                                        // 0..63: LDS row within a pass
    int logical = (tid & 7) ^ ((r >> 1) & 7);
    int row = IS_A ? r : ((r >> 5) * 64 + (r & 31));
    return ((unsigned)row * (unsigned)ld + (unsigned)(logical * 8)) * 2u;
  } else {
    int krow = tid >> 4, pc = tid & 15;                      // 32 k-rows per pass
    int f = ((krow & 3) << 2) | ((krow >> 2) & 3);
    int logical = pc ^ f;
    int row = IS_A ? ((logical >> 3) * 128 + (logical & 7) * 8) : ((logical >> 2) * 64 + (logical & 3) * 8);
    return ((unsigned)krow * (unsigned)ld + (unsigned)row) * 2u;
  }
}

// two LDS-DMA loads per thread (512 threads x 16 B x 2 = 16 KiB); soff = 0x80000000 makes both read as zero
template <bool KMAJ>
__device__ __forceinline__ void stage_half(__amdgpu_buffer_rsrc_t rsrc, char* lds_half, int wid, unsigned vbase,
                                           unsigned soff, int ld) {
  const unsigned pass = (KMAJ ? 128u : 32u) * (unsigned)ld * 2u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_half + wid * 1024), 16, vbase + soff, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_half + 8192 + wid * 1024), 16, vbase + soff + pass, 0, 0, 0);
}

// ---- fragment reads --------------------------------------------------------------------------
// returns the 8 halves op[row0 + (lane & 15)][ks*32 + 8*(lane >> 4) + j], j = 0..7
template <bool KMAJ, int R>
__device__ __forceinline__ h8 read_frag(const char* lds_tile, int row0, int ks, int lane) {
  if (KMAJ) {
    int row = row0 + (lane & 15);
    int c = ks * 4 + (lane >> 4);
    int phys = c ^ ((row >> 1) & 7);
    return *reinterpret_cast<const h8*>(lds_tile + row * 128 + phys * 16);
  } else {
    int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    int krow = ks * 32 + 8 * g + q;
    int c = (row0 >> 3) + (pp >> 1);
    int f0 = (q << 2) | ((2 * g) & 3);
    int f1 = (q << 2) | ((2 * g + 1) & 3);
    const char* a0 = lds_tile + krow * (R * 2) + (((c & ~15) | ((c & 15) ^ f0)) << 4) + 8 * (pp & 1);
    const char* a1 = lds_tile + (krow + 4) * (R * 2) + (((c & ~15) | ((c & 15) ^ f1)) << 4) + 8 * (pp & 1);
    fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)LDS_PTR(a0));
    fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)LDS_PTR(a1));
    h8 r;
    r[0] = (half_t)lo[0]; r[1] = (half_t)lo[1]; r[2] = (half_t)lo[2]; r[3] = (half_t)lo[3];
    r[4] = (half_t)hi[0]; r[5] = (half_t)hi[1]; r[6] = (half_t)hi[2]; r[7] = (half_t)hi[3];
    return r;
  }
}

// ---- epilogue -------------------------------------------------------------------------------------
// MFMA layout: lane (c = lane & 15, g = lane >> 4) owns C[m0 + c][16j + 4g .. +3] of every 16x16 tile j: four
// 8-byte pieces per row, and 16 consecutive lanes hold 16 different rows.  The memory pipeline coalesces a
// wave-instruction over CONSECUTIVE lanes: the same 8 rows x 128 B written with lanes in that order take 3.4x as long as
// with lane l on row l >> 3, bytes 16 (l & 7) .. +15 (scratch/ubench/burst_store.hip: 3.7 us against 1.1 us per 256x256 tile
// per CU; in the kernel the stores were 4.6 us of every work item, whatever the cache policy and whether the lines were
// in L2 or not).  So every 16-row strip goes through a 2 KiB per-wave LDS tile: written from the MFMA layout with
// ds_write_b64, read back one full row per 8 lanes with ds_read_b128, and stored / loaded in that order.  A wave's LDS
// operations execute in order, so neither a barrier nor a wait separates a strip's writes from its reads or one strip
// from the next; rows are 128 B with the 16-byte chunk index XORed by key(row) = (row & 7) ^ (row >> 3), which makes
// both the 8-byte writes and the 16-byte reads bank-conflict free.  The residual / auxiliary operand takes the same
// path backwards (full-line loads in lane order -> LDS -> MFMA layout) through a second tile.
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

typedef float f2v __attribute__((ext_vector_type(2)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));

constexpr int EPI_LDS_PER_WAVE = 4096;            // two 16 x 128 B tiles: results, operands

// two fp32 values -> one dword of two fp16 (round to nearest even): v_cvt_pk_f16_f32
__device__ __forceinline__ unsigned pk2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a, b}, h2v));
}
__device__ __forceinline__ f4 unpack2(unsigned lo, unsigned hi) {
  h2v a = __builtin_bit_cast(h2v, lo), b = __builtin_bit_cast(h2v, hi);
  return f4{(float)a[0], (float)a[1], (float)b[0], (float)b[1]};
}

// F >= 0: the epilogue flags at compile time; F < 0: p.flags at run time.  FULL: interior tile, no masks anywhere.
// scr: this wave's EPI_LDS_PER_WAVE bytes of LDS.
template <int MT, int F, bool FULL>
__device__ __forceinline__ void epilogue_run(const GemmArgs& p, f4 (&acc)[MT][4], int m_base, int n0, int lane, char* scr) {
  const int flags = F >= 0 ? (F & ~EPI_COLSUM) : (p.flags & ~EPI_COLSUM);
  const bool want_csum = F >= 0 ? (F & EPI_COLSUM) != 0 : p.csum != nullptr;
  const int c = lane & 15, g = lane >> 4;
  // MFMA side of the LDS tiles: piece j of this lane at mf_off ^ (32 j).  Two layouts, because the two directions use
  // different instructions: the operand tile is READ with ds_read_b64 (32-lane groups, 64 banks: rows c and c + 8 must
  // differ in their chunk -> key(c) = (c & 7) ^ (c >> 3)); the result tile is WRITTEN with ds_write_b64 (16-lane groups, 32
  // banks: the 16 rows of a group must cover both 8-byte halves of 8 chunks -> chunk ^ (c & 7), half ^ (c >> 3); measured
  // with the operand layout: SQ_LDS_BANK_CONFLICT 8.7 % of the LDS cycles of the launch).  The line side swaps the two
  // halves of the pieces of rows 8..15 to match.
  const int mf_off = c * 128 + (((g >> 1) ^ ((c & 7) ^ (c >> 3))) << 4) + 8 * (g & 1);
  const int mf_out = c * 128 + (((g >> 1) ^ (c & 7)) << 4) + 8 * ((g & 1) ^ (c >> 3));
  // line side: lane l reads / writes row (l >> 3) + 8 t, chunk l & 7
  const int r_l = lane >> 3, q_l = lane & 7;
  const int ln_off0 = r_l * 128 + ((q_l ^ r_l) << 4);                  // both layouts: key = r_l for rows 0..7
  const int ln_off1 = (r_l + 8) * 128 + ((q_l ^ r_l ^ 1) << 4);        // operand tile: key(r_l + 8) = r_l ^ 1
  const int ln_out1 = (r_l + 8) * 128 + ((q_l ^ r_l) << 4);            // result tile: same chunk, halves swapped
  const bool n_ok = FULL || n0 + 8 * q_l < p.N;                        // N % 8 == 0: a 16-byte piece is all-in or all-out
  // global address = uniform 64-bit base (scalar registers) + 32-bit lane offset
  const unsigned voff = ((unsigned)r_l * (unsigned)p.ldc + (unsigned)(8 * q_l)) * 2u;
  const size_t row8 = (size_t)p.ldc * 16u;                             // bytes between row r and row r + 8
  char* const scr_in = scr + 2048;
  f4 bias[4];
  if (flags & EPI_BIAS) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n = n0 + 16 * j + 4 * g;
      h4 bv = *reinterpret_cast<const h4*>(p.bias + ((FULL || n < p.N) ? n : 0));
      bias[j] = f4{(float)bv[0], (float)bv[1], (float)bv[2], (float)bv[3]};
    }
  }
  // EPI_LNFOLD: y = LN(x) W^T + b evaluated as rstd_r (x (gamma o W)^T)[r][n] - rstd_r mean_r c_n + d_n on the RAW rows x: the
  // normalised activations are never written or read (reference modules/module_clip.py:217-223,252-256: ln_1 -> in_proj,
  // ln_2 -> c_fc).  The row pair (a_r, b_r) = (rstd_r, -rstd_r mean_r) of each of this lane's MT rows and the column pairs
  // (c_n, d_n) of its 16 columns are requested up front.
  // Registers: the MT row pairs would be 2 MT live registers through the whole strip loop (with the second QuickGELU output
  // that took the kernel over its 256: scratch spills sit in the K loop's in-order vmcnt queue).  The LNFOLD kernels have no
  // residual / auxiliary operand, so their operand tile in LDS is free: the pairs are parked there (16 rows per strip, lanes
  // g = 0 write, every lane reads its row back per strip; a wave's LDS operations execute in order).  ROWSCALE (which does
  // use the operand tile) needs the rstd alone: MT registers.
  f4 fold_c[4], fold_d[4];
  float fold_a[MT];
  // 256x256 kernels built for LNFOLD: the K loop brought the wave's 128 row pairs (1 KiB) and 64 column pairs (c | d, 512 B) into
  // that tile by LDS-DMA during the item's last K-tile (gemm_f16_body): no global-memory latency in front of strip 0
  constexpr bool FOLD_PRE = MT == 8 && F >= 0 && (F & EPI_LNFOLD) != 0;
  if (FOLD_PRE) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fold_c[j] = *reinterpret_cast<const f4*>(scr + 2048 + 1024 + (16 * j + 4 * g) * 4);
      fold_d[j] = *reinterpret_cast<const f4*>(scr + 2048 + 1280 + (16 * j + 4 * g) * 4);
    }
  } else if (flags & EPI_LNFOLD) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 16 * j + 4 * g;
      const int nn = (FULL || n < p.N) ? n : 0;
      fold_c[j] = *reinterpret_cast<const f4*>(p.colterms + nn);
      fold_d[j] = *reinterpret_cast<const f4*>(p.colterms + p.N + nn);
    }
    f2 t[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = FULL ? m_base + 16 * i + c : min(m_base + 16 * i + c, p.M - 1);
      t[i] = *reinterpret_cast<const f2*>(p.rowstat + 2 * (size_t)m);
    }
    if (g == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i) *reinterpret_cast<f2*>(scr + 2048 + (16 * i + c) * 8) = t[i];
    }
  }
  // ROWSCALE: the rstd of strip i's row is requested RS_AHEAD strips early (a register each; all MT up front took this variant,
  // which already holds four operand strips in flight, over the register file)
  constexpr int RS_AHEAD = 2;
  // bounds-checked buffer load (rows past M read 0 and are never stored): ONE lane offset for all strips, the strip is an
  // immediate - per-strip 64-bit addresses cost two registers each
  const __amdgpu_buffer_rsrc_t rs_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.rowstat, 0, (flags & EPI_ROWSCALE) ? p.M * 8 : 0, 0x00020000);
  const int rs_off = (m_base + c) * 8;
  auto load_scale = [&](int i) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_rsrc, rs_off, i * 128, 0));
  };
  if (flags & EPI_ROWSCALE) {
#pragma unroll
    for (int i = 0; i < RS_AHEAD && i < MT; ++i) fold_a[i] = load_scale(i);
  }
  const bool has_src = flags & (EPI_DGELU | EPI_MULAUX | EPI_RESID);
  const char* src = reinterpret_cast<const char*>((flags & (EPI_DGELU | EPI_MULAUX)) ? p.aux_in : p.resid);
  constexpr int HB = MT > 4 ? 4 : MT;            // strips whose operand loads are in flight together (32 VGPRs)
  u4 rin[HB][2];
  const bool two = (flags & EPI_QGELU) && p.aux_out;
  float st1[MT], st2[MT];                        // EPI_ROWSTAT: per strip, this lane's part of its row's sum and sum of squares
  f4 csum[4];                                    // EPI_COLSUM: column sums of the fp16 values written, over this wave's rows
#pragma unroll
  for (int j = 0; j < 4; ++j) csum[j] = f4{0.f, 0.f, 0.f, 0.f};
  auto load_strip = [&](int slot, int strip) {
    const char* sb = src + ((size_t)(m_base + 16 * strip) * p.ldc + n0) * 2u;
    if (FULL) {
      rin[slot][0] = *reinterpret_cast<const u4*>(sb + voff);
      rin[slot][1] = *reinterpret_cast<const u4*>(sb + row8 + voff);
    } else {                                     // clamped, never masked: rows / columns outside are not stored
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int m = min(m_base + 16 * strip + 8 * t + r_l, p.M - 1);
        rin[slot][t] = *reinterpret_cast<const u4*>(src + ((size_t)m * p.ldc + (n_ok ? n0 + 8 * q_l : 0)) * 2u);
      }
    }
  };
  // the four 8-byte pieces of a strip (MFMA layout) -> LDS -> the lane's two 16-byte row pieces -> global.  The global
  // stores of a strip are issued after the LDS operations of the NEXT one, so a strip's LDS round trip is covered by its
  // successor's arithmetic instead of being waited for.
  u4 pv0, pv1;
  half_t* pv_dst = nullptr;
  int pv_i = 0;
  bool have_prev = false;
  auto flush = [&]() {
    if (!have_prev) return;
#if HMMC_DBG == 1 || HMMC_DBG == 4
    asm volatile("" :: "v"(pv0), "v"(pv1));
    return;
#endif
    char* sb = reinterpret_cast<char*>(pv_dst) + ((size_t)(m_base + 16 * pv_i) * p.ldc + n0) * 2u;
    if (FULL) {
      // streaming stores: a tile's output is next read by another kernel, long after the operand panels that the K loops
      // keep re-reading from L2 have passed
      __builtin_nontemporal_store(pv0, reinterpret_cast<u4*>(sb + voff));
      __builtin_nontemporal_store(pv1, reinterpret_cast<u4*>(sb + row8 + voff));
    } else {
      const int m0 = m_base + 16 * pv_i + r_l;
      if (m0 < p.M && n_ok) *reinterpret_cast<u4*>(sb + voff) = pv0;
      if (m0 + 8 < p.M && n_ok) *reinterpret_cast<u4*>(sb + row8 + voff) = pv1;
    }
  };
  auto store_strip = [&](half_t* dst, int i, const u2 (&pc)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<u2*>(scr + (mf_out ^ (32 * j))) = pc[j];
    const u4 o0 = *reinterpret_cast<const u4*>(scr + ln_off0);
    const u4 t1 = *reinterpret_cast<const u4*>(scr + ln_out1);
    const u4 o1 = u4{t1[2], t1[3], t1[0], t1[1]};
    flush();
    pv0 = o0; pv1 = o1; pv_dst = dst; pv_i = i; have_prev = true;
  };
  // operand strip: full-line pieces (lane order) -> LDS -> the four 8-byte pieces of the MFMA layout, one strip ahead
  u2 inr[2][4];
  auto in_issue = [&](int i) {
    *reinterpret_cast<u4*>(scr_in + ln_off0) = rin[i % HB][0];
    *reinterpret_cast<u4*>(scr_in + ln_off1) = rin[i % HB][1];
    // the slot is free again: request strip i + HB now, so the second half of the operand arrives while the first is processed
    if (i + HB < MT) load_strip(i % HB, i + HB);
#pragma unroll
    for (int j = 0; j < 4; ++j) inr[i & 1][j] = *reinterpret_cast<const u2*>(scr_in + (mf_off ^ (32 * j)));
  };
  if (has_src) {
#pragma unroll
    for (int ii = 0; ii < HB; ++ii) load_strip(ii, ii);
    in_issue(0);
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    f4 in[4];
    if (has_src) {
      if (i + 1 < MT) in_issue(i + 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) in[j] = unpack2(inr[i & 1][j][0], inr[i & 1][j][1]);
    }
    // pc / qc: the strip's result (and auxiliary result) as packed fp16, four 8-byte pieces in the MFMA layout
    u2 pc[4], qc[4];
    float rs1 = 0.f, rs2 = 0.f;                    // EPI_ROWSTAT: this lane's part of the row's sum and sum of squares
    if ((flags & EPI_ROWSCALE) && i + RS_AHEAD < MT) fold_a[i + RS_AHEAD] = load_scale(i + RS_AHEAD);
    f2 fr = f2{1.f, 0.f};                          // EPI_LNFOLD: the row pair of this lane's row of strip i
    if (flags & EPI_LNFOLD) fr = *reinterpret_cast<const f2*>(scr + 2048 + (16 * i + c) * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f4 v = acc[i][j];
      if (flags & EPI_LNFOLD) v = fr[0] * v + (fr[1] * fold_c[j] + fold_d[j]);
      if (flags & EPI_BIAS) v += bias[j];
      if (flags & EPI_QGELU) {
        // QuickGELU with the reference's fp16 rounding points, two elements at a time.  The values that torch holds as
        // fp16 tensors (h, t = 1.702 h, sigmoid(t), the product) stay PACKED: v_fma_mix_f32 reads either half as an fp32
        // operand without a conversion, and the final product of two fp16 values is one v_pk_mul_f16 (exact product, one
        // rounding: what the fp32 multiply + pack did).  15 VALU + 4 transcendental instructions per pair, against 33 + 4
        // when every intermediate went through v_cvt_f32_f16 / v_cvt_f16_f32 (the epilogue is VALU-bound: c_fc forward
        // ran 905 us against 610 us without an epilogue).
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const unsigned hpk = pk2(v[2 * e], v[2 * e + 1]);                 // h = fp16(acc + bias)
          float p0, p1, a0, a1;
          asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(p0) : "v"(hpk), "v"(1.702f));
          asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(p1) : "v"(hpk), "v"(1.702f));
          const unsigned tpk = pk2(p0, p1);                                 // fp32 product, then fp16: torch's two roundings
          asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(a0) : "v"(tpk), "v"(-1.4426950408889634f));
          asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(a1) : "v"(tpk), "v"(-1.4426950408889634f));
          const float s0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a0));
          const float s1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a1));
          const unsigned spk = pk2(s0, s1);
          unsigned opk;
          asm("v_pk_mul_f16 %0, %1, %2" : "=v"(opk) : "v"(hpk), "v"(spk));
          pc[j][e] = opk;
          // aux: the pre-activation h, or (EPI_SAVE_DGELU) QuickGELU'(h) = s + 1.702 h s (1 - s) so that the backward
          // GEMM multiplies by it (EPI_MULAUX) instead of evaluating exp and rcp again
          qc[j][e] = (flags & EPI_SAVE_DGELU) ? pk2(__builtin_fmaf(p0 * (1.0f - s0), s0, s0), __builtin_fmaf(p1 * (1.0f - s1), s1, s1)) : hpk;
        }
        continue;
      }
      f4 out;
      if (flags & EPI_DGELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[r] = v[r] * qgelu_grad(in[j][r]);
      } else if (flags & EPI_MULAUX) {
        out = v * in[j];
      } else if (flags & EPI_RESID) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[r] = in[j][r] + r16(v[r]);
      } else {
        out = v;
      }
      if (flags & EPI_ROWSCALE) {
        // the tensor written is rstd_r x the gradient, which is what the folded weight gradient and LayerNorm backward consume
        out = out * fold_a[i];
      }
      const unsigned lo = pk2(out[0], out[1]), hi = pk2(out[2], out[3]);
      pc[j] = u2{lo, hi};
      if (flags & EPI_ROWSTAT) {
        // (sum, sum of squares) of the ROUNDED values (v_dot2c_f32_f16: two elements per instruction, fp32 accumulation).
        // From the scalars, not from pc[j][e]: hipcc 7.2 folds bit_cast<half2>(pc[j][1]) to the FIRST dword of the pair.
        const h2v ones = h2v{(_Float16)1.0f, (_Float16)1.0f};
        const h2v hl = __builtin_bit_cast(h2v, lo), hh = __builtin_bit_cast(h2v, hi);
        rs1 = __builtin_amdgcn_fdot2(hl, ones, rs1, false);
        rs2 = __builtin_amdgcn_fdot2(hl, hl, rs2, false);
        rs1 = __builtin_amdgcn_fdot2(hh, ones, rs1, false);
        rs2 = __builtin_amdgcn_fdot2(hh, hh, rs2, false);
      }
    }
    if (want_csum) {                               // sums of the ROUNDED values, as a pass over the stored tensor would see
      const bool row_ok = FULL || m_base + 16 * i + c < p.M;
      if (row_ok) {
        // ROWSCALE: the bias gradient is the column sum of the UNSCALED gradient: the stored values divided by the row's factor
        // again (a second set of packed unscaled values took this variant 30 registers over the file: scratch in the K loop)
        const float inv = (flags & EPI_ROWSCALE) ? __builtin_amdgcn_rcpf(fold_a[i]) : 1.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (flags & EPI_ROWSCALE) csum[j] += unpack2(pc[j][0], pc[j][1]) * inv;
          else csum[j] += unpack2(pc[j][0], pc[j][1]);
        }
      }
    }
    if (flags & EPI_ROWSTAT) { st1[i] = rs1; st2[i] = rs2; }
    store_strip(p.C, i, pc);
    if (two) store_strip(p.aux_out, i, qc);      // the pre-activation or QuickGELU'(h), for the backward pass
  }
  flush();
  if (flags & EPI_ROWSTAT) {
    // this lane's 16 columns of each of its MT rows -> the row's 64-column block: the other columns sit in lanes c + 16, c + 32,
    // c + 48.  Reduced here, after the strips: inside the strip loop the shuffles (ds_bpermute) would sit in the in-order LDS
    // queue between the tile writes and reads the loop keeps in flight, and their results would be waited for per strip.
    auto xsum = [](float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); };
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float s1 = xsum(st1[i]), s2 = xsum(st2[i]);
      const int m = m_base + 16 * i + c;
      if (g == 0 && (FULL || (m < p.M && n0 < p.N))) *reinterpret_cast<f2*>(p.stat_part + 2 * ((size_t)(n0 >> 6) * p.M + m)) = f2{s1, s2};
    }
  }
  if (want_csum) {                               // 16 lanes c -> one partial row per wave: row block (m_base / (16 MT))
    float* dst = p.csum + (size_t)(m_base / (16 * MT)) * p.N + n0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f4 t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = csum[j][r];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
        t[r] = v;
      }
      const int n = n0 + 16 * j + 4 * g;
      if (c == 0 && (FULL || n < p.N)) *reinterpret_cast<f4*>(dst + 16 * j + 4 * g) = t;
    }
  }
}

template <int MT, int F>
__device__ __forceinline__ void epilogue_impl(const GemmArgs& p, f4 (&acc)[MT][4], int m_base, int n0, int lane, char* scr) {
  if (m_base + 16 * MT <= p.M && n0 + 64 <= p.N) epilogue_run<MT, F, true>(p, acc, m_base, n0, lane, scr);    // wave-uniform
  else epilogue_run<MT, F, false>(p, acc, m_base, n0, lane, scr);
}

// EPI >= 0: the kernel was instantiated for exactly these flags (straight-line epilogue, its own register budget);
// EPI < 0: generic kernel, flags read at run time
template <int MT, int EPI>
__device__ __forceinline__ void epilogue_f16(const GemmArgs& p, f4 (&acc)[MT][4], int m_base, int n0, int lane, char* scr) {
  epilogue_impl<MT, EPI>(p, acc, m_base, n0, lane, scr);
}

// split-K partial sums: fp32 slab [split][M][N], 16-byte stores straight from the MFMA layout
template <int MT, int NT>
__device__ __forceinline__ void epilogue_slab(float* slab0, int M, int N, f4 (&acc)[MT][NT], int mrow, int ncol, int split) {
  float* ws = slab0 + (size_t)split * M * N;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int m = mrow + i * 16;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      int n = ncol + j * 16;
      if (n < N) *reinterpret_cast<f4*>(ws + (size_t)m * N + n) = acc[i][j];
    }
  }
}

// BM x BN block tile, WM x WN waves, each wave (BM/WM) x (BN/WN) = MT x NT MFMA tiles of 16x16.
// Persistent: the grid is sized to the chip and every workgroup walks work items (output tile x K-split)
// item, item + gridDim, ...  The LDS-DMA prefetch runs one K-tile ahead across item boundaries, so the
// first tile of the next output tile is already in flight while this one's epilogue stores drain.
template <bool AK, bool BK, int BM, int BN, int WM, int WN, int EPI, bool GRP>
__device__ __forceinline__ void gemm_f16_body(const GemmArgs& p, const GemmGroup& gp) {
  constexpr int NTH = 64 * WM * WN;
  constexpr int MT = BM / WM / 16, NT = BN / WN / 16;
  constexpr int A_BYTES = BM * BKT * 2, B_BYTES = BN * BKT * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;

  // XCD-aware bijective remap: blocks b and b+8 share an XCD (and its L2); give each XCD a
  // contiguous run of logical ids so neighbouring tiles (same A panel) hit the same L2.
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int ntiles = GRP ? gp.tile0[gp.n] : ntn * ntm;
  const int nitems = ntiles * p.splitk;
  const int nkt = (p.K + BKT - 1) / BKT;
  // (split, problem, tile row, tile column) of a work item; one problem unless GRP
  auto decode = [&](int it, int& sp, int& pj, int& tm_, int& tn_) {
    sp = it / ntiles;
    int t2 = it - sp * ntiles;
    if constexpr (GRP) {
      pj = 0;
#pragma unroll
      for (int j = 1; j < GROUP_MAX; ++j) pj = (j < gp.n && t2 >= gp.tile0[j]) ? j : pj;
      t2 -= gp.tile0[pj];
      const int w = gp.pr[pj].ntn;
      tm_ = t2 / w; tn_ = t2 - tm_ * w;
    } else {
      pj = 0;
      tm_ = t2 / ntn; tn_ = t2 - tm_ * ntn;
    }
  };

  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);

  f4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  static_assert(NT == 4, "the epilogue assumes 64 columns per wave");
  static_assert((BM + BN) * 8 / NTH == 8, "the counted vmcnt below assumes 8 LDS-DMA loads per thread per K-tile");

  // current position (item, kt) and the decoded tile of the item
  int item = lid;
  if (item >= nitems) return;
  int split, cprob, tm, tn;
  decode(item, split, cprob, tm, tn);
  int kt = split * p.ktps, kt_end = min(nkt, kt + p.ktps);     // host guarantees kt < kt_end for every item
  int buf = 0;
#if HMMC_DBG >= 3
  // 128 stamps per wave in two registers (v_writelane: stamp n lives in lane n), from the start of the wave's third item
  unsigned st0 = 0, st1 = 0;
  int sn = 0, items_done = 0;
#define HMMC_STAMP() do { if (items_done >= 2 && sn < 128) { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memrealtime(); \
    st0 = (lane == sn) ? t_ : st0; st1 = (lane == sn - 64) ? t_ : st1; ++sn; } } while (0)
#else
#define HMMC_STAMP() do {} while (0)
#endif
  if constexpr (MT != 8) {
    stage_tile<AK, BM, NTH>(ra, smem, wid, tid, tm * BM, kt * BKT, p.lda);
    stage_tile<BK, BN, NTH>(rb, smem + A_BYTES, wid, tid, tn * BN, kt * BKT, p.ldb);
  }

  int n_prob = 0;
  auto next_pos = [&](int& n_item, int& n_split, int& n_tm, int& n_tn, int& n_kt, int& n_end) {
    n_item = item; n_split = split; n_tm = tm; n_tn = tn; n_kt = kt + 1; n_end = kt_end; n_prob = cprob;
    if (n_kt >= kt_end) {
      n_item = item + nblk;
      if (n_item < nitems) {
        decode(n_item, n_split, n_prob, n_tm, n_tn);
        n_kt = n_split * p.ktps; n_end = min(nkt, n_kt + p.ktps);
      }
    }
  };
  // writes the finished tile and clears the accumulators for the next item (a zero C operand on the first K-tile
  // instead of the clear measured 1-5 % slower: it doubles the MFMA blocks of the main loop)
  auto finish_item = [&]() {
    HMMC_STAMP();
#if HMMC_DBG == 2 || HMMC_DBG == 5
    _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j) asm volatile("" :: "v"(acc[i][j]));
#else
    if constexpr (GRP)
      epilogue_slab<MT, NT>(p.ws + gp.pr[cprob].slab_off, gp.pr[cprob].M, gp.pr[cprob].N, acc, tm * BM + wm * (MT * 16) + (lane & 15),
                            tn * BN + wn * (NT * 16) + 4 * (lane >> 4), split);
    else if (p.splitk > 1 || p.slab)
      epilogue_slab<MT, NT>(p.ws, p.M, p.N, acc, tm * BM + wm * (MT * 16) + (lane & 15), tn * BN + wn * (NT * 16) + 4 * (lane >> 4), split);
    else
      epilogue_f16<MT, EPI>(p, acc, tm * BM + wm * (MT * 16), tn * BN + wn * (NT * 16), lane, smem + 2 * STAGE_BYTES + wid * EPI_LDS_PER_WAVE);
#endif
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#if HMMC_DBG >= 3
    HMMC_STAMP();
    ++items_done;
#endif
  };

  if constexpr (MT == 8 && NT == 4) {
    // ---- 256x256 tile, ping-pong schedule.  The two waves of a SIMD are wid and wid + 4, i.e. the wm = 0 and wm = 1
    // waves of one wn.  A K-tile is 2 segments of 32 MFMAs per wave (a 64x64 half of the wave's tile x K = 64); each is a
    // LOAD segment (LDS fragment reads + LDS-DMA of half-tiles of later K-tiles), a barrier, a MATRIX segment, a barrier.
    // The wm = 1 group runs one barrier behind, so on every SIMD one wave's matrix segment sits beside its partner's load
    // segment.  K-tile t lives in stage t & 1 as four half images A0, A1 (128 rows each), B0, B1:
    //   segment   reads (ds_read)              MFMA                      stages (LDS-DMA, 2 loads/thread and half)
    //     1       B0 (4) + A0 (8) + B1 (4)     (a0, b0), (a0, b1)        A1 of K-tile t+1
    //     2       A1 (8)                       (a1, b1), (a1, b0)        A0, B0, B1 of K-tile t+2 (the images segment 1 read)
    // Every load segment ends with s_waitcnt vmcnt(8) lgkmcnt(0): all but the four most recently staged halves have landed,
    // and the wave's own fragment reads have retired, before the barrier.
    // WAR: an image is restaged in the first load segment after a barrier that follows the last read of it by EITHER group
    // (segment 2 of K-tile t restages what both groups read in their segment 1; segment 1 of K-tile t+1 restages A1, read in
    // both groups' segment 2 of K-tile t), and those reads retired before that barrier.
    // RAW: every half is waited for in the load segment one K-tile (four barriers) after the one that staged it, and that
    // wait precedes the barriers both groups pass before the read: a half has a whole K-tile to arrive, against half a
    // K-tile for the last-staged half of the four-phase schedule of rounds 1-2 (-DHMMC_PHASES=4), which also paid eight
    // barriers per K-tile instead of four and ran its heaviest load segment (12 reads + a half) beside only 16 MFMAs.
    // Measured on a layer's twelve GEMMs (scratch/ab_lib.sh): 6 460 -> 5 960 us; the weight gradients and the m-major data
    // gradients, whose operands stream from HBM, gain most (+12-14 %).  Past the last K-tile the DMA is still issued, out of
    // range (reads as zero into a dead image), so the count stays exact; epilogue stores only make the wait stricter.
    constexpr int HALF = 128 * BKT * 2;
    // operands of the problem being STAGED (the prefetch runs ahead of the computation across item boundaries, so under GRP
    // it can already be in the next problem); without GRP these never change
    unsigned vA = half_vbase<AK, true>(tid, p.lda), vB = half_vbase<BK, false>(tid, p.ldb);
    int s_lda = p.lda, s_ldb = p.ldb;
    auto s_bind = [&](int pj) {
      if constexpr (GRP) {
        const GroupProb& q = gp.pr[pj];
        ra = __builtin_amdgcn_make_buffer_rsrc((void*)q.A, 0, (int)q.a_bytes, 0x00020000);
        rb = __builtin_amdgcn_make_buffer_rsrc((void*)q.B, 0, (int)q.b_bytes, 0x00020000);
        s_lda = q.lda; s_ldb = q.ldb;
        vA = half_vbase<AK, true>(tid, s_lda); vB = half_vbase<BK, false>(tid, s_ldb);
      }
    };
    s_bind(cprob);
    int s_item = item, s_tm = tm, s_tn = tn, s_kt = kt, s_end = kt_end, s_buf = 0;
    bool s_ok = true;
    auto s_advance = [&]() {
      ++s_kt; s_buf ^= 1;
      if (s_kt >= s_end) {
        s_item += nblk;
        s_ok = s_item < nitems;
        if (s_ok) {
          int sp, pj;
          decode(s_item, sp, pj, s_tm, s_tn);
          s_bind(pj);
          s_kt = sp * p.ktps; s_end = min(nkt, s_kt + p.ktps);
        }
      }
    };
    auto stage_a = [&](int h) {
      unsigned so = AK ? ((unsigned)(s_tm * BM + h * 64) * (unsigned)s_lda + (unsigned)(s_kt * BKT)) * 2u
                       : ((unsigned)(s_kt * BKT) * (unsigned)s_lda + (unsigned)(s_tm * BM + h * 64)) * 2u;
      if (!s_ok) so = 0x80000000u;
      stage_half<AK>(ra, smem + s_buf * (4 * HALF) + h * HALF, wid, vA, so, s_lda);
    };
    auto stage_b = [&](int h) {
      unsigned so = BK ? ((unsigned)(s_tn * BN + h * 32) * (unsigned)s_ldb + (unsigned)(s_kt * BKT)) * 2u
                       : ((unsigned)(s_kt * BKT) * (unsigned)s_ldb + (unsigned)(s_tn * BN + h * 32)) * 2u;
      if (!s_ok) so = 0x80000000u;
      stage_half<BK>(rb, smem + s_buf * (4 * HALF) + (2 + h) * HALF, wid, vB, so, s_ldb);
    };
#define HMMC_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); \
                        __builtin_amdgcn_sched_barrier(0); } while (0)
#define HMMC_MM(I0, J0, BF) do { __builtin_amdgcn_s_setprio(1); \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
      acc[I0 + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(BF[ks][j], af[ks][i], acc[I0 + i][J0 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0); } while (0)
#if HMMC_PHASES == 2
    stage_a(0); stage_b(0); stage_b(1); stage_a(1);
    s_advance();
    stage_a(0); stage_b(0); stage_b(1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // A0, B0, B1 of the first K-tile have landed
#else
#if HMMC_PF == 6
#define HMMC_VMW "s_waitcnt vmcnt(8)"
#define HMMC_ST1() stage_b(1)
#define HMMC_ST2() do { stage_a(1); s_advance(); } while (0)
#define HMMC_ST3() stage_a(0)
#define HMMC_ST4() stage_b(0)
    stage_a(0); stage_b(0); stage_b(1); stage_a(1);
    s_advance();
    stage_a(0); stage_b(0);
#else      // distance 4: two half-tiles in flight at every wait (sensitivity experiment)
#define HMMC_VMW "s_waitcnt vmcnt(4)"
#define HMMC_ST1() stage_a(0)
#define HMMC_ST2() stage_b(0)
#define HMMC_ST3() stage_b(1)
#define HMMC_ST4() do { stage_a(1); s_advance(); } while (0)
    stage_a(0); stage_b(0); stage_b(1); stage_a(1);
    s_advance();
#endif
    asm volatile(HMMC_VMW ::: "memory");
#endif
    HMMC_BAR();
    if (wm == 1) HMMC_BAR();
    const int arow = wm * 64, brow = wn * 32;
    // EPI_LNFOLD: the epilogue's row pairs (this wave's 128 rows: 1 KiB) and column pairs (its 64 columns: c | d, 512 B) arrive
    // by two LDS-DMA loads issued at the top of the item's LAST K-tile into the operand half of the wave's epilogue scratch
    // (free in these kernels: no residual / auxiliary operand).  They are older than that K-tile's eight staging loads, so its
    // closing s_waitcnt vmcnt(8) covers them; its first wait allows the two extra loads in flight (vmcnt(10)).
    constexpr bool FOLD_PRE = EPI >= 0 && (EPI & EPI_LNFOLD) != 0;
    __amdgpu_buffer_rsrc_t rrow = ra, rcol = ra;
    if constexpr (FOLD_PRE) {
      rrow = __builtin_amdgcn_make_buffer_rsrc((void*)p.rowstat, 0, p.M * 8, 0x00020000);
      rcol = __builtin_amdgcn_make_buffer_rsrc((void*)p.colterms, 0, p.N * 8, 0x00020000);
    }
    while (true) {
      const char* base = smem + buf * (4 * HALF);
      h8 af[2][4], b0f[2][2], b1f[2][2];
      const bool last_kt = FOLD_PRE && kt + 1 >= kt_end;
      if constexpr (FOLD_PRE) {
        if (last_kt) {
          char* const sin = smem + 2 * STAGE_BYTES + wid * EPI_LDS_PER_WAVE + 2048;
          const unsigned roff = (unsigned)(tm * BM + wm * 128) * 8u + (unsigned)lane * 16u;
          const int n0w = tn * BN + wn * 64;
          const unsigned coff = lane < 16 ? (unsigned)(n0w + 4 * lane) * 4u
                                          : (lane < 32 ? (unsigned)(p.N + n0w + 4 * (lane - 16)) * 4u : 0x80000000u);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rrow, LDS_PTR(sin), 16, roff, 0, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rcol, LDS_PTR(sin + 1024), 16, coff, 0, 0, 0);
        }
      }
#if HMMC_PHASES == 2
      // segment 1
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) b0f[ks][j] = read_frag<BK, 128>(base + 2 * HALF, brow + j * 16, ks, lane);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) af[ks][i] = read_frag<AK, 128>(base, arow + i * 16, ks, lane);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) b1f[ks][j] = read_frag<BK, 128>(base + 3 * HALF, brow + j * 16, ks, lane);
      stage_a(1); s_advance();
      if (last_kt) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      HMMC_BAR();
      HMMC_MM(0, 0, b0f);
      HMMC_MM(0, 2, b1f);
      HMMC_BAR();
      HMMC_STAMP();
      // segment 2
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) af[ks][i] = read_frag<AK, 128>(base + HALF, arow + i * 16, ks, lane);
      stage_a(0); stage_b(0); stage_b(1);
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      HMMC_BAR();
      HMMC_MM(4, 2, b1f);
      HMMC_MM(4, 0, b0f);
      HMMC_BAR();
      HMMC_STAMP();
#else
      // phase 1
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) b0f[ks][j] = read_frag<BK, 128>(base + 2 * HALF, brow + j * 16, ks, lane);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) af[ks][i] = read_frag<AK, 128>(base, arow + i * 16, ks, lane);
      HMMC_ST1();
      asm volatile(HMMC_VMW ::: "memory");
      HMMC_BAR();
      HMMC_MM(0, 0, b0f);
      HMMC_BAR();
      HMMC_STAMP();
      // phase 2
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) b1f[ks][j] = read_frag<BK, 128>(base + 3 * HALF, brow + j * 16, ks, lane);
      HMMC_ST2();
      asm volatile(HMMC_VMW ::: "memory");
      HMMC_BAR();
      HMMC_MM(0, 2, b1f);
      HMMC_BAR();
      HMMC_STAMP();
      // phase 3
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) af[ks][i] = read_frag<AK, 128>(base + HALF, arow + i * 16, ks, lane);
      HMMC_ST3();
      asm volatile(HMMC_VMW ::: "memory");
      HMMC_BAR();
      HMMC_MM(4, 2, b1f);
      HMMC_BAR();
      HMMC_STAMP();
      // phase 4
      HMMC_ST4();
      asm volatile(HMMC_VMW ::: "memory");
      HMMC_BAR();
      HMMC_MM(4, 0, b0f);
      HMMC_BAR();
      HMMC_STAMP();
#endif
      int n_item, n_split, n_tm, n_tn, n_kt, n_end;
      next_pos(n_item, n_split, n_tm, n_tn, n_kt, n_end);
      if (kt + 1 >= kt_end) {
        // Item boundary: the leading group waits one barrier for the other group's last matrix segment, so both
        // groups run their (VALU- and store-bound) epilogues TOGETHER and fill each other's issue bubbles, instead
        // of one after the other with the matrix pipe idle both times; afterwards the wm = 1 group falls one
        // barrier behind again.  Barrier counts stay equal: +1 per item for wm = 0, +1 per item but the last
        // (and the one at kernel start) for wm = 1.
        if (wm == 0) HMMC_BAR();
        finish_item();
        if (wm == 1 && n_item < nitems) HMMC_BAR();
      }
      if (n_item >= nitems) break;
      item = n_item; split = n_split; tm = n_tm; tn = n_tn; kt = n_kt; kt_end = n_end; cprob = n_prob;
      buf ^= 1;
    }
#undef HMMC_BAR
#undef HMMC_MM
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // trailing out-of-range DMA must land before the LDS is released
#if HMMC_DBG >= 3
    if (p.ws) {
      unsigned* so = reinterpret_cast<unsigned*>(p.ws) + ((size_t)bid * 8 + wid) * 128;
      so[lane] = st0; so[64 + lane] = st1;
    }
#endif
  } else {
    // ---- 128x128 tile, two workgroups per CU overlap each other: simple two-barrier loop
    while (true) {
      int n_item, n_split, n_tm, n_tn, n_kt, n_end;
      next_pos(n_item, n_split, n_tm, n_tn, n_kt, n_end);
      const bool has_next = n_item < nitems;
      char* sa = smem + buf * STAGE_BYTES;
      char* sb = sa + A_BYTES;
      if (has_next) {
        char* na = smem + (buf ^ 1) * STAGE_BYTES;
        stage_tile<AK, BM, NTH>(ra, na, wid, tid, n_tm * BM, n_kt * BKT, p.lda);
        stage_tile<BK, BN, NTH>(rb, na + A_BYTES, wid, tid, n_tn * BN, n_kt * BKT, p.ldb);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        h8 af[MT], bf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = read_frag<BK, BN>(sb, wn * (NT * 16) + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = read_frag<AK, BM>(sa, wm * (MT * 16) + i * 16, ks, lane);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 1 >= kt_end) finish_item();
      if (!has_next) break;
      item = n_item; split = n_split; tm = n_tm; tn = n_tn; kt = n_kt; kt_end = n_end;
      buf ^= 1;
    }
  }
}

template <bool AK, bool BK, int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_f16_kernel(GemmArgs p) {
  gemm_f16_body<AK, BK, BM, BN, WM, WN, EPI, false>(p, GemmGroup{});
}
// the grouped weight-gradient launch (m-major operands, 256x256 tiles, slabs only)
__global__ __launch_bounds__(512, 2) void gemm_f16_wgrad_group_kernel(GemmArgs p, GemmGroup gp) {
  gemm_f16_body<false, false, 256, 256, 2, 4, 0, true>(p, gp);
}

// the slabs of a grouped launch -> the fp16 results: problem j, element e of its [M_j][N_j]
// C32[j] != nullptr: problem j leaves as fp32 sums [M_j][N_j] (dense) instead of fp16: the folded weight gradients, which are
// corrected and scaled by gamma afterwards (ln_fold.hip: fold_grad_finish)
struct GroupOut { half_t* C[GROUP_MAX]; float* C32[GROUP_MAX]; long slab_off[GROUP_MAX]; long elems[GROUP_MAX + 1]; int N[GROUP_MAX], ldc[GROUP_MAX]; int n, S; };
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(const float* __restrict__ ws, GroupOut go) {
  const long total4 = go.elems[go.n] / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 4;
    int j = 0;
#pragma unroll
    for (int t = 1; t < GROUP_MAX; ++t) j = (t < go.n && e >= go.elems[t]) ? t : j;
    const long le = e - go.elems[j], slab = go.elems[j + 1] - go.elems[j];
    const float* src = ws + go.slab_off[j] + le;
    f4 s = *reinterpret_cast<const f4*>(src);
    for (int k = 1; k < go.S; ++k) s += *reinterpret_cast<const f4*>(src + k * slab);
    if (go.C32[j]) { *reinterpret_cast<f4*>(go.C32[j] + le) = s; continue; }
    const int m = (int)(le / go.N[j]), n = (int)(le - (long)m * go.N[j]);
    h4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (half_t)s[r];
    *reinterpret_cast<h4*>(go.C[j] + (size_t)m * go.ldc[j] + n) = o;
  }
}

// out[m][n] = fp16( sum_s ws[s][m][n] )
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, half_t* __restrict__ C,
                                                            int M, int N, int ldc, int S) {
  size_t total4 = (size_t)M * N / 4;
  size_t slab = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    f4 s = *reinterpret_cast<const f4*>(ws + i * 4);
    for (int k = 1; k < S; ++k) {
      f4 t = *reinterpret_cast<const f4*>(ws + k * slab + i * 4);
      s += t;
    }
    size_t e = i * 4;
    int m = (int)(e / N), n = (int)(e - (size_t)m * N);
    h4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (half_t)s[r];
    *reinterpret_cast<h4*>(C + (size_t)m * ldc + n) = o;
  }
}

struct TileCfg { int bm, bn, splitk; };

int g_reserved_cus = 0;      // hmmc_gemm_reserve_cus

// compute units the persistent grids may occupy on the current device
int gemm_cus() {
  const int num_cu = hmmc_num_cus();
  return num_cu - g_reserved_cus > 8 ? num_cu - g_reserved_cus : 8;
}

// Large tower GEMMs take the 256x256 tile; anything that would leave most of a 256-wide tile empty, or
// that cannot fill the chip with 256x256 tiles even after splitting K, takes 128x128.
TileCfg pick_cfg(int M, int N, int K, bool allow_split) {
  const int nkt = (K + BKT - 1) / BKT;
  auto tiles_of = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
  const int cus = gemm_cus();
  TileCfg c;
  bool big = (M % 256 == 0 || M >= 2048) && (N % 256 == 0 || N >= 2048) && M >= 256 && N >= 256;
  if (big) {
    long t = tiles_of(256, 256);
    long reach = allow_split ? t * (nkt / 8 > 0 ? nkt / 8 : 1) : t;
    if (reach < cus * 3 / 4) big = false;    // cannot occupy most of the CUs
  }
  c.bm = c.bn = big ? 256 : 128;
  c.splitk = 1;
  if (allow_split) {
    // one resident workgroup per CU (two for the small tile): split K just far enough to fill the chip once,
    // so the fp32 slab traffic (splitk * M * N * 8 bytes written + read) stays small
    long t = tiles_of(c.bm, c.bn);
    long slots = big ? cus : 2 * cus;
    if (t * 2 <= slots && nkt >= 8) {
      long s = slots / t;
      if (s > nkt / 4) s = nkt / 4;
      c.splitk = (int)(s > 1 ? s : 1);
    }
  }
  return c;
}

template <bool AK, bool BK, int BM, int BN, int WM, int WN, int EPI>
void launch_one(const GemmArgs& p, dim3 grid, hipStream_t stream) {
  constexpr int SMEM = 2 * (BM + BN) * BKT * 2 + WM * WN * EPI_LDS_PER_WAVE;      // two stages + the epilogue's LDS tiles
  if (SMEM > 64 * 1024) {
    static bool done[HMMC_MAX_DEVICES] = {false};
    hmmc_allow_lds((const void*)gemm_f16_kernel<AK, BK, BM, BN, WM, WN, EPI>, SMEM, done);
  }
  hipLaunchKernelGGL((gemm_f16_kernel<AK, BK, BM, BN, WM, WN, EPI>), grid, dim3(64 * WM * WN), SMEM, stream, p);
}

// One kernel per (operand layout, epilogue) the towers use, so that a heavy epilogue (QuickGELU, residual reads, column
// sums) cannot cost the plain kernels registers; every other combination runs the generic kernel (EPI = -1).
template <int BM, int BN, int WM, int WN>
void launch_cfg(const GemmArgs& p, bool ak, bool bk, dim3 grid, hipStream_t stream) {
  const int f = p.csum ? -1 : p.flags;
  if (ak && bk) {                 // forward: y = x W^T
    switch (f) {
      case 0: return launch_one<true, true, BM, BN, WM, WN, 0>(p, grid, stream);
      case EPI_BIAS: return launch_one<true, true, BM, BN, WM, WN, EPI_BIAS>(p, grid, stream);
      case EPI_BIAS | EPI_RESID: return launch_one<true, true, BM, BN, WM, WN, EPI_BIAS | EPI_RESID>(p, grid, stream);
      case EPI_BIAS | EPI_QGELU: return launch_one<true, true, BM, BN, WM, WN, EPI_BIAS | EPI_QGELU>(p, grid, stream);
      case EPI_BIAS | EPI_QGELU | EPI_SAVE_DGELU:
        return launch_one<true, true, BM, BN, WM, WN, EPI_BIAS | EPI_QGELU | EPI_SAVE_DGELU>(p, grid, stream);
      case EPI_LNFOLD: return launch_one<true, true, BM, BN, WM, WN, EPI_LNFOLD>(p, grid, stream);
      case EPI_LNFOLD | EPI_QGELU: return launch_one<true, true, BM, BN, WM, WN, EPI_LNFOLD | EPI_QGELU>(p, grid, stream);
      case EPI_LNFOLD | EPI_QGELU | EPI_SAVE_DGELU:
        return launch_one<true, true, BM, BN, WM, WN, EPI_LNFOLD | EPI_QGELU | EPI_SAVE_DGELU>(p, grid, stream);
      case EPI_BIAS | EPI_RESID | EPI_ROWSTAT: return launch_one<true, true, BM, BN, WM, WN, EPI_BIAS | EPI_RESID | EPI_ROWSTAT>(p, grid, stream);
      default: return launch_one<true, true, BM, BN, WM, WN, -1>(p, grid, stream);
    }
  } else if (ak && !bk) {         // dgrad: dx = dy W
    if (f == 0) return launch_one<true, false, BM, BN, WM, WN, 0>(p, grid, stream);
    if (p.csum && (p.flags & ~EPI_COLSUM) == EPI_MULAUX) return launch_one<true, false, BM, BN, WM, WN, EPI_MULAUX | EPI_COLSUM>(p, grid, stream);
    if (p.csum && (p.flags & ~EPI_COLSUM) == (EPI_MULAUX | EPI_ROWSCALE))
      return launch_one<true, false, BM, BN, WM, WN, EPI_MULAUX | EPI_COLSUM | EPI_ROWSCALE>(p, grid, stream);
    if (f == EPI_DGELU) return launch_one<true, false, BM, BN, WM, WN, EPI_DGELU>(p, grid, stream);
    return launch_one<true, false, BM, BN, WM, WN, -1>(p, grid, stream);
  } else if (!ak && !bk) {        // wgrad: dW = dy^T x
    if (f == 0) return launch_one<false, false, BM, BN, WM, WN, 0>(p, grid, stream);
    return launch_one<false, false, BM, BN, WM, WN, -1>(p, grid, stream);
  }
  return launch_one<false, true, BM, BN, WM, WN, -1>(p, grid, stream);
}

}  // namespace

// ---- optional live timing (bench.py): HIP events recorded on the launch stream around every hmmc_gemm_f16 call.
// Process-wide and off by default; the only mutable state in the library, used by the benchmark alone.
#include <atomic>
#include <mutex>
#include <vector>
namespace {
struct GemmProfRec { hipEvent_t e0, e1; double flops, bytes; int layout; };
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;                 // hmmc_gemm_f16 may be called from several host threads (autograd, one per device)
std::vector<GemmProfRec> g_prof;
}  // namespace

// launch sites of other translation units (gemm_f32.hip): see options.h
long hmmc_prof_begin(double flops, double bytes, int slot, hipStream_t stream) {
  if (!g_prof_on) return -1;
  GemmProfRec rec{};
  if (hipEventCreate(&rec.e0) != hipSuccess) return -1;
  if (hipEventCreate(&rec.e1) != hipSuccess) { (void)hipEventDestroy(rec.e0); return -1; }
  rec.flops = flops; rec.bytes = bytes; rec.layout = slot;
  (void)hipEventRecord(rec.e0, stream);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(rec);
  return (long)g_prof.size() - 1;
}
void hmmc_prof_end(long token, hipStream_t stream) {
  if (token < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (token < (long)g_prof.size()) (void)hipEventRecord(g_prof[token].e1, stream);
}

extern "C" int hmmc_gemm_profile_start(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_prof.clear();
  g_prof_on = true;
  return HMMC_OK;
}

// out arrays of 4: slot 0 = forward (k-major x k-major), 1 = dgrad (k-major x m-major), 2 = wgrad (m-major A) of hmmc_gemm_f16,
// 3 = hmmc_gemm_f32 (every orientation).  bytes = algorithmic operand bytes (A + B + C and the epilogue's bias / residual / aux
// tensors, each touched once)
extern "C" int hmmc_gemm_profile_stop(double* flops, double* bytes, double* seconds, long* launches) {
  g_prof_on = false;
  if (!flops || !bytes || !seconds || !launches) return HMMC_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < 4; ++i) { flops[i] = 0; bytes[i] = 0; seconds[i] = 0; launches[i] = 0; }
  if (hipDeviceSynchronize() != hipSuccess) return HMMC_ERR_LAUNCH;
  for (auto& r : g_prof) {
    float ms = 0.f;
    if (r.layout >= 0 && r.layout < 4 && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      flops[r.layout] += r.flops; bytes[r.layout] += r.bytes; seconds[r.layout] += ms * 1e-3; launches[r.layout] += 1;
    }
    (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
  }
  g_prof.clear();
  return HMMC_OK;
}

// Compute units the persistent grid leaves free.  The 256x256 kernel holds a CU's whole register file and most of its
// LDS for the length of a launch, so a collective's workgroups (RCCL under data parallelism) would otherwise only be
// placed at kernel boundaries, and the next GEMM's workgroups would then queue behind them.
extern "C" int hmmc_gemm_reserve_cus(int cus) {
  if (cus < 0 || cus > 128) return HMMC_ERR_ARG;
  g_reserved_cus = cus;
  return HMMC_OK;
}

// rows of the fp32 [rows][N] partial matrix an EPI_COLSUM launch writes into `workspace`
extern "C" size_t hmmc_gemm_f16_colsum_rows(int M, int N, int K) {
  TileCfg c = pick_cfg(M, N, K, false);
  return (size_t)((M + c.bm - 1) / c.bm) * 2;   // two wave rows per tile in both configurations
}

// ---- operands of 2 GiB and more -----------------------------------------------------------------------------------------
// The kernels address their operands through 32-bit buffer offsets (with bit 31 as the "read zeros" escape of the staging
// code), so one launch sees at most 2 GiB of A and of B.  Larger operands - the MLP activations of ViT-B/16 at B = 128,
// F = 24 on ONE GPU are 605 184 tokens x 3072 x 2 B = 3.7 GB - are cut on the host, with 64-bit base pointers per piece:
//   a k-major A (tokens x features: forward, data gradient) along M, every piece an independent GEMM on its rows;
//   m-major operands (weight gradient: tokens are the K dimension) along K, every piece adding its split-K slabs to ONE
//   fp32 slab set that a single reduce turns into the result (the weight-gradient path always has epilogue 0).
// k_chunk_rows(ld): K rows one piece may span; m_chunk_rows(lda): rows of a k-major A.
constexpr uint64_t PIECE_BYTES = (1ull << 31) - (1ull << 24);       // room for the 256-row / 64-k overreach of the last tile
static inline long k_chunk_rows(int ld) { long r = (long)(PIECE_BYTES / ((uint64_t)ld * 2)); return r - r % BKT; }
static inline long m_chunk_rows(int lda) { long r = (long)(PIECE_BYTES / ((uint64_t)lda * 2)); return r - r % 256; }
static inline bool needs_k_pieces(int K, int lda, int ldb, bool a_kmajor, bool b_kmajor) {
  return (!a_kmajor && (uint64_t)(K + BKT) * lda * 2 >= PIECE_BYTES) || (!b_kmajor && (uint64_t)(K + BKT) * ldb * 2 >= PIECE_BYTES);
}

extern "C" size_t hmmc_gemm_f16_workspace(int M, int N, int K) {
  TileCfg c = pick_cfg(M, N, K, true);
  // a weight gradient over more tokens than one piece holds (assuming operands as wide as M and N) needs a slab per piece
  const int ld = M > N ? M : N;
  if (needs_k_pieces(K, ld, ld, false, false)) {
    const long kc = k_chunk_rows(ld);
    size_t slabs = 0;
    for (long k0 = 0; k0 < K; k0 += kc) slabs += (size_t)pick_cfg(M, N, (int)((K - k0 < kc) ? K - k0 : kc), true).splitk;
    return slabs * M * N * sizeof(float);
  }
  return c.splitk > 1 ? (size_t)c.splitk * M * N * sizeof(float) : 0;
}

// one launch (+ its split-K reduce).  slab_mode 0: ordinary call.  slab_mode 1 (pieces along K): the fp32 partial slabs go to
// `workspace` (at least one, even without a split), no reduce; *slabs_out = slabs written.
struct GemmExtra { const float* rowstat; const float* colterms; float* stat_part; };      // operands of EPI_LNFOLD / EPI_ROWSTAT

static int gemm_f16_one(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                        int a_kmajor, int b_kmajor, const void* bias, const void* resid, void* aux_out,
                        const void* aux_in, int epilogue, void* workspace, size_t ws_bytes, hipStream_t stream,
                        int slab_mode, int* slabs_out, const GemmExtra& ex = GemmExtra{nullptr, nullptr, nullptr}) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return HMMC_ERR_ARG;
  if ((epilogue & EPI_LNFOLD) && (!ex.rowstat || !ex.colterms || (N & 3) || (((uintptr_t)ex.rowstat) & 15) || (((uintptr_t)ex.colterms) & 15)))
    return HMMC_ERR_ARG;
  if ((epilogue & EPI_ROWSCALE) && (!ex.rowstat || (((uintptr_t)ex.rowstat) & 7))) return HMMC_ERR_ARG;
  if ((epilogue & EPI_ROWSTAT) && (!ex.stat_part || (((uintptr_t)ex.stat_part) & 7))) return HMMC_ERR_ARG;
  if ((epilogue & EPI_ROWSTAT) && (N & 63)) return HMMC_ERR_UNSUPPORTED;             // whole 64-column blocks only
  if ((epilogue & (EPI_LNFOLD | EPI_ROWSTAT)) && !(a_kmajor && b_kmajor)) return HMMC_ERR_UNSUPPORTED;
  if ((epilogue & EPI_ROWSCALE) && !a_kmajor) return HMMC_ERR_UNSUPPORTED;
  if ((epilogue & EPI_LNFOLD) && (epilogue & (EPI_RESID | EPI_DGELU | EPI_MULAUX))) return HMMC_ERR_UNSUPPORTED;   // share an LDS tile
  if ((lda & 7) || (ldb & 7) || (ldc & 7) || (N & 7)) return HMMC_ERR_UNSUPPORTED;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)resid | (uintptr_t)aux_in | (uintptr_t)aux_out) & 15) return HMMC_ERR_UNSUPPORTED;
  if ((a_kmajor || b_kmajor) && (K % BKT)) return HMMC_ERR_UNSUPPORTED;   // k tail of a k-major operand
  if (!a_kmajor && (M & 7)) return HMMC_ERR_UNSUPPORTED;
  if ((epilogue & EPI_BIAS) && !bias) return HMMC_ERR_ARG;
  if ((epilogue & EPI_RESID) && !resid) return HMMC_ERR_ARG;
  if ((epilogue & (EPI_DGELU | EPI_MULAUX)) && !aux_in) return HMMC_ERR_ARG;
  // extents of the operand buffers (bytes); 32-bit buffer offsets
  uint64_t a_bytes = a_kmajor ? ((uint64_t)(M - 1) * lda + K) * 2 : ((uint64_t)(K - 1) * lda + M) * 2;
  uint64_t b_bytes = b_kmajor ? ((uint64_t)(N - 1) * ldb + K) * 2 : ((uint64_t)(K - 1) * ldb + N) * 2;
  uint64_t a_reach = a_kmajor ? (uint64_t)(M + 256) * lda * 2 : (uint64_t)(K + BKT) * lda * 2;
  uint64_t b_reach = b_kmajor ? (uint64_t)(N + 256) * ldb * 2 : (uint64_t)(K + BKT) * ldb * 2;
  if (a_reach >= (1ull << 32) || b_reach >= (1ull << 32) || a_bytes >= (1ull << 31) || b_bytes >= (1ull << 31))
    return HMMC_ERR_UNSUPPORTED;

  GemmArgs p;
  p.A = (const half_t*)A; p.B = (const half_t*)B; p.C = (half_t*)C;
  p.bias = (const half_t*)bias; p.resid = (const half_t*)resid; p.aux_out = (half_t*)aux_out;
  p.aux_in = (const half_t*)aux_in;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.flags = epilogue;
  p.a_bytes = (unsigned)a_bytes; p.b_bytes = (unsigned)b_bytes;
  p.rowstat = ex.rowstat; p.colterms = ex.colterms; p.stat_part = ex.stat_part;
  int nkt = (K + BKT - 1) / BKT;
  TileCfg cfg = pick_cfg(M, N, K, epilogue == 0);
  p.csum = nullptr;
  if (epilogue & EPI_COLSUM) {
    if ((N & 3) || !workspace || ws_bytes < hmmc_gemm_f16_colsum_rows(M, N, K) * N * sizeof(float)) return HMMC_ERR_WORKSPACE;
    p.csum = (float*)workspace;
  }
  int splitk = cfg.splitk;
  if (splitk > 1 && (!workspace || ws_bytes < (size_t)splitk * M * N * sizeof(float))) splitk = 1;
  p.ktps = (nkt + splitk - 1) / splitk;
  splitk = (nkt + p.ktps - 1) / p.ktps;        // every split owns at least one K-tile
  if (slab_mode) {
    if (!workspace || ws_bytes < (size_t)splitk * M * N * sizeof(float)) return HMMC_ERR_WORKSPACE;
    *slabs_out = splitk;
  }
  p.slab = slab_mode;
  p.splitk = splitk;
  p.ws = (float*)workspace;
  long tiles = (long)((M + cfg.bm - 1) / cfg.bm) * ((N + cfg.bn - 1) / cfg.bn);
  long items = tiles * splitk;
  const int cus = gemm_cus();
  long resident = (long)cus * (cfg.bm == 256 ? 1 : 2);          // workgroups the LDS budget keeps resident
  dim3 grid((unsigned)(items < resident ? items : resident));
  GemmProfRec rec{};
  if (g_prof_on) {
    (void)hipEventCreate(&rec.e0); (void)hipEventCreate(&rec.e1);
    rec.flops = 2.0 * M * N * K;
    double mn = (double)M * N;
    rec.bytes = 2.0 * ((double)M * K + (double)N * K + mn) + ((epilogue & EPI_BIAS) ? 2.0 * N : 0.0) +
                2.0 * mn * (((epilogue & EPI_RESID) ? 1 : 0) + ((epilogue & (EPI_DGELU | EPI_MULAUX)) ? 1 : 0) + (aux_out ? 1 : 0));
    if (epilogue & EPI_LNFOLD) rec.bytes += 8.0 * M + 8.0 * N;
    if (epilogue & EPI_ROWSTAT) rec.bytes += 8.0 * M * (N / 64);
    rec.layout = a_kmajor ? (b_kmajor ? 0 : 1) : 2;
    (void)hipEventRecord(rec.e0, stream);
  }
  if (cfg.bm == 256) launch_cfg<256, 256, 2, 4>(p, a_kmajor, b_kmajor, grid, stream);
  else launch_cfg<128, 128, 2, 2>(p, a_kmajor, b_kmajor, grid, stream);
  if (splitk > 1 && !slab_mode) {
    size_t nb = ((size_t)M * N / 4 + 255) / 256;
    int blocks = (int)(nb < 2048 ? nb : 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)workspace, (half_t*)C, M, N,
                       ldc, splitk);
  }
  if (rec.e0) { (void)hipEventRecord(rec.e1, stream); std::lock_guard<std::mutex> lk(g_prof_mu); g_prof.push_back(rec); }
  return hmmc_launch_status();
}

// ---- grouped weight gradients ---------------------------------------------------------------------------------------------
// dW_j[Np_j, Kp_j] = dY_j[T, Np_j]^T X_j[T, Kp_j], j < nprob <= 4, in ONE persistent launch + one reduce (see GemmGroup).
namespace {
struct GroupPlan { int splitk, ktps; long tiles, elems; };
// K split of a grouped launch: the estimated time of rounds x (K-tiles per item + fixed cost per item) plus the slab round
// trip, with the measured constants of the m-major K loop (1.97 us per K-tile, ~8 us per item, ~5 TB/s for the slabs)
GroupPlan group_plan(const int* Np, const int* Kp, int nprob, int T) {
  GroupPlan g{1, 1, 0, 0};
  for (int j = 0; j < nprob; ++j) {
    g.tiles += (long)((Np[j] + 255) / 256) * ((Kp[j] + 255) / 256);
    g.elems += (long)Np[j] * Kp[j];
  }
  const int nkt = (T + BKT - 1) / BKT;
  const int cus = gemm_cus();
  double best = 1e30;
  for (int s = 1; s <= 64 && s * 4 <= nkt; ++s) {
    const long items = g.tiles * s;
    const long rounds = (items + cus - 1) / cus;
    const int kti = (nkt + s - 1) / s;
    const double t = rounds * (kti * 1.97 + 8.0) + (s > 0 ? 8.0 * s * g.elems / 5.0e6 : 0.0);
    if (t < best) { best = t; g.splitk = s; }
  }
  g.ktps = (nkt + g.splitk - 1) / g.splitk;
  g.splitk = (nkt + g.ktps - 1) / g.ktps;
  return g;
}
bool group_ok(const int* Np, const int* Kp, int nprob, int T) {
  if (hmmc_option(HMMC_OPT_NO_WGRAD_GROUP)) return false;       // A/B runs: one launch per gradient, as before round 3
  // (at least 8 K-tiles: group_plan keeps >= 4 per item.  Round 5 lowered this from 2048 tokens: the text tower at 32 captions per
  // GPU - 1 024 tokens - ran 4 launches + 4 reduces per layer)
  if (!Np || !Kp || nprob < 1 || nprob > GROUP_MAX || T < 512) return false;
  for (int j = 0; j < nprob; ++j) {
    if (Np[j] < 256 || Kp[j] < 256 || (Np[j] % 256) || (Kp[j] % 256)) return false;       // 256x256 tiles only
    if ((uint64_t)(T + BKT) * (uint64_t)(Np[j] > Kp[j] ? Np[j] : Kp[j]) * 2 >= PIECE_BYTES) return false;   // 32-bit offsets
  }
  return true;
}
}  // namespace

// workspace bytes of hmmc_gemm_f16_wgrad_group, or 0 when these shapes should take one hmmc_gemm_f16 call per gradient
extern "C" size_t hmmc_gemm_f16_wgrad_group_workspace(const int* Np, const int* Kp, int nprob, int T) {
  if (!group_ok(Np, Kp, nprob, T)) return 0;
  const GroupPlan g = group_plan(Np, Kp, nprob, T);
  return (size_t)g.splitk * g.elems * sizeof(float);
}

extern "C" int hmmc_gemm_f16_wgrad_group(const void* const* dY, const void* const* X, void* const* dW, float* const* dW32,
                                         const int* Np, const int* Kp, int nprob, int T, void* workspace, size_t ws_bytes,
                                         hipStream_t stream) {
  if (!dY || !X || !dW || !group_ok(Np, Kp, nprob, T)) return HMMC_ERR_UNSUPPORTED;
  const GroupPlan g = group_plan(Np, Kp, nprob, T);
  if (!workspace || ws_bytes < (size_t)g.splitk * g.elems * sizeof(float)) return HMMC_ERR_WORKSPACE;
  GemmGroup gp{};
  GroupOut go{};
  gp.n = go.n = nprob;
  go.S = g.splitk;
  long tile0 = 0, off = 0, e0 = 0;
  double flops = 0, bytes = 0;
  for (int j = 0; j < nprob; ++j) {
    float* const w32 = dW32 ? dW32[j] : nullptr;
    if (!dY[j] || !X[j] || (!dW[j] && !w32) || (((uintptr_t)dY[j] | (uintptr_t)X[j] | (uintptr_t)dW[j] | (uintptr_t)w32) & 15)) return HMMC_ERR_ARG;
    go.C32[j] = w32;
    GroupProb& q = gp.pr[j];
    q.A = (const half_t*)dY[j]; q.B = (const half_t*)X[j];
    q.lda = Np[j]; q.ldb = Kp[j]; q.M = Np[j]; q.N = Kp[j];
    q.a_bytes = (unsigned)(((uint64_t)(T - 1) * Np[j] + Np[j]) * 2);
    q.b_bytes = (unsigned)(((uint64_t)(T - 1) * Kp[j] + Kp[j]) * 2);
    q.ntn = (Kp[j] + 255) / 256;
    q.slab_off = off;
    gp.tile0[j] = (int)tile0;
    go.C[j] = (half_t*)dW[j]; go.slab_off[j] = off; go.elems[j] = e0; go.N[j] = Kp[j]; go.ldc[j] = Kp[j];
    tile0 += (long)((Np[j] + 255) / 256) * q.ntn;
    off += (long)g.splitk * Np[j] * Kp[j];
    e0 += (long)Np[j] * Kp[j];
    flops += 2.0 * Np[j] * Kp[j] * (double)T;
    bytes += 2.0 * ((double)T * Np[j] + (double)T * Kp[j] + (double)Np[j] * Kp[j]);
  }
  gp.tile0[nprob] = (int)tile0;
  go.elems[nprob] = e0;
  GemmArgs p{};
  p.A = gp.pr[0].A; p.B = gp.pr[0].B; p.C = nullptr;
  p.M = Np[0]; p.N = Kp[0]; p.K = T; p.lda = Np[0]; p.ldb = Kp[0]; p.ldc = Kp[0];
  p.flags = 0; p.splitk = g.splitk; p.ktps = g.ktps; p.slab = 1;
  p.a_bytes = gp.pr[0].a_bytes; p.b_bytes = gp.pr[0].b_bytes;
  p.ws = (float*)workspace;
  const long items = tile0 * g.splitk;
  const int cus = gemm_cus();
  constexpr int SMEM = 2 * (256 + 256) * BKT * 2 + 8 * EPI_LDS_PER_WAVE;
  static bool done[HMMC_MAX_DEVICES] = {false};
  hmmc_allow_lds((const void*)gemm_f16_wgrad_group_kernel, SMEM, done);
  GemmProfRec rec{};
  if (g_prof_on) {
    (void)hipEventCreate(&rec.e0); (void)hipEventCreate(&rec.e1);
    rec.flops = flops; rec.bytes = bytes; rec.layout = 2;
    (void)hipEventRecord(rec.e0, stream);
  }
  hipLaunchKernelGGL(gemm_f16_wgrad_group_kernel, dim3((unsigned)(items < cus ? items : cus)), dim3(512), SMEM, stream, p, gp);
  const size_t nb = ((size_t)e0 / 4 + 255) / 256;
  hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, stream, (const float*)workspace, go);
  if (rec.e0) { (void)hipEventRecord(rec.e1, stream); std::lock_guard<std::mutex> lk(g_prof_mu); g_prof.push_back(rec); }
  return hmmc_launch_status();
}

static int gemm_f16_any(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                        int a_kmajor, int b_kmajor, const void* bias, const void* resid, void* aux_out,
                        const void* aux_in, int epilogue, void* workspace, size_t ws_bytes, hipStream_t stream, const GemmExtra& ex);

extern "C" int hmmc_gemm_f16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                             int a_kmajor, int b_kmajor, const void* bias, const void* resid, void* aux_out,
                             const void* aux_in, int epilogue, void* workspace, size_t ws_bytes, hipStream_t stream) {
  if (epilogue & (EPI_LNFOLD | EPI_ROWSTAT | EPI_ROWSCALE)) return HMMC_ERR_ARG;      // those take hmmc_gemm_f16_fold's operands
  return gemm_f16_any(A, B, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, bias, resid, aux_out, aux_in, epilogue, workspace, ws_bytes,
                      stream, GemmExtra{nullptr, nullptr, nullptr});
}

// hmmc_gemm_f16 (k-major A) with a LayerNorm folded in (HMMC_EPI_LNFOLD: rowstat + colterms), the row statistics of its output
// emitted for the NEXT folded GEMM (HMMC_EPI_ROWSTAT: stat_part), or its output rows scaled by their rstd (HMMC_EPI_ROWSCALE:
// rowstat; the data gradient in front of a folded LayerNorm); operands below 2 GiB.
extern "C" int hmmc_gemm_f16_fold(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int b_kmajor,
                                  const void* bias, const void* resid, void* aux_out, const void* aux_in, int epilogue,
                                  const float* rowstat, const float* colterms, float* stat_part, void* workspace, size_t ws_bytes,
                                  hipStream_t stream) {
  if (epilogue & ~(EPI_BIAS | EPI_RESID | EPI_QGELU | EPI_SAVE_DGELU | EPI_MULAUX | EPI_COLSUM | EPI_LNFOLD | EPI_ROWSTAT | EPI_ROWSCALE))
    return HMMC_ERR_UNSUPPORTED;
  return gemm_f16_any(A, B, C, M, N, K, lda, ldb, ldc, 1, b_kmajor, bias, resid, aux_out, aux_in, epilogue, workspace, ws_bytes, stream,
                      GemmExtra{rowstat, colterms, stat_part});
}

static int gemm_f16_any(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                        int a_kmajor, int b_kmajor, const void* bias, const void* resid, void* aux_out,
                        const void* aux_in, int epilogue, void* workspace, size_t ws_bytes, hipStream_t stream, const GemmExtra& ex) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda <= 0 || ldb <= 0 || ldc <= 0) return HMMC_ERR_ARG;
  const bool big_m = a_kmajor && (uint64_t)(M + 256) * lda * 2 >= PIECE_BYTES;
  const bool big_k = needs_k_pieces(K, lda, ldb, a_kmajor != 0, b_kmajor != 0);
  if (!big_m && !big_k)
    return gemm_f16_one(A, B, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, bias, resid, aux_out, aux_in, epilogue, workspace,
                        ws_bytes, stream, 0, nullptr, ex);
  if (epilogue & (EPI_LNFOLD | EPI_ROWSTAT | EPI_ROWSCALE)) return HMMC_ERR_UNSUPPORTED;
  const char* a8 = (const char*)A;
  const char* b8 = (const char*)B;
  if (big_m && !big_k) {                         // pieces of whole 256-row tiles; the column-sum partials follow the rows
    const long mc = m_chunk_rows(lda);
    if (mc < 256) return HMMC_ERR_UNSUPPORTED;
    size_t cs_off = 0;
    for (long m0 = 0; m0 < M; m0 += mc) {
      const int mp = (int)((M - m0 < mc) ? M - m0 : mc);
      const size_t roff = (size_t)m0 * ldc * 2;
      void* ws = workspace;
      size_t wsb = ws_bytes;
      if (epilogue & EPI_COLSUM) {
        if (pick_cfg(mp, N, K, false).bm != pick_cfg(M, N, K, false).bm) return HMMC_ERR_UNSUPPORTED;
        ws = (char*)workspace + cs_off;
        wsb = ws_bytes > cs_off ? ws_bytes - cs_off : 0;
        cs_off += hmmc_gemm_f16_colsum_rows(mp, N, K) * N * sizeof(float);
      }
      int rc = gemm_f16_one(a8 + (size_t)m0 * lda * 2, B, (char*)C + roff, mp, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, bias,
                            resid ? (const char*)resid + roff : nullptr, aux_out ? (char*)aux_out + roff : nullptr,
                            aux_in ? (const char*)aux_in + roff : nullptr, epilogue, ws, wsb, stream, 0, nullptr);
      if (rc) return rc;
    }
    return HMMC_OK;
  }
  if (big_m || epilogue != 0 || a_kmajor || b_kmajor) return HMMC_ERR_UNSUPPORTED;     // K pieces: the weight-gradient layout only
  long kc = k_chunk_rows(lda > ldb ? lda : ldb);
  if (kc < BKT) return HMMC_ERR_UNSUPPORTED;
  int slabs = 0;
  for (long k0 = 0; k0 < K; k0 += kc) {
    const int kp = (int)((K - k0 < kc) ? K - k0 : kc);
    const size_t used = (size_t)slabs * M * N * sizeof(float);
    if (!workspace || ws_bytes <= used) return HMMC_ERR_WORKSPACE;
    int wrote = 0;
    int rc = gemm_f16_one(a8 + (size_t)k0 * lda * 2, b8 + (size_t)k0 * ldb * 2, C, M, N, kp, lda, ldb, ldc, 0, 0, nullptr, nullptr,
                          nullptr, nullptr, 0, (char*)workspace + used, ws_bytes - used, stream, 1, &wrote);
    if (rc) return rc;
    slabs += wrote;
  }
  size_t nb = ((size_t)M * N / 4 + 255) / 256;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, stream, (const float*)workspace,
                     (half_t*)C, M, N, ldc, slabs);
  return hmmc_launch_status();
}
