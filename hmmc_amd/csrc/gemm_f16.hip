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
// Structure: 128x128x64 block tile, 4 waves (2x2, 64x64 each), 16x16x32 f16 MFMA with the
// operands swapped (MFMA-A = weight rows, MFMA-B = activation rows) so each lane owns 4
// consecutive n of one output row (8-byte stores).  Tiles are staged by LDS-DMA
// (buffer_load ... lds, 16 B/lane) through a bounds-checked buffer descriptor: rows past the
// end of a matrix read as zero, so ragged M/N/K need no branches and cannot fault.
// LDS images are lane-linear; bank conflicts are removed by XOR-swizzling the per-lane SOURCE
// chunk and applying the same XOR on the read (k-major: ds_read_b128; m-major:
// ds_read_b64_tr_b16 hardware transpose).  Two LDS stages; the prefetch of tile t+1 stays in
// flight across the compute of tile t (counted vmcnt + raw s_barrier).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BKT = 64;
constexpr int TILE_BYTES = BM * BKT * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;       // A + B
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;       // double buffered: 64 KiB

enum { EPI_BIAS = 1, EPI_RESID = 2, EPI_QGELU = 4, EPI_DGELU = 8 };

struct GemmArgs {
  const half_t* A; const half_t* B; half_t* C;
  const half_t* bias; const half_t* resid; half_t* aux_out; const half_t* aux_in;
  float* ws;
  int M, N, K, lda, ldb, ldc;
  int flags, splitk, ktps;
  unsigned a_bytes, b_bytes;
};

// ---- LDS-DMA staging -------------------------------------------------------------------------
// k-major tile image: [128 rows][8 chunks of 16 B]; phys chunk = logical ^ ((row >> 1) & 7)
// m-major tile image: [64 k-rows][16 chunks of 16 B]; phys chunk = logical ^ f(krow),
//                     f(krow) = ((krow & 3) << 2) | ((krow >> 2) & 3)
template <bool KMAJ>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int wid, int tid,
                                           int r0, int k0, int ld) {
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    unsigned goff;
    if (KMAJ) {
      int row = ps * 32 + (tid >> 3);
      int logical = (tid & 7) ^ ((row >> 1) & 7);
      goff = ((unsigned)(r0 + row) * (unsigned)ld + (unsigned)(k0 + logical * 8)) * 2u;
    } else {
      int krow = ps * 16 + (tid >> 4);
      int f = ((krow & 3) << 2) | ((krow >> 2) & 3);
      int logical = (tid & 15) ^ f;
      goff = ((unsigned)(k0 + krow) * (unsigned)ld + (unsigned)(r0 + logical * 8)) * 2u;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + ps * 4096 + wid * 1024), 16, goff, 0, 0, 0);
  }
}

// ---- fragment reads --------------------------------------------------------------------------
// returns the 8 halves op[row0 + (lane & 15)][ks*32 + 8*(lane >> 4) + j], j = 0..7
template <bool KMAJ>
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
    const char* a0 = lds_tile + krow * 256 + ((c ^ f0) << 4) + 8 * (pp & 1);
    const char* a1 = lds_tile + (krow + 4) * 256 + ((c ^ f1) << 4) + 8 * (pp & 1);
    fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)LDS_PTR(a0));
    fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)LDS_PTR(a1));
    h8 r;
    r[0] = (half_t)lo[0]; r[1] = (half_t)lo[1]; r[2] = (half_t)lo[2]; r[3] = (half_t)lo[3];
    r[4] = (half_t)hi[0]; r[5] = (half_t)hi[1]; r[6] = (half_t)hi[2]; r[7] = (half_t)hi[3];
    return r;
  }
}

template <bool AK, bool BK>
__global__ __launch_bounds__(256) void gemm_f16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;

  // XCD-aware bijective remap: blocks b and b+8 share an XCD (and its L2); give each XCD a
  // contiguous run of logical ids so neighbouring tiles (same A panel) hit the same L2.
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int ntiles = ntn * ntm;
  const int split = lid / ntiles;
  const int tile = lid - split * ntiles;
  const int tm = tile / ntn, tn = tile - tm * ntn;

  const int nkt = (p.K + BKT - 1) / BKT;
  const int kt0 = split * p.ktps;
  const int kt1 = min(nkt, kt0 + p.ktps);

  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);

  f4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  if (kt0 < kt1) {
    stage_tile<AK>(ra, smem, wid, tid, tm * BM, kt0 * BKT, p.lda);
    stage_tile<BK>(rb, smem + TILE_BYTES, wid, tid, tn * BN, kt0 * BKT, p.ldb);
  }
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    char* sa = smem + cur * STAGE_BYTES;
    char* sb = sa + TILE_BYTES;
    if (kt + 1 < kt1) {
      char* na = smem + (cur ^ 1) * STAGE_BYTES;
      stage_tile<AK>(ra, na, wid, tid, tm * BM, (kt + 1) * BKT, p.lda);
      stage_tile<BK>(rb, na + TILE_BYTES, wid, tid, tn * BN, (kt + 1) * BKT, p.ldb);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    h8 af[4][2], bf[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        af[i][ks] = read_frag<AK>(sa, wm * 64 + i * 16, ks, lane);
        bf[i][ks] = read_frag<BK>(sb, wn * 64 + i * 16, ks, lane);
      }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j][ks], af[i][ks], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: lane owns C[m][n..n+3], m = (lane & 15), n = 4 * (lane >> 4) within each 16x16 tile
  const int mrow = tm * BM + wm * 64 + (lane & 15);
  const int ncol = tn * BN + wn * 64 + 4 * (lane >> 4);
  if (p.splitk > 1) {
    float* ws = p.ws + (size_t)split * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = mrow + i * 16;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int n = ncol + j * 16;
        if (n < p.N) *reinterpret_cast<f4*>(ws + (size_t)m * p.N + n) = acc[i][j];
      }
    }
    return;
  }
  const int flags = p.flags;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = mrow + i * 16;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n = ncol + j * 16;
      if (n >= p.N) continue;
      f4 v = acc[i][j];
      if (flags & EPI_BIAS) {
        h4 b = *reinterpret_cast<const h4*>(p.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)b[r];
      }
      size_t off = (size_t)m * p.ldc + n;
      h4 o;
      if (flags & EPI_QGELU) {
        h4 hh;
#pragma unroll
        for (int r = 0; r < 4; ++r) { hh[r] = (half_t)v[r]; o[r] = (half_t)qgelu_f16((float)hh[r]); }
        if (p.aux_out) *reinterpret_cast<h4*>(p.aux_out + off) = hh;
      } else if (flags & EPI_DGELU) {
        h4 hh = *reinterpret_cast<const h4*>(p.aux_in + off);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)(v[r] * qgelu_grad((float)hh[r]));
      } else if (flags & EPI_RESID) {
        h4 rr = *reinterpret_cast<const h4*>(p.resid + off);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)((float)rr[r] + r16(v[r]));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
      }
      *reinterpret_cast<h4*>(p.C + off) = o;
    }
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

int pick_splitk(int M, int N, int K) {
  int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int nkt = (K + BKT - 1) / BKT;
  if (tiles >= 512 || nkt < 8) return 1;
  int s = (1024 + tiles - 1) / tiles;           // aim at ~4 blocks per CU
  if (s > nkt / 4) s = nkt / 4;
  return s > 1 ? s : 1;
}

}  // namespace

extern "C" size_t hmmc_gemm_f16_workspace(int M, int N, int K) {
  int s = pick_splitk(M, N, K);
  return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

extern "C" int hmmc_gemm_f16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                             int a_kmajor, int b_kmajor, const void* bias, const void* resid, void* aux_out,
                             const void* aux_in, int epilogue, void* workspace, size_t ws_bytes, hipStream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return HMMC_ERR_ARG;
  if ((lda & 7) || (ldb & 7) || (ldc & 3) || (N & 7)) return HMMC_ERR_UNSUPPORTED;
  if (((uintptr_t)A | (uintptr_t)B) & 15 || ((uintptr_t)C & 7)) return HMMC_ERR_UNSUPPORTED;
  if ((a_kmajor || b_kmajor) && (K % BKT)) return HMMC_ERR_UNSUPPORTED;   // k tail of a k-major operand
  if (!a_kmajor && (M & 7)) return HMMC_ERR_UNSUPPORTED;
  if ((epilogue & EPI_BIAS) && !bias) return HMMC_ERR_ARG;
  if ((epilogue & EPI_RESID) && !resid) return HMMC_ERR_ARG;
  if ((epilogue & EPI_DGELU) && !aux_in) return HMMC_ERR_ARG;
  // extents of the operand buffers (bytes); 32-bit buffer offsets
  uint64_t a_bytes = a_kmajor ? ((uint64_t)(M - 1) * lda + K) * 2 : ((uint64_t)(K - 1) * lda + M) * 2;
  uint64_t b_bytes = b_kmajor ? ((uint64_t)(N - 1) * ldb + K) * 2 : ((uint64_t)(K - 1) * ldb + N) * 2;
  uint64_t a_reach = a_kmajor ? (uint64_t)(M + BM) * lda * 2 : (uint64_t)(K + BKT) * lda * 2;
  uint64_t b_reach = b_kmajor ? (uint64_t)(N + BN) * ldb * 2 : (uint64_t)(K + BKT) * ldb * 2;
  if (a_reach >= (1ull << 32) || b_reach >= (1ull << 32) || a_bytes >= (1ull << 31) || b_bytes >= (1ull << 31))
    return HMMC_ERR_UNSUPPORTED;

  GemmArgs p;
  p.A = (const half_t*)A; p.B = (const half_t*)B; p.C = (half_t*)C;
  p.bias = (const half_t*)bias; p.resid = (const half_t*)resid; p.aux_out = (half_t*)aux_out;
  p.aux_in = (const half_t*)aux_in;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.flags = epilogue;
  p.a_bytes = (unsigned)a_bytes; p.b_bytes = (unsigned)b_bytes;
  int nkt = (K + BKT - 1) / BKT;
  int splitk = epilogue ? 1 : pick_splitk(M, N, K);
  if (splitk > 1 && (!workspace || ws_bytes < (size_t)splitk * M * N * sizeof(float))) splitk = 1;
  p.splitk = splitk;
  p.ktps = (nkt + splitk - 1) / splitk;
  p.ws = (float*)workspace;
  int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  dim3 grid(tiles * splitk), block(256);
  if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_f16_kernel<true, true>), grid, block, SMEM_BYTES, stream, p);
  else if (a_kmajor && !b_kmajor) hipLaunchKernelGGL((gemm_f16_kernel<true, false>), grid, block, SMEM_BYTES, stream, p);
  else if (!a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_f16_kernel<false, true>), grid, block, SMEM_BYTES, stream, p);
  else hipLaunchKernelGGL((gemm_f16_kernel<false, false>), grid, block, SMEM_BYTES, stream, p);
  if (splitk > 1) {
    size_t nb = ((size_t)M * N / 4 + 255) / 256;
    int blocks = (int)(nb < 2048 ? nb : 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)workspace, (half_t*)C, M, N,
                       ldc, splitk);
  }
  return hmmc_launch_status();
}
