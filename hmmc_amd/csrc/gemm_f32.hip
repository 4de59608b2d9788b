// fp32 MFMA GEMM (exact f32 FMA chain, v_mfma_f32_16x16x4_f32) for the fp32 side of the hot path:
// the temporal transformer (reference modules/module_cross.py:114-149,193-207), the similarity
// matrices of loose_similarity / contrastive_loss (modules/modeling.py:207-229,286-313), the MoCo
// projector MLPs (:788-807) and the MLM head (modules/module_cross.py:308-357).
//
//   C[m][n] = epilogue( alpha * sum_k A(m,k) * B(k,n) ),   A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
//
// One of each operand's strides must be 1 (either orientation is read with 16-byte loads), so
// x W^T, dy W, dy^T x, q K^T and q queue all go through the same kernel with no transposed copy.
// 64x64x16 block tile, 4 waves (2x2, 32x32 each), register-staged double buffering; MFMA operands
// swapped (A-operand = B rows) so a lane owns 4 consecutive n of one row: 16-byte stores.
#include "common.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16;
constexpr int LDS_LD = 80;   // floats per k-row: 64 + 16 keeps the two k-rows of a 32-lane half on disjoint banks

enum { EPI_BIAS = 1, EPI_RESID = 2, EPI_QGELU = 4, EPI_DGELU = 8, EPI_RELU = 16 };

struct G32 {
  const float* A; const float* B; float* C; const float* bias; const float* resid; float* aux_out; const float* aux_in;
  int M, N, K; long sam, sak, sbk, sbn; int ldc; float alpha; int flags; int avec, bvec, cvec;
};

__device__ __forceinline__ float qgelu32(float h) { return h / (1.0f + __expf(-1.702f * h)); }

// load this thread's 4 elements of a 64 x 16 operand tile (rows r0.., k-range k0..) into regs
__device__ __forceinline__ f4 load_op(const float* P, long sr, long sk, int R, int K, int r0, int k0, int tid, bool vec) {
  f4 v = {0.f, 0.f, 0.f, 0.f};
  if (sk == 1) {            // k contiguous: thread -> (row = tid/4, k = 4*(tid%4))
    int r = r0 + (tid >> 2), k = k0 + (tid & 3) * 4;
    if (r < R) {
      const float* p = P + (long)r * sr + k;
      if (vec && k + 3 < K) v = *reinterpret_cast<const f4*>(p);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (k + j < K) v[j] = p[j];
      }
    }
  } else {                  // row contiguous: thread -> (k = tid/16, row = 4*(tid%16))
    int k = k0 + (tid >> 4), r = r0 + (tid & 15) * 4;
    if (k < K) {
      const float* p = P + (long)k * sk + r;
      if (vec && r + 3 < R) v = *reinterpret_cast<const f4*>(p);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (r + j < R) v[j] = p[j];
      }
    }
  }
  return v;
}

// Loop-invariant part of load_op: the thread's source pointer for K-step 0, its per-step advance, and whether its
// 4 elements can be fetched as one aligned 16-byte load for every K-step that lies fully inside K (row / alignment
// conditions do not depend on k).  Steps that reach past K go through load_op.
struct OpLoader {
  const float* ptr; long adv; bool vec4, any;
};
__device__ __forceinline__ OpLoader make_loader(const float* P, long sr, long sk, int R, int r0, int tid, bool vec) {
  OpLoader L;
  if (sk == 1) {
    int r = r0 + (tid >> 2);
    L.ptr = P + (long)r * sr + (tid & 3) * 4; L.adv = TK; L.any = r < R; L.vec4 = vec && L.any;
  } else {
    int r = r0 + (tid & 15) * 4;
    L.ptr = P + (long)(tid >> 4) * sk + r; L.adv = (long)TK * sk; L.any = r < R; L.vec4 = vec && r + 3 < R;
  }
  return L;
}
__device__ __forceinline__ f4 load_inner(const OpLoader& L, long sk, int R, int r0, int kt, int tid) {
  const float* p = L.ptr + (long)kt * L.adv;
  if (L.vec4) return *reinterpret_cast<const f4*>(p);
  f4 v = {0.f, 0.f, 0.f, 0.f};
  if (L.any) {
    if (sk == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = p[j];
    } else {
      int r = r0 + (tid & 15) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) if (r + j < R) v[j] = p[j];
    }
  }
  return v;
}

// write them into the LDS image [k][row]
__device__ __forceinline__ void store_op(float* S, f4 v, bool kcontig, int tid) {
  if (kcontig) {
    int r = tid >> 2, k = (tid & 3) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) S[(k + j) * LDS_LD + r] = v[j];
  } else {
    int k = tid >> 4, r = (tid & 15) * 4;
    *reinterpret_cast<f4*>(S + k * LDS_LD + r) = v;
  }
}

// epilogue of one 16x16 MFMA tile: this lane's 4 consecutive outputs C[m][n .. n+3]
__device__ __forceinline__ void store_tile16(const G32& p, f4 v, int m, int n) {
  if (m >= p.M || n >= p.N) return;
  long off = (long)m * p.ldc + n;
  bool full = p.cvec && n + 3 < p.N;
  float o[4], hsave[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float t = v[r] * p.alpha;
    if ((p.flags & EPI_BIAS) && n + r < p.N) t += p.bias[n + r];
    hsave[r] = t;
    if (p.flags & EPI_QGELU) t = qgelu32(t);
    else if (p.flags & EPI_RELU) t = fmaxf(t, 0.f);
    else if ((p.flags & EPI_DGELU) && n + r < p.N) t *= qgelu_grad(p.aux_in[off + r]);
    if ((p.flags & EPI_RESID) && n + r < p.N) t += p.resid[off + r];
    o[r] = t;
  }
  if (full) {
    *reinterpret_cast<f4*>(p.C + off) = f4{o[0], o[1], o[2], o[3]};
    if (p.aux_out) *reinterpret_cast<f4*>(p.aux_out + off) = f4{hsave[0], hsave[1], hsave[2], hsave[3]};
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < p.N) { p.C[off + r] = o[r]; if (p.aux_out) p.aux_out[off + r] = hsave[r]; }
  }
}

// ---- small problems: four waves per 16 x 32 output tile, operands straight from global memory into the MFMA ------------
// A problem with fewer 64x64 tiles than CUs (the temporal transformer at a few hundred tokens, the similarity heads) is
// latency-bound in the kernel below: each of its few workgroups walks K in 16-deep steps of 16 MFMAs behind an LDS round trip
// and a barrier.  Here a workgroup owns a 16 x 32 tile and its four waves split K between them (wave w takes the 16-deep
// steps w, w + 4, ...), so the dependent MFMA chain is K/4 long and there are 4x as many waves to hide the load latency;
// the partial tiles meet in LDS in a fixed order.  Because the k index an MFMA lane group consumes is free to choose, lane
// (r = lane & 15, kg = lane >> 4) simply takes k = k0 + 4 kg + s in sub-step s: for a k-contiguous operand that is one
// 16-byte load of its own row, for a row-contiguous one four coalesced 4-byte loads.  No LDS or barrier inside the K loop;
// the loads of the next PD steps are in flight while a step computes.
// Every load is unconditional (k clamped into the row) and the zeroing of k >= K happens when a fragment is consumed: a
// branch around a load would make the compiler wait for all loads in flight at the join and serialise the pipeline.
template <bool VEC>
__device__ __forceinline__ f4 small_frag(const float* rowp, long sk, int K, int k) {
  if constexpr (VEC) {                                           // k-contiguous, 16-byte aligned rows, K % 4 == 0
    return *reinterpret_cast<const f4*>(rowp + (k < K ? k : 0));
  } else {
    f4 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = rowp[(long)min(k + s, K - 1) * sk];
    return v;
  }
}

__device__ __forceinline__ f4 zero_past(f4 v, int k, int K) {
#pragma unroll
  for (int s = 0; s < 4; ++s) v[s] = k + s < K ? v[s] : 0.f;
  return v;
}

template <bool AV, bool BV>
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(G32 p) {
  __shared__ f4 red[4][2][64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntn = (p.N + 31) / 32;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
  const int m0 = tm * 16, n0 = tn * 32;
  const int r = lane & 15, kg = lane >> 4;
  const float* pa = p.A + (long)min(m0 + r, p.M - 1) * p.sam;
  const float* pb0 = p.B + (long)min(n0 + r, p.N - 1) * p.sbn;
  const float* pb1 = p.B + (long)min(n0 + 16 + r, p.N - 1) * p.sbn;
  f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
  constexpr int PD = 4;
  const int nkt = (p.K + 15) / 16;
  const int nj = nkt > w ? (nkt - w + 3) / 4 : 0;               // this wave's steps: kt = w + 4 j
  f4 fa[PD], fb0[PD], fb1[PD];
  auto fetch = [&](int j, f4& a, f4& b0, f4& b1) {
    const int k = (w + 4 * j) * 16 + 4 * kg;
    a = small_frag<AV>(pa, p.sak, p.K, k);
    b0 = small_frag<BV>(pb0, p.sbk, p.K, k);
    b1 = small_frag<BV>(pb1, p.sbk, p.K, k);
  };
#pragma unroll
  for (int s = 0; s < PD; ++s) fetch(s, fa[s], fb0[s], fb1[s]);
  auto step = [&](int j, f4& a, f4& b0, f4& b1) {
    const int k = (w + 4 * j) * 16 + 4 * kg;
    const f4 ca = zero_past(a, k, p.K), cb0 = zero_past(b0, k, p.K), cb1 = zero_past(b1, k, p.K);
    fetch(j + PD, a, b0, b1);
    __builtin_amdgcn_sched_barrier(0);     // keep the refill ahead of this step's MFMAs (the scheduler would sink it to the loop end)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cb0[s], ca[s], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cb1[s], ca[s], acc[1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int j = 0; j < nj; j += PD) {       // no branches inside: steps past nj multiply zeros (k >= K), and the counted waits stay exact
    step(j, fa[0], fb0[0], fb1[0]);
    step(j + 1, fa[1], fb0[1], fb1[1]);
    step(j + 2, fa[2], fb0[2], fb1[2]);
    step(j + 3, fa[3], fb0[3], fb1[3]);
  }
  red[w][0][lane] = acc[0];
  red[w][1][lane] = acc[1];
  __syncthreads();
  if (w < 2) {
    const f4 t = (red[0][w][lane] + red[1][w][lane]) + (red[2][w][lane] + red[3][w][lane]);
    store_tile16(p, t, m0 + r, n0 + 16 * w + 4 * kg);
  }
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(G32 p) {
  __shared__ __attribute__((aligned(16))) float sA[2][TK * LDS_LD];
  __shared__ __attribute__((aligned(16))) float sB[2][TK * LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int ntn = (p.N + TN - 1) / TN;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
  const int m0 = tm * TM, n0 = tn * TN;
  const bool ak = p.sak == 1, bk = p.sbk == 1;

  f4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  // Register-staged pipeline, PD K-steps deep: the global loads of step kt + PD are issued before step kt is computed and
  // reach LDS one step before they are needed, so a lone workgroup on a CU (small M, or K-long weight gradients with
  // few output tiles) still has PD loads in flight instead of waiting out one full memory latency per 16-deep K-step.
  constexpr int PD = 3;
  const int nkt = (p.K + TK - 1) / TK;
  const int nfull = p.K / TK;                     // K-steps that lie fully inside K: no k bounds checks
  const OpLoader LA = make_loader(p.A, p.sam, p.sak, p.M, m0, tid, p.avec);
  const OpLoader LB = make_loader(p.B, p.sbn, p.sbk, p.N, n0, tid, p.bvec);
  auto fetch_a = [&](int kt) { return kt < nfull ? load_inner(LA, p.sak, p.M, m0, kt, tid)
                                                 : load_op(p.A, p.sam, p.sak, p.M, p.K, m0, kt * TK, tid, p.avec); };   // past K: zeros
  auto fetch_b = [&](int kt) { return kt < nfull ? load_inner(LB, p.sbk, p.N, n0, kt, tid)
                                                 : load_op(p.B, p.sbn, p.sbk, p.N, p.K, n0, kt * TK, tid, p.bvec); };
  f4 ra[PD], rb[PD];
#pragma unroll
  for (int s = 0; s < PD; ++s) { ra[s] = fetch_a(s); rb[s] = fetch_b(s); }
  store_op(sA[0], ra[0], ak, tid);
  store_op(sB[0], rb[0], bk, tid);
  __syncthreads();
  auto step = [&](int kt, f4& ra_next, f4& rb_next, f4& ra_slot, f4& rb_slot) {
    // ra_slot held step kt (already in LDS): refill it with step kt + PD; ra_next holds step kt + 1
    const int cur = kt & 1;
    ra_slot = fetch_a(kt + PD);
    rb_slot = fetch_b(kt + PD);
    const float* a = sA[cur];
    const float* b = sB[cur];
#pragma unroll
    for (int kk = 0; kk < TK; kk += 4) {
      float af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = a[(kk + (lane >> 4)) * LDS_LD + wm * 32 + i * 16 + (lane & 15)];
        bf[i] = b[(kk + (lane >> 4)) * LDS_LD + wn * 32 + i * 16 + (lane & 15)];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) {
      store_op(sA[cur ^ 1], ra_next, ak, tid);
      store_op(sB[cur ^ 1], rb_next, bk, tid);
    }
    __syncthreads();
  };
  for (int kt = 0; kt < nkt; kt += PD) {          // slots rotate with period PD: static register indices
    step(kt, ra[1], rb[1], ra[0], rb[0]);
    if (kt + 1 < nkt) step(kt + 1, ra[2], rb[2], ra[1], rb[1]);
    if (kt + 2 < nkt) step(kt + 2, ra[0], rb[0], ra[2], rb[2]);
  }
  // lane owns C[m = .. + (lane & 15)][n = .. + 4*(lane >> 4) + r]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      store_tile16(p, acc[i][j], m0 + wm * 32 + i * 16 + (lane & 15), n0 + wn * 32 + j * 16 + 4 * (lane >> 4));
}

}  // namespace

extern "C" int hmmc_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak, long sbk,
                             long sbn, int ldc, float alpha, const float* bias, const float* resid, float* aux_out,
                             const float* aux_in, int epilogue, hipStream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return HMMC_ERR_ARG;
  if ((sak != 1 && sam != 1) || (sbk != 1 && sbn != 1)) return HMMC_ERR_UNSUPPORTED;
  long a_ld = sak == 1 ? sam : sak, b_ld = sbk == 1 ? sbn : sbk;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 3) return HMMC_ERR_UNSUPPORTED;
  if ((epilogue & EPI_BIAS) && !bias) return HMMC_ERR_ARG;
  if ((epilogue & EPI_RESID) && !resid) return HMMC_ERR_ARG;
  if ((epilogue & EPI_DGELU) && !aux_in) return HMMC_ERR_ARG;
  G32 p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.resid = resid; p.aux_out = aux_out; p.aux_in = aux_in;
  p.M = M; p.N = N; p.K = K; p.sam = sam; p.sak = sak; p.sbk = sbk; p.sbn = sbn; p.ldc = ldc; p.alpha = alpha;
  p.flags = epilogue;
  // 16-byte accesses only where the leading dimension and base keep every row aligned
  p.avec = !(a_ld & 3) && !((uintptr_t)A & 15);
  p.bvec = !(b_ld & 3) && !((uintptr_t)B & 15);
  p.cvec = !(ldc & 3) && !((uintptr_t)C & 15) && (!aux_out || !((uintptr_t)aux_out & 15));
  long blocks = (long)((M + TM - 1) / TM) * ((N + TN - 1) / TN);
  static const int num_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  // fewer 64x64 tiles than CUs: latency-bound, take the split-K 16x32 kernel.  With both operands row-contiguous (weight
  // gradients) its fragments cost four 4-byte loads each, and it only wins below half a wave of tiles (measured: 512x512
  // outputs 104 -> 33 us at K = 3072, 17 -> 9 us at K = 384; 1536x512 outputs 18 -> 21 us at K = 384)
  if (blocks < num_cu && (sak == 1 || sbk == 1 || blocks * 2 <= num_cu)) {
    long small = (long)((M + 15) / 16) * ((N + 31) / 32);
    const bool av = sak == 1 && p.avec && !(K & 3), bv = sbk == 1 && p.bvec && !(K & 3);
    auto k = av ? (bv ? gemm_f32_small_kernel<true, true> : gemm_f32_small_kernel<true, false>)
                : (bv ? gemm_f32_small_kernel<false, true> : gemm_f32_small_kernel<false, false>);
    hipLaunchKernelGGL(k, dim3((unsigned)small), dim3(256), 0, stream, p);
    return hmmc_launch_status();
  }
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
  return hmmc_launch_status();
}
