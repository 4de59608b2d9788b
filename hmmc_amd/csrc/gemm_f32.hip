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
#include "options.h"
#include <cstdlib>

namespace {

constexpr int TM = 64, TN = 64, TK = 16;
constexpr int LDS_LD = 80;   // floats per k-row: 64 + 16 keeps the two k-rows of a 32-lane half on disjoint banks

enum { EPI_BIAS = 1, EPI_RESID = 2, EPI_QGELU = 4, EPI_DGELU = 8, EPI_RELU = 16 };

struct G32 {
  const float* A; const float* B; float* C; const float* bias; const float* resid; float* aux_out; const float* aux_in;
  int M, N, K; long sam, sak, sbk, sbn; int ldc; float alpha; int flags; int avec, bvec, cvec;
  // eval scorer (TOPK kernels): B rows are videos x P slots (slot 0 = video embedding, 1..F = frames, rest zero padding)
  int tk_F, tk_P, tk_k, tk_nv; float* tk_video; float* tk_frame; float* tk_score;
};

__device__ __forceinline__ float qgelu32(float h) { return h / (1.0f + __expf(-1.702f * h)); }

// ---- operand staging of the 64x64x16 kernel ---------------------------------------------------------------------------
// A thread owns 4 elements of a 64 x 16 operand tile per K-step.  MODE 0: k-contiguous rows read with one 16-byte load
// (thread -> row tid/4, k 4*(tid%4)..+3; needs 16-byte aligned rows and K % 4 == 0); MODE 1: row-contiguous read with one
// 16-byte load (thread -> k tid/16, rows 4*(tid%16)..+3; needs R % 4 == 0); MODE 2: any unit-stride orientation, four 4-byte
// loads.  Every load is unconditional with its row and k clamped into the operand: rows past R feed output rows that are
// never stored, and k >= K is zeroed when the registers are written to LDS - a branch around a load makes the compiler
// drain every load in flight at the join, which serialises the prefetch pipeline.
enum { OP_KVEC = 0, OP_RVEC = 1, OP_SCALAR = 2 };

template <int MODE>
__device__ __forceinline__ f4 fetch_op(const float* P, long sr, long sk, int R, int K, int r0, int k0, int tid) {
  if constexpr (MODE == OP_KVEC) {
    const int r = min(r0 + (tid >> 2), R - 1), k = k0 + (tid & 3) * 4;
    return *reinterpret_cast<const f4*>(P + (long)r * sr + (k < K ? k : 0));
  } else if constexpr (MODE == OP_RVEC) {
    const int k = min(k0 + (tid >> 4), K - 1), r = r0 + (tid & 15) * 4;
    return *reinterpret_cast<const f4*>(P + (long)k * sk + (r < R ? r : 0));
  } else {
    const bool kc = sk == 1;
    f4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = kc ? r0 + (tid >> 2) : r0 + (tid & 15) * 4 + j;
      const int k = kc ? k0 + (tid & 3) * 4 + j : k0 + (tid >> 4);
      v[j] = P[(long)min(r, R - 1) * sr + (long)min(k, K - 1) * sk];
    }
    return v;
  }
}

// write them into the LDS image [k][row], zeroing k >= K
template <int MODE>
__device__ __forceinline__ void store_op(float* S, f4 v, long sk, int K, int k0, int tid) {
  const bool kc = MODE == OP_KVEC || (MODE == OP_SCALAR && sk == 1);
  if (kc) {
    const int r = tid >> 2, k = (tid & 3) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) S[(k + j) * LDS_LD + r] = k0 + k + j < K ? v[j] : 0.f;
  } else {
    const int k = tid >> 4, r = (tid & 15) * 4;
    const bool in = k0 + k < K;
    *reinterpret_cast<f4*>(S + k * LDS_LD + r) = f4{in ? v[0] : 0.f, in ? v[1] : 0.f, in ? v[2] : 0.f, in ? v[3] : 0.f};
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

// ---- the same scheme on larger tiles: (16 RM) x (16 RN) outputs per workgroup, K split over its four waves ------------------
// The 16 x 32 tile above re-reads 0.19 B of operand per flop from L2, which caps it at a third of the matrix rate; a 64 x 64
// tile needs 0.06 B/flop, and one wave per SIMD issuing 16 independent 16x16x4 MFMAs per sub-step keeps the matrix pipe full
// without LDS staging or barriers in the K loop.  Used where the 64x64 LDS kernel is bound by the length of K (a workgroup
// per CU or fewer: the temporal transformer, the MoCo query gradient) - each wave's dependent chain is K/4 long here.
// The four partial tiles meet in LDS; wave w finishes the tiles t with t % 4 == w, summed in a fixed order.
template <bool AV, bool BV, int RM, int RN>
__global__ __launch_bounds__(256) void gemm_f32_wavek_kernel(G32 p) {
  extern __shared__ __attribute__((aligned(16))) char wk_smem[];
  f4 (*red)[RM * RN][64] = reinterpret_cast<f4 (*)[RM * RN][64]>(wk_smem);          // [wave][tile][lane]
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntn = (p.N + 16 * RN - 1) / (16 * RN);
  const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
  const int m0 = tm * 16 * RM, n0 = tn * 16 * RN;
  const int r = lane & 15, kg = lane >> 4;
  const float* pa[RM];
  const float* pb[RN];
#pragma unroll
  for (int i = 0; i < RM; ++i) pa[i] = p.A + (long)min(m0 + 16 * i + r, p.M - 1) * p.sam;
#pragma unroll
  for (int j = 0; j < RN; ++j) pb[j] = p.B + (long)min(n0 + 16 * j + r, p.N - 1) * p.sbn;
  f4 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  constexpr int PD = RM + RN <= 4 ? 4 : RM + RN <= 8 ? 3 : 2;    // steps of operand fragments in flight (RM + RN registers x 4 each)
  const int nkt = (p.K + 15) / 16;
  const int nj = nkt > w ? (nkt - w + 3) / 4 : 0;                // this wave's steps: kt = w + 4 j
  f4 fa[PD][RM], fb[PD][RN];
  auto fetch = [&](int j, f4 (&a)[RM], f4 (&b)[RN]) {
    const int k = (w + 4 * j) * 16 + 4 * kg;
#pragma unroll
    for (int i = 0; i < RM; ++i) a[i] = small_frag<AV>(pa[i], p.sak, p.K, k);
#pragma unroll
    for (int jn = 0; jn < RN; ++jn) b[jn] = small_frag<BV>(pb[jn], p.sbk, p.K, k);
  };
#pragma unroll
  for (int s = 0; s < PD; ++s) fetch(s, fa[s], fb[s]);
  auto step = [&](int j, f4 (&a)[RM], f4 (&b)[RN]) {
    const int k = (w + 4 * j) * 16 + 4 * kg;
    f4 ca[RM], cb[RN];
#pragma unroll
    for (int i = 0; i < RM; ++i) ca[i] = zero_past(a[i], k, p.K);
#pragma unroll
    for (int jn = 0; jn < RN; ++jn) cb[jn] = zero_past(b[jn], k, p.K);
    fetch(j + PD, a, b);
    __builtin_amdgcn_sched_barrier(0);     // keep the refill ahead of this step's MFMAs
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int jn = 0; jn < RN; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(cb[jn][s], ca[i][s], acc[i][jn], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int j = 0; j < nj; j += PD) {       // no branches inside: steps past nj multiply zeros (k >= K)
#pragma unroll
    for (int s = 0; s < PD; ++s) step(j + s, fa[s], fb[s]);
  }
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int jn = 0; jn < RN; ++jn) red[w][i * RN + jn][lane] = acc[i][jn];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int jn = 0; jn < RN; ++jn) {
      const int t = i * RN + jn;
      if ((t & 3) != w) continue;                                 // wave-uniform
      const f4 v = (red[0][t][lane] + red[1][t][lane]) + (red[2][t][lane] + red[3][t][lane]);
      store_tile16(p, v, m0 + 16 * i + r, n0 + 16 * jn + 4 * kg);
    }
}

// ---- eval scorer epilogue: video logit + mean of the top-k frame logits, straight from the accumulators -----------------
// (main_task_retrieval.py:332-336: loose_similarity(query, visual), topk(loose_similarity(query, frames), k, dim=2).mean(2);
// the [queries, videos, F] tensor of the reference is never written.)  The NV values of one (query, video) pair sit in
// the 4 lanes with the same lane & 15: v[e] has slot slot0[e / 4] + 4 (lane >> 4) + (e & 3).  k rounds of: best
// (value, lowest slot on ties, as torch.topk counts equal values separately) in the lane, two xor-shuffles (16, 32) across
// the four lanes, the owner retires the winner.  Returns the video logit (slot 0) and the top-k mean, on all four lanes.
template <int NV>
__device__ __forceinline__ void topk_video(float (&v)[NV], int lane, int F, int k, float& s_video, float& s_frames) {
  const int g = lane >> 4;
  int slot[NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) slot[e] = 16 * (e >> 2) + 4 * g + (e & 3);
  float own0 = (g == 0) ? v[0] : 0.f;                                   // slot 0 lives in lane group 0
  own0 += __shfl_xor(own0, 16, 64);
  own0 += __shfl_xor(own0, 32, 64);
  s_video = own0;
#pragma unroll
  for (int e = 0; e < NV; ++e)
    if (slot[e] == 0 || slot[e] > F) v[e] = -INFINITY;
  float sum = 0.f;
  for (int t = 0; t < k; ++t) {
    float bv = -INFINITY;
    int bs = 1 << 20;
#pragma unroll
    for (int e = 0; e < NV; ++e)
      if (v[e] > bv || (v[e] == bv && slot[e] < bs)) { bv = v[e]; bs = slot[e]; }
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int os = __shfl_xor(bs, off, 64);
      if (ov > bv || (ov == bv && os < bs)) { bv = ov; bs = os; }
    }
    sum += bv;
#pragma unroll
    for (int e = 0; e < NV; ++e)
      if (slot[e] == bs) v[e] = -INFINITY;
  }
  s_frames = sum / (float)k;
}

__device__ __forceinline__ void topk_emit(const G32& p, int m, int video, int lane, float s_video, float s_frames) {
  if ((lane >> 4) != 0 || m >= p.M || video >= p.tk_nv) return;
  const long o = (long)m * p.tk_nv + video;
  if (p.tk_video) p.tk_video[o] = s_video;
  if (p.tk_frame) p.tk_frame[o] = s_frames;
  if (p.tk_score) p.tk_score[o] = s_video + s_frames;
}

template <int AMODE, int BMODE, bool TOPK = false>
__global__ __launch_bounds__(256) void gemm_f32_kernel(G32 p) {
  __shared__ __attribute__((aligned(16))) float sA[2][TK * LDS_LD];
  __shared__ __attribute__((aligned(16))) float sB[2][TK * LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int ntn = (p.N + TN - 1) / TN;
  int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
  if constexpr (TOPK) {
    // 15 000 x 48 000 outputs: walk the tiles in groups of 16 row tiles so that the ~256 workgroups resident together
    // share 16 query tiles and 16 video tiles (4 MiB of operands in L2) instead of 1 and 256 (33 MiB)
    constexpr int GROUP = 16;
    const int ntm = (p.M + TM - 1) / TM;
    const int grp = blockIdx.x / (GROUP * ntn), within = blockIdx.x % (GROUP * ntn);
    const int gm = min(GROUP, ntm - grp * GROUP);
    tm = grp * GROUP + within % gm;
    tn = within / gm;
  }
  const int m0 = tm * TM, n0 = tn * TN;

  f4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  // Two pipelines.  Global -> registers runs PD K-steps ahead (a step is 0.2 us of MFMA, a miss to HBM 1.5 us), so a lone
  // workgroup on a CU keeps PD loads in flight.  LDS -> fragment registers runs one step ahead: the 16 fragment words of
  // step kt + 1 are read while the MFMAs of step kt execute, so neither the LDS latency nor the staging stores sit between
  // two steps' MFMAs; one barrier per step.  The loop body has no branches: the steps of the last round that lie past K
  // multiply zeros (store_op zeroes k >= K).
  constexpr int PD = 8;
  const int nkt = (p.K + TK - 1) / TK;
  f4 ra[PD], rb[PD];
#pragma unroll
  for (int s = 0; s < PD; ++s) {
    ra[s] = fetch_op<AMODE>(p.A, p.sam, p.sak, p.M, p.K, m0, s * TK, tid);
    rb[s] = fetch_op<BMODE>(p.B, p.sbn, p.sbk, p.N, p.K, n0, s * TK, tid);
  }
  struct Frags { float a[TK / 4][2], b[TK / 4][2]; };
  auto read_frags = [&](int buf, Frags& f) {
    const float* a = sA[buf];
    const float* b = sB[buf];
#pragma unroll
    for (int kk = 0; kk < TK / 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f.a[kk][i] = a[(4 * kk + (lane >> 4)) * LDS_LD + wm * 32 + i * 16 + (lane & 15)];
        f.b[kk][i] = b[(4 * kk + (lane >> 4)) * LDS_LD + wn * 32 + i * 16 + (lane & 15)];
      }
  };
  Frags f0, f1;
  store_op<AMODE>(sA[0], ra[0], p.sak, p.K, 0, tid);
  store_op<BMODE>(sB[0], rb[0], p.sbk, p.K, 0, tid);
  store_op<AMODE>(sA[1], ra[1], p.sak, p.K, TK, tid);
  store_op<BMODE>(sB[1], rb[1], p.sbk, p.K, TK, tid);
  ra[0] = fetch_op<AMODE>(p.A, p.sam, p.sak, p.M, p.K, m0, PD * TK, tid);
  rb[0] = fetch_op<BMODE>(p.B, p.sbn, p.sbk, p.N, p.K, n0, PD * TK, tid);
  ra[1] = fetch_op<AMODE>(p.A, p.sam, p.sak, p.M, p.K, m0, (PD + 1) * TK, tid);
  rb[1] = fetch_op<BMODE>(p.B, p.sbn, p.sbk, p.N, p.K, n0, (PD + 1) * TK, tid);
  __syncthreads();
  read_frags(0, f0);
  __syncthreads();                                 // step 0 overwrites buffer 0: every wave must hold its fragments of step 0 first
  // step kt: fragments of kt are in `cur`; read those of kt + 1 from buffer (kt + 1) & 1, multiply, then overwrite buffer kt & 1
  // (every wave has its fragments of step kt in registers since the barrier that ended step kt - 1) with step kt + 2 from
  // register slot (kt + 2) % PD and refill that slot with step kt + 2 + PD
  // A wave issues in order, so whatever sits between the last MFMA of one step and the first of the next is exposed: the
  // LDS reads, the staging stores (which wait for the global data) and the refill are therefore placed between the four
  // 4-MFMA groups of the step, where they issue in the shadow of the 32-cycle MFMAs.
  auto mfma2 = [&](const Frags& f, int kk, int i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[kk][j], f.a[kk][i], acc[i][j], 0, 0, 0);
  };
  auto read_part = [&](int buf, Frags& f, int kk) {
    const float* a = sA[buf];
    const float* b = sB[buf];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f.a[kk][i] = a[(4 * kk + (lane >> 4)) * LDS_LD + wm * 32 + i * 16 + (lane & 15)];
      f.b[kk][i] = b[(4 * kk + (lane >> 4)) * LDS_LD + wn * 32 + i * 16 + (lane & 15)];
    }
  };
#define HMMC_PIN() __builtin_amdgcn_sched_barrier(0)
  auto step = [&](int kt, const Frags& cur, Frags& nxt, f4& sa, f4& sb) {
    // eight pairs of MFMAs (64 cycles of the matrix pipe each); between them, in order: the four fragment-read pairs of step
    // kt + 1, the two staging stores of step kt + 2, the two refills
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      mfma2(cur, kk, 0); read_part((kt + 1) & 1, nxt, 2 * kk); HMMC_PIN();
      mfma2(cur, kk, 1); read_part((kt + 1) & 1, nxt, 2 * kk + 1); HMMC_PIN();
    }
    mfma2(cur, 2, 0); store_op<AMODE>(sA[kt & 1], sa, p.sak, p.K, (kt + 2) * TK, tid); HMMC_PIN();
    mfma2(cur, 2, 1); store_op<BMODE>(sB[kt & 1], sb, p.sbk, p.K, (kt + 2) * TK, tid); HMMC_PIN();
    mfma2(cur, 3, 0);
    sa = fetch_op<AMODE>(p.A, p.sam, p.sak, p.M, p.K, m0, (kt + 2 + PD) * TK, tid);
    sb = fetch_op<BMODE>(p.B, p.sbn, p.sbk, p.N, p.K, n0, (kt + 2 + PD) * TK, tid);
    HMMC_PIN();
    mfma2(cur, 3, 1);
    __syncthreads();
  };
#undef HMMC_PIN
  for (int kt = 0; kt < nkt; kt += PD) {          // slots and buffers rotate with period PD (even): static register indices
#pragma unroll
    for (int s = 0; s < PD; s += 2) {
      step(kt + s, f0, f1, ra[(s + 2) % PD], rb[(s + 2) % PD]);
      step(kt + s + 1, f1, f0, ra[(s + 3) % PD], rb[(s + 3) % PD]);
    }
  }
  // lane owns C[m = .. + (lane & 15)][n = .. + 4*(lane >> 4) + r]
  if constexpr (TOPK) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wm * 32 + i * 16 + (lane & 15);
      float sv, sf;
      if (p.tk_P == 16) {                          // one video per 16-column MFMA tile
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v[4] = {acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha};
          topk_video<4>(v, lane, p.tk_F, p.tk_k, sv, sf);
          topk_emit(p, m, (n0 + wn * 32 + j * 16) >> 4, lane, sv, sf);
        }
      } else {                                     // P == 32: the wave's two tiles are one video
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[i][e >> 2][e & 3] * p.alpha;
        topk_video<8>(v, lane, p.tk_F, p.tk_k, sv, sf);
        topk_emit(p, m, (n0 + wn * 32) >> 5, lane, sv, sf);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      store_tile16(p, acc[i][j], m0 + wm * 32 + i * 16 + (lane & 15), n0 + wn * 32 + j * 16 + 4 * (lane >> 4));
}

// ---- large aligned problems: (32 MI) x 64 tiles staged by LDS-DMA -----------------------------------------------------------
// The register-staged kernel above pays, per 16-deep K-step of 16 MFMAs per wave, 16 four-byte LDS reads, the staging stores,
// two global loads and a barrier, and sits at 55-60 % of the f32 matrix rate on the temporal transformer at 3 072 tokens and
// the pre-training heads.  Here the operand tiles go global -> LDS without touching registers (buffer_load ... lds, 16 B per
// lane, bounds-checked: rows past the end of an operand read as zero), a K-step is 32 deep (32 MFMAs per wave and barrier), and
// NST - 1 steps are in flight behind a counted vmcnt.  Images (a 16-byte chunk = 4 floats):
//   k-contiguous operand:   [rows][8 chunks]   phys chunk = logical ^ ((row >> 1) & 7)           read: one ds_read_b128 per 16 k
//   row-contiguous operand: [32 k][16 chunks]  phys chunk = logical ^ (((k >> 2) & 3) << 2)      read: four ds_read_b32 per 16 k
// (LDS side of a DMA is lane-linear, so the XOR is applied to the SOURCE chunk a lane fetches).  As in the kernels above the
// MFMA's k slot kg of sub-step s is k = 16 kb + 4 kg + s for both operands.
// Needs 16-byte aligned rows on both operands and K % 32 == 0 (the k tail of a k-contiguous row would read its neighbour).
template <bool AKC, bool BKC, int MI, bool TOPK = false>
__device__ __forceinline__ void gemm_f32_dma_body(const G32& p, unsigned a_bytes, unsigned b_bytes, int a_fast) {
  constexpr int BM = 32 * MI, BN = 64, BK = 32, NST = 3;
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4, STAGE = A_BYTES + B_BYTES;
  constexpr int LPT = MI + 2;                                     // LDS-DMA loads per thread and stage
  constexpr int C_LD = 68;                                        // floats per row of the epilogue's LDS tile (272 B: 16-byte rows, +4 banks per row)
  static_assert(BM * C_LD * 4 <= NST * STAGE, "the result tile reuses the stages");
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  // tile order: the workgroups resident together walk the SMALLER operand's tiles fastest, so the larger operand's tile is
  // fetched once and shared (the MLM vocabulary projection: 3 MB of activations against 101 MB of weights)
  int tm = a_fast ? blockIdx.x % ntm : blockIdx.x / ntn, tn = a_fast ? blockIdx.x / ntm : blockIdx.x % ntn;
  if constexpr (TOPK) {
    // the eval scorer's 15 000 x 48 000 outputs: groups of 16 row tiles, as in gemm_f32_kernel<.., TOPK> (the workgroups resident
    // together share 16 query tiles and ~48 video tiles instead of one video tile and every query tile)
    constexpr int GROUP = 16;
    const int grp = blockIdx.x / (GROUP * ntn), within = blockIdx.x % (GROUP * ntn);
    const int gm = min(GROUP, ntm - grp * GROUP);
    tm = grp * GROUP + within % gm;
    tn = within / gm;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int r = lane & 15, kg = lane >> 4;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)b_bytes, 0x00020000);
  const unsigned lda = (unsigned)(AKC ? p.sam : p.sak), ldb = (unsigned)(BKC ? p.sbn : p.sbk);
  // byte offset of the chunk this thread fetches in pass ps of an operand tile at (row0, k0 = 0); the k advance is uniform
  auto src_off = [&](bool kc, int rows, int ps, int row0, unsigned ld) -> unsigned {
    const int P = ps * 256 + tid;
    if (kc) {
      const int row = P >> 3, lc = (P & 7) ^ ((row >> 1) & 7);
      return ((unsigned)(row0 + row) * ld + (unsigned)(lc * 4)) * 4u;
    }
    const int cpr = rows / 4, krow = P / cpr, pc = P % cpr;       // chunks per k-row: 8 / 16 / 32 (32 / 64 / 128 rows)
    const int lc = pc ^ ((((krow >> 2) & 3) << 2) & (cpr - 1));
    return ((unsigned)krow * ld + (unsigned)(row0 + lc * 4)) * 4u;
  };
  unsigned offa[MI], offb[2];
#pragma unroll
  for (int ps = 0; ps < MI; ++ps) offa[ps] = src_off(AKC, BM, ps, m0, lda);
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) offb[ps] = src_off(BKC, BN, ps, n0, ldb);
  const unsigned stepa = AKC ? BK * 4u : BK * lda * 4u, stepb = BKC ? BK * 4u : BK * ldb * 4u;
  const int nkt = p.K / BK;
  auto issue = [&](int kt) {                                      // past the last step: out of range, reads as zero into a dead stage
    char* st = dsm + (kt % NST) * STAGE;
    const bool live = kt < nkt;
#pragma unroll
    for (int ps = 0; ps < MI; ++ps)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(st + ps * 4096 + wid * 1024), 16, live ? offa[ps] + (unsigned)kt * stepa : 0xfffffff0u, 0, 0, 0);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(st + A_BYTES + ps * 4096 + wid * 1024), 16, live ? offb[ps] + (unsigned)kt * stepb : 0xfffffff0u, 0, 0, 0);
  };
  f4 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  // 4 sub-step values of MFMA tile `row0` of an image, k block kb
  auto frag = [&](bool kc, const char* img, int rows, int row0, int kb) -> f4 {
    const int row = row0 + r;
    if (kc) return *reinterpret_cast<const f4*>(img + row * 128 + (((kb * 4 + kg) ^ ((row >> 1) & 7)) << 4));
    f4 v;
    const char* base = img + (16 * kb + 4 * kg) * (rows * 4) + (((row >> 2) ^ ((kg << 2) & (rows / 4 - 1))) << 4) + (row & 3) * 4;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) v[s2] = *reinterpret_cast<const float*>(base + s2 * (rows * 4));
    return v;
  };
  static_assert(NST == 3, "the counted wait below leaves one stage in flight");
  issue(0); issue(1);
  for (int kt = 0; kt < nkt; ++kt) {
    // all but the newest stage have landed (this thread's part)
    if constexpr (LPT == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (LPT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    // ... everyone's part, and every wave's fragment reads of stage kt - 1 were consumed by its MFMAs.  The raw barrier:
    // __syncthreads() would wait for vmcnt(0) first, i.e. for the stage that is meant to stay in flight.
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue(kt + 2);
    const char* st = dsm + (kt % NST) * STAGE;
    f4 a[2][MI], b[2][2];                                         // both k blocks requested before the first MFMA
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < MI; ++i) a[kb][i] = frag(AKC, st, BM, wm * (16 * MI) + 16 * i, kb);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[kb][j] = frag(BKC, st + A_BYTES, BN, wn * 32 + 16 * j, kb);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[kb][j][s2], a[kb][i][s2], acc[i][j], 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the trailing out-of-range loads, before the stages are reused
  if constexpr (TOPK) {                                           // eval scorer: the epilogue of gemm_f32_kernel<.., TOPK>, same accumulator layout
    static_assert(!TOPK || MI == 2, "the scorer's epilogue walks 2 x 2 accumulator tiles per wave");
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * (16 * MI) + 16 * i + r;
      float sv, sf;
      if (p.tk_P == 16) {                                         // one video per 16-column MFMA tile
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v[4] = {acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha};
          topk_video<4>(v, lane, p.tk_F, p.tk_k, sv, sf);
          topk_emit(p, m, (n0 + wn * 32 + j * 16) >> 4, lane, sv, sf);
        }
      } else {                                                    // P == 32: the wave's two tiles are one video
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[i][e >> 2][e & 3] * p.alpha;
        topk_video<8>(v, lane, p.tk_F, p.tk_k, sv, sf);
        topk_emit(p, m, (n0 + wn * 32) >> 5, lane, sv, sf);
      }
    }
    return;
  }
  // Epilogue through LDS: from the MFMA layout (lane = one row, 16 bytes) to lane order (16 consecutive lanes = 256 consecutive
  // bytes of one row), so that the stores - and the residual / auxiliary loads inside store_tile16 - are whole 256-byte row
  // segments instead of 64-byte pieces of 16 different rows per instruction.
  __syncthreads();
  float* ct = reinterpret_cast<float*>(dsm);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *reinterpret_cast<f4*>(ct + (wm * (16 * MI) + 16 * i + r) * C_LD + wn * 32 + 16 * j + 4 * kg) = acc[i][j];
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < BM / 16; ++ps) {
    const int row = ps * 16 + (tid >> 4), col = 4 * (tid & 15);
    store_tile16(p, *reinterpret_cast<const f4*>(ct + row * C_LD + col), m0 + row, n0 + col);
  }
}
// (the body lives in a __device__ function: buffer-resource types in a __global__ template make hipcc 7.2 drop the host stub)
template <bool AKC, bool BKC, int MI>
__global__ __launch_bounds__(256) void gemm_f32_dma_kernel(G32 p, unsigned a_bytes, unsigned b_bytes, int a_fast) {
  gemm_f32_dma_body<AKC, BKC, MI>(p, a_bytes, b_bytes, a_fast);
}
__global__ __launch_bounds__(256) void gemm_f32_dma_topk_kernel(G32 p, unsigned a_bytes, unsigned b_bytes) {
  gemm_f32_dma_body<true, true, 2, true>(p, a_bytes, b_bytes, 0);
}

// packed[video * P + slot][:] = unit row: slot 0 = visual[video], 1..F = frames[video][slot - 1], slots past F zero
// (loose_similarity normalises both sides without an epsilon, modules/modeling.py:211-213)
__global__ __launch_bounds__(256) void eval_pack_kernel(const float* __restrict__ visual, const float* __restrict__ frames,
                                                        float* __restrict__ packed, int nv, int F, int E, int P) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // one wave per packed row
  if (row >= (long)nv * P) return;
  const int video = (int)(row / P), slot = (int)(row % P);
  float* dst = packed + row * E;
  if (slot > F) {
    for (int e = lane * 4; e < E; e += 256) *reinterpret_cast<f4*>(dst + e) = f4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  const float* src = slot == 0 ? visual + (long)video * E : frames + ((long)video * F + slot - 1) * E;
  float ss = 0.f;
  for (int e = lane * 4; e < E; e += 256) {
    const f4 x = *reinterpret_cast<const f4*>(src + e);
    ss += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
  }
  const float inv = 1.0f / sqrtf(wave_sum(ss));
  for (int e = lane * 4; e < E; e += 256) {
    const f4 x = *reinterpret_cast<const f4*>(src + e);
    *reinterpret_cast<f4*>(dst + e) = f4{x[0] * inv, x[1] * inv, x[2] * inv, x[3] * inv};
  }
}

template <int AMODE>
void launch_f32(int bmode, dim3 grid, hipStream_t stream, const G32& p) {
  switch (bmode) {
    case OP_KVEC: hipLaunchKernelGGL((gemm_f32_kernel<AMODE, OP_KVEC>), grid, dim3(256), 0, stream, p); break;
    case OP_RVEC: hipLaunchKernelGGL((gemm_f32_kernel<AMODE, OP_RVEC>), grid, dim3(256), 0, stream, p); break;
    default: hipLaunchKernelGGL((gemm_f32_kernel<AMODE, OP_SCALAR>), grid, dim3(256), 0, stream, p); break;
  }
}

}  // namespace

static int gemm_f32_impl(const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak, long sbk,
                         long sbn, int ldc, float alpha, const float* bias, const float* resid, float* aux_out,
                         const float* aux_in, int epilogue, hipStream_t stream);

extern "C" int hmmc_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak, long sbk,
                             long sbn, int ldc, float alpha, const float* bias, const float* resid, float* aux_out,
                             const float* aux_in, int epilogue, hipStream_t stream) {
  // bench.py's live timing (off unless hmmc_gemm_profile_start was called): slot 3 of hmmc_gemm_profile_stop
  const double mn = (double)M * N;
  const long tok = hmmc_prof_begin(2.0 * mn * K, 4.0 * ((double)M * K + (double)N * K + mn) + ((epilogue & EPI_BIAS) ? 4.0 * N : 0.0) +
                                   4.0 * mn * (((epilogue & EPI_RESID) ? 1 : 0) + ((epilogue & EPI_DGELU) ? 1 : 0) + (aux_out ? 1 : 0)), 3, stream);
  const int rc = gemm_f32_impl(A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, bias, resid, aux_out, aux_in, epilogue, stream);
  hmmc_prof_end(tok, stream);
  return rc;
}

static int gemm_f32_impl(const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak, long sbk,
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
  const int num_cu = hmmc_num_cus();
  // Two kernels.  The 64x64 LDS kernel runs one 16-deep K-step per ~0.4 us while a workgroup is alone on its CU (0.2 us of
  // MFMA plus the in-order issue of its staging and the barrier) and ~0.6x that per further co-resident workgroup, so a
  // problem with few tiles is bound by the length of K; the split-K 16x32 kernel has no such chain but re-reads its operands
  // from L2 for every 16x32 tile (0.19 B/flop; both operands k-contiguous cost more because a 16-byte load then touches 16
  // rows).  Estimated times in us (fitted to measurements at 96..3072 rows of the temporal-transformer shapes) pick one.
  const double mnk = (double)M * N * K;
  const double t_small = 3.0 + mnk * ((sak == 1 && sbk == 1) ? 6.0e-8 : 3.9e-8);
  // (many tiles per CU: the 64x64 kernel settles at one tile per CU per K/16 x 0.333 us - the MLM head's 72 tiles per CU)
  const double t_tiled_few = 3.0 + (K / 16.0) * 0.40 * (0.4 + 0.6 * (double)((blocks + num_cu - 1) / num_cu));
  const double t_tiled_many = 3.0 + (K / 16.0) * 0.333 * ((double)blocks / num_cu + 0.5);       // + half a round of imbalance
  const double t_tiled = t_tiled_few > t_tiled_many ? t_tiled_few : t_tiled_many;
  // The wave-split-K kernel on (16 RM) x 64 tiles, for problems it covers with at most one workgroup per CU (more than one per CU
  // share the matrix pipe at ~0.7 of its rate: measured).  Matrix-rate bound: 2 RM RN K cycles per tile; the smallest RM that
  // fits the CUs gives the shortest chain.  Measured against the other two kernels on every fp32 shape of the path
  // (scratch/gemm32_pick.py): ahead for K >= 1536 (data / weight gradients and c_proj of the temporal transformer, the MoCo
  // query gradient, the MLM data gradient), behind for K = 512, where the fixed ~6 us of fill and reduction decide.
  {
    const bool off = hmmc_option(HMMC_OPT_NO_F32_WAVEK);                          // A/B runs
#ifdef HMMC_SCRATCH      // scratch/gemm32_pick.py builds: HMMC_F32_PICK = 1 small, 2 tiled, 3.. wave-split-K RM = 2, 3, 4, 6,
    static const char* force_s = std::getenv("HMMC_F32_PICK");                    // 7 / 9 LDS-DMA 64x64 / 32x64
    const int force = force_s ? atoi(force_s) : 0;
#else
    const int force = 0;
#endif
    const bool av = sak == 1 && p.avec && !(K & 3), bv = sbk == 1 && p.bvec && !(K & 3);
    static const int rms[4] = {2, 3, 4, 6};
    double best = t_small < t_tiled ? t_small : t_tiled;
    int rm = 0;
    if (force >= 3 && force <= 6) rm = rms[force - 3];
    for (int c = 0; c < 4 && !off && !force; ++c) {
      const long tiles = (long)((M + 16 * rms[c] - 1) / (16 * rms[c])) * ((N + 63) / 64);
      if (tiles > num_cu) continue;
      const double t = 6.0 + (double)K * rms[c] * 4 * (2.0 / 1900.0) / 0.85;
      if (t < best) { best = t; rm = rms[c]; }
      break;                                                                      // larger tiles only lengthen the chain
    }
    // The LDS-DMA kernel (16-byte aligned rows on both sides, whole 32-deep K-steps, 32-bit byte offsets): (32 MI) x 64 tiles,
    // three (MI = 2) or four (MI = 1) workgroups per CU.  Measured on the temporal transformer at 3 072 tokens and the
    // pre-training heads (scratch/gemm32_dma.py): ~8-10 us of launch, fill and drain + rounds x the tile's time at 0.78 (64x64) /
    // 0.71 (32x64) of the matrix rate; ahead of the kernels above wherever there are several tiles per CU.
    {
      const uint64_t a_ext = (sak == 1 ? (uint64_t)(M - 1) * sam + K : (uint64_t)(K - 1) * sak + M) * 4;
      const uint64_t b_ext = (sbk == 1 ? (uint64_t)(N - 1) * sbn + K : (uint64_t)(K - 1) * sbk + N) * 4;
      const uint64_t a_reach = (sak == 1 ? (uint64_t)(M + 64) * sam : (uint64_t)K * sak + M + 64) * 4;
      const uint64_t b_reach = (sbk == 1 ? (uint64_t)(N + 64) * sbn : (uint64_t)K * sbk + N + 64) * 4;
      const bool dma_ok = p.avec && p.bvec && !(K & 31) && a_ext < (1ull << 31) && b_ext < (1ull << 31) && a_reach < 0xfffffff0ull &&
                          b_reach < 0xfffffff0ull;
      const bool dma_off = hmmc_option(HMMC_OPT_NO_F32_DMA);                      // A/B runs
      int mi = 0;
      if (dma_ok && (force == 7 || force == 9)) mi = force == 7 ? 2 : 1;
      for (int c = 1; c <= 2 && dma_ok && !dma_off && !force; ++c) {
        const long tiles = (long)((M + 32 * c - 1) / (32 * c)) * ((N + 63) / 64);
        const double tile_us = 32.0 * c * 64 * K * 2 / (256.0 * 2400.0);          // one tile at the CU's full f32 matrix rate
        if (tiles < 2 * num_cu) continue;                                         // alone on its CU a workgroup hides nothing: 0.6 of the rate
        const double t = 8.0 + (double)((tiles + num_cu - 1) / num_cu) * tile_us / (c == 2 ? 0.78 : 0.71);
        if (t < best) { best = t; mi = c; rm = 0; }
      }
      if (mi) {
        const long tiles = (long)((M + 32 * mi - 1) / (32 * mi)) * ((N + 63) / 64);
        const int smem = 3 * (32 * mi + 64) * 32 * 4;
        const int a_fast = (long)M < (long)N;                     // A is the smaller operand: walk its tiles fastest
#define HMMC_DMA1(AKC, BKC, MI_) hipLaunchKernelGGL((gemm_f32_dma_kernel<AKC, BKC, MI_>), dim3((unsigned)tiles), dim3(256), smem, stream, p, \
                                                    (unsigned)a_ext, (unsigned)b_ext, a_fast)
#define HMMC_DMA(AKC, BKC) do { if (mi == 1) HMMC_DMA1(AKC, BKC, 1); else HMMC_DMA1(AKC, BKC, 2); } while (0)
        const bool akc = sak == 1, bkc = sbk == 1;
        if (akc && bkc) HMMC_DMA(true, true); else if (akc) HMMC_DMA(true, false); else if (bkc) HMMC_DMA(false, true); else HMMC_DMA(false, false);
#undef HMMC_DMA
#undef HMMC_DMA1
        return hmmc_launch_status();
      }
    }
    if (rm) {
      const long tiles = (long)((M + 16 * rm - 1) / (16 * rm)) * ((N + 63) / 64);
      const int smem = 4 * rm * 4 * 1024;
#define HMMC_WK1(AVV, BVV, RMM) do { static bool done_[HMMC_MAX_DEVICES] = {false}; \
        if (smem > 64 * 1024) hmmc_allow_lds((const void*)gemm_f32_wavek_kernel<AVV, BVV, RMM, 4>, smem, done_); \
        hipLaunchKernelGGL((gemm_f32_wavek_kernel<AVV, BVV, RMM, 4>), dim3((unsigned)tiles), dim3(256), smem, stream, p); } while (0)
#define HMMC_WK(AVV, BVV) do { if (rm == 2) HMMC_WK1(AVV, BVV, 2); else if (rm == 3) HMMC_WK1(AVV, BVV, 3); \
                               else if (rm == 4) HMMC_WK1(AVV, BVV, 4); else HMMC_WK1(AVV, BVV, 6); } while (0)
      if (av && bv) HMMC_WK(true, true); else if (av) HMMC_WK(true, false); else if (bv) HMMC_WK(false, true); else HMMC_WK(false, false);
#undef HMMC_WK
#undef HMMC_WK1
      return hmmc_launch_status();
    }
  }
#ifdef HMMC_SCRATCH
  static const char* force2_s = std::getenv("HMMC_F32_PICK");
  const int force2 = force2_s ? atoi(force2_s) : 0;
#else
  const int force2 = 0;
#endif
  if (force2 == 1 || (force2 != 2 && t_small < t_tiled)) {
    long small = (long)((M + 15) / 16) * ((N + 31) / 32);
    const bool av = sak == 1 && p.avec && !(K & 3), bv = sbk == 1 && p.bvec && !(K & 3);
    auto k = av ? (bv ? gemm_f32_small_kernel<true, true> : gemm_f32_small_kernel<true, false>)
                : (bv ? gemm_f32_small_kernel<false, true> : gemm_f32_small_kernel<false, false>);
    hipLaunchKernelGGL(k, dim3((unsigned)small), dim3(256), 0, stream, p);
    return hmmc_launch_status();
  }
  const int amode = !p.avec ? OP_SCALAR : sak == 1 ? ((K & 3) ? OP_SCALAR : OP_KVEC) : ((M & 3) ? OP_SCALAR : OP_RVEC);
  const int bmode = !p.bvec ? OP_SCALAR : sbk == 1 ? ((K & 3) ? OP_SCALAR : OP_KVEC) : ((N & 3) ? OP_SCALAR : OP_RVEC);
  const dim3 grid((unsigned)blocks);
  switch (amode) {
    case OP_KVEC: launch_f32<OP_KVEC>(bmode, grid, stream, p); break;
    case OP_RVEC: launch_f32<OP_RVEC>(bmode, grid, stream, p); break;
    default: launch_f32<OP_SCALAR>(bmode, grid, stream, p); break;
  }
  return hmmc_launch_status();
}

// ---- eval scorer (reference main_task_retrieval.py:321-357 _run_on_single_gpu, modules/modeling.py:207-229) -----------------
extern "C" int hmmc_eval_slots(int F) { return F + 1 <= 16 ? 16 : F + 1 <= 32 ? 32 : 0; }

extern "C" int hmmc_eval_pack(const float* visual, const float* frames, float* packed, int nv, int F, int E,
                              hipStream_t stream) {
  const int P = hmmc_eval_slots(F);
  if (!visual || !frames || !packed || nv <= 0 || F <= 0) return HMMC_ERR_ARG;
  if (!P || (E & 3) || (((uintptr_t)visual | (uintptr_t)frames | (uintptr_t)packed) & 15)) return HMMC_ERR_UNSUPPORTED;
  const long rows = (long)nv * P;
  hipLaunchKernelGGL(eval_pack_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, visual, frames, packed, nv, F, E, P);
  return hmmc_launch_status();
}

extern "C" int hmmc_eval_score(const float* queries_unit, const float* packed, float* out_video, float* out_frame,
                               float* out_score, int nq, int nv, int F, int E, int k, float scale, hipStream_t stream) {
  const int P = hmmc_eval_slots(F);
  if (!queries_unit || !packed || nq <= 0 || nv <= 0 || F <= 0 || k <= 0 || k > F) return HMMC_ERR_ARG;
  if (!out_video && !out_frame && !out_score) return HMMC_ERR_ARG;
  if (!P || (E & 3) || (((uintptr_t)queries_unit | (uintptr_t)packed) & 15) || (long)nv * P >= (1l << 31)) return HMMC_ERR_UNSUPPORTED;
  G32 p{};
  p.A = queries_unit; p.B = packed; p.C = nullptr;
  p.M = nq; p.N = nv * P; p.K = E; p.sam = E; p.sak = 1; p.sbk = 1; p.sbn = E; p.ldc = 0; p.alpha = scale; p.flags = 0;
  p.avec = p.bvec = 1; p.cvec = 0;
  p.tk_F = F; p.tk_P = P; p.tk_k = k; p.tk_nv = nv; p.tk_video = out_video; p.tk_frame = out_frame; p.tk_score = out_score;
  const long blocks = (long)((nq + TM - 1) / TM) * ((p.N + TN - 1) / TN);
  // the LDS-DMA kernel (same 64x64 tile and accumulator layout: 115-120 against 85-93 TFLOP/s at the VATEX size) where its
  // conditions hold: whole 32-deep K-steps, 32-bit byte offsets
  const bool dma_off = hmmc_option(HMMC_OPT_NO_F32_DMA);
  const uint64_t a_ext = (uint64_t)nq * E * 4, b_ext = (uint64_t)p.N * E * 4;
  if (!dma_off && !(E & 31) && a_ext + 64ull * E * 4 < (1ull << 31) && b_ext + 64ull * E * 4 < (1ull << 31)) {
    hipLaunchKernelGGL(gemm_f32_dma_topk_kernel, dim3((unsigned)blocks), dim3(256), 3 * (64 + 64) * 32 * 4, stream, p, (unsigned)a_ext,
                       (unsigned)b_ext);
    return hmmc_launch_status();
  }
  hipLaunchKernelGGL((gemm_f32_kernel<OP_KVEC, OP_KVEC, true>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
  return hmmc_launch_status();
}
