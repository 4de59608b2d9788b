// burst store ubench: every workgroup (8 waves) writes a 256x256 fp16 tile in the GEMM epilogue's access shape
// (per wave 128x64 outputs = 16 stores of 8 rows x 128 B), then idles `gap` us; reports issue time and ack time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(char* out, size_t ld, int iters, int ntn, int gap_ticks, unsigned* stamps, int stagger_ticks, int mfma_iters, int pat) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 2, wn = wid & 3;
  uint4v v = {(unsigned)lane, 1u, 2u, 3u};
  if (stagger_ticks) {
    unsigned long long t = __builtin_amdgcn_s_memrealtime();
    unsigned long long until = t + (unsigned long long)((blockIdx.x * 2654435761u >> 8) % (unsigned)stagger_ticks);
    while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
  }
  for (int it = 0; it < iters; ++it) {
    int tile = blockIdx.x + it * gridDim.x;
    int tm = tile / ntn, tn = tile % ntn;
    char* base = out + ((size_t)(tm * 256 + wm * 128) * ld) + (size_t)(tn * 256 + wn * 64) * 2;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        int row, chunk;
        const int c = lane & 15, g = lane >> 4;
        if (pat == 0) { row = lane >> 3; chunk = lane & 7; }                                   // ideal: 8 consecutive lanes = one 128-B row
        else if (pat == 1) { row = c & 7; chunk = 4 * (c >> 3) + 2 * (g & 1) + (g >> 1); }     // the GEMM epilogue's lane order
        else if (pat == 2) { row = (lane >> 1) & 7; chunk = (lane & 1) + 2 * (lane >> 4); }    // 2 consecutive lanes = 32 B
        else { row = (lane >> 2) & 7; chunk = (lane & 3) + 4 * (lane >> 5); }                  // 4 consecutive lanes = 64 B
        char* p = base + (size_t)(i * 16 + t * 8 + row) * ld + chunk * 16;
        __builtin_nontemporal_store(v, (uint4v*)p);
      }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { unsigned* s = stamps + ((size_t)(blockIdx.x * 8 + wid) * iters + it) * 2; s[0] = (unsigned)(t1 - t0); s[1] = (unsigned)(t2 - t0); }
    if (mfma_iters) {
      h8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane ^ e)); }
      f4 c[16];
      for (int e = 0; e < 16; ++e) c[e] = f4{0.f, 0.f, 0.f, 0.f};
      for (int m = 0; m < mfma_iters; ++m) {
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[e], 0, 0, 0);
      }
      float acc = 0.f;
      for (int e = 0; e < 16; ++e) acc += c[e][0];
      if (acc == 12345.678f) v[1] = 7u;      // keep the MFMAs live
    } else {
      unsigned long long until = t2 + gap_ticks;
      while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
    }
  }
}
int main(int argc, char** argv) {
  const int iters = 12, ntn = 12;
  size_t ld = 6144;
  size_t bytes = (size_t)((256 * iters + ntn - 1) / ntn + 1) * 256 * ld;
  char* d; if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc fail\n"); return 1; }
  hipMemset(d, 0, bytes);
  unsigned* st; hipMalloc(&st, 256 * 8 * iters * 2 * 4);
  for (int pat : {0, 1, 2, 3}) for (int mf : {0}) for (int stag : {0}) for (int grid : {8, 256}) {
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, d, ld, iters, ntn, 1500, st, stag, mf, pat);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, d, ld, iters, ntn, 1500, st, stag, mf, pat);
    hipDeviceSynchronize();
    std::vector<unsigned> h(grid * 8 * iters * 2);
    hipMemcpy(h.data(), st, h.size() * 4, hipMemcpyDeviceToHost);
    std::vector<double> iss, ack;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < 8; ++w) for (int it = 2; it < iters; ++it) {
      size_t o = ((size_t)(b * 8 + w) * iters + it) * 2; iss.push_back(h[o] / 100.0); ack.push_back(h[o + 1] / 100.0);
    }
    std::sort(iss.begin(), iss.end()); std::sort(ack.begin(), ack.end());
    printf("pat %d mfma %2d stagger %4d grid %3d: issue us median %.2f p90 %.2f max %.2f | ack us median %.2f p90 %.2f max %.2f\n", pat, mf, stag, grid,
           iss[iss.size() / 2], iss[iss.size() * 9 / 10], iss.back(), ack[ack.size() / 2], ack[ack.size() * 9 / 10], ack.back());
  }
  return 0;
}
