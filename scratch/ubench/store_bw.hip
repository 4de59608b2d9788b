// per-CU global store throughput for the GEMM epilogue's access shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
// pattern 0: lane -> row = lane>>3 (8 rows), 128 B per row   (stride ld bytes between rows)
// pattern 1: lane -> row = lane&15 (16 rows), 64 B per row
// pattern 2: 1 KiB contiguous
// pattern 3: pattern 1 with nontemporal
// pattern 4: pattern 0 nontemporal
template <int PAT>
__global__ __launch_bounds__(256) void k(char* out, size_t ld, int iters, size_t wg_stride) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  char* base = out + (size_t)blockIdx.x * wg_stride + (size_t)wid * 128 ;   // wave owns a 128-byte column strip? (pattern 0/1: 64 cols fp16 = 128 B)
  uint4v v = {(unsigned)lane, 1u, 2u, 3u};
  for (int it = 0; it < iters; ++it) {
    char* tile = base + (size_t)it * 128 * ld;            // 128 rows per iteration
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (PAT == 0 || PAT == 4) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          char* p = tile + (size_t)(i * 16 + t * 8 + (lane >> 3)) * ld + (lane & 7) * 16;
          if (PAT == 4) __builtin_nontemporal_store(v, (uint4v*)p); else *(uint4v*)p = v;
        }
      } else if (PAT == 1 || PAT == 3) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          char* p = tile + (size_t)(i * 16 + (lane & 15)) * ld + q * 64 + (lane >> 4) * 16;
          if (PAT == 3) __builtin_nontemporal_store(v, (uint4v*)p); else *(uint4v*)p = v;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          char* p = out + (size_t)blockIdx.x * wg_stride + ((size_t)((it * 8 + i) * 2 + t) * 4 + wid) * 1024 + lane * 16;
          *(uint4v*)p = v;
        }
      }
    }
  }
}
int main(int argc, char** argv) {
  int iters = 64;
  size_t ld = 6144;                        // bytes per row (N = 3072 fp16)
  size_t wg_stride = (size_t)iters * 128 * ld;   // each WG its own row band; waves side by side (4 x 128 B = 512 B of each row)
  for (int grid : {8, 64, 256}) {
    size_t bytes = (size_t)grid * wg_stride + (1 << 20);
    char* d; if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc fail\n"); return 1; }
    hipMemset(d, 0, bytes);
    for (int pat = 0; pat < 5; ++pat) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&]() {
        switch (pat) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, ld, iters, wg_stride); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, ld, iters, wg_stride); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, ld, iters, wg_stride); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d, ld, iters, wg_stride); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, d, ld, iters, wg_stride); break;
        }
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      double per_wg = (double)iters * 4 * 16 * 1024;      // bytes per WG
      printf("grid %3d pattern %d: %8.1f us  %7.2f GB/s per CU  %7.2f TB/s total\n", grid, pat, ms * 1e3, per_wg / (ms * 1e-3) / 1e9,
             per_wg * grid / (ms * 1e-3) / 1e12);
    }
    hipFree(d);
  }
  return 0;
}
