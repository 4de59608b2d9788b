// Does the SOURCE of the store data matter?  Every workgroup (8 waves) writes a 256x256 fp16 tile, 16 stores of 8 rows x 128 B per
// wave in ideal lane order (lane l -> row l >> 3, 16-byte chunk l & 7), with the data coming from
//   src 0: one register quad for all 16 stores (scratch/ubench/burst_store.hip)
//   src 1: 16 different register quads (64 VGPRs, as accumulators would be)
//   src 2: a ds_read_b128 per store from the wave's LDS tile (the GEMM epilogue's path)
//   src 3: as 2, with 8 LDS-DMA loads of 1 KB per wave issued just before the stores (the K-loop's prefetch in flight)
// hipcc --offload-arch=gfx950 -O3 scratch/ubench/store_src.hip -o /tmp/store_src
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(unsigned)(size_t)(p))
__global__ __launch_bounds__(512) void k(char* out, const char* in, size_t ld, int iters, int ntn, int gap_ticks, unsigned* stamps, int src) {
  __shared__ __attribute__((aligned(16))) char lds[8 * 4096 + 8 * 8192];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 2, wn = wid & 3;
  uint4v v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = uint4v{(unsigned)lane + i, 1u, 2u, 3u};
  char* scr = lds + wid * 4096;
  *reinterpret_cast<uint4v*>(scr + lane * 16) = v[0];
  *reinterpret_cast<uint4v*>(scr + 1024 + lane * 16) = v[1];
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 1 << 30, 0x00020000);
  const int row = lane >> 3, chunk = lane & 7;
  for (int it = 0; it < iters; ++it) {
    int tile = blockIdx.x + it * gridDim.x;
    int tm = tile / ntn, tn = tile % ntn;
    char* base = out + ((size_t)(tm * 256 + wm * 128) * ld) + (size_t)(tn * 256 + wn * 64) * 2;
    __syncthreads();
    if (src == 3) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + 8 * 4096 + wid * 8192 + q * 1024), 16,
                                                 (unsigned)((tile * 8 + wid) * 8 + q) * 1024u + lane * 16u, 0, 0, 0);
    }
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        char* p = base + (size_t)(i * 16 + t * 8 + row) * ld + chunk * 16;
        uint4v d;
        if (src == 0) d = v[0];
        else if (src == 1) d = v[2 * i + t];
        else d = *reinterpret_cast<const uint4v*>(scr + t * 1024 + ((lane * 16) ^ (i * 16)));
        __builtin_nontemporal_store(d, (uint4v*)p);
      }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { unsigned* s = stamps + ((size_t)(blockIdx.x * 8 + wid) * iters + it) * 2; s[0] = (unsigned)(t1 - t0); s[1] = (unsigned)(t2 - t0); }
    unsigned long long until = t2 + gap_ticks;
    while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(v[i]));
  }
}
int main() {
  const int iters = 12, ntn = 12;
  size_t ld = 6144;
  size_t bytes = (size_t)((256 * iters + ntn - 1) / ntn + 1) * 256 * ld;
  char* d; if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc fail\n"); return 1; }
  char* in; hipMalloc(&in, (size_t)1 << 30); hipMemset(in, 1, (size_t)1 << 30);
  hipMemset(d, 0, bytes);
  unsigned* st; hipMalloc(&st, 256 * 8 * iters * 2 * 4);
  for (int src : {0, 1, 2, 3}) for (int grid : {8, 256}) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, d, in, ld, iters, ntn, 1500, st, src); hipDeviceSynchronize(); }
    std::vector<unsigned> h(grid * 8 * iters * 2);
    hipMemcpy(h.data(), st, h.size() * 4, hipMemcpyDeviceToHost);
    std::vector<double> iss, ack;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < 8; ++w) for (int it = 2; it < iters; ++it) {
      size_t o = ((size_t)(b * 8 + w) * iters + it) * 2; iss.push_back(h[o] / 100.0); ack.push_back(h[o + 1] / 100.0);
    }
    std::sort(iss.begin(), iss.end()); std::sort(ack.begin(), ack.end());
    printf("src %d grid %3d: issue us median %.2f p90 %.2f max %.2f | ack us median %.2f p90 %.2f max %.2f\n", src, grid,
           iss[iss.size() / 2], iss[iss.size() * 9 / 10], iss.back(), ack[ack.size() / 2], ack[ack.size() * 9 / 10], ack.back());
  }
  return 0;
}
