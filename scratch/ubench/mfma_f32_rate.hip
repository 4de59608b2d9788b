// Issue rate of the exact-f32 MFMAs on gfx950: cycles per instruction with 4 independent accumulators per wave,
// at 1 / 2 / 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o scratch/_dbg/mfma_f32_rate scratch/ubench/mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ void k(float* out, int iters, long long* cyc) {
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  long long t0 = 0, t1 = 0;
  if constexpr (KIND == 0) {
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int kind = 0; kind < 2; ++kind)
    for (int waves = 1; waves <= 4; waves *= 2) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      dim3 grid(256), block(256 * waves);      // one workgroup per CU, `waves` waves per SIMD
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, iters, cyc);
        else hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      double flops = (kind == 0 ? 2048.0 : 4096.0) * 4 * iters * 256 * 4 * waves;
      printf("%s waves/SIMD %d: %.3f ms  %.1f TFLOP/s  (%.1f shader-clock ticks per MFMA per wave)\n", kind == 0 ? "16x16x4f32" : "32x32x2f32",
             waves, ms, flops / ms / 1e9, (double)c / (4.0 * iters));
    }
  return 0;
}
