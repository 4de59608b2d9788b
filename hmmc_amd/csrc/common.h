// Shared device helpers for the HMMC gfx950 kernels.  CDNA4 only: 64-lane waves,
// MFMA 16x16x32 f16, LDS-DMA buffer loads.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

#define HMMC_OK 0
#define HMMC_ERR_ARG (-1)
#define HMMC_ERR_UNSUPPORTED (-2)
#define HMMC_ERR_WORKSPACE (-3)
#define HMMC_ERR_LAUNCH (-4)

static inline int hmmc_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? HMMC_OK : HMMC_ERR_LAUNCH;
}

#define HMMC_MAX_DEVICES 16
static inline int hmmc_current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev >= 0 && dev < HMMC_MAX_DEVICES ? dev : 0;
}

// kernels that ask for more than 64 KiB of dynamic LDS must opt in, once per device (`done` = the call site's own
// per-device flags; a racing second thread merely repeats the idempotent call)
static inline void hmmc_allow_lds(const void* kernel, int bytes, bool (&done)[HMMC_MAX_DEVICES]) {
  const int dev = hmmc_current_device();
  if (done[dev]) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done[dev] = true;
}

// compute units of the current device (queried once per device)
static inline int hmmc_num_cus() {
  static int cus[HMMC_MAX_DEVICES] = {0};
  const int dev = hmmc_current_device();
  if (cus[dev] == 0) {
    int n = 0;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus[dev] = n > 0 ? n : 256;
  }
  return cus[dev];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// round-trip through fp16 (the reference rounds at every fp16 tensor op)
__device__ __forceinline__ float r16(float x) { return (float)(half_t)x; }
// the same with the fp32 value pinned first: torch computes `half_tensor * python_float` as an fp32 product that is THEN
// rounded to fp16 (two roundings); left alone, hipcc fuses product and conversion into v_fma_mixlo_f16 (one rounding),
// which differs on fp32 ties (0.02 % of QuickGELU inputs)
__device__ __forceinline__ float opq(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ float r16s(float x) { return (float)(half_t)opq(x); }

// QuickGELU evaluated with the reference's fp16 rounding points
// (modules/module_clip.py:226-228: x * sigmoid(1.702 * x) on an fp16 tensor).  exp(-t) = exp2(-log2(e) t): one
// multiply + v_exp_f32; v_rcp_f32 is 1 ulp, far inside the fp16 rounding that follows.
__device__ __forceinline__ float qgelu_f16(float h) {
  float t = r16s(1.702f * h);
  float s = r16(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f)));
  return r16(h * s);
}
// d/dh [h * sigmoid(1.702 h)] = s + 1.702 h s (1 - s)
__device__ __forceinline__ float qgelu_grad(float h) {
  float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(h * (-1.702f * 1.4426950408889634f)));
  return __builtin_fmaf(1.702f * h * (1.0f - s), s, s);
}
