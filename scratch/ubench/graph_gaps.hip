// Gap between dependent small kernels: plain stream launches vs one hipGraph replay (is a captured tower call worth it?).
// hipcc --offload-arch=gfx950 -O3 scratch/ubench/graph_gaps.hip -o /tmp/graph_gaps && /tmp/graph_gaps
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long cycles) { long t0 = clock64(); while (clock64() - t0 < cycles) {} }
__global__ void tiny(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
  const int N = 400, n = 1 << 16;
  float* d; hipMalloc(&d, n * 4); hipMemset(d, 0, n * 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {64, 1024}) {
    for (int it = 0; it < 3; ++it) {
      auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 400000l);      // ~4 ms at 100 MHz: the queue fills while it runs
      hipEventRecord(e0, s);
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, n);
      hipEventRecord(e1, s);
      auto t1 = std::chrono::steady_clock::now();
      hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it == 2) printf("grid %4d stream: %.2f us per kernel on the GPU, host enqueue %.2f us per kernel\n", grid, ms * 1e3 / N,
                          std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, n);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int it = 0; it < 3; ++it) {
      auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 400000l);
      hipEventRecord(e0, s);
      hipGraphLaunch(ge, s);
      hipEventRecord(e1, s);
      auto t1 = std::chrono::steady_clock::now();
      hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it == 2) printf("grid %4d graph : %.2f us per kernel on the GPU, host enqueue %.2f us per kernel\n", grid, ms * 1e3 / N,
                          std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
