// Throughput of LDS atomics on gfx950: lane-ops per clock per CU for ds_add_f32 (no return), ds_add_rtn_u32, ds_add_u32.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  __shared__ float sf[4096];
  __shared__ unsigned su[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) { sf[i] = 0.f; su[i] = 0; }
  __syncthreads();
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int a = (threadIdx.x + j * 256 + it * 7) & 4095;  // conflict-free: consecutive lanes, consecutive dwords
      if (KIND == 0) atomicAdd(&sf[a], 1.0f);
      if (KIND == 1) acc += atomicAdd(&su[a], 1u);
      if (KIND == 2) atomicAdd(&su[a], 1u);
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = sf[threadIdx.x] + (float)su[threadIdx.x] + (float)acc;
}
int main() {
  float* d; hipMalloc(&d, 1024 * 256 * 4);
  const int iters = 2000, blocks = 1024;
  for (int kind = 0; kind < 3; ++kind) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
      if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
      if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * 8;
    printf("kind %d (%s): %.3f ms, %.1f G lane-atomics/s, %.3f per clk per CU (256 CUs @2.4GHz)\n", kind,
           kind == 0 ? "ds_add_f32" : kind == 1 ? "ds_add_rtn_u32" : "ds_add_u32", ms, ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.4e9);
  }
  return 0;
}
