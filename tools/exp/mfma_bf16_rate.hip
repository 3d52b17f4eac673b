// v_mfma_f32_16x16x32_bf16 issue rate on gfx950: cycles per MFMA for (a) 8 independent accumulators, (b) 4, (c) 2,
// (d) one dependent chain; and the fp32 16x16x4 for reference.  hipcc --offload-arch=gfx950 -O3 mfma_bf16_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, bool BF>
__global__ void k(float *out, long long *cyc, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (__bf16)(threadIdx.x * 0.001f + i), b[i] = (__bf16)(1.f + i * 0.01f);
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0, 0, 0, 0};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / CHAINS; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (BF) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[0], (float)b[0], acc[c], 0, 0, 0);
      }
  }
  long long t1 = clock64();
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CHAINS, bool BF>
void run(const char *name, int waves_per_simd) {
  float *out; long long *cyc, h;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
  const int iters = 2000;
  k<CHAINS, BF><<<256, 256 * waves_per_simd>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<CHAINS, BF><<<256, 256 * waves_per_simd>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = iters * 8.0;
  printf("%-28s waves/SIMD %d: %.1f clk/MFMA per wave (counter), kernel %.1f us -> %.2f ns per MFMA per SIMD\n", name, waves_per_simd, h / n,
         ms * 1e3, ms * 1e6 / (n * waves_per_simd));
}

int main() {
  run<8, true>("bf16 16x16x32, 8 chains", 1);
  run<4, true>("bf16 16x16x32, 4 chains", 1);
  run<2, true>("bf16 16x16x32, 2 chains", 1);
  run<1, true>("bf16 16x16x32, 1 chain", 1);
  run<1, true>("bf16 16x16x32, 1 chain", 2);
  run<2, true>("bf16 16x16x32, 2 chains", 2);
  run<8, false>("f32 16x16x4, 8 chains", 1);
  run<1, false>("f32 16x16x4, 1 chain", 1);
  return 0;
}
