// Texture-addresser cost of vector-memory loads on gfx950: clocks per wave-level load instruction per CU for 4 / 8 / 16
// bytes per lane, (a) lanes contiguous, (b) lanes 8 bytes apart with 16-byte reads (the bilinear tap pattern: neighbours
// overlap), (c) lanes scattered over rows.  Working set L1/L2 resident.   hipcc --offload-arch=gfx950 ta_rate.hip -o ta_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int BYTES, int MODE>
__global__ void __launch_bounds__(256) k(const float *src, float *out, int iters, int rowstride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // per-lane byte offset inside a 256 KB window private to the workgroup
  uint32_t off;
  if (MODE == 0) off = lane * BYTES;                        // contiguous
  else if (MODE == 1) off = lane * 8;                       // tap pattern: 8 bytes apart
  else off = (lane & 7) * 8 + (lane >> 3) * rowstride;      // 8 lanes per row, rows `rowstride` bytes apart
  const char *base = (const char *)src + (size_t)(blockIdx.x % 64) * (1 << 18) + wave * 4096;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 1 << 18, 0x00020000);
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t so = (uint32_t)(((i * 8 + j) & 15) * 256);
      if (BYTES == 4) acc += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, so, 0));
      else if (BYTES == 8) { u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, so, 0); acc += __uint_as_float(v.x) + __uint_as_float(v.y); }
      else { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, so, 0); acc += __uint_as_float(v.x) + __uint_as_float(v.w); }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}
template <int BYTES, int MODE>
void run(const float *src, float *out, const char *name, int rowstride = 7680) {
  const int iters = 2000, blocks = 256 * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<BYTES, MODE>), dim3(blocks), dim3(256), 0, 0, src, out, 10, rowstride);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<BYTES, MODE>), dim3(blocks), dim3(256), 0, 0, src, out, iters, rowstride);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
  printf("%-34s %2d B/lane: %7.3f ms  %6.1f ns per wave-load per CU  (%.1f clk at 2.1 GHz)  %.1f GB/s chip\n", name, BYTES, ms,
         ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.1, (double)blocks * 4 * iters * 8 * 64 * BYTES / ms / 1e6);
}
int main() {
  float *src, *out;
  hipMalloc(&src, 64 << 18); hipMalloc(&out, 4);
  hipMemset(src, 0, 64 << 18);
  run<4, 0>(src, out, "contiguous"); run<8, 0>(src, out, "contiguous"); run<16, 0>(src, out, "contiguous");
  run<8, 1>(src, out, "8 B apart (tap pattern)"); run<16, 1>(src, out, "8 B apart, 16 B reads (overlap)");
  run<4, 2>(src, out, "8 lanes/row, rows 7680 B apart"); run<8, 2>(src, out, "8 lanes/row, rows 7680 B apart"); run<16, 2>(src, out, "8 lanes/row, rows 7680 B apart");
  run<16, 2>(src, out, "8 lanes/row, rows 256 B apart", 256);
  return 0;
}
