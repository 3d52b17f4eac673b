#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const float* in, float* out, int n) {
  __shared__ float smem[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) smem[i] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // dword: lane l of wave w reads element (threadIdx.x*3 + 5), soffset 16 bytes; odd lanes of wave 1 forced OOB
  unsigned voff = (threadIdx.x * 3 + 5) * 4;
  if (wave == 1 && (lane & 1)) voff = 0x7FFFFFF0u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(smem + wave * 64), 4, voff, 16, 0, 0);
  // dwordx4: lane reads 4 floats at element threadIdx.x*4 + 1000 (soffset 0)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(smem + 1024 + wave * 256), 16, (threadIdx.x * 4 + 1000) * 4, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256) out[i] = smem[i];
}
int main() {
  const int n = 2000;  // so that the x4 loads of the last lanes run past the end (elements >= 2000 -> 0)
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *din, *dout;
  hipMalloc(&din, n * 4); hipMalloc(&dout, 2048 * 4);
  hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, din, dout, n);
  std::vector<float> o(2048);
  hipMemcpy(o.data(), dout, 2048 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) {
    int w = t >> 6, l = t & 63;
    float exp = (float)(t * 3 + 5 + 4);
    if (w == 1 && (l & 1)) exp = 0.f;
    if (t * 3 + 5 + 4 >= n) exp = 0.f;
    if (o[t] != exp) { if (bad < 10) printf("dword t=%d got %g exp %g\n", t, o[t], exp); ++bad; }
  }
  for (int t = 0; t < 256; ++t) for (int j = 0; j < 4; ++j) {
    int e = t * 4 + 1000 + j;
    float exp = e < n ? (float)e : 0.f;
    float got = o[1024 + t * 4 + j];
    if (got != exp) { if (bad < 20) printf("x4 t=%d j=%d got %g exp %g\n", t, j, got, exp); ++bad; }
  }
  printf("untouched smem[300]=%g smem[1023]=%g\n", o[300], o[1023]);
  printf("bad=%d\n", bad);
  return 0;
}
