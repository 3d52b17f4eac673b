// Scene-cut metric: SSIM with a 3-D 11x11x11 gaussian over the [3,32,32] thumbnail volume,
// replicate padding 5 on every axis (pytorch_msssim/__init__.py:83-136, used by
// tools.py:27-30).  One workgroup, everything in LDS/registers; the window is separable
// (it is built as an outer product, __init__.py:21-26) so each field takes three 11-tap passes.
#include "common.hpp"

#include <math.h>

using namespace drba;

namespace {

struct Gauss11 {
  float g[11];
};

constexpr int VOL = 3 * 32 * 32;

__device__ __forceinline__ float block_sum(float v, float *red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;  // valid in thread 0
}

__global__ void __launch_bounds__(1024) ssim3d_kernel(const float *__restrict__ x1, const float *__restrict__ x2,
                                                      float *__restrict__ out, Gauss11 G) {
  __shared__ float bufA[VOL], bufB[VOL], red[16], lim[2];
  const int tid = threadIdx.x;
  // val_range from img1 (pytorch_msssim/__init__.py:85-97)
  float mx = -INFINITY, mn = INFINITY;
  for (int e = tid; e < VOL; e += 1024) {
    mx = fmaxf(mx, x1[e]);
    mn = fminf(mn, x1[e]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    mx = fmaxf(mx, __shfl_down(mx, o, 64));
    mn = fminf(mn, __shfl_down(mn, o, 64));
  }
  if ((tid & 63) == 0) {
    bufA[tid >> 6] = mx;
    bufB[tid >> 6] = mn;
  }
  __syncthreads();
  if (tid == 0) {
    float a = bufA[0], b = bufB[0];
    for (int i = 1; i < 16; ++i) {
      a = fmaxf(a, bufA[i]);
      b = fminf(b, bufB[i]);
    }
    lim[0] = a;
    lim[1] = b;
  }
  __syncthreads();
  const float L = ((lim[0] > 128.f) ? 255.f : 1.f) - ((lim[1] < -0.5f) ? -1.f : 0.f);
  const float C1 = (float)((0.01 * (double)L) * (0.01 * (double)L));
  const float C2 = (float)((0.03 * (double)L) * (0.03 * (double)L));

  float blur[5][3];  // 5 fields x this thread's 3 voxels
#pragma unroll
  for (int f = 0; f < 5; ++f) {
    __syncthreads();
    for (int e = tid; e < VOL; e += 1024) {
      const float a = x1[e], b = x2[e];
      bufA[e] = f == 0 ? a : f == 1 ? b : f == 2 ? a * a : f == 3 ? b * b : a * b;
    }
    __syncthreads();
    for (int e = tid; e < VOL; e += 1024) {  // along x
      const int x = e & 31, base = e - x;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) s += G.g[k] * bufA[base + min(max(x + k - 5, 0), 31)];
      bufB[e] = s;
    }
    __syncthreads();
    for (int e = tid; e < VOL; e += 1024) {  // along y
      const int x = e & 31, y = (e >> 5) & 31, c = e >> 10;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) s += G.g[k] * bufB[(c << 10) + (min(max(y + k - 5, 0), 31) << 5) + x];
      bufA[e] = s;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; ++j) {  // along the channel axis
      const int e = tid + j * 1024, c = e >> 10, yx = e & 1023;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) s += G.g[k] * bufA[(min(max(c + k - 5, 0), 2) << 10) + yx];
      blur[f][j] = s;
    }
  }
  float part = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float mu1 = blur[0][j], mu2 = blur[1][j];
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = blur[2][j] - mu1_sq, s2 = blur[3][j] - mu2_sq, s12 = blur[4][j] - mu12;
    const float v1 = 2.0f * s12 + C2, v2 = s1 + s2 + C2;
    part += ((2.f * mu12 + C1) * v1) / ((mu1_sq + mu2_sq + C1) * v2);
  }
  const float tot = block_sum(part, red);
  if (tid == 0) out[0] = tot / (float)VOL;
}

}  // namespace

extern "C" int drba_ssim3d_32(const float *x1, const float *x2, float *out, void *stream) {
  if (!x1 || !x2 || !out) return DRBA_EINVAL;
  Gauss11 G;
  float sum = 0.f;
  for (int k = 0; k < 11; ++k) {  // gaussian(11, 1.5): double exp -> fp32, normalised in fp32
    G.g[k] = (float)exp(-(double)((k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5));
    sum += G.g[k];
  }
  for (int k = 0; k < 11; ++k) G.g[k] /= sum;
  DRBA_LAUNCH(ssim3d_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x1, x2, out, G);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}
