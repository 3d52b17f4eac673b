// Pointwise / gather pieces of the GMFSS(_UNION) path that sit between the convolutions:
// MetricNet's 14-channel input, PixelShuffle, the timestep-map hole fill and the >25x swap
// masks of Model.inference, clamp.  One lane per pixel, XCD-banded 32x8 tiles.
#include "common.hpp"

using namespace drba;

namespace {

// zeros-padding bilinear sample at pixel coordinates (sx, sy) (grid_sample align_corners=True)
__device__ __forceinline__ float sample_zeros(const float *__restrict__ pl, int W, int H, float sx, float sy) {
  if (!(isfinite(sx) && isfinite(sy))) return 0.f;
  const float fx = floorf(sx), fy = floorf(sy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = sx - fx, wy1 = sy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  float v = 0.f;
  const bool okx0 = x0 >= 0 && x0 < W, okx1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool oky0 = y0 >= 0 && y0 < H, oky1 = y0 + 1 >= 0 && y0 + 1 < H;
  if (okx0 && oky0) v += pl[(size_t)y0 * W + x0] * (wx0 * wy0);
  if (okx1 && oky0) v += pl[(size_t)y0 * W + x0 + 1] * (wx1 * wy0);
  if (okx0 && oky1) v += pl[(size_t)(y0 + 1) * W + x0] * (wx0 * wy1);
  if (okx1 && oky1) v += pl[(size_t)(y0 + 1) * W + x0 + 1] * (wx1 * wy1);
  return v;
}

// GMFlow's flow_warp coordinate (geometry.py:53-84): g = 2*(c+f)/(size-1) - 1, un-normalised by grid_sample
__device__ __forceinline__ float fw_coord(int c, float f, int n) {
  const float g = 2.f * ((float)c + f) / (float)(n - 1) - 1.f;
  return (g + 1.f) * (((float)n - 1.f) / 2.f);
}

// MetricNet.forward input assembly (model_gmfss_union/MetricNet.py:45-60 + geometry.py:87-108):
// [img0 3, img1 3, -mean|img0 - backwarp(img1, f01)|, -mean|img1 - backwarp(img0, f10)|,
//  f01/((W-1)/2, (H-1)/2), f10/(...), fwd_occ, bwd_occ]
__global__ void __launch_bounds__(256)
metric_input_kernel(const float *__restrict__ img0, const float *__restrict__ img1, const float *__restrict__ f01,
                    const float *__restrict__ f10, float *__restrict__ out, int H, int W) {
  const size_t P = (size_t)H * W;
  const Tile2D tp = tile_pixel(W, H);
  if (!tp.valid) return;
  const int x = tp.x, y = tp.y;
  const size_t p = (size_t)y * W + x;
  const float a0 = f01[p], a1 = f01[P + p], b0 = f10[p], b1 = f10[P + p];
  // photometric terms: MetricNet's own backwarp (linspace grid + flow/((size-1)/2), zeros padding)
  const float sx01 = warp_coord(x, W, a0), sy01 = warp_coord(y, H, a1);
  const float sx10 = warp_coord(x, W, b0), sy10 = warp_coord(y, H, b1);
  float m0 = 0.f, m1 = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float i0 = img0[(size_t)c * P + p], i1 = img1[(size_t)c * P + p];
    out[(size_t)c * P + p] = i0;
    out[(size_t)(3 + c) * P + p] = i1;
    m0 += fabsf(i0 - sample_zeros(img1 + (size_t)c * P, W, H, sx01, sy01));
    m1 += fabsf(i1 - sample_zeros(img0 + (size_t)c * P, W, H, sx10, sy10));
  }
  out[6 * P + p] = -(m0 / 3.f);
  out[7 * P + p] = -(m1 / 3.f);
  const float hx = ((float)W - 1.f) / 2.f, hy = ((float)H - 1.f) / 2.f;
  out[8 * P + p] = a0 / hx;
  out[9 * P + p] = a1 / hy;
  out[10 * P + p] = b0 / hx;
  out[11 * P + p] = b1 / hy;
  // forward-backward consistency: |f + warp(b, f)| > 0.01*(|f|+|b|) + 0.5
  const float mag = sqrtf(a0 * a0 + a1 * a1) + sqrtf(b0 * b0 + b1 * b1);
  const float thr = 0.01f * mag + 0.5f;
  const float wx = fw_coord(x, a0, W), wy = fw_coord(y, a1, H);
  const float wb0 = sample_zeros(f10, W, H, wx, wy), wb1 = sample_zeros(f10 + P, W, H, wx, wy);
  const float vx = fw_coord(x, b0, W), vy = fw_coord(y, b1, H);
  const float wf0 = sample_zeros(f01, W, H, vx, vy), wf1 = sample_zeros(f01 + P, W, H, vx, vy);
  const float df = sqrtf((a0 + wb0) * (a0 + wb0) + (a1 + wb1) * (a1 + wb1));
  const float db = sqrtf((b0 + wf0) * (b0 + wf0) + (b1 + wf1) * (b1 + wf1));
  out[12 * P + p] = df > thr ? 1.f : 0.f;
  out[13 * P + p] = db > thr ? 1.f : 0.f;
}

// PixelShuffle(2): out[c, 2h+i, 2w+j] = in[4c + 2i + j, h, w]
__global__ void __launch_bounds__(256)
pixel_shuffle2_kernel(const float *__restrict__ in, float *__restrict__ out, int C, int H, int W) {
  const Tile2D tp = tile_pixel(2 * W, 2 * H);
  if (!tp.valid) return;
  const int X = tp.x, Y = tp.y, h = Y >> 1, w = X >> 1, sub = ((Y & 1) << 1) | (X & 1);
  const size_t Pi = (size_t)H * W, Po = Pi * 4;
  for (int c = 0; c < C; ++c) out[(size_t)c * Po + (size_t)Y * (2 * W) + X] = in[(size_t)(4 * c + sub) * Pi + (size_t)h * W + w];
}

// GMFSS.py:120-122: where either ones-splat < 0.999 both warped timestep maps are reset to 1
__global__ void __launch_bounds__(256)
timestep_fix_kernel(const float *__restrict__ t0, const float *__restrict__ t1, const float *__restrict__ c0,
                    const float *__restrict__ c1, float *__restrict__ o0, float *__restrict__ o1, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const bool bad = (c0[i] < 0.999f) || (c1[i] < 0.999f);
    o0[i] = bad ? 1.f : t0[i];
    o1[i] = bad ? 1.f : t1[i];
  }
}

// GMFSS.py:125-150: x[m0], y[m1] = y[m0], x[m1] with m0 = t0/t1 > thr, m1 = t1/t0 > thr (maps broadcast over C)
__global__ void __launch_bounds__(256)
swap_select_kernel(const float *x, const float *y, const float *__restrict__ t0,  // (x / ox and y / oy may be the same arrays)
                   const float *__restrict__ t1, float *ox, float *oy, int C, size_t P,
                   float thr) {
  const bool inplace = x == ox && y == oy;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
    const float a = t0[p], b = t1[p];
    const bool m0 = (a / b) > thr, m1 = (b / a) > thr;
    // in place (ox == x, oy == y: the splats wrote GridNet's input slices themselves) a pixel neither mask selects has nothing to
    // move -- the masks mark scene changes, i.e. almost no pixel of almost every frame: the pass reads the two maps and little else
    if (inplace && !m0 && !m1) continue;
    for (int c = 0; c < C; ++c) {
      const float xv = x[(size_t)c * P + p], yv = y[(size_t)c * P + p];
      ox[(size_t)c * P + p] = m0 ? yv : xv;
      oy[(size_t)c * P + p] = m1 ? xv : yv;
    }
  }
}

__global__ void __launch_bounds__(256)
clamp_kernel(const float *__restrict__ in, float *__restrict__ out, float lo, float hi, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = in[i];
    out[i] = v != v ? v : fminf(fmaxf(v, lo), hi);  // torch.clamp propagates NaN
  }
}

}  // namespace

extern "C" {

int drba_metric_input(const float *img0, const float *img1, const float *flow01, const float *flow10, float *out, int H,
                      int W, void *stream) {
  if (!img0 || !img1 || !flow01 || !flow10 || !out || H <= 1 || W <= 1) return DRBA_EINVAL;
  DRBA_LAUNCH(metric_input_kernel, dim3(tiles_for(W, H)), dim3(kBlock), 0, (hipStream_t)stream, img0, img1,
                     flow01, flow10, out, H, W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_pixel_shuffle2(const float *in, float *out, int C, int H, int W, void *stream) {
  if (!in || !out || C <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  DRBA_LAUNCH(pixel_shuffle2_kernel, dim3(tiles_for(2 * W, 2 * H)), dim3(kBlock), 0, (hipStream_t)stream, in, out,
                     C, H, W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_timestep_fix(const float *t0, const float *t1, const float *cover0, const float *cover1, float *out0,
                      float *out1, size_t n, void *stream) {
  if (!t0 || !t1 || !cover0 || !cover1 || !out0 || !out1 || n == 0) return DRBA_EINVAL;
  DRBA_LAUNCH(timestep_fix_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, t0, t1, cover0,
                     cover1, out0, out1, n);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_swap_select(const float *x, const float *y, const float *t0, const float *t1, float *out_x, float *out_y, int C,
                     int H, int W, float thr, void *stream) {
  if (!x || !y || !t0 || !t1 || !out_x || !out_y || C <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  DRBA_LAUNCH(swap_select_kernel, dim3(grid_for((size_t)H * W)), dim3(kBlock), 0, (hipStream_t)stream, x, y, t0,
                     t1, out_x, out_y, C, (size_t)H * W, thr);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_clamp(const float *in, float *out, float lo, float hi, size_t n, void *stream) {
  if (!in || !out || n == 0) return DRBA_EINVAL;
  DRBA_LAUNCH(clamp_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, in, out, lo, hi, n);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // extern "C"
