// Forward splat (softsplat), backward warp, flow distance and the fused RIFE/DRM splat
// pipelines.  All HBM-bound: one lane per pixel, coalesced NCHW plane reads, fp32 hardware
// atomics into a pixel-interleaved accumulator ([H*W][C+1]) so the 4 bilinear corners of a
// source pixel touch at most two cache-line runs regardless of C.
#include "common.hpp"

using namespace drba;

namespace {

// ------------------------------------------------------------------------------------------
// softsplat.py:312-357 — scatter.  Reference launches one thread per (n,c,y,x) element and
// recomputes the target/weights per channel; here one lane owns a source pixel and loops
// channels, so flow, floor() and the 4 weights are computed once.
// mode: 0 sum, 1 avg, 2 linear, 3 soft.  CP = channels in the accumulator (C or C+1).
__global__ void __launch_bounds__(256) splat_scatter(const float *__restrict__ in, const float *__restrict__ flow,
                              const float *__restrict__ metric, float *__restrict__ acc, int C, int H,
                              int W, int mode) {
  const int n = blockIdx.y;
  const size_t P = (size_t)H * W;
  const int CP = (mode == 0) ? C : C + 1;
  in += (size_t)n * C * P;
  flow += (size_t)n * 2 * P;
  if (metric) metric += (size_t)n * P;
  acc += (size_t)n * P * CP;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
    const float X = (float)x + flow[p], Y = (float)y + flow[P + p];
    if (!(isfinite(X) && isfinite(Y))) continue;
    const float fx = floorf(X), fy = floorf(Y);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx0 = (fx + 1.f) - X, wx1 = X - fx, wy0 = (fy + 1.f) - Y, wy1 = Y - fy;
    const float wgt[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};  // NW NE SW SE
    const bool okx0 = x0 >= 0 && x0 < W, okx1 = x0 + 1 >= 0 && x0 + 1 < W;
    const bool oky0 = y0 >= 0 && y0 < H, oky1 = y0 + 1 >= 0 && y0 + 1 < H;
    const bool ok[4] = {okx0 && oky0, okx1 && oky0, okx0 && oky1, okx1 && oky1};
    if (!(ok[0] || ok[1] || ok[2] || ok[3])) continue;
    float *dst[4];
    const long long base = (long long)y0 * W + x0;  // may be negative when out of bounds; only used if ok
    dst[0] = acc + (base)*CP;
    dst[1] = acc + (base + 1) * CP;
    dst[2] = acc + (base + W) * CP;
    dst[3] = acc + (base + W + 1) * CP;
    float m = 1.f;
    if (mode == 2) m = metric[p];
    if (mode == 3) m = expf(metric[p]);
    for (int c = 0; c < C; ++c) {
      const float v = (mode >= 2) ? in[(size_t)c * P + p] * m : in[(size_t)c * P + p];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ok[k]) atomic_add_f32(dst[k] + c, v * wgt[k]);
    }
    if (mode != 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ok[k]) atomic_add_f32(dst[k] + C, m * wgt[k]);
    }
  }
}

// softsplat_torch.py:48-65 — divide by the splatted normaliser.
__global__ void __launch_bounds__(256) splat_normalize(const float *__restrict__ acc, float *__restrict__ out, int C, int H, int W,
                                int mode, int eps) {
  const int n = blockIdx.y;
  const size_t P = (size_t)H * W;
  const int CP = (mode == 0) ? C : C + 1;
  acc += (size_t)n * P * CP;
  out += (size_t)n * C * P;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
    const float *a = acc + p * CP;
    if (mode == 0) {
      for (int c = 0; c < C; ++c) out[(size_t)c * P + p] = a[c];
      continue;
    }
    float nrm = a[C];
    if (eps == 0) nrm = nrm + 0.0000001f;
    else if (eps == 1) nrm = (nrm == 0.f) ? 1.f : nrm;
    else nrm = fmaxf(nrm, 0.0000001f) + (nrm != nrm ? nrm : 0.f);  // clip(min=1e-7); NaN stays NaN
    for (int c = 0; c < C; ++c) out[(size_t)c * P + p] = a[c] / nrm;
  }
}

// ------------------------------------------------------------------------------------------
// warplayer.py:8-22 / MetricNet.py:10-20.  One lane per output pixel; taps computed once and
// reused by every channel plane (gathers are L2-served: neighbouring lanes hit neighbouring
// source pixels for smooth flows).
template <bool ZEROS>
__global__ void __launch_bounds__(256) backwarp_kernel(const float *__restrict__ in, const float *__restrict__ flow,
                                float *__restrict__ out, int C, int H, int W) {
  const int n = blockIdx.y;
  const size_t P = (size_t)H * W;
  in += (size_t)n * C * P;
  flow += (size_t)n * 2 * P;
  out += (size_t)n * C * P;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
    const float sx = warp_coord(x, W, flow[p]), sy = warp_coord(y, H, flow[P + p]);
    if (!ZEROS) {
      const Taps t = taps_border(sx, sy, W, H);
      for (int c = 0; c < C; ++c) out[(size_t)c * P + p] = sample(in + (size_t)c * P, W, t);
    } else {
      const float fx = floorf(sx), fy = floorf(sy);
      const int x0 = (int)fx, y0 = (int)fy;
      const float wx1 = sx - fx, wy1 = sy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
      const bool okx0 = x0 >= 0 && x0 < W, okx1 = x0 + 1 >= 0 && x0 + 1 < W;
      const bool oky0 = y0 >= 0 && y0 < H, oky1 = y0 + 1 >= 0 && y0 + 1 < H;
      const bool fin = isfinite(sx) && isfinite(sy);
      for (int c = 0; c < C; ++c) {
        const float *pl = in + (size_t)c * P;
        float v = 0.f;
        if (fin) {
          if (okx0 && oky0) v += pl[(size_t)y0 * W + x0] * (wx0 * wy0);
          if (okx1 && oky0) v += pl[(size_t)y0 * W + x0 + 1] * (wx1 * wy0);
          if (okx0 && oky1) v += pl[(size_t)(y0 + 1) * W + x0] * (wx0 * wy1);
          if (okx1 && oky1) v += pl[(size_t)(y0 + 1) * W + x0 + 1] * (wx1 * wy1);
        }
        out[(size_t)c * P + p] = v;
      }
    }
  }
}

__global__ void __launch_bounds__(256) distance_kernel(const float *__restrict__ flow, float *__restrict__ out, int H, int W) {
  const int n = blockIdx.y;
  const size_t P = (size_t)H * W;
  flow += (size_t)n * 2 * P;
  out += (size_t)n * P;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
    const float u = flow[p], v = flow[P + p];
    out[p] = sqrtf(u * u + v * v);
  }
}

// ------------------------------------------------------------------------------------------
// Generic "value + ones" avg-splat scatter shared by the fused pipelines: accumulates
// [v0*w, (v1*w,) w] at (x+fx, y+fy).  NV = number of value channels (1 or 2).
template <int NV>
__device__ __forceinline__ void scatter_avg(float *__restrict__ acc, int x, int y, float fx_, float fy_,
                                            const float (&v)[NV], int H, int W) {
  const float X = (float)x + fx_, Y = (float)y + fy_;
  if (!(isfinite(X) && isfinite(Y))) return;
  const float fx = floorf(X), fy = floorf(Y);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx0 = (fx + 1.f) - X, wx1 = X - fx, wy0 = (fy + 1.f) - Y, wy1 = Y - fy;
  const float wgt[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
  const bool okx[2] = {x0 >= 0 && x0 < W, x0 + 1 >= 0 && x0 + 1 < W};
  const bool oky[2] = {y0 >= 0 && y0 < H, y0 + 1 >= 0 && y0 + 1 < H};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!(okx[k & 1] && oky[k >> 1])) continue;
    float *d = acc + ((long long)(y0 + (k >> 1)) * W + (x0 + (k & 1))) * (NV + 1);
#pragma unroll
    for (int c = 0; c < NV; ++c) atomic_add_f32(d + c, v[c] * wgt[k]);
    atomic_add_f32(d + NV, wgt[k]);
  }
}

// ------------------------------------------------------------------------------------------
// Fused avg-splat pipelines of the RIFE path, LDS-tiled.
//   MODE 0 (rife.py:59-73)  : value = (fx, fy), splat flow = the flow itself;
//                             out = 2 * (hole ? max(H,W) : -avg)
//   MODE 1 (drm.py:65-107)  : value = u = d_other/(d_self+d_other)*t*2, splat flow = self*u;
//                             out = hole ? u : avg
// A workgroup owns a TX x TY output tile whose accumulators live in LDS.  It visits every source
// pixel within R of the tile and adds the corners that land inside the tile with ds_add_f32 --
// no global atomics for "short" flows (all four corners within R of the source).  The rare
// "long" pixels are scattered by a pre-pass with global atomics into `gacc` (zeroed), which the
// finish step adds in.  Each (source, corner) pair is accumulated exactly once either way.
constexpr int kTX = 64, kTY = 32, kR = 16;

template <int MODE>
struct SplatSrc {
  float v[2];
  float fx, fy;
  bool finite, is_short;
};

template <int MODE>
__device__ __forceinline__ SplatSrc<MODE> splat_source(const float *__restrict__ fs, const float *__restrict__ fo,
                                                       size_t P, size_t p, int x, int y, float t, float eps) {
  SplatSrc<MODE> s;
  const float su = fs[p], sv = fs[P + p];
  if (MODE == 0) {
    s.v[0] = su;
    s.v[1] = sv;
    s.fx = su;
    s.fy = sv;
  } else {
    const float ou = fo[p], ov = fo[P + p];
    const float ds = sqrtf(su * su + sv * sv) + eps, d_o = sqrtf(ou * ou + ov * ov) + eps;
    const float u = (d_o / (ds + d_o)) * t * 2.f;
    s.v[0] = u;
    s.v[1] = 0.f;
    s.fx = su * u;
    s.fy = sv * u;
  }
  const float X = (float)x + s.fx, Y = (float)y + s.fy;
  s.finite = isfinite(X) && isfinite(Y);
  // all four corners within kR of the source <=> floor(f) >= -kR and floor(f)+1 <= kR
  s.is_short = s.fx >= -(float)kR && s.fx < (float)(kR - 1) && s.fy >= -(float)kR && s.fy < (float)(kR - 1);
  return s;
}

// pre-pass: long (but finite) pixels -> global atomics into gacc [P][NV+1]
template <int MODE>
__global__ void __launch_bounds__(256) splat_long_prepass(const float *__restrict__ fs, const float *__restrict__ fo,
                                                          float t, const float *__restrict__ t_dev, float eps,
                                                          float *__restrict__ gacc, int H, int W) {
  constexpr int NV = MODE == 0 ? 2 : 1;
  if (t_dev) t = *t_dev;  // timestep from device memory: lets one captured HIP graph serve every t
  const int n = blockIdx.y;
  const size_t P = (size_t)H * W;
  fs += (size_t)n * 2 * P;
  if (fo) fo += (size_t)n * 2 * P;
  gacc += (size_t)n * P * (NV + 1);
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
    const SplatSrc<MODE> s = splat_source<MODE>(fs, fo, P, p, x, y, t, eps);
    if (!s.finite || s.is_short) continue;
    if (MODE == 0) {
      const float v[2] = {s.v[0], s.v[1]};
      scatter_avg<2>(gacc, x, y, s.fx, s.fy, v, H, W);
    } else {
      const float v[1] = {s.v[0]};
      scatter_avg<1>(gacc, x, y, s.fx, s.fy, v, H, W);
    }
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) splat_tiled(const float *__restrict__ fs, const float *__restrict__ fo, float t,
                                                   const float *__restrict__ t_dev, float eps,
                                                   const float *__restrict__ gacc, float *__restrict__ out, int H,
                                                   int W) {
  constexpr int NV = MODE == 0 ? 2 : 1;
  if (t_dev) t = *t_dev;
  __shared__ float acc[NV + 1][kTY * kTX];
  const int n = blockIdx.z;
  const size_t P = (size_t)H * W;
  fs += (size_t)n * 2 * P;
  if (fo) fo += (size_t)n * 2 * P;
  gacc += (size_t)n * P * (NV + 1);
  out += (size_t)n * (MODE == 0 ? 2 : 1) * P;
  const int tx0 = blockIdx.x * kTX, ty0 = blockIdx.y * kTY;
  const int tid = threadIdx.x;
  for (int i = tid; i < (NV + 1) * kTY * kTX; i += 256) (&acc[0][0])[i] = 0.f;
  __syncthreads();
  constexpr int SW = kTX + 2 * kR, SH = kTY + 2 * kR;
  for (int e = tid; e < SW * SH; e += 256) {
    const int ry = e / SW, rx = e - ry * SW;
    const int y = ty0 - kR + ry, x = tx0 - kR + rx;
    if (x < 0 || x >= W || y < 0 || y >= H) continue;
    const SplatSrc<MODE> s = splat_source<MODE>(fs, fo, P, (size_t)y * W + x, x, y, t, eps);
    if (!s.finite || !s.is_short) continue;
    const float X = (float)x + s.fx, Y = (float)y + s.fy;
    const float fxf = floorf(X), fyf = floorf(Y);
    const int x0 = (int)fxf, y0 = (int)fyf;
    const float wx0 = (fxf + 1.f) - X, wx1 = X - fxf, wy0 = (fyf + 1.f) - Y, wy1 = Y - fyf;
    const float wgt[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
      if (cx < 0 || cx >= W || cy < 0 || cy >= H) continue;        // corner outside the image: dropped
      const int lx = cx - tx0, ly = cy - ty0;
      if (lx < 0 || lx >= kTX || ly < 0 || ly >= kTY) continue;    // corner belongs to another tile
      const int li = ly * kTX + lx;
#pragma unroll
      for (int c = 0; c < NV; ++c) atomicAdd(&acc[c][li], s.v[c] * wgt[k]);
      atomicAdd(&acc[NV][li], wgt[k]);
    }
  }
  __syncthreads();
  const float fill = (float)max(H, W);
  for (int i = tid; i < kTY * kTX; i += 256) {
    const int ly = i / kTX, lx = i - ly * kTX;
    const int y = ty0 + ly, x = tx0 + lx;
    if (x >= W || y >= H) continue;
    const size_t p = (size_t)y * W + x;
    const float *g = gacc + p * (NV + 1);
    const float sw = acc[NV][i] + g[NV];
    const float nrm = sw + 0.0000001f;
    const bool gap = (sw / nrm) < 0.999f;
    if (MODE == 0) {
      const float a0 = acc[0][i] + g[0], a1 = acc[1][i] + g[1];
      out[p] = (gap ? fill : -1.f * (a0 / nrm)) * 2.f;
      out[P + p] = (gap ? fill : -1.f * (a1 / nrm)) * 2.f;
    } else {
      const float a0 = acc[0][i] + g[0];
      const SplatSrc<MODE> s = splat_source<MODE>(fs, fo, P, p, x, y, t, eps);  // the unaligned value for holes
      out[p] = gap ? s.v[0] : a0 / nrm;
    }
  }
}

// ------------------------------------------------------------------------------------------ small elementwise
__global__ void __launch_bounds__(256) drm_ratio_kernel(const float *__restrict__ f10, const float *__restrict__ f12, float eps,
                                 float *__restrict__ r10, float *__restrict__ r12, int H, int W) {
  const int n = blockIdx.y;
  const size_t P = (size_t)H * W;
  f10 += (size_t)n * 2 * P;
  f12 += (size_t)n * 2 * P;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
    float a = sqrtf(f10[p] * f10[p] + f10[P + p] * f10[P + p]);
    float b = sqrtf(f12[p] * f12[p] + f12[P + p] * f12[P + p]);
    if (eps != 0.f) {
      a += eps;
      b += eps;
    }
    const float s = a + b;
    if (r10) r10[(size_t)n * P + p] = a / s;
    if (r12) r12[(size_t)n * P + p] = b / s;
  }
}

__global__ void __launch_bounds__(256) affine_kernel(const float *__restrict__ a, float mul, float add, float *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = a[i] * mul + add;
}

__global__ void __launch_bounds__(256) mul_map_kernel(const float *__restrict__ x, const float *__restrict__ map, float *__restrict__ out,
                               int C, size_t P) {
  const int n = blockIdx.y;
  x += (size_t)n * C * P;
  out += (size_t)n * C * P;
  map += (size_t)n * P;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
    const float m = map[p];
    for (int c = 0; c < C; ++c) out[(size_t)c * P + p] = x[(size_t)c * P + p] * m;
  }
}

__global__ void __launch_bounds__(256) fill_holes_kernel(const float *__restrict__ aligned, const float *__restrict__ cover,
                                  const float *__restrict__ value, float *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = (cover[i] < 0.999f) ? value[i] : aligned[i];
}

// drm.py:10-62.  The scalar bracket walk (double, like Python floats) is data independent;
// every lane replays it and applies the matching map update in fp32.
__global__ void __launch_bounds__(256) drm_retime_kernel(const float *__restrict__ drm, float *__restrict__ out, double t, double prec, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double x = 0.5, lo = 0.0, hi = 1.0;
    const double frac = 0.5;
    const float fm = drm[i];
    float xm = fm, lom = xm * 0.f, him = xm * 0.f + 1.f;
    int guard = 0;
    while (fabs(x - t) > prec && guard++ < 4096) {
      if (x > t) {
        hi = x;
        x = x - (x - lo) * frac;
        him = xm;
        xm = xm - (xm - lom) * fm;
      }
      if (x < t) {
        lo = x;
        x = x + (hi - x) * frac;
        lom = xm;
        xm = xm + (him - xm) * fm;
      }
    }
    out[i] = xm;
  }
}

}  // namespace

// ============================================================================================ C ABI
extern "C" {

size_t drba_softsplat_ws_floats(int N, int C, int H, int W) { return (size_t)N * H * W * (C + 1); }

int drba_softsplat(const float *in, const float *flow, const float *metric, float *out, float *ws, int N, int C,
                   int H, int W, int mode, int eps, void *stream) {
  if (!in || !flow || !out || !ws || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  if (mode < 0 || mode > 3 || eps < 0 || eps > 2) return DRBA_EINVAL;
  if (mode >= 2 && !metric) return DRBA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t P = (size_t)H * W;
  const int CP = mode == 0 ? C : C + 1;
  if (hipMemsetAsync(ws, 0, (size_t)N * P * CP * sizeof(float), s) != hipSuccess) return DRBA_ELAUNCH;
  dim3 g(grid_for(P), N);
  hipLaunchKernelGGL(splat_scatter, g, dim3(kBlock), 0, s, in, flow, metric, ws, C, H, W, mode);
  hipLaunchKernelGGL(splat_normalize, g, dim3(kBlock), 0, s, ws, out, C, H, W, mode, eps);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_backwarp(const float *in, const float *flow, float *out, int N, int C, int H, int W, int padding,
                  void *stream) {
  if (!in || !flow || !out || N <= 0 || C <= 0 || H <= 1 || W <= 1 || padding < 0 || padding > 1) return DRBA_EINVAL;
  dim3 g(grid_for((size_t)H * W), N);
  if (padding == 0)
    hipLaunchKernelGGL(backwarp_kernel<false>, g, dim3(kBlock), 0, (hipStream_t)stream, in, flow, out, C, H, W);
  else
    hipLaunchKernelGGL(backwarp_kernel<true>, g, dim3(kBlock), 0, (hipStream_t)stream, in, flow, out, C, H, W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_flow_distance(const float *flow, float *out, int N, int H, int W, void *stream) {
  if (!flow || !out || N <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  dim3 g(grid_for((size_t)H * W), N);
  hipLaunchKernelGGL(distance_kernel, g, dim3(kBlock), 0, (hipStream_t)stream, flow, out, H, W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_flow_reverse(const float *flow, float *out, float *ws, int N, int H, int W, void *stream) {
  if (!flow || !out || !ws || N <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t P = (size_t)H * W;
  if (hipMemsetAsync(ws, 0, (size_t)N * P * 3 * sizeof(float), s) != hipSuccess) return DRBA_ELAUNCH;
  hipLaunchKernelGGL(splat_long_prepass<0>, dim3(grid_for(P), N), dim3(kBlock), 0, s, flow, (const float *)nullptr, 0.f,
                     (const float *)nullptr, 0.f, ws, H, W);
  dim3 g((W + kTX - 1) / kTX, (H + kTY - 1) / kTY, N);
  hipLaunchKernelGGL(splat_tiled<0>, g, dim3(kBlock), 0, s, flow, (const float *)nullptr, 0.f, (const float *)nullptr,
                     0.f, ws, out, H, W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_drm_rife_linear(const float *flow_self, const float *flow_other, float t, const float *t_dev, float eps,
                         float *out, float *ws, int N, int H, int W, void *stream) {
  if (!flow_self || !flow_other || !out || !ws || N <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t P = (size_t)H * W;
  if (hipMemsetAsync(ws, 0, (size_t)N * P * 2 * sizeof(float), s) != hipSuccess) return DRBA_ELAUNCH;
  hipLaunchKernelGGL(splat_long_prepass<1>, dim3(grid_for(P), N), dim3(kBlock), 0, s, flow_self, flow_other, t, t_dev,
                     eps, ws, H, W);
  dim3 g((W + kTX - 1) / kTX, (H + kTY - 1) / kTY, N);
  hipLaunchKernelGGL(splat_tiled<1>, g, dim3(kBlock), 0, s, flow_self, flow_other, t, t_dev, eps, ws, out, H, W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_drm_ratio(const float *flow10, const float *flow12, float eps, float *drm10, float *drm12, int N, int H,
                   int W, void *stream) {
  if (!flow10 || !flow12 || (!drm10 && !drm12) || N <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  dim3 g(grid_for((size_t)H * W), N);
  hipLaunchKernelGGL(drm_ratio_kernel, g, dim3(kBlock), 0, (hipStream_t)stream, flow10, flow12, eps, drm10, drm12, H, W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_affine(const float *a, float mul, float add, float *out, size_t n, void *stream) {
  if (!a || !out || n == 0) return DRBA_EINVAL;
  hipLaunchKernelGGL(affine_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, a, mul, add, out, n);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_mul_map(const float *x, const float *map, float *out, int N, int C, int H, int W, void *stream) {
  if (!x || !map || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  dim3 g(grid_for((size_t)H * W), N);
  hipLaunchKernelGGL(mul_map_kernel, g, dim3(kBlock), 0, (hipStream_t)stream, x, map, out, C, (size_t)H * W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_fill_holes(const float *aligned, const float *cover, const float *value, float *out, size_t n, void *stream) {
  if (!aligned || !cover || !value || !out || n == 0) return DRBA_EINVAL;
  hipLaunchKernelGGL(fill_holes_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, aligned, cover, value, out, n);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_drm_retime(const float *drm, float *out, double t, double precision, size_t n, void *stream) {
  if (!drm || !out || n == 0 || !(precision > 0)) return DRBA_EINVAL;
  hipLaunchKernelGGL(drm_retime_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, drm, out, t, precision, n);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // extern "C"
