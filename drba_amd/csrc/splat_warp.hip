// Forward splat (softsplat), backward warp, flow distance and the fused RIFE/DRM splat
// pipelines.  All HBM-bound.  The generic splat is a counting sort by target pixel followed by a
// gather (no float atomics); the fused 1-2 channel RIFE splats are LDS-tiled: a workgroup owns an
// output tile, accumulates the source pixels around it with ds_add_f32 and stores the normalised
// result, and only flows longer than the tile halo use global fp32 atomics.
#include "common.hpp"

using namespace drba;

namespace {

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


// ------------------------------------------------------------------------------------------
// Generic softsplat for many channels (C >= gather_min_c()): sort once, gather per channel -- no float atomics.
// ds_add_f32 sustains only ~0.25 lane-adds per clock per CU (measured: run time ~ number of adds) and L2 float
// atomics far less, so with 4*(C+1) adds per source pixel the kernels above are add-bound for the 64..192-channel
// GMFSS feature pyramids.  Here the sources are counting-sorted by the pixel their footprint starts at
// (key = floor(target), one int atomic per source for the histogram and one for the slot), and every OUTPUT pixel
// then reads the four key segments whose 2x2 footprint covers it and sums weight * value over a chunk of channels:
// plain coalesced loads and stores, cost independent of the flow length, no halo limit, no overflow case.
// Keys live on a (W+1) x (H+1) grid because a footprint may start one pixel outside the image.
constexpr int kChunk = 16, kScanPerBlock = 4096;

struct __attribute__((aligned(16))) SplatRec {
  int src;      // source pixel index within the image
  float X, Y;   // target coordinates
  float m;      // 1, metric or exp(metric)
};

__device__ __forceinline__ float splat_weight_1d(float X, float fl, int d) { return d ? X - fl : (fl + 1.f) - X; }

// key of source pixel (x, y), or -1 when it contributes nothing (non-finite target or footprint outside the image)
__device__ __forceinline__ long long splat_key(const float *__restrict__ flow, size_t P, size_t p, int x, int y, int H, int W,
                                               float &X, float &Y) {
  X = (float)x + flow[p];
  Y = (float)y + flow[P + p];
  if (!(isfinite(X) && isfinite(Y))) return -1;
  const float fx = floorf(X), fy = floorf(Y);
  if (fx < -1.f || fx > (float)(W - 1) || fy < -1.f || fy > (float)(H - 1)) return -1;
  return (long long)((int)fy + 1) * (W + 1) + ((int)fx + 1);
}

__global__ void __launch_bounds__(256) splat_sort_count(const float *__restrict__ flow, int *__restrict__ cnt, int H, int W) {
  const int n = blockIdx.y;
  const size_t P = (size_t)H * W, Pk = (size_t)(H + 1) * (W + 1);
  flow += (size_t)n * 2 * P;
  cnt += (size_t)n * Pk;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
    float X, Y;
    const long long k = splat_key(flow, P, p, x, y, H, W, X, Y);
    if (k >= 0) atomicAdd(&cnt[k], 1);
  }
}

// exclusive scan of cnt[0..L) into start[0..L], three small kernels (block-local scan, scan of block sums, add)
__global__ void __launch_bounds__(256) scan_block(const int *__restrict__ cnt, int *__restrict__ start, int *__restrict__ bsum, size_t L) {
  __shared__ int wsum[4];
  const size_t base = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * 16;
  int v[16], tot = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] = base + i < L ? cnt[base + i] : 0;
    tot += v[i];
  }
  int inc = tot;  // inclusive scan over the wave, then over the 4 waves
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if ((threadIdx.x & 63) >= d) inc += o;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
  int run = woff + inc - tot;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (base + i < L) start[base + i] = run;
    run += v[i];
  }
  if (threadIdx.x == 255) bsum[blockIdx.x] = woff + inc;
}

__global__ void __launch_bounds__(256) scan_sums(int *__restrict__ bsum, int nb, int *__restrict__ total) {
  __shared__ int wsum[4];
  int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 256) {
    const int i = b0 + threadIdx.x;
    const int v = i < nb ? bsum[i] : 0;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if ((threadIdx.x & 63) >= d) inc += o;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
    if (i < nb) bsum[i] = carry + woff + inc - v;
    carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(256) scan_add(int *__restrict__ start, const int *__restrict__ bsum, size_t L) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < L) start[i] += bsum[i / kScanPerBlock];
}

__global__ void __launch_bounds__(256)
splat_sort_fill(const float *__restrict__ flow, const float *__restrict__ metric, const int *__restrict__ start,
                int *__restrict__ cnt, SplatRec *__restrict__ rec, int H, int W, int mode) {
  const int n = blockIdx.y;
  const size_t P = (size_t)H * W, Pk = (size_t)(H + 1) * (W + 1);
  flow += (size_t)n * 2 * P;
  if (metric) metric += (size_t)n * P;
  start += (size_t)n * Pk;  // one scan over all images: segment positions are global, rec is shared
  cnt += (size_t)n * Pk;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
    SplatRec r;
    const long long k = splat_key(flow, P, p, x, y, H, W, r.X, r.Y);
    if (k < 0) continue;
    r.src = (int)p;
    r.m = 1.f;
    if (mode == 2) r.m = metric[p];
    if (mode == 3) r.m = expf(metric[p]);
    const int slot = start[k] + atomicSub(&cnt[k], 1) - 1;
    rec[slot] = r;
  }
}

__global__ void __launch_bounds__(256)
splat_sorted_gather(const float *__restrict__ in, const int *__restrict__ start, const SplatRec *__restrict__ rec,
                    float *__restrict__ out, int C, int H, int W, int mode, int eps, int chunks) {
  const int n = blockIdx.y / chunks, chunk = blockIdx.y - n * chunks;
  const int c0 = chunk * kChunk, gc = min(kChunk, C - c0);
  const size_t P = (size_t)H * W, Pk = (size_t)(H + 1) * (W + 1);
  in += ((size_t)n * C + c0) * P;
  start += (size_t)n * Pk;
  out += ((size_t)n * C + c0) * P;
  const Tile2D tp = tile_pixel(W, H);
  if (!tp.valid) return;
  const int x = tp.x, y = tp.y;
  float acc[kChunk];
#pragma unroll
  for (int c = 0; c < kChunk; ++c) acc[c] = 0.f;
  float nrm = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    // footprints starting at (x-1, y-dy) [dx = 1] and (x, y-dy) [dx = 0] have adjacent keys: one contiguous range
    const size_t k1 = (size_t)(y - dy + 1) * (W + 1) + x;  // key of (x-1, y-dy)
    const int s0 = start[k1], s1 = start[k1 + 1], s2 = start[k1 + 2];
    for (int s = s0; s < s2; ++s) {
      const SplatRec r = rec[s];
      const int dx = s < s1 ? 1 : 0;
      const float w = r.m * (splat_weight_1d(r.X, floorf(r.X), dx) * splat_weight_1d(r.Y, floorf(r.Y), dy));
      nrm += w;
#pragma unroll
      for (int c = 0; c < kChunk; ++c) acc[c] += in[(size_t)min(c, gc - 1) * P + r.src] * w;  // branch-free: tail repeats
    }
  }
  if (mode != 0) {
    if (eps == 0) nrm = nrm + 0.0000001f;
    else if (eps == 1) nrm = (nrm == 0.f) ? 1.f : nrm;
    else nrm = fmaxf(nrm, 0.0000001f) + (nrm != nrm ? nrm : 0.f);  // clip(min=1e-7); NaN stays NaN
  }
  const size_t p = (size_t)y * W + x;
#pragma unroll
  for (int c = 0; c < kChunk; ++c)
    if (c < gc) out[(size_t)c * P + p] = mode == 0 ? acc[c] : acc[c] / nrm;
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

static size_t sort_keys(int N, int H, int W) { return (size_t)N * (H + 1) * (W + 1); }

size_t drba_softsplat_ws_floats(int N, int C, int H, int W) {
  (void)C;  // [cnt: L][start: L + 1][block sums][rec: 4 * N*P]
  const size_t L = sort_keys(N, H, W);
  return 2 * L + 8 + (L + kScanPerBlock - 1) / kScanPerBlock + 8 + 4 * (size_t)N * H * W;
}

int drba_softsplat(const float *in, const float *flow, const float *metric, float *out, float *ws, int N, int C,
                   int H, int W, int mode, int eps, void *stream) {
  if (!in || !flow || !out || !ws || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  if (mode < 0 || mode > 3 || eps < 0 || eps > 2) return DRBA_EINVAL;
  if (mode >= 2 && !metric) return DRBA_EINVAL;
  if ((size_t)N * (H + 1) * (W + 1) >= (1u << 31)) return DRBA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t P = (size_t)H * W;
  const size_t L = sort_keys(N, H, W);
  const int nb = (int)((L + kScanPerBlock - 1) / kScanPerBlock);
  int *cnt = (int *)ws;
  int *start = cnt + L;                                // L + 1 entries
  int *bsum = start + ((L + 1 + 3) & ~(size_t)3);      // nb entries
  SplatRec *rec = (SplatRec *)(((uintptr_t)(bsum + nb) + 15) & ~(uintptr_t)15);
  if (hipMemsetAsync(cnt, 0, L * sizeof(int), s) != hipSuccess) return DRBA_ELAUNCH;
  hipLaunchKernelGGL(splat_sort_count, dim3(grid_for(P), N), dim3(kBlock), 0, s, flow, cnt, H, W);
  hipLaunchKernelGGL(scan_block, dim3(nb), dim3(kBlock), 0, s, cnt, start, bsum, L);
  hipLaunchKernelGGL(scan_sums, dim3(1), dim3(kBlock), 0, s, bsum, nb, start + L);
  hipLaunchKernelGGL(scan_add, dim3((unsigned)((L + 255) / 256)), dim3(kBlock), 0, s, start, bsum, L);
  hipLaunchKernelGGL(splat_sort_fill, dim3(grid_for(P), N), dim3(kBlock), 0, s, flow, metric, start, cnt, rec, H, W, mode);
  const int chunks = (C + kChunk - 1) / kChunk;
  hipLaunchKernelGGL(splat_sorted_gather, dim3(tiles_for(W, H), N * chunks), dim3(kBlock), 0, s, in, start, rec, out, C, H,
                     W, mode, eps, chunks);
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
