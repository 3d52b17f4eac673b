// Forward splat (softsplat), backward warp, flow distance and the fused RIFE/DRM splat
// pipelines.  All HBM-bound.  The generic splat is a counting sort by target pixel followed by a
// gather (no float atomics); the fused 1-2 channel RIFE splats are LDS-tiled: a workgroup owns an
// output tile, accumulates the source pixels around it with ds_add_f32 and stores the normalised
// result, and only flows longer than the tile halo use global fp32 atomics.
#include "common.hpp"

#include <string.h>

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
                                            const float (&v)[NV], int H, int W, int *__restrict__ dirty = nullptr,
                                            int tiles_x = 0) {
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
    // the 32 x 16 output tile this corner lies in has something to collect from the accumulator (kSX, kSY below)
    if (dirty) atomicOr(dirty + ((y0 + (k >> 1)) >> 4) * tiles_x + ((x0 + (k & 1)) >> 5), 1);
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
#ifndef DRBA_SPLAT_R
#define DRBA_SPLAT_R 16
#endif
constexpr int kRMax = DRBA_SPLAT_R;  // largest tile halo: sources with longer flows take the global-atomic pre-pass

template <int MODE>
struct SplatSrc {
  float v[2];
  float fx, fy;
  bool finite, is_short;
};

// value and splat flow of a source from its raw flow vectors (su, sv) [and (ou, ov) of the other direction]
template <int MODE>
__device__ __forceinline__ SplatSrc<MODE> splat_source_from(float su, float sv, float ou, float ov, int x, int y, float t,
                                                            float eps) {
  SplatSrc<MODE> s;
  if (MODE == 0) {
    s.v[0] = su;
    s.v[1] = sv;
    s.fx = su;
    s.fy = sv;
  } else {
    const float ds = sqrtf(su * su + sv * sv) + eps, d_o = sqrtf(ou * ou + ov * ov) + eps;
    const float u = (d_o / (ds + d_o)) * t * 2.f;
    s.v[0] = u;
    s.v[1] = 0.f;
    s.fx = su * u;
    s.fy = sv * u;
  }
  const float X = (float)x + s.fx, Y = (float)y + s.fy;
  s.finite = isfinite(X) && isfinite(Y);
  // all four corners within kRMax of the source <=> floor(f) >= -kRMax and floor(f)+1 <= kRMax
  s.is_short = s.fx >= -(float)kRMax && s.fx < (float)(kRMax - 1) && s.fy >= -(float)kRMax && s.fy < (float)(kRMax - 1);
  return s;
}

template <int MODE>
__device__ __forceinline__ SplatSrc<MODE> splat_source(const float *__restrict__ fs, const float *__restrict__ fo,
                                                       size_t P, size_t p, int x, int y, float t, float eps) {
  const float su = fs[p], sv = fs[P + p];
  float ou = 0.f, ov = 0.f;
  if (MODE != 0) {
    ou = fo[p];
    ov = fo[P + p];
  }
  return splat_source_from<MODE>(su, sv, ou, ov, x, y, t, eps);
}

// Several (flow_self, flow_other, t) -> out jobs of ONE geometry in one launch (drba_drm_rife_linear_batch): the DRM maps of a
// group of steps were 2 launches each, 16 dependent launches of 20-55 us per group with ~6.5 us between them.  n == 0: the
// single-tensor form (items contiguous along N).
struct SplatJobs {
  int n;
  const float *fs[DRBA_MAX_STAGE_ITEMS], *fo[DRBA_MAX_STAGE_ITEMS];
  float t[DRBA_MAX_STAGE_ITEMS];
  float *out[DRBA_MAX_STAGE_ITEMS];
};

// Workspace of the two fused splats (drba_rife_splat_ws_floats): [kReachWords ints: the reach map, scratch -- any content on entry
// and on return][accumulators [N][P][NV + 1], zero on entry and on return][one dirty flag per (item, tile), zero on entry and on return].
// The reach map has a FIXED place in front so that no other geometry's accumulators ever lie over it: the entry points share one
// zeroed buffer (ops._zero_workspace) and the map is not zeroed after use.
constexpr int kReachWords = 1 << 18;  // 1 MB: 8 items of a 4K map are 130 560 tiles; larger launches run without the map

// pre-pass, one workgroup per 32 x 16 tile of SOURCES: long (but finite) pixels -> global atomics into gacc [P][NV+1]; the
// short ones -> the tile's reach, the smallest halo R with all four corners of each within R of its source (what
// splat_source_from's is_short tests against kR): an output tile whose 3 x 3 neighbourhood of source tiles has reach <= R
// receives nothing from beyond R pixels and scans a (32 + 2R) x (16 + 2R) window instead of 64 x 48.
template <int MODE>
__global__ void __launch_bounds__(256) splat_long_prepass(const float *__restrict__ fs, const float *__restrict__ fo,
                                                          float t, const float *__restrict__ t_dev, float eps,
                                                          float *__restrict__ ws, int H, int W, const SplatJobs jobs) {
  constexpr int NV = MODE == 0 ? 2 : 1;
  if (t_dev) t = *t_dev;  // timestep from device memory: lets one captured HIP graph serve every t
  const int n = blockIdx.z;
  const size_t P = (size_t)H * W;
  if (jobs.n) {
    fs = jobs.fs[n], fo = jobs.fo[n], t = jobs.t[n];
  } else {
    fs += (size_t)n * 2 * P;
    if (fo) fo += (size_t)n * 2 * P;
  }
  const int tiles_x = gridDim.x, tiles = tiles_x * gridDim.y;
  int *reach = reinterpret_cast<int *>(ws);
  float *gacc = ws + kReachWords;
  int *dirty = reinterpret_cast<int *>(gacc + (size_t)gridDim.z * P * (NV + 1)) + (size_t)n * tiles;
  gacc += (size_t)n * P * (NV + 1);
  const int tid = threadIdx.x;
  int r = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int x = blockIdx.x * 32 + (tid & 31), y = blockIdx.y * 16 + (tid >> 5) + 8 * i;
    if (x >= W || y >= H) continue;
    const size_t p = (size_t)y * W + x;
    const SplatSrc<MODE> s = splat_source<MODE>(fs, fo, P, p, x, y, t, eps);
    if (!s.finite) continue;
    if (s.is_short) {
      // fx in [-R, R - 1)  <=>  R >= -fx and R > fx + 1
      const int rx = s.fx < 0.f ? (int)ceilf(-s.fx) : (int)floorf(s.fx) + 2;
      const int ry = s.fy < 0.f ? (int)ceilf(-s.fy) : (int)floorf(s.fy) + 2;
      r = max(r, max(rx, ry));
      continue;
    }
    if (MODE == 0) {
      const float v[2] = {s.v[0], s.v[1]};
      scatter_avg<2>(gacc, x, y, s.fx, s.fy, v, H, W, dirty, tiles_x);
    } else {
      const float v[1] = {s.v[0]};
      scatter_avg<1>(gacc, x, y, s.fx, s.fy, v, H, W, dirty, tiles_x);
    }
  }
  if ((size_t)gridDim.z * tiles > (size_t)kReachWords) return;  // (no room for the map: the tiles scan the full halo)
  __shared__ int wr[4];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) r = max(r, __shfl_xor(r, d, 64));
  if ((tid & 63) == 0) wr[tid >> 6] = r;
  __syncthreads();
  if (tid == 0) reach[(size_t)n * tiles + blockIdx.y * tiles_x + blockIdx.x] = max(max(wr[0], wr[1]), max(wr[2], wr[3]));
}

// Tile kernel.  On gfx950 ds_add_f32 retires ~0.33 lane-adds per clock per CU while integer LDS atomics retire ~13
// (tools/exp/lds_atomic_rate.hip), so the tile does not accumulate floats atomically: it counting-sorts the sources
// that touch it by the output pixel their 2x2 footprint starts at (one ds_add_rtn_u32 per source gives the rank, a
// block scan the segment starts), stores their records in LDS, and every output pixel then sums its four segments
// with plain LDS reads.  Sources beyond the LDS record capacity (strong local convergence) fall back to the same
// global-atomic accumulator the long-flow pre-pass uses.
constexpr int kSX = 32, kSY = 16;                       // output tile of the sorted kernel
static_assert(kRMax <= kSY, "the reach map reads the 3 x 3 tiles around an output tile: a halo may not span more than one tile (DRBA_SPLAT_R)");
constexpr int kKX = kSX + 1, kKY = kSY + 1;             // key grid: footprint origins (-1..kSX-1) x (-1..kSY-1)
constexpr int kNKEY = kKX * kKY;
constexpr int kCAP = 1536;                              // records held in LDS (3 per output pixel)

// R: the halo this tile scans (kR, or less where the reach map allows it)
template <int MODE, int R>
__device__ __forceinline__ void splat_tile_body(const float *__restrict__ fs, const float *__restrict__ fo, float t, float eps,
                                                float *gacc, int *dirty, float *__restrict__ out, int H, int W, int *cnt,
                                                int *wsum, float4 *rec, int *spilled_) {
  constexpr int NV = MODE == 0 ? 2 : 1;
  constexpr int kSWX = kSX + 2 * R, kSWY = kSY + 2 * R;  // source window
  constexpr int kSPT = (kSWX * kSWY + 255) / 256;        // window sources per thread
  constexpr int kR = R;                                  // (the window arithmetic below)
  volatile int &spilled = *spilled_;
  const size_t P = (size_t)H * W;
  const int was_dirty = __builtin_nontemporal_load(dirty);
  const int tx0 = blockIdx.x * kSX, ty0 = blockIdx.y * kSY;
  const int tid = threadIdx.x;
  for (int i = tid; i <= kNKEY; i += 256) cnt[i] = 0;
  if (tid == 0) spilled = 0;
  __syncthreads();

  // pass 1: all of this thread's window sources are loaded first (independent loads, issued back to back), then
  // each gets its key and, from one ds_add_rtn_u32, its rank within the key
  float raw[kSPT][MODE == 0 ? 2 : 4];
#pragma unroll
  for (int i = 0; i < kSPT; ++i) {
    const int e = min(tid + i * 256, kSWX * kSWY - 1);
    const int ry = e / kSWX, rx = e - ry * kSWX;
    const int y = min(max(ty0 - kR + ry, 0), H - 1), x = min(max(tx0 - kR + rx, 0), W - 1);  // clamped: always loadable
    const size_t p = (size_t)y * W + x;
    raw[i][0] = fs[p];
    raw[i][1] = fs[P + p];
    if (MODE != 0) {
      raw[i][2] = fo[p];
      raw[i][3] = fo[P + p];
    }
  }
  int kr[kSPT];  // key << 16 | rank, or -1
  float4 val[kSPT];
#pragma unroll
  for (int i = 0; i < kSPT; ++i) {
    kr[i] = -1;
    const int e = tid + i * 256;
    const int ry = e / kSWX, rx = e - ry * kSWX;
    const int y = ty0 - kR + ry, x = tx0 - kR + rx;
    if (e >= kSWX * kSWY || x < 0 || x >= W || y < 0 || y >= H) continue;
    const SplatSrc<MODE> sp = splat_source_from<MODE>(raw[i][0], raw[i][1], MODE == 0 ? 0.f : raw[i][2],
                                                      MODE == 0 ? 0.f : raw[i][3], x, y, t, eps);
    if (!sp.finite || !sp.is_short) continue;
    const float X = (float)x + sp.fx, Y = (float)y + sp.fy;
    const int kx = (int)floorf(X) - tx0 + 1, ky = (int)floorf(Y) - ty0 + 1;
    if (kx < 0 || kx >= kKX || ky < 0 || ky >= kKY) continue;
    const int key = ky * kKX + kx;
    kr[i] = (key << 16) | (atomicAdd(&cnt[key], 1) & 0xFFFF);
    val[i] = make_float4(sp.v[0], sp.v[1], X - (float)tx0, Y - (float)ty0);
  }
  __syncthreads();

  // exclusive scan of cnt[0..kNKEY) in place (3 keys per thread), total -> cnt[kNKEY]
  {
    constexpr int PER = (kNKEY + 255) / 256;
    int v[PER], tot = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int k = tid * PER + j;
      v[j] = k < kNKEY ? cnt[k] : 0;
      tot += v[j];
    }
    int inc = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if ((tid & 63) >= d) inc += o;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    int run = inc - tot;
    for (int w = 0; w < (tid >> 6); ++w) run += wsum[w];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int k = tid * PER + j;
      if (k < kNKEY) cnt[k] = run;
      run += v[j];
    }
    if (tid == 255) cnt[kNKEY] = run;
  }
  __syncthreads();

  // pass 2: place the records
#pragma unroll
  for (int i = 0; i < kSPT; ++i) {
    if (kr[i] < 0) continue;
    const int slot = cnt[kr[i] >> 16] + (kr[i] & 0xFFFF);
    if (slot < kCAP) {
      rec[slot] = val[i];
    } else {  // LDS record space exhausted: the corners of this source that lie in THIS tile go through the global
              // accumulator (a neighbouring tile handles its own corners of the same source)
      const float X = val[i].z + (float)tx0, Y = val[i].w + (float)ty0;
      const float fl_x = floorf(X), fl_y = floorf(Y);
      const int x0 = (int)fl_x, y0 = (int)fl_y;
      spilled = 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
        if (cx < tx0 || cx >= tx0 + kSX || cy < ty0 || cy >= ty0 + kSY || cx >= W || cy >= H) continue;
        const float w = ((k & 1) ? X - fl_x : (fl_x + 1.f) - X) * ((k >> 1) ? Y - fl_y : (fl_y + 1.f) - Y);
        float *d = gacc + ((size_t)cy * W + cx) * (NV + 1);
        atomic_add_f32(d, val[i].x * w);
        if (NV == 2) atomic_add_f32(d + 1, val[i].y * w);
        atomic_add_f32(d + NV, w);
      }
    }
  }
  // overflow adds (L2 atomics) complete before this tile reads its accumulator entries back; a workgroup-scope fence
  // is enough (same CU, lines not yet in L1) -- an agent-scope fence writes the XCD's L2 back from every workgroup
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __syncthreads();

#ifndef DRBA_SPLAT_ALWAYS_COLLECT  // A/B builds: every tile reads its accumulator entries, as before ABI 7
#define DRBA_SPLAT_ALWAYS_COLLECT 0
#endif
  const bool collect = DRBA_SPLAT_ALWAYS_COLLECT || was_dirty != 0 || spilled != 0;  // uniform over the workgroup
  if (was_dirty && tid == 0) *dirty = 0;                // self-cleaning, like the accumulator itself
  // gather: output pixel (lx, ly) takes, for dy in {0,1}, the segments of keys (lx, ly-dy+1) [dx = 1] and (lx+1, ly-dy+1) [dx = 0]
  const float fill = (float)max(H, W);
  for (int i = tid; i < kSX * kSY; i += 256) {
    const int ly = i / kSX, lx = i - ly * kSX;
    const int y = ty0 + ly, x = tx0 + lx;
    if (x >= W || y >= H) continue;
    float a0 = 0.f, a1 = 0.f, sw = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int k1 = (ly - dy + 1) * kKX + lx;
      const int s0 = cnt[k1], s1 = cnt[k1 + 1], s2 = min(cnt[k1 + 2], kCAP);
      for (int s_ = s0; s_ < s2; ++s_) {
        const float4 r = rec[s_];
        const int dx = s_ < s1 ? 1 : 0;
        const float X = r.z, Y = r.w;  // tile-relative target: floor and fraction are shift-invariant (tx0, ty0 integers)
        const float fl_x = floorf(X), fl_y = floorf(Y);
        const float wx = dx ? X - fl_x : (fl_x + 1.f) - X;
        const float wy = dy ? Y - fl_y : (fl_y + 1.f) - Y;
        const float w = wx * wy;
        a0 += r.x * w;
        if (NV == 2) a1 += r.y * w;
        sw += w;
      }
    }
    const size_t p = (size_t)y * W + x;
    float g0 = 0.f, g1 = 0.f, gw = 0.f;
    if (collect) {
      float *g = gacc + p * (NV + 1);
      g0 = g[0], g1 = NV == 2 ? g[1] : 0.f, gw = g[NV];
      if (g0 != 0.f || g1 != 0.f || gw != 0.f) {  // self-cleaning accumulator: zero on entry, zero again on return
        g[0] = 0.f;                               // (no memset launch per call; nothing is written in the usual case)
        if (NV == 2) g[1] = 0.f;
        g[NV] = 0.f;
      }
    }
    sw += gw;
    const float nrm = sw + 0.0000001f;
    const bool gap = (sw / nrm) < 0.999f;
    if (MODE == 0) {
      a0 += g0;
      a1 += g1;
      out[p] = (gap ? fill : -1.f * (a0 / nrm)) * 2.f;
      out[P + p] = (gap ? fill : -1.f * (a1 / nrm)) * 2.f;
    } else {
      a0 += g0;
      float o = a0 / nrm;
      if (gap) o = splat_source<MODE>(fs, fo, P, p, x, y, t, eps).v[0];  // the unaligned value for holes (rare: not loaded otherwise)
      out[p] = o;
    }
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) splat_tiled(const float *__restrict__ fs, const float *__restrict__ fo, float t,
                                                   const float *__restrict__ t_dev, float eps, float *ws,
                                                   float *__restrict__ out, int H, int W, const SplatJobs jobs) {
  constexpr int NV = MODE == 0 ? 2 : 1;
  if (t_dev) t = *t_dev;
  __shared__ int cnt[kNKEY + 1];   // per key: count, then (after the scan) segment start; [kNKEY] = total
  __shared__ int wsum[4];
  __shared__ float4 rec[kCAP];     // (v0, v1, X - tx0, Y - ty0) of a source, grouped by key
  __shared__ int spilled;          // this tile put records beyond kCAP into the global accumulator
  static_assert(kSX == 32 && kSY == 16, "scatter_avg and the pre-pass work on 32 x 16 tiles");
  const int n = blockIdx.z;
  const size_t P = (size_t)H * W;
  if (jobs.n) {
    fs = jobs.fs[n], fo = jobs.fo[n], t = jobs.t[n], out = jobs.out[n];
  } else {
    fs += (size_t)n * 2 * P;
    if (fo) fo += (size_t)n * 2 * P;
    out += (size_t)n * (MODE == 0 ? 2 : 1) * P;
  }
  const int tiles_x = gridDim.x, tiles_y = gridDim.y, tiles = tiles_x * tiles_y;
  const int *reach = reinterpret_cast<const int *>(ws) + (size_t)n * tiles;
  float *gacc = ws + kReachWords;
  // dirty[tile] != 0: the long-flow pre-pass scattered into this tile.  Only then (or after a spill of its own) does the
  // tile read its accumulator entries -- 8-12 bytes per output pixel that are zero in all but a few tiles of a frame.
  int *dirty = reinterpret_cast<int *>(gacc + (size_t)gridDim.z * P * (NV + 1)) + ((size_t)n * tiles_y + blockIdx.y) * tiles_x + blockIdx.x;
  gacc += (size_t)n * P * (NV + 1);
  // the halo: what the source tiles around this one reach (all of them lie within kRMax = 16 <= a tile's smaller side)
  int r = kRMax;
  if ((size_t)gridDim.z * tiles <= (size_t)kReachWords) {
    r = 0;
    const int bx = blockIdx.x, by = blockIdx.y;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = by + dy, xx = bx + dx;
        if (yy >= 0 && yy < tiles_y && xx >= 0 && xx < tiles_x) r = max(r, __builtin_nontemporal_load(reach + yy * tiles_x + xx));
      }
    r = __builtin_amdgcn_readfirstlane(r);
  }
  if (r <= 4) splat_tile_body<MODE, 4>(fs, fo, t, eps, gacc, dirty, out, H, W, cnt, wsum, rec, &spilled);
  else if (r <= 8) splat_tile_body<MODE, 8>(fs, fo, t, eps, gacc, dirty, out, H, W, cnt, wsum, rec, &spilled);
  else splat_tile_body<MODE, kRMax>(fs, fo, t, eps, gacc, dirty, out, H, W, cnt, wsum, rec, &spilled);
}


// ------------------------------------------------------------------------------------------
// Generic softsplat (softsplat.py:248-367 == softsplat_torch.py:19-179; mode 0 sum, 1 avg, 2 linear, 3 soft):
// sort once, gather per channel -- no float atomics.
// ds_add_f32 sustains only ~0.3 lane-adds per clock per CU and L2 float atomics far less, so a scatter with
// 4*(C+1) adds per source pixel is add-bound for the 64..192-channel GMFSS feature pyramids.  Here the sources are counting-sorted by the pixel their footprint starts at
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

// Feature tensors (C >= 16, C % 4 == 0): the gather's cost is the NUMBER of source loads (one per record per channel,
// each a scattered 4-byte read), so the source is first rewritten channel-quad interleaved ([C/4][P][4]: one coalesced
// pass) and a record then fetches 4 channels with one 16-byte load.  Same products per channel as
// splat_sorted_gather.
__global__ void __launch_bounds__(256) quad_interleave_kernel(const float *__restrict__ in, float *__restrict__ out, int C4, size_t P) {
  const size_t total = (size_t)C4 * P;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t q = i / P, p = i - q * P;
    const float *src = in + (q * 4) * P + p;
    *reinterpret_cast<float4 *>(out + i * 4) = make_float4(src[0], src[P], src[2 * P], src[3 * P]);
  }
}

__global__ void __launch_bounds__(256)
splat_sorted_gather_quad(const float *__restrict__ inq, const int *__restrict__ start, const SplatRec *__restrict__ rec,
                         float *__restrict__ out, int C, int H, int W, int mode, int eps, int chunks) {
  const int n = blockIdx.y / chunks, chunk = blockIdx.y - n * chunks;
  const int c0 = chunk * kChunk, gq = min(kChunk, C - c0) / 4;  // channel quads in this chunk (C % 4 == 0)
  const size_t P = (size_t)H * W, Pk = (size_t)(H + 1) * (W + 1);
  const float4 *src = reinterpret_cast<const float4 *>(inq) + ((size_t)n * (C / 4) + c0 / 4) * P;
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
    const size_t k1 = (size_t)(y - dy + 1) * (W + 1) + x;  // key of (x-1, y-dy), see splat_sorted_gather
    const int s0 = start[k1], s1 = start[k1 + 1], s2 = start[k1 + 2];
    for (int s = s0; s < s2; ++s) {
      const SplatRec r = rec[s];
      const int dx = s < s1 ? 1 : 0;
      const float w = r.m * (splat_weight_1d(r.X, floorf(r.X), dx) * splat_weight_1d(r.Y, floorf(r.Y), dy));
      nrm += w;
#pragma unroll
      for (int q = 0; q < kChunk / 4; ++q) {
        const float4 v = src[(size_t)min(q, gq - 1) * P + r.src];  // branch-free: tail quads repeat the last one
        acc[4 * q] += v.x * w, acc[4 * q + 1] += v.y * w, acc[4 * q + 2] += v.z * w, acc[4 * q + 3] += v.w * w;
      }
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
    if (c < 4 * gq) out[(size_t)c * P + p] = mode == 0 ? acc[c] : acc[c] / nrm;
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

static bool splat_quad(int C) { return C >= 16 && (C & 3) == 0; }

size_t drba_softsplat_ws_floats(int N, int C, int H, int W) {
  // [cnt: L][start: L + 1][block sums][rec: 4 * N*P][quad-interleaved copy of the input: N*C*P, feature tensors only]
  const size_t L = sort_keys(N, H, W);
  return 2 * L + 8 + (L + kScanPerBlock - 1) / kScanPerBlock + 8 + 4 * (size_t)N * H * W + 4 +
         (splat_quad(C) ? (size_t)N * C * H * W : 0);
}

// ws layout: [cnt: L][start: L + 1][block sums: nb][rec: N * P][quad-interleaved copy of the input: N*C*P, feature tensors only]
struct SplatWs {
  int *cnt, *start, *bsum;
  SplatRec *rec;
  float *inq;
  size_t L;
  int nb;
};
static SplatWs splat_ws(float *ws, int N, int H, int W) {
  SplatWs w;
  w.L = sort_keys(N, H, W);
  w.nb = (int)((w.L + kScanPerBlock - 1) / kScanPerBlock);
  w.cnt = (int *)ws;
  w.start = w.cnt + w.L;                                  // L + 1 entries
  w.bsum = w.start + ((w.L + 1 + 3) & ~(size_t)3);        // nb entries
  w.rec = (SplatRec *)(((uintptr_t)(w.bsum + w.nb) + 15) & ~(uintptr_t)15);
  w.inq = (float *)(((uintptr_t)(w.rec + (size_t)N * H * W) + 15) & ~(uintptr_t)15);
  return w;
}
// the gather of one input through the index in ws
static int splat_gather(const float *in, float *out, const SplatWs &w, int N, int C, int H, int W, int mode, int eps, hipStream_t s) {
  const size_t P = (size_t)H * W;
  const int chunks = (C + kChunk - 1) / kChunk;
  if (splat_quad(C)) {
    DRBA_LAUNCH(quad_interleave_kernel, dim3(grid_for((size_t)N * (C / 4) * P)), dim3(kBlock), 0, s, in, w.inq, N * (C / 4), P);
    DRBA_LAUNCH(splat_sorted_gather_quad, dim3(tiles_for(W, H), N * chunks), dim3(kBlock), 0, s, w.inq, w.start, w.rec, out, C,
                       H, W, mode, eps, chunks);
  } else {
    DRBA_LAUNCH(splat_sorted_gather, dim3(tiles_for(W, H), N * chunks), dim3(kBlock), 0, s, in, w.start, w.rec, out, C, H,
                       W, mode, eps, chunks);
  }
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_softsplat(const float *in, const float *flow, const float *metric, float *out, float *ws, int N, int C,
                   int H, int W, int mode, int eps, void *stream) {
  if (!in || !flow || !out || !ws || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  if (mode < 0 || mode > 3 || eps < 0 || eps > 2) return DRBA_EINVAL;
  if (mode >= 2 && !metric) return DRBA_EINVAL;
  if ((size_t)N * (H + 1) * (W + 1) >= (1u << 31)) return DRBA_EINVAL;
  const int rc = drba_softsplat_index(flow, metric, ws, N, H, W, mode, stream);
  if (rc != DRBA_OK) return rc;
  return splat_gather(in, out, splat_ws(ws, N, H, W), N, C, H, W, mode, eps, (hipStream_t)stream);
}

// ABI 9: the three pieces of drba_softsplat for a caller that keeps the quad-interleaved copy of a feature tensor ([N][C/4][H*W][4],
// drba_quad_interleave) across calls -- GMFSS splats every pyramid level of a frame for each output frame of two steps, and
// drba_softsplat / drba_softsplat_again rewrite the copy into the workspace every time (12 x 28 us per 1080p step).
int drba_quad_interleave(const float *in, float *out, int N, int C, int H, int W, void *stream) {
  if (!in || !out || N <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0) return DRBA_EINVAL;
  const size_t P = (size_t)H * W;
  DRBA_LAUNCH(quad_interleave_kernel, dim3(grid_for((size_t)N * (C / 4) * P)), dim3(kBlock), 0, (hipStream_t)stream, in, out, N * (C / 4), P);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

// the sorted index of (flow, metric, mode) into ws (drba_softsplat_ws_floats(N, any C, H, W) floats), no gather
int drba_softsplat_index(const float *flow, const float *metric, float *ws, int N, int H, int W, int mode, void *stream) {
  if (!flow || !ws || N <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  if (mode < 0 || mode > 3) return DRBA_EINVAL;
  if (mode >= 2 && !metric) return DRBA_EINVAL;
  if ((size_t)N * (H + 1) * (W + 1) >= (1u << 31)) return DRBA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t P = (size_t)H * W;
  const SplatWs w = splat_ws(ws, N, H, W);
  if (hipMemsetAsync(w.cnt, 0, w.L * sizeof(int), s) != hipSuccess) return DRBA_ELAUNCH;
  DRBA_LAUNCH(splat_sort_count, dim3(grid_for(P), N), dim3(kBlock), 0, s, flow, w.cnt, H, W);
  DRBA_LAUNCH(scan_block, dim3(w.nb), dim3(kBlock), 0, s, w.cnt, w.start, w.bsum, w.L);
  DRBA_LAUNCH(scan_sums, dim3(1), dim3(kBlock), 0, s, w.bsum, w.nb, w.start + w.L);
  DRBA_LAUNCH(scan_add, dim3((unsigned)((w.L + 255) / 256)), dim3(kBlock), 0, s, w.start, w.bsum, w.L);
  DRBA_LAUNCH(splat_sort_fill, dim3(grid_for(P), N), dim3(kBlock), 0, s, flow, metric, w.start, w.cnt, w.rec, H, W, mode);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

// drba_softsplat_again for a source already in the quad-interleaved layout (C >= 16, C % 4 == 0): same products per channel
int drba_softsplat_gather_quad(const float *in_quad, float *out, float *ws, int N, int C, int H, int W, int mode, int eps, void *stream) {
  if (!in_quad || !out || !ws || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  if (mode < 0 || mode > 3 || eps < 0 || eps > 2) return DRBA_EINVAL;
  if (!splat_quad(C)) return DRBA_EUNSUPPORTED;
  const SplatWs w = splat_ws(ws, N, H, W);
  const int chunks = (C + kChunk - 1) / kChunk;
  DRBA_LAUNCH(splat_sorted_gather_quad, dim3(tiles_for(W, H), N * chunks), dim3(kBlock), 0, (hipStream_t)stream, in_quad, w.start, w.rec, out,
              C, H, W, mode, eps, chunks);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_softsplat_again(const float *in, float *out, float *ws, int N, int C, int H, int W, int mode, int eps, void *stream) {
  if (!in || !out || !ws || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  if (mode < 0 || mode > 3 || eps < 0 || eps > 2) return DRBA_EINVAL;
  return splat_gather(in, out, splat_ws(ws, N, H, W), N, C, H, W, mode, eps, (hipStream_t)stream);
}

int drba_backwarp(const float *in, const float *flow, float *out, int N, int C, int H, int W, int padding,
                  void *stream) {
  if (!in || !flow || !out || N <= 0 || C <= 0 || H <= 1 || W <= 1 || padding < 0 || padding > 1) return DRBA_EINVAL;
  dim3 g(grid_for((size_t)H * W), N);
  if (padding == 0)
    DRBA_LAUNCH(backwarp_kernel<false>, g, dim3(kBlock), 0, (hipStream_t)stream, in, flow, out, C, H, W);
  else
    DRBA_LAUNCH(backwarp_kernel<true>, g, dim3(kBlock), 0, (hipStream_t)stream, in, flow, out, C, H, W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_flow_distance(const float *flow, float *out, int N, int H, int W, void *stream) {
  if (!flow || !out || N <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  dim3 g(grid_for((size_t)H * W), N);
  DRBA_LAUNCH(distance_kernel, g, dim3(kBlock), 0, (hipStream_t)stream, flow, out, H, W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

size_t drba_rife_splat_ws_floats(int N, int H, int W, int values) {
  if (N <= 0 || H <= 0 || W <= 0 || values < 1 || values > 2) return 0;
  // the reach map (fixed size, scratch), accumulators [N][H*W][values + 1], then one flag per (item, 32 x 16 output tile)
  return (size_t)kReachWords + (size_t)N * H * W * (values + 1) + (size_t)N * ((W + kSX - 1) / kSX) * ((H + kSY - 1) / kSY);
}

int drba_flow_reverse(const float *flow, float *out, float *ws, int N, int H, int W, void *stream) {
  if (!flow || !out || !ws || N <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  SplatJobs none;
  none.n = 0;
  dim3 g((W + kSX - 1) / kSX, (H + kSY - 1) / kSY, N);  // source tiles of the pre-pass = output tiles of the splat
  DRBA_LAUNCH(splat_long_prepass<0>, g, dim3(kBlock), 0, s, flow, (const float *)nullptr, 0.f,
                     (const float *)nullptr, 0.f, ws, H, W, none);
  DRBA_LAUNCH(splat_tiled<0>, g, dim3(kBlock), 0, s, flow, (const float *)nullptr, 0.f, (const float *)nullptr,
                     0.f, ws, out, H, W, none);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_drm_rife_linear(const float *flow_self, const float *flow_other, float t, const float *t_dev, float eps,
                         float *out, float *ws, int N, int H, int W, void *stream) {
  if (!flow_self || !flow_other || !out || !ws || N <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  SplatJobs none;
  none.n = 0;
  dim3 g((W + kSX - 1) / kSX, (H + kSY - 1) / kSY, N);
  DRBA_LAUNCH(splat_long_prepass<1>, g, dim3(kBlock), 0, s, flow_self, flow_other, t, t_dev,
                     eps, ws, H, W, none);
  DRBA_LAUNCH(splat_tiled<1>, g, dim3(kBlock), 0, s, flow_self, flow_other, t, t_dev, eps, ws, out, H, W, none);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_drm_rife_linear_batch(const drba_drm_job_t *jobs, int n_jobs, float eps, float *ws, int H, int W, void *stream) {
  if (!jobs || n_jobs <= 0 || n_jobs > DRBA_MAX_STAGE_ITEMS || !ws || H <= 0 || W <= 0) return DRBA_EINVAL;
  SplatJobs J;
  memset(&J, 0, sizeof(J));
  J.n = n_jobs;
  for (int k = 0; k < n_jobs; ++k) {
    if (!jobs[k].flow_self || !jobs[k].flow_other || !jobs[k].out) return DRBA_EINVAL;
    J.fs[k] = jobs[k].flow_self, J.fo[k] = jobs[k].flow_other, J.t[k] = jobs[k].t, J.out[k] = jobs[k].out;
  }
  hipStream_t s = (hipStream_t)stream;
  dim3 g((W + kSX - 1) / kSX, (H + kSY - 1) / kSY, n_jobs);
  DRBA_LAUNCH(splat_long_prepass<1>, g, dim3(kBlock), 0, s, J.fs[0], J.fo[0], 0.f, (const float *)nullptr, eps, ws,
              H, W, J);
  DRBA_LAUNCH(splat_tiled<1>, g, dim3(kBlock), 0, s, J.fs[0], J.fo[0], 0.f, (const float *)nullptr, eps, ws, J.out[0], H, W, J);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_drm_ratio(const float *flow10, const float *flow12, float eps, float *drm10, float *drm12, int N, int H,
                   int W, void *stream) {
  if (!flow10 || !flow12 || (!drm10 && !drm12) || N <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  dim3 g(grid_for((size_t)H * W), N);
  DRBA_LAUNCH(drm_ratio_kernel, g, dim3(kBlock), 0, (hipStream_t)stream, flow10, flow12, eps, drm10, drm12, H, W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_affine(const float *a, float mul, float add, float *out, size_t n, void *stream) {
  if (!a || !out || n == 0) return DRBA_EINVAL;
  DRBA_LAUNCH(affine_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, a, mul, add, out, n);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_mul_map(const float *x, const float *map, float *out, int N, int C, int H, int W, void *stream) {
  if (!x || !map || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DRBA_EINVAL;
  dim3 g(grid_for((size_t)H * W), N);
  DRBA_LAUNCH(mul_map_kernel, g, dim3(kBlock), 0, (hipStream_t)stream, x, map, out, C, (size_t)H * W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_fill_holes(const float *aligned, const float *cover, const float *value, float *out, size_t n, void *stream) {
  if (!aligned || !cover || !value || !out || n == 0) return DRBA_EINVAL;
  DRBA_LAUNCH(fill_holes_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, aligned, cover, value, out, n);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_drm_retime(const float *drm, float *out, double t, double precision, size_t n, void *stream) {
  if (!drm || !out || n == 0 || !(precision > 0)) return DRBA_EINVAL;
  DRBA_LAUNCH(drm_retime_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, drm, out, t, precision, n);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // extern "C"
