// Shared helpers for the gfx950 kernel library.  Written for MI355X only: wave64,
// 256 CUs in 8 XCDs, fp32 hardware atomics to device (coarse-grained) memory.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/drba_hip.h"

#define DRBA_CHECK_LAUNCH()                                   \
  do {                                                        \
    if (hipGetLastError() != hipSuccess) return DRBA_ELAUNCH; \
  } while (0)

namespace drba {

// Kernel trace for bench.py's roofline object (api_misc.hip): while drba_trace_begin() is in effect every launch
// attaches an event pair to its own dispatch packet (hipExtLaunchKernelGGL), so the pair brackets exactly the kernel
// -- the duration rocprofv3's kernel trace reports -- without the barrier packets an event recorded on the stream
// adds before and after the launch.  Every kernel launch of the library goes through DRBA_LAUNCH.
struct TimedLaunch {
  hipEvent_t start, stop;  // both null when the trace is full
};
extern bool g_trace_on;
TimedLaunch trace_launch(const void *host_fn, const char *fallback_name, dim3 grid, hipStream_t stream);

#define DRBA_LAUNCH(kernel, grid, block, lds, stream, ...)                                                     \
  do {                                                                                                         \
    if (drba::g_trace_on) {                                                                                    \
      const drba::TimedLaunch tl_ = drba::trace_launch((const void *)(kernel), #kernel, grid, stream);         \
      if (tl_.start) {                                                                                         \
        hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, tl_.start, tl_.stop, 0, __VA_ARGS__);          \
        break;                                                                                                 \
      }                                                                                                        \
    }                                                                                                          \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                         \
  } while (0)
#define DRBA_LAUNCH_TIMED DRBA_LAUNCH

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device: raised once per (kernel, device), not once
// per process -- a process that drives a second GPU (RIFE(device="cuda:1")) would otherwise launch > 64 KB-LDS kernels there
// without it.  Thread-safe; the largest size applied per (kernel, device) is remembered, failures are not (api_misc.hip).
hipError_t max_dynamic_lds(const void *kernel, int bytes);

// Debug range check of the two-term fp16 family (drba_set_range_check, include/drba_hip.h): when on, an entry point that ran a
// family-4 kernel scans the output it has just enqueued for non-finite values (one extra kernel, one stream synchronisation,
// one 4-byte copy) and returns DRBA_EUNSUPPORTED instead of handing inf / NaN on: the family's operands overflow fp16 at
// |activation| >= 65504 * 16 (weights, attention Q / V: 65504), which the 24-bit families do not.  Off by default.
extern bool g_range_check;
int range_scan(const float *out, size_t n, void *stream);  // DRBA_OK / DRBA_EUNSUPPORTED (non-finite found) / DRBA_ELAUNCH
static inline int range_checked(int rc, const float *out, size_t n, void *stream) {
  return (rc != DRBA_OK || !g_range_check) ? rc : range_scan(out, n, stream);
}

// Always-on overflow report of the same family (drba_status_word, include/drba_hip.h; ABI 8): every family-4 kernel takes the
// device address of the current device's status bytes (null until the caller asked for the word: then nothing is reported),
// folds each value it stores into a NaN accumulator (v * 0 + nf: NaN as soon as one v is inf / NaN, one VALU operation per stored
// value, no branch, no compare-to-SGPR) and writes 1 into its byte at the end of the workgroup if that came out NaN.  The word is
// host-mapped: the host reads it without touching the stream.
unsigned char *status_bytes();  // device address for the current device, or nullptr
__device__ __forceinline__ float nf_fold(float nf, float v) { return __builtin_fmaf(v, 0.f, nf); }
__device__ __forceinline__ void nf_report(unsigned char *status, int which, float nf) {
  if (status && nf != nf) status[which] = 1;
}
// ... and the weights of the family at pack time: the two-term form holds a weight as fp16(w) + 2^-11 fp16(...) with no
// pre-scale, so |w| >= 65504 (or a non-finite one) cannot be packed: the *_pack entry points return DRBA_EUNSUPPORTED and the
// host falls to a 24-bit family for that layer.
static inline bool two_term_weights_ok(const float *w, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (!(w[i] > -65504.f && w[i] < 65504.f)) return false;  // (NaN fails both comparisons)
  return true;
}

// Environment switches select between kernel variants of THIS library for A/B measurements (never another backend).  The
// release build -- the Makefile's default -- compiles them out: env_int() returns the default without reading the
// environment; `make TUNING=1` (-DDRBA_TUNING_SWITCHES) builds the measuring library the tools/ scripts may use.
#ifdef DRBA_TUNING_SWITCHES
static inline int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v ? atoi(v) : dflt;
}
#else
static inline int env_int(const char *, int dflt) { return dflt; }
#endif

constexpr int kBlock = 256;      // 4 waves: one per SIMD of a CU
constexpr int kMaxBlocks = 2048; // 256 CUs x 8: grid-stride beyond this (guide G11)

static inline int grid_for(size_t n, int per_block = kBlock) {
  size_t b = (n + per_block - 1) / per_block;
  if (b > (size_t)kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

// XCD-aware 2-D tiling for gather kernels.  Workgroup b runs on XCD b % 8 (observed dispatch order;
// used for speed only): remap so that each XCD owns a contiguous band of tiles and the source rows a
// bilinear gather shares between neighbouring tiles are fetched into ONE private L2 instead of up to
// eight (PMC: 4x the algorithmic read traffic with the linear block order).  Bijective for any count.
__device__ __forceinline__ int xcd_band(int b, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, xcd = b & 7, k = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
// ... and inside its band an XCD walks the tiles in STRIPS of kStripW columns, top to bottom, instead of whole rows: a
// row of tiles of the 1080p stage-input gather moves 2.6 MB of sources and 3.2 MB of output through the XCD's 4 MB L2, so
// the halo rows a tile shares with the one below it were gone by the time that one ran (PMC, row-major walk: reads
// 1.34x / 1.70x algorithmic on the scale-1 / scale-2 gathers -- exactly the tiles' halo ratio).  With 8 tiles between
// vertical neighbours the shared rows are still resident.  nblocks == tiles_x * tiles_y; bijective for any shape.
constexpr int kStripW = 8;
__device__ __forceinline__ void xcd_strip_tile(int b, int nblocks, int tiles_x, int &tx, int &ty) {
  const int i = xcd_band(b, nblocks);  // position in the global order: XCD b % 8 owns a contiguous range of it
  const int tiles_y = nblocks / tiles_x;
  const int RB = (tiles_y + 7) >> 3;   // tile rows per band (the last band may be shorter)
  const int per_band = RB * tiles_x;
  const int j = i / per_band, k = i - j * per_band;
  const int nrows = min(RB, tiles_y - j * RB);
  const int nfull = tiles_x / kStripW, full = nfull * kStripW * nrows;
  int row, col;
  if (k < full) {
    const int s = k / (kStripW * nrows), r = k - s * kStripW * nrows;
    row = r / kStripW;
    col = s * kStripW + (r - row * kStripW);
  } else {  // the narrower last strip
    const int wl = max(tiles_x - nfull * kStripW, 1), r = k - full;
    row = r / wl;
    col = nfull * kStripW + (r - row * wl);
  }
  tx = col;
  ty = j * RB + row;
}
// A launch over (tiles x items) as ONE grid dimension with the items of a tile back to back on ONE XCD (round 4).  The items of
// a group of DRBA steps read the same six source frames -- frame I(k) is img0 of two items and img1 of up to two more, at
// sample points a few pixels apart --, but launched as blockIdx.y = item the whole frame of item 0 was gathered before
// item 1 started: every item fetched its 159 MB per source from HBM again.  XCD x (= linear workgroup id % 8) walks its band
// of tiles and runs all n items of a tile consecutively, so the later items find the neighbourhood in the XCD's L2.
// vb = the block id xcd_strip_tile() expects (over `ntiles`), item = which item.  Falls back to item-major order when the
// tile count is not a multiple of 8 (the XCD bands would not be equal).
__device__ __forceinline__ void tile_item_block(int n_items, int &vb, int &item, int &ntiles) {
  const int b = blockIdx.x;
  ntiles = gridDim.x / n_items;
  if ((ntiles & 7) == 0 && n_items > 1) {
    const int xcd = b & 7, k = b >> 3;
    item = k % n_items;
    vb = (k / n_items) * 8 + xcd;
  } else {
    item = b / ntiles;
    vb = b - item * ntiles;
  }
}

constexpr int kTileW = 32, kTileH = 8;  // pixels per 256-thread workgroup: a wave covers 32 x 2

struct Tile2D {
  int x, y;
  bool valid;
};
// thread -> pixel of a W x H image tiled 32 x 8; gridDim.x must be tiles_x * tiles_y
__device__ __forceinline__ Tile2D tile_pixel(int W, int H) {
  const int tiles_x = (W + kTileW - 1) / kTileW;
  int tx, ty;
  xcd_strip_tile(blockIdx.x, gridDim.x, tiles_x, tx, ty);
  Tile2D p;
  p.x = tx * kTileW + (threadIdx.x & (kTileW - 1));
  p.y = ty * kTileH + (threadIdx.x >> 5);
  p.valid = p.x < W && p.y < H;
  return p;
}
static inline int tiles_for(int W, int H) { return ((W + kTileW - 1) / kTileW) * ((H + kTileH - 1) / kTileH); }

// fp32 atomic add that lowers to global_atomic_add_f32 (no CAS loop).  All buffers we
// scatter into are torch device allocations (coarse-grained), where the hardware op is valid.
__device__ __forceinline__ void atomic_add_f32(float *p, float v) { unsafeAtomicAdd(p, v); }

__device__ __forceinline__ float lrelu02(float v) { return v > 0.f ? v : 0.2f * v; }

// Lane exchanges inside a quad / a row of 16 as DPP operand modifiers of a VALU move (no LDS crossbar trip, no lgkmcnt wait):
// __shfl_xor(v, 1 | 2 | 8) compiles to ds_bpermute_b32 -- the scale >= 2 stage-input gather issued 104 of them per lane for
// the 2 x 2 downsample of its 52 channels, every one an LDS-pipe instruction with ~100 clocks of latency in front of a
// dependent add.  All lanes of the quad / row must be active (a disabled source lane reads as 0).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_xor1(float v) { return dpp_f32<0xB1>(v); }  // quad_perm [1,0,3,2]
__device__ __forceinline__ float quad_xor2(float v) { return dpp_f32<0x4E>(v); }  // quad_perm [2,3,0,1]

// torch.linspace(-1, 1, n)[i] in fp32: symmetric two-sided formula (ATen RangeFactories).
__device__ __forceinline__ float linspace_m1p1(int i, int n) {
  const float step = 2.0f / (float)(n - 1);
  return (i < n / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

// Pixel-space sampling coordinate of warp(): base grid + flow/((size-1)/2), then
// grid_sample's align_corners=True un-normalisation (g+1)*((size-1)/2) (the form ATen's
// vectorised CPU kernel uses).  Reproduces the reference's fp32 round trip
// (warplayer.py:17-22) instead of the algebraic x+flow.
__device__ __forceinline__ float warp_coord(int i, int n, float flow) {
  const float half = ((float)n - 1.0f) / 2.0f;
  const float g = linspace_m1p1(i, n) + flow / half;
  return (g + 1.0f) * half;
}

struct Taps {  // bilinear taps of grid_sample: clamped indices + weights
  int x0, x1, y0, y1;
  float wnw, wne, wsw, wse;
};

// padding_mode='border': clip the coordinate to [0, size-1] first, then bilinear.
__device__ __forceinline__ Taps taps_border(float x, float y, int W, int H) {
  x = fminf(fmaxf(x, 0.f), (float)(W - 1));
  y = fminf(fmaxf(y, 0.f), (float)(H - 1));
  const float fx = floorf(x), fy = floorf(y);
  Taps t;
  t.x0 = (int)fx;
  t.y0 = (int)fy;
  const float wx1 = x - fx, wy1 = y - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  t.wnw = wx0 * wy0;
  t.wne = wx1 * wy0;
  t.wsw = wx0 * wy1;
  t.wse = wx1 * wy1;
  t.x1 = min(t.x0 + 1, W - 1);  // the clamped-away tap always carries weight 0
  t.y1 = min(t.y0 + 1, H - 1);
  // NaN coordinates: fminf/fmaxf drop the NaN -> sample at (W-1 or 0); matches clip semantics
  t.x0 = min(max(t.x0, 0), W - 1);
  t.y0 = min(max(t.y0, 0), H - 1);
  return t;
}

// The two taps of a row are adjacent (x1 == x0 + 1, or x1 == x0 == W-1 at the right border), so each row is one
// 8-byte load instead of two 4-byte ones: the gather kernels are bound by the texture-address unit (TA busy 86-98 %,
// ~20 cycles per wave-level gather instruction measured), i.e. by the NUMBER of vector-memory instructions.
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ float sample(const float *__restrict__ p, int W, const Taps &t) {
  const int xb = min(t.x0, W - 2);  // W >= 2; xb != x0 only when x0 == x1 == W-1
  const f32x2u a = *reinterpret_cast<const f32x2u *>(p + (size_t)t.y0 * W + xb);
  const f32x2u b = *reinterpret_cast<const f32x2u *>(p + (size_t)t.y1 * W + xb);
  const bool edge = t.x0 != xb;
  const float a0 = edge ? a.y : a.x, b0 = edge ? b.y : b.x;
  return a0 * t.wnw + a.y * t.wne + b0 * t.wsw + b.y * t.wse;
}

// Same for two channels stored pair-interleaved ([C/2][H][W][2]): the 2 x 2 taps of BOTH channels of a pair are two
// 16-byte loads, i.e. half the gather instructions per channel again.  pp points at the pair's plane.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(8)));
__device__ __forceinline__ void sample_pair(const float *__restrict__ pp, int W, const Taps &t, float &v0, float &v1) {
  const int xb = min(t.x0, W - 2);
  const f32x4u a = *reinterpret_cast<const f32x4u *>(pp + ((size_t)t.y0 * W + xb) * 2);
  const f32x4u b = *reinterpret_cast<const f32x4u *>(pp + ((size_t)t.y1 * W + xb) * 2);
  const bool edge = t.x0 != xb;
  const float a00 = edge ? a.z : a.x, b00 = edge ? b.z : b.x;
  const float a01 = edge ? a.w : a.y, b01 = edge ? b.w : b.y;
  v0 = a00 * t.wnw + a.z * t.wne + b00 * t.wsw + b.z * t.wse;
  v1 = a01 * t.wnw + a.w * t.wne + b01 * t.wsw + b.w * t.wse;
}

// The three channels of a frame kept as [H][W][4] (c0, c1, c2, 0) at the four taps: two 16-byte loads per tap row.  Same
// products in the same order as sample() per channel (the right-border case: x1 == x0 carries weight 0).
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sample_x4(const float *__restrict__ px, int W, const Taps &t, float (&v)[3]) {
  const int xb = min(t.x0, W - 2);
  const bool edge = t.x0 != xb;
  const f32x4v a0 = *reinterpret_cast<const f32x4v *>(px + ((size_t)t.y0 * W + xb) * 4), a1 = *reinterpret_cast<const f32x4v *>(px + ((size_t)t.y0 * W + xb + 1) * 4);
  const f32x4v b0 = *reinterpret_cast<const f32x4v *>(px + ((size_t)t.y1 * W + xb) * 4), b1 = *reinterpret_cast<const f32x4v *>(px + ((size_t)t.y1 * W + xb + 1) * 4);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float al = edge ? a1[c] : a0[c], bl = edge ? b1[c] : b0[c];
    v[c] = al * t.wnw + a1[c] * t.wne + bl * t.wsw + b1[c] * t.wse;
  }
}

// Source taps of F.interpolate(bilinear, align_corners=False) along one axis.
struct Lerp {
  int i0, i1;
  float w0, w1;
};
__device__ __forceinline__ Lerp lerp_src(int dst, float scale, int size) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  Lerp l;
  l.i0 = min((int)s, size - 1);
  l.i1 = l.i0 + (l.i0 < size - 1 ? 1 : 0);
  l.w1 = s - (float)l.i0;
  l.w0 = 1.f - l.w1;
  return l;
}

// One bilinear upsample value from its four taps, each 1-D interpolation as fma(w0, a, w1 * b) -- the form ATen's
// upsample kernels evaluate (established bit for bit on to_inp, ifnet_glue.hip).  Spelled out so that every kernel that
// upsamples the previous head output (ifblock_input_lds, stage_conv0) produces the same bits.
__device__ __forceinline__ float lerp2_fma(float wy0, float wy1, float wx0, float wx1, float p00, float p01, float p10, float p11) {
  const float top = __fmaf_rn(wx0, p00, __fmul_rn(wx1, p01));
  const float bot = __fmaf_rn(wx0, p10, __fmul_rn(wx1, p11));
  return __fmaf_rn(wy0, top, __fmul_rn(wy1, bot));
}

}  // namespace drba
