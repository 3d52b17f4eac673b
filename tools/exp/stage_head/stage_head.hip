// EXPERIMENT (not part of libdrba_hip.so): measured and NOT adopted -- see DESIGN.md "Tried and rejected".
//   tools/exp/stage_head/run.py builds it into a private copy of the library, checks it against the product's
//   drba_ifblock_input_lds + drba_conv3x3 pair and times both.  MI355X, 1088x1920, per sample: fused 372 us (445 us with the
//   folded flow update) against 231 + 117 us for the pair, whether the taps come from the LDS-staged windows or from
//   memory (slow path): ~5000 VALU instructions per wave and tile for the tap arithmetic, the bf16 splits and the window
//   index arithmetic bound it, not memory; one workgroup per CU instead of two costs 8 %.
//
// Warped stage input FUSED with the IFBlock's first convolution (IFNet_HDv3.py:85-88,146-156 + conv0[0] :65-68):
//     y = LeakyReLU_0.2(conv3x3_s2(cat(warp(img0), warp(img1), warp(f0), warp(f1), timestep, mask, feat, flow/s)) + b)
// ifblock_input_lds writes that 52-channel concat at stage resolution (434 MB per sample at 1088x1920, scale 1) and the
// stride-2 convolution reads it straight back: 868 MB of the 1.33 GB the pair moves per sample.  Here a workgroup builds
// the concat for the haloed window of ONE 4 x TWO tile of conv outputs in LDS -- 16 channels at a time -- and contracts it
// on the matrix cores; the only HBM traffic left is the gather's reads, the folded flow update and the 16/32-channel
// half-resolution output.
//
// Arithmetic of the 52 channels: exactly ifblock_input_lds's (same helpers, same evaluation order).  The convolution is
// evaluated like conv_split.hip's: fp32 operands as three bf16 terms, six partial products per multiply on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation (differs from the fp32-MFMA kernel by a few ulp of the sum).
//
// GEMM view per workgroup (4 waves, wave = output row of the tile): M = TWO output pixels per wave (MT tiles of 16),
// N = NT tiles of 16 output channels, K = 16 channels x 2 taps per MFMA (lane group kq: channels 8*(kq & 1)..+7 of tap
// 2*pair + (kq >> 1); the tenth tap of the five pairs has zero weights), four channel chunks:
//     chunk 0: img0 (3) img1 (3) timestep mask | feat (8)      chunk 1: warp(f0) 16      chunk 2: warp(f1) 16
//     chunk 3: flow / s (4) + 12 zero channels
// LDS tile: [plane h/m/l][channel group][slot][8 x bf16]; slot = row * RS + (col & 1) * HALF + (col >> 1): the window's
// even and odd columns are stored apart, so the 16 pixels of an MFMA row block (stride 2 in the window) are 16 consecutive
// 16-byte slots for every tap.
#include "../../../drba_amd/csrc/common.hpp"

#include <stdlib.h>
#include <string.h>

using namespace drba;

namespace drba_sh {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kChunks = 4, kPairs = 5;
// input channel of (chunk, k) in the concat's order (IFNet_HDv3.py:151-156: img0 0-2, img1 3-5, f0 6-21, f1 22-37,
// timestep 38, mask 39, feat 40-47, flow 48-51); -1: zero padding of K
static const int kChunkMap[kChunks][16] = {
    {0, 1, 2, 3, 4, 5, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47},
    {6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21},
    {22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37},
    {48, 49, 50, 51, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}};

// Geometry (scale 1: every window pixel is one full-resolution sample point, one lane each)
struct Geo {
  static constexpr int TWO = 16, THO = 4;                   // conv outputs per workgroup
  static constexpr int WR = 2 * THO + 1, WC = 2 * TWO + 1;  // haloed window of the convolution's input: 9 x 33
  static constexpr int NPIX = WR * WC;                      // 297 lanes of the 320 carry a window pixel
  static constexpr int THREADS = 320;                       // 5 waves: all gather, waves 0..3 own one output row each
  static constexpr int HALF = TWO + 1, RS = 2 * HALF;       // even columns: TWO + 1 slots, odd columns: TWO (+1 unused)
  static constexpr int NSLOT = WR * RS;
  static constexpr int NPIXP = ((NSLOT - 4 + 15) / 16) * 16 + 4;  // 16 * NPIXP == 64 (mod 256): groups on distinct banks
  static constexpr int PRH = 7, PRW = 20;                   // capacity of tmp_prev's footprint (7 x 19 needed)
  // Source window of a warp, staged in LDS: the convolution window displaced by the flow at the tile's centre, a margin
  // of R pixels for the flow's variation inside the tile (+1 for the second bilinear tap), columns from a multiple of 4
  static constexpr int R = 3;
  static constexpr int SH = WR + 2 * R + 1, SW = 44;        // 16 rows x 44 columns (33 + 2R + 1 + 3 of alignment slack, to x4)
  static constexpr int PAIR_UNITS = 4 * SH * (SW / 2);      // 16-byte units: 4 pair planes (8 channels), 2 pixels per unit
  static constexpr int IMG_UNITS = 6 * SH * (SW / 4);       // img0 (3 planes, window 0) + img1 (3 planes, window 1)
  static constexpr int TILE_BYTES = 3 * 2 * NPIXP * 16;
  static constexpr int PREV_BYTES = 13 * PRH * PRW * 4;
  static constexpr int WBUF_UNITS = kPairs * 3 * 64;        // one chunk's weights (Cout = 16)
  static constexpr int SBUF_BYTES = PAIR_UNITS * 16;
  static constexpr int LDS_BYTES = TILE_BYTES + PREV_BYTES + WBUF_UNITS * 16 + SBUF_BYTES + 32;
};
static_assert(Geo::IMG_UNITS <= Geo::PAIR_UNITS, "the image windows share the feature windows' buffer");
static_assert(2 * Geo::LDS_BYTES <= 160 * 1024, "two workgroups per CU");

// fp32 -> (h, m, l) bf16, round-to-nearest-even at every step (conv_split.hip); two values packed per word
__device__ __forceinline__ void split2(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto pk = [](float x, float y) -> unsigned {
    const bf16x2 p = __builtin_convertvector(f32x2{x, y}, bf16x2);
    return __builtin_bit_cast(unsigned, p);
  };
  h = pk(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = pk(ra, rb);
  l = pk(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}

// sample() / sample_pair() of common.hpp in two halves: the four taps are fetched (from the staged window, or from
// memory where a lane's taps leave it), the arithmetic -- identical, operation for operation -- runs on the fetched values
struct Raw2 {
  f32x2u a, b;
};
struct Raw4 {
  f32x4u a, b;
};
__device__ __forceinline__ Raw2 ld_taps(const float *__restrict__ p, int W, const Taps &t) {
  const int xb = min(t.x0, W - 2);
  Raw2 r;
  r.a = *reinterpret_cast<const f32x2u *>(p + (size_t)t.y0 * W + xb);
  r.b = *reinterpret_cast<const f32x2u *>(p + (size_t)t.y1 * W + xb);
  return r;
}
__device__ __forceinline__ float fin_taps(const Raw2 &r, int W, const Taps &t) {
  const bool edge = t.x0 != min(t.x0, W - 2);
  const float a0 = edge ? r.a.y : r.a.x, b0 = edge ? r.b.y : r.b.x;
  return a0 * t.wnw + r.a.y * t.wne + b0 * t.wsw + r.b.y * t.wse;
}
__device__ __forceinline__ Raw4 ld_pair(const float *__restrict__ pp, int W, const Taps &t) {
  const int xb = min(t.x0, W - 2);
  Raw4 r;
  r.a = *reinterpret_cast<const f32x4u *>(pp + ((size_t)t.y0 * W + xb) * 2);
  r.b = *reinterpret_cast<const f32x4u *>(pp + ((size_t)t.y1 * W + xb) * 2);
  return r;
}
__device__ __forceinline__ void fin_pair(const Raw4 &r, int W, const Taps &t, float &v0, float &v1) {
  const bool edge = t.x0 != min(t.x0, W - 2);
  const float a00 = edge ? r.a.z : r.a.x, b00 = edge ? r.b.z : r.b.x;
  const float a01 = edge ? r.a.w : r.a.y, b01 = edge ? r.b.w : r.b.y;
  v0 = a00 * t.wnw + r.a.z * t.wne + b00 * t.wsw + r.b.z * t.wse;
  v1 = a01 * t.wnw + r.a.w * t.wne + b01 * t.wsw + r.b.w * t.wse;
}

// Workgroup barrier that orders LDS traffic only (conv_split.hip): __syncthreads() would also drain the wave's global-memory
// queue (s_waitcnt vmcnt(0)), i.e. wait for the NEXT phase's window loads that are meant to fly under this phase's work.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A lane's view of one staged source window
struct Win {
  int bx, by;      // image coordinates of the window's cell (0, 0); bx a multiple of 4
  int rx, ry0, ry1;  // the lane's tap origin (column of the left taps, rows of the upper / lower taps) inside it
  bool in;         // all four taps are inside the window
};

// Experiment builds only (-DDRBA_SH_CLOCKS): clocks between consecutive marks, summed over the workgroups' first lanes
#ifdef DRBA_SH_CLOCKS
__device__ unsigned long long g_sh_clk[32];
#define SH_MARK(i)                                                                    \
  do {                                                                                \
    const long long now_ = (long long)__builtin_readcyclecounter();                   \
    if (tid == 0) atomicAdd(&g_sh_clk[i], (unsigned long long)(now_ - clk_last_));    \
    clk_last_ = now_;                                                                 \
  } while (0)
#else
#define SH_MARK(i)
#endif

template <bool FOLD>
__global__ void __launch_bounds__(Geo::THREADS, 2)
stage_head_kernel(const float *__restrict__ img0, const float *__restrict__ img1, const float *__restrict__ f0p,
                  const float *__restrict__ f1p, const float *__restrict__ tmap, float tscalar,
                  const float *__restrict__ flow, const float *__restrict__ tmp_prev, int hp, int wp, float inv_prev_scale,
                  float prev_scale, float *__restrict__ flow_out, const u32x4 *__restrict__ wfrag,
                  const float *__restrict__ bias, float *__restrict__ out, int H, int W, int ho, int wo, int tiles_x) {
#if defined(__HIP_DEVICE_COMPILE__)
  using G = Geo;
  constexpr int TWO = G::TWO, THO = G::THO, WR = G::WR, WC = G::WC, NPIX = G::NPIX, HALF = G::HALF, RS = G::RS;
  constexpr int NPIXP = G::NPIXP, PRH = G::PRH, PRW = G::PRW, NTHR = G::THREADS, WU = G::WBUF_UNITS, R = G::R;
  constexpr int SH = G::SH, SW = G::SW;
  constexpr int WLD = (WU + NTHR - 1) / NTHR;                  // 16-byte weight units per thread and chunk
  constexpr int PLD = (G::PAIR_UNITS + NTHR - 1) / NTHR;       // window units per thread: 8 feature channels
  constexpr int ILD = (G::IMG_UNITS + NTHR - 1) / NTHR;        // ... the two images
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4 *tile = reinterpret_cast<u32x4 *>(smem);  // [plane 3][group 2][NPIXP]
  float(*prev)[PRH][PRW] = reinterpret_cast<float(*)[PRH][PRW]>(smem + G::TILE_BYTES);
  u32x4 *wbuf = reinterpret_cast<u32x4 *>(smem + G::TILE_BYTES + G::PREV_BYTES);  // [pair][plane][lane]
  float *fo = reinterpret_cast<float *>(wbuf);    // [4][8][32] folded flow of the owned block (before wbuf's first use)
  u32x4 *sbuf = wbuf + WU;                        // staged source window
  int *offs = reinterpret_cast<int *>(sbuf + G::PAIR_UNITS);
  constexpr int C0 = FOLD ? 0 : 4;  // first channel of tmp_prev that is needed

  const int tid = threadIdx.x, lane = tid & 63;
#ifdef DRBA_SH_CLOCKS
  long long clk_last_ = (long long)__builtin_readcyclecounter();
#endif
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const size_t P = (size_t)H * W, p_prev = (size_t)hp * wp, p_out = (size_t)ho * wo;
  const int t = xcd_band(blockIdx.x, gridDim.x);
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int ox0 = tx * TWO, oy0 = ty * THO;
  const int wx0 = 2 * ox0 - 1, wy0 = 2 * oy0 - 1;  // full-resolution coordinates of window pixel (0, 0)

  // ---- this lane's window pixel
  const bool act = tid < NPIX;
  const int pix = min(tid, NPIX - 1);
  const int wr = pix / WC, wc = pix - wr * WC;
  const int y = wy0 + wr, x = wx0 + wc;
  const bool on = act && y >= 0 && y < H && x >= 0 && x < W;  // outside the image: the convolution's zero padding
  const bool owned = wr >= 1 && wc >= 1;  // row / column 0 of the window belong to the tiles above / to the left
  const int slot = wr * RS + (wc & 1) * HALF + (wc >> 1);
  const int Y = min(max(y, 0), H - 1), X = min(max(x, 0), W - 1);
  const size_t q = (size_t)Y * W + X;

  // ---- requests that depend on nothing: the running flow, the timestep, chunk 0's weights, tmp_prev's footprint
  // (every global load of the fast path is unconditional -- clamped coordinates instead of predicates: a load under a
  // divergent branch makes the compiler drain the whole memory queue, s_waitcnt vmcnt(0), at the next use after the merge,
  // and the requests that are meant to fly under the MFMAs would be waited for right away)
  float fl_in[4] = {0.f, 0.f, 0.f, 0.f};
  if (flow) {
#pragma unroll
    for (int c = 0; c < 4; ++c) fl_in[c] = flow[(size_t)c * P + q];
  }
  const float tmv = tmap ? tmap[q] : tscalar;
  static_assert(WLD * NTHR == WU, "every thread moves the same number of weight units");
  u32x4 wreg[WLD];
  auto w_issue = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < WLD; ++i) wreg[i] = wfrag[(size_t)chunk * WU + tid + NTHR * i];
  };
  auto w_store = [&]() {
#pragma unroll
    for (int i = 0; i < WLD; ++i) wbuf[tid + NTHR * i] = wreg[i];
  };
  w_issue(0);
  const int xa = max(wx0, 0), ya = max(wy0, 0), xb = min(wx0 + WC - 1, W - 1), yb = min(wy0 + WR - 1, H - 1);
  const int rx0 = lerp_src(xa, inv_prev_scale, wp).i0, ry0 = lerp_src(ya, inv_prev_scale, hp).i0;
  const int rw = min(lerp_src(xb, inv_prev_scale, wp).i1 - rx0 + 1, PRW), rh = min(lerp_src(yb, inv_prev_scale, hp).i1 - ry0 + 1, PRH);
  {
    // all of the footprint's loads in flight at once (a rolled loop is one dependent memory round trip per iteration)
    constexpr int NLD = ((13 - C0) * PRH * PRW + NTHR - 1) / NTHR;
    const int n_el = (13 - C0) * rh * rw;
    float pv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = min(tid + NTHR * k, n_el - 1);
      const int c = i / (rh * rw), r = (i - c * rh * rw) / rw, col = i - c * rh * rw - r * rw;
      pv[k] = tmp_prev[(size_t)(C0 + c) * p_prev + (size_t)(ry0 + r) * wp + rx0 + col];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + NTHR * k;
      if (i < n_el) {
        const int c = i / (rh * rw), r = (i - c * rh * rw) / rw, col = i - c * rh * rw - r * rw;
        prev[C0 + c][r][col] = pv[k];
      }
    }
  }
  SH_MARK(1);
  lds_barrier();  // #1
  SH_MARK(2);  // prev staged

  // taps of the previous head output's x prev_scale upsample at (X, Y), relative to the staged footprint
  const Lerp pa = lerp_src(Y, inv_prev_scale, hp), pb = lerp_src(X, inv_prev_scale, wp);
  const int r0 = min(pa.i0 - ry0, PRH - 1), r1 = min(pa.i1 - ry0, PRH - 1);
  const int c0 = min(pb.i0 - rx0, PRW - 1), c1 = min(pb.i1 - rx0, PRW - 1);
  auto prev_up = [&](int c) -> float {
    const float top = pb.w0 * prev[c][r0][c0] + pb.w1 * prev[c][r0][c1];
    const float bot = pb.w0 * prev[c][r1][c0] + pb.w1 * prev[c][r1][c1];
    return pa.w0 * top + pa.w1 * bot;
  };
  float fls[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (FOLD) {
      const float fd = prev_up(c) * prev_scale;  // ifblock_update: flow_in + up(tmp) * scale
      fls[c] = flow ? fl_in[c] + fd : fd;
      if (act && owned) fo[(c * 8 + wr - 1) * 32 + wc - 1] = fls[c];  // parked: written back as whole 128-byte rows
    } else {
      fls[c] = fl_in[c];
    }
  }
  if (tid == (WR / 2) * WC + WC / 2) {  // the window's centre pixel names the displacement of both source windows
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float f = floorf(fls[c]);
      offs[c] = (on && f >= -30000.f && f <= 30000.f) ? (int)f : 0;  // (NaN / inf: no displacement)
    }
    offs[4] = 0;  // raised by any lane whose taps leave the staged windows
  }
  const Taps t0 = taps_border(warp_coord(X, W, fls[0]), warp_coord(Y, H, fls[1]), W, H);
  const Taps t1 = taps_border(warp_coord(X, W, fls[2]), warp_coord(Y, H, fls[3]), W, H);
  SH_MARK(3);
  lds_barrier();  // #2
  SH_MARK(4);  // offsets (and the parked flow) visible

  auto window = [&](int k, const Taps &tt) -> Win {
    Win w;
    w.bx = (wx0 + offs[2 * k] - R) & ~3;
    w.by = wy0 + offs[2 * k + 1] - R;
    w.rx = min(tt.x0, W - 2) - w.bx;
    w.ry0 = tt.y0 - w.by, w.ry1 = tt.y1 - w.by;
    w.in = w.rx >= 0 && w.rx <= SW - 2 && w.ry0 >= 0 && w.ry1 <= SH - 1;
    return w;
  };
  const Win w0 = window(0, t0), w1 = window(1, t1);
  if (on && !(w0.in && w1.in)) offs[4] = 1;  // (read after barrier #3)

  // ---- window loaders: unit u of a window buffer is the 16 bytes at float4 index u (rows of SW/2 or SW/4 units); rows
  // and columns that leave the image are clamped (no tap ever points at them)
  u32x4 sreg[PLD];
  auto pair_issue = [&](const float *__restrict__ fp, int half, const Win &w) {
#pragma unroll
    for (int i = 0; i < PLD; ++i) {
      const int u = min(tid + NTHR * i, G::PAIR_UNITS - 1);
      const int pl = u / (SH * (SW / 2)), rem = u - pl * (SH * (SW / 2)), r = rem / (SW / 2), j = rem - r * (SW / 2);
      const int yy = min(max(w.by + r, 0), H - 1), xx = min(max(w.bx + 2 * j, 0), W - 2);
      sreg[i] = *reinterpret_cast<const u32x4 *>(fp + (size_t)(4 * half + pl) * 2 * P + ((size_t)yy * W + xx) * 2);
    }
  };
  auto img_issue = [&]() {
#pragma unroll
    for (int i = 0; i < ILD; ++i) {
      const int u = min(tid + NTHR * i, G::IMG_UNITS - 1);
      const int pl = u / (SH * (SW / 4)), rem = u - pl * (SH * (SW / 4)), r = rem / (SW / 4), j = rem - r * (SW / 4);
      const Win &w = pl < 3 ? w0 : w1;
      const float *src = pl < 3 ? img0 + (size_t)pl * P : img1 + (size_t)(pl - 3) * P;
      const int yy = min(max(w.by + r, 0), H - 1), xx = min(max(w.bx + 4 * j, 0), W - 4);
      sreg[i] = *reinterpret_cast<const u32x4 *>(src + (size_t)yy * W + xx);
    }
  };
  auto s_store = [&](int units, int n) {
#pragma unroll
    for (int i = 0; i < PLD; ++i) {
      const int u = tid + NTHR * i;
      if (i < n && u < units) sbuf[u] = sreg[i];
    }
  };
  // 8 channel values of the window pixel -> three bf16 planes of channel group g
  auto put8 = [&](int g, const float *v) {
    u32x4 hh, mm, ll;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned a, b, c;
      split2(v[2 * i], v[2 * i + 1], a, b, c);
      hh[i] = a, mm[i] = b, ll[i] = c;
    }
    tile[(0 * 2 + g) * NPIXP + slot] = hh;
    tile[(1 * 2 + g) * NPIXP + slot] = mm;
    tile[(2 * 2 + g) * NPIXP + slot] = ll;
  };
  // 8 warped feature channels (pair planes 4*half .. +3 of fp) of this lane's pixel from the staged window -> group `half`
  auto pair_taps = [&](int half, const Win &w, const Taps &tt) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
    if (on) {
      typedef float f32x2a __attribute__((ext_vector_type(2)));  // 8-byte aligned: one ds_read_b64 per pixel of a pair plane
      const f32x2a *sp = reinterpret_cast<const f32x2a *>(sbuf);
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) {
        Raw4 r;
        const f32x2a a0 = sp[(pl * SH + w.ry0) * SW + w.rx], a1 = sp[(pl * SH + w.ry0) * SW + w.rx + 1];
        const f32x2a b0 = sp[(pl * SH + w.ry1) * SW + w.rx], b1 = sp[(pl * SH + w.ry1) * SW + w.rx + 1];
        r.a = (f32x4u){a0.x, a0.y, a1.x, a1.y};
        r.b = (f32x4u){b0.x, b0.y, b1.x, b1.y};
        fin_pair(r, W, tt, v[2 * pl], v[2 * pl + 1]);
      }
    }
    if (act) put8(half, v);
  };

  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mma_chunk = [&]() {
    if (wave >= THO) return;  // the fifth wave only gathers
    const int g = kq & 1, tsel = kq >> 1;
#pragma unroll
    for (int p = 0; p < kPairs; ++p) {
      int tap = 2 * p + tsel;
      tap = tap > 8 ? 8 : tap;  // the tenth tap: zero weights, any written slot will do
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int base = (2 * wave + ky) * RS + (kx & 1) * HALF + (kx >> 1) + m;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, tile[(0 * 2 + g) * NPIXP + base]);
      const bf16x8 am = __builtin_bit_cast(bf16x8, tile[(1 * 2 + g) * NPIXP + base]);
      const bf16x8 al = __builtin_bit_cast(bf16x8, tile[(2 * 2 + g) * NPIXP + base]);
      const bf16x8 bh = __builtin_bit_cast(bf16x8, wbuf[(p * 3 + 0) * 64 + lane]);
      const bf16x8 bm = __builtin_bit_cast(bf16x8, wbuf[(p * 3 + 1) * 64 + lane]);
      const bf16x8 bl = __builtin_bit_cast(bf16x8, wbuf[(p * 3 + 2) * 64 + lane]);
      f32x4 c = acc;
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);  // smallest terms first
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
      acc = c;
    }
  };

  // ---- the folded flow of the owned 8 x 32 block goes out as whole rows; the image windows come in
  img_issue();
  if (FOLD) {
    // (straight-line too: a lane with nothing to write gets an offset beyond num_records and its store is dropped)
    const int c = tid >> 6, rr = (tid >> 3) & 7, j = tid & 7;
    const int yy = 2 * oy0 + rr, xx = 2 * ox0 + 4 * j;
    const bool wr_ok = tid < 256 && yy < H && xx < W;  // W % 4 == 0: a float4 is inside or outside as a whole
    const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc((void *)flow_out, 0, (unsigned)(P * 16), 0x00020000);
    const unsigned off = wr_ok ? (unsigned)(((size_t)c * P + (size_t)yy * W + xx) * 4) : 0xffffffffu;
    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(fo + (c * 8 + rr) * 32 + 4 * j), frsrc, off, 0, 0);
  }
  s_store(G::IMG_UNITS, ILD);
  pair_issue(f0p, 0, w0);
  SH_MARK(5);
  lds_barrier();  // #3
  SH_MARK(6);  // image windows staged (the parked flow has been read: wbuf may be written)

  auto epilogue = [&]() {  // + bias, LeakyReLU(0.2); lane holds cout lane % 16 of 4 consecutive output columns
    if (wave >= THO) return;
    const int oy = oy0 + wave, ox = ox0 + kq * 4;
    if (oy >= ho || ox >= wo) return;
    const float b = bias ? bias[m] : 0.f;
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = lrelu02(acc[k] + b);
    float *dst = out + (size_t)m * p_out + (size_t)oy * wo + ox;
    if (ox + 3 < wo && (wo & 3) == 0) {
      *reinterpret_cast<f32x4 *>(dst) = r;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ox + k < wo) dst[k] = r[k];
    }
  };
  auto flow_chunk = [&]() {  // chunk 3: interpolate(flow) * 1 / scale at scale 1 (IFNet_HDv3.py:87) + zero channels
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
    if (on) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = (fls[c] * 1.f) / 1.f;
    }
    if (act) {
      put8(0, v);
      put8(1, v + 8);
    }
  };

  // ---- SLOW PATH (workgroup-uniform; never rejoins the fast path): some lane's taps leave the staged windows (flow that
  // varies by more than R pixels inside the tile, or points far outside the image) -> every lane gathers its taps from
  // memory, phase by phase, like ifblock_input_lds does.  Same values, same arithmetic.
  if (offs[4]) {
    {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = 0.f;
      if (on) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          v[c] = fin_taps(ld_taps(img0 + (size_t)c * P, W, t0), W, t0);
          v[3 + c] = fin_taps(ld_taps(img1 + (size_t)c * P, W, t1), W, t1);
        }
        v[6] = tmv;
#pragma unroll
        for (int c = 0; c < 9; ++c) v[7 + c] = prev_up(4 + c);
      }
      if (act) {
        put8(0, v);
        put8(1, v + 8);
      }
    }
    w_store();
    __syncthreads();
    mma_chunk();
    __syncthreads();
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const float *fp = side ? f1p : f0p;
      const Taps &tt = side ? t1 : t0;
      w_issue(1 + side);
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = 0.f;
      if (on) {
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) fin_pair(ld_pair(fp + (size_t)c2 * 2 * P, W, tt), W, tt, v[2 * c2], v[2 * c2 + 1]);
      }
      if (act) {
        put8(0, v);
        put8(1, v + 8);
      }
      w_store();
      __syncthreads();
      mma_chunk();
      __syncthreads();
    }
    w_issue(3);
    flow_chunk();
    w_store();
    __syncthreads();
    mma_chunk();
    epilogue();
    return;
  }

  // ---- chunk 0: img0, img1, timestep, mask | feat
  {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
    if (on) {
      const float *sp = reinterpret_cast<const float *>(sbuf);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const Win &w = c < 3 ? w0 : w1;
        const Taps &tt = c < 3 ? t0 : t1;
        const float *row0 = sp + (c * SH + w.ry0) * SW + w.rx, *row1 = sp + (c * SH + w.ry1) * SW + w.rx;
        Raw2 r;
        r.a = (f32x2u){row0[0], row0[1]};
        r.b = (f32x2u){row1[0], row1[1]};
        v[c] = fin_taps(r, W, tt);
      }
      v[6] = tmv;
#pragma unroll
      for (int c = 0; c < 9; ++c) v[7 + c] = prev_up(4 + c);  // mask (tmp[4]) and feat (tmp[5:13])
    }
    if (act) {
      put8(0, v);
      put8(1, v + 8);
    }
  }
  w_store();
  w_issue(1);
  SH_MARK(7);
  lds_barrier();  // #4
  SH_MARK(8);  // chunk 0's tile and weights ready; the image windows have been read
  s_store(G::PAIR_UNITS, PLD);  // f0, channels 0..7
  pair_issue(f0p, 1, w0);
  mma_chunk();
  SH_MARK(9);
  lds_barrier();  // #5
  SH_MARK(10);
  // ---- chunk 1: warp(f0)
  pair_taps(0, w0, t0);
  SH_MARK(11);
  lds_barrier();  // #6
  SH_MARK(12);  // window read by everyone
  s_store(G::PAIR_UNITS, PLD);  // f0, channels 8..15
  pair_issue(f1p, 0, w1);
  SH_MARK(13);
  lds_barrier();  // #7
  SH_MARK(14);
  pair_taps(1, w0, t0);
  w_store();
  w_issue(2);
  SH_MARK(15);
  lds_barrier();  // #8
  SH_MARK(16);
  s_store(G::PAIR_UNITS, PLD);  // f1, channels 0..7
  pair_issue(f1p, 1, w1);
  mma_chunk();
  SH_MARK(17);
  lds_barrier();  // #9
  SH_MARK(18);
  // ---- chunk 2: warp(f1)
  pair_taps(0, w1, t1);
  SH_MARK(19);
  lds_barrier();  // #10
  SH_MARK(20);
  s_store(G::PAIR_UNITS, PLD);  // f1, channels 8..15
  SH_MARK(21);
  lds_barrier();  // #11
  SH_MARK(22);
  pair_taps(1, w1, t1);
  w_store();
  w_issue(3);
  SH_MARK(23);
  lds_barrier();  // #12
  SH_MARK(24);
  mma_chunk();
  SH_MARK(25);
  lds_barrier();  // #13
  SH_MARK(26);
  // ---- chunk 3
  flow_chunk();
  w_store();
  SH_MARK(27);
  lds_barrier();  // #14
  SH_MARK(28);
  mma_chunk();
  SH_MARK(29);
  epilogue();
#endif
}

static inline float bf16_round(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  float r;
  memcpy(&r, &u, 4);
  return r;
}
static inline unsigned short bf16_bits(float exact) {
  unsigned u;
  memcpy(&u, &exact, 4);
  return (unsigned short)(u >> 16);
}

template <bool FOLD>
int launch(const float *img0, const float *img1, const float *f0p, const float *f1p, const float *tmap, float tsc,
           const float *flow, const float *tmp_prev, int hp, int wp, float ips, float prev_scale, float *flow_out,
           const float *packed_w, const float *bias, float *out, int H, int W, int ho, int wo, hipStream_t s) {
  using G = Geo;
  static const int pad = getenv("DRBA_SH_PAD") ? atoi(getenv("DRBA_SH_PAD")) : 0;  // occupancy probe: extra LDS bytes per workgroup
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(stage_head_kernel<FOLD>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES + pad);
  if (attr != hipSuccess) return DRBA_ELAUNCH;
  const int tiles_x = (wo + G::TWO - 1) / G::TWO, tiles_y = (ho + G::THO - 1) / G::THO;
  DRBA_LAUNCH((stage_head_kernel<FOLD>), dim3(tiles_x * tiles_y), dim3(G::THREADS), G::LDS_BYTES + pad, s, img0, img1, f0p, f1p, tmap, tsc,
              flow, tmp_prev, hp, wp, ips, prev_scale, flow_out, reinterpret_cast<const u32x4 *>(packed_w), bias, out, H, W, ho, wo,
              tiles_x);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // namespace drba_sh

extern "C" {

size_t drba_stage_head_packed_floats(int Cout);
int drba_stage_head_pack(const float *w, float *packed, int Cout);

size_t drba_stage_head_packed_floats(int Cout) {
  using namespace drba_sh;
  if (Cout != 16) return 0;
  return (size_t)kChunks * kPairs * 3 * 64 * 4;  // 16-byte units of 4 floats
}

// w: conv0[0].weight [Cout, 52, 3, 3].  packed (16-byte units): [chunk][tap pair][nt][plane h/m/l][lane] = 8 bf16 of
//   w[nt*16 + (lane & 15)][kChunkMap[chunk][8*((lane >> 4) & 1) + i]][tap 2*pair + (lane >> 5)], i = 0..7 (0 for tap 9)
int drba_stage_head_pack(const float *w, float *packed, int Cout) {
  using namespace drba_sh;
  const size_t n = drba_stage_head_packed_floats(Cout);
  if (!w || !packed || n == 0) return DRBA_EINVAL;
  memset(packed, 0, n * sizeof(float));
  unsigned short *dst = reinterpret_cast<unsigned short *>(packed);
  const int NT = Cout / 16;
  for (int chunk = 0; chunk < kChunks; ++chunk)
    for (int p = 0; p < kPairs; ++p)
      for (int nt = 0; nt < NT; ++nt)
        for (int lane = 0; lane < 64; ++lane) {
          const int kq = lane >> 4, tap = 2 * p + (kq >> 1), co = nt * 16 + (lane & 15);
          if (tap > 8) continue;
          for (int i = 0; i < 8; ++i) {
            const int ci = kChunkMap[chunk][8 * (kq & 1) + i];
            if (ci < 0) continue;
            const float x = w[((size_t)co * 52 + ci) * 9 + tap];
            const float hh = bf16_round(x), mm = bf16_round(x - hh), ll = bf16_round(x - hh - mm);
            const float term[3] = {hh, mm, ll};
            for (int pl = 0; pl < 3; ++pl) {
              const size_t unit = ((((size_t)chunk * kPairs + p) * NT + nt) * 3 + pl);
              dst[(unit * 64 + lane) * 8 + i] = bf16_bits(term[pl]);
            }
          }
        }
  return DRBA_OK;
}

int drba_stage_head(const float *img0, const float *img1, const float *f0_pair, const float *f1_pair,
                    const float *timestep_map, float timestep_scalar, const float *flow, const float *tmp_prev, int hp,
                    int wp, float prev_scale, float *flow_out, const float *packed_w, const float *bias, float *out, int H,
                    int W, int h, int w, int Cout, float scale, void *stream) {
  using namespace drba_sh;
  if (!img0 || !img1 || !f0_pair || !f1_pair || !tmp_prev || !packed_w || !out || H <= 1 || W <= 1 || h <= 0 || w <= 0 ||
      !(scale > 0.f))
    return DRBA_EINVAL;
  if (hp <= 0 || wp <= 0 || !(prev_scale > 0.f)) return DRBA_EINVAL;
  if (!flow_out && !flow) return DRBA_EINVAL;             // without the fold the finished flow must be given
  if (Cout != 16) return DRBA_EUNSUPPORTED;  // block4: conv0[0] is 52 -> 16
  if (scale != 1.f) return DRBA_EUNSUPPORTED;  // one sample point per window pixel (the full-resolution stage)
  if (prev_scale != 2.f * scale) return DRBA_EUNSUPPORTED;  // IFNet's pyramid; bounds the staged footprint
  if (h != H || w != W) return DRBA_EINVAL;
  if ((W & 3) || W < 8 || H < 2) return DRBA_EUNSUPPORTED;  // 16-byte window loads and whole-row flow stores
  if (((uintptr_t)img0 | (uintptr_t)img1 | (uintptr_t)f0_pair | (uintptr_t)f1_pair | (uintptr_t)flow_out) & 15) return DRBA_EUNSUPPORTED;
  if (flow_out && flow_out == flow) return DRBA_EINVAL;  // halo pixels are read by neighbouring tiles
  const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
  const float ips = (float)(1.0 / (double)prev_scale);
  hipStream_t s = (hipStream_t)stream;
  if (flow_out)
    return launch<true>(img0, img1, f0_pair, f1_pair, timestep_map, timestep_scalar, flow, tmp_prev, hp, wp, ips, prev_scale,
                        flow_out, packed_w, bias, out, H, W, ho, wo, s);
  return launch<false>(img0, img1, f0_pair, f1_pair, timestep_map, timestep_scalar, flow, tmp_prev, hp, wp, ips, prev_scale,
                       flow_out, packed_w, bias, out, H, W, ho, wo, s);
}

#ifdef DRBA_SH_CLOCKS
int drba_stage_head_clocks(unsigned long long *host32, int reset) {
  if (reset) {
    unsigned long long z[32] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(drba_sh::g_sh_clk), z, sizeof(z)) == hipSuccess ? 0 : -1;
  }
  return hipMemcpyFromSymbol(host32, HIP_SYMBOL(drba_sh::g_sh_clk), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
