// EXPERIMENT (round 4), NOT the product kernel -- drba_amd/csrc/stage_conv.hip is.  Same entry points and results
// (tools/stage_conv_check.py, tests/gpu_checks.py green with it built in), kept for the record with its measurements:
//   * the bilinear taps of a tile are read from SOURCE BOXES staged in LDS by coalesced 16-byte LDS-DMA (bounding box of
//     the tile's tap coordinates, per source frame; per-lane gathers stay as the workgroup-uniform fallback), weights
//     streamed by LDS-DMA, everything requested two channel groups ahead by alternating halves of the workgroup;
//   * measured (profiles/r04_stage_conv_box_timing.txt, MI355X, 8 samples 1088x1920): box path 257 us per sample = gather
//     path 257 us (zero / gentle flows: every tile boxed), against 180-220 us for the product kernel; in the step 4.2 ms
//     per 8 samples against 2.5 ms (bench 770 against 823 frames/s);
//   * why (profiles/r04_stage_conv_box_clocks.txt, in-kernel clocks per workgroup, 42.6k ticks): the waits for the DMAs are
//     2 % -- the memory system is NOT what a workgroup waits for once requests run two groups ahead --, the 14-iteration
//     loop is 57 % (MFMA phase 670 ticks per group, sampling from LDS 520, barrier 330, DMA issue 270) and the phases in
//     front of it 41 %: footprint loads 6.1k, flow from the terms + taps + bounding boxes 8.0k, the first box's round
//     trip 3.6k.  With two 9-wave workgroups per CU those serial phases are not hidden; replacing the gathers changes
//     neither them nor the MFMA / LDS work of the loop, and sampling from LDS costs the LDS pipe what the gathers cost
//     the texture addresser.
// Build: tools/exp/stage_conv_clocks.sh (compiles this file in place of stage_conv.hip on the GPU box).
// The full-resolution stage input fused with the IFBlock's first convolution (IFNet_HDv3.py:85-88 -> conv0[0], :64-66):
//   x   = cat(warp(img0, flow[:2]), warp(img1, flow[2:4]), warp(f0, ..), warp(f1, ..), timestep, mask, feat, flow)   52 ch
//   y0  = LeakyReLU_0.2(conv3x3(x, stride 2, pad 1))                                                     52 -> 16 ch
// at scale 1 (the last stage of a scale-1.0 run).  Unfused, `ifblock_input_lds` writes x (52 x H x W floats: 435 MB per
// 1080p sample) and the stride-2 convolution reads it back: 1.74 GB of a step's HBM traffic for a tensor nothing else
// reads -- the two kernels are the first and the third entry of a step's single-stream time (0.34 + 0.27 of 3.5 ms).
// Here x exists only as LDS tiles:
//   * a workgroup (9 waves) owns 8 x 16 conv outputs = a 16 x 32 block of full-resolution sample points, one point per
//     lane of waves 0..7, plus the block's upper row and left column (the stride-2 window reaches one point up / left):
//     32 + 17 = 49 points on wave 8.  561 points for 512 outputs x 4: 1.096 of the gather work (one-point halo only).
//   * the 52 channels come in 13 groups of 4 in the order the gather produces them ({img0 x3, timestep}, {img1 x3, mask},
//     8 x {f0 pair, f1 pair}, feat 0..3, feat 4..7, flow) -- exactly the K = 4 of v_mfma_f32_16x16x4_f32.  Per group:
//     every lane parks its 4 values in a [4][17][33] window (double-buffered), one barrier, then waves 0..7 run the 9
//     taps of their 16-pixel output row on the matrix cores (A = window, read with the stride-2 column step; B = the
//     group's 9 weight fragments, resident in LDS for the whole kernel: 30 KB).  The loads of group k+1 are issued
//     before group k's barrier, so they fly under its MFMAs.
//   * the gathers are buffer loads: the per-point tap offsets are computed once (VGPR offset), the channel plane is the
//     instruction's scalar offset -- no 64-bit address arithmetic per load (the unfused kernel spends 2 VALU per load).
// The folded flow update (flow = flow_prev + up(tmp_prev[0:4]) * 2, ifblock_update's arithmetic) is written by the lanes
// that own a pixel, as in ifblock_input_lds<.., FOLD = true, ..>; per-point arithmetic is that kernel's, term by term.
// Exact fp32 products (fp32 MFMA): the result differs from the unfused pair only by the accumulation order.
#include "common.hpp"
#include "flow_terms.hpp"

#include <stdio.h>
#include <string.h>

using namespace drba;

namespace drba_stage_conv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int COUT = 16, CIN = 52, NG = 13;  // 13 channel groups of 4
constexpr int TOW = 16;                      // conv outputs per workgroup: TOH x 16 (one MFMA tile per output row)
constexpr int WC = 2 * TOW + 1;              // window columns
constexpr int RS = 34;                       // window row stride (floats)
constexpr int PC = 20;                       // tmp_prev footprint capacity, columns (33 points at half resolution + taps: <= 18)
constexpr int WG_FLOATS = 9 * 64;            // one group's weight fragments
constexpr int W_FLOATS = NG * WG_FLOATS;     // packed weights: [group][tap][lane] = w[cout = lane & 15][chan(group, lane >> 4)][tap]
// Source boxes (round 4): the bilinear taps of a tile's 17 x 33 sample points cover, per source frame, a box of source
// pixels that is hardly larger than the tile when the flow is smooth.  The box is copied global -> LDS with coalesced
// 16-byte LDS-DMA units and the taps are read from LDS; capacities:
constexpr int BW_F = 40, UW_F = BW_F / 2;    // feature box: pixels per row; [px][2] pair layout, a 16-byte unit = 2 pixels (34 + 1 + 5 of flow variation)
constexpr int BW_I = 48, UW_I = BW_I / 4;    // image box (planar): a 16-byte unit = 4 pixels (x origin a multiple of 4)
constexpr int NSLOT = 3;                     // box slots = weight buffers: DMAs are requested TWO channel groups ahead
constexpr int HALF = 256;                    // lanes of an issuing half (waves 0..3 / waves 4..7, alternating by group)
constexpr int kFeatDmas = 2;                 // DMA instructions per lane of a half for one source's feature box (Geo::FEAT_DMAS)
// TOH output rows: waves 0..TOH-1 own the 2 TOH x 32 block of sample points (and run output row `wave` on the matrix
// cores), wave TOH the upper row + left column.
template <int TOH_>
struct Geo {
  static constexpr int TOH = TOH_;
  static constexpr int WR = 2 * TOH + 1;         // window rows
  static constexpr int CS = WR * RS + 1;         // channel stride: odd, so the 4 channels of an A fragment fall on distinct banks
  static constexpr int THREADS = 64 * (TOH + 1);
  static constexpr int PR = TOH + 3;             // tmp_prev footprint capacity, rows
  static constexpr int WL = NSLOT * WG_FLOATS;   // weight fragments of three groups (streamed by LDS-DMA two groups ahead)
  static constexpr int TR = TOH / 2 + 4, TC = 12;  // term footprint capacity: (2 TOH + 1 rows, 33 columns) at >= 1/4 resolution
  static constexpr int TERM_FLOATS = kMaxTerms * 4 * TR * TC;
  static constexpr int BH = WR + 6;              // box rows: the window's rows, the second tap row, 5 rows of flow variation
  static constexpr int BOXF = BH * BW_F * 2;     // one source's feature-pair box (floats)
  static constexpr int BOXI = BH * BW_I;         // one image channel's box
  static constexpr int SLOT = (2 * BOXF > 3 * BOXI ? 2 * BOXF : 3 * BOXI);  // a group's boxes: two feature pairs, or three image channels
  static constexpr int IMG_DMAS = (3 * BH * UW_I + HALF - 1) / HALF;   // DMA instructions per lane of a half for an image group
  static constexpr int FEAT_DMAS = (BH * UW_F + HALF - 1) / HALF;      // ... for one source's feature-pair box
  static constexpr int PREV_C = 9;               // channels of tmp_prev kept for the whole kernel: mask, feat (4..12)
  static constexpr int LDS_FLOATS = WL + 2 * 4 * CS + PREV_C * PR * PC + NSLOT * SLOT + 16;
  static_assert(32 + WR <= 64, "upper row + left column on one wave");
  static_assert(TOH >= 8, "two issuing halves of four waves");
  static_assert(TERM_FLOATS <= SLOT, "the term footprints live in box slot 1 until the flow has been formed");
  static_assert(4 * PR * PC <= SLOT, "tmp_prev's flow channels live in box slot 2 until the flow has been formed");
  static_assert(WG_FLOATS / 4 <= HALF, "a group's weights are one DMA per lane of a half");
};

// channel of the stage input held by slot j of group g (the order ifblock_input_lds emits them in)
__host__ __device__ constexpr int chan_of(int g, int j) {
  return g == 0 ? (j < 3 ? j : 38) : g == 1 ? (j < 3 ? 3 + j : 39) : g < 10 ? ((j < 2 ? 6 : 22 - 2) + 2 * (g - 2) + j) : 40 + 4 * (g - 10) + j;
}
static_assert(chan_of(2, 0) == 6 && chan_of(2, 1) == 7 && chan_of(2, 2) == 22 && chan_of(2, 3) == 23, "pair groups");
static_assert(chan_of(9, 1) == 21 && chan_of(9, 3) == 37 && chan_of(10, 0) == 40 && chan_of(12, 3) == 51, "tail groups");

template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

struct StageItems {
  drba_stage_item_t it[DRBA_MAX_STAGE_ITEMS];
};

struct Raw {  // gather path: the loads of one channel group, in flight across the previous group's barrier
  u32x4 q[4];
  u32x2 h[6];
};

typedef __attribute__((address_space(3))) void *lds_ptr;
typedef short s16x2 __attribute__((ext_vector_type(2)));

// componentwise minimum of two packed int16 pairs over the wave (every lane gets the result)
__device__ __forceinline__ uint32_t wave_min_pk16(uint32_t v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)v, m, 64);
    const s16x2 a = __builtin_bit_cast(s16x2, v), b = __builtin_bit_cast(s16x2, o);
    v = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(a, b));
  }
  return v;
}
__device__ __forceinline__ uint32_t pk16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }

// FMODE: 0 = the finished flow is read; 1 = FOLD (flow_prev + the previous stage's update, written to flow_out); 2 = LAZY
// (the flow is the sum of the terms, flow_terms.hpp, + the previous stage's update; nothing but the convolution is written)
template <int FMODE, class G_>
__global__ void __launch_bounds__(G_::THREADS)
stage_conv0(const StageItems items, const FlowTermsArg T, const float *__restrict__ wpk, const float *__restrict__ bias, int hp, int wp,
            float inv_prev_scale, float prev_scale, int H, int W, int Ho, int Wo, int tiles_x, int box_allowed) {
#ifdef DRBA_SC_CLOCKS  // experiment builds (tools/exp/stage_conv_clocks.sh): where a wave's time goes, printed by two workgroups
  long long ck[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck_t = __builtin_readcyclecounter();
  const long long ck_start = ck_t;
#define DRBA_CK(i)                                      \
  do {                                                  \
    const long long now_ = __builtin_readcyclecounter(); \
    ck[i] += now_ - ck_t;                               \
    ck_t = now_;                                        \
  } while (0)
#else
#define DRBA_CK(i) do { } while (0)
#endif
  constexpr bool FOLD = FMODE != 0, WRITES = FMODE == 1, LAZY = FMODE == 2;
  constexpr int TOH = G_::TOH, WR = G_::WR, CS = G_::CS, THREADS = G_::THREADS, PR = G_::PR, BH = G_::BH;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *wl = lds;                         // [3][9][64]
  float *win = lds + G_::WL;               // [2][4][CS]
  float *prev = win + 2 * 4 * CS;          // [9][PR][PC]: channels 4..12 (mask, feat) of tmp_prev's footprint
  float *box = prev + G_::PREV_C * PR * PC;  // [3][SLOT]: the source boxes of three channel groups
  int *bi = reinterpret_cast<int *>(box + NSLOT * G_::SLOT);  // [8] bounding boxes of the tap coordinates
  float *tl = box + G_::SLOT;              // [kMaxTerms][4][TR * TC]: in slot 1 (read before the first box lands there)
  float *prevf = box + 2 * G_::SLOT;       // [4][PR][PC]: tmp_prev's flow channels, in slot 2 (read before a box lands there)
  // the item is picked by blockIdx.y out of the by-value argument: the compiler does not see that its fields are
  // wave-uniform (it would address every load per lane and wrap every buffer load in a waterfall loop) -- state it
  typedef __attribute__((address_space(1))) float *gptr;
  typedef __attribute__((address_space(1))) const float *cgptr;
  auto uniform = [](const float *p) -> gptr {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (gptr)(((uint64_t)hi << 32) | lo);
  };
  struct {
    cgptr img0, img1, f0_pair, f1_pair, timestep_map, flow, tmp_prev;
    gptr flow_out, out;
    float timestep_scalar;
    const float *term[kMaxTerms];
  } item;
  {
    const drba_stage_item_t &src = items.it[blockIdx.y];
    item.img0 = uniform(src.img0), item.img1 = uniform(src.img1), item.f0_pair = uniform(src.f0_pair), item.f1_pair = uniform(src.f1_pair);
    item.timestep_map = uniform(src.timestep_map), item.flow = uniform(src.flow), item.tmp_prev = uniform(src.tmp_prev);
    item.flow_out = uniform(src.flow_out), item.out = uniform(src.out);
    item.timestep_scalar = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(src.timestep_scalar)));
#pragma unroll
    for (int i = 0; i < kMaxTerms; ++i) item.term[i] = (const float *)uniform(src.term[i]);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t P = (size_t)H * W, p_prev = (size_t)hp * wp;
  int tx, ty;
  xcd_strip_tile(blockIdx.x, gridDim.x, tiles_x, tx, ty);
  const int ox0 = tx * TOW, oy0 = ty * TOH;
  const int X0 = 2 * ox0 - 1, Y0 = 2 * oy0 - 1;  // full-resolution coordinates of window (row 0, column 0)

  // ---- this lane's sample point
  int wr, wc;  // window row / column
  bool active = true;
  if (wave < TOH) {
    wr = 1 + 2 * wave + (lane >> 5), wc = 1 + (lane & 31);
  } else if (lane < 32) {
    wr = 0, wc = 1 + lane;
  } else {
    wr = min(lane - 32, WR - 1), wc = 0;
    active = lane - 32 < WR;
  }
  const int Xr = X0 + wc, Yr = Y0 + wr;
  const bool inimg = Xr >= 0 && Xr < W && Yr >= 0 && Yr < H;  // outside: the convolution's zero padding
  const int X = min(max(Xr, 0), W - 1), Y = min(max(Yr, 0), H - 1);
  const uint32_t q = (uint32_t)Y * W + X;
  const bool owner = wave < TOH && inimg;

  // ---- prologue loads, all issued before the first wait: the footprint of the window's sample points in tmp_prev, this
  // point's running flow and timestep (one memory latency instead of three in a row)
  constexpr int C0 = FOLD ? 0 : 4;  // first channel of tmp_prev that is needed
  const int Xa = max(X0, 0), Ya = max(Y0, 0), Xb = min(X0 + WC - 1, W - 1), Yb = min(Y0 + WR - 1, H - 1);
  const int rx0 = lerp_src(Xa, inv_prev_scale, wp).i0, ry0 = lerp_src(Ya, inv_prev_scale, hp).i0;
  const int rw = lerp_src(Xb, inv_prev_scale, wp).i1 - rx0 + 1, rh = lerp_src(Yb, inv_prev_scale, hp).i1 - ry0 + 1;
  const int pr_r = tid / PC, pr_c = tid - pr_r * PC;  // one (row, column) of the footprint per lane (PR * PC <= THREADS)
  const bool pr_on = pr_r < rh && pr_c < rw;
  float pv[13];
  {
    const cgptr tp = item.tmp_prev + (size_t)(ry0 + min(pr_r, rh - 1)) * wp + rx0 + min(pr_c, rw - 1);
#pragma unroll
    for (int c = C0; c < 13; ++c) pv[c] = tp[(size_t)c * p_prev];
  }
  float fr[4] = {0.f, 0.f, 0.f, 0.f};
  if (item.flow) {
    const cgptr fin = item.flow;
#pragma unroll
    for (int c = 0; c < 4; ++c) fr[c] = fin[(size_t)c * P + q];
  }
  const float tmv = item.timestep_map ? item.timestep_map[q] : item.timestep_scalar;
  if (tid < 8) bi[tid] = 0x7fffffff;
  if (pr_on) {
#pragma unroll
    for (int c = C0; c < 13; ++c) (c < 4 ? prevf + c * PR * PC : prev + (c - 4) * PR * PC)[pr_r * PC + pr_c] = pv[c];
  }
  int trx0[kMaxTerms], try0[kMaxTerms];
  if (LAZY) terms_stage<G_::TR, G_::TC, THREADS>(tl, T, item.term, Xa, Ya, Xb, Yb, tid, trx0, try0);
  __syncthreads();
  DRBA_CK(0);  // prologue loads + first barrier

  // taps of the previous head output's upsample at (X, Y), relative to the staged footprint
  const Lerp la = lerp_src(Y, inv_prev_scale, hp), lb = lerp_src(X, inv_prev_scale, wp);
  const int pr0 = (la.i0 - ry0) * PC, pr1 = (la.i1 - ry0) * PC, pc0 = lb.i0 - rx0, pc1 = lb.i1 - rx0;
  auto prev_up = [&](int c) -> float {  // c < 4: the flow channels
    const float *pp = prevf + c * PR * PC;
    return lerp2_fma(la.w0, la.w1, lb.w0, lb.w1, pp[pr0 + pc0], pp[pr0 + pc1], pp[pr1 + pc0], pp[pr1 + pc1]);
  };
  float fls[4];
  const bool have_terms = LAZY && terms_flow<G_::TR, G_::TC>(tl, T, trx0, try0, X, Y, fls);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (LAZY) {
      const float fd = __fmul_rn(prev_up(c), prev_scale);
      fls[c] = have_terms ? __fadd_rn(fls[c], fd) : fd;
    } else if (FOLD) {
      // ifblock_update: flow_in + up(tmp) * scale, product and sum rounded separately as torch evaluates them (and as
      // ifblock_input_lds does: the two kernels hand identical flows to warp_blend_fold)
      const float fd = __fmul_rn(prev_up(c), prev_scale);
      fls[c] = item.flow ? __fadd_rn(fr[c], fd) : fd;
    } else {
      fls[c] = fr[c];
    }
  }
  if (WRITES && owner) {
    const gptr fout = item.flow_out;
#pragma unroll
    for (int c = 0; c < 4; ++c) fout[(size_t)c * P + q] = fls[c];
  }
  const Taps t0 = taps_border(warp_coord(X, W, fls[0]), warp_coord(Y, H, fls[1]), W, H);
  const Taps t1 = taps_border(warp_coord(X, W, fls[2]), warp_coord(Y, H, fls[3]), W, H);
  // ifblock_input_lds' tap form: the pair of a row loaded at min(x0, W-2), the right-border case folded into the weights
  struct TapW {
    uint32_t o0, o1;  // gather path: element offsets of the two tap rows; box path: float offsets into the box
    float w00, w01, w10, w11;
  };
  auto tapw = [&](const Taps &t) -> TapW {
    const int xb = min(t.x0, W - 2);
    const bool edge = t.x0 != xb;
    TapW k;
    k.o0 = (uint32_t)(t.y0 * W + xb), k.o1 = (uint32_t)(t.y1 * W + xb);
    k.w00 = edge ? 0.f : t.wnw, k.w01 = edge ? t.wnw : t.wne;
    k.w10 = edge ? 0.f : t.wsw, k.w11 = edge ? t.wsw : t.wse;
    return k;
  };
  TapW k0 = tapw(t0), k1 = tapw(t1);
  // a point outside the image is the convolution's zero padding: its tap and upsample weights are zeroed once (x * 0 for
  // finite x) instead of selecting 0 for each of the 52 parked values
  const float zin = inimg ? 1.f : 0.f;
  if (!inimg) {
    k0.w00 = k0.w01 = k0.w10 = k0.w11 = 0.f;
    k1.w00 = k1.w01 = k1.w10 = k1.w11 = 0.f;
  }
  const float uw0 = la.w0 * zin, uw1 = la.w1 * zin;  // prev_up's row weights for the PARKED mask / feat (the flow fold used the true ones)
  auto prev_up_z = [&](int c) -> float {  // c >= 4: mask, feat
    const float *pp = prev + (c - 4) * PR * PC;
    return lerp2_fma(uw0, uw1, lb.w0, lb.w1, pp[pr0 + pc0], pp[pr0 + pc1], pp[pr1 + pc0], pp[pr1 + pc1]);
  };

  // ---- the boxes: bounding box of every lane's taps per source (every lane holds in-image taps: clamped points repeat
  // their neighbours'), packed int16 (x, y) minima of (x, y) and of (-x, -y); one LDS atomic per wave and value
  const int xb0 = min(t0.x0, W - 2), xb1 = min(t1.x0, W - 2);
  {
    const uint32_t m0 = wave_min_pk16(pk16(xb0, t0.y0)), n0 = wave_min_pk16(pk16(-xb0, -t0.y1));
    const uint32_t m1 = wave_min_pk16(pk16(xb1, t1.y0)), n1 = wave_min_pk16(pk16(-xb1, -t1.y1));
    if (lane == 0) {
      atomicMin(bi + 0, (int)(short)(m0 & 0xffffu)), atomicMin(bi + 1, (int)(short)(m0 >> 16));
      atomicMin(bi + 2, (int)(short)(n0 & 0xffffu)), atomicMin(bi + 3, (int)(short)(n0 >> 16));
      atomicMin(bi + 4, (int)(short)(m1 & 0xffffu)), atomicMin(bi + 5, (int)(short)(m1 >> 16));
      atomicMin(bi + 6, (int)(short)(n1 & 0xffffu)), atomicMin(bi + 7, (int)(short)(n1 >> 16));
    }
  }
  __syncthreads();  // (also: everybody has read the term footprints and tmp_prev's flow channels -- box slots 1 and 2 may be overwritten)
  DRBA_CK(1);  // flow, taps, bounding boxes + second barrier
  struct Box {
    int x0f, x0i, y0, h, uwf, uwi;  // x origin of the feature / image box, first row, rows, 16-byte units per row
    bool fits;
  };
  auto box_of = [&](int s) -> Box {
    const int minx = __builtin_amdgcn_readfirstlane(bi[4 * s]), miny = __builtin_amdgcn_readfirstlane(bi[4 * s + 1]);
    const int maxx = -__builtin_amdgcn_readfirstlane(bi[4 * s + 2]), maxy = -__builtin_amdgcn_readfirstlane(bi[4 * s + 3]);
    Box b;
    b.x0f = minx & ~1, b.x0i = minx & ~3, b.y0 = miny, b.h = maxy - miny + 1;
    const int wf = maxx + 2 - b.x0f, wi = maxx + 2 - b.x0i;  // pixels x0 .. maxx + 1
    b.uwf = (wf + 1) >> 1, b.uwi = (wi + 3) >> 2;
    b.fits = b.h <= BH && wf <= BW_F && wi <= BW_I;
    return b;
  };
  const Box b0 = box_of(0), b1 = box_of(1);
  const bool boxed = box_allowed && b0.fits && b1.fits;  // workgroup-uniform

  const uint32_t img_bytes = (uint32_t)(3 * P * 4), feat_bytes = (uint32_t)(16 * P * 4);
  const __amdgpu_buffer_rsrc_t r_i0 = __builtin_amdgcn_make_buffer_rsrc((void *)item.img0, 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_i1 = __builtin_amdgcn_make_buffer_rsrc((void *)item.img1, 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_f0 = __builtin_amdgcn_make_buffer_rsrc((void *)item.f0_pair, 0, feat_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_f1 = __builtin_amdgcn_make_buffer_rsrc((void *)item.f1_pair, 0, feat_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void *)wpk, 0, (uint32_t)(W_FLOATS * 4), 0x00020000);
  const uint32_t plane = (uint32_t)(P * 4);

  // DMAs are issued by one HALF of the workgroup per channel group (waves 0..3 for even, waves 4..7 for odd groups), two
  // groups ahead of their use, and waited for -- s_waitcnt vmcnt(0) in front of a barrier -- by the half that issued them,
  // one iteration later: a half never waits for the requests it has just made, so every request has two iterations
  // (MFMAs + sampling of two groups, by two resident workgroups) to cross the memory system.  (Round 4's first version
  // requested one group ahead and waited in the same iteration: 265 us per 1080p sample, every iteration one exposed
  // round trip -- the gather path at 260 us was bound the same way by its weights' DMA.)
  const int hl = tid & (HALF - 1), hw = wave & 3;   // lane / wave inside its half
  const bool half0 = wave < 4, half1 = wave >= 4 && wave < 8;
  auto my_turn = [&](int g) -> bool { return (g & 1) ? half1 : half0; };  // does this wave issue for channel group g?
  // a group's weight fragments, global -> LDS (2304 bytes: lanes 0..143 of the half, 16 bytes each)
  auto issue_w = [&](int g) {
    if (hl < WG_FLOATS / 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_ptr)(wl + (g % NSLOT) * WG_FLOATS + hw * 256), 16, (uint32_t)hl * 16u,
                                               (uint32_t)(g * WG_FLOATS * 4), 0, 0);
  };
  const int park = wr * RS + wc;
  const int a_off = (lane >> 4) * CS + (2 * wave) * RS + 2 * (lane & 15);  // waves 0..TOH-1: output row `wave`, pixels lane & 15
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // the 9 taps of channel group g on the matrix cores: A = the parked window, B = the group's weight fragments
  auto mfma_group = [&](int g) {
    if (wave < TOH) {
      const float *ab = win + (g & 1) * 4 * CS + a_off;
      const float *bb = wl + (g % NSLOT) * WG_FLOATS + lane;
      float a[9], b[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) a[t] = ab[(t / 3) * RS + (t % 3)], b[t] = bb[t * 64];
#pragma unroll
      for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc, 0, 0, 0);
    }
  };
  auto park_values = [&](int g, const float (&v)[4]) {
    float *wb = win + (g & 1) * 4 * CS;
    if (active) {
#pragma unroll
      for (int c = 0; c < 4; ++c) wb[c * CS + park] = v[c];
    }
  };
  // the channel groups that are not gathered from a frame: feat 0..3, feat 4..7 (upsampled head output), the flow
  auto tail_values = [&](auto G, float (&v)[4]) {
    constexpr int g = decltype(G)::value;
    if constexpr (g < 12) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = prev_up_z(5 + 4 * (g - 10) + c);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = inimg ? fls[c] : 0.f;  // interpolate(flow) * 1. / scale at scale 1: the flow itself
    }
  };

  if (boxed) {
    // ================================================================ box path
    // Iteration g: the boxes of group g + 2 and the weights of group g + 1 are requested (LDS-DMA) by this iteration's half,
    // group g - 1 runs on the matrix cores, group g is sampled from its boxes and parked; ONE barrier per group, in front of
    // which the OTHER half waits for what it requested an iteration ago (the boxes of group g + 1, the weights of group g).
    // Slot (g + 2) % 3 held group g - 1's boxes (sampled before the previous barrier), the weight buffers rotate the same way.
    // per-lane DMA source offsets (bytes): unit u of a feature box is (row u / UW_F, 16-byte column u % UW_F); rows / columns
    // beyond the box are not requested, coordinates beyond the frame are clamped (never sampled)
    // (sized by a namespace constant: with the dependent G_::FEAT_DMAS as the bound, hipcc 7.2's host pass silently dropped
    // the kernel's stub -- the object linked, the library failed to load with an undefined kernel symbol)
    static_assert(G_::FEAT_DMAS == kFeatDmas, "feature box DMAs per lane");
    uint32_t fo0[kFeatDmas], fo1[kFeatDmas];
    bool fon0[kFeatDmas], fon1[kFeatDmas];
#pragma unroll
    for (int i = 0; i < kFeatDmas; ++i) {
      const int u = hl + i * HALF;
      const int r = u / UW_F, c = u - r * UW_F;
      fon0[i] = r < b0.h && c < b0.uwf, fon1[i] = r < b1.h && c < b1.uwf;
      fo0[i] = (uint32_t)(min(b0.y0 + r, H - 1) * W + min(b0.x0f + 2 * c, W - 2)) * 8u;
      fo1[i] = (uint32_t)(min(b1.y0 + r, H - 1) * W + min(b1.x0f + 2 * c, W - 2)) * 8u;
    }
    auto issue_img = [&](const __amdgpu_buffer_rsrc_t &rs, const Box &b, float *slot) {
#pragma unroll
      for (int i = 0; i < G_::IMG_DMAS; ++i) {
        const int u = hl + i * HALF;
        const int ch = u / (BH * UW_I), e = u - ch * (BH * UW_I);
        const int r = e / UW_I, c = e - r * UW_I;
        if (ch < 3 && r < b.h && c < b.uwi)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(slot + (i * HALF + hw * 64) * 4), 16,
                                                   (uint32_t)(min(b.y0 + r, H - 1) * W + min(b.x0i + 4 * c, W - 4)) * 4u + (uint32_t)ch * plane, 0, 0, 0);
      }
    };
    auto issue_box = [&](auto G) {
      constexpr int g = decltype(G)::value;
      float *slot = box + (g % NSLOT) * G_::SLOT;
      if constexpr (g == 0) issue_img(r_i0, b0, slot);
      else if constexpr (g == 1) issue_img(r_i1, b1, slot);
      else if constexpr (g < 10) {
        constexpr int c2 = g - 2;  // [C/2, H, W, 2]: the pair's plane is 2P floats
#pragma unroll
        for (int i = 0; i < kFeatDmas; ++i) {
          if (fon0[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_f0, (lds_ptr)(slot + (i * HALF + hw * 64) * 4), 16, fo0[i], c2 * 2 * plane, 0, 0);
          if (fon1[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_f1, (lds_ptr)(slot + G_::BOXF + (i * HALF + hw * 64) * 4), 16, fo1[i], c2 * 2 * plane, 0, 0);
        }
      }
    };
    // sample offsets (floats) of this lane's taps inside the boxes
    const int sf0a = ((t0.y0 - b0.y0) * BW_F + (xb0 - b0.x0f)) * 2, sf0b = ((t0.y1 - b0.y0) * BW_F + (xb0 - b0.x0f)) * 2;
    const int sf1a = ((t1.y0 - b1.y0) * BW_F + (xb1 - b1.x0f)) * 2, sf1b = ((t1.y1 - b1.y0) * BW_F + (xb1 - b1.x0f)) * 2;
    const int si0a = (t0.y0 - b0.y0) * BW_I + (xb0 - b0.x0i), si0b = (t0.y1 - b0.y0) * BW_I + (xb0 - b0.x0i);
    const int si1a = (t1.y0 - b1.y0) * BW_I + (xb1 - b1.x0i), si1b = (t1.y1 - b1.y0) * BW_I + (xb1 - b1.x0i);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto sample_group = [&](auto G, float (&v)[4]) {
      constexpr int g = decltype(G)::value;
      const float *slot = box + (g % NSLOT) * G_::SLOT;
      if constexpr (g < 2) {
        const TapW &k = g == 0 ? k0 : k1;
        const int oa = g == 0 ? si0a : si1a, ob = g == 0 ? si0b : si1b;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float *pc = slot + c * G_::BOXI;
          v[c] = pc[oa] * k.w00 + pc[oa + 1] * k.w01 + pc[ob] * k.w10 + pc[ob + 1] * k.w11;
        }
        v[3] = g == 0 ? (inimg ? tmv : 0.f) : prev_up_z(4);
      } else if constexpr (g < 10) {
        auto pair = [&](const float *pb, int oa, int ob, const TapW &k, float &v0, float &v1) {
          const f32x2 a = *reinterpret_cast<const f32x2 *>(pb + oa), a2 = *reinterpret_cast<const f32x2 *>(pb + oa + 2);
          const f32x2 b = *reinterpret_cast<const f32x2 *>(pb + ob), b2 = *reinterpret_cast<const f32x2 *>(pb + ob + 2);
          v0 = a.x * k.w00 + a2.x * k.w01 + b.x * k.w10 + b2.x * k.w11;
          v1 = a.y * k.w00 + a2.y * k.w01 + b.y * k.w10 + b2.y * k.w11;
        };
        pair(slot, sf0a, sf0b, k0, v[0], v[1]);
        pair(slot + G_::BOXF, sf1a, sf1b, k1, v[2], v[3]);
      } else {
        tail_values(G, v);
      }
    };
    if (half0) {  // "iteration -2": group 0's boxes, waited for right away (the one exposed round trip of the workgroup)
      issue_box(std::integral_constant<int, 0>{});
    } else if (half1) {  // "iteration -1": group 1's boxes and group 0's weights
      issue_box(std::integral_constant<int, 1>{});
      issue_w(0);
    }
    if (half0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DRBA_CK(2);  // first boxes requested and landed
    static_for<NG + 1>([&](auto G) {
      constexpr int g = decltype(G)::value;
      if (my_turn(g)) {
        if constexpr (g + 2 < 10) issue_box(std::integral_constant<int, g + 2>{});
        if constexpr (g + 1 < NG) issue_w(g + 1);
      }
      DRBA_CK(3);
      if constexpr (g >= 1) mfma_group(g - 1);
      DRBA_CK(4);
      if constexpr (g < NG) {
        float v[4];
        sample_group(G, v);
        park_values(g, v);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DRBA_CK(5);
        if (my_turn(g + 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // requested during iteration g - 1
        DRBA_CK(6);
        __builtin_amdgcn_s_barrier();
        DRBA_CK(7);
      }
    });
  } else {
    // ================================================================ gather path (tiles whose taps do not fit the boxes:
    // flow discontinuities, and frames whose rows are not 16-byte multiples): every lane loads its own taps -- buffer loads,
    // the tap offsets in a VGPR, the channel plane as the scalar offset.  Same loop: the weights are requested two groups
    // ahead by alternating halves, the loads of group g + 1 stay in flight across group g's barrier (the wait in front of
    // it covers the weights only: they were requested before those loads).
    auto issue = [&](auto G, Raw &r) {
      constexpr int g = decltype(G)::value;
      if constexpr (g < 2) {
        const __amdgpu_buffer_rsrc_t &rs = g == 0 ? r_i0 : r_i1;
        const TapW &k = g == 0 ? k0 : k1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          r.h[2 * c] = __builtin_amdgcn_raw_buffer_load_b64(rs, k.o0 * 4u, c * plane, 0);
          r.h[2 * c + 1] = __builtin_amdgcn_raw_buffer_load_b64(rs, k.o1 * 4u, c * plane, 0);
        }
      } else if constexpr (g < 10) {
        constexpr int c2 = g - 2;
        r.q[0] = __builtin_amdgcn_raw_buffer_load_b128(r_f0, k0.o0 * 8u, c2 * 2 * plane, 0);
        r.q[1] = __builtin_amdgcn_raw_buffer_load_b128(r_f0, k0.o1 * 8u, c2 * 2 * plane, 0);
        r.q[2] = __builtin_amdgcn_raw_buffer_load_b128(r_f1, k1.o0 * 8u, c2 * 2 * plane, 0);
        r.q[3] = __builtin_amdgcn_raw_buffer_load_b128(r_f1, k1.o1 * 8u, c2 * 2 * plane, 0);
      }
    };
    auto finish = [&](auto G, const Raw &r, float (&v)[4]) {
      constexpr int g = decltype(G)::value;
      if constexpr (g < 2) {
        const TapW &k = g == 0 ? k0 : k1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float ax = __uint_as_float(r.h[2 * c].x), ay = __uint_as_float(r.h[2 * c].y);
          const float bx = __uint_as_float(r.h[2 * c + 1].x), by = __uint_as_float(r.h[2 * c + 1].y);
          v[c] = ax * k.w00 + ay * k.w01 + bx * k.w10 + by * k.w11;
        }
        v[3] = g == 0 ? (inimg ? tmv : 0.f) : prev_up_z(4);
      } else if constexpr (g < 10) {
        auto pair = [&](const u32x4 &a, const u32x4 &b, const TapW &k, float &v0, float &v1) {
          v0 = __uint_as_float(a.x) * k.w00 + __uint_as_float(a.z) * k.w01 + __uint_as_float(b.x) * k.w10 + __uint_as_float(b.z) * k.w11;
          v1 = __uint_as_float(a.y) * k.w00 + __uint_as_float(a.w) * k.w01 + __uint_as_float(b.y) * k.w10 + __uint_as_float(b.w) * k.w11;
        };
        pair(r.q[0], r.q[1], k0, v[0], v[1]);
        pair(r.q[2], r.q[3], k1, v[2], v[3]);
      } else {
        tail_values(G, v);
      }
    };
    Raw raw[2];
    if (half1) issue_w(0);
    issue(std::integral_constant<int, 0>{}, raw[0]);
    static_for<NG + 1>([&](auto G) {
      constexpr int g = decltype(G)::value;
      if constexpr (g + 1 < NG) {
        if (my_turn(g)) issue_w(g + 1);
      }
      if constexpr (g + 1 < 10) issue(std::integral_constant<int, g + 1>{}, raw[(g + 1) & 1]);
      DRBA_CK(3);
      if constexpr (g >= 1) mfma_group(g - 1);
      DRBA_CK(4);
      if constexpr (g < NG) {
        float v[4];
        finish(G, raw[g & 1], v);
        park_values(g, v);
        // the weights of group g (requested during iteration g - 1, before the 6 / 4 / 0 gather loads of group g + 1 that
        // may stay in flight) have landed
        constexpr int kInFlight = g + 1 < 2 ? 6 : g + 1 < 10 ? 4 : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DRBA_CK(5);
        if (my_turn(g + 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kInFlight) : "memory");
        DRBA_CK(6);
        __builtin_amdgcn_s_barrier();
        DRBA_CK(7);
      }
    });
  }
#ifdef DRBA_SC_CLOCKS
  if ((blockIdx.x == 700 || blockIdx.x == 2001) && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == 5 || wave == 8))
    printf("wg %d wave %d %s: prologue %lld  flow+box %lld  first box %lld | per 14 iterations: issue %lld  mfma %lld  sample+park %lld  vmcnt %lld  barrier %lld | total %lld\n",
           (int)blockIdx.x, wave, boxed ? "box" : "gather", ck[0], ck[1], ck[2], ck[3], ck[4], ck[5], ck[6], ck[7],
           (long long)__builtin_readcyclecounter() - ck_start);
#endif

  // ---- epilogue: bias, LeakyReLU(0.2), 4 consecutive pixels of one output channel per lane
  if (wave < TOH) {
    const int co = lane & 15, oy = oy0 + wave, ox = ox0 + 4 * (lane >> 4);
    if (oy < Ho && ox < Wo) {
      const float bs = bias ? bias[co] : 0.f;
      const gptr dst = item.out + ((size_t)co * Ho + oy) * Wo + ox;
      f32x4 y;
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = lrelu02(acc[k] + bs);
      if ((Wo & 3) == 0) {
        *(__attribute__((address_space(1))) f32x4 *)dst = y;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (ox + k < Wo) dst[k] = y[k];
      }
    }
  }
}

}  // namespace drba_stage_conv

extern "C" {

size_t drba_stage_conv0_packed_floats(void) { return (size_t)drba_stage_conv::W_FLOATS; }

int drba_stage_conv0_pack(const float *w, float *packed) {
  using namespace drba_stage_conv;
  if (!w || !packed) return DRBA_EINVAL;
  for (int g = 0; g < NG; ++g)
    for (int t = 0; t < 9; ++t)
      for (int l = 0; l < 64; ++l) {
        const int co = l & 15, ci = chan_of(g, l >> 4);
        packed[(g * 9 + t) * 64 + l] = w[((size_t)co * CIN + ci) * 9 + t];
      }
  return DRBA_OK;
}

int drba_stage_conv0_supported(int H, int W, float scale, float prev_scale, int Cout) {
  return (H >= 2 && W >= 2 && scale == 1.f && prev_scale == 2.f && Cout == drba_stage_conv::COUT) ? 1 : 0;
}

int drba_stage_conv0_batch(const drba_stage_item_t *items, int n_items, const drba_flow_terms_t *terms, int hp, int wp, float prev_scale,
                           int H, int W, const float *packed_w, const float *bias, void *stream) {
  using namespace drba_stage_conv;
  if (!items || n_items <= 0 || n_items > DRBA_MAX_STAGE_ITEMS || !packed_w || H < 2 || W < 2 || hp <= 0 || wp <= 0) return DRBA_EINVAL;
  if (prev_scale != 2.f) return DRBA_EUNSUPPORTED;  // IFNet's pyramid: the stage before scale 1 ran at scale 2 (bounds the staged footprint)
  if ((uint64_t)H * W * 16 * 4 >= (1ull << 32)) return DRBA_EUNSUPPORTED;  // buffer-load offsets are 32-bit
  StageItems its;
  memset(&its, 0, sizeof(its));
  const bool lazy = terms != nullptr;
  const bool fold = !lazy && items[0].flow_out != nullptr;
  FlowTermsArg T;
  if (!flow_terms_arg(terms, T)) return DRBA_EINVAL;
  for (int i = 0; i < T.n; ++i)
    if (T.scale[i] < 4.f) return DRBA_EUNSUPPORTED;  // earlier stages of the pyramid only (bounds their footprints)
  for (int k = 0; k < n_items; ++k) {
    const drba_stage_item_t &I = items[k];
    if (!I.img0 || !I.img1 || !I.f0_pair || !I.f1_pair || !I.tmp_prev || !I.out) return DRBA_EINVAL;
    if (lazy) {
      if (I.flow || I.flow_out) return DRBA_EINVAL;
      for (int i = 0; i < T.n; ++i)
        if (!I.term[i]) return DRBA_EINVAL;
    } else {
      if ((I.flow_out != nullptr) != fold || (!fold && !I.flow)) return DRBA_EINVAL;
      if ((I.flow == nullptr) != (items[0].flow == nullptr)) return DRBA_EINVAL;
    }
    its.it[k] = I;
  }
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const float ips = 0.5f;
  hipStream_t s = (hipStream_t)stream;
  // the box path copies 16-byte units: rows that are 16-byte multiples in both layouts, 16-byte aligned planes, coordinates
  // that fit the packed int16 reduction; anything else runs the gather path on every tile
  int box_allowed = (W % 4 == 0 && W >= 8 && W <= 32000 && H <= 32000) ? 1 : 0;
  for (int k = 0; k < n_items && box_allowed; ++k)
    if ((((uintptr_t)items[k].img0 | (uintptr_t)items[k].img1 | (uintptr_t)items[k].f0_pair | (uintptr_t)items[k].f1_pair) & 15) != 0) box_allowed = 0;
  if (((uintptr_t)packed_w & 15) != 0) return DRBA_EINVAL;
  static const int box_env = env_int("DRBA_SC_BOX", 1);  // TUNING builds: DRBA_SC_BOX = 0 runs the gather path on every tile
  if (!box_env) box_allowed = 0;
#ifdef DRBA_TUNING_SWITCHES  /* measuring builds: resident workgroups per CU of each instantiation, once */
#define DRBA_SC_OCC(FO, G_)                                                                                                   \
  do {                                                                                                                        \
    static bool told = false;                                                                                                 \
    if (!told) {                                                                                                              \
      int nb = -1;                                                                                                            \
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, stage_conv0<FO, G_>, G_::THREADS, lds_bytes);                    \
      fprintf(stderr, "stage_conv0<%d>: %d workgroups of %d threads per CU with %zu bytes of LDS\n", FO, nb, G_::THREADS, lds_bytes); \
      told = true;                                                                                                            \
    }                                                                                                                         \
  } while (0)
#else
#define DRBA_SC_OCC(FO, G_) do { } while (0)
#endif
#define DRBA_SC_GO(FO, TT)                                                                                                 \
  do {                                                                                                                     \
    using G_ = Geo<TT>;                                                                                                    \
    constexpr size_t lds_bytes = (size_t)G_::LDS_FLOATS * 4;                                                               \
    if (max_dynamic_lds((const void *)stage_conv0<FO, G_>, (int)lds_bytes) != hipSuccess) return DRBA_ELAUNCH;             \
    DRBA_SC_OCC(FO, G_);                                                                                                   \
    const int tiles_x = (Wo + TOW - 1) / TOW, tiles_y = (Ho + TT - 1) / TT;                                                \
    DRBA_LAUNCH((stage_conv0<FO, G_>), dim3(tiles_x * tiles_y, n_items), dim3(G_::THREADS), lds_bytes, s, its, T, packed_w, bias, \
                hp, wp, ips, prev_scale, H, W, Ho, Wo, tiles_x, box_allowed);                                              \
  } while (0)
#define DRBA_SC_GO2(T_)          \
  do {                           \
    if (lazy) DRBA_SC_GO(2, T_); \
    else if (fold) DRBA_SC_GO(1, T_); \
    else DRBA_SC_GO(0, T_);      \
  } while (0)
  DRBA_SC_GO2(8);  // (4- and 6-row workgroups were measured in round 3 and lost: 196-201 us against 196)
#undef DRBA_SC_GO2
#undef DRBA_SC_GO
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // extern "C"
