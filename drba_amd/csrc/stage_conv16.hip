// The scale-1 stage input fused with the IFBlock's first convolution (IFNet_HDv3.py:85-88 -> conv0[0], :64-66), two-term fp16
// form (kernel family 4; stage_conv.hip is the exact-fp32 form of the same fusion: same inputs, same outputs, same flows).
//   x   = cat(warp(img0, flow[:2]), warp(img1, flow[2:4]), warp(f0, ..), warp(f1, ..), timestep, mask, feat, flow)   52 ch
//   y0  = LeakyReLU_0.2(conv3x3(x, stride 2, pad 1))                                                     52 -> COUT ch
// What changes against stage_conv.hip is the convolution's arithmetic and, for it, the shape of the channel loop:
//   * every gathered value is taken as h + 2^-11 l with two fp16 terms (conv_split.hip "Two-term form": x' = x / 16,
//     h = fp16(x'), l = fp16((x' - h) * 2^11)) and the convolution runs on v_mfma_f32_16x16x32_f16: K = 32 = four (tap, octet of
//     channels) units, three products per K step (h h, h l, l h; joined as hh + (hl + lh) / 2048).  18 K steps x 3 MFMAs of ~16
//     clocks per 16-pixel output row where the fp32 form issues 117 MFMAs of 32 clocks;
//   * the 52 channels come in FOUR groups instead of 13 -- {img0 x3, timestep, img1 x3, mask | flow x4, 0 x4},
//     {f0 0..7 | f1 0..7}, {f0 8..15 | f1 8..15}, {feat 0..7} -- two octets of channels each (the last: one).  A group's window
//     lives in LDS as [plane h, l][octet][position][8 x fp16]: a lane parks an octet of its sample point with ONE 16-byte
//     write per plane (4 ds_write_b128 per group instead of 16 ds_write_b32), an MFMA operand is ONE 16-byte read.  Window
//     columns are stored split by parity (even columns first): the stride-2 reads of 16 neighbouring outputs are 256
//     contiguous bytes, and the two octets of a tap sit a multiple of 256 bytes apart (conflict-free ds_read_b128 groups);
//   * a workgroup is 8 waves for a (2 TOH + 1) x 33 window of sample points, one point per lane in row-major order (495 of
//     512 lanes at TOH = 7: no ninth "halo" wave), waves 0..TOH-1 run output row `wave` on the matrix cores;
//   * one window buffer (35 KB) and one group of weight fragments (10 KB) at a time: 62 KB per workgroup, two per CU.  Per
//     group: park, barrier, MFMAs, barrier -- 8 barriers instead of 13, the next group's gathers (16 x 16 bytes per lane)
//     in flight across both.
// The sampling arithmetic (flow from the terms / the fold, taps, bilinear products) is stage_conv.hip's, term by term: the
// flows it hands on are bit-identical to ifblock_input_lds'.
#include "common.hpp"
#include "conv_split.hpp"
#include "flow_terms.hpp"

#include <string.h>

using namespace drba;

namespace drba_stage_conv16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr;

constexpr int CIN = 52;
constexpr int TOW = 16;             // conv outputs per tile row (one MFMA M tile)
constexpr int WC = 2 * TOW + 1;     // window columns
[[maybe_unused]] constexpr int PS = 20;              // units from a window row's even columns (17) to its odd ones (16): = 4 mod 8, the 8 lanes of a ds_write_b128 group land on distinct banks
constexpr int RS = 36;              // window row stride (16-byte units)
constexpr int PC = 20;              // tmp_prev footprint capacity, columns
constexpr int NG = 4;               // channel groups
constexpr int KS_TOTAL = 18;        // K steps: 5 + 5 + 5 + 3
__host__ __device__ constexpr int ks_first(int g) { return 5 * g; }
__host__ __device__ constexpr int ks_count(int g) { return g < 3 ? 5 : 3; }
[[maybe_unused]] constexpr float kScale = 1.f / (float)(1 << kSplitActShift), kUnscale = (float)(1 << kSplitActShift);

// stage-input channel held by element i of octet o of group g (-1: padding, zero weights)
__host__ __device__ constexpr int chan_of(int g, int o, int i) {
  return g == 0 ? (o == 0 ? (i < 3 ? i : i == 3 ? 38 : i < 7 ? i - 1 : 39) : (i < 4 ? 48 + i : -1))
       : g == 1 ? (o == 0 ? 6 + i : 22 + i)
       : g == 2 ? (o == 0 ? 14 + i : 30 + i)
                : (o == 0 ? 40 + i : -1);
}
// (tap, octet) of K unit kq of K step j of group g; tap 9 = padding (zero weights)
__host__ __device__ constexpr int unit_tap(int g, int j, int kq) { return g < 3 ? (2 * j + (kq >> 1) < 9 ? 2 * j + (kq >> 1) : 9) : (4 * j + kq < 9 ? 4 * j + kq : 9); }
__host__ __device__ constexpr int unit_octet(int g, int kq) { return g < 3 ? (kq & 1) : 0; }
static_assert(chan_of(0, 0, 3) == 38 && chan_of(0, 0, 4) == 3 && chan_of(0, 0, 7) == 39 && chan_of(0, 1, 3) == 51, "group 0");
static_assert(chan_of(1, 1, 7) == 29 && chan_of(2, 0, 0) == 14 && chan_of(2, 1, 7) == 37 && chan_of(3, 0, 7) == 47, "feature groups");

template <int TOH_, int NT_>
struct Geo {
  static constexpr int TOH = TOH_, NT = NT_, COUT = 16 * NT_;
  static constexpr int WR = 2 * TOH + 1;                       // window rows
  static constexpr int NPOINTS = WR * WC;
  static constexpr int THREADS = (NPOINTS + 63) / 64 * 64;
  static constexpr int NPOS = (WR * RS + 15) / 16 * 16;        // units per (plane, octet): a multiple of 16, the two octets of a tap on the same banks
  static constexpr int WG_UNITS = 5 * 2 * NT * 64;             // one group's weight fragments: [K step][plane][n tile][lane]
  static constexpr int WD = (WG_UNITS / 64 + THREADS / 64 - 1) / (THREADS / 64);  // LDS-DMA instructions (1 KB each) per wave and group
  static constexpr int PR = TOH + 3;                           // tmp_prev footprint capacity, rows
  static constexpr int TR = TOH / 2 + 4, TC = 12;              // term footprint capacity (terms at >= 1/4 resolution)
  static constexpr int U_WIN = 0, U_W = 4 * NPOS, U_PREV = U_W + WG_UNITS, U_TERM = U_PREV + PR * PC * 4;
  static constexpr int LDS_UNITS = U_TERM + kMaxTerms * TR * TC;
  static_assert(PR * PC <= THREADS, "one footprint pixel per lane");
  static_assert(TOH <= THREADS / 64, "one output row per wave");
};

struct StageItems {
  drba_stage_item_t it[DRBA_MAX_STAGE_ITEMS];
};

__device__ __forceinline__ void lds_barrier() {
  // LDS traffic of this wave done, then the workgroup barrier; global loads stay in flight (no vmcnt wait)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// (a, b) * 2^-shift -> packed h and packed (remainder * 2^11)
__device__ __forceinline__ void split2(float a, float b, uint32_t &h, uint32_t &l) {
  const f32x2 v = (f32x2){a, b} * kScale;
  const f16x2 hh = __builtin_convertvector(v, f16x2);
  const f32x2 r = (v - __builtin_convertvector(hh, f32x2)) * 2048.f;
  const f16x2 ll = __builtin_convertvector(r, f16x2);
  h = __builtin_bit_cast(uint32_t, hh);
  l = __builtin_bit_cast(uint32_t, ll);
}

// Experiment builds only (-DDRBA_SC16_CLOCKS): s_memtime stamps at the phase boundaries of workgroups 700 / 2001, waves 0, 3, 7
#ifdef DRBA_SC16_CLOCKS
#define SC16_CLK(i) clk[i] = (long long)__builtin_readcyclecounter()
#else
#define SC16_CLK(i)
#endif

// FMODE: 0 = the finished flow is read; 1 = FOLD (flow_prev + the previous stage's update, written to flow_out); 2 = LAZY
// (the flow is the sum of the terms, flow_terms.hpp, + the previous stage's update; nothing but the convolution is written)
template <int FMODE, class G_, bool X4>
__global__ void __launch_bounds__(G_::THREADS, 4)  // 4 waves per SIMD = two workgroups per CU: <= 128 registers
stage_conv16(const StageItems items, const FlowTermsArg T, const u32x4 *__restrict__ wpk, const float *__restrict__ bias, int hp, int wp,
             float inv_prev_scale, float prev_scale, int H, int W, int Ho, int Wo, int tiles_x, int n_items, unsigned char *status) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool FOLD = FMODE != 0, WRITES = FMODE == 1, LAZY = FMODE == 2;
  constexpr int TOH = G_::TOH, NT = G_::NT, WR = G_::WR, THREADS = G_::THREADS, NPOS = G_::NPOS, NPOINTS = G_::NPOINTS;
  extern __shared__ __attribute__((aligned(16))) u32x4 lds16[];
  u32x4 *win = lds16 + G_::U_WIN;                               // [plane][octet][NPOS]
  u32x4 *wl = lds16 + G_::U_W;                                  // [K step][plane][n tile][64]
  float *prev = reinterpret_cast<float *>(lds16 + G_::U_PREV);  // [PR][PC][16]: flow | mask, - | feat 0..3 | feat 4..7
  float *tl = reinterpret_cast<float *>(lds16 + G_::U_TERM);    // [kMaxTerms][TR * TC][4]
  int vb_, vitem_, ntiles_;
  tile_item_block(n_items, vb_, vitem_, ntiles_);  // one grid dimension: the items of a tile back to back on one XCD (common.hpp)
  typedef __attribute__((address_space(1))) float *gptr;
  typedef __attribute__((address_space(1))) const float *cgptr;
  auto uniform = [](const float *p) -> gptr {  // the item is picked by the block id: state that its fields are wave-uniform
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (gptr)(((uint64_t)hi << 32) | lo);
  };
  struct {
    cgptr img0, img1, img0_x4, img1_x4, f0_pair, f1_pair, timestep_map, flow, tmp_prev;
    gptr flow_out, out;
    float timestep_scalar;
    const float *term[kMaxTerms];
  } item;
  {
    const drba_stage_item_t &src = items.it[vitem_];
    item.img0 = uniform(src.img0), item.img1 = uniform(src.img1), item.img0_x4 = uniform(src.img0_x4), item.img1_x4 = uniform(src.img1_x4), item.f0_pair = uniform(src.f0_pair), item.f1_pair = uniform(src.f1_pair);
    item.timestep_map = uniform(src.timestep_map), item.flow = uniform(src.flow), item.tmp_prev = uniform(src.tmp_prev);
    item.flow_out = uniform(src.flow_out), item.out = uniform(src.out);
    item.timestep_scalar = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(src.timestep_scalar)));
#pragma unroll
    for (int i = 0; i < kMaxTerms; ++i) item.term[i] = (const float *)uniform(src.term[i]);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef DRBA_SC16_CLOCKS
  long long clk[20];
#endif
  SC16_CLK(0);
  const size_t P = (size_t)H * W, p_prev = (size_t)hp * wp;
  int tx, ty;
  xcd_strip_tile(vb_, ntiles_, tiles_x, tx, ty);
  const int ox0 = tx * TOW, oy0 = ty * TOH;
  const int X0 = 2 * ox0 - 1, Y0 = 2 * oy0 - 1;  // full-resolution coordinates of window (row 0, column 0)

  // ---- this lane's sample point: window position tid in row-major order
  const bool active = tid < NPOINTS;
  const int pt = min(tid, NPOINTS - 1);
  const int wr = pt / WC, wc = pt - wr * WC;
  const int Xr = X0 + wc, Yr = Y0 + wr;
  const bool inimg = Xr >= 0 && Xr < W && Yr >= 0 && Yr < H;  // outside: the convolution's zero padding
  const int X = min(max(Xr, 0), W - 1), Y = min(max(Yr, 0), H - 1);
  const uint32_t q = (uint32_t)Y * W + X;
  const bool owner = active && wr >= 1 && wc >= 1 && inimg;   // the tile that owns the pixel writes the folded flow

  // ---- prologue loads, all issued before the first wait: group 0's weight fragments, the footprint of the window's sample
  // points in tmp_prev, this point's running flow and timestep
  constexpr int C0 = FOLD ? 0 : 4;  // first channel of tmp_prev that is needed
  const int Xa = max(X0, 0), Ya = max(Y0, 0), Xb = min(X0 + WC - 1, W - 1), Yb = min(Y0 + WR - 1, H - 1);
  const int rx0 = lerp_src(Xa, inv_prev_scale, wp).i0, ry0 = lerp_src(Ya, inv_prev_scale, hp).i0;
  const int rw = lerp_src(Xb, inv_prev_scale, wp).i1 - rx0 + 1, rh = lerp_src(Yb, inv_prev_scale, hp).i1 - ry0 + 1;
  // a group's weight fragments go global -> LDS by LDS-DMA (no registers), 1-KB pieces dealt round robin to the waves.  A wave's
  // pieces are OLDER than the 16 gathers it issues next, so `s_waitcnt vmcnt(16)` -- at most the 16 youngest loads outstanding --
  // covers them however many pieces the wave had
  const __amdgpu_buffer_rsrc_t r_w =
      __builtin_amdgcn_make_buffer_rsrc((void *)wpk, 0, (uint32_t)(KS_TOTAL * 2 * NT * 64 * 16), 0x00020000);
  auto wdma = [&](int g) {
    const int n = ks_count(g) * 2 * NT;  // pieces
#pragma unroll
    for (int i = 0; i < G_::WD; ++i) {
      const int k = i * (THREADS / 64) + wave;
      if (k < n)  // (wave-uniform: a wave without a piece issues nothing)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_ptr)(wl + k * 64), 16, (uint32_t)((ks_first(g) * 2 * NT + k) * 1024 + lane * 16), 0, 0, 0);
    }
  };
  wdma(0);
  const int pr_r = tid / PC, pr_c = tid - pr_r * PC;  // one (row, column) of the footprint per lane
  const bool pr_on = pr_r < rh && pr_c < rw;
  float pv[13];
  if (pr_on) {  // (only the lanes that own a footprint pixel: a wave without one issues no load -- the L1 charges a gather per instruction)
    const cgptr tp = item.tmp_prev + (size_t)(ry0 + pr_r) * wp + rx0 + pr_c;
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
  float bs[NT];  // the epilogue's bias, fetched here: at the end it was a dependent global load of 3k clocks
#pragma unroll
  for (int n = 0; n < NT; ++n) bs[n] = bias ? bias[16 * n + (lane & 15)] : 0.f;
  if (pr_on) {
    float *d = prev + (pr_r * PC + pr_c) * 16;
    if (FOLD) *reinterpret_cast<f32x4 *>(d) = (f32x4){pv[0], pv[1], pv[2], pv[3]};
    d[4] = pv[4];
    *reinterpret_cast<f32x4 *>(d + 8) = (f32x4){pv[5], pv[6], pv[7], pv[8]};
    *reinterpret_cast<f32x4 *>(d + 12) = (f32x4){pv[9], pv[10], pv[11], pv[12]};
  }
  int trx0[kMaxTerms], try0[kMaxTerms];
  if (LAZY) terms_stage<G_::TR, G_::TC, THREADS>(tl, T, item.term, Xa, Ya, Xb, Yb, tid, trx0, try0);
  SC16_CLK(1);
  __syncthreads();
  SC16_CLK(2);

  // taps of the previous head output's upsample at (X, Y), relative to the staged footprint
  const Lerp la = lerp_src(Y, inv_prev_scale, hp), lb = lerp_src(X, inv_prev_scale, wp);
  const int pr0 = (la.i0 - ry0) * PC, pr1 = (la.i1 - ry0) * PC, pc0 = lb.i0 - rx0, pc1 = lb.i1 - rx0;
  // channels 4k .. 4k+3 of the previous head output, upsampled to (X, Y) with row weights (wy0, wy1): four ds_read_b128
  auto prev_up4 = [&](int k, float wy0, float wy1) -> f32x4 {
    const f32x4 q00 = *reinterpret_cast<const f32x4 *>(prev + (pr0 + pc0) * 16 + 4 * k), q01 = *reinterpret_cast<const f32x4 *>(prev + (pr0 + pc1) * 16 + 4 * k);
    const f32x4 q10 = *reinterpret_cast<const f32x4 *>(prev + (pr1 + pc0) * 16 + 4 * k), q11 = *reinterpret_cast<const f32x4 *>(prev + (pr1 + pc1) * 16 + 4 * k);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = lerp2_fma(wy0, wy1, lb.w0, lb.w1, q00[j], q01[j], q10[j], q11[j]);
    return v;
  };
  const f32x4 pu0 = FOLD ? prev_up4(0, la.w0, la.w1) : (f32x4){0.f, 0.f, 0.f, 0.f};
  float fls[4];
  const bool have_terms = LAZY && terms_flow<G_::TR, G_::TC>(tl, T, trx0, try0, X, Y, fls);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (LAZY) {
      const float fd = __fmul_rn(pu0[c], prev_scale);
      fls[c] = have_terms ? __fadd_rn(fls[c], fd) : fd;
    } else if (FOLD) {
      // ifblock_update: flow_in + up(tmp) * scale, product and sum rounded separately as torch evaluates them (and as
      // ifblock_input_lds does: the kernels hand identical flows to warp_blend_fold)
      const float fd = __fmul_rn(pu0[c], prev_scale);
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
    uint32_t o0, o1;  // element offsets of the two tap rows
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
  const uint32_t img_bytes = (uint32_t)((X4 ? 4 : 3) * P * 4), feat_bytes = (uint32_t)(16 * P * 4);
  const __amdgpu_buffer_rsrc_t r_i0 = __builtin_amdgcn_make_buffer_rsrc((void *)(X4 ? item.img0_x4 : item.img0), 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_i1 = __builtin_amdgcn_make_buffer_rsrc((void *)(X4 ? item.img1_x4 : item.img1), 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_f0 = __builtin_amdgcn_make_buffer_rsrc((void *)item.f0_pair, 0, feat_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_f1 = __builtin_amdgcn_make_buffer_rsrc((void *)item.f1_pair, 0, feat_bytes, 0x00020000);
  const uint32_t plane = (uint32_t)(P * 4);

  // ---- the gathers: image taps 8 bytes per row, feature pairs ([C/2, H, W, 2]) 16 bytes per row
  u32x2 ri[X4 ? 1 : 12];  // group 0, planar frames: img0 / img1, 3 channels x 2 tap rows
  u32x4 rx[X4 ? 8 : 1];   // group 0, [H][W][4] frames: img0 / img1, 2 tap rows x 2 columns (a pixel's channels in one 16-byte unit)
  u32x4 rf[16];   // groups 1, 2: 4 pairs of f0, 4 pairs of f1, 2 tap rows each
  // Issue order: the UPPER tap rows of every plane first, the lower rows behind them.  The lower tap row of a point is the
  // upper tap row of the point below it -- the same cache lines, requested by the other half of the wave one instruction
  // earlier: issued back to back, every second gather hit a line whose miss was still in flight, and the L1 holds its whole
  // (in-order) pipeline until that line arrives (TCP_PENDING_STALL_CYCLES: 30 % of the kernel's time, round 5).
  auto issue_img = [&]() {
    if constexpr (X4) {
      rx[0] = __builtin_amdgcn_raw_buffer_load_b128(r_i0, k0.o0 * 16u, 0, 0), rx[1] = __builtin_amdgcn_raw_buffer_load_b128(r_i0, k0.o0 * 16u, 16, 0);
      rx[4] = __builtin_amdgcn_raw_buffer_load_b128(r_i1, k1.o0 * 16u, 0, 0), rx[5] = __builtin_amdgcn_raw_buffer_load_b128(r_i1, k1.o0 * 16u, 16, 0);
      asm volatile("" ::: "memory");
      rx[2] = __builtin_amdgcn_raw_buffer_load_b128(r_i0, k0.o1 * 16u, 0, 0), rx[3] = __builtin_amdgcn_raw_buffer_load_b128(r_i0, k0.o1 * 16u, 16, 0);
      rx[6] = __builtin_amdgcn_raw_buffer_load_b128(r_i1, k1.o1 * 16u, 0, 0), rx[7] = __builtin_amdgcn_raw_buffer_load_b128(r_i1, k1.o1 * 16u, 16, 0);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) ri[2 * c] = __builtin_amdgcn_raw_buffer_load_b64(r_i0, k0.o0 * 4u, c * plane, 0);
#pragma unroll
      for (int c = 0; c < 3; ++c) ri[6 + 2 * c] = __builtin_amdgcn_raw_buffer_load_b64(r_i1, k1.o0 * 4u, c * plane, 0);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int c = 0; c < 3; ++c) ri[2 * c + 1] = __builtin_amdgcn_raw_buffer_load_b64(r_i0, k0.o1 * 4u, c * plane, 0);
#pragma unroll
      for (int c = 0; c < 3; ++c) ri[6 + 2 * c + 1] = __builtin_amdgcn_raw_buffer_load_b64(r_i1, k1.o1 * 4u, c * plane, 0);
    }
  };
  auto issue_f0 = [&](int first_pair) {
#pragma unroll
    for (int c = 0; c < 4; ++c) rf[2 * c] = __builtin_amdgcn_raw_buffer_load_b128(r_f0, k0.o0 * 8u, (first_pair + c) * 2 * plane, 0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) rf[2 * c + 1] = __builtin_amdgcn_raw_buffer_load_b128(r_f0, k0.o1 * 8u, (first_pair + c) * 2 * plane, 0);
  };
  auto issue_f1 = [&](int first_pair) {
#pragma unroll
    for (int c = 0; c < 4; ++c) rf[8 + 2 * c] = __builtin_amdgcn_raw_buffer_load_b128(r_f1, k1.o0 * 8u, (first_pair + c) * 2 * plane, 0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) rf[8 + 2 * c + 1] = __builtin_amdgcn_raw_buffer_load_b128(r_f1, k1.o1 * 8u, (first_pair + c) * 2 * plane, 0);
  };
  const int park = wr * RS + (wc & 1) * PS + (wc >> 1);
  // an octet of this point's values -> one 16-byte unit per plane
  auto park8 = [&](int octet, const float (&v)[8]) {
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t a, b;
      split2(v[2 * i], v[2 * i + 1], a, b);
      h[i] = a, l[i] = b;
    }
    if (active) {
      win[octet * NPOS + park] = h;
      win[(2 + octet) * NPOS + park] = l;
    }
  };
  auto img3 = [&](int which, const TapW &k, float *v) {  // the three channels of img0 (which = 0) / img1 (1) at this point
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float ax, ay, bx, by;
      if constexpr (X4) {
        ax = __uint_as_float(rx[4 * which][c]), ay = __uint_as_float(rx[4 * which + 1][c]);
        bx = __uint_as_float(rx[4 * which + 2][c]), by = __uint_as_float(rx[4 * which + 3][c]);
      } else {
        const u32x2 *r = ri + 6 * which;
        ax = __uint_as_float(r[2 * c].x), ay = __uint_as_float(r[2 * c].y);
        bx = __uint_as_float(r[2 * c + 1].x), by = __uint_as_float(r[2 * c + 1].y);
      }
      v[c] = ax * k.w00 + ay * k.w01 + bx * k.w10 + by * k.w11;
    }
  };
  auto feat8 = [&](const u32x4 *r, const TapW &k, float (&v)[8]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4 a = r[2 * c], b = r[2 * c + 1];
      v[2 * c] = __uint_as_float(a.x) * k.w00 + __uint_as_float(a.z) * k.w01 + __uint_as_float(b.x) * k.w10 + __uint_as_float(b.z) * k.w11;
      v[2 * c + 1] = __uint_as_float(a.y) * k.w00 + __uint_as_float(a.w) * k.w01 + __uint_as_float(b.y) * k.w10 + __uint_as_float(b.w) * k.w11;
    }
  };

  // ---- the matrix phase of one group: waves 0..TOH-1, output row `wave`, pixels lane & 15 (A = the window, rows = pixels;
  // B = the weight fragments, columns = output channels)
  const int m = lane & 15, kq = lane >> 4;
  auto tap_off = [](int t) -> int { return (t / 3) * RS + ((t % 3) & 1) * PS + ((t % 3) >> 1); };
  const int a_row = 2 * wave * RS + m;
  f32x4 hh[NT], lo[NT];  // h h | h l + l h
#pragma unroll
  for (int n = 0; n < NT; ++n) hh[n] = lo[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mma_group = [&](auto G) {
    constexpr int g = decltype(G)::value;
    if (wave < TOH) {
#pragma unroll
      for (int j = 0; j < ks_count(g); ++j) {
        const int tap = min(g < 3 ? 2 * j + (kq >> 1) : 4 * j + kq, 8);  // (tap 9: zero weights, any valid address)
        const u32x4 *a = win + (g < 3 ? (kq & 1) * NPOS : 0) + a_row + tap_off(tap);
        const f16x8 a_h = __builtin_bit_cast(f16x8, a[0]), a_l = __builtin_bit_cast(f16x8, a[2 * NPOS]);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const f16x8 b_h = __builtin_bit_cast(f16x8, wl[((j * 2 + 0) * NT + n) * 64 + lane]);
          const f16x8 b_l = __builtin_bit_cast(f16x8, wl[((j * 2 + 1) * NT + n) * 64 + lane]);
          lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, b_l, lo[n], 0, 0, 0);
          hh[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, b_h, hh[n], 0, 0, 0);
          lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l, b_h, lo[n], 0, 0, 0);
        }
      }
    }
  };

  // ---- group 0: {img0 x3, timestep, img1 x3, mask | flow x4, 0 x4}.  The feature gathers of group 1 fly under it.
  // (the compiler-level fences pin the ISSUE ORDER of the gathers: left alone hipcc hoists all 28 of them to one place and
  // spills what it has just loaded)
  SC16_CLK(3);
  issue_img();
  issue_f0(0);
  asm volatile("" ::: "memory");
  {
    float v[8];
    img3(0, k0, v);
    v[3] = inimg ? tmv : 0.f;
    img3(1, k1, v + 4);
    // (the values as operands: the image registers are dead before the next gathers are issued)
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]) : : "memory");
    issue_f1(0);
    asm volatile("" ::: "memory");
    v[7] = prev_up4(1, uw0, uw1)[0];
    park8(0, v);
    float f[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) f[c] = inimg ? fls[c] : 0.f, f[4 + c] = 0.f;  // interpolate(flow) * 1. / scale at scale 1: the flow itself
    park8(1, f);
  }
  SC16_CLK(4);
  lds_barrier();
  SC16_CLK(5);
  mma_group(std::integral_constant<int, 0>{});
  SC16_CLK(6);
  lds_barrier();
  SC16_CLK(7);
  // ---- groups 1, 2: {f0 | f1}, 8 channels of each.  The group's weights are fetched while its gathers are turned into
  // operands; group 2's gathers are issued as group 1's registers come free and stay in flight across its matrix phase.
  wdma(1);
  {
    float v[8];
    feat8(rf, k0, v);
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
    issue_f0(4);
    asm volatile("" ::: "memory");
    park8(0, v);
    feat8(rf + 8, k1, v);
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
    issue_f1(4);
    asm volatile("" ::: "memory");
    park8(1, v);
  }
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // the weight DMA is older than the 16 gathers just issued
  SC16_CLK(8);
  lds_barrier();
  SC16_CLK(9);
  mma_group(std::integral_constant<int, 1>{});
  SC16_CLK(10);
  lds_barrier();
  SC16_CLK(11);
  wdma(2);
  {
    float v[8];
    feat8(rf, k0, v);
    park8(0, v);
    feat8(rf + 8, k1, v);
    park8(1, v);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SC16_CLK(12);
  lds_barrier();
  SC16_CLK(13);
  mma_group(std::integral_constant<int, 2>{});
  SC16_CLK(14);
  lds_barrier();
  SC16_CLK(15);
  // ---- group 3: feat 0..7 of the previous head output
  wdma(3);
  {
    const f32x4 fa = prev_up4(2, uw0, uw1), fb = prev_up4(3, uw0, uw1);
    const float v[8] = {fa[0], fa[1], fa[2], fa[3], fb[0], fb[1], fb[2], fb[3]};
    park8(0, v);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SC16_CLK(16);
  lds_barrier();
  SC16_CLK(17);
  mma_group(std::integral_constant<int, 3>{});
  SC16_CLK(18);

  // ---- epilogue: join the three products, bias, LeakyReLU(0.2); 4 consecutive pixels of one output channel per lane
  if (wave < TOH) {
    const int oy = oy0 + wave, ox = ox0 + 4 * kq;
    if (oy < Ho && ox < Wo) {
      float nf = 0.f;  // the family's overflow report (common.hpp): an operand past fp16's range has made the sums inf / NaN
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = 16 * n + m;
        const gptr dst = item.out + ((size_t)co * Ho + oy) * Wo + ox;
        f32x4 y = (hh[n] + lo[n] * (1.f / 2048.f)) * kUnscale;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          y[k] = lrelu02(y[k] + bs[n]);
          nf = nf_fold(nf, y[k]);
        }
        if ((Wo & 3) == 0) {
          *(__attribute__((address_space(1))) f32x4 *)dst = y;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (ox + k < Wo) dst[k] = y[k];
        }
      }
      nf_report(status, DRBA_STATUS_STAGE, nf);
    }
  }
#ifdef DRBA_SC16_CLOCKS
  SC16_CLK(19);
  if ((blockIdx.x == 700 || blockIdx.x == 2001 || blockIdx.x == 20001) && lane == 0 && (wave == 0 || wave == 3 || wave == 7))
    printf("wg %d wave %d: prologue %lld sync %lld flow+taps %lld | G0 park %lld barA %lld mma %lld barB %lld | G1 park %lld barA %lld mma %lld barB %lld | "
           "G2 park %lld barA %lld mma %lld barB %lld | G3 park %lld barA %lld mma %lld | epilogue %lld | total %lld\n",
           (int)blockIdx.x, wave, clk[1] - clk[0], clk[2] - clk[1], clk[3] - clk[2], clk[4] - clk[3], clk[5] - clk[4], clk[6] - clk[5], clk[7] - clk[6],
           clk[8] - clk[7], clk[9] - clk[8], clk[10] - clk[9], clk[11] - clk[10], clk[12] - clk[11], clk[13] - clk[12], clk[14] - clk[13],
           clk[15] - clk[14], clk[16] - clk[15], clk[17] - clk[16], clk[18] - clk[17], clk[19] - clk[18], clk[19] - clk[0]);
#endif
#endif
}


// ----------------------------------------------------------------------------------------------------------------------------
// The SCALE-2 stage fused the same way (IFNet_HDv3.py:85-88 at scale 2 -> conv0[0]): the stage input lives at half resolution,
// each of its pixels the mean of the 2 x 2 full-resolution sample points under it (bilinear, align_corners=False, factor 0.5),
// every sample point warped by its own flow.  Unfused, `ifblock_input_lds` writes the 52-channel half-resolution tensor (108 MB
// per 1080p sample, 435 MB per 4K sample) for the stride-2 convolution to read it back: at 4K scale 0.5 -- where this is the LAST
// stage -- the pair is 29 % of a step.  Here a lane owns one window position = FOUR sample points: their flows (terms + the
// previous head output's update, term by term as everywhere), taps and the channels that come from LDS (mask, feat, flow,
// timestep) are formed once, up front, the taps kept compact (two row offsets, the right column's x weight, the lower row's y
// weight per frame: 32 registers for the four points); per channel group the points' gathers are issued and consumed one
// (point, frame) at a time -- 8 x 16 bytes in flight -- and averaged into the group's 16 values, which are then split and
// parked exactly as at scale 1.  The flow is given as terms (the lazy form) and the frames as [H][W][4]; anything else takes
// the unfused pair.  NT = Cout / 16 output tiles per wave (1080p: block 3, 32 channels; 4K scale 0.5: block 4, 16).
template <class G_>
__global__ void __launch_bounds__(G_::THREADS, 4)
stage_conv16_s2(const StageItems items, const FlowTermsArg T, const u32x4 *__restrict__ wpk, const float *__restrict__ bias, int hp, int wp,
                float inv_prev_scale, float prev_scale, int H, int W, int h, int w, int Ho, int Wo, int tiles_x, int n_items,
                unsigned char *status) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TOH = G_::TOH, NT = G_::NT, WR = G_::WR, THREADS = G_::THREADS, NPOS = G_::NPOS, NPOINTS = G_::NPOINTS;
  extern __shared__ __attribute__((aligned(16))) u32x4 lds16[];
  u32x4 *win = lds16 + G_::U_WIN;
  u32x4 *wl = lds16 + G_::U_W;
  float *prev = reinterpret_cast<float *>(lds16 + G_::U_PREV);
  float *tl = reinterpret_cast<float *>(lds16 + G_::U_TERM);
  int vb_, vitem_, ntiles_;
  tile_item_block(n_items, vb_, vitem_, ntiles_);
  typedef __attribute__((address_space(1))) float *gptr;
  typedef __attribute__((address_space(1))) const float *cgptr;
  auto uniform = [](const float *p) -> gptr {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (gptr)(((uint64_t)hi << 32) | lo);
  };
  struct {
    cgptr img0_x4, img1_x4, f0_pair, f1_pair, timestep_map, tmp_prev;
    gptr out;
    float timestep_scalar;
    const float *term[kMaxTerms];
  } item;
  {
    const drba_stage_item_t &src = items.it[vitem_];
    item.img0_x4 = uniform(src.img0_x4), item.img1_x4 = uniform(src.img1_x4), item.f0_pair = uniform(src.f0_pair), item.f1_pair = uniform(src.f1_pair);
    item.timestep_map = uniform(src.timestep_map), item.tmp_prev = uniform(src.tmp_prev), item.out = uniform(src.out);
    item.timestep_scalar = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(src.timestep_scalar)));
#pragma unroll
    for (int i = 0; i < kMaxTerms; ++i) item.term[i] = (const float *)uniform(src.term[i]);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t P = (size_t)H * W, p_prev = (size_t)hp * wp;
  int tx, ty;
  xcd_strip_tile(vb_, ntiles_, tiles_x, tx, ty);
  const int ox0 = tx * TOW, oy0 = ty * TOH;
  const int xs0 = 2 * ox0 - 1, ys0 = 2 * oy0 - 1;  // stage (half-resolution) coordinates of window (row 0, column 0)

  // ---- this lane's window position and its 2 x 2 sample points (2 xs + i, 2 ys + j)
  const bool active = tid < NPOINTS;
  const int pt = min(tid, NPOINTS - 1);
  const int wr = pt / WC, wc = pt - wr * WC;
  const int xr = xs0 + wc, yr = ys0 + wr;
  const bool inimg = xr >= 0 && xr < w && yr >= 0 && yr < h;  // outside: the convolution's zero padding
  const int xs = min(max(xr, 0), w - 1), ys = min(max(yr, 0), h - 1);

  // ---- prologue loads: group 0's weight fragments, tmp_prev's and the terms' footprints under the window's sample points
  const int Xa = 2 * max(xs0, 0), Ya = 2 * max(ys0, 0), Xb = min(2 * min(xs0 + WC - 1, w - 1) + 1, W - 1), Yb = min(2 * min(ys0 + WR - 1, h - 1) + 1, H - 1);
  const int rx0 = lerp_src(Xa, inv_prev_scale, wp).i0, ry0 = lerp_src(Ya, inv_prev_scale, hp).i0;
  const int rw = min(lerp_src(Xb, inv_prev_scale, wp).i1 - rx0 + 1, PC), rh = min(lerp_src(Yb, inv_prev_scale, hp).i1 - ry0 + 1, G_::PR);
  const __amdgpu_buffer_rsrc_t r_w =
      __builtin_amdgcn_make_buffer_rsrc((void *)wpk, 0, (uint32_t)(KS_TOTAL * 2 * NT * 64 * 16), 0x00020000);
  auto wdma = [&](int g) {
    const int n = ks_count(g) * 2 * NT;  // pieces
#pragma unroll
    for (int i = 0; i < G_::WD; ++i) {
      const int k = i * (THREADS / 64) + wave;
      if (k < n)  // (wave-uniform: a wave without a piece issues nothing)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_ptr)(wl + k * 64), 16, (uint32_t)((ks_first(g) * 2 * NT + k) * 1024 + lane * 16), 0, 0, 0);
    }
  };
  wdma(3);  // the groups run in the order 3, 0, 1, 2 here: the one that needs no gather first (its eight values die early)
  const int pr_r = tid / PC, pr_c = tid - pr_r * PC;
  const bool pr_on = pr_r < rh && pr_c < rw;
  float pv[13];
  if (pr_on) {
    const cgptr tp = item.tmp_prev + (size_t)(ry0 + pr_r) * wp + rx0 + pr_c;
#pragma unroll
    for (int c = 0; c < 13; ++c) pv[c] = tp[(size_t)c * p_prev];
  }
  float bs[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) bs[n] = bias ? bias[16 * n + (lane & 15)] : 0.f;
  // the timestep at the four sample points (two adjacent pixels per row)
  float tmv[4];
  if (item.timestep_map) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x2 t2 = *reinterpret_cast<const __attribute__((address_space(1))) f32x2 *>(item.timestep_map + (size_t)(2 * ys + j) * W + 2 * xs);
      tmv[2 * j] = t2.x, tmv[2 * j + 1] = t2.y;
    }
  } else {
    tmv[0] = tmv[1] = tmv[2] = tmv[3] = item.timestep_scalar;
  }
  if (pr_on) {
    float *d = prev + (pr_r * PC + pr_c) * 16;
    *reinterpret_cast<f32x4 *>(d) = (f32x4){pv[0], pv[1], pv[2], pv[3]};
    d[4] = pv[4];
    *reinterpret_cast<f32x4 *>(d + 8) = (f32x4){pv[5], pv[6], pv[7], pv[8]};
    *reinterpret_cast<f32x4 *>(d + 12) = (f32x4){pv[9], pv[10], pv[11], pv[12]};
  }
  int trx0[kMaxTerms], try0[kMaxTerms];
  terms_stage<G_::TR, G_::TC, THREADS>(tl, T, item.term, Xa, Ya, Xb, Yb, tid, trx0, try0);
  __syncthreads();

  // ---- the four sample points: flow, compact taps, and the channels that come from LDS (averaged: 0.25 each, the factor-0.5
  // bilinear downsample of a 2 x 2 block)
  uint32_t to0[4][2], to1[4][2];  // [point][frame]: element offsets of the two tap rows (pair of a row loaded at min(x0, W - 2))
  float twx[4][2], twy[4][2];     // weight of the RIGHT loaded pixel (the right-border case folded in), weight of the LOWER row
  float a_ts = 0.f, a_mask = 0.f, a_flow[4] = {0.f, 0.f, 0.f, 0.f}, a_feat[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sp = 0; sp < 4; ++sp) {
    const int X = 2 * xs + (sp & 1), Y = 2 * ys + (sp >> 1);
    const Lerp la = lerp_src(Y, inv_prev_scale, hp), lb = lerp_src(X, inv_prev_scale, wp);
    const int pr0 = min(la.i0 - ry0, G_::PR - 1) * PC, pr1 = min(la.i1 - ry0, G_::PR - 1) * PC, pc0 = min(lb.i0 - rx0, PC - 1), pc1 = min(lb.i1 - rx0, PC - 1);
    auto prev_up4 = [&](int k) -> f32x4 {
      const f32x4 q00 = *reinterpret_cast<const f32x4 *>(prev + (pr0 + pc0) * 16 + 4 * k), q01 = *reinterpret_cast<const f32x4 *>(prev + (pr0 + pc1) * 16 + 4 * k);
      const f32x4 q10 = *reinterpret_cast<const f32x4 *>(prev + (pr1 + pc0) * 16 + 4 * k), q11 = *reinterpret_cast<const f32x4 *>(prev + (pr1 + pc1) * 16 + 4 * k);
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = lerp2_fma(la.w0, la.w1, lb.w0, lb.w1, q00[j], q01[j], q10[j], q11[j]);
      return v;
    };
    const f32x4 pu0 = prev_up4(0);
    float fls[4];
    const bool have_terms = terms_flow<G_::TR, G_::TC>(tl, T, trx0, try0, X, Y, fls);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float fd = __fmul_rn(pu0[c], prev_scale);
      fls[c] = have_terms ? __fadd_rn(fls[c], fd) : fd;
      a_flow[c] += 0.25f * fls[c];
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      // taps_border's arithmetic, keeping the fractions instead of the four products: clip to [0, size - 1], floor, x1 / y1 clamped
      float x = warp_coord(X, W, fls[2 * f]), y = warp_coord(Y, H, fls[2 * f + 1]);
      x = fminf(fmaxf(x, 0.f), (float)(W - 1)), y = fminf(fmaxf(y, 0.f), (float)(H - 1));
      const float fx = floorf(x), fy = floorf(y);
      const float wx1 = x - fx, wy1 = y - fy;
      const int x0 = min(max((int)fx, 0), W - 1), y0 = min(max((int)fy, 0), H - 1), y1 = min(y0 + 1, H - 1);
      const int xb = min(x0, W - 2);  // the pair of a row is loaded at min(x0, W - 2); at the right border (wx1 == 0) its RIGHT pixel is x0
      to0[sp][f] = (uint32_t)(y0 * W + xb), to1[sp][f] = (uint32_t)(y1 * W + xb);
      twx[sp][f] = x0 != xb ? 1.f - wx1 : wx1;
      twy[sp][f] = wy1;
    }
    const f32x4 p1 = prev_up4(1), p2 = prev_up4(2), p3 = prev_up4(3);
    a_mask += 0.25f * p1[0];
#pragma unroll
    for (int c = 0; c < 4; ++c) a_feat[c] += 0.25f * p2[c], a_feat[4 + c] += 0.25f * p3[c];
    a_ts += 0.25f * tmv[sp];
    // (one sample point at a time: the four are independent and hipcc would otherwise interleave them -- 4 x the live registers)
    asm volatile("" : "+v"(a_ts), "+v"(a_mask), "+v"(a_flow[0]), "+v"(a_flow[1]), "+v"(a_flow[2]), "+v"(a_flow[3]), "+v"(a_feat[0]), "+v"(a_feat[1]),
                 "+v"(a_feat[2]), "+v"(a_feat[3]), "+v"(a_feat[4]), "+v"(a_feat[5]), "+v"(a_feat[6]), "+v"(a_feat[7]), "+v"(to0[sp][0]), "+v"(to0[sp][1]),
                 "+v"(to1[sp][0]), "+v"(to1[sp][1]), "+v"(twx[sp][0]), "+v"(twx[sp][1]), "+v"(twy[sp][0]), "+v"(twy[sp][1]) : : "memory");
  }
  const uint32_t img_bytes = (uint32_t)(4 * P * 4), feat_bytes = (uint32_t)(16 * P * 4);
  const __amdgpu_buffer_rsrc_t r_i0 = __builtin_amdgcn_make_buffer_rsrc((void *)item.img0_x4, 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_i1 = __builtin_amdgcn_make_buffer_rsrc((void *)item.img1_x4, 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_f0 = __builtin_amdgcn_make_buffer_rsrc((void *)item.f0_pair, 0, feat_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_f1 = __builtin_amdgcn_make_buffer_rsrc((void *)item.f1_pair, 0, feat_bytes, 0x00020000);
  const uint32_t plane = (uint32_t)(P * 4);

  const int park = wr * RS + (wc & 1) * PS + (wc >> 1);
  auto park8 = [&](int octet, const float (&v)[8]) {
    u32x4 hh_, ll_;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t a, b;
      split2(inimg ? v[2 * i] : 0.f, inimg ? v[2 * i + 1] : 0.f, a, b);
      hh_[i] = a, ll_[i] = b;
    }
    if (active) {
      win[octet * NPOS + park] = hh_;
      win[(2 + octet) * NPOS + park] = ll_;
    }
  };
  // (the tap weights of (point, frame) are formed at the gathers from the two fractions: wxl = 1 - wxr, wy0 = 1 - wy1, the four
  // products as taps_border forms them, times the 1 / 4 of the downsample)

  // ---- the matrix phase of one group (as at scale 1)
  const int m = lane & 15, kq = lane >> 4;
  auto tap_off = [](int t) -> int { return (t / 3) * RS + ((t % 3) & 1) * PS + ((t % 3) >> 1); };
  const int a_row = 2 * wave * RS + m;
  f32x4 hh[NT], lo[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) hh[n] = lo[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mma_group = [&](auto G) {
    constexpr int g = decltype(G)::value;
    if (wave < TOH) {
#pragma unroll
      for (int j = 0; j < ks_count(g); ++j) {
        const int tap = min(g < 3 ? 2 * j + (kq >> 1) : 4 * j + kq, 8);
        const u32x4 *a = win + (g < 3 ? (kq & 1) * NPOS : 0) + a_row + tap_off(tap);
        const f16x8 a_h = __builtin_bit_cast(f16x8, a[0]), a_l = __builtin_bit_cast(f16x8, a[2 * NPOS]);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const f16x8 b_h = __builtin_bit_cast(f16x8, wl[((j * 2 + 0) * NT + n) * 64 + lane]);
          const f16x8 b_l = __builtin_bit_cast(f16x8, wl[((j * 2 + 1) * NT + n) * 64 + lane]);
          lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, b_l, lo[n], 0, 0, 0);
          hh[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, b_h, hh[n], 0, 0, 0);
          lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l, b_h, lo[n], 0, 0, 0);
        }
      }
    }
  };
  auto group_barriers = [&](auto G, bool last) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the group's weight DMA has landed (every gather is consumed)
    lds_barrier();
    mma_group(G);
    if (!last) lds_barrier();
  };
  // ---- group 3 first: feat 0..7 of the previous head output (no gathers)
  park8(0, a_feat);
  group_barriers(std::integral_constant<int, 3>{}, false);
  wdma(0);

  // ---- group 0: {img0 x3, timestep, img1 x3, mask | flow x4, 0 x4}.  The gathers go in HALF STEPS -- one tap row of one frame
  // at one sample point: two pixels of 16 bytes -- double-buffered: half step k + 1 is issued before half step k is consumed
  // (16 half steps: frame, point, row)
  {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    u32x4 rb[2][2];
    auto issue = [&](int hs, u32x4 (&r)[2]) {
      const int f = hs >> 3, sp = (hs >> 1) & 3, row = hs & 1;
      const __amdgpu_buffer_rsrc_t &rs = f == 0 ? r_i0 : r_i1;
      const uint32_t o = (row ? to1[sp][f] : to0[sp][f]) * 16u;
      r[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0), r[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 16, 0);
    };
    issue(0, rb[0]);
#pragma unroll
    for (int hs = 0; hs < 16; ++hs) {
      if (hs + 1 < 16) issue(hs + 1, rb[(hs + 1) & 1]);
      const int f = hs >> 3, sp = (hs >> 1) & 3, row = hs & 1;
      const float wxr = twx[sp][f], wxl = 1.f - wxr, wy = row ? twy[sp][f] : 1.f - twy[sp][f];
      const float wl_ = 0.25f * (wxl * wy), wr_ = 0.25f * (wxr * wy);
      const u32x4 (&r)[2] = rb[hs & 1];
#pragma unroll
      for (int c = 0; c < 3; ++c) v[4 * f + c] += __uint_as_float(r[0][c]) * wl_ + __uint_as_float(r[1][c]) * wr_;
      // (the sums as operands: a half step's registers are free before the one after the next is issued)
      asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]) : : "memory");
    }
    v[3] = a_ts, v[7] = a_mask;
    park8(0, v);
    // interpolate(flow, 1 / scale) * 1. / scale (IFNet_HDv3.py:87): the mean of the four sample points' flows, over the scale
    const float fl[8] = {(a_flow[0] * 1.f) / 2.f, (a_flow[1] * 1.f) / 2.f, (a_flow[2] * 1.f) / 2.f, (a_flow[3] * 1.f) / 2.f, 0.f, 0.f, 0.f, 0.f};
    park8(1, fl);
  }
  group_barriers(std::integral_constant<int, 0>{}, false);
  // ---- groups 1, 2: {f0 | f1}, 8 channels (4 pairs) of each; half step = the four pairs of one tap row of one frame at one point
#pragma unroll
  for (int g = 1; g <= 2; ++g) {
    wdma(g);
    u32x4 rb[2][4];
    auto issue = [&](int hs, u32x4 (&r)[4]) {
      const int f = hs >> 3, sp = (hs >> 1) & 3, row = hs & 1;
      const __amdgpu_buffer_rsrc_t &rs = f == 0 ? r_f0 : r_f1;
      const uint32_t o = (row ? to1[sp][f] : to0[sp][f]) * 8u;
#pragma unroll
      for (int c = 0; c < 4; ++c) r[c] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, (4 * (g - 1) + c) * 2 * plane, 0);
    };
    issue(0, rb[0]);
    float v[8];
#pragma unroll
    for (int hs = 0; hs < 16; ++hs) {
      if (hs + 1 < 16) issue(hs + 1, rb[(hs + 1) & 1]);
      const int f = hs >> 3, sp = (hs >> 1) & 3, row = hs & 1;
      if ((hs & 7) == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = 0.f;
      }
      const float wxr = twx[sp][f], wxl = 1.f - wxr, wy = row ? twy[sp][f] : 1.f - twy[sp][f];
      const float wl_ = 0.25f * (wxl * wy), wr_ = 0.25f * (wxr * wy);
      const u32x4 (&r)[4] = rb[hs & 1];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v[2 * c] += __uint_as_float(r[c].x) * wl_ + __uint_as_float(r[c].z) * wr_;
        v[2 * c + 1] += __uint_as_float(r[c].y) * wl_ + __uint_as_float(r[c].w) * wr_;
      }
      asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
      if ((hs & 7) == 7) park8(f, v);
    }
    if (g == 1) group_barriers(std::integral_constant<int, 1>{}, false);
    else group_barriers(std::integral_constant<int, 2>{}, true);
  }

  // ---- epilogue
  if (wave < TOH) {
    const int oy = oy0 + wave, ox = ox0 + 4 * kq;
    if (oy < Ho && ox < Wo) {
      float nf = 0.f;  // the family's overflow report (common.hpp): an operand past fp16's range has made the sums inf / NaN
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = 16 * n + m;
        const gptr dst = item.out + ((size_t)co * Ho + oy) * Wo + ox;
        f32x4 y = (hh[n] + lo[n] * (1.f / 2048.f)) * kUnscale;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          y[k] = lrelu02(y[k] + bs[n]);
          nf = nf_fold(nf, y[k]);
        }
        if ((Wo & 3) == 0) {
          *(__attribute__((address_space(1))) f32x4 *)dst = y;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (ox + k < Wo) dst[k] = y[k];
        }
      }
      nf_report(status, DRBA_STATUS_STAGE, nf);
    }
  }
#endif
}

}  // namespace drba_stage_conv16

extern "C" {

size_t drba_stage_conv16_packed_floats(int Cout) {
  using namespace drba_stage_conv16;
  return (Cout == 16 || Cout == 32) ? (size_t)KS_TOTAL * 2 * (Cout / 16) * 64 * 4 : 0;
}

/* HOST: w [Cout, 52, 3, 3] -> [K step 18][plane h, l][n tile][lane][8 x fp16]; lane = (cout = 16 n + (l & 15), kq = l >> 4), element i =
 * the weight of (unit_tap, chan_of(group, unit_octet, i)); every weight as two fp16 terms (split_weight_terms) */
int drba_stage_conv16_pack(const float *w, int Cout, float *packed) {
  using namespace drba_stage_conv16;
  if (!w || !packed || (Cout != 16 && Cout != 32)) return DRBA_EINVAL;
  const int NT = Cout / 16;
  if (!two_term_weights_ok(w, (size_t)Cout * CIN * 9)) return DRBA_EUNSUPPORTED;
  memset(packed, 0, sizeof(float) * drba_stage_conv16_packed_floats(Cout));
  unsigned short *dst = reinterpret_cast<unsigned short *>(packed);
  for (int g = 0; g < NG; ++g)
    for (int j = 0; j < ks_count(g); ++j)
      for (int n = 0; n < NT; ++n)
        for (int l = 0; l < 64; ++l)
          for (int i = 0; i < 8; ++i) {
            const int kq = l >> 4, co = 16 * n + (l & 15), tap = unit_tap(g, j, kq), ci = chan_of(g, unit_octet(g, kq), i);
            if (tap >= 9 || ci < 0) continue;
            unsigned short t[3];
            split_weight_terms(w[((size_t)co * CIN + ci) * 9 + tap], 2, t);
            const size_t ks = (size_t)ks_first(g) + j;
            dst[((((ks * 2 + 0) * NT + n) * 64) + l) * 8 + i] = t[0];
            dst[((((ks * 2 + 1) * NT + n) * 64) + l) * 8 + i] = t[1];
          }
  return DRBA_OK;
}

int drba_stage_conv16_supported(int H, int W, float scale, float prev_scale, int Cout) {
  if (scale == 2.f) return (H >= 4 && W >= 4 && prev_scale == 4.f && (Cout == 16 || Cout == 32)) ? 1 : 0;  // (lazy flow + [H][W][4] frames only)
  return (H >= 2 && W >= 2 && scale == 1.f && prev_scale == 2.f && Cout == 16) ? 1 : 0;
}

int drba_stage_conv16_batch(const drba_stage_item_t *items, int n_items, const drba_flow_terms_t *terms, int hp, int wp, float prev_scale,
                            int H, int W, float scale, int Cout, const float *packed_w, const float *bias, void *stream) {
  using namespace drba_stage_conv16;
  if (!items || n_items <= 0 || n_items > DRBA_MAX_STAGE_ITEMS || !packed_w || H < 2 || W < 2 || hp <= 0 || wp <= 0) return DRBA_EINVAL;
  if (((uintptr_t)packed_w & 15) != 0) return DRBA_EINVAL;
  if (!drba_stage_conv16_supported(H, W, scale, prev_scale, Cout)) return DRBA_EUNSUPPORTED;
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
    if ((I.img0_x4 != nullptr) != (items[0].img0_x4 != nullptr) || (I.img0_x4 != nullptr) != (I.img1_x4 != nullptr)) return DRBA_EINVAL;
    if ((((uintptr_t)I.img0_x4 | (uintptr_t)I.img1_x4) & 15) != 0) return DRBA_EINVAL;
    its.it[k] = I;
  }
  const bool x4 = items[0].img0_x4 != nullptr;
  hipStream_t s = (hipStream_t)stream;
  if (scale == 2.f) {
    // the scale-2 stage: sample points at full resolution, the stage input at half, the convolution's output at a quarter
    if (!lazy || !x4) return DRBA_EUNSUPPORTED;
    for (int i = 0; i < T.n; ++i)
      if (T.scale[i] < 8.f) return DRBA_EUNSUPPORTED;
    const int h = H / 2, w = W / 2, Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
    const int tiles_x = (Wo + TOW - 1) / TOW, tiles_y = (Ho + 7 - 1) / 7;
#define DRBA_SC16_S2(NT_)                                                                                                       \
    do {                                                                                                                          \
      using G_ = Geo<7, NT_>;                                                                                                     \
      if (max_dynamic_lds((const void *)stage_conv16_s2<G_>, 160 * 1024) != hipSuccess) return DRBA_ELAUNCH;                       \
      DRBA_LAUNCH((stage_conv16_s2<G_>), dim3(tiles_x * tiles_y * n_items), dim3(G_::THREADS), (size_t)G_::LDS_UNITS * 16, s, its, T, \
                  reinterpret_cast<const drba_stage_conv16::u32x4 *>(packed_w), bias, hp, wp, 0.25f, prev_scale, H, W, h, w, Ho, Wo, \
                  tiles_x, n_items, status_bytes());                                                                              \
    } while (0)
    if (Cout == 16) DRBA_SC16_S2(1);
    else DRBA_SC16_S2(2);
#undef DRBA_SC16_S2
    DRBA_CHECK_LAUNCH();
    for (int k = 0; k < n_items && g_range_check; ++k) {
      const int rc = range_scan(items[k].out, (size_t)Cout * Ho * Wo, stream);
      if (rc != DRBA_OK) return rc;
    }
    return DRBA_OK;
  }
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
#define DRBA_SC16_GO_(FO, NT_, TOH_, XX)                                                                                       \
  do {                                                                                                                     \
    using G_ = Geo<TOH_, NT_>;                                                                                             \
    const size_t lds_bytes = (size_t)G_::LDS_UNITS * 16;                                                                   \
    if (max_dynamic_lds((const void *)stage_conv16<FO, G_, XX>, 160 * 1024) != hipSuccess) return DRBA_ELAUNCH;            \
    const int tiles_x = (Wo + TOW - 1) / TOW, tiles_y = (Ho + G_::TOH - 1) / G_::TOH;                                   \
    DRBA_LAUNCH((stage_conv16<FO, G_, XX>), dim3(tiles_x * tiles_y * n_items), dim3(G_::THREADS), lds_bytes, s, its, T,        \
                reinterpret_cast<const drba_stage_conv16::u32x4 *>(packed_w), bias, hp, wp, 0.5f, prev_scale, H, W, Ho, Wo, tiles_x,   \
                n_items, status_bytes());                                                                                  \
  } while (0)
#define DRBA_SC16_GO(FO, NT_, TOH_)          \
  do {                                       \
    if (x4) DRBA_SC16_GO_(FO, NT_, TOH_, true); \
    else DRBA_SC16_GO_(FO, NT_, TOH_, false);   \
  } while (0)
#define DRBA_SC16_GO2(NT_, TOH_)            \
  do {                                      \
    if (lazy) DRBA_SC16_GO(2, NT_, TOH_);   \
    else if (fold) DRBA_SC16_GO(1, NT_, TOH_); \
    else DRBA_SC16_GO(0, NT_, TOH_);        \
  } while (0)
  // output rows per workgroup: 7 (8 waves, two workgroups per CU).  TUNING builds: DRBA_SC16_TOH = 3 / 5 -> 4 / 6 waves; measured
  // on the 8-sample 1080p launch (profiles/r05_stage_conv16_variants.txt): 192 / 195 us per sample against 178
  if (Cout != 16) return DRBA_EUNSUPPORTED;
#ifdef DRBA_TUNING_SWITCHES
  static const int toh = env_int("DRBA_SC16_TOH", 7);
  if (toh == 3) DRBA_SC16_GO2(1, 3);
  else if (toh == 5) DRBA_SC16_GO2(1, 5);
  else
#endif
    DRBA_SC16_GO2(1, 7);
#undef DRBA_SC16_GO2
#undef DRBA_SC16_GO
#undef DRBA_SC16_GO_
  DRBA_CHECK_LAUNCH();
  for (int k = 0; k < n_items && g_range_check; ++k) {
    const int rc = range_scan(items[k].out, (size_t)Cout * Ho * Wo, stream);
    if (rc != DRBA_OK) return rc;
  }
  return DRBA_OK;
}

}  // extern "C"
