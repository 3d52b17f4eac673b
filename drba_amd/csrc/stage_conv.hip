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
// TOH output rows: waves 0..TOH-1 own the 2 TOH x 32 block of sample points (and run output row `wave` on the matrix
// cores), wave TOH the upper row + left column.  WRES: all weights resident in LDS (30 KB) instead of streamed per group.
template <int TOH_, bool WRES_>
struct Geo {
  static constexpr int TOH = TOH_;
  static constexpr bool WRES = WRES_;
  static constexpr int WR = 2 * TOH + 1;         // window rows
  static constexpr int CS = WR * RS + 1;         // channel stride: odd, so the 4 channels of an A fragment fall on distinct banks
  static constexpr int THREADS = 64 * (TOH + 1);
  static constexpr int PR = TOH + 3;             // tmp_prev footprint capacity, rows
  static constexpr int WL = WRES ? W_FLOATS : 2 * WG_FLOATS;
  static constexpr int WV = (WG_FLOATS + THREADS - 1) / THREADS;  // streamed weights: floats per lane and group
  static constexpr int TR = TOH / 2 + 4, TC = 12;  // term footprint capacity: (2 TOH + 1 rows, 33 columns) at >= 1/4 resolution
  static constexpr int TERM_FLOATS = kMaxTerms * 4 * TR * TC;
  static constexpr int LDS_FLOATS = WL + 2 * 4 * CS + 16 * PR * PC + TERM_FLOATS;
  static_assert(32 + WR <= 64, "upper row + left column on one wave");
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

struct Raw {  // the loads of one channel group, in flight across the previous group's barrier
  u32x4 q[4];
  u32x2 h[6];
};

__device__ __forceinline__ void lds_barrier() {
  // LDS writes of this wave done, then the workgroup barrier; global loads stay in flight (no vmcnt wait)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

#ifndef DRBA_SC_DEPTH
#define DRBA_SC_DEPTH 1
#endif
// FMODE: 0 = the finished flow is read; 1 = FOLD (flow_prev + the previous stage's update, written to flow_out); 2 = LAZY
// (the flow is the sum of the terms, flow_terms.hpp, + the previous stage's update; nothing but the convolution is written)
#ifdef DRBA_SC_WPE  // experiment builds: cap the registers for this many waves per SIMD
#define DRBA_SC_ATTR __attribute__((amdgpu_waves_per_eu(DRBA_SC_WPE)))
#else
#define DRBA_SC_ATTR
#endif
template <int FMODE, class G_>
__global__ void __launch_bounds__(G_::THREADS) DRBA_SC_ATTR
stage_conv0(const StageItems items, const FlowTermsArg T, const float *__restrict__ wpk, const float *__restrict__ bias, int hp, int wp,
            float inv_prev_scale, float prev_scale, int H, int W, int Ho, int Wo, int tiles_x, int n_items) {
  constexpr bool FOLD = FMODE != 0, WRITES = FMODE == 1, LAZY = FMODE == 2;
  constexpr int TOH = G_::TOH, WR = G_::WR, CS = G_::CS, THREADS = G_::THREADS, PR = G_::PR;
  constexpr bool WRES = G_::WRES;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *wl = lds;                         // WRES: [13][9][64]; else [2][9][64]
  float *win = lds + G_::WL;               // [2][4][CS]
  float *prev = win + 2 * 4 * CS;          // [PR][PC][16]: a footprint pixel's 13 channels as four 16-byte words (flow | mask | feat 0..3 | feat 4..7)
  float *tl = prev + 16 * PR * PC;         // [kMaxTerms][TR * TC][4]
  int vb_, vitem_, ntiles_;
  tile_item_block(n_items, vb_, vitem_, ntiles_);  // one grid dimension: the items of a tile back to back on one XCD (common.hpp)
  // the item is picked by the block id out of the by-value argument: the compiler does not see that its fields are
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
    const drba_stage_item_t &src = items.it[vitem_];
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
  xcd_strip_tile(vb_, ntiles_, tiles_x, tx, ty);
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

  // ---- prologue loads, all issued before the first wait: the resident weights, the footprint of the window's sample
  // points in tmp_prev, this point's running flow and timestep (one memory latency instead of four in a row)
  constexpr int C0 = FOLD ? 0 : 4;  // first channel of tmp_prev that is needed
  const int Xa = max(X0, 0), Ya = max(Y0, 0), Xb = min(X0 + WC - 1, W - 1), Yb = min(Y0 + WR - 1, H - 1);
  const int rx0 = lerp_src(Xa, inv_prev_scale, wp).i0, ry0 = lerp_src(Ya, inv_prev_scale, hp).i0;
  const int rw = lerp_src(Xb, inv_prev_scale, wp).i1 - rx0 + 1, rh = lerp_src(Yb, inv_prev_scale, hp).i1 - ry0 + 1;
  constexpr int WV4 = WRES ? (W_FLOATS / 4 + THREADS - 1) / THREADS : 1;
  f32x4 wv[WV4];
  float ws[2][G_::WV];  // streamed weights: this lane's floats of the group in flight
  auto wload = [&](int g, float (&w)[G_::WV]) {
#pragma unroll
    for (int i = 0; i < G_::WV; ++i) w[i] = wpk[g * WG_FLOATS + min(tid + i * THREADS, WG_FLOATS - 1)];
  };
  auto wpark = [&](int g, const float (&w)[G_::WV]) {
#pragma unroll
    for (int i = 0; i < G_::WV; ++i)
      if (tid + i * THREADS < WG_FLOATS) wl[(g & 1) * WG_FLOATS + tid + i * THREADS] = w[i];
  };
  if constexpr (WRES) {
#pragma unroll
    for (int i = 0; i < WV4; ++i) wv[i] = reinterpret_cast<const f32x4 *>(wpk)[min(tid + i * THREADS, W_FLOATS / 4 - 1)];
  } else {
    wload(0, ws[0]);
  }
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
  if constexpr (WRES) {
#pragma unroll
    for (int i = 0; i < WV4; ++i)
      if (tid + i * THREADS < W_FLOATS / 4) reinterpret_cast<f32x4 *>(wl)[tid + i * THREADS] = wv[i];
  }
  if (pr_on) {
#pragma unroll
    for (int c = C0; c < 13; ++c) prev[(pr_r * PC + pr_c) * 16 + (c < 5 ? c : c + 3)] = pv[c];  // words: flow | mask, - | feat 0..3 | feat 4..7
  }
  int trx0[kMaxTerms], try0[kMaxTerms];
  if (LAZY) terms_stage<G_::TR, G_::TC, THREADS>(tl, T, item.term, Xa, Ya, Xb, Yb, tid, trx0, try0);
  __syncthreads();

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
  auto prev_up = [&](int c) -> float { return pu0[c]; };  // c < 4: the flow delta
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
  auto prev_up_z4 = [&](int k) -> f32x4 { return prev_up4(k, uw0, uw1); };  // mask / feat (channels 4..12), zero outside the image
  const uint32_t img_bytes = (uint32_t)(3 * P * 4), feat_bytes = (uint32_t)(16 * P * 4);
  const __amdgpu_buffer_rsrc_t r_i0 = __builtin_amdgcn_make_buffer_rsrc((void *)item.img0, 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_i1 = __builtin_amdgcn_make_buffer_rsrc((void *)item.img1, 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_f0 = __builtin_amdgcn_make_buffer_rsrc((void *)item.f0_pair, 0, feat_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_f1 = __builtin_amdgcn_make_buffer_rsrc((void *)item.f1_pair, 0, feat_bytes, 0x00020000);
  const uint32_t plane = (uint32_t)(P * 4);

  auto issue = [&](auto G, Raw &r) {
    constexpr int g = decltype(G)::value;
#ifdef DRBA_SC_EXP_NOLOAD  // timing experiment only (wrong results)
    for (int i = 0; i < 4; ++i) r.q[i] = (u32x4){q, q, q, q};
    for (int i = 0; i < 6; ++i) r.h[i] = (u32x2){q, q};
    return;
#endif
    if constexpr (g < 2) {
      const __amdgpu_buffer_rsrc_t &rs = g == 0 ? r_i0 : r_i1;
      const TapW &k = g == 0 ? k0 : k1;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        r.h[2 * c] = __builtin_amdgcn_raw_buffer_load_b64(rs, k.o0 * 4u, c * plane, 0);
        r.h[2 * c + 1] = __builtin_amdgcn_raw_buffer_load_b64(rs, k.o1 * 4u, c * plane, 0);
      }
    } else if constexpr (g < 10) {
      constexpr int c2 = g - 2;  // [C/2, H, W, 2]: the pair's plane is 2P floats
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
      v[3] = g == 0 ? (inimg ? tmv : 0.f) : prev_up_z4(1)[0];
    } else if constexpr (g < 10) {
      auto pair = [&](const u32x4 &a, const u32x4 &b, const TapW &k, float &v0, float &v1) {
        v0 = __uint_as_float(a.x) * k.w00 + __uint_as_float(a.z) * k.w01 + __uint_as_float(b.x) * k.w10 + __uint_as_float(b.z) * k.w11;
        v1 = __uint_as_float(a.y) * k.w00 + __uint_as_float(a.w) * k.w01 + __uint_as_float(b.y) * k.w10 + __uint_as_float(b.w) * k.w11;
      };
      pair(r.q[0], r.q[1], k0, v[0], v[1]);
      pair(r.q[2], r.q[3], k1, v[2], v[3]);
    } else if constexpr (g < 12) {
      {
        const f32x4 ft = prev_up_z4(g - 8);  // word 2: feat 0..3, word 3: feat 4..7
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = ft[c];
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = inimg ? fls[c] : 0.f;  // interpolate(flow) * 1. / scale at scale 1: the flow itself
    }
  };

  // ---- 13 groups: park, barrier, 9 taps on the matrix cores.  Group g + 1's loads are issued before group g's barrier.
  // (Measured and not kept, tools/exp/stage_conv_variants.sh: loads two or three groups ahead, 196 -> 220 / 229 us per
  // 1080p sample; the same wave forming group g + 1's values between the MFMAs of group g, 213 us; weights streamed per
  // group instead of resident, 198-234 us; 5- and 7-wave workgroups, 196-201 us.  PMC: the texture addresser is the
  // busiest unit, 63 % -- 29 cycles per vector-memory instruction, 71 of them per wave; MFMA 29 %, VALU 25 %, LDS 22 %.)
  const int park = wr * RS + wc;
  const int a_off = (lane >> 4) * CS + (2 * wave) * RS + 2 * (lane & 15);  // waves 0..TOH-1: output row `wave`, pixels lane & 15
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int DEPTH = DRBA_SC_DEPTH;  // channel groups whose loads are in flight beyond the one being parked
  Raw raw[DEPTH + 1];
  static_for<DEPTH>([&](auto G) { issue(G, raw[decltype(G)::value]); });
  static_for<NG>([&](auto G) {
    constexpr int g = decltype(G)::value;
    if constexpr (g + DEPTH < 10) issue(std::integral_constant<int, g + DEPTH>{}, raw[(g + DEPTH) % (DEPTH + 1)]);
    if constexpr (!WRES && g + 1 < NG) wload(g + 1, ws[(g + 1) & 1]);
    float v[4];
    finish(G, raw[g % (DEPTH + 1)], v);
    float *wb = win + (g & 1) * 4 * CS;
    if (active) {
#pragma unroll
      for (int c = 0; c < 4; ++c) wb[c * CS + park] = v[c];
    }
    if constexpr (!WRES) wpark(g, ws[g & 1]);
    lds_barrier();
    if (wave < TOH) {
      const float *ab = wb + a_off;
      const float *bb = (WRES ? wl + g * WG_FLOATS : wl + (g & 1) * WG_FLOATS) + lane;
      float a[9], b[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) a[t] = ab[(t / 3) * RS + (t % 3)], b[t] = bb[t * 64];
#ifdef DRBA_SC_EXP_NOMFMA  // timing experiment only (wrong results)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t & 3] += a[t] * b[t];
#else
#pragma unroll
      for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc, 0, 0, 0);
#endif
    }
  });

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
  // workgroup shape (TUNING builds: DRBA_SC_TOH = 4 / 6 / 8 output rows, DRBA_SC_WRES = 1 keeps all weights in LDS)
  static const int toh = env_int("DRBA_SC_TOH", 8), wres = env_int("DRBA_SC_WRES", 1);
#define DRBA_SC_GO(FO, TT, R)                                                                                            \
  do {                                                                                                                     \
    using G_ = Geo<TT, R>;                                                                                                 \
    static const int lds_pad = env_int("DRBA_SC_LDS_PAD", 0); /* TUNING builds: extra LDS bytes per workgroup (occupancy probe) */ \
    const size_t lds_bytes = (size_t)G_::LDS_FLOATS * 4 + (size_t)lds_pad;                                                 \
    if (max_dynamic_lds((const void *)stage_conv0<FO, G_>, 160 * 1024) != hipSuccess) return DRBA_ELAUNCH;                 \
    const int tiles_x = (Wo + TOW - 1) / TOW, tiles_y = (Ho + TT - 1) / TT;                                                \
    DRBA_LAUNCH((stage_conv0<FO, G_>), dim3(tiles_x * tiles_y * n_items), dim3(G_::THREADS), lds_bytes, s, its, T, packed_w, bias, \
                hp, wp, ips, prev_scale, H, W, Ho, Wo, tiles_x, n_items);                                                           \
  } while (0)
#define DRBA_SC_GO2(T_, R)       \
  do {                           \
    if (lazy) DRBA_SC_GO(2, T_, R);   \
    else if (fold) DRBA_SC_GO(1, T_, R); \
    else DRBA_SC_GO(0, T_, R);        \
  } while (0)
  if (wres) {
    if (toh == 4) DRBA_SC_GO2(4, true);
    else if (toh == 6) DRBA_SC_GO2(6, true);
    else DRBA_SC_GO2(8, true);
  } else {
    if (toh == 4) DRBA_SC_GO2(4, false);
    else if (toh == 6) DRBA_SC_GO2(6, false);
    else DRBA_SC_GO2(8, false);
  }
#undef DRBA_SC_GO2
#undef DRBA_SC_GO
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // extern "C"
