// Non-GEMM pieces of the IFNet pipeline, fused so that no full-resolution concat, warped
// feature map or intermediate resize ever reaches HBM:
//   ifblock_input : warp x4 + concat + bilinear downsample  -> the stage's conv input
//   ifblock_update: PixelShuffle'd head output -> bilinear upsample + flow accumulate
//   warp_blend    : final two warps + sigmoid blend
// plus bilinear resize and the uint8<->fp32 frame conversions used by to_inp/to_out.
#include "common.hpp"
#include "flow_terms.hpp"

#include <string.h>

#include <stdlib.h>

using namespace drba;

namespace {

// ---- frame source / sink (tools.py:33-38,59-72): bit-exact with ATen's CPU upsample_bilinear2d --------------------
// ATen evaluates the source coordinate as fma(scale, dst + 0.5, -0.5) and each 1-D interpolation as
// fma(w0, a, w1 * b) (its vectorised kernels are compiled with contraction; established by comparing every
// contraction variant against F.interpolate on the 480p/1080p/4K frame sizes: only this one matches in every bit).
// hipcc would pick its own contraction, so the operations are spelled out.
__device__ __forceinline__ Lerp lerp_src_aten(int dst, float scale, int size) {
  float s = __fmaf_rn(scale, (float)dst + 0.5f, -0.5f);
  if (s < 0.f) s = 0.f;
  Lerp l;
  l.i0 = min((int)s, size - 1);
  l.i1 = l.i0 + (l.i0 < size - 1 ? 1 : 0);
  l.w1 = __fsub_rn(s, (float)l.i0);
  l.w0 = __fsub_rn(1.f, l.w1);
  return l;
}
__device__ __forceinline__ float lerp2_aten(const Lerp &ly, const Lerp &lx, float a, float b, float c, float d) {
  const float top = __fmaf_rn(lx.w0, a, __fmul_rn(lx.w1, b));
  const float bot = __fmaf_rn(lx.w0, c, __fmul_rn(lx.w1, d));
  return __fmaf_rn(ly.w0, top, __fmul_rn(ly.w1, bot));
}

// F.interpolate(bilinear, align_corners=False): out = wy0*(wx0*a + wx1*b) + wy1*(wx0*c + wx1*d)
// (ATen's separable evaluation order: innermost axis first).
__global__ void __launch_bounds__(256) resize_bilinear_kernel(const float *__restrict__ in, float *__restrict__ out, int NC, int Hin,
                                       int Win, int Hout, int Wout, float sy, float sx) {
  const size_t total = (size_t)NC * Hout * Wout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wout);
    const int oy = (int)((i / Wout) % Hout);
    const int c = (int)(i / ((size_t)Wout * Hout));
    const Lerp ly = lerp_src_aten(oy, sy, Hin), lx = lerp_src_aten(ox, sx, Win);
    const float *p = in + (size_t)c * Hin * Win;
    const float *r0 = p + (size_t)ly.i0 * Win, *r1 = p + (size_t)ly.i1 * Win;
    out[i] = lerp2_aten(ly, lx, r0[lx.i0], r0[lx.i1], r1[lx.i0], r1[lx.i1]);
  }
}

// to_inp = resize(to_tensor(img), dst_size) in ONE pass: uint8 HWC -> /255. -> bilinear -> fp32 [1,3,Hout,Wout].
// (the reference materialises the full-size fp32 frame in between, tools.py:33-34,59-60)
__global__ void __launch_bounds__(256) to_inp_kernel(const uint8_t *__restrict__ in, float *__restrict__ out, float *__restrict__ out_x4,
                                                     int Hin, int Win, int Hout, int Wout, float sy, float sx) {
  const Tile2D p = tile_pixel(Wout, Hout);
  if (!p.valid) return;
  const Lerp ly = lerp_src_aten(p.y, sy, Hin), lx = lerp_src_aten(p.x, sx, Win);
  const uint8_t *r0 = in + (size_t)ly.i0 * Win * 3, *r1 = in + (size_t)ly.i1 * Win * 3;
  const size_t P = (size_t)Hout * Wout, o = (size_t)p.y * Wout + p.x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a = (float)r0[lx.i0 * 3 + c] / 255.f, b = (float)r0[lx.i1 * 3 + c] / 255.f;
    const float cc = (float)r1[lx.i0 * 3 + c] / 255.f, d = (float)r1[lx.i1 * 3 + c] / 255.f;
    const float v = lerp2_aten(ly, lx, a, b, cc, d);
    out[c * P + o] = v;
    if (out_x4) out_x4[4 * o + c] = v;
  }
  if (out_x4) out_x4[4 * o + 3] = 0.f;
}

// to_out = to_cv2(resize(x, src_size)) in ONE pass: fp32 [1,3,Hin,Win] -> bilinear -> *255. truncated -> uint8 HWC,
// channels reversed when `rev` (the BGR -> RGB flip of the encoder pipe, tools.py:202, otherwise a host-side copy).
__global__ void __launch_bounds__(256) to_out_kernel(const float *__restrict__ in, uint8_t *__restrict__ out, int Hin, int Win,
                                                     int Hout, int Wout, float sy, float sx, int rev) {
  const Tile2D p = tile_pixel(Wout, Hout);
  if (!p.valid) return;
  const Lerp ly = lerp_src_aten(p.y, sy, Hin), lx = lerp_src_aten(p.x, sx, Win);
  const size_t P = (size_t)Hin * Win;
  uint8_t *o = out + ((size_t)p.y * Wout + p.x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float *r0 = in + c * P + (size_t)ly.i0 * Win, *r1 = in + c * P + (size_t)ly.i1 * Win;
    const float v = __fmul_rn(lerp2_aten(ly, lx, r0[lx.i0], r0[lx.i1], r1[lx.i0], r1[lx.i1]), 255.f);
    o[rev ? 2 - c : c] = (uint8_t)(int)v;
  }
}

// The frame sizes of the 1080p / 4K configurations change the HEIGHT only (1080 -> 1088, 2160 -> 2176: the width is already a
// multiple of the padding): the horizontal interpolation is the identity -- scale_x == 1 gives s = x, weights (1, 0), and
// fma(1, a, 0 * b) == a for finite b -- so a lane takes FOUR consecutive pixels of a row with 16-byte loads / stores instead of
// one pixel with 12 scalar loads and 3 byte (or 3 dword) stores: same values bit for bit, a third of the time (round 4: the
// one-pixel kernels were 46 + 2 x 34 us of a 2.43 ms 1080p step at 0.9 TB/s).
typedef float f32x4g __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) to_inp_rows_kernel(const uint8_t *__restrict__ in, float *__restrict__ out, float *__restrict__ out_x4,
                                                          int Hin, int W, int Hout, float sy) {
  const int W4 = W >> 2;
  const size_t total = (size_t)W4 * Hout, P = (size_t)Hout * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / W4), x = (int)(i - (size_t)y * W4) * 4;
    const Lerp ly = lerp_src_aten(y, sy, Hin);
    // 4 pixels x 3 channels = 12 bytes per row, 4-byte aligned (W % 4 == 0)
    const uint32_t *r0 = reinterpret_cast<const uint32_t *>(in + ((size_t)ly.i0 * W + x) * 3);
    const uint32_t *r1 = reinterpret_cast<const uint32_t *>(in + ((size_t)ly.i1 * W + x) * 3);
    const uint32_t a[3] = {r0[0], r0[1], r0[2]}, b[3] = {r1[0], r1[1], r1[2]};
    f32x4g o[3];
#pragma unroll
    for (int px = 0; px < 4; ++px)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int k = px * 3 + c;
        const float top = (float)((a[k >> 2] >> (8 * (k & 3))) & 0xffu) / 255.f, bot = (float)((b[k >> 2] >> (8 * (k & 3))) & 0xffu) / 255.f;
        o[c][px] = __fmaf_rn(ly.w0, top, __fmul_rn(ly.w1, bot));
      }
#pragma unroll
    for (int c = 0; c < 3; ++c) *reinterpret_cast<f32x4g *>(out + (size_t)c * P + (size_t)y * W + x) = o[c];
    if (out_x4) {  // the same values pixel-major, (c0, c1, c2, 0): 64 contiguous bytes per lane
#pragma unroll
      for (int px = 0; px < 4; ++px)
        *reinterpret_cast<f32x4g *>(out_x4 + ((size_t)y * W + x + px) * 4) = (f32x4g){o[0][px], o[1][px], o[2][px], 0.f};
    }
  }
}
__global__ void __launch_bounds__(256) to_out_rows_kernel(const float *__restrict__ in, uint8_t *__restrict__ out, int Hin, int W, int Hout,
                                                          float sy, int rev) {
  const int W4 = W >> 2;
  const size_t total = (size_t)W4 * Hout, P = (size_t)Hin * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / W4), x = (int)(i - (size_t)y * W4) * 4;
    const Lerp ly = lerp_src_aten(y, sy, Hin);
    uint32_t bytes[12];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f32x4g t = *reinterpret_cast<const f32x4g *>(in + (size_t)c * P + (size_t)ly.i0 * W + x);
      const f32x4g u = *reinterpret_cast<const f32x4g *>(in + (size_t)c * P + (size_t)ly.i1 * W + x);
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const float v = __fmul_rn(__fmaf_rn(ly.w0, t[px], __fmul_rn(ly.w1, u[px])), 255.f);
        bytes[px * 3 + (rev ? 2 - c : c)] = (uint32_t)(uint8_t)(int)v;
      }
    }
    uint32_t *o = reinterpret_cast<uint32_t *>(out + ((size_t)y * W + x) * 3);
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = bytes[4 * k] | (bytes[4 * k + 1] << 8) | (bytes[4 * k + 2] << 16) | (bytes[4 * k + 3] << 24);
  }
}

// tools.py:33-34: HWC uint8 -> [1,3,H,W] fp32 / 255.
__global__ void __launch_bounds__(256) u8_to_f32_kernel(const uint8_t *__restrict__ in, float *__restrict__ out, int H, int W) {
  const size_t P = (size_t)H * W;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
    const uint8_t *s = in + p * 3;
    out[p] = (float)s[0] / 255.f;
    out[P + p] = (float)s[1] / 255.f;
    out[2 * P + p] = (float)s[2] / 255.f;
  }
}

// tools.py:37-38: (x*255.).astype(uint8): truncation toward zero, no clamp, no rounding
// (out-of-range values wrap modulo 256 like the x86 float->int32->uint8 conversion numpy performs).
__global__ void __launch_bounds__(256) f32_to_u8_kernel(const float *__restrict__ in, uint8_t *__restrict__ out, int H, int W) {
  const size_t P = (size_t)H * W;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = in[(size_t)c * P + p] * 255.f;
      out[p * 3 + c] = (uint8_t)(int)v;
    }
  }
}

// The interpolations of one step (`-t 2`: two frames) are independent until their convolutions are stacked along N: the
// stage-input and flow-update kernels of ALL items of a stage are one launch (blockIdx.y = item).  On the small maps of
// the first stages these kernels are latency-bound, and every launch also costs the ~5.5 us a dependent dispatch waits
// for its predecessor -- a step's serial chains are bound by exactly that.
constexpr int kMaxItems = DRBA_MAX_STAGE_ITEMS;
struct StageItems {
  drba_stage_item_t it[kMaxItems];
};
struct UpdateItems {
  const float *tmp[kMaxItems];
  const float *flow_in[kMaxItems];
  float *flow_out[kMaxItems];
};
// the kernels' bodies keep their single-item names
#define DRBA_UNPACK_STAGE_ITEM(items) DRBA_UNPACK_STAGE_ITEM_AT(items, blockIdx.y)
#define DRBA_UNPACK_STAGE_ITEM_AT(items, which)                                                                   \
  const drba_stage_item_t &item_ = (items).it[which];                                                             \
  const float *__restrict__ img0 = item_.img0, *__restrict__ img1 = item_.img1, *__restrict__ f0 = item_.f0,      \
                           *__restrict__ f1 = item_.f1, *__restrict__ f0p = item_.f0_pair,                        \
                           *__restrict__ f1p = item_.f1_pair, *__restrict__ tmap = item_.timestep_map,            \
                           *__restrict__ flow = item_.flow, *__restrict__ tmp_prev = item_.tmp_prev;             \
  const float tscalar = item_.timestep_scalar;                                                                    \
  float *__restrict__ out = item_.out

// [C, H, W] -> [C/2, H, W, 2]: the layout the stage-input gathers read the encoder features in (see sample_pair)
__global__ void __launch_bounds__(256) pair_interleave_kernel(const float *__restrict__ in, float *__restrict__ out, int C2, size_t P) {
  const size_t total = (size_t)C2 * P;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t c2 = i / P, p = i - c2 * P;
    float2 v;
    v.x = in[(2 * c2) * P + p];
    v.y = in[(2 * c2 + 1) * P + p];
    reinterpret_cast<float2 *>(out)[i] = v;
  }
}

// [3, H, W] -> [H, W, 4] = (c0, c1, c2, 0): a pixel's channels as one 16-byte unit (drba_stage_item_t.img0_x4)
__global__ void __launch_bounds__(256) rgbx_kernel(const float *__restrict__ in, float *__restrict__ out, size_t P) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (size_t)gridDim.x * blockDim.x)
    reinterpret_cast<float4 *>(out)[i] = make_float4(in[i], in[P + i], in[2 * P + i], 0.f);
}

// ------------------------------------------------------------------------------------------
// IFNet_HDv3.py:146 / :151-156 + IFBlock.forward :85-88.
// One lane per LOW-RES output pixel.  For integer scale s >= 2 the align_corners=False
// downsample touches only the central 2x2 full-res samples of each s x s cell, so the four
// warps are evaluated at 4/s^2 of the full-res pixels (all of them only at s <= 2).
// mask/feat of the previous stage are NOT read from full-resolution tensors: they are the
// x s_prev bilinear upsample of the previous head output `tmp_prev` [13, hp, wp]
// (IFNet_HDv3.py:92-95), evaluated here at the sample points from the (L2-resident) low-res tensor.
// SINGLE: scale == 1, exactly one sample point per output pixel (keeps the register count low).
// Variant 0: one lane per OUTPUT pixel holding all (up to 4) sample points.
template <bool HAS_FLOW, bool SINGLE>
__global__ void __launch_bounds__(256)
ifblock_input_pixel(const StageItems items, int hp, int wp, float inv_prev_scale, int H, int W, int h, int w, float scale) {
  DRBA_UNPACK_STAGE_ITEM(items);
  constexpr int NS = SINGLE ? 1 : 2;
  const size_t P = (size_t)H * W, p_lo = (size_t)h * w, p_prev = (size_t)hp * wp;
  const Tile2D tp_ = tile_pixel(w, h);  // 32 x 8 output tile per workgroup, XCD-banded
  if (tp_.valid) {
    const int oy = tp_.y, ox = tp_.x;
    const size_t o = (size_t)oy * w + ox;
    const Lerp ly = lerp_src(oy, scale, H), lx = lerp_src(ox, scale, W);
    // zero-weight taps are skipped: they would contribute exactly +0.
    const bool use_x1 = !SINGLE && lx.w1 != 0.f, use_y1 = !SINGLE && ly.w1 != 0.f;
    const int Xs[2] = {lx.i0, lx.i1}, Ys[2] = {ly.i0, ly.i1};
    size_t q[NS][NS];
    bool live[NS][NS];
    Taps t0[NS][NS], t1[NS][NS];
    float fl[NS][NS][4];
    Lerp py_[NS], px_[NS];  // source taps of the previous stage's upsample at each sample row / column
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      if (HAS_FLOW) {
        py_[j] = lerp_src(Ys[j], inv_prev_scale, hp);
        px_[j] = lerp_src(Xs[j], inv_prev_scale, wp);
      }
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        live[j][i] = (i == 0 || use_x1) && (j == 0 || use_y1);
        q[j][i] = (size_t)Ys[j] * W + Xs[i];
        if (HAS_FLOW && live[j][i]) {
#pragma unroll
          for (int c = 0; c < 4; ++c) fl[j][i][c] = flow[(size_t)c * P + q[j][i]];
          t0[j][i] = taps_border(warp_coord(Xs[i], W, fl[j][i][0]), warp_coord(Ys[j], H, fl[j][i][1]), W, H);
          t1[j][i] = taps_border(warp_coord(Xs[i], W, fl[j][i][2]), warp_coord(Ys[j], H, fl[j][i][3]), W, H);
        }
      }
    }
    // separable lerp, innermost axis first: wy0*(wx0*V00 + wx1*V01) + wy1*(wx0*V10 + wx1*V11)
    auto lerp4 = [&](auto &&val) -> float {
      if (SINGLE) return ly.w0 * (lx.w0 * val(0, 0) + lx.w1 * 0.f) + ly.w1 * 0.f;
      const float v00 = val(0, 0);
      const float v01 = live[0][NS - 1] ? val(0, NS - 1) : 0.f;
      const float top = lx.w0 * v00 + lx.w1 * v01;
      float bot = 0.f;
      if (use_y1) {
        const float v10 = val(NS - 1, 0);
        const float v11 = live[NS - 1][NS - 1] ? val(NS - 1, NS - 1) : 0.f;
        bot = lx.w0 * v10 + lx.w1 * v11;
      }
      return ly.w0 * top + ly.w1 * bot;
    };
    // previous head output channel c, bilinearly upsampled to full-res sample point (j, i)
    auto prev_up = [&](int c, int j, int i) -> float {
      const float *tp = tmp_prev + (size_t)c * p_prev;
      const Lerp &a = py_[j], &b = px_[i];
      const float top = b.w0 * tp[(size_t)a.i0 * wp + b.i0] + b.w1 * tp[(size_t)a.i0 * wp + b.i1];
      const float bot = b.w0 * tp[(size_t)a.i1 * wp + b.i0] + b.w1 * tp[(size_t)a.i1 * wp + b.i1];
      return a.w0 * top + a.w1 * bot;
    };
    float *dst = out + o;
    if (HAS_FLOW) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float *pl0 = img0 + (size_t)c * P, *pl1 = img1 + (size_t)c * P;
        dst[(size_t)c * p_lo] = lerp4([&](int j, int i) { return sample(pl0, W, t0[j][i]); });
        dst[(size_t)(3 + c) * p_lo] = lerp4([&](int j, int i) { return sample(pl1, W, t1[j][i]); });
      }
      if (SINGLE && f0p) {  // pair-interleaved feature planes: both channels of a pair from two 16-byte loads
        for (int c2 = 0; c2 < 8; ++c2) {
          float a0, a1, b0, b1;
          sample_pair(f0p + (size_t)c2 * 2 * P, W, t0[0][0], a0, a1);
          sample_pair(f1p + (size_t)c2 * 2 * P, W, t1[0][0], b0, b1);
          dst[(size_t)(6 + 2 * c2) * p_lo] = lerp4([&](int, int) { return a0; });
          dst[(size_t)(7 + 2 * c2) * p_lo] = lerp4([&](int, int) { return a1; });
          dst[(size_t)(22 + 2 * c2) * p_lo] = lerp4([&](int, int) { return b0; });
          dst[(size_t)(23 + 2 * c2) * p_lo] = lerp4([&](int, int) { return b1; });
        }
      } else {
        for (int c = 0; c < 16; ++c) {
          const float *pl0 = f0 + (size_t)c * P, *pl1 = f1 + (size_t)c * P;
          dst[(size_t)(6 + c) * p_lo] = lerp4([&](int j, int i) { return sample(pl0, W, t0[j][i]); });
          dst[(size_t)(22 + c) * p_lo] = lerp4([&](int j, int i) { return sample(pl1, W, t1[j][i]); });
        }
      }
      dst[(size_t)38 * p_lo] = lerp4([&](int j, int i) { return tmap ? tmap[q[j][i]] : tscalar; });
      for (int c = 0; c < 9; ++c)  // mask (tmp[4]) then feat (tmp[5:13])
        dst[(size_t)(39 + c) * p_lo] = lerp4([&](int j, int i) { return prev_up(4 + c, j, i); });
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float v = lerp4([&](int j, int i) { return fl[j][i][c]; });
        dst[(size_t)(48 + c) * p_lo] = (v * 1.f) / scale;  // interpolate(flow) * 1. / scale (IFNet_HDv3.py:87)
      }
    } else {
      for (int c = 0; c < 3; ++c) {
        const float *pl0 = img0 + (size_t)c * P, *pl1 = img1 + (size_t)c * P;
        dst[(size_t)c * p_lo] = lerp4([&](int j, int i) { return pl0[q[j][i]]; });
        dst[(size_t)(3 + c) * p_lo] = lerp4([&](int j, int i) { return pl1[q[j][i]]; });
      }
      if (f0p) {  // pair-interleaved features (the only layout head_fused writes on the hot path): channel c of pixel q at [c / 2][q][c % 2]
        for (int c = 0; c < 16; ++c) {
          const float *pl0 = f0p + (size_t)(c >> 1) * 2 * P + (c & 1), *pl1 = f1p + (size_t)(c >> 1) * 2 * P + (c & 1);
          dst[(size_t)(6 + c) * p_lo] = lerp4([&](int j, int i) { return pl0[2 * q[j][i]]; });
          dst[(size_t)(22 + c) * p_lo] = lerp4([&](int j, int i) { return pl1[2 * q[j][i]]; });
        }
      } else {
        for (int c = 0; c < 16; ++c) {
          const float *pl0 = f0 + (size_t)c * P, *pl1 = f1 + (size_t)c * P;
          dst[(size_t)(6 + c) * p_lo] = lerp4([&](int j, int i) { return pl0[q[j][i]]; });
          dst[(size_t)(22 + c) * p_lo] = lerp4([&](int j, int i) { return pl1[q[j][i]]; });
        }
      }
      dst[(size_t)38 * p_lo] = lerp4([&](int j, int i) { return tmap ? tmap[q[j][i]] : tscalar; });
    }
  }
}

// SINGLE (scale == 1): one lane per output pixel, one sample point.
// otherwise: FOUR lanes per output pixel, one per sample point (j,i) of the 2x2 bilinear footprint;
// the separable lerp  wy0*(wx0*V00 + wx1*V01) + wy1*(wx0*V10 + wx1*V11)  is formed across the lane
// quad with two xor-shuffles (fp add is commutative, so every lane gets the reference's exact sum).
// This keeps the per-lane state at one sample point (86 VGPRs instead of 211) and gives the
// small low-resolution stages 4x more lanes in flight.
template <bool HAS_FLOW, bool SINGLE, int UNR>
__global__ void __launch_bounds__(256)
ifblock_input_kernel(const StageItems items, int hp, int wp, float inv_prev_scale, int H, int W, int h, int w, float scale) {
  DRBA_UNPACK_STAGE_ITEM(items);
  constexpr int LPO = SINGLE ? 1 : 4;  // lanes per output pixel
  const size_t P = (size_t)H * W, p_lo = (size_t)h * w, p_prev = (size_t)hp * wp;
  // workgroup = XCD-banded output tile: 32 x 8 px (SINGLE) or 16 x 4 px x 4 sample lanes; every lane runs
  // (the shuffles need whole quads), out-of-image lanes are clamped and do not store
  constexpr int TWo = SINGLE ? 32 : 16, THo = SINGLE ? 8 : 4;
  {
    const int tiles_x = (w + TWo - 1) / TWo;
    int tx, ty;
    xcd_strip_tile(blockIdx.x, gridDim.x, tiles_x, tx, ty);
    const int lo = threadIdx.x / LPO;
    const int ox_raw = tx * TWo + (lo % TWo), oy_raw = ty * THo + (lo / TWo);
    const bool valid = ox_raw < w && oy_raw < h;
    const int ox = min(ox_raw, w - 1), oy = min(oy_raw, h - 1);
    const size_t o = (size_t)oy * w + ox;
    const int sub = (int)(threadIdx.x % LPO), sj = sub >> 1, si = sub & 1;
    const Lerp ly = lerp_src(oy, scale, H), lx = lerp_src(ox, scale, W);
    const int X = si ? lx.i1 : lx.i0, Y = sj ? ly.i1 : ly.i0;
    const float wx = si ? lx.w1 : lx.w0, wy = sj ? ly.w1 : ly.w0;
    const size_t q = (size_t)Y * W + X;
    auto comb = [&](float v) -> float {
      if (SINGLE) return ly.w0 * (lx.w0 * v + lx.w1 * 0.f) + ly.w1 * 0.f;
      const float a = wx * v;
      const float row = a + quad_xor1(a);  // wx0*V_j0 + wx1*V_j1 (the quad's four lanes are the 2 x 2 sample points)
      const float b = wy * row;
      return b + quad_xor2(b);             // wy0*top + wy1*bot
    };
    const bool writer = valid && sub == 0;
    float *dst = out + o;
    if (HAS_FLOW) {
      const float fl0 = flow[q], fl1 = flow[P + q], fl2 = flow[2 * P + q], fl3 = flow[3 * P + q];
      const Taps t0 = taps_border(warp_coord(X, W, fl0), warp_coord(Y, H, fl1), W, H);
      const Taps t1 = taps_border(warp_coord(X, W, fl2), warp_coord(Y, H, fl3), W, H);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v0 = comb(sample(img0 + (size_t)c * P, W, t0)), v1 = comb(sample(img1 + (size_t)c * P, W, t1));
        if (writer) {
          dst[(size_t)c * p_lo] = v0;
          dst[(size_t)(3 + c) * p_lo] = v1;
        }
      }
      if (f0p) {  // pair-interleaved feature planes: both channels of a pair from two 16-byte loads
#pragma unroll UNR
        for (int c2 = 0; c2 < 8; ++c2) {
          float a0, a1, b0, b1;
          sample_pair(f0p + (size_t)c2 * 2 * P, W, t0, a0, a1);
          sample_pair(f1p + (size_t)c2 * 2 * P, W, t1, b0, b1);
          a0 = comb(a0);
          a1 = comb(a1);
          b0 = comb(b0);
          b1 = comb(b1);
          if (writer) {
            dst[(size_t)(6 + 2 * c2) * p_lo] = a0;
            dst[(size_t)(7 + 2 * c2) * p_lo] = a1;
            dst[(size_t)(22 + 2 * c2) * p_lo] = b0;
            dst[(size_t)(23 + 2 * c2) * p_lo] = b1;
          }
        }
      } else {
#pragma unroll UNR
        for (int c = 0; c < 16; ++c) {
          const float v0 = comb(sample(f0 + (size_t)c * P, W, t0)), v1 = comb(sample(f1 + (size_t)c * P, W, t1));
          if (writer) {
            dst[(size_t)(6 + c) * p_lo] = v0;
            dst[(size_t)(22 + c) * p_lo] = v1;
          }
        }
      }
      {
        const float v = comb(tmap ? tmap[q] : tscalar);
        if (writer) dst[(size_t)38 * p_lo] = v;
      }
      // mask (tmp[4]) and feat (tmp[5:13]) = x s_prev upsample of the previous head output at (X, Y)
      const Lerp a = lerp_src(Y, inv_prev_scale, hp), b = lerp_src(X, inv_prev_scale, wp);
      const size_t o00 = (size_t)a.i0 * wp + b.i0, o01 = (size_t)a.i0 * wp + b.i1;
      const size_t o10 = (size_t)a.i1 * wp + b.i0, o11 = (size_t)a.i1 * wp + b.i1;
#pragma unroll UNR
      for (int c = 0; c < 9; ++c) {
        const float *tp = tmp_prev + (size_t)(4 + c) * p_prev;
        const float top = b.w0 * tp[o00] + b.w1 * tp[o01];
        const float bot = b.w0 * tp[o10] + b.w1 * tp[o11];
        const float v = comb(a.w0 * top + a.w1 * bot);
        if (writer) dst[(size_t)(39 + c) * p_lo] = v;
      }
      const float fls[4] = {fl0, fl1, fl2, fl3};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float v = comb(fls[c]);
        if (writer) dst[(size_t)(48 + c) * p_lo] = (v * 1.f) / scale;  // interpolate(flow) * 1. / scale (IFNet_HDv3.py:87)
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v0 = comb(img0[(size_t)c * P + q]), v1 = comb(img1[(size_t)c * P + q]);
        if (writer) {
          dst[(size_t)c * p_lo] = v0;
          dst[(size_t)(3 + c) * p_lo] = v1;
        }
      }
      if (f0p) {  // pair-interleaved features: both channels of a pair in one 8-byte load
#pragma unroll UNR
        for (int c2 = 0; c2 < 8; ++c2) {
          const f32x2u a = *reinterpret_cast<const f32x2u *>(f0p + ((size_t)c2 * P + q) * 2);
          const f32x2u b = *reinterpret_cast<const f32x2u *>(f1p + ((size_t)c2 * P + q) * 2);
          const float a0 = comb(a.x), a1 = comb(a.y), b0 = comb(b.x), b1 = comb(b.y);
          if (writer) {
            dst[(size_t)(6 + 2 * c2) * p_lo] = a0;
            dst[(size_t)(7 + 2 * c2) * p_lo] = a1;
            dst[(size_t)(22 + 2 * c2) * p_lo] = b0;
            dst[(size_t)(23 + 2 * c2) * p_lo] = b1;
          }
        }
      } else {
#pragma unroll UNR
        for (int c = 0; c < 16; ++c) {
          const float v0 = comb(f0[(size_t)c * P + q]), v1 = comb(f1[(size_t)c * P + q]);
          if (writer) {
            dst[(size_t)(6 + c) * p_lo] = v0;
            dst[(size_t)(22 + c) * p_lo] = v1;
          }
        }
      }
      const float v = comb(tmap ? tmap[q] : tscalar);
      if (writer) dst[(size_t)38 * p_lo] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Warped stage input with the previous head output staged through LDS, optionally with the previous stage's flow
// update folded in (IFNet_HDv3.py:85-95,146-160).
//
// The kernels above evaluate mask / feat (x s_prev upsample of tmp_prev's channels 4..12) with 36 scattered global
// loads per sample point; they are bound by the NUMBER of vector-memory instructions (TA busy 86-98 %).  A workgroup's
// sample points cover a 32 x 8 (scale 1, 2) or 16s x 4s full-resolution block, whose footprint in tmp_prev is at most
// 18 x 6 pixels: the 13-channel footprint (<= 5.6 KB) is staged once per workgroup and the taps become LDS reads.
// With FOLD the running flow is not read as the finished sum: flow = flow_prev + up(tmp_prev[0:4]) * s_prev is formed
// here from the same LDS tile (ifblock_update's arithmetic), written to flow_out and used for the warps.  That needs
// every full-resolution pixel to be a sample point exactly once, i.e. scale <= 2 (the two full-resolution stages of
// the 1080p path): one 8P-byte read-modify-write pass and one launch per sample and stage disappear.
constexpr int kPrevRW = 20, kPrevRH = 6;  // LDS footprint capacity (columns padded to 20)
// VS ("vector stores", taken when the output is a whole number of tiles and rows are 16-byte aligned): a lane computes
// one value per channel, so stored directly a wave instruction writes 64 (scale 1) or only 16 (scale >= 2, one writer lane
// per 2 x 2 sample quad) floats of ONE channel -- 52 (+4) store instructions per wave on kernels that are bound by the
// number of vector-memory instructions.  Instead the values are parked in a wave-private LDS strip and written back
// as 16 bytes per lane: scale 1: 4 channels x (32 x 2 pixels) per store, 14 stores; scale >= 2: all 52 channels x 16
// pixels in 4 stores; the folded flow update (a 32 x 2 full-resolution block per wave either way) in one.
typedef float f32x4a __attribute__((ext_vector_type(4)));
// FMODE: 0 = the finished flow is read; 1 = FOLD (above: flow_prev + the previous stage's update, written to flow_out);
// 2 = LAZY: the flow is the sum of the terms (flow_terms.hpp) + the previous stage's update, nothing is written -- any scale.
constexpr int kTermR = 5, kTermC = 12;  // term footprint capacity under a tile (32 x 8 px at scale <= 2, 16s x 4s beyond; terms at >= 4 x scale)
template <bool SINGLE, int FMODE, bool VS>
__global__ void __launch_bounds__(256)
ifblock_input_lds(const StageItems items, const FlowTermsArg T, int hp, int wp, float inv_prev_scale, float prev_scale, int H, int W,
                  int h, int w, float scale, int n_items) {
  constexpr bool FOLD = FMODE != 0, WRITES = FMODE == 1, LAZY = FMODE == 2;
  int vb_, vitem_, ntiles_;
  tile_item_block(n_items, vb_, vitem_, ntiles_);  // one grid dimension: the items of a tile back to back on one XCD
  DRBA_UNPACK_STAGE_ITEM_AT(items, vitem_);
  float *__restrict__ flow_out = item_.flow_out;
  // [row][column][16]: the 13 channels of a footprint pixel (padded to 16) are four 16-byte LDS words, so a sample point
  // reads its 2 x 2 taps of four channels with 4 ds_read_b128 (16 reads for all 13 channels instead of 52 ds_read_b32)
  __shared__ __attribute__((aligned(16))) float prev[kPrevRH][kPrevRW][16];
  __shared__ __attribute__((aligned(16))) float tl[LAZY ? kMaxTerms * 4 * kTermR * kTermC : 4];
  int trx0[kMaxTerms], try0[kMaxTerms];
  constexpr int STG = VS ? (SINGLE ? 4 * 64 : 52 * 16 + 4 * 64) : 1;  // floats per wave
  __shared__ __attribute__((aligned(16))) float stg_all[4 * STG];
  constexpr int LPO = SINGLE ? 1 : 4;
  constexpr int TWo = SINGLE ? 32 : 16, THo = SINGLE ? 8 : 4;
  constexpr int C0 = FOLD ? 0 : 4;  // first channel of tmp_prev that is needed
  const size_t P = (size_t)H * W, p_lo = (size_t)h * w, p_prev = (size_t)hp * wp;
  const int tiles_x = (w + TWo - 1) / TWo;
  int tx, ty;
  xcd_strip_tile(vb_, ntiles_, tiles_x, tx, ty);
  // footprint of this workgroup's sample points in tmp_prev (same for every lane: computed from the tile corners)
  const int ox_a = tx * TWo, oy_a = ty * THo;
  const int ox_b = min(ox_a + TWo - 1, w - 1), oy_b = min(oy_a + THo - 1, h - 1);
  const int Xa = lerp_src(ox_a, scale, W).i0, Ya = lerp_src(oy_a, scale, H).i0;
  const int Xb = SINGLE ? ox_b : lerp_src(ox_b, scale, W).i1, Yb = SINGLE ? oy_b : lerp_src(oy_b, scale, H).i1;
  const int rx0 = lerp_src(Xa, inv_prev_scale, wp).i0, ry0 = lerp_src(Ya, inv_prev_scale, hp).i0;
  const int rw = lerp_src(Xb, inv_prev_scale, wp).i1 - rx0 + 1, rh = lerp_src(Yb, inv_prev_scale, hp).i1 - ry0 + 1;
  {
    // thread -> (footprint pixel e of the CAPACITY grid, half of the channels): compile-time divisors.  (The loop used to run over
    // (13 - C0) * rh * rw values with two runtime integer divisions each -- ~70 VALU instructions per value, five values per thread.)
    static_assert(kPrevRH * kPrevRW <= 128, "one footprint pixel per thread of a half workgroup");
    constexpr int NCH = 13 - C0, HALF = (NCH + 1) / 2;
    const int e = threadIdx.x & 127, cg = threadIdx.x >> 7;
    const int r = e / kPrevRW, col = e - r * kPrevRW;
    if (r < rh && col < rw) {
      const float *src = tmp_prev + (size_t)(ry0 + r) * wp + rx0 + col;
      float v[HALF];
#pragma unroll
      for (int k = 0; k < HALF; ++k) {
        const int c = C0 + cg * HALF + k;
        v[k] = c < 13 ? src[(size_t)c * p_prev] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < HALF; ++k) {
        const int c = C0 + cg * HALF + k;
        if (c < 13) prev[r][col][c] = v[k];
      }
    }
  }
  if (LAZY) terms_stage<kTermR, kTermC, 256>(tl, T, item_.term, Xa, Ya, Xb, Yb, threadIdx.x, trx0, try0);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *ws = stg_all + wave * STG;
  const int lo = threadIdx.x / LPO;
  const int ox_raw = ox_a + (lo % TWo), oy_raw = oy_a + (lo / TWo);
  const bool valid = ox_raw < w && oy_raw < h;
  const int ox = min(ox_raw, w - 1), oy = min(oy_raw, h - 1);
  const size_t o = (size_t)oy * w + ox;
  const int sub = (int)(threadIdx.x % LPO), sj = sub >> 1, si = sub & 1;
  const Lerp ly = lerp_src(oy, scale, H), lx = lerp_src(ox, scale, W);
  const int X = si ? lx.i1 : lx.i0, Y = sj ? ly.i1 : ly.i0;
  const float wx = si ? lx.w1 : lx.w0, wy = sj ? ly.w1 : ly.w0;
  const size_t q = (size_t)Y * W + X;
  auto comb = [&](float v) -> float {
    if (SINGLE) return v;  // scale 1: the "downsample" has weights (1, 0) x (1, 0) -- v * 1 + 0 * 0 + 0 * 0, the identity
    const float a = wx * v;
    const float row = a + quad_xor1(a);
    const float b = wy * row;
    return b + quad_xor2(b);
  };
  const bool writer = valid && sub == 0;
  float *dst = out + o;
  // ---- VS plumbing.  A 32 x 2 block (scale 1: the wave's output pixels; FOLD at scale 2: its full-resolution sample
  // points) is parked as [4 slots][64] and written by lane l as the 16 bytes of slot l >> 4, pixels 4*(l & 15)..+3.
  const int g4 = lane >> 4, qd = lane & 15;
  // The strip is exchanged between the lanes of ONE wave: LDS operations of a wave retire in order, so no s_barrier is
  // needed, but the compiler must keep the parked writes in front of the other lanes' reads and the reads in front of
  // the next round's writes -- wavefront-scope fences + wave barriers pin that order (they emit no instruction).
  auto strip_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto flush32x2 = [&](const float *strip, float *base, size_t plane, int width, int x0, int y0, int ch) {
    strip_sync();
    const f32x4a v = *reinterpret_cast<const f32x4a *>(strip + g4 * 64 + qd * 4);
    strip_sync();
    *reinterpret_cast<f32x4a *>(base + (size_t)ch * plane + (size_t)(y0 + (qd >> 3)) * width + x0 + (qd & 7) * 4) = v;
  };
  // scale 1: emit(slot, v) parks channel values, flush(ch of this lane's 16-lane group) writes 4 channels at once
  auto park = [&](int slot, float v) { ws[slot * 64 + lane] = v; };
  auto flush_out = [&](int ch) { flush32x2(ws, out, p_lo, w, ox_a, oy_a + 2 * wave, ch); };
  // scale >= 2: channel values of the wave's 16 output pixels parked as [52][16]
  auto emit = [&](int ch, float v) {
    if (VS) {
      if (!SINGLE && sub == 0) ws[ch * 16 + (lo & 15)] = v;
    } else if (writer) {
      dst[(size_t)ch * p_lo] = v;
    }
  };
  // taps of the previous head output's upsample at (X, Y), relative to the staged footprint
  const Lerp a = lerp_src(Y, inv_prev_scale, hp), b = lerp_src(X, inv_prev_scale, wp);
  const int r0 = a.i0 - ry0, r1 = a.i1 - ry0, c0 = b.i0 - rx0, c1 = b.i1 - rx0;
  auto prev_up4 = [&](int k) -> f32x4a {  // channels 4k .. 4k+3 of the previous head output, upsampled to (X, Y)
    const f32x4a q00 = *reinterpret_cast<const f32x4a *>(&prev[r0][c0][4 * k]), q01 = *reinterpret_cast<const f32x4a *>(&prev[r0][c1][4 * k]);
    const f32x4a q10 = *reinterpret_cast<const f32x4a *>(&prev[r1][c0][4 * k]), q11 = *reinterpret_cast<const f32x4a *>(&prev[r1][c1][4 * k]);
    f32x4a v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = lerp2_fma(a.w0, a.w1, b.w0, b.w1, q00[j], q01[j], q10[j], q11[j]);
    return v;
  };
  const f32x4a pu0 = FOLD ? prev_up4(0) : (f32x4a){0.f, 0.f, 0.f, 0.f};  // flow delta
  auto prev_up = [&](int c) -> float { return pu0[c & 3]; };                // c < 4; mask / feat are read where they are emitted (registers)
  float fls[4];
  const bool have_terms = LAZY && terms_flow<kTermR, kTermC>(tl, T, trx0, try0, X, Y, fls);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (LAZY) {
      const float fd = __fmul_rn(prev_up(c), prev_scale);
      fls[c] = have_terms ? __fadd_rn(fls[c], fd) : fd;
    } else if (FOLD) {
      // ifblock_update: flow_in + up(tmp) * scale -- product and sum rounded separately, as torch evaluates them
      const float fd = __fmul_rn(prev_up(c), prev_scale);
      fls[c] = flow ? __fadd_rn(flow[(size_t)c * P + q], fd) : fd;
      // scale <= 2: the sample points are the full-resolution pixels, each exactly once
      if (VS) {
        if (SINGLE) park(c, fls[c]);
        else ws[52 * 16 + c * 64 + sj * 32 + (lo & 15) * 2 + si] = fls[c];
      } else if (valid) {
        flow_out[(size_t)c * P + q] = fls[c];
      }
    } else {
      fls[c] = flow[(size_t)c * P + q];
    }
  }
  if (VS && WRITES) {
    if (SINGLE) flush32x2(ws, flow_out, P, W, ox_a, oy_a + 2 * wave, g4);
    else flush32x2(ws + 52 * 16, flow_out, P, W, 2 * ox_a, 2 * (oy_a + wave), g4);  // scale 2: rows 2y, 2y+1, columns 2x..
  }
  const Taps t0 = taps_border(warp_coord(X, W, fls[0]), warp_coord(Y, H, fls[1]), W, H);
  const Taps t1 = taps_border(warp_coord(X, W, fls[2]), warp_coord(Y, H, fls[3]), W, H);
  // sample() / sample_pair() of common.hpp with everything that does not depend on the channel taken out of the channel
  // loops: 32-bit element offsets of the two tap rows, and the right-border case (x0 == W-1: the pair is loaded one
  // column to the left and its SECOND element is the left tap) folded into the four weights instead of four selects per
  // sample.  Same products, same order, same zeros added: bit-identical to the helpers for finite inputs.
  struct TapW {
    int o0, o1;
    float w00, w01, w10, w11;
  };
  auto tapw = [&](const Taps &t) -> TapW {
    const int xb = min(t.x0, W - 2);
    const bool edge = t.x0 != xb;
    TapW k;
    k.o0 = t.y0 * W + xb, k.o1 = t.y1 * W + xb;
    k.w00 = edge ? 0.f : t.wnw, k.w01 = edge ? t.wnw : t.wne;
    k.w10 = edge ? 0.f : t.wsw, k.w11 = edge ? t.wsw : t.wse;
    return k;
  };
  const TapW k0 = tapw(t0), k1 = tapw(t1);
  auto sample = [&](const float *__restrict__ p, int, const TapW &k) -> float {
    const f32x2u a = *reinterpret_cast<const f32x2u *>(p + k.o0), b = *reinterpret_cast<const f32x2u *>(p + k.o1);
    return a.x * k.w00 + a.y * k.w01 + b.x * k.w10 + b.y * k.w11;
  };
  auto sample_pair = [&](const float *__restrict__ pp, int, const TapW &k, float &v0, float &v1) {
    const f32x4u a = *reinterpret_cast<const f32x4u *>(pp + 2 * k.o0), b = *reinterpret_cast<const f32x4u *>(pp + 2 * k.o1);
    v0 = a.x * k.w00 + a.z * k.w01 + b.x * k.w10 + b.z * k.w11;
    v1 = a.y * k.w00 + a.w * k.w01 + b.y * k.w10 + b.w * k.w11;
  };
  const float tmv = comb(tmap ? tmap[q] : tscalar);
  // the three channels of a frame at this point: from its [H][W][4] copy when the item has one (two 16-byte loads per tap row
  // instead of three 8-byte ones: 8 gathers per point instead of 12), else from the planes
  const float *__restrict__ img0x = item_.img0_x4, *__restrict__ img1x = item_.img1_x4;
  auto sample3 = [&](const float *__restrict__ planar, const float *__restrict__ x4, const TapW &k, float (&v)[3]) {
    if (x4) {
      const f32x4a a0 = *reinterpret_cast<const f32x4a *>(x4 + 4 * (size_t)k.o0), a1 = *reinterpret_cast<const f32x4a *>(x4 + 4 * (size_t)k.o0 + 4);
      const f32x4a b0 = *reinterpret_cast<const f32x4a *>(x4 + 4 * (size_t)k.o1), b1 = *reinterpret_cast<const f32x4a *>(x4 + 4 * (size_t)k.o1 + 4);
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = a0[c] * k.w00 + a1[c] * k.w01 + b0[c] * k.w10 + b1[c] * k.w11;
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = sample(planar + (size_t)c * P, W, k);
    }
  };
  float i0v[3], i1v[3];
  sample3(img0, img0x, k0, i0v);
  sample3(img1, img1x, k1, i1v);
  if (VS && SINGLE) {
    // batches of four channels: {img0 0..2, timestep}, {img1 0..2, mask}, 8 x {f0 pair, f1 pair}, feat 0..3, feat 4..7, flow
#pragma unroll
    for (int c = 0; c < 3; ++c) park(c, comb(i0v[c]));
    park(3, tmv);
    flush_out(g4 < 3 ? g4 : 38);
#pragma unroll
    for (int c = 0; c < 3; ++c) park(c, comb(i1v[c]));
    park(3, comb(prev_up4(1)[0]));  // mask = channel 4
    flush_out(g4 < 3 ? 3 + g4 : 39);
#pragma unroll 1
    for (int c2 = 0; c2 < 8; ++c2) {
      float a0, a1, b0, b1;
      if (f0p) {
        sample_pair(f0p + (size_t)c2 * 2 * P, W, k0, a0, a1);
        sample_pair(f1p + (size_t)c2 * 2 * P, W, k1, b0, b1);
      } else {
        a0 = sample(f0 + (size_t)(2 * c2) * P, W, k0), a1 = sample(f0 + (size_t)(2 * c2 + 1) * P, W, k0);
        b0 = sample(f1 + (size_t)(2 * c2) * P, W, k1), b1 = sample(f1 + (size_t)(2 * c2 + 1) * P, W, k1);
      }
      park(0, comb(a0)), park(1, comb(a1)), park(2, comb(b0)), park(3, comb(b1));
      flush_out(6 + 2 * c2 + (g4 & 1) + (g4 >> 1) * 16);
    }
    {
      const f32x4a p1 = prev_up4(1), p2 = prev_up4(2), p3 = prev_up4(3);  // channels 4..15: feat = 5..12
      const float ft[8] = {p1[1], p1[2], p1[3], p2[0], p2[1], p2[2], p2[3], p3[0]};
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int c = 0; c < 4; ++c) park(c, comb(ft[4 * half + c]));
        flush_out(40 + 4 * half + g4);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) park(c, (comb(fls[c]) * 1.f) / scale);  // interpolate(flow) * 1. / scale (IFNet_HDv3.py:87)
    flush_out(48 + g4);
    return;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v0 = comb(i0v[c]), v1 = comb(i1v[c]);
    emit(c, v0);
    emit(3 + c, v1);
  }
  if (f0p) {
#pragma unroll 1
    for (int c2 = 0; c2 < 8; ++c2) {
      float a0, a1, b0, b1;
      sample_pair(f0p + (size_t)c2 * 2 * P, W, k0, a0, a1);
      sample_pair(f1p + (size_t)c2 * 2 * P, W, k1, b0, b1);
      a0 = comb(a0), a1 = comb(a1), b0 = comb(b0), b1 = comb(b1);
      emit(6 + 2 * c2, a0);
      emit(7 + 2 * c2, a1);
      emit(22 + 2 * c2, b0);
      emit(23 + 2 * c2, b1);
    }
  } else {
#pragma unroll 1
    for (int c = 0; c < 16; ++c) {
      const float v0 = comb(sample(f0 + (size_t)c * P, W, k0)), v1 = comb(sample(f1 + (size_t)c * P, W, k1));
      emit(6 + c, v0);
      emit(22 + c, v1);
    }
  }
  emit(38, tmv);
  {
    const f32x4a p1 = prev_up4(1), p2 = prev_up4(2), p3 = prev_up4(3);  // mask (tmp[4]) and feat (tmp[5:13])
    const float mf[9] = {p1[0], p1[1], p1[2], p1[3], p2[0], p2[1], p2[2], p2[3], p3[0]};
#pragma unroll
    for (int c = 0; c < 9; ++c) emit(39 + c, comb(mf[c]));
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) emit(48 + c, (comb(fls[c]) * 1.f) / scale);  // interpolate(flow) * 1. / scale (IFNet_HDv3.py:87)
  if (VS && !SINGLE) {
    // [52][16] -> lane l of pass i writes the 16 bytes of channel (64 i + l) >> 2, pixels 4 * (l & 3)..+3 of the wave's row
    strip_sync();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = i * 64 + lane, ch = idx >> 2, qq = idx & 3;
      if (ch < 52) {
        const f32x4a v = *reinterpret_cast<const f32x4a *>(ws + ch * 16 + qq * 4);
        *reinterpret_cast<f32x4a *>(out + (size_t)ch * p_lo + (size_t)(oy_a + wave) * w + ox_a + qq * 4) = v;
      }
    }
  }
}

// warp_blend with the LAST stage's flow update folded in: flow = flow_prev + up(tmp[0:4]) * scale is formed per pixel
// (never stored: nothing reads the final flow), the mask is tmp's channel 4; tmp's footprint is staged through LDS.
// LAZY: the flow before the last stage is the sum of the terms (flow_terms.hpp); the items of a stage in one launch.
constexpr int kWbTermR = 6, kWbTermC = 18;  // term footprint capacity under a 32 x 8 tile (terms at >= 2 x the last scale >= 2)
struct BlendItems {
  const float *img0[kMaxItems], *img1[kMaxItems], *flow[kMaxItems], *tmp[kMaxItems];
  const float *img0x[kMaxItems], *img1x[kMaxItems];  // the frames as [H][W][4] (drba_stage_item_t.img0_x4), or NULL
  float *out[kMaxItems];
  const float *term[kMaxItems][kMaxTerms];
};
// S1 (round 6): the last stage ran at the frame's own resolution (scale == 1, h == H, w == W: the default 1080p configuration).
// The factor-1 "upsample" has weights exactly (1, 0) on the pixel itself and its right / lower neighbour -- lerp_src(x, 1, W) is
// (x, x + 1, 1, 0) -- so up(c) IS tmp[c][y][x] for finite data, and staging a 10 x 34 footprint of five channels through LDS (6.6
// scalar loads per pixel, two runtime integer divisions per staged value) to evaluate it was the larger half of the kernel's 1091
// VALU instructions per wave (PMC, `profiles/r06_warp_blend_pmc.txt`: the vector ALUs 58 % busy, i.e. the kernel was VALU-bound, not
// gather-bound).  The pixel's five values are five coalesced loads here.  (A non-finite NEIGHBOUR made the staged form NaN through
// 0 * inf as ATen's interpolate does; this form hands on the pixel's own value.)
template <bool LAZY, bool S1 = false>
__global__ void __launch_bounds__(256)
warp_blend_fold_kernel(const BlendItems items, const FlowTermsArg T, int h, int w, float inv_scale, float scale, int H, int W, int n_items) {
  int vb_, vitem_, ntiles_;
  tile_item_block(n_items, vb_, vitem_, ntiles_);  // one grid dimension: the items of a tile back to back on one XCD
  const float *__restrict__ img0 = items.img0[vitem_], *__restrict__ img1 = items.img1[vitem_];
  const float *__restrict__ flow = items.flow[vitem_], *__restrict__ tmp = items.tmp[vitem_];
  const float *__restrict__ img0x = items.img0x[vitem_], *__restrict__ img1x = items.img1x[vitem_];
  float *__restrict__ out = items.out[vitem_];
  __shared__ __attribute__((aligned(16))) float prev[S1 ? 1 : 10][S1 ? 1 : 36][8];  // [row][column][flow 0..3 | mask, 3 x padding]: 16-byte LDS words
  __shared__ __attribute__((aligned(16))) float tl[LAZY ? kMaxTerms * 4 * kWbTermR * kWbTermC : 4];
  int trx0[kMaxTerms], try0[kMaxTerms];
  const size_t P = (size_t)H * W, p_lo = (size_t)h * w;
  const int tiles_x = (W + kTileW - 1) / kTileW;
  int tx, ty;
  xcd_strip_tile(vb_, ntiles_, tiles_x, tx, ty);
  const int Xa = tx * kTileW, Ya = ty * kTileH, Xb = min(Xa + kTileW - 1, W - 1), Yb = min(Ya + kTileH - 1, H - 1);
  const int x = Xa + (threadIdx.x & (kTileW - 1)), y = Ya + (threadIdx.x >> 5);
  const size_t p = (size_t)min(y, H - 1) * W + min(x, W - 1);
  float own[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  int rx0 = 0, ry0 = 0;
  if constexpr (S1) {
#pragma unroll
    for (int c = 0; c < 5; ++c) own[c] = tmp[(size_t)c * P + p];  // issued before the terms' loads and the barrier
  } else {
    rx0 = lerp_src(Xa, inv_scale, w).i0, ry0 = lerp_src(Ya, inv_scale, h).i0;
    const int rw = lerp_src(Xb, inv_scale, w).i1 - rx0 + 1, rh = lerp_src(Yb, inv_scale, h).i1 - ry0 + 1;
    // footprint pixel e of the capacity grid per thread and round (compile-time divisors; the loop used to decompose a running
    // index with two runtime integer divisions per staged value)
#pragma unroll
    for (int k = 0; k < (10 * 36 + 255) / 256; ++k) {
      const int e = threadIdx.x + k * 256;
      const int r = e / 36, col = e - r * 36;
      if (r < rh && col < rw && r < 10) {
        const float *src = tmp + (size_t)(ry0 + r) * w + rx0 + col;
        float v[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) v[c] = src[(size_t)c * p_lo];
#pragma unroll
        for (int c = 0; c < 5; ++c) prev[r][col][c] = v[c];
      }
    }
  }
  if (LAZY) terms_stage<kWbTermR, kWbTermC, 256>(tl, T, items.term[vitem_], Xa, Ya, Xb, Yb, threadIdx.x, trx0, try0);
  if (LAZY || !S1) __syncthreads();
  if (x >= W || y >= H) return;
  const Lerp ly = lerp_src(y, inv_scale, h), lx = lerp_src(x, inv_scale, w);
  const int r0 = S1 ? 0 : ly.i0 - ry0, r1 = S1 ? 0 : ly.i1 - ry0, c0 = S1 ? 0 : lx.i0 - rx0, c1 = S1 ? 0 : lx.i1 - rx0;
  typedef float f32x4w __attribute__((ext_vector_type(4)));
  const f32x4w q00 = *reinterpret_cast<const f32x4w *>(&prev[r0][c0][0]), q01 = *reinterpret_cast<const f32x4w *>(&prev[r0][c1][0]);
  const f32x4w q10 = *reinterpret_cast<const f32x4w *>(&prev[r1][c0][0]), q11 = *reinterpret_cast<const f32x4w *>(&prev[r1][c1][0]);
  auto up = [&](int c) -> float {
    if constexpr (S1) return own[c];
    if (c < 4) return lerp2_fma(ly.w0, ly.w1, lx.w0, lx.w1, q00[c & 3], q01[c & 3], q10[c & 3], q11[c & 3]);
    return lerp2_fma(ly.w0, ly.w1, lx.w0, lx.w1, prev[r0][c0][4], prev[r0][c1][4], prev[r1][c0][4], prev[r1][c1][4]);
  };
  float fl[4];
  const bool have_terms = LAZY && terms_flow<kWbTermR, kWbTermC>(tl, T, trx0, try0, x, y, fl);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float fd = __fmul_rn(up(c), scale);  // ifblock_update's arithmetic: product and sum rounded separately
    if (LAZY) fl[c] = have_terms ? __fadd_rn(fl[c], fd) : fd;
    else fl[c] = flow ? __fadd_rn(flow[(size_t)c * P + p], fd) : fd;
  }
  const Taps t0 = taps_border(warp_coord(x, W, fl[0]), warp_coord(y, H, fl[1]), W, H);
  const Taps t1 = taps_border(warp_coord(x, W, fl[2]), warp_coord(y, H, fl[3]), W, H);
  const float mk = up(4);
  const float m = 1.f / (1.f + expf(-mk));
  if (img0x) {  // [H][W][4] frames: the two taps of a row are two 16-byte loads (4 + 4 gathers instead of 6 + 6 of 8 bytes)
    float a[3], b[3];
    sample_x4(img0x, W, t0, a);
    sample_x4(img1x, W, t1, b);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(size_t)c * P + p] = a[c] * m + b[c] * (1.f - m);
    return;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a = sample(img0 + (size_t)c * P, W, t0), b = sample(img1 + (size_t)c * P, W, t1);
    out[(size_t)c * P + p] = a * m + b * (1.f - m);
  }
}

// IFNet_HDv3.py:92-95 + :160: tmp [13,h,w] -> x`scale` bilinear; flow_out = flow_in + tmp[:4]*scale.
// mask / feat are written only when requested (the fused pipeline re-derives them from tmp on the fly).
__global__ void __launch_bounds__(256)
ifblock_update_kernel(const UpdateItems items, float *__restrict__ mask, float *__restrict__ feat, int h, int w, int H, int W,
                      float scale, float inv_scale) {
  const float *__restrict__ tmp = items.tmp[blockIdx.y];
  const float *flow_in = items.flow_in[blockIdx.y];
  float *flow_out = items.flow_out[blockIdx.y];
  const size_t P = (size_t)H * W, p_lo = (size_t)h * w;
  const int nch = (mask || feat) ? 13 : 4;
  const Tile2D tp_ = tile_pixel(W, H);
  if (tp_.valid) {
    const int y = tp_.y, x = tp_.x;
    const size_t p = (size_t)y * W + x;
    const Lerp ly = lerp_src(y, inv_scale, h), lx = lerp_src(x, inv_scale, w);
    const size_t o00 = (size_t)ly.i0 * w + lx.i0, o01 = (size_t)ly.i0 * w + lx.i1;
    const size_t o10 = (size_t)ly.i1 * w + lx.i0, o11 = (size_t)ly.i1 * w + lx.i1;
    for (int c = 0; c < nch; ++c) {
      const float *t = tmp + (size_t)c * p_lo;
      const float top = lx.w0 * t[o00] + lx.w1 * t[o01];
      const float bot = lx.w0 * t[o10] + lx.w1 * t[o11];
      const float v = ly.w0 * top + ly.w1 * bot;
      if (c < 4) {
        const float fd = v * scale;
        flow_out[(size_t)c * P + p] = flow_in ? flow_in[(size_t)c * P + p] + fd : fd;
      } else if (c == 4) {
        if (mask) mask[p] = v;
      } else if (feat) {
        feat[(size_t)(c - 5) * P + p] = v;
      }
    }
  }
}

// IFNet_HDv3.py:163-167: warped_img0*sigmoid(mask) + warped_img1*(1-sigmoid(mask)); mask = x`scale`
// upsample of the last head output's channel 4 (mask_lo, [h, w]).
__global__ void __launch_bounds__(256)
warp_blend_kernel(const float *__restrict__ img0, const float *__restrict__ img1, const float *__restrict__ flow,
                  const float *__restrict__ mask_lo, int h, int w, float inv_scale, float *__restrict__ out, int H,
                  int W) {
  const size_t P = (size_t)H * W;
  const Tile2D tp_ = tile_pixel(W, H);
  if (tp_.valid) {
    const int y = tp_.y, x = tp_.x;
    const size_t p = (size_t)y * W + x;
    const Taps t0 = taps_border(warp_coord(x, W, flow[p]), warp_coord(y, H, flow[P + p]), W, H);
    const Taps t1 = taps_border(warp_coord(x, W, flow[2 * P + p]), warp_coord(y, H, flow[3 * P + p]), W, H);
    const Lerp ly = lerp_src(y, inv_scale, h), lx = lerp_src(x, inv_scale, w);
    const float top = lx.w0 * mask_lo[(size_t)ly.i0 * w + lx.i0] + lx.w1 * mask_lo[(size_t)ly.i0 * w + lx.i1];
    const float bot = lx.w0 * mask_lo[(size_t)ly.i1 * w + lx.i0] + lx.w1 * mask_lo[(size_t)ly.i1 * w + lx.i1];
    const float mk = ly.w0 * top + ly.w1 * bot;
    const float m = 1.f / (1.f + expf(-mk));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a = sample(img0 + (size_t)c * P, W, t0), b = sample(img1 + (size_t)c * P, W, t1);
      out[(size_t)c * P + p] = a * m + b * (1.f - m);
    }
  }
}

}  // namespace

extern "C" {

int drba_resize_bilinear(const float *in, float *out, int NC, int Hin, int Win, int Hout, int Wout, float scale_y,
                         float scale_x, void *stream) {
  if (!in || !out || NC <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return DRBA_EINVAL;
  DRBA_LAUNCH(resize_bilinear_kernel, dim3(grid_for((size_t)NC * Hout * Wout)), dim3(kBlock), 0,
                     (hipStream_t)stream, in, out, NC, Hin, Win, Hout, Wout, scale_y, scale_x);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_pair_interleave(const float *in, float *out, int C, int H, int W, void *stream) {
  if (!in || !out || C <= 0 || (C & 1) || H <= 0 || W <= 0) return DRBA_EINVAL;
  const size_t P = (size_t)H * W;
  DRBA_LAUNCH(pair_interleave_kernel, dim3(grid_for((size_t)(C / 2) * P)), dim3(kBlock), 0, (hipStream_t)stream, in,
                     out, C / 2, P);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_rgbx(const float *img, float *out, int H, int W, void *stream) {
  if (!img || !out || H <= 0 || W <= 0 || ((uintptr_t)out & 15) != 0) return DRBA_EINVAL;
  const size_t P = (size_t)H * W;
  DRBA_LAUNCH(rgbx_kernel, dim3(grid_for(P)), dim3(kBlock), 0, (hipStream_t)stream, img, out, P);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_u8hwc_to_f32nchw(const uint8_t *in, float *out, int H, int W, void *stream) {
  if (!in || !out || H <= 0 || W <= 0) return DRBA_EINVAL;
  DRBA_LAUNCH(u8_to_f32_kernel, dim3(grid_for((size_t)H * W)), dim3(kBlock), 0, (hipStream_t)stream, in, out, H, W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_f32nchw_to_u8hwc(const float *in, uint8_t *out, int H, int W, void *stream) {
  if (!in || !out || H <= 0 || W <= 0) return DRBA_EINVAL;
  DRBA_LAUNCH(f32_to_u8_kernel, dim3(grid_for((size_t)H * W)), dim3(kBlock), 0, (hipStream_t)stream, in, out, H, W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_to_inp_x4(const uint8_t *img_hwc, float *out, float *out_x4, int Hin, int Win, int Hout, int Wout, float scale_y, float scale_x,
                   void *stream) {
  if (!img_hwc || !out || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || ((uintptr_t)out_x4 & 15) != 0) return DRBA_EINVAL;
  if (Win == Wout && scale_x == 1.f && (Wout & 3) == 0 && (((uintptr_t)img_hwc & 3) | ((uintptr_t)out & 15)) == 0) {
    DRBA_LAUNCH(to_inp_rows_kernel, dim3(grid_for((size_t)(Wout >> 2) * Hout)), dim3(kBlock), 0, (hipStream_t)stream, img_hwc, out, out_x4,
                Hin, Wout, Hout, scale_y);
    DRBA_CHECK_LAUNCH();
    return DRBA_OK;
  }
  DRBA_LAUNCH(to_inp_kernel, dim3(tiles_for(Wout, Hout)), dim3(kBlock), 0, (hipStream_t)stream, img_hwc, out, out_x4, Hin, Win, Hout,
              Wout, scale_y, scale_x);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_to_inp(const uint8_t *img_hwc, float *out, int Hin, int Win, int Hout, int Wout, float scale_y, float scale_x,
                void *stream) {
  return drba_to_inp_x4(img_hwc, out, nullptr, Hin, Win, Hout, Wout, scale_y, scale_x, stream);
}

int drba_to_out(const float *in, uint8_t *out_hwc, int Hin, int Win, int Hout, int Wout, float scale_y, float scale_x,
                int reverse_channels, void *stream) {
  if (!in || !out_hwc || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return DRBA_EINVAL;
  if (Win == Wout && scale_x == 1.f && (Wout & 3) == 0 && (((uintptr_t)out_hwc & 3) | ((uintptr_t)in & 15)) == 0) {
    DRBA_LAUNCH(to_out_rows_kernel, dim3(grid_for((size_t)(Wout >> 2) * Hout)), dim3(kBlock), 0, (hipStream_t)stream, in, out_hwc, Hin,
                Wout, Hout, scale_y, reverse_channels);
    DRBA_CHECK_LAUNCH();
    return DRBA_OK;
  }
  DRBA_LAUNCH(to_out_kernel, dim3(tiles_for(Wout, Hout)), dim3(kBlock), 0, (hipStream_t)stream, in, out_hwc, Hin, Win, Hout,
              Wout, scale_y, scale_x, reverse_channels);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

static int stage_items_ok(const drba_stage_item_t *items, int n, bool lds) {
  if (!items || n <= 0 || n > kMaxItems) return DRBA_EINVAL;
  for (int k = 0; k < n; ++k) {
    const drba_stage_item_t &I = items[k];
    if (!I.img0 || !I.img1 || !I.out) return DRBA_EINVAL;
    if ((I.f0_pair == nullptr) != (I.f1_pair == nullptr) || (I.f0 == nullptr) != (I.f1 == nullptr)) return DRBA_EINVAL;
    if (!I.f0 && !I.f0_pair) return DRBA_EINVAL;  // the features in one layout at least: planar [16,H,W] or pair-interleaved [8,H,W,2]
    if ((I.img0_x4 != nullptr) != (I.img1_x4 != nullptr) || (((uintptr_t)I.img0_x4 | (uintptr_t)I.img1_x4) & 15) != 0) return DRBA_EINVAL;
    // one kernel instantiation serves the batch: the items agree on what is optional
    if ((I.flow == nullptr) != (items[0].flow == nullptr) || (I.flow_out == nullptr) != (items[0].flow_out == nullptr) ||
        (I.f0_pair == nullptr) != (items[0].f0_pair == nullptr) || (I.f0 == nullptr) != (items[0].f0 == nullptr) ||
        (I.tmp_prev == nullptr) != (items[0].tmp_prev == nullptr))
      return DRBA_EINVAL;
    if (lds && !I.tmp_prev) return DRBA_EINVAL;
    if (lds && !I.flow_out && !I.flow) return DRBA_EINVAL;  // without the fold the finished flow must be given
  }
  return DRBA_OK;
}

int drba_ifblock_input_batch(const drba_stage_item_t *items, int n_items, int hp, int wp, float prev_scale, int H, int W, int h,
                             int w, float scale, void *stream) {
  const int rc = stage_items_ok(items, n_items, false);
  if (rc != DRBA_OK) return rc;
  if (H <= 1 || W <= 1 || h <= 0 || w <= 0 || !(scale > 0.f)) return DRBA_EINVAL;
  const bool has_flow = items[0].flow != nullptr;
  if (has_flow && (!items[0].tmp_prev || hp <= 0 || wp <= 0 || !(prev_scale > 0.f))) return DRBA_EINVAL;
  StageItems its;
  memset(&its, 0, sizeof(its));
  for (int k = 0; k < n_items; ++k) its.it[k] = items[k];
  const bool single = scale == 1.f;
  hipStream_t s = (hipStream_t)stream;
  const float ips = has_flow ? (float)(1.0 / (double)prev_scale) : 1.f;
  // variant: 0 = lane per output pixel; 1 = lane per sample point; 2 = lane per sample point, channel loops unrolled x4.
  // DRBA_IFIN_VARIANT overrides the default (A/B experiments, TUNING builds only).
  static const int forced = env_int("DRBA_IFIN_VARIANT", -1);
  const int var = forced >= 0 ? forced : (single ? 0 : 1);
  dim3 b(kBlock);
  const int quad_tiles = single ? tiles_for(w, h) : ((w + 15) / 16) * ((h + 3) / 4);
#define DRBA_ARGS its, hp, wp, ips, H, W, h, w, scale
#define DRBA_IFIN(HF, SG)                                                                                          \
  do {                                                                                                             \
    if (var == 0) {                                                                                                \
      DRBA_LAUNCH((ifblock_input_pixel<HF, SG>), dim3(tiles_for(w, h), n_items), b, 0, s, DRBA_ARGS);              \
    } else if (var == 1) {                                                                                         \
      DRBA_LAUNCH((ifblock_input_kernel<HF, SG, 1>), dim3(quad_tiles, n_items), b, 0, s, DRBA_ARGS);               \
    } else {                                                                                                       \
      DRBA_LAUNCH((ifblock_input_kernel<HF, SG, 4>), dim3(quad_tiles, n_items), b, 0, s, DRBA_ARGS);               \
    }                                                                                                              \
  } while (0)
  if (has_flow) {
    if (single) DRBA_IFIN(true, true);
    else DRBA_IFIN(true, false);
  } else {
    if (single) DRBA_IFIN(false, true);
    else DRBA_IFIN(false, false);
  }
#undef DRBA_ARGS
#undef DRBA_IFIN
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_ifblock_input(const float *img0, const float *img1, const float *f0, const float *f1, const float *f0_pair,
                       const float *f1_pair, const float *timestep_map, float timestep_scalar, const float *flow,
                       const float *tmp_prev,
                       int hp, int wp, float prev_scale, float *out, int H, int W, int h, int w, float scale,
                       void *stream) {
  const drba_stage_item_t it = {img0, img1, f0, f1, f0_pair, f1_pair, timestep_map, timestep_scalar, flow, tmp_prev, nullptr, out};
  return drba_ifblock_input_batch(&it, 1, hp, wp, prev_scale, H, W, h, w, scale, stream);
}

// mode 0 / 1 (flow given / folded update written, chosen by items[0].flow_out) or 2 (lazy: the flow is `terms` + the fold)
static int ifblock_input_lds_launch(const drba_stage_item_t *items, int n_items, const drba_flow_terms_t *terms, int hp, int wp,
                                    float prev_scale, int H, int W, int h, int w, float scale, void *stream) {
  const bool lazy = terms != nullptr;
  if (!items || n_items <= 0 || n_items > kMaxItems) return DRBA_EINVAL;
  if (lazy) {
    for (int k = 0; k < n_items; ++k) {
      const drba_stage_item_t &I = items[k];
      if (!I.img0 || !I.img1 || !I.out || !I.tmp_prev || I.flow || I.flow_out) return DRBA_EINVAL;
      if ((I.f0_pair == nullptr) != (I.f1_pair == nullptr) || (I.f0_pair == nullptr) != (items[0].f0_pair == nullptr)) return DRBA_EINVAL;
      if ((I.f0 == nullptr) != (I.f1 == nullptr) || (!I.f0 && !I.f0_pair)) return DRBA_EINVAL;  // planar or pair-interleaved features
      for (int i = 0; i < terms->n && i < kMaxTerms; ++i)
        if (!I.term[i]) return DRBA_EINVAL;
    }
  } else {
    const int rc = stage_items_ok(items, n_items, true);
    if (rc != DRBA_OK) return rc;
  }
  if (H <= 1 || W <= 1 || h <= 0 || w <= 0 || !(scale > 0.f)) return DRBA_EINVAL;
  if (hp <= 0 || wp <= 0 || !(prev_scale > 0.f)) return DRBA_EINVAL;
  const bool fold = !lazy && items[0].flow_out != nullptr;
  if (fold && scale > 2.f) return DRBA_EUNSUPPORTED;   // the fold WRITES the flow: every full-resolution pixel must be sampled once
  if (scale != 1.f && scale != 2.f && scale != 4.f && scale != 8.f && scale != 16.f && scale != 32.f) return DRBA_EUNSUPPORTED;
  if (prev_scale != 2.f * scale) return DRBA_EUNSUPPORTED;  // IFNet's pyramid; bounds the staged footprint to 18 x 6 pixels
  FlowTermsArg T;
  if (!flow_terms_arg(terms, T)) return DRBA_EINVAL;
  for (int i = 0; i < T.n; ++i)
    if (T.scale[i] < 2.f * prev_scale) return DRBA_EUNSUPPORTED;  // earlier stages of the pyramid only (bounds their footprints)
  StageItems its;
  memset(&its, 0, sizeof(its));
  for (int k = 0; k < n_items; ++k) its.it[k] = items[k];
  const bool single = scale == 1.f;
  hipStream_t s = (hipStream_t)stream;
  const float ips = (float)(1.0 / (double)prev_scale);
  const int tiles = single ? tiles_for(w, h) : ((w + 15) / 16) * ((h + 3) / 4);
  // vector-store form: whole tiles only, 16-byte aligned rows and planes (DRBA_IFIN_VS=0 forces the scalar stores: A/B runs, TUNING builds)
  static const bool vs_allowed = env_int("DRBA_IFIN_VS", 1) != 0;
  const int two = single ? 32 : 16, tho = single ? 8 : 4;
  bool vs = vs_allowed && w % two == 0 && h % tho == 0 && (W & 3) == 0;
  for (int k = 0; k < n_items; ++k)
    vs = vs && ((uintptr_t)items[k].out & 15) == 0 &&
         (!fold || (((uintptr_t)items[k].flow_out & 15) == 0 && (single || (W == 2 * w && H == 2 * h))));
#define DRBA_IFL(SG, FM, VS_) \
  DRBA_LAUNCH((ifblock_input_lds<SG, FM, VS_>), dim3(tiles * n_items), dim3(kBlock), 0, s, its, T, hp, wp, ips, prev_scale, H, W, h, w, scale, n_items)
#define DRBA_IFL2(SG, FM) \
  do {                    \
    if (vs) DRBA_IFL(SG, FM, true); \
    else DRBA_IFL(SG, FM, false);   \
  } while (0)
  if (lazy) {
    if (single) DRBA_IFL2(true, 2);
    else DRBA_IFL2(false, 2);
  } else if (fold) {
    if (single) DRBA_IFL2(true, 1);
    else DRBA_IFL2(false, 1);
  } else {
    if (single) DRBA_IFL2(true, 0);
    else DRBA_IFL2(false, 0);
  }
#undef DRBA_IFL2
#undef DRBA_IFL
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_ifblock_input_lds_batch(const drba_stage_item_t *items, int n_items, int hp, int wp, float prev_scale, int H, int W,
                                 int h, int w, float scale, void *stream) {
  return ifblock_input_lds_launch(items, n_items, nullptr, hp, wp, prev_scale, H, W, h, w, scale, stream);
}

int drba_ifblock_input_lazy_batch(const drba_stage_item_t *items, int n_items, const drba_flow_terms_t *terms, int hp, int wp,
                                  float prev_scale, int H, int W, int h, int w, float scale, void *stream) {
  if (!terms) return DRBA_EINVAL;
  return ifblock_input_lds_launch(items, n_items, terms, hp, wp, prev_scale, H, W, h, w, scale, stream);
}

int drba_ifblock_input_lds(const float *img0, const float *img1, const float *f0, const float *f1, const float *f0_pair,
                           const float *f1_pair, const float *timestep_map, float timestep_scalar, const float *flow,
                           const float *tmp_prev, int hp, int wp, float prev_scale, float *flow_out, float *out, int H,
                           int W, int h, int w, float scale, void *stream) {
  const drba_stage_item_t it = {img0, img1, f0, f1, f0_pair, f1_pair, timestep_map, timestep_scalar, flow, tmp_prev, flow_out, out};
  return drba_ifblock_input_lds_batch(&it, 1, hp, wp, prev_scale, H, W, h, w, scale, stream);
}

int drba_warp_blend_fold(const float *img0, const float *img1, const float *flow, const float *tmp_last, int h, int w,
                         float scale, float *out, int H, int W, void *stream) {
  if (!img0 || !img1 || !tmp_last || !out || H <= 1 || W <= 1 || h <= 0 || w <= 0 || !(scale >= 1.f)) return DRBA_EINVAL;
  BlendItems its;
  memset(&its, 0, sizeof(its));
  its.img0[0] = img0, its.img1[0] = img1, its.flow[0] = flow, its.tmp[0] = tmp_last, its.out[0] = out;
  FlowTermsArg T;
  flow_terms_arg(nullptr, T);
  DRBA_LAUNCH((warp_blend_fold_kernel<false>), dim3(tiles_for(W, H)), dim3(kBlock), 0, (hipStream_t)stream, its, T, h, w,
              (float)(1.0 / (double)scale), scale, H, W, 1);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_warp_blend_lazy_batch(const drba_stage_item_t *items, int n_items, const drba_flow_terms_t *terms, int h, int w, float scale,
                               int H, int W, void *stream) {
  if (!items || !terms || n_items <= 0 || n_items > kMaxItems || H <= 1 || W <= 1 || h <= 0 || w <= 0 || !(scale >= 1.f)) return DRBA_EINVAL;
  FlowTermsArg T;
  if (!flow_terms_arg(terms, T)) return DRBA_EINVAL;
  for (int i = 0; i < T.n; ++i)
    if (T.scale[i] < 2.f * scale || T.scale[i] < 2.f) return DRBA_EUNSUPPORTED;  // earlier stages of the pyramid (bounds their footprints)
  BlendItems its;
  memset(&its, 0, sizeof(its));
  for (int k = 0; k < n_items; ++k) {
    const drba_stage_item_t &I = items[k];
    if (!I.img0 || !I.img1 || !I.tmp_prev || !I.out) return DRBA_EINVAL;
    if ((I.img0_x4 != nullptr) != (I.img1_x4 != nullptr) || (((uintptr_t)I.img0_x4 | (uintptr_t)I.img1_x4) & 15) != 0) return DRBA_EINVAL;
    its.img0[k] = I.img0, its.img1[k] = I.img1, its.tmp[k] = I.tmp_prev, its.out[k] = I.out;
    its.img0x[k] = I.img0_x4, its.img1x[k] = I.img1_x4;
    for (int i = 0; i < T.n; ++i) {
      if (!I.term[i]) return DRBA_EINVAL;
      its.term[k][i] = I.term[i];
    }
  }
  if (scale == 1.f && h == H && w == W)  // the last stage at the frame's own resolution: no footprint staging (kernel comment "S1")
    DRBA_LAUNCH((warp_blend_fold_kernel<true, true>), dim3(tiles_for(W, H) * n_items), dim3(kBlock), 0, (hipStream_t)stream, its, T, h, w,
                1.f, scale, H, W, n_items);
  else
    DRBA_LAUNCH((warp_blend_fold_kernel<true>), dim3(tiles_for(W, H) * n_items), dim3(kBlock), 0, (hipStream_t)stream, its, T, h, w,
                (float)(1.0 / (double)scale), scale, H, W, n_items);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_ifblock_update_batch(const float *const *tmp, const float *const *flow_in, float *const *flow_out, int n_items, int h,
                              int w, int H, int W, float scale, void *stream) {
  if (!tmp || !flow_out || n_items <= 0 || n_items > kMaxItems || h <= 0 || w <= 0 || H <= 0 || W <= 0 || !(scale > 0.f))
    return DRBA_EINVAL;
  UpdateItems its;
  memset(&its, 0, sizeof(its));
  for (int k = 0; k < n_items; ++k) {
    if (!tmp[k] || !flow_out[k]) return DRBA_EINVAL;
    its.tmp[k] = tmp[k], its.flow_in[k] = flow_in ? flow_in[k] : nullptr, its.flow_out[k] = flow_out[k];
  }
  DRBA_LAUNCH(ifblock_update_kernel, dim3(tiles_for(W, H), n_items), dim3(kBlock), 0, (hipStream_t)stream, its, (float *)nullptr,
              (float *)nullptr, h, w, H, W, scale, (float)(1.0 / (double)scale));
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_ifblock_update(const float *tmp, const float *flow_in, float *flow_out, float *mask, float *feat, int h,
                        int w, int H, int W, float scale, void *stream) {
  if (!tmp || !flow_out || h <= 0 || w <= 0 || H <= 0 || W <= 0 || !(scale > 0.f)) return DRBA_EINVAL;
  UpdateItems its;
  memset(&its, 0, sizeof(its));
  its.tmp[0] = tmp, its.flow_in[0] = flow_in, its.flow_out[0] = flow_out;
  DRBA_LAUNCH(ifblock_update_kernel, dim3(tiles_for(W, H)), dim3(kBlock), 0, (hipStream_t)stream, its, mask, feat, h, w, H, W,
              scale, (float)(1.0 / (double)scale));
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

int drba_warp_blend(const float *img0, const float *img1, const float *flow, const float *mask_lo, int h, int w,
                    float scale, float *out, int H, int W, void *stream) {
  if (!img0 || !img1 || !flow || !mask_lo || !out || H <= 1 || W <= 1 || h <= 0 || w <= 0 || !(scale > 0.f))
    return DRBA_EINVAL;
  DRBA_LAUNCH(warp_blend_kernel, dim3(tiles_for(W, H)), dim3(kBlock), 0, (hipStream_t)stream, img0, img1,
                     flow, mask_lo, h, w, (float)(1.0 / (double)scale), out, H, W);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // extern "C"
