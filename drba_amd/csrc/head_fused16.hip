// IFNet's context encoder (IFNet_HDv3.py:23-47, `Head`) in one kernel, two-term fp16 form (kernel family 4; head_fused.hip is the
// fp32-MFMA form of the same fusion: same tile, same rings, same buffers, same outputs).
//
// What changes is the arithmetic and, for it, the layout of the intermediates in LDS.  Every fp32 operand is taken as
// h + 2^-11 l with two fp16 terms (conv_split.hip "Two-term form": x' = x / 16, h = fp16(x'), l = fp16((x' - h) * 2^11)), and a
// layer is evaluated with v_mfma_f32_16x16x32_f16: K = 32 = two taps x 16 input channels, three products per K step (h h into one
// accumulator, h l + l h into a second one, joined as acc + acl / 2048).  A 16 -> 16 layer is 5 K steps = 15 MFMAs of ~17 clocks
// per 16-position tile where the fp32 form issues 36 MFMAs of 32 clocks; the transposed convolution 4 phases x 2 steps x 3 = 24
// against 64; cnn0 (K = 27 padded to 32) 3 against 7.
//   * roles: the WEIGHTS are the MFMA's A operand (rows = output channels), the activations B (columns = positions), so that a
//     lane leaves a tile holding 4 consecutive output channels of ONE position -- which is what the next layer's operand
//     layout wants to be written with (below), and what the pair-interleaved output wants to be stored with;
//   * intermediates X0 / X1 / X2: [plane h, l][channel half 0-7, 8-15][position of the 14 x 40 region][8 x fp16] -- a lane's B
//     operand for K step j (tap 2j + (kq >> 1), channels 8 (kq & 1) .. + 7) is ONE 16-byte read at a per-lane constant offset
//     from the tile's first position, the 16 lanes of a channel group read 256 contiguous bytes; a tile's result is one 8-byte
//     write per plane.  2 + 2 bytes per element: the buffers are as large as the fp32 form's (two workgroups per CU);
//   * the frame window stays fp32 in LDS (cnn0 gathers its 8 taps per lane with scalar reads and splits them in registers);
//   * weights: split on the host, fragment order, from L2 into registers per layer (10 / 16 x 16 bytes per lane).
// Positions outside the half-resolution map are stored as zero (the next layer's padding), per lane.
#include "common.hpp"
#include "conv_split.hpp"

#include <string.h>

using namespace drba;

namespace drba_head16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

[[maybe_unused]] constexpr int C = 16;  // feature channels
constexpr int H2 = 8, W2 = 32;              // half-resolution positions per workgroup (output tile 16 x 64)
constexpr int RR = H2 + 6, RC = W2 + 6;     // region: 14 x 38 positions (cnn0's outputs)
constexpr int RS = 40;                      // region row stride
constexpr int NPOS = RR * RS;               // 560 positions per (plane, channel half)
[[maybe_unused]] constexpr int IR = 2 * RR + 1, IC = 2 * RC + 1;  // frame window: 29 x 77
constexpr int IRS = 80, ICS = IR * IRS;     // its row / channel stride (floats)
constexpr int GP = 48;                      // guard positions in front of / behind a buffer (taps of discarded edge positions reach 41)
constexpr int XU = 4 * NPOS;                // 16-byte units of one intermediate: [plane][half][position]
constexpr int BUF_U = XU + 2 * GP;          // 2336 units = 37376 bytes >= the frame window's 3 * ICS * 4 = 27840
constexpr int THREADS = 512;
constexpr float kScale = 1.f / (float)(1 << kSplitActShift), kUnscale = (float)(1 << kSplitActShift);
// packed weights (16-byte units): cnn0 [plane 2][64] | cnn1, cnn2 [step 5][plane 2][64] | deconv [phase 4][step 2][plane 2][64];
// then the biases as floats [4][16]
constexpr int U0 = 2 * 64, U1 = 10 * 64, U3 = 16 * 64;
constexpr int OFF_U1 = U0, OFF_U2 = U0 + U1, OFF_U3 = U0 + 2 * U1, OFF_UB = U0 + 2 * U1 + U3;
constexpr int W_FLOATS = OFF_UB * 4 + 64;
constexpr int LDS_BYTES = 2 * BUF_U * 16;   // 74752: two workgroups per CU

static_assert(3 * ICS * 4 <= BUF_U * 16, "frame window fits the buffer it shares");

// (a, b) * 2^-shift -> packed h and packed (remainder * 2^11)
__device__ __forceinline__ void split2(float a, float b, unsigned &h, unsigned &l) {
  const f32x2 v = (f32x2){a, b} * kScale;
  const f16x2 hh = __builtin_convertvector(v, f16x2);
  const f32x2 r = (v - __builtin_convertvector(hh, f32x2)) * 2048.f;
  const f16x2 ll = __builtin_convertvector(r, f16x2);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, ll);
}

struct Acc2 {
  f32x4 hi, lo;
};
__device__ __forceinline__ void mma3(Acc2 &c, const u32x4 &wh, const u32x4 &wl, const u32x4 &xh, const u32x4 &xl) {
  const f16x8 a_h = __builtin_bit_cast(f16x8, wh), a_l = __builtin_bit_cast(f16x8, wl);
  const f16x8 b_h = __builtin_bit_cast(f16x8, xh), b_l = __builtin_bit_cast(f16x8, xl);
  c.lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l, b_h, c.lo, 0, 0, 0);
  c.lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, b_l, c.lo, 0, 0, 0);
  c.hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, b_h, c.hi, 0, 0, 0);
}
__device__ __forceinline__ f32x4 join(const Acc2 &c) { return (c.hi + c.lo * (1.f / 2048.f)) * kUnscale; }

__global__ void __launch_bounds__(THREADS, 4)  // 4 waves per SIMD = two workgroups per CU: <= 128 registers
head_fused16(const float *__restrict__ img, const u32x4 *__restrict__ wpk, float *__restrict__ f_out, float *__restrict__ fp_out, int H,
             int W, int tiles_x, unsigned char *status) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) u32x4 lds16[];
  u32x4 *bufA = lds16 + GP;                       // IN (fp32), then X1
  u32x4 *bufB = lds16 + BUF_U + GP;               // X0, then X2
  float *inw = reinterpret_cast<float *>(lds16 + GP);
  const float *biases = reinterpret_cast<const float *>(wpk + OFF_UB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int Hh = H >> 1, Wh = W >> 1;
  const size_t P = (size_t)H * W;
  int tx, ty;
  xcd_strip_tile(blockIdx.x, gridDim.x, tiles_x, tx, ty);
  const int m0 = ty * H2, n0 = tx * W2;
  const int rm = m0 - 3, rn = n0 - 3;             // region origin (half-resolution coordinates)
  img += (size_t)blockIdx.y * 3 * P;
  if (f_out) f_out += (size_t)blockIdx.y * C * P;
  fp_out += (size_t)blockIdx.y * C * P;

  // ---- the frame window (zero outside the image), as in head_fused.hip
  {
    const int y0 = 2 * rm - 1, x0 = 2 * rn - 1;
    constexpr int NI = (3 * IR * IRS + THREADS - 1) / THREADS;
    float v[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int i = min(tid + k * THREADS, 3 * IR * IRS - 1);
      const int c = i / (IR * IRS), r = (i - c * IR * IRS) / IRS, col = i - c * IR * IRS - r * IRS;
      const int y = y0 + r, x = x0 + col;
      const bool ok = col < IC && y >= 0 && y < H && x >= 0 && x < W;
      const float t = img[(size_t)c * P + (size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)];
      v[k] = ok ? t : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (tid + k * THREADS < 3 * IR * IRS) inw[tid + k * THREADS] = v[k];
  }
  __syncthreads();

  // A tile's result -> the next layer's operand planes: lane (position q, channels 4 kq .. + 3): bias, LeakyReLU, zero outside
  // the map, split, one 8-byte write per plane at [plane][half kq >> 1][q] + 8 (kq & 1) bytes
  auto xstore = [&](u32x4 *X, int q, const Acc2 &c, const f32x4 &bias, int ncol) {
    const int r = q / RS, col = q - r * RS;
    const bool ok = (unsigned)(rm + r) < (unsigned)Hh && col < ncol && (unsigned)(rn + col) < (unsigned)Wh;
    f32x4 v = join(c) + bias;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ok ? lrelu02(v[i]) : 0.f;
    u32x2 h, l;
    unsigned a, b;
    split2(v[0], v[1], a, b);
    h[0] = a, l[0] = b;
    split2(v[2], v[3], a, b);
    h[1] = a, l[1] = b;
    unsigned char *base = reinterpret_cast<unsigned char *>(X + (kq >> 1) * NPOS + q) + 8 * (kq & 1);
    *reinterpret_cast<u32x2 *>(base) = h;
    *reinterpret_cast<u32x2 *>(base + 2 * NPOS * 16) = l;
  };

  // ---- cnn0: 3 -> 16, stride 2: K = 27 (channel, tap) padded to 32, one K step; lane (position m, kq) gathers k = 8 kq .. + 7
  {
    const u32x4 wh = wpk[lane], wl = wpk[64 + lane];
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(biases + 4 * kq);
    int off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = min(8 * kq + i, 26), ci = k / 9, t = k - ci * 9;  // (k >= 27: any valid address, its weights are zero)
      off[i] = ci * ICS + (t / 3) * IRS + (t % 3);
    }
    for (int t = wave; t < RR * 3; t += 8) {       // 16-position tiles, three per region row (the last one runs over the row's end)
      const int r = t / 3, c0 = (t - r * 3) * 16;
      const float *a = inw + 2 * r * IRS + 2 * (c0 + m);
      u32x4 xh, xl;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned h, l;
        split2(a[off[2 * i]], a[off[2 * i + 1]], h, l);
        xh[i] = h, xl[i] = l;
      }
      Acc2 c = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
      mma3(c, wh, wl, xh, xl);
      if (c0 + m < RS) xstore(bufB, r * RS + c0 + m, c, bias, RC);
    }
  }
  __syncthreads();

  // ---- cnn1 / cnn2: 16 -> 16 on the flattened region (tile = 16 consecutive positions, tap = constant offset)
  auto conv16 = [&](const u32x4 *src, u32x4 *dst, int woff, int layer, int q_lo, int q_hi) {
    u32x4 wh[5], wl[5];
    int off[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      wh[j] = wpk[woff + (2 * j) * 64 + lane], wl[j] = wpk[woff + (2 * j + 1) * 64 + lane];
      const int tap = min(2 * j + (kq >> 1), 8);   // (tap 9: zero weights, any valid address)
      off[j] = (kq & 1) * NPOS + (tap / 3 - 1) * RS + (tap % 3 - 1) + m;
    }
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(biases + 16 * layer + 4 * kq);
    // two tiles per pass: independent accumulator chains
    for (int q0 = q_lo + 16 * wave; q0 < q_hi; q0 += 2 * 16 * 8) {
      const int q1 = q0 + 16 * 8;
      const bool two = q1 < q_hi;
      const u32x4 *a0 = src + q0, *a1 = src + (two ? q1 : q0);
      Acc2 c0 = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}}, c1 = c0;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        mma3(c0, wh[j], wl[j], a0[off[j]], a0[off[j] + 2 * NPOS]);
        mma3(c1, wh[j], wl[j], a1[off[j]], a1[off[j] + 2 * NPOS]);
      }
      xstore(dst, q0 + m, c0, bias, RS);
      if (two) xstore(dst, q1 + m, c1, bias, RS);
    }
  };
  conv16(bufB, bufA, OFF_U1, 1, 1 * RS, (RR - 1) * RS);   // X0 -> X1: region rows 1..12
  __syncthreads();
  conv16(bufA, bufB, OFF_U2, 2, 2 * RS, (RR - 2) * RS);   // X1 -> X2: region rows 2..11
  __syncthreads();

  // ---- ConvTranspose2d(16, 16, 4, 2, 1): out(2M + py, 2N + px) = sum over 2 x 2 taps j of X2 (head_fused.hip's phase algebra);
  // K step s of a phase holds taps 2 s + (kq >> 1)
  {
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(biases + 48 + 4 * kq);
    float nf = 0.f;  // the family's overflow report (common.hpp): an overflow of any layer's fp16 planes reaches these values
    for (int u = wave; u < H2 * 2; u += 8) {      // (region row, 16-position half) units
      const int r = 3 + (u >> 1), c0 = 3 + (u & 1) * 16;
      const int M = rm + r, N = rn + c0 + m;       // this lane: position (M, N), output channels 4 kq .. + 3
      const bool inside = M < Hh && N < Wh;
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        Acc2 acc[2];
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          acc[px].hi = acc[px].lo = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const int j = 2 * s + (kq >> 1);
            const int dr = (j >> 1) == 0 ? (py ? 1 : 0) : (py ? 0 : -1);
            const int dc = (j & 1) == 0 ? (px ? 1 : 0) : (px ? 0 : -1);
            const u32x4 *a = bufB + (kq & 1) * NPOS + (r + dr) * RS + c0 + m + dc;
            const int wu = OFF_U3 + (((py * 2 + px) * 2 + s) * 2) * 64 + lane;
            mma3(acc[px], wpk[wu], wpk[wu + 64], a[0], a[2 * NPOS]);
          }
        }
        const f32x4 v0 = join(acc[0]) + bias, v1 = join(acc[1]) + bias;  // columns 2N, 2N + 1 of row 2M + py
        if (inside) {
#pragma unroll
          for (int e = 0; e < 4; ++e) nf = nf_fold(nf_fold(nf, v0[e]), v1[e]);
          const int y = 2 * M + py, x = 2 * N;
          // pair layout [C/2][H][W][2]: channel pairs 2 kq, 2 kq + 1; (x, x + 1) x (even, odd channel) = 16 contiguous bytes
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float *pp = fp_out + ((size_t)(2 * kq + e) * H + y) * (size_t)W * 2 + (size_t)x * 2;
            *reinterpret_cast<f32x4 *>(pp) = (f32x4){v0[2 * e], v0[2 * e + 1], v1[2 * e], v1[2 * e + 1]};
          }
          if (f_out) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              *reinterpret_cast<f32x2 *>(f_out + (size_t)(4 * kq + e) * P + (size_t)y * W + x) = (f32x2){v0[e], v1[e]};
          }
        }
      }
    }
    nf_report(status, DRBA_STATUS_HEAD, nf);
  }
#endif
}

}  // namespace drba_head16

extern "C" {

size_t drba_head_fused16_packed_floats(void) { return (size_t)drba_head16::W_FLOATS; }

/* HOST: as drba_head_fused_pack; every weight as two fp16 terms (split_weight_terms), fragment order of the kernel above */
int drba_head_fused16_pack(const float *w0, const float *b0, const float *w1, const float *b1, const float *w2, const float *b2,
                           const float *w3, const float *b3, float *packed) {
  using namespace drba_head16;
  if (!w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !packed) return DRBA_EINVAL;
  if (!two_term_weights_ok(w0, 16 * 27) || !two_term_weights_ok(w1, 16 * 144) || !two_term_weights_ok(w2, 16 * 144) ||
      !two_term_weights_ok(w3, 16 * 256))
    return DRBA_EUNSUPPORTED;
  memset(packed, 0, sizeof(float) * W_FLOATS);
  unsigned short *dst = reinterpret_cast<unsigned short *>(packed);
  auto put = [&](int unit_h, int lane, int i, float w) {  // unit_h: the h fragment's 64-lane block; the l fragment follows it
    unsigned short t[3];
    split_weight_terms(w, 2, t);
    dst[((size_t)(unit_h) + lane) * 8 + i] = t[0];
    dst[((size_t)(unit_h) + 64 + lane) * 8 + i] = t[1];
  };
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 8; ++i) {
      const int co = l & 15, k = 8 * (l >> 4) + i;
      if (k < 27) put(0, l, i, w0[co * 27 + k]);  // [co][ci][ky][kx] flattened: k = ci * 9 + ky * 3 + kx
    }
  const float *ws[2] = {w1, w2};
  for (int n = 0; n < 2; ++n)
    for (int j = 0; j < 5; ++j)
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 8; ++i) {
          const int co = l & 15, kq = l >> 4, tap = 2 * j + (kq >> 1), ci = 8 * (kq & 1) + i;
          if (tap < 9) put((n ? OFF_U2 : OFF_U1) + (2 * j) * 64, l, i, ws[n][(co * 16 + ci) * 9 + tap]);
        }
  // transposed convolution: output row 2M + py takes input rows (ky): py = 0 -> (M, ky 1), (M-1, ky 3); py = 1 -> (M+1, ky 0), (M, ky 2)
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px)
      for (int s = 0; s < 2; ++s)
        for (int l = 0; l < 64; ++l)
          for (int i = 0; i < 8; ++i) {
            const int co = l & 15, kq = l >> 4, j = 2 * s + (kq >> 1), ci = 8 * (kq & 1) + i;
            const int ky = (j >> 1) == 0 ? (py ? 0 : 1) : (py ? 2 : 3), kx = (j & 1) == 0 ? (px ? 0 : 1) : (px ? 2 : 3);
            put(OFF_U3 + (((py * 2 + px) * 2 + s) * 2) * 64, l, i, w3[((ci * 16 + co) * 4 + ky) * 4 + kx]);
          }
  const float *bs[4] = {b0, b1, b2, b3};
  for (int n = 0; n < 4; ++n)
    for (int c = 0; c < 16; ++c) packed[OFF_UB * 4 + n * 16 + c] = bs[n][c];
  return DRBA_OK;
}

int drba_head_fused16(const float *img, const float *packed_w, float *f_out, float *f_pair_out, int N, int H, int W, void *stream) {
  using namespace drba_head16;
  if (!img || !packed_w || !f_pair_out || N <= 0 || H < 2 || W < 2) return DRBA_EINVAL;  // f_out may be NULL: pair layout only
  if ((H & 1) || (W & 3) || N > 65535) return DRBA_EUNSUPPORTED;
  if ((((uintptr_t)f_out | (uintptr_t)f_pair_out | (uintptr_t)packed_w) & 15) != 0) return DRBA_EINVAL;
  const int tiles_x = (W / 2 + W2 - 1) / W2, tiles_y = (H / 2 + H2 - 1) / H2;
  if (max_dynamic_lds((const void *)head_fused16, LDS_BYTES) != hipSuccess) return DRBA_ELAUNCH;
  DRBA_LAUNCH(head_fused16, dim3(tiles_x * tiles_y, N), dim3(THREADS), LDS_BYTES, (hipStream_t)stream, img,
              reinterpret_cast<const u32x4 *>(packed_w), f_out, f_pair_out, H, W, tiles_x, status_bytes());
  DRBA_CHECK_LAUNCH();
  return range_checked(DRBA_OK, f_pair_out, (size_t)N * 16 * H * W, stream);
}

}  // extern "C"
