// IFNet's context encoder (IFNet_HDv3.py:23-47, `Head`: Conv2d(3,16,3,2,1) -> LeakyReLU -> Conv2d(16,16,3,1,1) -> LeakyReLU ->
// Conv2d(16,16,3,1,1) -> LeakyReLU -> ConvTranspose2d(16,16,4,2,1)) in ONE kernel.  Layer by layer it is four launches that move
// 25 + 33 | 33 + 33 | 33 + 33 | 33 + 134 MB per 1080p frame through HBM at 33-38 % of the fp32 MFMA peak, plus the
// pair-interleave copy of the result (268 MB) the warped gathers read: 225 us per frame, 9 % of a 1080p step and a quarter
// of a 4K scale-0.5 step (the encoder runs at the frame's full resolution whatever the flow scale).  Here a workgroup
// (8 waves) owns a 16 x 64 tile of the output = 8 x 32 half-resolution positions and keeps everything between the frame
// and the features in LDS:
//   IN  [3][29][77]   the frame under the tile with the 3 + 3 + 3 (+1) half-resolution rings the three 3x3 convolutions and
//                     the transposed one reach (zero outside the image: the first convolution's padding)
//   X0  [16][14][38]  cnn0 + LeakyReLU        X1 [16][12][36]  cnn1 + LeakyReLU        X2 [16][10][34]  cnn2 + LeakyReLU
//   (all three in one 14 x 40 "region" coordinate system, channel stride 560 = 16 mod 32: A fragments are conflict-free;
//    positions outside the half-resolution map are stored as zero -- the next layer's padding; X1 overlays IN, X2 X0)
// and every layer is exact-fp32 MFMA (v_mfma_f32_16x16x4_f32: 16 pixels x 16 output channels x 4 input channels of one
// tap), M tiles taken from the flattened region so that a tile's taps are one LDS read at a constant offset; a layer's
// weight fragments go from L2 straight into registers (coalesced 256-byte loads) and stay there for all tiles of a wave;
// two to four independent accumulator chains per wave.  The transposed convolution runs as its four output phases
// (2 x 2 taps each) and writes BOTH layouts from registers: [16, H, W] (what the unwarped first stage, tests and callers
// read) and the pair-interleaved [8, H, W, 2] the warped gathers read (lanes of a channel pair exchange halves).
// Redundant work for the rings: 1.47 x the MFMAs of the four layers; HBM traffic: the frame once, the features once per
// layout.  LDS: two 35 KB buffers, two workgroups per CU.  Results differ from the layer-by-layer path only by the
// accumulation order.  Measured alone (tools/exp/head_time.py): 200 us per 1080p frame against 215 for the four layers +
// the copy, 755 against 817 at 4K; 32 x 64 tiles (one 114 KB workgroup per CU, 1.30 x ring work) are faster alone (189 /
// 735) but the same in the step (826-837 frames/s either way, +1.2 % over the layers): their workgroups wait for a CU with
// 114 KB of LDS free while the main stream's kernels run, the kernel's launches stretch to 610 us in the step and hold
// stage_conv0's workgroups off in turn.  The matrix cores are 45 % busy.
#include "common.hpp"

#include <string.h>

using namespace drba;

namespace drba_head {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 16;                       // feature channels
constexpr int H2 = 8, W2 = 32;              // half-resolution positions per workgroup (output tile 16 x 64)
constexpr int RR = H2 + 6, RC = W2 + 6;     // region: 14 x 38 positions (cnn0's outputs)
constexpr int RS = 40;                      // region row stride
constexpr int CS = RR * RS;                 // 560: channel stride, = 16 (mod 32)
constexpr int IR = 2 * RR + 1, IC = 2 * RC + 1;  // frame window: 29 x 77
constexpr int IRS = 80, ICS = IR * IRS;     // its row / channel stride (2320)
constexpr int GUARD = 64;                   // floats in front of / behind each buffer (taps of discarded edge positions)
constexpr int BUF = C * CS;                 // 8960 floats >= 3 * ICS = 6960
constexpr int THREADS = 512;
// packed weights: cnn0 [7][64], cnn1 [9 taps][4 groups][64], cnn2 likewise, deconv [4 phases][4 taps][4 groups][64], biases [4][16]
constexpr int W0 = 7 * 64, W1 = 36 * 64, W3 = 64 * 64, WB = 4 * 16;
constexpr int OFF_W1 = W0, OFF_W2 = W0 + W1, OFF_W3 = W0 + 2 * W1, OFF_B = W0 + 2 * W1 + W3;
constexpr int W_FLOATS = OFF_B + WB;        // 9216
constexpr int LDS_FLOATS = 2 * (BUF + 2 * GUARD);  // 72.7 KB: two workgroups per CU (the weights go from L2 to registers)

static_assert(CS % 32 == 16, "channel stride");
static_assert(3 * ICS <= BUF, "frame window fits the region buffer it shares");


__global__ void __launch_bounds__(THREADS)
head_fused(const float *__restrict__ img, const float *__restrict__ wpk, float *__restrict__ f_out, float *__restrict__ fp_out, int H,
           int W, int tiles_x) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const float *wl = wpk;                          // weight fragments: coalesced 256-byte loads, L2-resident, held in registers per layer
  float *bufA = lds + GUARD;                      // IN, then X1
  float *bufB = bufA + BUF + 2 * GUARD;           // X0, then X2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int Hh = H >> 1, Wh = W >> 1;
  const size_t P = (size_t)H * W;
  int tx, ty;
  xcd_strip_tile(blockIdx.x, gridDim.x, tiles_x, tx, ty);
  const int m0 = ty * H2, n0 = tx * W2;           // the tile's first half-resolution row / column
  const int rm = m0 - 3, rn = n0 - 3;             // region origin (half-resolution coordinates)
  img += (size_t)blockIdx.y * 3 * P;
  if (f_out) f_out += (size_t)blockIdx.y * C * P;
  fp_out += (size_t)blockIdx.y * C * P;

  // ---- weights and the frame window (zero outside the image)
  {
    // all of a lane's loads are issued before the first is parked (unconditional loads at clamped addresses: a guarded
    // load per iteration serialises one memory latency per element, 14 of them per lane: 30 us per 1080p frame)
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
      if (tid + k * THREADS < 3 * IR * IRS) bufA[tid + k * THREADS] = v[k];
  }
  __syncthreads();
  // Positions outside the half-resolution map are stored as zero.  Only border tiles have any (a workgroup-uniform test);
  // there the four positions a lane stores (q .. q+3, q % 4 == 0, RS % 4 == 0: one region row) share one row test.
  const bool interior = rm >= 0 && rm + RR <= Hh && rn >= 0 && rn + RS <= Wh;
  auto store4 = [&](float *dst, int q, const f32x4 &acc, float bias, int ncol) {  // ncol: valid columns of the region row (RC or RS)
    f32x4 o;
    if (interior && ncol == RS) {
#pragma unroll
      for (int v = 0; v < 4; ++v) o[v] = lrelu02(acc[v] + bias);
    } else {
      const int r = q / RS, c = q - r * RS;
      const bool rowok = (unsigned)(rm + r) < (unsigned)Hh;
#pragma unroll
      for (int v = 0; v < 4; ++v) o[v] = (rowok && c + v < ncol && (unsigned)(rn + c + v) < (unsigned)Wh) ? lrelu02(acc[v] + bias) : 0.f;
    }
    *reinterpret_cast<f32x4 *>(dst + q) = o;
  };

  // ---- cnn0: 3 -> 16, stride 2, + LeakyReLU: IN -> X0 (bufB).  K = 27 as 7 groups of 4 (k = 4 i + kq -> channel, tap)
  {
    float b[7];
    int off[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      b[i] = wl[i * 64 + lane];
      const int k = min(4 * i + kq, 26), ci = k / 9, t = k - ci * 9;
      off[i] = ci * ICS + (t / 3) * IRS + (t % 3);
    }
    const float bias = wl[OFF_B + m];
    for (int t = wave; t < RR * 3; t += 8) {       // 16-position tiles, three per region row (the last one runs over the row's end)
      const int r = t / 3, c0 = (t - r * 3) * 16;
      const float *a = bufA + 2 * r * IRS + 2 * (c0 + m);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 7; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[off[i]], b[i], acc, 0, 0, 0);
      const int c = c0 + 4 * kq;                    // this lane: positions c .. c+3 of row r, output channel m
      if (c < RS) store4(bufB + m * CS, r * RS + c, acc, bias, RC);
    }
  }
  __syncthreads();

  // ---- cnn1 / cnn2: 16 -> 16, + LeakyReLU, on the flattened region (tile = 16 consecutive positions, tap = constant offset)
  auto conv16 = [&](const float *src, float *dst, int woff, int boff, int q_lo, int q_hi) {
    float b[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) b[i] = wl[woff + i * 64 + lane];
    const float bias = wl[OFF_B + boff + m];
    // two tiles per pass: two independent accumulator chains keep the matrix pipe busy with 2 waves per SIMD
    for (int q0 = q_lo + 16 * wave; q0 < q_hi; q0 += 2 * 16 * 8) {
      const int q1 = q0 + 16 * 8;
      const bool two = q1 < q_hi;
      const float *a0 = src + kq * CS + q0 + m, *a1 = src + kq * CS + (two ? q1 : q0) + m;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int o = 4 * g * CS + (t / 3 - 1) * RS + (t % 3 - 1);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[o], b[t * 4 + g], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[o], b[t * 4 + g], acc1, 0, 0, 0);
        }
      store4(dst + m * CS, q0 + 4 * kq, acc0, bias, RS);
      if (two) store4(dst + m * CS, q1 + 4 * kq, acc1, bias, RS);
    }
  };
  conv16(bufB, bufA, OFF_W1, 16, 1 * RS, (RR - 1) * RS);   // X0 -> X1: region rows 1..20
  __syncthreads();
  conv16(bufA, bufB, OFF_W2, 32, 2 * RS, (RR - 2) * RS);   // X1 -> X2: region rows 2..19
  __syncthreads();

  // ---- ConvTranspose2d(16, 16, 4, 2, 1): out(2M + py, 2N + px) = sum over 2 x 2 taps of X2 (rows M - 1 + py .., see the pack)
  {
    const float bias = wl[OFF_B + 48 + m];
    float b3[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) b3[i] = wl[OFF_W3 + i * 64 + lane];
    for (int u = wave; u < H2 * 2; u += 8) {      // (region row, 16-position half) units
      const int r = 3 + (u >> 1), c0 = 3 + (u & 1) * 16;
      const int M = rm + r, N = rn + c0 + 4 * kq;  // this lane: positions N .. N+3, output channel m
      f32x4 acc4[2][2];
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) acc4[py][px] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {  // four independent accumulator chains
              const int dr = (j >> 1) == 0 ? (py ? 1 : 0) : (py ? 0 : -1);   // tap jy: py = 0 -> rows M, M-1; py = 1 -> rows M+1, M
              const int dc = (j & 1) == 0 ? (px ? 1 : 0) : (px ? 0 : -1);
              acc4[py][px] = __builtin_amdgcn_mfma_f32_16x16x4f32(bufB[(kq + 4 * g) * CS + (r + dr) * RS + c0 + m + dc],
                                                                  b3[((py * 2 + px) * 4 + j) * 4 + g], acc4[py][px], 0, 0, 0);
            }
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        const f32x4 *acc = acc4[py];
        // lane (channel m, group kq): columns 2N .. 2N+7 of row 2M + py: (acc[0][v], acc[1][v]) = columns 2(N+v), 2(N+v)+1
        const int y = 2 * M + py, x = 2 * N;
        float v8[8], o8[8];
#pragma unroll
        for (int v = 0; v < 4; ++v) v8[2 * v] = acc[0][v] + bias, v8[2 * v + 1] = acc[1][v] + bias;
#pragma unroll
        for (int k = 0; k < 8; ++k) o8[k] = quad_xor1(v8[k]);  // the other channel of the pair (lane m ^ 1): a DPP move, not a bpermute
        if (M < Hh && x < W) {
          float *pf = f_out + (size_t)m * P + (size_t)y * W + x;
          if (!f_out) {  // pair layout only (the hot path: every consumer reads the pair layout; -134 MB written per 1080p frame)
          } else if (x + 7 < W) {
            *reinterpret_cast<f32x4 *>(pf) = (f32x4){v8[0], v8[1], v8[2], v8[3]};
            *reinterpret_cast<f32x4 *>(pf + 4) = (f32x4){v8[4], v8[5], v8[6], v8[7]};
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
              if (x + k < W) pf[k] = v8[k];
          }
          // pair layout [C/2][H][W][2]: the even channel's lane writes columns x .. x+3, the odd one's x+4 .. x+7
          const int odd = m & 1, xs = x + 4 * odd;
          float *pp = fp_out + ((size_t)(m >> 1) * H + y) * (size_t)W * 2 + (size_t)xs * 2;
          float e[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float mine = v8[4 * odd + k], other = o8[4 * odd + k];
            e[2 * k] = odd ? other : mine, e[2 * k + 1] = odd ? mine : other;
          }
          if (xs + 3 < W) {
            *reinterpret_cast<f32x4 *>(pp) = (f32x4){e[0], e[1], e[2], e[3]};
            *reinterpret_cast<f32x4 *>(pp + 4) = (f32x4){e[4], e[5], e[6], e[7]};
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (xs + k < W) pp[2 * k] = e[2 * k], pp[2 * k + 1] = e[2 * k + 1];
          }
        }
      }
    }
  }
}

}  // namespace drba_head

extern "C" {

size_t drba_head_fused_packed_floats(void) { return (size_t)drba_head::W_FLOATS; }

/* HOST: w0 [16,3,3,3], w1 / w2 [16,16,3,3], w3 [16(in),16(out),4,4] (ConvTranspose2d layout), b0..b3 [16] -> packed */
int drba_head_fused_pack(const float *w0, const float *b0, const float *w1, const float *b1, const float *w2, const float *b2,
                         const float *w3, const float *b3, float *packed) {
  using namespace drba_head;
  if (!w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !packed) return DRBA_EINVAL;
  memset(packed, 0, sizeof(float) * W_FLOATS);
  for (int i = 0; i < 7; ++i)
    for (int l = 0; l < 64; ++l) {
      const int co = l & 15, k = 4 * i + (l >> 4);
      if (k < 27) packed[i * 64 + l] = w0[co * 27 + k];  // [co][ci][ky][kx] flattened: k = ci * 9 + ky * 3 + kx
    }
  const float *ws[2] = {w1, w2};
  for (int n = 0; n < 2; ++n)
    for (int t = 0; t < 9; ++t)
      for (int g = 0; g < 4; ++g)
        for (int l = 0; l < 64; ++l)
          packed[(n ? OFF_W2 : OFF_W1) + (t * 4 + g) * 64 + l] = ws[n][((l & 15) * 16 + 4 * g + (l >> 4)) * 9 + t];
  // transposed convolution: output row 2M + py takes input rows (ky): py = 0 -> (M, ky 1), (M-1, ky 3); py = 1 -> (M+1, ky 0), (M, ky 2)
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px)
      for (int j = 0; j < 4; ++j) {
        const int ky = (j >> 1) == 0 ? (py ? 0 : 1) : (py ? 2 : 3), kx = (j & 1) == 0 ? (px ? 0 : 1) : (px ? 2 : 3);
        for (int g = 0; g < 4; ++g)
          for (int l = 0; l < 64; ++l) {
            const int co = l & 15, ci = 4 * g + (l >> 4);
            packed[OFF_W3 + (((py * 2 + px) * 4 + j) * 4 + g) * 64 + l] = w3[((ci * 16 + co) * 4 + ky) * 4 + kx];
          }
      }
  const float *bs[4] = {b0, b1, b2, b3};
  for (int n = 0; n < 4; ++n)
    for (int c = 0; c < 16; ++c) packed[OFF_B + n * 16 + c] = bs[n][c];
  return DRBA_OK;
}

int drba_head_fused(const float *img, const float *packed_w, float *f_out, float *f_pair_out, int N, int H, int W, void *stream) {
  using namespace drba_head;
  if (!img || !packed_w || !f_pair_out || N <= 0 || H < 2 || W < 2) return DRBA_EINVAL;  // f_out may be NULL: pair layout only
  if ((H & 1) || (W & 3) || N > 65535) return DRBA_EUNSUPPORTED;  // even rows; 16-byte aligned rows in both layouts
  if ((((uintptr_t)f_out | (uintptr_t)f_pair_out | (uintptr_t)packed_w) & 15) != 0) return DRBA_EINVAL;
  const int tiles_x = (W / 2 + W2 - 1) / W2, tiles_y = (H / 2 + H2 - 1) / H2;
  constexpr size_t lds_bytes = (size_t)LDS_FLOATS * 4;
  if (max_dynamic_lds((const void *)head_fused, (int)lds_bytes) != hipSuccess) return DRBA_ELAUNCH;
  DRBA_LAUNCH(head_fused, dim3(tiles_x * tiles_y, N), dim3(THREADS), lds_bytes, (hipStream_t)stream, img, packed_w, f_out, f_pair_out, H, W,
              tiles_x);
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // extern "C"
