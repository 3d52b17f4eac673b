// 3x3 convolution (stride 1, pad 1), split-bf16 on the matrix cores, every operand streamed by LDS-DMA.
//
// Same arithmetic as conv_split.hip (an fp32 value is the sum of three bf16 terms h + m + l, a product is evaluated as
// the six partial products >= 2^-16 of the leading one, fp32 accumulation: fp32-level error), different machine:
//   * conv_split_mfma fetches the fp32 window into REGISTERS (48 dword loads per lane), splits it with VALU code in a
//     staging phase, writes three bf16 planes to LDS, and reads its weight fragments from L2 into registers inside the MFMA
//     loop.  All of that sits in the wave's in-order vmcnt queue: the weight fragments of a tap wait behind the next
//     tile's window, the staging phase waits for HBM, and per 8x32 tile the matrix cores work 2.9 us of 17
//     ("conv_split_mfma").
//   * here NOTHING passes through registers on its way in.  The fp32 window ([32 channels][10 rows][40 columns], the
//     NCHW planes as they lie in HBM) and the host-split weight fragments of one kernel row are copied global -> LDS by
//     `buffer_load ... lds` (16 bytes per lane), two steps ahead of their use, and waited for with COUNTED s_waitcnt
//     vmcnt(N) + one workgroup barrier per step (a step = one kernel row dy of one 32-channel chunk = 3 taps).
//     A lane builds its A operand from 8 conflict-free ds_read_b32 (8 channels of one pixel) and splits it ON THE WAY
//     into the MFMAs: 44 VALU instructions per 12 MFMAs, issued under the matrix pipe's 17 cycles per instruction
//     (VALU and MFMA are separate pipes; two waves share a SIMD, one splits while the other's MFMAs run).
//     The HBM format does not change (fp32 NCHW in, fp32 NCHW out): the kernel sits behind the same drba_conv3x3 entry
//     point as another configuration id and the autotuner keeps it where it is faster.
//
// Workgroup = 8 waves (2 per SIMD), one per CU (the LDS is the limit: 155 KB), persistent over work items.
// Work item = (image, 8 x 32 pixel tile, tile of 16*NT output channels); wave w owns row w: 2 blocks of 16 pixels
// (MFMA M) x NT blocks of 16 output channels (MFMA N); K = 32 channels of one tap.
//   K order inside an MFMA: lane (m = lane % 16, kq = lane / 16) supplies k = 8 kq .. 8 kq + 7, mapped to channels
//   4 i + kq (i = 0..7) of the chunk -- with the channel stride of the window 400 dwords == 16 (mod 32) the 32 lanes a
//   ds_read_b32 services together (kq = 0, 1) hit 32 distinct banks.  The weights are packed with the same mapping.
// LDS (bytes): A[2] 2 x 51200 (window of a chunk, double buffered) | W[3] 3 x 9 KB x NT (ring of kernel-row fragment
//   blocks [dx][nt][plane h/m/l][64 lanes][16 B]) | bias/beta[2] 2 x 512 | dump 1 KB (target of the padding DMAs that
//   keep every wave's instruction count equal, so that the waits are compile-time immediates).
// DMA schedule (chunk c of the workgroup's flat chunk sequence, steps dy = 0, 1, 2; "allow n" = s_waitcnt vmcnt(n)):
//   top of (c,0): allow |g2| (+ the stores of an epilogue in between), barrier, issue g0 = W(c,2), A(c+1) units 0..31
//   top of (c,1): allow |g0| (+ stores),                                barrier, issue g1 = W(c+1,0), A(c+1) units 32..49
//   top of (c,2): allow 4 + |g1|,                                       barrier, issue g2 = W(c+1,1), bias/beta(c+1)
//   so every block has two whole steps (~2 us) to land and a wave never waits for anything younger than it needs.
// ResConv layers (residual == input, Cin == Cout, NT = 2): the chunks of an item are taken in rotated order so that the
// item's own 32 channels are the last window staged, and the epilogue reads the residual from LDS (exact fp32).
// Reference operator: nn.Conv2d(c, c, 3, 1, 1) + bias, * beta + x, LeakyReLU(0.2)
// (models/rife_426_heavy/IFNet_HDv3.py:11-16,50-59) and the stride-1 convolutions of FeatureNet / MetricNet / GridNet.
#include "common.hpp"
#include "conv_split.hpp"

#include <string.h>

using namespace drba;

namespace drba_conv_dma {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr;

constexpr int CK = 32;                 // channels per chunk = K of one bf16 MFMA
constexpr int TH = 8, TW = 32;         // pixels per work item: one row per wave, two 16-pixel MFMA row blocks
constexpr int WR = TH + 2, WC = TW + 8;  // window: rows y0-1 .. y0+8, columns x0-4 .. x0+35 (16-byte units)
constexpr int CS = WR * WC;            // 400 dwords per channel, == 16 (mod 32)
constexpr int A_BYTES = CK * CS * 4;   // 51200
constexpr int A_INSTR = A_BYTES / 1024;  // 50 wave-level DMA instructions of 64 lanes x 16 B
constexpr int A_P0 = 32, A_P1 = A_INSTR - A_P0;  // pieces: 4 + 3 instructions per wave (8 waves)
constexpr int A_S0 = 4, A_S1 = 3;      // slots per wave
static_assert(CS % 32 == 16 && A_BYTES % 1024 == 0 && A_P1 <= 8 * A_S1, "window layout");
[[maybe_unused]] constexpr unsigned kOOB = 0x7FFFFFF0u;  // beyond any num_records: the load returns 0 (zero padding), never faults

template <int NT_>
struct DmaCfg {
  static constexpr int NT = NT_, NTC = 16 * NT;
  static constexpr int W_STEP = 3 * NT * 3 * 1024;     // bytes of one kernel row's fragments: [dx][nt][plane][64][16]
  static constexpr int W_INSTR = W_STEP / 1024;        // 9 NT
  static constexpr int W_S = (W_INSTR + 7) / 8;        // slots per wave
  static constexpr int OFF_A = 0, OFF_W = 2 * A_BYTES, OFF_BB = OFF_W + 3 * W_STEP, OFF_DUMP = OFF_BB + 1024;
  static constexpr int LDS_BYTES = OFF_DUMP + 1024;
  static constexpr int G0 = W_S + A_S0, G1 = W_S + A_S1, G2 = W_S + 2;  // DMA instructions per wave in the three groups
  static constexpr int STORES = 2 * NT;                // epilogue store instructions per wave and item
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// fp32 -> (h, m, l) bf16 with round-to-nearest-even at every step; the three terms of 2 values packed (conv_split.hip)
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

struct Item {  // scalar (wave-uniform) description of a work item
  int x0, y0, cz, n;
};

#define DRBA_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <class Cfg, bool PRE, bool RL>
__global__ void __launch_bounds__(512, 1)
conv_dma_mfma(const float *__restrict__ in, const u32x4 *__restrict__ wfrag, const float *__restrict__ bias,
              const float *__restrict__ beta, const float *__restrict__ res, const float *__restrict__ res2,
              float *__restrict__ out, int Cin, int H, int W, int Cout, int act, float post_slope, float pre_slope,
              int n_ctiles, int nbx, int nby, int total) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NT = Cfg::NT;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // the ONLY LDS object of the kernel

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int HW = H * W;
  const int nchunks = Cin / CK;
  const unsigned plane_bytes = (unsigned)HW * 4u;
  const unsigned img_bytes = (unsigned)Cin * plane_bytes;

  auto decode = [&](int work) -> Item {
    int t = xcd_band(work, total);
    Item c;
    c.cz = t % n_ctiles;
    t /= n_ctiles;
    const int bx = t % nbx;
    t /= nbx;
    c.x0 = bx * TW, c.y0 = (t % nby) * TH, c.n = t / nby;
    return c;
  };
  // chunk taken as the qi-th of an item (RL: rotated so that the item's own 32 output channels come last)
  auto chunk_of = [&](const Item &c, int qi) -> int {
    if (!RL) return qi;
    int q = qi + c.cz + 1;
    return q >= nchunks ? q - nchunks : q;
  };

  // ---- per-lane constants of the window DMA: slot t covers LDS units 64 k_t .. 64 k_t + 63 of the window, unit u =
  // (channel u / 100, row (u % 100) / 10, 16-byte column group (u % 100) % 10); its source offset relative to the
  // window's origin is fixed for the whole kernel, only the in-image test depends on the item
  unsigned a_rel[A_S0 + A_S1];
  int a_rc[A_S0 + A_S1];   // row | (column << 8)
  int a_k[A_S0 + A_S1];    // DMA instruction index (scalar), -1: padding slot
#pragma unroll
  for (int t = 0; t < A_S0 + A_S1; ++t) {
    const int k = t < A_S0 ? t * 8 + wave : A_P0 + (t - A_S0) * 8 + wave;
    a_k[t] = k < A_INSTR ? k : -1;
    const int u = k * 64 + lane;
    const int c = u / (WR * (WC / 4)), e = u - c * (WR * (WC / 4));
    const int r = e / (WC / 4), j = e - r * (WC / 4);
    a_rel[t] = (unsigned)c * plane_bytes + (unsigned)(r * W + 4 * j) * 4u;
    a_rc[t] = r | ((4 * j) << 8);
  }
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)wfrag, 0, n_ctiles * nchunks * 3 * Cfg::W_STEP, 0x00020000);
  const __amdgpu_buffer_rsrc_t bias_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)bias, 0, bias ? Cout * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t beta_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)beta, 0, beta ? Cout * 4 : 0, 0x00020000);

  // window of chunk q of item c -> A buffer `ab` (0 / 1), piece 0 (slots 0..3) or 1 (slots 4..6)
  auto issue_A = [&](const Item &c, int q, int ab, int piece) {
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)c.n * Cin * HW), 0, img_bytes, 0x00020000);
    const unsigned soff = (unsigned)(q * CK) * plane_bytes;
    const unsigned tb = (unsigned)(((c.y0 - 1) * W + c.x0 - 4) * 4);  // wraps for the first row / column: valid lanes add up >= 0
    const int t0 = piece ? A_S0 : 0, t1 = piece ? A_S0 + A_S1 : A_S0;
#pragma unroll
    for (int t = t0; t < t1; ++t) {
      const int r = a_rc[t] & 0xff, col = a_rc[t] >> 8;
      const bool ok = a_k[t] >= 0 && (unsigned)(c.y0 - 1 + r) < (unsigned)H && (unsigned)(c.x0 - 4 + col) < (unsigned)W;
      const unsigned voff = ok ? a_rel[t] + tb : kOOB;
      const int dst = a_k[t] >= 0 ? Cfg::OFF_A + ab * A_BYTES + a_k[t] * 1024 : Cfg::OFF_DUMP;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + dst), 16, voff, soff, 0, 0);
    }
  };
  // fragments of kernel row dy of chunk q, cout tile cz -> W ring slot dy
  auto issue_W = [&](const Item &c, int q, int dy) {
    const unsigned soff = (unsigned)(((c.cz * nchunks + q) * 3 + dy) * Cfg::W_STEP);
#pragma unroll
    for (int i = 0; i < Cfg::W_S; ++i) {
      const int k = i * 8 + wave;
      const bool ok = k < Cfg::W_INSTR;
      const unsigned voff = ok ? (unsigned)(k * 1024 + lane * 16) : kOOB;
      const int dst = ok ? Cfg::OFF_W + dy * Cfg::W_STEP + k * 1024 : Cfg::OFF_DUMP;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr)(lds + dst), 16, voff, soff, 0, 0);
    }
  };
  // bias / beta of the item's NTC output channels -> bb[parity]: [bias: 64 lanes x 4 B][beta: 64 x 4 B] (a 4-byte DMA writes all
  // 64 lanes, zeros for the lanes past NTC; every wave writes the same bytes)
  auto issue_BB = [&](const Item &c, int parity) {
    const int co = c.cz * Cfg::NTC + lane;
    const unsigned voff = (lane < Cfg::NTC && co < Cout) ? (unsigned)co * 4u : kOOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(bias_rsrc, (lds_ptr)(lds + Cfg::OFF_BB + parity * 512), 4, voff, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(beta_rsrc, (lds_ptr)(lds + Cfg::OFF_BB + parity * 512 + 256), 4, voff, 0, 0, 0);
  };
  static_assert(Cfg::NTC <= 64, "bias/beta slots hold 64 channels");

  int work = blockIdx.x;
  if (work >= total) return;
  Item cur = decode(work);
  int qi = 0;
  // ---- prologue: everything the first chunk needs, in the order the steady-state waits assume:
  // [A piece 0, A piece 1, bias/beta] then g1-like [W(c0,0)] then g2-like [W(c0,1)]; the first wait allows |g2| = G2,
  // so pad the last group to G2 instructions and the one before it to anything (it is waited for completely)
  {
    const int q0 = chunk_of(cur, 0);
    issue_A(cur, q0, 0, 0);
    issue_A(cur, q0, 0, 1);
    issue_W(cur, q0, 0);
    issue_W(cur, q0, 1);
    issue_BB(cur, 0);  // W_S + 2 = G2 instructions since the last thing the first wait needs
  }
  int par = 0;          // parity of the current chunk in the workgroup's flat chunk sequence (A buffer, bias/beta slot)
  bool stored = false;  // an epilogue's stores were issued since the last group (they sit in the vmcnt queue too)

  f32x4 acc[2][NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < NT; ++c) acc[b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  const float *ldsf = reinterpret_cast<const float *>(lds);
  const u32x4 *ldsq = reinterpret_cast<const u32x4 *>(lds);
  const int a_lane = kq * CS + wave * WC + m + 3;  // dword of (channel kq, window row w, column of pixel m, tap dx = 0)
  const int w_lane = Cfg::OFF_W / 16 + lane;

  // one step: taps (dy, dx = 0..2) of the chunk in A buffer `ab`, fragments in W slot dy
  auto compute = [&](int ab, int dy) {
    const float *ap = ldsf + ab * (A_BYTES / 4) + a_lane + dy * WC;
    const u32x4 *wp = ldsq + w_lane + dy * (Cfg::W_STEP / 16);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      u32x4 bw[NT][3];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bw[nt][pl] = wp[((dx * NT + nt) * 3 + pl) * 64];
#pragma unroll
      for (int mw = 0; mw < 2; ++mw) {
        float raw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[i] = ap[4 * i * CS + dx + 16 * mw];
        u32x4 h, mm, l;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float a = raw[2 * i], b = raw[2 * i + 1];
          if (PRE) {
            a = a > 0.f ? a : a * pre_slope;
            b = b > 0.f ? b : b * pre_slope;
          }
          unsigned hh, hm, hl;
          split2(a, b, hh, hm, hl);
          h[i] = hh, mm[i] = hm, l[i] = hl;
        }
        const bf16x8 ah = __builtin_bit_cast(bf16x8, h), am = __builtin_bit_cast(bf16x8, mm), al = __builtin_bit_cast(bf16x8, l);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const bf16x8 bh = __builtin_bit_cast(bf16x8, bw[nt][0]), bm = __builtin_bit_cast(bf16x8, bw[nt][1]);
          const bf16x8 bl = __builtin_bit_cast(bf16x8, bw[nt][2]);
          f32x4 c = acc[mw][nt];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);  // smallest terms first
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
          acc[mw][nt] = c;
        }
      }
    }
  };

  while (true) {
    // ---- the chunk after this one in the workgroup's sequence (the same item's next chunk, or the next item's first);
    // past the end the current chunk is fetched again (into the free buffers, never used)
    Item nxt = cur;
    int nqi = qi + 1;
    bool have_next = true;
    if (nqi == nchunks) {
      nqi = 0;
      const int nw = work + (int)gridDim.x;
      if (nw < total) nxt = decode(nw);
      else have_next = false, nqi = qi;
    }
    const int q = chunk_of(cur, qi), nq = chunk_of(nxt, nqi);

    // ---- step (c, 0)
    if (stored) DRBA_WAIT_VM(Cfg::G2 + Cfg::STORES);
    else DRBA_WAIT_VM(Cfg::G2);
    __builtin_amdgcn_s_barrier();
    issue_W(cur, q, 2);
    issue_A(nxt, nq, par ^ 1, 0);
    compute(par, 0);
    // ---- step (c, 1)
    if (stored) DRBA_WAIT_VM(Cfg::G0 + Cfg::STORES);
    else DRBA_WAIT_VM(Cfg::G0);
    stored = false;
    __builtin_amdgcn_s_barrier();
    issue_W(nxt, nq, 0);
    issue_A(nxt, nq, par ^ 1, 1);
    compute(par, 1);
    // ---- step (c, 2)
    DRBA_WAIT_VM(A_S0 + Cfg::G1);
    __builtin_amdgcn_s_barrier();
    issue_W(nxt, nq, 1);
    issue_BB(nxt, par ^ 1);
    compute(par, 2);

    if (qi + 1 == nchunks) {
      // ---- epilogue (conv_split.hip): y = acc + bias; ResConv: y = y * beta + x; otherwise y += res (+ res2); activation
      // (0 none, 1 LeakyReLU(0.2), 2 PReLU(post_slope), 3 ReLU, 4 tanh * 10) selected once around the tile.
      // Whole-line stores: the accumulator leaves lane (m, kq) with 4 consecutive x of ONE cout; the two column blocks
      // of the row are regrouped across lanes (one xor-8 exchange + one permutation) so that lanes 8c..8c+7 hold the 32
      // consecutive pixels of cout c -- every store moves 8 whole 128-byte row segments, couts 0-7 (A) then 8-15 (B).
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int c8 = ln >> 3, blk = (ln >> 2) & 1, q4 = ln & 3;
      const int src = q4 * 16 + blk * 8 + c8;
      const bool hi = (ln & 8) != 0;
      const size_t img = (size_t)cur.n * Cout * HW;
      const unsigned obytes = (unsigned)Cout * plane_bytes;
      const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(out + img), 0, obytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rrsrc =
          __builtin_amdgcn_make_buffer_rsrc((void *)((res ? res : out) + img), 0, res ? obytes : 0u, 0x00020000);
      const __amdgpu_buffer_rsrc_t r2rsrc =
          __builtin_amdgcn_make_buffer_rsrc((void *)((res2 ? res2 : out) + img), 0, res2 ? obytes : 0u, 0x00020000);
      const float *bb = ldsf + (Cfg::OFF_BB + par * 512) / 4;
      const int y = cur.y0 + wave, xb = cur.x0 + blk * 16 + q4 * 4;
      auto epilogue = [&](auto post) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int chA = nt * 16 + c8, coA = cur.cz * Cfg::NTC + chA;
          const float bsA = bb[chA], bsB = bb[chA + 8], btA = bb[64 + chA], btB = bb[64 + chA + 8];
          const bool in_img = y < H && xb < W;
          const unsigned base = (unsigned)(((coA * H + y) * W + xb) * 4);
          const unsigned oa = (in_img && coA < Cout) ? base : 0xffffffffu;
          const unsigned ob = (in_img && coA + 8 < Cout) ? base + 8u * plane_bytes : 0xffffffffu;
          const f32x4 v0 = acc[0][nt], v1 = acc[1][nt];
          f32x4 a, b;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float got = __shfl_xor(hi ? v0[k] : v1[k], 8, 64);
            a[k] = __shfl(hi ? got : v0[k], src, 64);
            b[k] = __shfl(hi ? v1[k] : got, src, 64);
          }
          f32x4 xa = (f32x4){0.f, 0.f, 0.f, 0.f}, xb_ = xa, xa2 = xa, xb2 = xa;
          if (RL) {
            // the item's own channels are the window in A[par]: channel chA, window row w + 1, columns 4 + blk*16 + 4 q4 ..+3
            const float *xp = ldsf + par * (A_BYTES / 4) + chA * CS + (wave + 1) * WC + 4 + blk * 16 + q4 * 4;
            xa = *reinterpret_cast<const f32x4 *>(xp);
            xb_ = *reinterpret_cast<const f32x4 *>(xp + 8 * CS);
          } else if (res) {
            xa = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, oa, 0, 0));
            xb_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ob, 0, 0));
          }
          if (!RL && res2) {
            xa2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2rsrc, oa, 0, 0));
            xb2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2rsrc, ob, 0, 0));
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float ua = a[k] + bsA, ub = b[k] + bsB;
            if (beta) {
              ua = ua * btA + xa[k];
              ub = ub * btB + xb_[k];
            } else {
              if (RL || res) ua = ua + xa[k], ub = ub + xb_[k];
              if (!RL && res2) ua = ua + xa2[k], ub = ub + xb2[k];
            }
            a[k] = post(ua);
            b[k] = post(ub);
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), orsrc, oa, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, b), orsrc, ob, 0, 0);
        }
      };
      switch (act) {
        case 1: epilogue([](float v) { return lrelu02(v); }); break;
        case 2: epilogue([post_slope](float v) { return v > 0.f ? v : post_slope * v; }); break;
        case 3: epilogue([](float v) { return fmaxf(v, 0.f); }); break;
        case 4: epilogue([](float v) { return tanhf(v) * 10.f; }); break;
        default: epilogue([](float v) { return v; }); break;
      }
      stored = true;
      zero_acc();
      if (!have_next) break;
      work += (int)gridDim.x;
    }
    cur = nxt;
    qi = nqi;
    par ^= 1;
  }
  // drain: DMAs of the never-used lookahead are still writing this workgroup's LDS
  DRBA_WAIT_VM(0);
#endif
}

// ------------------------------------------------------------------------------------------ host side
using D2 = DmaCfg<2>;
constexpr int kNum = 1;

template <class Cfg, bool PRE, bool RL>
hipError_t lds_limit() {
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_dma_mfma<Cfg, PRE, RL>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
  return e;
}

template <class Cfg>
int launch(const float *in, const float *wpk, const float *bias, const float *beta, const float *res, const float *res2,
           float *out, int N, int Cin, int H, int W, int Cout, int act, float post_slope, int pre_act, float pre_slope,
           hipStream_t s) {
  const int n_ct = (Cout + Cfg::NTC - 1) / Cfg::NTC;
  const int nbx = (W + TW - 1) / TW, nby = (H + TH - 1) / TH;
  const long long total = (long long)nbx * nby * N * n_ct;
  if (total >= (1ll << 31)) return DRBA_EUNSUPPORTED;
  long long grid = 256;  // one persistent workgroup per CU
  if (grid > total) grid = (total + 7) / 8 * 8;
  dim3 g((unsigned)grid);
  const u32x4 *wf = reinterpret_cast<const u32x4 *>(wpk);
  auto go = [&](auto kernel, hipError_t lds_ok) -> int {
    if (lds_ok != hipSuccess) return DRBA_ELAUNCH;
    DRBA_LAUNCH(kernel, g, dim3(512), Cfg::LDS_BYTES, s, in, wf, bias, beta, res, res2, out, Cin, H, W, Cout, act, post_slope,
                pre_slope, n_ct, nbx, nby, (int)total);
    return DRBA_OK;
  };
  const bool rl = Cfg::NTC == CK && res && res == in && !res2 && !pre_act && Cin == Cout;
  const int rc = rl ? go(conv_dma_mfma<Cfg, false, true>, lds_limit<Cfg, false, true>())
                    : (pre_act ? go(conv_dma_mfma<Cfg, true, false>, lds_limit<Cfg, true, false>())
                               : go(conv_dma_mfma<Cfg, false, false>, lds_limit<Cfg, false, false>()));
  if (rc != DRBA_OK) return rc;
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

static inline float bf16_round(float x) {  // round-to-nearest-even fp32 -> bf16 (finite inputs), as the fp32 value it represents
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

}  // namespace drba_conv_dma

namespace drba {

int conv_dma_num_cfgs() { return drba_conv_dma::kNum; }

bool conv_dma_supports(int Cin, int Cout, int id) {
  return id >= 0 && id < drba_conv_dma::kNum && Cin > 0 && Cout > 0 && Cin % drba_conv_dma::CK == 0;
}

size_t conv_dma_packed_floats(int Cin, int Cout, int id) {
  if (!conv_dma_supports(Cin, Cout, id)) return 0;
  using C = drba_conv_dma::D2;
  const size_t n_ct = (Cout + C::NTC - 1) / C::NTC, nch = Cin / drba_conv_dma::CK;
  return n_ct * nch * 3 * (C::W_STEP / 4);
}

// packed (16-byte units): [cout tile][chunk][dy][dx][nt][plane h/m/l][lane] = 8 bf16, element i =
//   w[cz*NTC + nt*16 + (lane & 15)][q*32 + 4*i + (lane >> 4)][3*dy + dx], zero outside Cout
int conv_dma_pack(const float *w, float *packed, int Cin, int Cout, int id) {
  using namespace drba_conv_dma;
  if (!w || !packed || !conv_dma_supports(Cin, Cout, id)) return DRBA_EINVAL;
  using C = D2;
  const int n_ct = (Cout + C::NTC - 1) / C::NTC, nch = Cin / CK;
  memset(packed, 0, sizeof(float) * conv_dma_packed_floats(Cin, Cout, id));
  unsigned short *dst = reinterpret_cast<unsigned short *>(packed);
  for (int cz = 0; cz < n_ct; ++cz)
    for (int q = 0; q < nch; ++q)
      for (int tap = 0; tap < 9; ++tap)
        for (int nt = 0; nt < C::NT; ++nt)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = cz * C::NTC + nt * 16 + (lane & 15);
            if (co >= Cout) continue;
            for (int i = 0; i < 8; ++i) {
              const int ci = q * CK + 4 * i + (lane >> 4);
              const float x = w[((size_t)co * Cin + ci) * 9 + tap];
              const float h = bf16_round(x), m = bf16_round(x - h), l = bf16_round(x - h - m);
              const float term[3] = {h, m, l};
              for (int pl = 0; pl < 3; ++pl) {
                const size_t unit = ((((size_t)cz * nch + q) * 9 + tap) * C::NT + nt) * 3 + pl;
                dst[(unit * 64 + lane) * 8 + i] = bf16_bits(term[pl]);
              }
            }
          }
  return DRBA_OK;
}

int conv_dma_launch(int id, const float *in, const float *packed_w, const float *bias, const float *beta,
                    const float *residual, const float *residual2, float *out, int N, int Cin, int H, int W, int Cout,
                    int act, float post_slope, int pre_act, float pre_slope, void *stream) {
  using namespace drba_conv_dma;
  if (!conv_dma_supports(Cin, Cout, id)) return DRBA_EUNSUPPORTED;
  if ((W & 3) != 0) return DRBA_EUNSUPPORTED;  // the window moves in 16-byte units
  if ((size_t)Cin * H * W * 4 >= (1ull << 31) - 64) return DRBA_EUNSUPPORTED;  // 32-bit byte offsets inside an image, below kOOB
  if ((size_t)Cout * H * W * 4 >= (1ull << 31) - 64) return DRBA_EUNSUPPORTED;
  return launch<D2>(in, packed_w, bias, beta, residual, residual2, out, N, Cin, H, W, Cout, act, post_slope, pre_act,
                    pre_slope, (hipStream_t)stream);
}

}  // namespace drba
