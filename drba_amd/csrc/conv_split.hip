// 3x3 convolution (stride 1, pad 1) for MFMA-bound layers: fp32 operands are split into three bf16 terms
// (x = h + m + l, 8 + 8 + 8 mantissa bits: the split is exact up to the last bit of the fp32 mantissa) and the product
// is evaluated on the bf16 matrix cores as the six partial products whose weight is >= 2^-16 of the leading one,
//     x*w ~= h_x h_w + (h_x m_w + m_x h_w) + (m_x m_w + h_x l_w + l_x h_w),
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  The dropped terms (m*l, l*m, l*l) are below 2^-24 relative to the
// product -- the size of an fp32 rounding error -- so the result differs from the fp32 MFMA kernel (conv.hip) by a few
// ulp of the accumulated sum (measured in tests/gpu_checks.py check_conv_layers), while six 16x16x32 bf16 MFMAs (6 x 16
// cycles) replace the eight 16x16x4 fp32 MFMAs (8 x 32 cycles) a 32-channel tap needs: 2.7x less matrix-core time.
// Same entry points and epilogue as conv.hip (drba_conv3x3 cfg ids >= the fp32 table); the host autotuner keeps
// whichever kernel is faster for a layer, which is this one where the fp32 kernel is MFMA-bound (large maps:
// GridNet / FeatureNet at full resolution, 4K IFBlocks) and the fp32 one where launches are latency-bound.
//
// GEMM view per workgroup (4 waves): M = pixels of a TH x TW tile (16 consecutive columns per MFMA row block; the
// waves split the rows), N = NT tiles of 16 output channels, K = 32 input channels per tap per chunk.
//   A (activations): the chunk's haloed window is loaded as fp32 (coalesced along x), PReLU pre-activation applied,
//      split, and written to LDS as [plane h/m/l][channel group of 8][pixel][8 x bf16]: a lane's A operand (pixel
//      lane % 16 + tap shift, channels 8*(lane/16)..+7) is one 16-byte read and the 16 lanes of a channel group read
//      256 consecutive bytes.  The next chunk's global loads are in flight under the current chunk's MFMAs.
//   B (weights): split on the host and packed in fragment order ([cout tile][chunk][tap][nt][plane][lane][8 x bf16]):
//      one 16-byte global load per lane per fragment (L1/L2 resident: every workgroup reads the same few KB), fetched
//      one tap ahead of its use.
// Accumulator layout as in conv.hip: lane holds cout lane % 16 for pixels 4*(lane/16)..+3 -> one float4 store along x.
#include "common.hpp"
#include "conv_split.hpp"

#include <string.h>

#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

using namespace drba;

namespace drba_conv_split {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 32;  // channels per chunk = the K of one bf16 MFMA

// MODE 2 (two-term form only): conv3x3 stride 2, pad 1 -- the tile is TH x TW OUTPUT pixels, its window the (2 TH + 1) x
// (2 TW + 1) input pixels under it (each split once, every tap of every output reads the planes at 2 * pixel + tap);
// Cin need not be a multiple of 32: the last chunk's missing channels read as zero through the buffer range check and
// their weights are packed as zero.
// MODE 0: conv3x3 stride 1.  MODE 1: one row phase py (both column phases px) of ConvTranspose2d(k=4, s=2, p=1), a
// 2x2-tap convolution over the same haloed window (tap offsets and phase algebra as in conv.hip).
// PL = 3: the three-term bf16 split above.  PL = 2: the two-term fp16 split (header comment "Two-term form").
// CS (round 6): the four waves of a workgroup split the tile's ROWS 4 / CS ways and its COUT tiles CS ways -- CS = 2: two row
// groups x two cout halves -- instead of rows only.  A wave then works RW = 2 rows against NT = 2 cout tiles where the plain 4 x 32 x 64
// tile works 1 row against 4: the same tile, the same staged window, but a weight fragment a wave fetches feeds FOUR MFMA sets
// instead of two and a wave fetches half of them -- the per-wave weight stream through the vector L1 (every wave fetches its
// fragments itself: 2.07 GB per 64-channel 136 x 240 N8 launch, 49 of the L1's 64 bytes per clock and CU averaged over the whole
// launch, PMC `SQ_INSTS_VMEM_RD`) halves, and so do the accumulators (64 registers instead of 128).
// BP (MODE 1, round 6): BOTH row phases of a transposed-convolution tile in one work item.  The two phases of a window were two
// work items back to back (the second found the window in L2) -- and each loaded, split and staged the whole window again: with one
// 32-channel chunk a work item is ~5 k clocks of staging for ~1.5 k of MFMAs.  Here the staged window serves the four (row phase,
// column phase) accumulator sets (NPX = 4: p = 2 py + px).
template <int MODE_, int RW_, int MW_, int NT_, int PL_ = 3, int CS_ = 1, int BP_ = 0>
struct SplitCfg {
  static constexpr int MODE = MODE_, RW = RW_, MW = MW_, NT = NT_, PL = PL_, CS = CS_;
  static constexpr bool BP = BP_ != 0;
  static_assert(!BP || MODE == 1, "row phases exist in the transposed form only");
  static constexpr int NTT = NT * CS;  // cout tiles of the workgroup's tile (NT: of one wave)
  static constexpr int NTAP = MODE != 1 ? 9 : 4, NPX = MODE != 1 ? 1 : (BP ? 4 : 2);
  static constexpr int TH = (4 / CS) * RW, TW = 16 * MW, NTC = 16 * NTT;
  static constexpr int TR = MODE == 2 ? 2 * TH + 1 : TH + 2, TC = MODE == 2 ? 2 * TW + 1 : TW + 2, NPIX = TR * TC;
  // LDS: [plane PL][group 4][NPIXP][8 x 16 bit]; NPIXP*16 bytes == 64 (mod 256) spreads the four channel groups of a wave
  // read over distinct banks
  static constexpr int NPIXP = ((NPIX + 15) / 16) * 16 + 4;
  static constexpr int LDS_BYTES = PL * 4 * NPIXP * 16;
  static constexpr int ITEMS = NPIX * 4;                 // (pixel, channel group) staging items per chunk
  static constexpr int LIT = (ITEMS + 255) / 256;        // per thread
  static constexpr int FRAG_U4 = NTAP * NTT * PL * 64;   // 16-byte units of packed weights per (cout tile[, phase], chunk)
#ifndef DRBA_SPLIT_BDEPTH_BIG
#define DRBA_SPLIT_BDEPTH_BIG 2
#endif
#ifndef DRBA_SPLIT_BDEPTH2_BIG
#define DRBA_SPLIT_BDEPTH2_BIG 2
#endif
#ifndef DRBA_SPLIT_BDEPTH2_SMALL
#define DRBA_SPLIT_BDEPTH2_SMALL 4
#endif
  // weight fetch distance in steps (>= ~700 cycles of MFMAs); the two-term form's steps are half as long
  static constexpr int BDEPTH = PL == 3 ? ((RW * MW >= 4) ? DRBA_SPLIT_BDEPTH_BIG : 4)
                                        : ((RW * MW >= 4) ? DRBA_SPLIT_BDEPTH2_BIG : DRBA_SPLIT_BDEPTH2_SMALL);
#ifndef DRBA_SPLIT_MINB3
#define DRBA_SPLIT_MINB3 1
#endif
  // workgroups per CU the register allocation aims at: the 4x32x32 tile needs 170 registers, two short of three per CU
  // (CS = 2 at three workgroups per CU: 168 registers with 372 bytes of scratch per lane, 81 against 74 us on the 64-channel N8 layer)
  static constexpr int MINB = (DRBA_SPLIT_MINB3 && MODE == 0 && RW * MW * NT <= 4 && PL == 3 && CS == 1) ? 3 : 2;
};

// fp32 -> (h, m, l) bf16 with round-to-nearest-even at every step; returns the three terms of 2 values packed
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

// Two-term form (PL = 2): x * 2^-kActShift = h + 2^-11 * l with h = fp16(x'), l = fp16((x' - h) * 2^11): the remainder is
// exact in fp32 (13 bits), l keeps 11 of them, so h + 2^-11 l carries 22 bits of x (relative error <= 2^-22; values
// below fp16's normal range keep an absolute error <= 2^-25 * 2^kActShift whether or not denormals are flushed, because
// the remainder then holds the whole value).  l is kept scaled by 2^11 so that it sits in fp16's normal range; the
// products therefore go to two accumulators -- h*h, and (h*l + l*h) whose weight is 2^-11 -- joined in the epilogue.
// l*l (2^-22 of the product) is dropped.  Three v_mfma_f32_16x16x32_f16 per fragment pair instead of six bf16 ones.
// The activations are pre-scaled by 2^-kActShift (undone exactly in the epilogue): finite up to 65504 * 2^kActShift.
constexpr int kActShift = drba::kSplitActShift;
__device__ __forceinline__ void split2_f16(float a, float b, unsigned &h, unsigned &l) {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = (f32x2){a, b} * (1.f / (float)(1 << kActShift));
  const f16x2 hh = __builtin_convertvector(v, f16x2);
  const f32x2 r = (v - __builtin_convertvector(hh, f32x2)) * 2048.f;
  const f16x2 ll = __builtin_convertvector(r, f16x2);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, ll);
}

struct TileCtx {
  int x0, y0, cz, n, py;
};

// Experiment builds only (tools/exp/conv_split_phases.hip defines DRBA_PHASE_CLOCKS): clocks spent per phase, summed over
// the waves' first lanes -- [0] waiting for the previous readers + staging (incl. the wait for the fetched data),
// [1] the MFMA phase, [2] the epilogue, [3] tiles, [4] first-fetch issue to first stage done.
#ifdef DRBA_PHASE_CLOCKS
__device__ long long g_phase[1024 * 4 * 4];  // [workgroup][wave][slot]: written once per wave at the end (no atomics)
#define DRBA_CLK(var) const long long var = (long long)__builtin_readcyclecounter()
#define DRBA_CLK_ADD(slot, a, b) clk_acc[slot] += (b) - (a)
#define DRBA_CLK_INIT long long clk_acc[4] = {0, 0, 0, 0}
#define DRBA_CLK_FLUSH                                                                     \
  if (lane == 0 && blockIdx.x < 1024) {                                                    \
    for (int k_ = 0; k_ < 4; ++k_) g_phase[(blockIdx.x * 4 + wave) * 4 + k_] += clk_acc[k_]; \
  }
#else
#define DRBA_CLK(var)
#define DRBA_CLK_ADD(slot, a, b)
#define DRBA_CLK_INIT
#define DRBA_CLK_FLUSH
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global-memory queue
// (s_waitcnt vmcnt(0)): here that would expose, once per tile, the latency of the epilogue stores just issued.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// RL ("residual from LDS", MODE 0, 32-cout tiles, residual == input, Cin == Cout): the residual of a ResConv IS the input
// tile, and x = h + m + l holds exactly (8 + 8 + 8 mantissa bits, each remainder exact), so the epilogue rebuilds it from
// the three bf16 planes the tile's LAST chunk left in LDS instead of reading it from HBM again.  The chunks of a tile
// are therefore taken in rotated order so that the last one is the chunk of the tile's own 32 output channels.
template <class Cfg, bool PRE, bool RL = false, bool PSH = false>  // PSH: MODE 0 with the PixelShuffle(2) store form (its own instantiation)
__global__ void __launch_bounds__(256, Cfg::MINB)  // at least two workgroups per CU: <= 256 registers
conv_split_mfma(const float *__restrict__ in, const u32x4 *__restrict__ wfrag, const float *__restrict__ bias,
                const float *__restrict__ beta, const float *__restrict__ res, const float *__restrict__ res2,
                float *__restrict__ out, int Cin, int H, int W, int Cout, int act, float post_slope, float pre_slope,
                int n_ctiles, int nbx, int nby, int total, int pixel_shuffle, int Hi, int Wi, unsigned char *status) {
  // H, W: the map the tiles are laid over (MODE 0 / 1: the input = Hi x Wi; MODE 2: the OUTPUT, the input being Hi x Wi)
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MODE = Cfg::MODE, NTAP = Cfg::NTAP, NPX = Cfg::NPX;
  constexpr int RW = Cfg::RW, MW = Cfg::MW, NT = Cfg::NT, TH = Cfg::TH, TW = Cfg::TW, TC = Cfg::TC;
  constexpr int NPIX = Cfg::NPIX, NPIXP = Cfg::NPIXP, LIT = Cfg::LIT, PL = Cfg::PL;
  static_assert(PL == 3 || !RL, "the residual is rebuilt exactly only from the three bf16 planes");
  extern __shared__ __attribute__((aligned(16))) u32x4 tile[];  // [plane][group][NPIXP]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const size_t HW = (size_t)H * W, HWi = (size_t)Hi * Wi;
  const int nchunks = (Cin + CK - 1) / CK;
  constexpr int CS = Cfg::CS, NTT = Cfg::NTT;
  const int row0 = (CS == 1 ? wave : wave % (4 / CS)) * RW;   // this wave's rows of the tile ...
  const int nt0 = CS == 1 ? 0 : (wave / (4 / CS)) * NT;      // ... and its first cout tile (CS > 1: the waves also split the couts)

  // Persistent workgroups: work item i of workgroup b is tile xcd_band(b + i * gridDim.x) -- gridDim.x is a multiple
  // of 8, so a workgroup stays inside its XCD's contiguous band of tiles (cout tile innermost, see conv.hip) -- and
  // the first chunk of the NEXT tile is fetched under the MFMAs of the current one: layers with one or two chunks
  // (Cin = 32, 64) would otherwise expose the whole load -> split -> LDS latency once per tile.
  auto decode = [&](int work) -> TileCtx {
    int t = xcd_band(work, total);
    TileCtx c;
    c.py = 0;
    if (MODE == 1 && !Cfg::BP) {  // the two row phases of a window back to back: the second one finds it in L2
      c.py = t & 1;
      t >>= 1;
    }
    c.cz = t % n_ctiles;
    t /= n_ctiles;
    const int bx = t % nbx;
    t /= nbx;
    c.x0 = bx * TW, c.y0 = (t % nby) * TH, c.n = t / nby;
    return c;
  };

  // ---- staging: item = (pixel p of the haloed window, channel group g); lanes run along p (coalesced along x)
  // Global reads go through buffer descriptors: the address is (scalar base) + (per-lane 32-bit offset), no 64-bit
  // vector address arithmetic, and a lane whose pixel is outside the image reads offset >= num_records -> 0 (the
  // zero padding) without a branch.
  const unsigned img_bytes = (unsigned)((size_t)Cin * HWi * 4);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)wfrag, 0, n_ctiles * (MODE != 1 ? 1 : 4) * nchunks * Cfg::FRAG_U4 * 16, 0x00020000);
  float pre[LIT][8];
  auto fetch = [&](const TileCtx &c, int q) {
    const __amdgpu_buffer_rsrc_t irsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)c.n * Cin * HWi), 0, img_bytes, 0x00020000);
    const int plane = (int)HWi * 4;
#pragma unroll
    for (int it = 0; it < LIT; ++it) {
      const int item = tid + 256 * it;
      const int g = item / NPIX, p = item - g * NPIX;  // group-major
      const int r = p / TC, cc = p - r * TC;
      const int y = (MODE == 2 ? 2 * c.y0 : c.y0) + r - 1, x = (MODE == 2 ? 2 * c.x0 : c.x0) + cc - 1;
      const bool ok = item < Cfg::ITEMS && y >= 0 && y < Hi && x >= 0 && x < Wi;
      const unsigned voff = ok ? (unsigned)(g * 8 * plane + (y * Wi + x) * 4) : 0xffffffffu;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        pre[it][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(irsrc, voff, (q * CK + i) * plane, 0));
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < LIT; ++it) {
      const int item = tid + 256 * it;
      if (item >= Cfg::ITEMS) continue;
      const int g = item / NPIX, slot = g * NPIXP + (item - g * NPIX);
      u32x4 h, mm, l;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a = pre[it][2 * i], b = pre[it][2 * i + 1];
        if (PRE) {
          a = a > 0.f ? a : a * pre_slope;
          b = b > 0.f ? b : b * pre_slope;
        }
        unsigned hh, hm = 0, hl;
        if constexpr (PL == 3) split2(a, b, hh, hm, hl);
        else split2_f16(a, b, hh, hl);
        h[i] = hh, mm[i] = hm, l[i] = hl;
      }
      tile[slot] = h;
      if constexpr (PL == 3) tile[4 * NPIXP + slot] = mm;
      tile[4 * (PL - 1) * NPIXP + slot] = l;
    }
  };

  const bool vec = !PSH && (W & 3) == 0 && (MW & 1) == 0;  // (the regrouped stores pair column blocks 2j, 2j + 1)

  int work = blockIdx.x;
  if (work >= total) return;
  DRBA_CLK_INIT;
  float nf = 0.f;  // PL = 2: NaN once a sum of this lane came out non-finite (nf_fold / nf_report, common.hpp)
#ifdef DRBA_SPLIT_STAGGER  // experiment: put the co-resident workgroups of a CU out of phase
  {
    const int b = blockIdx.x;
    const bool late = DRBA_SPLIT_STAGGER == 1 ? (b & 1) : DRBA_SPLIT_STAGGER == 2 ? (b >= (int)gridDim.x / 2) : DRBA_SPLIT_STAGGER == 4 ? ((b >> 8) & 1) : ((b >> 3) & 1);
    if (late) {
      for (int i = 0; i < DRBA_SPLIT_STAGGER_N; ++i) __builtin_amdgcn_s_sleep(127);
    }
  }
#endif
  TileCtx ctx = decode(work);
  auto first_chunk = [&](const TileCtx &c) -> int { return RL ? (c.cz + 1 == nchunks ? 0 : c.cz + 1) : 0; };
  fetch(ctx, first_chunk(ctx));
  while (true) {
    f32x4 acc[NPX][RW][MW][NT];
    f32x4 acl[PL == 2 ? NPX : 1][RW][MW][NT];  // PL = 2: the sum of the h*l + l*h products (weight 2^-11)
#pragma unroll
    for (int p = 0; p < NPX; ++p)
#pragma unroll
      for (int a = 0; a < RW; ++a)
#pragma unroll
        for (int b = 0; b < MW; ++b)
#pragma unroll
          for (int c = 0; c < NT; ++c) {
            acc[p][a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (PL == 2) acl[p][a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
    const int next = work + (int)gridDim.x;
    TileCtx nctx = ctx;
    if (next < total) nctx = decode(next);
    // per-channel epilogue constants: fetched here so that their latency sits under the tile's MFMAs
    // (MODE 0 with W % 4 == 0 stores in the regrouped layout of the epilogue: lane -> couts (lane >> 3) and + 8 of a tile)
    const bool regroup = MODE != 1 && vec;
    float bs[NT], bt[NT], bs2[NT], bt2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = ctx.cz * Cfg::NTC + (nt0 + nt) * 16 + (regroup ? (lane >> 3) : m), co2 = co + 8;
      bs[nt] = (bias && co < Cout) ? bias[co] : 0.f;
      bt[nt] = (beta && co < Cout) ? beta[co] : 0.f;
      bs2[nt] = (regroup && bias && co2 < Cout) ? bias[co2] : 0.f;
      bt2[nt] = (regroup && beta && co2 < Cout) ? beta[co2] : 0.f;
    }

    for (int qi = 0; qi < nchunks; ++qi) {
      int q = qi;  // the chunk worked on (RL: rotated so that the tile's own channels come last)
      if (RL) {
        q = qi + first_chunk(ctx);
        if (q >= nchunks) q -= nchunks;
      }
      // Weight fragments are fetched D steps (a step = one tap of one cout tile: RW*MW*6 MFMAs) ahead of their use: a
      // chunk's fragments (27 KB per cout tile) do not stay in the 32 KB L1 next to the activations, so a fetch is
      // an L2 round trip.  The first D are independent of LDS and stay in flight across the staging barriers.
      // A "tap slot" ts is a (column phase, tap) pair: 9 for the convolution, 2 x 4 for the transposed one.
      constexpr int NTS = NPX * NTAP, STEPS = NTS * NT, D = Cfg::BDEPTH;
      // byte offset of this (cout tile[, row phase], chunk); the px = 1 block of a transposed conv follows nchunks later
      const int wq = ((MODE != 1 ? ctx.cz : ctx.cz * 4 + 2 * ctx.py) * nchunks + q) * (Cfg::FRAG_U4 * 16);
      const int px_bytes = nchunks * (Cfg::FRAG_U4 * 16);
      auto wload = [&](int step, int pl) -> u32x4 {
        const int p = step / (NTAP * NT), r = step - p * (NTAP * NT);  // r = tap * NT + nt
        const int tap = r / NT, nt = r - tap * NT;                      // packed per tap: NTT fragments of the whole tile
        return __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16, wq + p * px_bytes + ((tap * NTT + nt0 + nt) * PL + pl) * 1024, 0);
      };
      u32x4 bw[D][PL];
#pragma unroll
      for (int d = 0; d < D; ++d)
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) bw[d][pl] = wload(d, pl);
      DRBA_CLK(c_s0);
      lds_barrier();  // every wave is done reading the previous chunk
      stage();
      lds_barrier();
      DRBA_CLK(c_s1);
      DRBA_CLK_ADD(0, c_s0, c_s1);
      // (measured, tools/exp/conv_split_phases.hip: without this request the MFMA phase of the 8x32 tile takes 9.8k clocks
      // instead of 15.9k -- vmcnt retires in issue order, so the weight fragments requested after it wait behind these HBM
      // loads, and the streaming traffic lengthens the L2 round trip of the fragments; issuing it behind the chunk's
      // last weight fetch brought 14.3k and the spills back, a fetch distance of 3 nothing)
#ifndef DRBA_EXP_NOFETCH
      if (qi + 1 < nchunks) fetch(ctx, q + 1 == nchunks ? 0 : q + 1);
      else if (next < total) fetch(nctx, first_chunk(nctx));
#endif

      // activation fragments: ONE buffer; the fragment of (rw, mw) for the next tap slot is read from LDS right after the
      // last MFMAs of the current slot that use it (nt = NT - 1), i.e. RW*MW*6 MFMAs (>= 200 clocks) before its first use
      // -- half the registers of a whole-slot double buffer (48 instead of 96 for the 8x32 tile, which spilled).
      // Transposed conv, tap = 2a + b: window row rw + dro[a] with dro = {1, 0} (py = 0) / {2, 1} (py = 1), window
      // column x + dco[px][b] with dco = {{1, 0}, {2, 1}} (conv.hip).
      u32x4 af[RW][MW][PL];
      auto load_piece = [&](int ts, int rw, int mw) {
        const int p = ts / NTAP, tap = ts - p * NTAP;
        const int ro = MODE != 1 ? tap / 3 : ((tap >> 1) ? 0 : 1) + (Cfg::BP ? (p >> 1) : ctx.py);
        const int co = MODE != 1 ? tap % 3 : ((tap & 1) ? 0 : 1) + (Cfg::BP ? (p & 1) : p);
        const int slot = MODE == 2 ? kq * NPIXP + (2 * (row0 + rw) + ro) * TC + 2 * (mw * 16 + m) + co
                                   : kq * NPIXP + (row0 + rw + ro) * TC + mw * 16 + m + co;
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) af[rw][mw][pl] = tile[4 * pl * NPIXP + slot];
      };
#pragma unroll
      for (int rw = 0; rw < RW; ++rw)
#pragma unroll
        for (int mw = 0; mw < MW; ++mw) load_piece(0, rw, mw);
#pragma unroll
      for (int ts = 0; ts < NTS; ++ts) {
        const int p = ts / NTAP;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int step = ts * NT + nt;
#pragma unroll
          for (int rw = 0; rw < RW; ++rw)
#pragma unroll
            for (int mw = 0; mw < MW; ++mw) {
              if constexpr (PL == 3) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, bw[step % D][0]);
                const bf16x8 bm = __builtin_bit_cast(bf16x8, bw[step % D][1]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, bw[step % D][PL - 1]);
                const bf16x8 ah = __builtin_bit_cast(bf16x8, af[rw][mw][0]);
                const bf16x8 am = __builtin_bit_cast(bf16x8, af[rw][mw][1]);
                const bf16x8 al = __builtin_bit_cast(bf16x8, af[rw][mw][PL - 1]);
                f32x4 c = acc[p][rw][mw][nt];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);  // smallest terms first
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
                acc[p][rw][mw][nt] = c;
              } else {
                const f16x8 bh = __builtin_bit_cast(f16x8, bw[step % D][0]), bl = __builtin_bit_cast(f16x8, bw[step % D][1]);
                const f16x8 ah = __builtin_bit_cast(f16x8, af[rw][mw][0]), al = __builtin_bit_cast(f16x8, af[rw][mw][1]);
                f32x4 c = acl[p][rw][mw][nt];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
                acl[p][rw][mw][nt] = c;
                acc[p][rw][mw][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[p][rw][mw][nt], 0, 0, 0);
              }
#ifndef DRBA_EXP_NOALOAD
              if (nt == NT - 1 && ts + 1 < NTS) load_piece(ts + 1, rw, mw);
#endif
            }
#ifndef DRBA_EXP_NOWLOAD
          if (step + D < STEPS) {
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) bw[step % D][pl] = wload(step + D, pl);
          }
#endif

          __builtin_amdgcn_sched_barrier(0);  // keep the fetch distances as written
        }
      }
      DRBA_CLK(c_m1);
      DRBA_CLK_ADD(1, c_s1, c_m1);
    }
    DRBA_CLK(c_e0);
    if constexpr (PL == 2) {  // join the two sums and undo the activation pre-scale (both exact powers of two)
#pragma unroll
      for (int p = 0; p < NPX; ++p)
#pragma unroll
        for (int a = 0; a < RW; ++a)
#pragma unroll
          for (int b = 0; b < MW; ++b)
#pragma unroll
            for (int c = 0; c < NT; ++c)
            {
              acc[p][a][b][c] = (acc[p][a][b][c] + acl[p][a][b][c] * (1.f / 2048.f)) * (float)(1 << kActShift);
              // the family's overflow report (common.hpp): an operand past fp16's range has made its sums inf / NaN
#pragma unroll
              for (int k = 0; k < 4; ++k) nf = nf_fold(nf, acc[p][a][b][c][k]);
            }
    }

    // ---- epilogue (conv.hip MODE 0): y = acc + bias; ResConv: y = y*beta + res; otherwise y += res (+ res2); then
    // the post activation: act 0 none, 1 LeakyReLU(0.2), 2 PReLU(post_slope), 3 ReLU, 4 tanh(y)*10.  The activation is
    // selected ONCE around the tile loops: selected per element, the inlined copies of the switch (each with a tanhf
    // expansion to jump over) made the epilogue 10k instructions and as slow as the tile's MFMAs.
    if constexpr (MODE != 1 && PSH) {
      // conv + PixelShuffle(2) (GridNet's tail, FusionNet.py:100-103): cout = 4 c13 + 2 si + sj lands at row 2y + si, column
      // 2x + sj of plane c13.  Lanes m and m ^ 1 (sj = 0 / 1: same plane and row) exchange their 4 pixels, after which the even
      // lane holds output columns 2 xb .. 2 xb + 3 and the odd one 2 xb + 4 .. + 7: one 16-byte store each, 128 contiguous bytes
      // per (plane, row) and wave instruction.  W % 4 == 0 and Cout % 4 == 0 (the launcher checks), no residual operands.
      const unsigned obytes = (unsigned)((size_t)Cout * HW * 4);
      const __amdgpu_buffer_rsrc_t orsrc =
          __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)ctx.n * Cout * HW), 0, obytes, 0x00020000);
      auto epilogue_ps = [&](auto post) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int rw = 0; rw < RW; ++rw)
#pragma unroll
            for (int mw = 0; mw < MW; ++mw) {
              f32x4 v = acc[0][rw][mw][nt], o;
              const int co = ctx.cz * Cfg::NTC + (nt0 + nt) * 16 + m;
              const int y = ctx.y0 + row0 + rw;
              const int xb = ctx.x0 + mw * 16 + kq * 4;
              const bool inside = co < Cout && y < H && xb < W;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                v[k] = post(v[k] + bs[nt]);
                o[k] = quad_xor1(v[k]);
              }
              const int c13 = co >> 2, si = (co >> 1) & 1, sj = co & 1;
              const f32x4 qv = sj ? (f32x4){o[2], v[2], o[3], v[3]} : (f32x4){v[0], o[0], v[1], o[1]};
              const unsigned base = (unsigned)(((c13 * (2 * H) + (2 * y + si)) * (2 * W) + 2 * xb) * 4);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, qv), orsrc, inside ? base + 16u * sj : 0xffffffffu, 0, 0);
            }
      };
      switch (act) {
        case 1: epilogue_ps([](float v) { return lrelu02(v); }); break;
        case 2: epilogue_ps([post_slope](float v) { return v > 0.f ? v : post_slope * v; }); break;
        case 3: epilogue_ps([](float v) { return fmaxf(v, 0.f); }); break;
        case 4: epilogue_ps([](float v) { return tanhf(v) * 10.f; }); break;
        default: epilogue_ps([](float v) { return v; }); break;
      }
    } else if constexpr (MODE != 1) {
      const size_t img = (size_t)ctx.n * Cout * HW;
      if (vec) {
        // Whole-line stores and residual reads.  The accumulator leaves lane (m, kq) with 4 consecutive x of ONE cout:
        // stored as it is, a wave instruction touches 16 couts x 64 bytes and no two neighbouring lanes are
        // neighbours in memory (measured: WRITE_SIZE 1.45x the output, the store queue back-pressuring at 64-byte
        // pieces).  Two column blocks (mw = 2j, 2j+1) of a (row, cout tile) are regrouped across lanes -- the halves
        // of each 16-lane row swap one block (couts 0-7 keep block 2j and receive block 2j+1 of the same cout from
        // lane ^ 8, couts 8-15 the other way round), then a lane permutation puts the 8 lanes of a (cout, row) next
        // to each other -- so that lanes 8c..8c+7 hold the 32 consecutive pixels of cout c: every store / residual load
        // instruction moves 8 whole 128-byte row segments, first couts 0-7 of the tile (A), then couts 8-15 (B).
        // (the lane-derived indices are formed here, from an opaque copy of the lane id: hoisted out of the tile loop
        // they would stay live across the MFMA loop, where every register counts)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int c8 = ln >> 3, blk = (ln >> 2) & 1, q4 = ln & 3;
        const int src = q4 * 16 + blk * 8 + c8;  // lane holding (cout c8 | c8 + 8, block blk, pixel group q4) after the swap
        const bool hi = (ln & 8) != 0;
        // Residual loads and stores go through buffer descriptors: a lane outside the image (or past Cout) gets an offset
        // beyond num_records -- its load returns 0, its store is dropped -- so the epilogue is straight-line code and the
        // compiler's s_waitcnt placement is exact.  With branches around them every path merge forced s_waitcnt vmcnt(0),
        // i.e. each store's full round trip was waited for before the next piece (vmcnt counts loads AND stores in issue
        // order); for the same reason pass 1 issues every residual load of the tile before pass 2 issues the first store.
        const unsigned obytes = (unsigned)((size_t)Cout * HW * 4);
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(out + img), 0, obytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rrsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)((res ? res : out) + img), 0, res ? obytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r2rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)((res2 ? res2 : out) + img), 0, res2 ? obytes : 0u, 0x00020000);
        auto offs = [&](int nt, int rw, int j, unsigned &oa, unsigned &ob) {
          const int coA = ctx.cz * Cfg::NTC + (nt0 + nt) * 16 + c8;
          const int y = ctx.y0 + row0 + rw;
          const int xb = ctx.x0 + (2 * j + blk) * 16 + q4 * 4;
          const bool in_img = y < H && xb < W;
          const unsigned base = (unsigned)(((coA * H + y) * W + xb) * 4);
          oa = (in_img && coA < Cout) ? base : 0xffffffffu;
          ob = (in_img && coA + 8 < Cout) ? base + 8u * (unsigned)HW * 4u : 0xffffffffu;
        };
        // residual pieces of cout tile nt + 1 are requested before the stores of tile nt are issued (ring of two)
        u32x4 ra[2][RW][MW / 2], rb[2][RW][MW / 2];  // (a second residual, GridNet's lateral sums, is read in pass 2)
        auto load_res = [&](int nt) {
#pragma unroll
          for (int rw = 0; rw < RW; ++rw)
#pragma unroll
            for (int j = 0; j < MW / 2; ++j) {
              unsigned oa, ob;
              offs(nt, rw, j, oa, ob);
              ra[nt & 1][rw][j] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, oa, 0, 0);
              rb[nt & 1][rw][j] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ob, 0, 0);
            }
        };
        if (!RL && (res || res2)) load_res(0);
        // KIND: which operands the layer has -- 0 none, 1 one residual, 2 two residuals (GridNet's lateral sums), 3 the ResConv
        // form y * beta + x.  Like the activation it is selected ONCE around the tile's loops: tested per piece (three uniform
        // branches on `beta`, `res`, `res2`, their masks spilled to VGPR lanes and read back) the epilogue of the 4 x 32 x 64
        // tile took 7.9k clocks instead of 5.3k (tools/exp/conv_split_phases.hip).
        auto epilogue = [&](auto kind_, auto post) {
          constexpr int KIND = decltype(kind_)::value;
          constexpr bool kBeta = KIND == 3, kRes = !RL && KIND >= 1, kRes2 = !RL && KIND == 2;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float bsA = bs[nt], bsB = bs2[nt], btA = bt[nt], btB = bt2[nt];
            if (kRes && nt + 1 < NT) load_res(nt + 1);
#pragma unroll
            for (int rw = 0; rw < RW; ++rw)
#pragma unroll
              for (int j = 0; j < MW / 2; ++j) {
                const f32x4 v0 = acc[0][rw][2 * j][nt], v1 = acc[0][rw][2 * j + 1][nt];
                f32x4 a, b;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const float got = dpp_f32<0x128>(hi ? v0[k] : v1[k]);  // row_ror:8 = lane ^ 8 inside a row of 16: a VALU move, not an LDS-crossbar trip
                  a[k] = __shfl(hi ? got : v0[k], src, 64);
                  b[k] = __shfl(hi ? v1[k] : got, src, 64);
                }
                unsigned oa, ob;
                offs(nt, rw, j, oa, ob);
                f32x4 xa = (f32x4){0.f, 0.f, 0.f, 0.f}, xb_ = xa, xa2 = xa, xb2 = xa;
                if (kRes) {
                  xa = __builtin_bit_cast(f32x4, ra[nt & 1][rw][j]);
                  xb_ = __builtin_bit_cast(f32x4, rb[nt & 1][rw][j]);
                }
                if (kRes2) {
                  xa2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2rsrc, oa, 0, 0));
                  xb2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2rsrc, ob, 0, 0));
                }
                if (RL) {
                  // channel nt*16 + c8 (A) / + 8 (B) of the chunk in LDS = group 2*nt (A) / 2*nt + 1 (B), element c8;
                  // the pixel's slot in the haloed window is (row + 1, column + 1)
                  const unsigned short *pl = reinterpret_cast<const unsigned short *>(tile);
                  const int px0 = (row0 + rw + 1) * TC + (2 * j + blk) * 16 + q4 * 4 + 1;
#pragma unroll
                  for (int k = 0; k < 4; ++k) {
                    const int sa = ((2 * nt) * NPIXP + px0 + k) * 8 + c8, sb = sa + NPIXP * 8;
                    const float ha = __uint_as_float((unsigned)pl[sa] << 16), hb = __uint_as_float((unsigned)pl[sb] << 16);
                    const float ma = __uint_as_float((unsigned)pl[4 * NPIXP * 8 + sa] << 16);
                    const float mb = __uint_as_float((unsigned)pl[4 * NPIXP * 8 + sb] << 16);
                    const float la = __uint_as_float((unsigned)pl[8 * NPIXP * 8 + sa] << 16);
                    const float lb = __uint_as_float((unsigned)pl[8 * NPIXP * 8 + sb] << 16);
                    xa[k] = ha + (ma + la);
                    xb_[k] = hb + (mb + lb);
                  }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  float ua = a[k] + bsA, ub = b[k] + bsB;
                  if (kBeta) {
                    ua = ua * btA + xa[k];
                    ub = ub * btB + xb_[k];
                  } else {
                    if (RL || kRes) ua = ua + xa[k], ub = ub + xb_[k];
                    if (kRes2) ua = ua + xa2[k], ub = ub + xb2[k];
                  }
                  a[k] = post(ua);
                  b[k] = post(ub);
                }
#ifdef DRBA_EXP_NOSTORE  // experiment: the stores are compiled in but never executed (H is never negative)
                if (H >= 0) continue;
#endif
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), orsrc, oa, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, b), orsrc, ob, 0, 0);
              }
          }
        };
        // activations 0-2 are one form, v > 0 ? v : slope * v with slope 1 / 0.2 / post_slope (exact: v * 1 == v); ReLU and
        // tanh are their own instantiations.  ReLU as torch.relu and the other families' fmaxf(v, 0) define it: +0 for every
        // v <= 0 (slope 0 would hand on -0.0, and NaN for -inf), a NaN kept.
        const float slope = act == 1 ? 0.2f : act == 2 ? post_slope : 1.f;
        auto with_act = [&](auto kind_) {
          if (act == 4) epilogue(kind_, [](float v) { return tanhf(v) * 10.f; });
          else if (act == 3) epilogue(kind_, [](float v) { return v > 0.f ? v : (v != v ? v : 0.f); });
          else epilogue(kind_, [slope](float v) { return v > 0.f ? v : slope * v; });
        };
        if (beta) with_act(std::integral_constant<int, 3>{});
        else if (!RL && res2) with_act(std::integral_constant<int, 2>{});  // (a missing first residual reads as zero: rrsrc has no records)
        else if (RL || res) with_act(std::integral_constant<int, 1>{});
        else with_act(std::integral_constant<int, 0>{});
      } else {
        // W not a multiple of 4: element-wise stores from the accumulator layout (lane = cout m, pixels 4*kq..+3)
        auto epilogue = [&](auto post) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int rw = 0; rw < RW; ++rw)
#pragma unroll
              for (int mw = 0; mw < MW; ++mw) {
                const f32x4 v = acc[0][rw][mw][nt];
                const int co = ctx.cz * Cfg::NTC + (nt0 + nt) * 16 + m;
                const int y = ctx.y0 + row0 + rw;
                const int xb = ctx.x0 + mw * 16 + kq * 4;
                if (co >= Cout || y >= H || xb >= W) continue;
                const size_t idx = img + ((size_t)co * H + y) * W + xb;
                if ((W & 3) == 0) {  // (odd MW tiles on aligned maps: one 16-byte store -- and residual load -- per lane)
                  f32x4 o, r1 = (f32x4){0.f, 0.f, 0.f, 0.f}, r2 = r1;
                  if (res) r1 = *reinterpret_cast<const f32x4 *>(res + idx);
                  if (res2) r2 = *reinterpret_cast<const f32x4 *>(res2 + idx);
#pragma unroll
                  for (int k = 0; k < 4; ++k) {
                    float u = v[k] + bs[nt];
                    if (beta) u = u * bt[nt] + r1[k];
                    else {
                      if (res) u = u + r1[k];
                      if (res2) u = u + r2[k];
                    }
                    o[k] = post(u);
                  }
                  *reinterpret_cast<f32x4 *>(out + idx) = o;
                  continue;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  if (xb + k >= W) continue;
                  float u = v[k] + bs[nt];
                  if (beta) u = u * bt[nt] + res[idx + k];
                  else {
                    if (res) u = u + res[idx + k];
                    if (res2) u = u + res2[idx + k];
                  }
                  out[idx + k] = post(u);
                }
              }
        };
        switch (act) {
          case 1: epilogue([](float v) { return lrelu02(v); }); break;
          case 2: epilogue([post_slope](float v) { return v > 0.f ? v : post_slope * v; }); break;
          case 3: epilogue([](float v) { return fmaxf(v, 0.f); }); break;
          case 4: epilogue([](float v) { return tanhf(v) * 10.f; }); break;
          default: epilogue([](float v) { return v; }); break;
        }
      }
    } else {
      // transposed conv, row phase py, both column phases: lane (cout m, column group kq) holds the input columns
      // xb..xb+3 for px = 0 (v) and px = 1 (v1), i.e. the 8 consecutive output columns 2*xb .. 2*xb+7 of row 2y+py
      // (conv.hip MODE 1; bias only)
      const int Hd = 2 * H, Wd = 2 * W;
      // Stores through a buffer descriptor over this image's output (32-bit byte offsets; the launcher refuses outputs of
      // 4 GB per image): a lane outside the map or past Cout gets an offset beyond num_records and its store is dropped, so the
      // loop is straight-line code -- with `continue`s around plain stores every merge point carried an s_waitcnt vmcnt(0)
      // and the 64-bit address arithmetic of every piece (the epilogue was 4.3k of the 32 -> 20 tile's 12.2k clocks).  The
      // PixelShuffle form is selected once around the loops.
      const unsigned obytes = (unsigned)((size_t)Cout * Hd * Wd * 4);
      const __amdgpu_buffer_rsrc_t orsrc =
          __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)ctx.n * Cout * Hd * Wd), 0, obytes, 0x00020000);
      const bool whole = (W & 3) == 0;  // every lane's 4 input columns are inside the map or none is
      auto epilogue_t = [&](auto ps_) {
        constexpr bool PS = decltype(ps_)::value;
#pragma unroll
        for (int pyi = 0; pyi < (Cfg::BP ? 2 : 1); ++pyi)
#pragma unroll
        for (int rw = 0; rw < RW; ++rw)
#pragma unroll
          for (int mw = 0; mw < MW; ++mw)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              f32x4 v = acc[Cfg::BP ? 2 * pyi : 0][rw][mw][nt], v1 = acc[Cfg::BP ? 2 * pyi + 1 : NPX - 1][rw][mw][nt];
              const int co = ctx.cz * Cfg::NTC + (nt0 + nt) * 16 + m;
              const int y = ctx.y0 + row0 + rw;
              const int xb = ctx.x0 + mw * 16 + kq * 4;
              const int oy = 2 * y + (Cfg::BP ? pyi : ctx.py);
              const float b0 = bs[nt];
              const bool inside = co < Cout && y < H && xb < W;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                v[k] += b0;
                v1[k] += b0;
              }
              if constexpr (!PS) {
                const unsigned base = (unsigned)(((co * Hd + oy) * Wd + 2 * xb) * 4);
                if (whole) {
                  const unsigned o = inside ? base : 0xffffffffu;
                  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){v[0], v1[0], v[1], v1[1]}), orsrc, o, 0, 0);
                  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){v[2], v1[2], v[3], v1[3]}), orsrc, o, 16, 0);
                } else {
#pragma unroll
                  for (int k = 0; k < 4; ++k) {
                    const unsigned o = (inside && xb + k < W) ? base + 8u * k : 0xffffffffu;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[k]), orsrc, o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1[k]), orsrc, o, 4, 0);
                  }
                }
              } else {
                // + PixelShuffle(2): cout = 4*c13 + 2*si + sj lands at row 2*oy+si, column 4i + 2px + sj of plane c13.
                // Lanes m and m^1 (sj = 0 / 1, same c13 and si) exchange their values, after which each holds the 4
                // consecutive columns 4i..4i+3 for every i; the even lane stores i = xb, xb+1, the odd lane xb+2, xb+3.
                const int c13 = co >> 2, si = (co >> 1) & 1, sj = co & 1;
                f32x4 o0, o1;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  o0[k] = quad_xor1(v[k]);
                  o1[k] = quad_xor1(v1[k]);
                }
                const unsigned base = (unsigned)(((c13 * (2 * Hd) + (2 * oy + si)) * (2 * Wd) + 4 * xb) * 4);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                  const int k = sj * 2 + kk;  // even lane: input columns xb, xb+1; odd lane: xb+2, xb+3
                  const f32x4 qv = sj ? (f32x4){o0[k], v[k], o1[k], v1[k]} : (f32x4){v[k], o0[k], v1[k], o1[k]};
                  const unsigned o = (inside && xb + k < W) ? base + 16u * k : 0xffffffffu;
                  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, qv), orsrc, o, 0, 0);
                }
              }
            }
      };
      if (pixel_shuffle) epilogue_t(std::true_type{});
      else epilogue_t(std::false_type{});
    }
    DRBA_CLK(c_e1);
    DRBA_CLK_ADD(2, c_e0, c_e1);
    DRBA_CLK_ADD(3, 0, 1);
    if (next >= total) {
      DRBA_CLK_FLUSH;
      break;
    }
    work = next;
    ctx = nctx;
  }
  if constexpr (PL == 2) nf_report(status, DRBA_STATUS_CONV_SPLIT, nf);
#endif
}

// ------------------------------------------------------------------------------------------ host side
//                  M  RW MW NT
using S0 = SplitCfg<0, 1, 2, 2>;  // 4x32 px x 32 cout
using S1 = SplitCfg<0, 1, 2, 4>;  // 4x32 px x 64 cout
using S2 = SplitCfg<0, 1, 2, 6>;  // 4x32 px x 96 cout
using S3 = SplitCfg<0, 1, 4, 2>;  // 4x64 px x 32 cout
using S4 = SplitCfg<0, 2, 2, 2>;  // 8x32 px x 32 cout
constexpr int kNum = 5;
using T0 = SplitCfg<1, 1, 2, 2>;  // transposed: 4x32 input px x 32 cout (both column phases)
using T1 = SplitCfg<1, 1, 2, 4>;  // 4x32 x 64 (8x32 and 4x64 tiles need > 256 registers with two phases' accumulators)
constexpr int kNumT = 2;
// the same tiles in the two-term fp16 form: ids kNum.. / kNumT.. of this file, exposed by conv.hip as the LAST family of
// the cfg tables (drba_conv3x3_cfg_family 4) so that a caller opts into them
using F0 = SplitCfg<0, 1, 2, 2, 2>;
using F1 = SplitCfg<0, 1, 2, 4, 2>;
using F2 = SplitCfg<0, 1, 2, 6, 2>;
using F3 = SplitCfg<0, 1, 4, 2, 2>;
using F4 = SplitCfg<0, 2, 2, 2, 2>;
using G0 = SplitCfg<1, 1, 2, 2, 2>;
using G1 = SplitCfg<1, 1, 2, 4, 2>;
// stride 2 (two-term form only): ids 2 * kNum .. 2 * kNum + kNumX - 1 of this file
using X0 = SplitCfg<2, 1, 2, 2, 2>;  // 4x32 output px x 32 cout (window 9 x 65)
using X1 = SplitCfg<2, 1, 2, 4, 2>;  // 4x32 x 64
using X2 = SplitCfg<2, 1, 1, 2, 2>;  // 4x16 x 32 (window 9 x 33)
using X3 = SplitCfg<2, 1, 1, 4, 2>;  // 4x16 x 64
constexpr int kNumX = 4;
// appended behind them (round 6; stride 1 again): the 4 x 32 x 64 tile with the waves splitting rows AND couts (CS = 2)
using Y0 = SplitCfg<0, 2, 2, 2, 2, 2>;  // 4x32 px x 64 cout
using Y1 = SplitCfg<0, 2, 2, 3, 2, 2>;  // 4x32 px x 96 cout
using Y2 = SplitCfg<2, 2, 2, 2, 2, 2>;  // stride 2: 4x32 output px x 64 cout (window 9 x 65, as X1)
using Y3 = SplitCfg<2, 2, 1, 2, 2, 2>;  // stride 2: 4x16 output px x 64 cout (window 9 x 33, as X3: a fragment feeds 2 MFMA sets instead of 1)
using Y4 = SplitCfg<2, 2, 1, 3, 2, 2>;  // stride 2: 4x16 output px x 96 cout
using Y5 = SplitCfg<0, 2, 1, 3, 2, 2>;  // stride 1: 4x16 px x 96 cout (one staged window for all couts of a 96-channel layer)
constexpr int kNumY = 6;  // (the same tile with 128 couts: 32.5 against 32.5 us on the 128-channel 34 x 60 layer -- not built)
struct Info {
  int NT, NTC, frag_u4, PL;
};
template <class C>
constexpr Info info() {
  return {C::NTT, C::NTC, C::FRAG_U4, C::PL};  // (NT here: the cout tiles of a packed tile)
}
const Info kInfo[2 * kNum + kNumX + kNumY] = {info<S0>(), info<S1>(), info<S2>(), info<S3>(), info<S4>(),
                                              info<F0>(), info<F1>(), info<F2>(), info<F3>(), info<F4>(),
                                              info<X0>(), info<X1>(), info<X2>(), info<X3>(), info<Y0>(), info<Y1>(), info<Y2>(), info<Y3>(), info<Y4>(), info<Y5>()};
using Z0 = SplitCfg<1, 2, 2, 2, 2, 2>;  // transposed, two-term, the waves split rows and couts: 4x32 input px x 64 cout (round 6)
using Z1 = SplitCfg<1, 2, 2, 1, 2, 2>;  // ... x 32 cout
using Z2 = SplitCfg<1, 2, 2, 1, 2, 2, 1>;  // Z1 with both row phases per work item (BP)
using Z3 = SplitCfg<1, 1, 2, 2, 2, 1, 1>;  // G0 with both row phases per work item
using Z4 = SplitCfg<1, 2, 1, 2, 2, 2, 1>;  // 4x16 input px x 64 cout, rows and couts split across the waves, both row phases
constexpr int kNumZ = 5;
const Info kInfoT[2 * kNumT + kNumZ] = {info<T0>(), info<T1>(), info<G0>(), info<G1>(), info<Z0>(), info<Z1>(), info<Z2>(), info<Z3>(), info<Z4>()};

template <class Cfg, bool PRE, bool RL = false, bool PSH = false>
hipError_t lds_limit() {
  if (Cfg::LDS_BYTES <= 64 * 1024) return hipSuccess;
  return max_dynamic_lds(reinterpret_cast<const void *>(conv_split_mfma<Cfg, PRE, RL, PSH>), Cfg::LDS_BYTES);
}
// the tiles that carry the PixelShuffle store form of MODE 0 (one more instantiation each): the two-term 4 x 32 x 64 ones
template <class Cfg>
constexpr bool kHasShuffle = Cfg::MODE == 0 && Cfg::PL == 2 && Cfg::RW * Cfg::MW == 2 * Cfg::CS && Cfg::NTT == 4 && Cfg::MW == 2;

static int resident_per_cu(const void *kernel, int lds_bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, int> known;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(kernel, dev);
  auto it = known.find(key);
  if (it != known.end()) return it->second;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, lds_bytes) != hipSuccess || occ < 1) occ = 1;
  return known[key] = occ > 3 ? 3 : occ;
}

template <class Cfg>
int launch(const float *in, const float *wpk, const float *bias, const float *beta, const float *res, const float *res2,
           float *out, int N, int Cin, int H, int W, int Cout, int act, float post_slope, int pre_act, float pre_slope,
           int pixel_shuffle, hipStream_t s) {
  const int n_ct = (Cout + Cfg::NTC - 1) / Cfg::NTC;
  const int Hi = H, Wi = W;  // the input map; the tiles are laid over the OUTPUT of a stride-2 layer
  if (Cfg::MODE == 2) H = (Hi - 1) / 2 + 1, W = (Wi - 1) / 2 + 1;
  const int nbx = (W + Cfg::TW - 1) / Cfg::TW, nby = (H + Cfg::TH - 1) / Cfg::TH;
  const long long total = (long long)nbx * nby * N * n_ct * ((Cfg::MODE != 1 || Cfg::BP) ? 1 : 2);  // x2: row phases as work items
  if (total >= (1ll << 31)) return DRBA_EUNSUPPORTED;
  const u32x4 *wf = reinterpret_cast<const u32x4 *>(wpk);
  unsigned char *status = Cfg::PL == 2 ? status_bytes() : nullptr;
  auto go = [&](auto kernel, hipError_t lds_ok) -> int {
    if (lds_ok != hipSuccess) return DRBA_ELAUNCH;
    // persistent grid: the workgroups the 256 CUs hold AT ONCE (registers and LDS of this instantiation, asked once), a
    // multiple of 8.  A grid above the residency runs its surplus workgroups as a second round on half-empty CUs: the
    // 227-register 4 x 32 x 64 tile was launched 3 per CU by its LDS size alone while 2 fit (64 ch 136x240 N8: 95 -> 86 us,
    // tools/exp/split_per_cu.sh).
    // (asked once per kernel instantiation and device: the PRE / RL variants of a Cfg decay to one function-pointer type, so
    // a static inside this lambda would be shared by all of them -- and by every GPU of the process)
    const int resident = resident_per_cu(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES);
    const int per_cu = env_int("DRBA_SPLIT_PER_CU", resident);
    long long grid = 256ll * per_cu;
    if (grid > total) grid = (total + 7) / 8 * 8;
    DRBA_LAUNCH(kernel, dim3((unsigned)grid), dim3(256), Cfg::LDS_BYTES, s, in, wf, bias, beta, res, res2, out, Cin, H, W, Cout,
                act, post_slope, pre_slope, n_ct, nbx, nby, (int)total, pixel_shuffle, Hi, Wi, status);
    return DRBA_OK;
  };
  int rc;
  if (Cfg::MODE == 0 && pixel_shuffle) {
    if constexpr (kHasShuffle<Cfg>) {
      if (res || res2 || beta || pre_act || (W & 3) || (Cout & 3)) return DRBA_EUNSUPPORTED;
      rc = go(conv_split_mfma<Cfg, false, false, true>, lds_limit<Cfg, false, false, true>());
    } else {
      return DRBA_EUNSUPPORTED;
    }
  } else if constexpr (Cfg::MODE == 0 && Cfg::NTC == CK && Cfg::PL == 3) {
    // ResConv shape (the residual is the layer's own input): rebuilt from the bf16 planes in LDS, no second read
    const bool rl = res && res == in && !res2 && !pre_act && Cin == Cout && (W & 3) == 0;
    rc = rl ? go(conv_split_mfma<Cfg, false, true>, lds_limit<Cfg, false, true>())
            : (pre_act ? go(conv_split_mfma<Cfg, true>, lds_limit<Cfg, true>()) : go(conv_split_mfma<Cfg, false>, lds_limit<Cfg, false>()));
  } else {
    rc = pre_act ? go(conv_split_mfma<Cfg, true>, lds_limit<Cfg, true>()) : go(conv_split_mfma<Cfg, false>, lds_limit<Cfg, false>());
  }
  if (rc != DRBA_OK) return rc;
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // namespace drba_conv_split

namespace drba {

// ids 0 .. kNum-1: three-term bf16; kNum .. 2*kNum-1: the same tiles in the two-term fp16 form; then the stride-2 tiles
// (two-term form, any Cin)
int conv_split_num_cfgs() { return drba_conv_split::kNum; }
int conv_split_f16_first() { return drba_conv_split::kNum; }
int conv_split_s2_first() { return 2 * drba_conv_split::kNum; }
int conv_split_s2_num_cfgs() { return drba_conv_split::kNumX; }
int conv_split_cs_first() { return 2 * drba_conv_split::kNum + drba_conv_split::kNumX; }
int conv_split_cs_num_cfgs() { return drba_conv_split::kNumY; }
int conv_split_cfg_stride(int id) { return (id >= 2 * drba_conv_split::kNum && id != 14 && id != 15 && id < 19) ? 2 : 1; }  // (Y0, Y1 = ids 14, 15 are the stride-1 tiles behind the X's)

bool conv_split_supports(int Cin, int Cout, int id) {
  using namespace drba_conv_split;
  if (id >= 2 * kNum + kNumX)  // two-term, CS = 2: the stride-1 tiles need whole chunks, the stride-2 tile takes any Cin
    return id < 2 * kNum + kNumX + kNumY && Cin > 0 && Cout > 0 && (conv_split_cfg_stride(id) == 2 || Cin % CK == 0);
  if (id >= 2 * kNum) return Cin > 0 && Cout > 0;
  return id >= 0 && Cin > 0 && Cout > 0 && Cin % CK == 0;
}

size_t conv_split_packed_floats(int Cin, int Cout, int id) {
  if (!conv_split_supports(Cin, Cout, id)) return 0;
  const drba_conv_split::Info &c = drba_conv_split::kInfo[id];
  const size_t n_ct = (Cout + c.NTC - 1) / c.NTC, nch = (Cin + drba_conv_split::CK - 1) / drba_conv_split::CK;
  return n_ct * nch * c.frag_u4 * 4;
}

// packed (16-byte units): [cout tile][chunk][tap][nt][plane h/m/l][lane] = 8 bf16 of
//   w[cz*NTC + nt*16 + (lane & 15)][q*32 + 8*(lane >> 4) + i][tap], i = 0..7, zero outside Cout
int conv_split_pack(const float *w, float *packed, int Cin, int Cout, int id) {
  using namespace drba_conv_split;
  if (!w || !packed || !conv_split_supports(Cin, Cout, id)) return DRBA_EINVAL;
  const Info &c = kInfo[id];
  if (c.PL == 2 && !two_term_weights_ok(w, (size_t)Cout * Cin * 9)) return DRBA_EUNSUPPORTED;
  const int n_ct = (Cout + c.NTC - 1) / c.NTC, nch = (Cin + CK - 1) / CK;
  memset(packed, 0, sizeof(float) * conv_split_packed_floats(Cin, Cout, id));
  unsigned short *dst = reinterpret_cast<unsigned short *>(packed);
  for (int cz = 0; cz < n_ct; ++cz)
    for (int q = 0; q < nch; ++q)
      for (int tap = 0; tap < 9; ++tap)
        for (int nt = 0; nt < c.NT; ++nt)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = cz * c.NTC + nt * 16 + (lane & 15);
            if (co >= Cout) continue;
            for (int i = 0; i < 8; ++i) {
              const int ci = q * CK + 8 * (lane >> 4) + i;
              if (ci >= Cin) continue;  // (stride-2 ids: the last chunk of a ragged Cin)
              unsigned short term[3];
              split_weight_terms(w[((size_t)co * Cin + ci) * 9 + tap], c.PL, term);
              for (int pl = 0; pl < c.PL; ++pl) {
                const size_t unit = ((((size_t)cz * nch + q) * 9 + tap) * c.NT + nt) * c.PL + pl;
                dst[(unit * 64 + lane) * 8 + i] = term[pl];
              }
            }
          }
  return DRBA_OK;
}

int conv_split_launch(int id, const float *in, const float *packed_w, const float *bias, const float *beta,
                      const float *residual, const float *residual2, float *out, int N, int Cin, int H, int W, int Cout,
                      int act, float post_slope, int pre_act, float pre_slope, void *stream, int pixel_shuffle) {
  using namespace drba_conv_split;
  if (!conv_split_supports(Cin, Cout, id)) return DRBA_EUNSUPPORTED;
  if ((size_t)Cin * H * W * 4 >= (1ull << 31)) return DRBA_EUNSUPPORTED;  // 32-bit byte offsets inside an image
  if ((size_t)Cout * H * W * 4 >= (1ull << 31)) return DRBA_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
#define DRBA_CASE(ID, T) \
  case ID:               \
    return launch<T>(in, packed_w, bias, beta, residual, residual2, out, N, Cin, H, W, Cout, act, post_slope, pre_act, \
                     pre_slope, pixel_shuffle, s);
  switch (id) {
    DRBA_CASE(0, S0)
    DRBA_CASE(1, S1)
    DRBA_CASE(2, S2)
    DRBA_CASE(3, S3)
    DRBA_CASE(4, S4)
    DRBA_CASE(5, F0)
    DRBA_CASE(6, F1)
    DRBA_CASE(7, F2)
    DRBA_CASE(8, F3)
    DRBA_CASE(9, F4)
    DRBA_CASE(10, X0)
    DRBA_CASE(11, X1)
    DRBA_CASE(12, X2)
    DRBA_CASE(13, X3)
    DRBA_CASE(14, Y0)
    DRBA_CASE(15, Y1)
    DRBA_CASE(16, Y2)
    DRBA_CASE(17, Y3)
    DRBA_CASE(18, Y4)
    DRBA_CASE(19, Y5)
  }
#undef DRBA_CASE
  return DRBA_EUNSUPPORTED;
}

// ---- transposed convolution (ConvTranspose2d k=4, s=2, p=1), cfg ids after conv.hip's fp32 deconv table
int deconv_split_num_cfgs() { return drba_conv_split::kNumT; }

int deconv_split_f16_first() { return drba_conv_split::kNumT; }

int deconv_split_total_cfgs() { return 2 * drba_conv_split::kNumT + drba_conv_split::kNumZ; }  // ids >= 2 kNumT: two-term, CS = 2
bool deconv_split_supports(int Cin, int Cout, int id) {
  return id >= 0 && id < deconv_split_total_cfgs() && Cin > 0 && Cout > 0 && Cin % drba_conv_split::CK == 0;
}

size_t deconv_split_packed_floats(int Cin, int Cout, int id) {
  if (!deconv_split_supports(Cin, Cout, id)) return 0;
  const drba_conv_split::Info &c = drba_conv_split::kInfoT[id];
  const size_t n_ct = (Cout + c.NTC - 1) / c.NTC, nch = Cin / drba_conv_split::CK;
  return n_ct * 4 * nch * c.frag_u4 * 4;
}

// w: [Cin, Cout, 4, 4].  packed (16-byte units): [cout tile][phase = 2*py + px][chunk][tap = 2a + b][nt][plane][lane],
// ky = py ? (a ? 2 : 0) : (a ? 3 : 1), kx likewise from (px, b) -- the phase algebra of conv.hip's drba_deconv4x4_pack
int deconv_split_pack(const float *w, float *packed, int Cin, int Cout, int id) {
  using namespace drba_conv_split;
  if (!w || !packed || !deconv_split_supports(Cin, Cout, id)) return DRBA_EINVAL;
  const Info &c = kInfoT[id];
  if (c.PL == 2 && !two_term_weights_ok(w, (size_t)Cin * Cout * 16)) return DRBA_EUNSUPPORTED;
  const int n_ct = (Cout + c.NTC - 1) / c.NTC, nch = Cin / CK;
  memset(packed, 0, sizeof(float) * deconv_split_packed_floats(Cin, Cout, id));
  unsigned short *dst = reinterpret_cast<unsigned short *>(packed);
  for (int cz = 0; cz < n_ct; ++cz)
    for (int phase = 0; phase < 4; ++phase) {
      const int py = phase >> 1, px = phase & 1;
      for (int q = 0; q < nch; ++q)
        for (int tap = 0; tap < 4; ++tap) {
          const int a = tap >> 1, b = tap & 1;
          const int ky = py ? (a ? 2 : 0) : (a ? 3 : 1);
          const int kx = px ? (b ? 2 : 0) : (b ? 3 : 1);
          for (int nt = 0; nt < c.NT; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
              const int co = cz * c.NTC + nt * 16 + (lane & 15);
              if (co >= Cout) continue;
              for (int i = 0; i < 8; ++i) {
                const int ci = q * CK + 8 * (lane >> 4) + i;
                unsigned short term[3];
                split_weight_terms(w[(((size_t)ci * Cout + co) * 4 + ky) * 4 + kx], c.PL, term);
                for (int pl = 0; pl < c.PL; ++pl) {
                  const size_t unit = ((((((size_t)cz * 4 + phase) * nch + q) * 4 + tap) * c.NT + nt) * c.PL + pl);
                  dst[(unit * 64 + lane) * 8 + i] = term[pl];
                }
              }
            }
        }
    }
  return DRBA_OK;
}

int deconv_split_launch(int id, const float *in, const float *packed_w, const float *bias, float *out, int N, int Cin, int H,
                        int W, int Cout, int pixel_shuffle, int pre_act, float pre_slope, void *stream) {
  using namespace drba_conv_split;
  if (!deconv_split_supports(Cin, Cout, id)) return DRBA_EUNSUPPORTED;
  if ((size_t)Cin * H * W * 4 >= (1ull << 31)) return DRBA_EUNSUPPORTED;  // 32-bit byte offsets inside an image
  if ((size_t)Cout * 4 * H * W * 4 >= (1ull << 32) - 64) return DRBA_EUNSUPPORTED;  // ... and inside an output image (buffer stores)
  hipStream_t s = (hipStream_t)stream;
#define DRBA_CASE(ID, T) \
  case ID:               \
    return launch<T>(in, packed_w, bias, nullptr, nullptr, nullptr, out, N, Cin, H, W, Cout, 0, 0.f, pre_act, pre_slope, \
                     pixel_shuffle, s);
  switch (id) {
    DRBA_CASE(0, T0)
    DRBA_CASE(1, T1)
    DRBA_CASE(2, G0)
    DRBA_CASE(3, G1)
    DRBA_CASE(4, Z0)
    DRBA_CASE(5, Z1)
    DRBA_CASE(6, Z2)
    DRBA_CASE(7, Z3)
    DRBA_CASE(8, Z4)
  }
#undef DRBA_CASE
  return DRBA_EUNSUPPORTED;
}

}  // namespace drba
