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

using namespace drba;

namespace drba_conv_split {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 32;  // channels per chunk = the K of one bf16 MFMA

template <int RW_, int MW_, int NT_>
struct SplitCfg {
  static constexpr int RW = RW_, MW = MW_, NT = NT_;
  static constexpr int TH = 4 * RW, TW = 16 * MW, NTC = 16 * NT;
  static constexpr int TR = TH + 2, TC = TW + 2, NPIX = TR * TC;
  // LDS: [plane 3][group 4][NPIXP][8 bf16]; NPIXP*16 bytes == 64 (mod 256) spreads the four channel groups of a wave
  // read over distinct banks
  static constexpr int NPIXP = ((NPIX + 15) / 16) * 16 + 4;
  static constexpr int BUF_U4 = 3 * 4 * NPIXP;           // 16-byte units per staging buffer
  static constexpr int LDS_BYTES = 2 * BUF_U4 * 16;      // two buffers: MFMAs on one while the next item is staged
  static constexpr int ITEMS = NPIX * 4;                 // (pixel, channel group) staging items per chunk
  static constexpr int LIT = (ITEMS + 255) / 256;        // per producer thread
  static constexpr int FRAG_U4 = 9 * NT * 3 * 64;        // 16-byte units of packed weights per (cout tile, chunk)
  static constexpr int BDEPTH = 3;                       // weight fetch distance in steps (>= ~700 cycles of MFMAs)
  static_assert(LDS_BYTES <= 160 * 1024, "two staging buffers must fit the CU's LDS");
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

struct TileCtx {
  int x0, y0, cz, n;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global-memory queue
// (s_waitcnt vmcnt(0)): here that would expose, once per tile, the latency of the epilogue stores just issued and of
// the weight fragments prefetched for the next item.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Workgroup = 8 waves, two per SIMD: waves 0-3 (consumers) issue the MFMAs of work item j from LDS buffer j & 1 while
// waves 4-7 (producers) fetch, split and stage item j+1 into the other buffer; one workgroup barrier per item.  The two
// roles are separate waves because a wave's vector-memory results return in issue order: in a single instruction
// stream the streaming weight-fragment loads queue behind the activation prefetch of the next item, and every MFMA
// step then waits out an HBM round trip (measured: the phases added up instead of overlapping, 322 us for the 32->32
// full-HD layer whose MFMA time is 125 us).  Workgroups are persistent (grid = 256 CUs) and walk their XCD's band of
// tiles; an item is one 32-channel chunk of one tile.
template <class Cfg, bool PRE>
__global__ void __launch_bounds__(512)
conv_split_mfma(const float *__restrict__ in, const u32x4 *__restrict__ wfrag, const float *__restrict__ bias,
                const float *__restrict__ beta, const float *__restrict__ res, const float *__restrict__ res2,
                float *__restrict__ out, int Cin, int H, int W, int Cout, int act, float post_slope, float pre_slope,
                int n_ctiles, int nbx, int nby, int total) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int RW = Cfg::RW, MW = Cfg::MW, NT = Cfg::NT, TH = Cfg::TH, TW = Cfg::TW, TC = Cfg::TC;
  constexpr int NPIX = Cfg::NPIX, NPIXP = Cfg::NPIXP, LIT = Cfg::LIT, BUF = Cfg::BUF_U4;
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];  // two buffers of [plane][group][NPIXP]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const size_t HW = (size_t)H * W;
  const int nchunks = Cin / CK;

  auto decode = [&](int work) -> TileCtx {
    int t = xcd_band(work, total);
    TileCtx c;
    c.cz = t % n_ctiles;
    t /= n_ctiles;
    const int bx = t % nbx;
    t /= nbx;
    c.x0 = bx * TW, c.y0 = (t % nby) * TH, c.n = t / nby;
    return c;
  };
  // items of this workgroup: tiles blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x a multiple of 8: one XCD), each
  // nchunks items long
  const int my_tiles = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int n_items = my_tiles * nchunks;
  if (n_items <= 0) return;

  if (producer) {
    // ---- fetch + split + stage.  item = (pixel p of the haloed window, channel group g); lanes run along p.
    // Buffer descriptors: (scalar base) + (per-lane 32-bit offset); a lane whose pixel is outside the image reads
    // offset >= num_records -> 0, the zero padding, without a branch.
    const int ptid = tid - 256;
    const unsigned img_bytes = (unsigned)((size_t)Cin * HW * 4);
    const int plane = (int)HW * 4;
    for (int j = 0; j <= n_items; ++j) {
      if (j < n_items) {
        const int ti = j / nchunks, q = j - ti * nchunks;
        const TileCtx c = decode((int)blockIdx.x + ti * (int)gridDim.x);
        const __amdgpu_buffer_rsrc_t irsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)c.n * Cin * HW), 0, img_bytes, 0x00020000);
        float pre[LIT][8];
#pragma unroll
        for (int it = 0; it < LIT; ++it) {
          const int item = ptid + 256 * it;
          const int g = item / NPIX, p = item - g * NPIX;  // group-major
          const int r = p / TC, cc = p - r * TC;
          const int y = c.y0 + r - 1, x = c.x0 + cc - 1;
          const bool ok = item < Cfg::ITEMS && y >= 0 && y < H && x >= 0 && x < W;
          const unsigned voff = ok ? (unsigned)(g * 8 * plane + (y * W + x) * 4) : 0xffffffffu;
#pragma unroll
          for (int i = 0; i < 8; ++i)
            pre[it][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(irsrc, voff, (q * CK + i) * plane, 0));
        }
        u32x4 *tile = lds + (j & 1) * BUF;
#pragma unroll
        for (int it = 0; it < LIT; ++it) {
          const int item = ptid + 256 * it;
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
            unsigned hh, hm, hl;
            split2(a, b, hh, hm, hl);
            h[i] = hh, mm[i] = hm, l[i] = hl;
          }
          tile[slot] = h;
          tile[4 * NPIXP + slot] = mm;
          tile[8 * NPIXP + slot] = l;
        }
      }
      lds_barrier();  // item j staged (consumers: item j-1 done)
    }
    return;
  }

  // ---- consumers
  const int m = lane & 15, kq = lane >> 4;
  const int row0 = wave * RW;
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)wfrag, 0, n_ctiles * nchunks * Cfg::FRAG_U4 * 16, 0x00020000);
  const bool vec = (W & 3) == 0;
  // Weight fragments are fetched D steps (a step = one tap of one cout tile: RW*MW*6 MFMAs) ahead of their use: a
  // chunk's fragments (27 KB per cout tile) do not stay in the 32 KB L1, so a fetch is an L2 round trip.  The first
  // D of an item are issued before the barrier that hands the item over (and before the previous tile's stores).
  constexpr int STEPS = 9 * NT, D = Cfg::BDEPTH;
  u32x4 bw[D][3];
  auto wload = [&](int wq, int step, int pl) -> u32x4 {
    return __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16, wq + (step * 3 + pl) * 1024, 0);
  };
  auto wq_of = [&](const TileCtx &c, int q) { return (c.cz * nchunks + q) * (Cfg::FRAG_U4 * 16); };  // scalar byte offset

  TileCtx ctx = decode((int)blockIdx.x);
  f32x4 acc[RW][MW][NT];
  float bs[NT], bt[NT];
  {
    const int wq = wq_of(ctx, 0);
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bw[d][pl] = wload(wq, d, pl);
  }
  lds_barrier();  // item 0 staged
  for (int j = 0; j < n_items; ++j) {
    const int ti = j / nchunks, q = j - ti * nchunks;
    if (q == 0) {
#pragma unroll
      for (int a = 0; a < RW; ++a)
#pragma unroll
        for (int b = 0; b < MW; ++b)
#pragma unroll
          for (int c = 0; c < NT; ++c) acc[a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // per-channel epilogue constants: fetched here so that their latency sits under the tile's MFMAs
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = ctx.cz * Cfg::NTC + nt * 16 + m;
        bs[nt] = (bias && co < Cout) ? bias[co] : 0.f;
        bt[nt] = (beta && co < Cout) ? beta[co] : 0.f;
      }
    }
    const u32x4 *tile = lds + (j & 1) * BUF;
    const int wq = wq_of(ctx, q);

    // activation fragments: one tap ahead
    u32x4 af[2][RW][MW][3];
    auto load_a = [&](int tap, int slot_) {
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int rw = 0; rw < RW; ++rw)
#pragma unroll
        for (int mw = 0; mw < MW; ++mw) {
          const int slot = kq * NPIXP + (row0 + rw + ky) * TC + mw * 16 + m + kx;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) af[slot_][rw][mw][pl] = tile[4 * pl * NPIXP + slot];
        }
    };
    load_a(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) load_a(tap + 1, (tap + 1) & 1);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int step = tap * NT + nt;
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bw[step % D][0]);
        const bf16x8 bm = __builtin_bit_cast(bf16x8, bw[step % D][1]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, bw[step % D][2]);
#pragma unroll
        for (int rw = 0; rw < RW; ++rw)
#pragma unroll
          for (int mw = 0; mw < MW; ++mw) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, af[tap & 1][rw][mw][0]);
            const bf16x8 am = __builtin_bit_cast(bf16x8, af[tap & 1][rw][mw][1]);
            const bf16x8 al = __builtin_bit_cast(bf16x8, af[tap & 1][rw][mw][2]);
            f32x4 c = acc[rw][mw][nt];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);  // smallest terms first
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
            acc[rw][mw][nt] = c;
          }
        if (step + D < STEPS) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) bw[step % D][pl] = wload(wq, step + D, pl);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the fetch distances as written
      }
    }

    // the next item's first weight fragments go out before this tile's stores
    const bool last_chunk = q + 1 == nchunks;
    const TileCtx cur = ctx;
    if (j + 1 < n_items) {
      if (last_chunk) ctx = decode((int)blockIdx.x + (ti + 1) * (int)gridDim.x);
      const int wn = wq_of(ctx, last_chunk ? 0 : q + 1);
      static_assert(STEPS % D == 0, "the ring restarts at slot 0 for every item");
#pragma unroll
      for (int d = 0; d < D; ++d)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bw[d][pl] = wload(wn, d, pl);
    }

    if (last_chunk) {
      // ---- epilogue (conv.hip MODE 0): y = acc + bias; ResConv: y = y*beta + res; otherwise y += res (+ res2); then
      // the post activation: act 0 none, 1 LeakyReLU(0.2), 2 PReLU(post_slope), 3 ReLU, 4 tanh(y)*10.  The activation
      // is selected ONCE around the tile loops: selected per element, the 32 inlined copies of the switch (each with
      // a tanhf expansion to jump over) made the epilogue 10k instructions and as slow as the tile's MFMAs.
      const size_t img = (size_t)cur.n * Cout * HW;
      auto epilogue = [&](auto post) {
#pragma unroll
        for (int rw = 0; rw < RW; ++rw)
#pragma unroll
          for (int mw = 0; mw < MW; ++mw)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              f32x4 v = acc[rw][mw][nt];
              const int co = cur.cz * Cfg::NTC + nt * 16 + m;
              const int y = cur.y0 + row0 + rw;
              const int xb = cur.x0 + mw * 16 + kq * 4;
              if (co >= Cout || y >= H || xb >= W) continue;
              const size_t idx = img + ((size_t)co * H + y) * W + xb;
              if (vec) {
                f32x4 r = (f32x4){0.f, 0.f, 0.f, 0.f}, r2 = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (res) r = *reinterpret_cast<const f32x4 *>(res + idx);
                if (res2) r2 = *reinterpret_cast<const f32x4 *>(res2 + idx);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  float u = v[k] + bs[nt];
                  if (beta) u = u * bt[nt] + r[k];
                  else {
                    if (res) u = u + r[k];
                    if (res2) u = u + r2[k];
                  }
                  v[k] = post(u);
                }
                *reinterpret_cast<f32x4 *>(out + idx) = v;
              } else {
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
    lds_barrier();  // item j+1 staged; buffer j & 1 free for item j+2
  }
#endif
}

// ------------------------------------------------------------------------------------------ host side
//                  RW MW NT
using S0 = SplitCfg<2, 2, 2>;  // 8x32 px x 32 cout
using S1 = SplitCfg<2, 2, 4>;  // 8x32 px x 64 cout
using S2 = SplitCfg<1, 2, 6>;  // 4x32 px x 96 cout
using S3 = SplitCfg<1, 4, 2>;  // 4x64 px x 32 cout
using S4 = SplitCfg<1, 4, 4>;  // 4x64 px x 64 cout
constexpr int kNum = 5;
struct Info {
  int NT, NTC, frag_u4;
};
template <class C>
constexpr Info info() {
  return {C::NT, C::NTC, C::FRAG_U4};
}
const Info kInfo[kNum] = {info<S0>(), info<S1>(), info<S2>(), info<S3>(), info<S4>()};

template <class Cfg, bool PRE>
hipError_t lds_limit() {
  if (Cfg::LDS_BYTES <= 64 * 1024) return hipSuccess;
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_split_mfma<Cfg, PRE>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
  return e;
}

template <class Cfg>
int launch(const float *in, const float *wpk, const float *bias, const float *beta, const float *res, const float *res2,
           float *out, int N, int Cin, int H, int W, int Cout, int act, float post_slope, int pre_act, float pre_slope,
           hipStream_t s) {
  const int n_ct = (Cout + Cfg::NTC - 1) / Cfg::NTC;
  const int nbx = (W + Cfg::TW - 1) / Cfg::TW, nby = (H + Cfg::TH - 1) / Cfg::TH;
  const long long total = (long long)nbx * nby * N * n_ct;
  if (total >= (1ll << 31)) return DRBA_EUNSUPPORTED;
  // persistent grid: one 8-wave workgroup per CU, a multiple of 8 (a workgroup stays on its XCD's band of tiles)
  long long grid = 256;
  if (grid > total) grid = (total + 7) / 8 * 8;
  dim3 g((unsigned)grid);
  const u32x4 *wf = reinterpret_cast<const u32x4 *>(wpk);
  auto go = [&](auto kernel, hipError_t lds_ok) -> int {
    if (lds_ok != hipSuccess) return DRBA_ELAUNCH;
    DRBA_LAUNCH_TIMED(kernel, g, dim3(512), Cfg::LDS_BYTES, s, in, wf, bias, beta, res, res2, out, Cin, H, W, Cout, act,
                      post_slope, pre_slope, n_ct, nbx, nby, (int)total);
    return DRBA_OK;
  };
  const int rc = pre_act ? go(conv_split_mfma<Cfg, true>, lds_limit<Cfg, true>())
                         : go(conv_split_mfma<Cfg, false>, lds_limit<Cfg, false>());
  if (rc != DRBA_OK) return rc;
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

// round-to-nearest-even fp32 -> bf16 (finite inputs), returned as the fp32 value it represents
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

}  // namespace drba_conv_split

namespace drba {

int conv_split_num_cfgs() { return drba_conv_split::kNum; }

bool conv_split_supports(int Cin, int Cout, int id) {
  return id >= 0 && id < drba_conv_split::kNum && Cin > 0 && Cout > 0 && Cin % drba_conv_split::CK == 0;
}

size_t conv_split_packed_floats(int Cin, int Cout, int id) {
  if (!conv_split_supports(Cin, Cout, id)) return 0;
  const drba_conv_split::Info &c = drba_conv_split::kInfo[id];
  const size_t n_ct = (Cout + c.NTC - 1) / c.NTC, nch = Cin / drba_conv_split::CK;
  return n_ct * nch * c.frag_u4 * 4;
}

// packed (16-byte units): [cout tile][chunk][tap][nt][plane h/m/l][lane] = 8 bf16 of
//   w[cz*NTC + nt*16 + (lane & 15)][q*32 + 8*(lane >> 4) + i][tap], i = 0..7, zero outside Cout
int conv_split_pack(const float *w, float *packed, int Cin, int Cout, int id) {
  using namespace drba_conv_split;
  if (!w || !packed || !conv_split_supports(Cin, Cout, id)) return DRBA_EINVAL;
  const Info &c = kInfo[id];
  const int n_ct = (Cout + c.NTC - 1) / c.NTC, nch = Cin / CK;
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
              const float x = w[((size_t)co * Cin + ci) * 9 + tap];
              const float h = bf16_round(x), m = bf16_round(x - h), l = bf16_round(x - h - m);
              const float term[3] = {h, m, l};
              for (int pl = 0; pl < 3; ++pl) {
                const size_t unit = ((((size_t)cz * nch + q) * 9 + tap) * c.NT + nt) * 3 + pl;
                dst[(unit * 64 + lane) * 8 + i] = bf16_bits(term[pl]);
              }
            }
          }
  return DRBA_OK;
}

int conv_split_launch(int id, const float *in, const float *packed_w, const float *bias, const float *beta,
                      const float *residual, const float *residual2, float *out, int N, int Cin, int H, int W, int Cout,
                      int act, float post_slope, int pre_act, float pre_slope, void *stream) {
  using namespace drba_conv_split;
  if (!conv_split_supports(Cin, Cout, id)) return DRBA_EUNSUPPORTED;
  if ((size_t)Cin * H * W * 4 >= (1ull << 31)) return DRBA_EUNSUPPORTED;  // 32-bit byte offsets inside an image
  hipStream_t s = (hipStream_t)stream;
#define DRBA_CASE(ID, T) \
  case ID:               \
    return launch<T>(in, packed_w, bias, beta, residual, residual2, out, N, Cin, H, W, Cout, act, post_slope, pre_act, \
                     pre_slope, s);
  switch (id) {
    DRBA_CASE(0, S0)
    DRBA_CASE(1, S1)
    DRBA_CASE(2, S2)
    DRBA_CASE(3, S3)
    DRBA_CASE(4, S4)
  }
#undef DRBA_CASE
  return DRBA_EUNSUPPORTED;
}

}  // namespace drba
