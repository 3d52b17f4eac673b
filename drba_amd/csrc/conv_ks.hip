// 3x3 convolution (stride 1, pad 1) of the multi-chunk layers on SMALL maps, split-bf16, K split across the waves.
//
// The ResConv layers of IFBlocks 0-2 (reference models/rife_426_heavy/IFNet_HDv3.py:50-59,69-78: 192 / 128 / 96 channels
// on 17x30 .. 68x120 maps at 1080p, 8 per block, plus block 0 once more in calc_flow, models/rife.py:41-47) are 0.3-2.7 GFLOP
// each: 2-10 us of matrix-core time spread over the chip, but 15-28 us per launch with conv_split_mfma / the fp32 split-K
// kernel -- a workgroup there walks its 3..6 chunks of 32 input channels one after the other (fetch -> split -> LDS ->
// barrier -> MFMA, 3-4 us each), and a map this small has fewer tiles than the chip has SIMDs.
// Here the unit of work is what ONE wave of conv_dma.hip does for one chunk: 2 output rows x 16 pixels x 32 output
// channels x 32 input channels = 216 MFMAs (12 blocks of 8 fp32 dwords per lane, split into three bf16 terms on the way
// into the MFMAs, every split reused for all kernel rows that touch the two output rows), and a workgroup is the KW =
// Cin / 32 waves that share an output unit: wave q does chunk q, all at once.
//   * each wave copies ITS chunk's window ([32 channels][4 rows][24 columns] fp32, NCHW as it lies in HBM) to its
//     private LDS region with `buffer_load ... lds` and waits for it alone (no workgroup barrier before the MFMAs);
//   * weight fragments (split on the host, [cout tile][chunk][tap][nt][plane][64 lanes][16 B]) go L2 -> registers, 6 per
//     MFMA group, two groups ahead: every wave reads a different chunk, so nothing is fetched twice inside a workgroup,
//     and with the window already in LDS the weights are the only loads in the wave's in-order vmcnt queue;
//   * the KW partial accumulators meet in LDS, one barrier, and the waves share the epilogue (bias, beta, residual --
//     for ResConv layers read from the window of chunk == cout tile in LDS, exact fp32 --, activation, 16-byte stores).
// 4 KW x (units) waves of 216 MFMAs: 2176 for the 128-channel layer at 34x60, batch 2 -- one per wave slot of the chip.
// Same arithmetic and error as conv_split.hip (three bf16 terms per fp32 operand, six partial products, fp32 accumulate).
// DW variant: maps whose width is not a multiple of 4 (block 0: 30 columns) move their windows in dwords instead of
// 16-byte units ([4 rows][18 columns]) and store scalars.
#include "common.hpp"
#include "conv_split.hpp"

#include <string.h>

#include <type_traits>
#include <utility>

using namespace drba;

namespace drba_conv_ks {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr;

constexpr int CK = 32, NT = 2, NTC = 32;
// PL = 16-bit terms per operand: 3 = bf16 h + m + l, 2 = the two-term fp16 form (conv_split.hip "Two-term form")
constexpr int frag_u4(int PL) { return 9 * NT * PL * 64; }  // 16-byte units of packed weights per (cout tile, chunk)
[[maybe_unused]] constexpr unsigned kOOB = 0x7FFFFFF0u;

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// window geometry: DW = false: columns x0-4 .. x0+19 in 16-byte units; DW = true: columns x0-1 .. x0+16 in dwords.
// Channel stride == 16 (mod 32) dwords: the 32 lanes a ds_read_b32 services together (kq = 0, 1) hit distinct banks.
template <bool DW>
struct Geo {
  static constexpr int WCOL = DW ? 18 : 24, C0 = DW ? 1 : 4;  // columns per row; column of pixel 0
  static constexpr int CS = DW ? 80 : 112;                    // dwords per channel (4 rows + bank padding)
  static constexpr int A_DW = CK * CS;                        // dwords per chunk window
  static constexpr int UNIT = DW ? 4 : 16;                    // bytes per lane and DMA instruction
  static constexpr int N_INSTR = A_DW * 4 / (64 * UNIT);
  static_assert(CS % 32 == 16 && CS >= 4 * WCOL && (A_DW * 4) % (64 * UNIT) == 0, "window layout");
};

template <int KW, bool PRE, bool RL, bool DW, int PL = 3>
__global__ void __launch_bounds__(KW * 64)
conv_ks(const float *__restrict__ in, const u32x4 *__restrict__ wfrag, const float *__restrict__ bias,
        const float *__restrict__ beta, const float *__restrict__ res, const float *__restrict__ res2, float *__restrict__ out,
        int H, int W, int Cout, int act, float post_slope, float pre_slope, int n_ctiles, int ncb, int nrp, int total,
        unsigned char *status) {
#if defined(__HIP_DEVICE_COMPILE__)
  using G = Geo<DW>;
  constexpr int CS = G::CS, WCOL = G::WCOL;
  constexpr int NM = (PL == 3 ? 6 : 3) * NT, NSPLIT = PL == 3 ? 11 : 6, B0 = PL == 3 ? 4 : 2;  // conv_dma.hip
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // [KW windows][KW x 4 partial accumulators]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = the chunk this wave contracts
  const int m = lane & 15, kq = lane >> 4;
  const int HW = H * W, Cin = KW * CK;
  const unsigned plane_bytes = (unsigned)HW * 4u;

  int t = xcd_band((int)blockIdx.x, total);
  const int cz = t % n_ctiles;
  t /= n_ctiles;
  const int cb = t % ncb;
  t /= ncb;
  const int rp = t % nrp, n = t / nrp;
  const int x0 = cb * 16, y0 = rp * 2;

  float *awin = reinterpret_cast<float *>(lds) + wave * G::A_DW;
  // ---- this wave's window: channels 32 q .., rows y0-1 .. y0+2; outside the image (and the bank padding) reads as zero
  {
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void *)(in + ((size_t)n * Cin + (size_t)wave * CK) * HW), 0, CK * plane_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < G::N_INSTR; ++k) {
      const int u = k * 64 + lane;  // unit of G::UNIT bytes
      constexpr int UPC = CS * 4 / G::UNIT, UPR = WCOL * 4 / G::UNIT;  // units per channel / per row
      const int ch = u / UPC, e = u - ch * UPC;
      const int r = e / UPR, j = e - r * UPR;
      const int gy = y0 - 1 + r, gx = x0 - G::C0 + j * (G::UNIT / 4);
      const bool ok = r < 4 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;  // (16-byte units: W % 4 == 0, whole in or out)
      const unsigned voff = ok ? (unsigned)ch * plane_bytes + (unsigned)(gy * W + gx) * 4u : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(awin + k * 64 * (G::UNIT / 4)), G::UNIT, voff, 0, 0, 0);
    }
  }
  // per-lane epilogue constants (cout nt*16 + m of the tile) and the first weight fragments ride under the window's latency
  const u32x4 *wq = wfrag + ((size_t)cz * KW + wave) * frag_u4(PL) + lane;

  f32x4 acc[2][NT], acl[2][NT];  // acl (PL = 2): the h*l + l*h products, weight 2^-11
  float rawr[2][8];
  u32x4 pl[2][PL];
  constexpr int D = PL == 3 ? 3 : 4;  // weight fragments are requested D - 1 groups ahead of their MFMAs (an L2 round trip ~ 2 bf16 groups)
  u32x4 bw[D][NT][PL];
  float sa[4], sb[4], ta[4], tb[4];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[o][c] = acl[o][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Blocks b = 0..11: window row ir = b / 3, tap column dx = b % 3; block b feeds the output rows o with 0 <= ir - o <= 2
  // (kernel row dy = ir - o): 18 MFMA groups of 12 (conv_dma.hip).
  const float *a_lane = awin + kq * CS + m + G::C0 - 1;
  auto read_raw = [&](int b, int i0, int i1) {
    const float *ap = a_lane + (b / 3) * WCOL + (b % 3);
#pragma unroll
    for (int i = i0; i < i1; ++i) rawr[b & 1][i] = ap[4 * i * CS];
  };
  auto pk = [](float x, float y) -> unsigned {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const bf16x2 p = __builtin_convertvector(f32x2{x, y}, bf16x2);
    return __builtin_bit_cast(unsigned, p);
  };
  auto split_group = [&](int b, int q) {  // group q (0..10) of the split of block b (conv_dma.hip)
    const int s = b & 1;
    auto unpack = [&](const u32x4 &v, int p) { ta[p] = __uint_as_float(v[p] << 16), tb[p] = __uint_as_float(v[p] & 0xffff0000u); };
    if constexpr (PL == 2) {  // two-term fp16 form, 6 groups (conv_dma.hip)
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      auto pkh = [](float x, float y) -> unsigned {
        return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x, y}, f16x2));
      };
      auto unpackh = [&](int p) {
        const unsigned u = pl[s][0][p];  // (through a scalar: bit_cast of a vector ELEMENT reads element 0 with hipcc 7.2)
        const f16x2 hv = __builtin_bit_cast(f16x2, u);
        ta[p] = (float)hv[0], tb[p] = (float)hv[1];
      };
      constexpr float kScale = 1.f / (float)(1 << kSplitActShift);
      if (q == 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          sa[p] = rawr[s][2 * p], sb[p] = rawr[s][2 * p + 1];
          if (PRE) {
            sa[p] = sa[p] > 0.f ? sa[p] : sa[p] * pre_slope;
            sb[p] = sb[p] > 0.f ? sb[p] : sb[p] * pre_slope;
          }
          sa[p] *= kScale, sb[p] *= kScale;
          pl[s][0][p] = pkh(sa[p], sb[p]);
        }
      } else if (q == 1 || q == 2) {
        unpackh(2 * (q - 1)), unpackh(2 * (q - 1) + 1);
      } else if (q == 3) {
#pragma unroll
        for (int p = 0; p < 4; ++p) sa[p] -= ta[p], sb[p] -= tb[p];
      } else if (q == 4) {
#pragma unroll
        for (int p = 0; p < 4; ++p) sa[p] *= 2048.f, sb[p] *= 2048.f;
      } else if (q == 5) {
#pragma unroll
        for (int p = 0; p < 4; ++p) pl[s][1][p] = pkh(sa[p], sb[p]);
      }
    } else if (q == 0) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        sa[p] = rawr[s][2 * p], sb[p] = rawr[s][2 * p + 1];
        if (PRE) {
          sa[p] = sa[p] > 0.f ? sa[p] : sa[p] * pre_slope;
          sb[p] = sb[p] > 0.f ? sb[p] : sb[p] * pre_slope;
        }
        pl[s][0][p] = pk(sa[p], sb[p]);
      }
    } else if (q == 1 || q == 2) {
      unpack(pl[s][0], 2 * (q - 1)), unpack(pl[s][0], 2 * (q - 1) + 1);
    } else if (q == 3 || q == 4) {
#pragma unroll
      for (int p = 2 * (q - 3); p < 2 * (q - 3) + 2; ++p) sa[p] -= ta[p], sb[p] -= tb[p];
    } else if (q == 5) {
#pragma unroll
      for (int p = 0; p < 4; ++p) pl[s][1][p] = pk(sa[p], sb[p]);
    } else if (q == 6 || q == 7) {
      unpack(pl[s][1], 2 * (q - 6)), unpack(pl[s][1], 2 * (q - 6) + 1);
    } else if (q == 8 || q == 9) {
#pragma unroll
      for (int p = 2 * (q - 8); p < 2 * (q - 8) + 2; ++p) sa[p] -= ta[p], sb[p] -= tb[p];
    } else if (q == 10) {
#pragma unroll
      for (int p = 0; p < 4; ++p) pl[s][2][p] = pk(sa[p], sb[p]);
    }
  };
  auto g_block = [](int g) { return g < 3 ? g : (g < 9 ? 3 + (g - 3) / 2 : (g < 15 ? 6 + (g - 9) / 2 : 9 + (g - 15))); };
  auto g_out = [](int g) { return g < 3 ? 0 : (g < 15 ? (g - 3) & 1 : 1); };
  auto load_B1 = [&](int g, int k) {  // k-th (0 .. PL NT - 1) fragment of group g's tap (dy, dx): one 16-byte load from L2
    const int b = g_block(g), dy = b / 3 - g_out(g), dx = b % 3;
    bw[g % D][k / PL][k % PL] = wq[(((dy * 3 + dx) * NT) * PL + k) * 64];
  };
  auto mma1 = [&](int g, int tt) {
    const int s = g_block(g) & 1, o = g_out(g), nt = tt % NT, term = tt / NT;
    if constexpr (PL == 2) {  // al bh, ah bl -> acl; ah bh -> acc
      const f16x8 a = __builtin_bit_cast(f16x8, pl[s][term == 0 ? 1 : 0]), b = __builtin_bit_cast(f16x8, bw[g % D][nt][term == 1 ? 1 : 0]);
      if (term < 2) acl[o][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acl[o][nt], 0, 0, 0);
      else acc[o][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[o][nt], 0, 0, 0);
      return;
    }
    constexpr int ia[6] = {2, 0, 1, 1, 0, 0}, ib[6] = {0, 2, 1, 0, 1, 0};  // al bh, ah bl, am bm, am bh, ah bm, ah bh
    acc[o][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pl[s][ia[term]]),
                                                         __builtin_bit_cast(bf16x8, bw[g % D][nt][ib[term]]), acc[o][nt], 0, 0, 0);
  };

  // the first D - 1 groups' fragments do not depend on the window: requested before waiting for it
#pragma unroll
  for (int g = 0; g < D - 1; ++g)
#pragma unroll
    for (int k = 0; k < PL * NT; ++k) load_B1(g, k);
  // (separate scalars, not arrays: the epilogue picks by a runtime cout tile, and a dynamically indexed array goes to scratch)
  const int co0 = cz * NTC + m, co1 = co0 + 16;
  const float bs0 = (bias && co0 < Cout) ? bias[co0] : 0.f, bs1 = (bias && co1 < Cout) ? bias[co1] : 0.f;
  const float bt0 = (beta && co0 < Cout) ? beta[co0] : 0.f, bt1 = (beta && co1 < Cout) ? beta[co1] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the window (wave-private: no barrier) -- and what was asked for beside it
  read_raw(0, 0, 8);
  read_raw(1, 0, 8);
#pragma unroll
  for (int q = 0; q < NSPLIT; ++q) split_group(0, q);

  static_for<18>([&](auto GG) {
    constexpr int g = decltype(GG)::value;
    constexpr int b = g < 3 ? g : (g < 9 ? 3 + (g - 3) / 2 : (g < 15 ? 6 + (g - 9) / 2 : 9 + (g - 15)));
    constexpr bool two = b >= 3 && b < 9;
    constexpr bool first = !two || ((g - 3) & 1) == 0;
    static_for<NM>([&](auto T) {
      constexpr int tt = decltype(T)::value;
      __builtin_amdgcn_sched_barrier(0);
      mma1(g, tt);
      if constexpr (b + 1 < 12) {
        if constexpr (!two) {
          if constexpr (tt < NSPLIT) split_group(b + 1, tt);
        } else if constexpr ((tt & 1) == 0) {
          constexpr int q = (first ? 0 : NM / 2) + (tt >> 1);
          if constexpr (q < NSPLIT) split_group(b + 1, q);
        }
      }
      if constexpr (first && b + 2 < 12 && tt < 4) read_raw(b + 2, 2 * tt, 2 * tt + 2);
      if constexpr (g + D - 1 < 18 && tt >= B0 && tt - B0 < PL * NT) load_B1(g + D - 1, tt - B0);
    });
  });
  __builtin_amdgcn_sched_barrier(0);

  // ---- the KW partial sums meet in LDS: red[(wave * 4 + o * 2 + nt) * 64 + lane] (16 bytes per lane, behind the windows)
  f32x4 *red = reinterpret_cast<f32x4 *>(lds + KW * G::A_DW * 4);
  float nf = 0.f;
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if constexpr (PL == 2) {
        acc[o][nt] = (acc[o][nt] + acl[o][nt] * (1.f / 2048.f)) * (float)(1 << kSplitActShift);  // exact powers of two
#pragma unroll
        for (int k = 0; k < 4; ++k) nf = nf_fold(nf, acc[o][nt][k]);  // the family's overflow report (common.hpp)
      }
      red[(wave * 4 + o * 2 + nt) * 64 + lane] = acc[o][nt];
    }
  if constexpr (PL == 2) nf_report(status, DRBA_STATUS_CONV_KS, nf);
  __syncthreads();

  // ---- epilogue, shared by the waves: accumulator (o, nt) = c is finished by wave c % KW.  y = sum + bias; ResConv:
  // y = y * beta + x; otherwise y += res (+ res2); activation (0 none, 1 LeakyReLU(0.2), 2 PReLU, 3 ReLU, 4 tanh * 10).
  // Lane (m, kq) holds cout cz*32 + nt*16 + m of pixels x0 + 4 kq .. + 3 of row y0 + o.
  const size_t img = (size_t)n * Cout * HW;
  const unsigned obytes = (unsigned)Cout * plane_bytes;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(out + img), 0, obytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)((res ? res : out) + img), 0, res ? obytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r2rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)((res2 ? res2 : out) + img), 0, res2 ? obytes : 0u, 0x00020000);
  const int xb = x0 + 4 * kq;
  auto finish = [&](auto post) {
#pragma unroll 1
    for (int c = wave; c < 4; c += KW) {
      const int o = c >> 1, nt = c & 1;
      f32x4 v = red[c * 64 + lane];
#pragma unroll
      for (int k = 1; k < KW; ++k) v += red[(k * 4 + c) * 64 + lane];
      const int col = nt * 16 + m, co = cz * NTC + col, y = y0 + o;
      const bool row_ok = y < H && co < Cout;
      const unsigned off = (unsigned)(((co * H + y) * W + xb) * 4);
      f32x4 x1 = (f32x4){0.f, 0.f, 0.f, 0.f}, x2 = x1;
      if (RL) {  // the layer's input IS the residual: channel `col` of chunk cz's window, row o + 1, columns C0 + 4 kq ..
        const float *xp = reinterpret_cast<const float *>(lds) + cz * G::A_DW + col * CS + (o + 1) * WCOL + G::C0 + 4 * kq;
#pragma unroll
        for (int k = 0; k < 4; ++k) x1[k] = xp[k];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned ok = (row_ok && xb + k < W) ? off + 4u * k : 0xffffffffu;
          if (res) x1[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrsrc, ok, 0, 0));
          if (res2) x2[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2rsrc, ok, 0, 0));
        }
      }
      const float bsn = nt ? bs1 : bs0, btn = nt ? bt1 : bt0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float u = v[k] + bsn;
        if (beta) u = u * btn + x1[k];
        else {
          if (RL || res) u = u + x1[k];
          if (!RL && res2) u = u + x2[k];
        }
        v[k] = post(u);
      }
      if (!DW && xb + 3 < W) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, row_ok ? off : 0xffffffffu, 0, 0);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[k]), orsrc,  // (bit_cast of a vector ELEMENT reads element 0 with hipcc 7.2)
                                                (row_ok && xb + k < W) ? off + 4u * k : 0xffffffffu, 0, 0);
      }
    }
  };
  switch (act) {
    case 1: finish([](float v) { return lrelu02(v); }); break;
    case 2: finish([post_slope](float v) { return v > 0.f ? v : post_slope * v; }); break;
    case 3: finish([](float v) { return fmaxf(v, 0.f); }); break;
    case 4: finish([](float v) { return tanhf(v) * 10.f; }); break;
    default: finish([](float v) { return v; }); break;
  }
#endif
}

// ------------------------------------------------------------------------------------------ host side
constexpr int kNum = 1;  // ids 0 .. kNum-1: three bf16 terms; kNum .. 2 kNum-1: two fp16 terms

template <int KW, bool PRE, bool RL, bool DW, int PL>
hipError_t lds_limit(int bytes) {
  (void)bytes;
  return max_dynamic_lds(reinterpret_cast<const void *>(conv_ks<KW, PRE, RL, DW, PL>), 160 * 1024);
}

template <int KW, int PL>
int launch(const float *in, const float *wpk, const float *bias, const float *beta, const float *res, const float *res2,
           float *out, int N, int H, int W, int Cout, int act, float post_slope, int pre_act, float pre_slope, hipStream_t s) {
  const int n_ct = (Cout + NTC - 1) / NTC, ncb = (W + 15) / 16, nrp = (H + 1) / 2;
  const long long total = (long long)n_ct * ncb * nrp * N;
  if (total >= (1ll << 31)) return DRBA_EUNSUPPORTED;
  const bool dw = (W & 3) != 0;
  const int lds_bytes = KW * (dw ? Geo<true>::A_DW : Geo<false>::A_DW) * 4 + KW * 4 * 1024;
  if (lds_bytes > 160 * 1024) return DRBA_EUNSUPPORTED;
  const u32x4 *wf = reinterpret_cast<const u32x4 *>(wpk);
  auto go = [&](auto kernel, hipError_t lds_ok) -> int {
    if (lds_ok != hipSuccess) return DRBA_ELAUNCH;
    DRBA_LAUNCH(kernel, dim3((unsigned)total), dim3(KW * 64), lds_bytes, s, in, wf, bias, beta, res, res2, out, H, W, Cout, act,
                post_slope, pre_slope, n_ct, ncb, nrp, (int)total, PL == 2 ? status_bytes() : nullptr);
    return DRBA_OK;
  };
  const bool rl = res && res == in && !res2 && !pre_act && Cout == KW * CK;
  int rc;
#define DRBA_GO(PRE_, RL_, DW_) go(conv_ks<KW, PRE_, RL_, DW_, PL>, lds_limit<KW, PRE_, RL_, DW_, PL>(lds_bytes))
  if (dw) rc = rl ? DRBA_GO(false, true, true) : (pre_act ? DRBA_GO(true, false, true) : DRBA_GO(false, false, true));
  else rc = rl ? DRBA_GO(false, true, false) : (pre_act ? DRBA_GO(true, false, false) : DRBA_GO(false, false, false));
#undef DRBA_GO
  if (rc != DRBA_OK) return rc;
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // namespace drba_conv_ks

namespace drba {

int conv_ks_num_cfgs() { return drba_conv_ks::kNum; }
int conv_ks_f16_first() { return drba_conv_ks::kNum; }
static int ks_planes(int id) { return id < drba_conv_ks::kNum ? 3 : 2; }

bool conv_ks_supports(int Cin, int Cout, int id) {
  const int kw = Cin / drba_conv_ks::CK;
  return id >= 0 && id < 2 * drba_conv_ks::kNum && Cout > 0 && Cin % drba_conv_ks::CK == 0 && (kw == 2 || kw == 3 || kw == 4 || kw == 6);
}

size_t conv_ks_packed_floats(int Cin, int Cout, int id) {
  if (!conv_ks_supports(Cin, Cout, id)) return 0;
  const size_t n_ct = (Cout + drba_conv_ks::NTC - 1) / drba_conv_ks::NTC, nch = Cin / drba_conv_ks::CK;
  return n_ct * nch * drba_conv_ks::frag_u4(ks_planes(id)) * 4;
}

// packed (16-byte units): [cout tile][chunk][dy][dx][nt][plane h/m/l or h/l][lane] = 8 x 16 bit (split_weight_terms), element i =
//   w[cz*32 + nt*16 + (lane & 15)][q*32 + 4*i + (lane >> 4)][3*dy + dx], zero outside Cout
int conv_ks_pack(const float *w, float *packed, int Cin, int Cout, int id) {
  using namespace drba_conv_ks;
  if (!w || !packed || !conv_ks_supports(Cin, Cout, id)) return DRBA_EINVAL;
  const int n_ct = (Cout + NTC - 1) / NTC, nch = Cin / CK, PL = ks_planes(id);
  if (PL == 2 && !two_term_weights_ok(w, (size_t)Cout * Cin * 9)) return DRBA_EUNSUPPORTED;
  memset(packed, 0, sizeof(float) * conv_ks_packed_floats(Cin, Cout, id));
  unsigned short *dst = reinterpret_cast<unsigned short *>(packed);
  for (int cz = 0; cz < n_ct; ++cz)
    for (int q = 0; q < nch; ++q)
      for (int tap = 0; tap < 9; ++tap)
        for (int nt = 0; nt < NT; ++nt)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = cz * NTC + nt * 16 + (lane & 15);
            if (co >= Cout) continue;
            for (int i = 0; i < 8; ++i) {
              const int ci = q * CK + 4 * i + (lane >> 4);
              unsigned short term[3];
              split_weight_terms(w[((size_t)co * Cin + ci) * 9 + tap], PL, term);
              for (int pl = 0; pl < PL; ++pl) {
                const size_t unit = ((((size_t)cz * nch + q) * 9 + tap) * NT + nt) * PL + pl;
                dst[(unit * 64 + lane) * 8 + i] = term[pl];
              }
            }
          }
  return DRBA_OK;
}

int conv_ks_launch(int id, const float *in, const float *packed_w, const float *bias, const float *beta, const float *residual,
                   const float *residual2, float *out, int N, int Cin, int H, int W, int Cout, int act, float post_slope,
                   int pre_act, float pre_slope, void *stream) {
  using namespace drba_conv_ks;
  if (!conv_ks_supports(Cin, Cout, id)) return DRBA_EUNSUPPORTED;
  if ((size_t)Cin * H * W * 4 >= (1ull << 31) - 64) return DRBA_EUNSUPPORTED;  // 32-bit byte offsets inside an image, below kOOB
  if ((size_t)Cout * H * W * 4 >= (1ull << 31) - 64) return DRBA_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
#define DRBA_CASE(K)                                                                                                          \
  case K:                                                                                                                     \
    return ks_planes(id) == 3                                                                                                 \
               ? launch<K, 3>(in, packed_w, bias, beta, residual, residual2, out, N, H, W, Cout, act, post_slope, pre_act, pre_slope, s) \
               : launch<K, 2>(in, packed_w, bias, beta, residual, residual2, out, N, H, W, Cout, act, post_slope, pre_act, pre_slope, s);
  switch (Cin / CK) {
    DRBA_CASE(2)
    DRBA_CASE(3)
    DRBA_CASE(4)
    DRBA_CASE(6)
  }
#undef DRBA_CASE
  return DRBA_EUNSUPPORTED;
}

}  // namespace drba
