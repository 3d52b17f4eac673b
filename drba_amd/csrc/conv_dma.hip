// 3x3 convolution (stride 1, pad 1) of 32-channel layers, split-bf16 on the matrix cores, operands moved by LDS-DMA.
//
// Same arithmetic as conv_split.hip (an fp32 value is the sum of three bf16 terms h + m + l, a product is evaluated as
// the six partial products >= 2^-16 of the leading one, fp32 accumulation: fp32-level error), different machine, built
// for the layer shape that dominates a 1080p / 4K RIFE step: the 32 -> 32 channel ResConv of IFBlock 4 at 1/4 resolution
// (reference models/rife_426_heavy/IFNet_HDv3.py:50-59,69-78: lrelu(conv3x3(x) * beta + x), 8 per block) and the other
// stride-1 layers with 32 input and at most 32 output channels (GridNet / FeatureNet heads).
//   * conv_split_mfma fetches the fp32 window into REGISTERS (48 dword loads per lane), splits it in a staging phase,
//     writes three bf16 planes to LDS and reads its weight fragments from L2 inside the MFMA loop.  All of that shares
//     the wave's in-order vmcnt queue: per 8x32 tile the matrix cores work 2.9 us of 17.
//   * here the layer's weight fragments (54 KB, split on the host) are copied to LDS ONCE per workgroup and stay; the
//     fp32 window of a tile ([32 channels][10 rows][40 columns], the NCHW planes as they lie in HBM, 16 bytes per lane)
//     is copied global -> LDS by a NINTH wave that does nothing else -- `buffer_load ... lds`, no registers -- one whole
//     tile ahead, double buffered; the eight MFMA waves have no load in their vmcnt queue at all and meet the loader
//     at ONE workgroup barrier per tile.  The HBM format does not change (fp32 NCHW in and out): the kernel is another
//     configuration id of drba_conv3x3 and the autotuner keeps it where it is faster.
//   * an MFMA wave owns 2 output rows x 16 pixels x 32 output channels.  It reads the 8 fp32 channels of its lane's
//     pixel with conflict-free ds_read_b32 (channel stride 400 dwords == 16 mod 32), splits them ON THE WAY into the
//     MFMAs, and uses every split block for all the kernel rows dy that touch its two output rows: 12 splits per tile
//     and wave instead of 18.  The block loop is written as slots of [one MFMA, four split instructions of the next
//     block, one LDS read] fenced with sched_barrier: the matrix pipe takes 17 cycles per MFMA, a wave 4 to issue one,
//     and the VALU work rides in the gaps (left to hipcc, each block's split sits in front of its own MFMAs).
//   K order inside an MFMA: lane (m = lane % 16, kq = lane / 16) supplies k = 8 kq .. 8 kq + 7, mapped to channels
//   4 i + kq (i = 0..7); the weights are packed with the same mapping.
// LDS (bytes): A[2] 2 x 51200 | W 55296 = [dy][dx][nt][plane h/m/l][64 lanes][16 B] | 2 tile numbers.
// ResConv layers (residual == input): the epilogue reads the residual from the window in LDS (exact fp32).
#include "common.hpp"
#include "conv_split.hpp"

#include <string.h>

#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

using namespace drba;

namespace drba_conv_dma {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr;

constexpr int CK = 32;                   // input channels = K of one bf16 MFMA
constexpr int NT = 2, NTC = 16 * NT;     // output channels: two MFMA column blocks
constexpr int TH = 8, TW = 32;           // pixels per work item
constexpr int WR = TH + 2, WC = TW + 8;  // window: rows y0-1 .. y0+8, columns x0-4 .. x0+35 (16-byte units)
constexpr int CS = WR * WC;              // 400 dwords per channel, == 16 (mod 32)
constexpr int A_BYTES = CK * CS * 4;     // 51200
[[maybe_unused]] constexpr int A_INSTR = A_BYTES / 1024;  // 50 wave-level DMA instructions of 64 lanes x 16 B
// PL = number of 16-bit terms per operand: 3 = bf16 h + m + l, 2 = the two-term fp16 form (conv_split.hip "Two-term form")
constexpr int w_bytes(int PL) { return 9 * NT * PL * 1024; }  // 55296 / 36864
constexpr int OFF_W = 2 * A_BYTES;
constexpr int off_idx(int PL) { return OFF_W + w_bytes(PL); }  // 2 x {x0, y0, image, tile number}: what the loader publishes per item
constexpr int lds_bytes(int PL) { return off_idx(PL) + 32; }
constexpr int NTHREADS = 9 * 64;         // 8 MFMA waves + the loader
static_assert(CS % 32 == 16 && A_BYTES % 1024 == 0 && lds_bytes(3) <= 160 * 1024, "window layout");
[[maybe_unused]] constexpr unsigned kOOB = 0x7FFFFFF0u;  // beyond any num_records: the load returns 0 (zero padding), never faults

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N - 1>) -- the block schedule below is a table indexed by
// the group number, which must be a constant in every copy of the body (hipcc does not fully unroll an 18 x 12 body on
// `#pragma unroll` alone)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct Item {  // scalar (wave-uniform) description of a work item
  int x0, y0, n;
};

// Experiment builds only (-DDRBA_EXP_CLOCKS): clocks of workgroup 0 / 131, waves 0 and 5, printed at the end --
// [0] barrier wait, [1] head (first reads + split), [2] MFMA blocks, [3] epilogue, [4] items
#ifdef DRBA_EXP_CLOCKS
#define DRBA_CLK(var) const long long var = (long long)__builtin_readcyclecounter()
#define DRBA_CLK_ADD(slot, a, b) clk_acc[slot] += (b) - (a)
#else
#define DRBA_CLK(var)
#define DRBA_CLK_ADD(slot, a, b)
#endif

template <bool PRE, bool RL, int PL = 3>
__global__ void __launch_bounds__(NTHREADS, 1)
conv_dma1(const float *__restrict__ in, const u32x4 *__restrict__ wfrag, const float *__restrict__ bias,
          const float *__restrict__ beta, const float *__restrict__ res, const float *__restrict__ res2,
          float *__restrict__ out, int H, int W, int Cout, int act, float post_slope, float pre_slope, int nbx, int nby,
          int total, int *__restrict__ counters, int *__restrict__ counters_next, unsigned char *status) {
#if defined(__HIP_DEVICE_COMPILE__)
  float nf = 0.f;  // PL = 2: NaN once a sum of this lane came out non-finite (nf_fold / nf_report, common.hpp)
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // the ONLY LDS object of the kernel
  constexpr int W_BYTES = w_bytes(PL), W_INSTR = W_BYTES / 1024, OFF_IDX = off_idx(PL);
  constexpr int NM = (PL == 3 ? 6 : 3) * NT;  // MFMAs of a group = slots of the block program
  constexpr int NSPLIT = PL == 3 ? 11 : 6;    // instruction groups of one block's split

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HW = H * W;
  const unsigned plane_bytes = (unsigned)HW * 4u;

  // Work distribution.  Tiles are numbered in (image, tile row, tile column) order and cut into 8 contiguous bands, one
  // per XCD (workgroup b runs on XCD b % 8 -- observed dispatch order, used for speed only -- so that the halo rows
  // neighbouring tiles share are fetched into one private L2).  A workgroup's FIRST tile is static (its rank inside its
  // band); every further one is taken from its band's counter, then from the other bands' (atomicAdd by the loader, one
  // tile ahead).  A static partition (tile = b + i * grid) assumes that all 256 workgroups start together: beside two
  // other streams' kernels they do not -- a workgroup whose CU is still busy starts late and, with a fixed share, ends
  // late, and the launch lasts twice its stand-alone time; taking tiles on demand, late workgroups simply do fewer.
  // counters[0..7]: tiles handed out per band beyond the static ones.  Launches on a stream alternate between two sets
  // of counters and each zeroes the set of the next one (counters_next), so nobody has to find out who finished last.
  const int band_q = total >> 3, band_r = total & 7;
  auto band_start = [&](int x) { return x < band_r ? x * (band_q + 1) : band_r * (band_q + 1) + (x - band_r) * band_q; };
  auto band_len = [&](int x) { return band_q + (x < band_r ? 1 : 0); };
  const int my_xcd = (int)(blockIdx.x & 7), n_static = (int)(gridDim.x >> 3);  // grid is a multiple of 8
  auto decode = [&](int t) -> Item {
    Item c;
    const int bx = t % nbx;
    t /= nbx;
    c.x0 = bx * TW, c.y0 = (t % nby) * TH, c.n = t / nby;
    return c;
  };
  int *lds_idx = reinterpret_cast<int *>(lds + OFF_IDX);  // [2][4]: the item behind barrier i in slot i & 1 (tile number -1: none)

  if (wave == 8) {
    // ================================================================ the loader
    __builtin_amdgcn_s_setprio(3);  // its few instructions go first: an MFMA wave waiting at the barrier costs a whole tile
    // window of item c -> A buffer ab: LDS unit u = 64 k + lane is (channel u / 100, row (u % 100) / 10, 16-byte column
    // group u % 10); rows / column groups outside the image read as zero through the buffer range check (the padding).
    // A lane's source offset relative to the window's origin is the same for every item: computed once (50 registers --
    // formed per item, the divisions made the loader the slowest wave of the workgroup: 20k clocks per tile)
    unsigned a_rel[A_INSTR];  // byte offset (a multiple of 16) | window row in the low 4 bits
    const int lane10 = lane % (WC / 4);
#pragma unroll
    for (int k = 0; k < A_INSTR; ++k) {
      const int u = k * 64 + lane;
      const int ch = u / (WR * (WC / 4)), e = u - ch * (WR * (WC / 4));
      const int r = e / (WC / 4), j = e - r * (WC / 4);
      a_rel[k] = ((unsigned)ch * plane_bytes + (unsigned)(r * W + 4 * j) * 4u) | (unsigned)r;
    }
    auto issue_A = [&](const Item &c, int ab) {
      const int ry = c.y0 - 1, cx = c.x0 - 4;
      const float *img = in + (size_t)c.n * CK * HW;
      if (ry >= 0 && ry + WR <= H && cx >= 0 && cx + WC <= W) {
        // interior tile (4 of 5 at 1080p): every unit is inside the image -- the window's origin goes into the buffer
        // base, the lane offsets are the precomputed ones: two instructions per DMA, no VALU work at all (the loader
        // shares its SIMD with two MFMA waves whose split keeps the VALU busy)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(img + ry * W + cx), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int k = 0; k < A_INSTR; ++k)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + ab * A_BYTES + k * 1024), 16, a_rel[k] & ~15u, 0, 0, 0);
      } else {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, CK * plane_bytes, 0x00020000);
        const unsigned tb = (unsigned)((ry * W + cx) * 4);  // wraps for the first row / column: valid lanes add up >= 0
#pragma unroll
        for (int k = 0; k < A_INSTR; ++k) {
          int j = lane10 + (k * 64) % (WC / 4);  // column group of unit 64 k + lane
          j = j >= WC / 4 ? j - WC / 4 : j;
          const bool ok = (unsigned)(ry + (int)(a_rel[k] & 15u)) < (unsigned)H && (unsigned)(cx + 4 * j) < (unsigned)W;
          const unsigned voff = ok ? (a_rel[k] & ~15u) + tb : kOOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + ab * A_BYTES + k * 1024), 16, voff, 0, 0, 0);
        }
      }
    };
    // Next tile for this workgroup, -1 when there is none: one atomic on the own band's counter while it lasts, then ONE
    // load of all eight counters and one attempt on the band with the most tiles left (its own workgroups keep draining
    // it whether or not that succeeds; a chain of failing atomics, each a round trip to memory, cost 10 us at the end of
    // every launch).  The own band's atomic is ISSUED before the window's 50 DMA instructions and its result read after
    // them: returns come back in issue order, so read after them it would wait for the whole window.
    auto band_tile = [&](int b, int k) -> int {
      const int len = band_len(b), ns = min(n_static, len);
      return ns + k < len ? band_start(b) + ns + k : -1;
    };
    auto ask = [&](int b) -> int {  // lane 0's return value is the ticket
      int k = 0;
      if (lane == 0) k = __hip_atomic_fetch_add(counters + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return k;
    };
    bool own_left = true;
#ifdef DRBA_EXP_STATIC  // experiment: the static partition (tile = b + i * grid in band order), no counters
    int static_i = 0;
#endif
    auto finish_grab = [&](int ticket) -> int {
#ifdef DRBA_EXP_STATIC
      const int w = (int)blockIdx.x + (++static_i) * (int)gridDim.x;
      return w < total ? xcd_band(w, total) : -1;
#endif
      if (own_left) {
        const int t = band_tile(my_xcd, __builtin_amdgcn_readfirstlane(ticket));
        if (t >= 0) return t;
        own_left = false;
      }
      int left = 0;
      if (lane < 8) {
        const int len = band_len(lane), ns = min(n_static, len);
        left = len - ns - __hip_atomic_load(counters + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      int best = -1, most = 0;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int l = __builtin_amdgcn_readlane(left, b);
        if (l > most) most = l, best = b;
      }
      return best < 0 ? -1 : band_tile(best, __builtin_amdgcn_readfirstlane(ask(best)));
    };
    auto publish = [&](int tile, int slot) {  // the decoded tile (the MFMA waves do not repeat the divisions behind the barrier)
      const Item c = decode(tile < 0 ? 0 : tile);
      if (lane == 0) *reinterpret_cast<int4 *>(lds_idx + 4 * slot) = make_int4(c.x0, c.y0, c.n, tile);
    };
    if (blockIdx.x == 0 && lane < 8) counters_next[lane] = 0;  // the set the NEXT launch on this stream counts in
    int cur = (int)(blockIdx.x >> 3) < band_len(my_xcd) ? band_start(my_xcd) + (int)(blockIdx.x >> 3) : -1;
    int ab = 0;
    publish(cur, 0);
    int nxt = -1;  // taken one tile ahead: the counter's round trip hides under the window's
    if (cur >= 0) {  // (the first window is fetched by the MFMA waves, see there)
#ifdef DRBA_EXP_STATIC
      const int ticket = 0;
#else
      const int ticket = own_left ? ask(my_xcd) : 0;
#endif
      nxt = finish_grab(ticket);
    }
#ifdef DRBA_EXP_CLOCKS
    long long lc[4] = {0, 0, 0, 0};
#endif
    while (true) {
      // barrier i: the window of item i (and, the first time, the weights) has landed and its tile number is published;
      // every MFMA wave is done with item i - 1, whose buffer the window of item i + 1 may now overwrite
      DRBA_CLK(l0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      DRBA_CLK(l1);
      __builtin_amdgcn_s_barrier();
      DRBA_CLK(l2);
      if (cur < 0) break;
      cur = nxt;
      ab ^= 1;
      publish(cur, ab);
      if (cur >= 0) {
#ifdef DRBA_EXP_STATIC
        const int ticket = 0;
#else
        const int ticket = own_left ? ask(my_xcd) : 0;
#endif
        issue_A(decode(cur), ab);
        nxt = finish_grab(ticket);
      }
      DRBA_CLK(l3);
#ifdef DRBA_EXP_CLOCKS
      lc[0] += l1 - l0, lc[1] += l2 - l1, lc[2] += l3 - l2, lc[3] += 1;
#endif
    }
#ifdef DRBA_EXP_CLOCKS
    if (blockIdx.x == 131 && lane == 0 && lc[3])
      printf("wg %d loader: rounds %lld  vmcnt wait %lld  barrier wait %lld  issue %lld (clocks per round)\n", (int)blockIdx.x, lc[3],
             lc[0] / lc[3], lc[1] / lc[3], lc[2] / lc[3]);
#endif
    return;
  }

  // ================================================================== the MFMA waves
  // the layer's weight fragments -> LDS, once: every MFMA wave copies its share while the loader fetches the first window
  {
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wfrag, 0, W_BYTES, 0x00020000);
#pragma unroll
    for (int i = 0; i < (W_INSTR + 7) / 8; ++i) {
      const int k = i * 8 + wave;
      if (k < W_INSTR)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr)(lds + OFF_W + k * 1024), 16, (unsigned)(k * 1024 + lane * 16), 0, 0, 0);
    }
  }
  // ... and the workgroup's FIRST window (its tile is static), 6-7 of the 50 instructions per wave: the MFMA waves have
  // nothing else to do yet, and the loader alone needs 3k clocks to set itself up and 3k more to issue a window
  {
    const int first = (int)(blockIdx.x >> 3) < band_len(my_xcd) ? band_start(my_xcd) + (int)(blockIdx.x >> 3) : -1;
    if (first >= 0) {
      const Item c = decode(first);
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)c.n * CK * HW), 0, CK * plane_bytes, 0x00020000);
      const int ry = c.y0 - 1, cx = c.x0 - 4;
#pragma unroll
      for (int i = 0; i < (A_INSTR + 7) / 8; ++i) {
        const int k = i * 8 + wave;
        if (k < A_INSTR) {
          const int u = k * 64 + lane;
          const int ch = u / (WR * (WC / 4)), e = u - ch * (WR * (WC / 4));
          const int r = e / (WC / 4), j = e - r * (WC / 4);
          const bool ok = (unsigned)(ry + r) < (unsigned)H && (unsigned)(cx + 4 * j) < (unsigned)W;
          const unsigned voff = ok ? (unsigned)ch * plane_bytes + (unsigned)((ry + r) * W + cx + 4 * j) * 4u : kOOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + k * 1024), 16, voff, 0, 0, 0);
        }
      }
    }
  }
  // The two MFMA waves of a SIMD (w, w + 4) compete for its matrix and VALU pipes; arbitration is by priority, then age.
  // Left at equal priority the younger wave gets the leftovers and finishes its blocks 40 % later (10.0k against 7.0k
  // clocks per tile, and the tile takes as long as the slower one; a constant higher priority for the younger half just
  // swaps the roles).  So each half is favoured for half of the tile: waves 0-3 by age during groups 0-8, waves 4-7 by
  // priority during groups 9-17.
  const int m = lane & 15, kq = lane >> 4;
  const int mw = wave & 1, rp = wave >> 1;  // column block, row pair of the tile
  const float *ldsf = reinterpret_cast<const float *>(lds);
  const u32x4 *ldsq = reinterpret_cast<const u32x4 *>(lds);
  // dword of (channel kq, window row 2 rp, column of pixel m at tap dx = 0)
  const int a_lane = kq * CS + (2 * rp) * WC + 16 * mw + m + 3;
  const int w_lane = OFF_W / 16 + lane;
#ifdef DRBA_EXP_CLOCKS
  long long clk_acc[5] = {0, 0, 0, 0, 0}, first_barrier = 0;
  const long long k_start = (long long)__builtin_readcyclecounter();
#endif

  // per-lane epilogue constants: cout nt*16 + m
  float bs[NT], bt[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = nt * 16 + m;
    bs[nt] = (bias && co < Cout) ? bias[co] : 0.f;
    bt[nt] = (beta && co < Cout) ? beta[co] : 0.f;
  }

  f32x4 acc[2][NT];    // [output row of the pair][cout tile]
  f32x4 acl[2][NT];    // PL = 2: the h*l + l*h products (weight 2^-11), joined in the epilogue
  float rawr[2][8];    // ring: raw A dwords of block b in rawr[b & 1]
  u32x4 pl[2][PL];     // h / m / l (PL = 2: h / l) operands of block b in pl[b & 1]
  u32x4 bw[2][NT][PL]; // weight fragments of MFMA group g in bw[g & 1]
  float sa[4], sb[4], ta[4], tb[4];  // split in flight: remainders (sa, sb) and unpacked terms (ta, tb) of the 4 pairs
  // Blocks of a tile, b = 0..11: window row ir = b / 3 of the wave's four, tap column dx = b % 3.  Block b feeds the output
  // rows o with 0 <= ir - o <= 2 (kernel row dy = ir - o): one MFMA group (12 MFMAs) for ir = 0, 3, two for ir = 1, 2.
  // Groups in issue order, g = 0..17: (b, o).
  auto read_raw = [&](int ab, int b, int i0, int i1) {
    const float *ap = ldsf + ab * (A_BYTES / 4) + a_lane + (b / 3) * WC + (b % 3);
#pragma unroll
    for (int i = i0; i < i1; ++i) rawr[b & 1][i] = ap[4 * i * CS];
  };
  auto pk = [](float x, float y) -> unsigned {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const bf16x2 p = __builtin_convertvector(f32x2{x, y}, bf16x2);
    return __builtin_bit_cast(unsigned, p);
  };
  // group q (0..10) of the split of block b: fp32 -> h + m + l, round-to-nearest-even at every step (conv_split.hip),
  // cut stage by stage into groups of 4 independent instructions
  auto split_group = [&](int b, int q) {
    const int s = b & 1;
    auto unpack = [&](const u32x4 &v, int p) { ta[p] = __uint_as_float(v[p] << 16), tb[p] = __uint_as_float(v[p] & 0xffff0000u); };
#ifdef DRBA_EXP_NOSPLIT  // experiment (wrong results): the operands are the raw bits, no split arithmetic
    if (q == 0)
      for (int p = 0; p < 4; ++p) {
        pl[s][0][p] = __float_as_uint(rawr[s][2 * p]), pl[s][1][p] = __float_as_uint(rawr[s][2 * p + 1]);
        pl[s][2][p] = pl[s][0][p] ^ pl[s][1][p];
      }
    return;
#endif
    if constexpr (PL == 2) {
      // two-term fp16 form: x' = x * 2^-shift, h = fp16(x'), l = fp16((x' - h) * 2^11), in 6 groups
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
      return;
    }
    if (q == 0) {
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
  // group table (compile-time after unrolling): block, output row, kernel row of group g
  auto g_block = [](int g) { return g < 3 ? g : (g < 9 ? 3 + (g - 3) / 2 : (g < 15 ? 6 + (g - 9) / 2 : 9 + (g - 15))); };
  auto g_out = [](int g) { return g < 3 ? 0 : (g < 15 ? (g - 3) & 1 : 1); };
  auto read_B1 = [&](int g, int k) {  // k-th (0 .. PL NT - 1) 16-byte fragment of group g's tap (dy, dx)
    const int b = g_block(g), dy = b / 3 - g_out(g), dx = b % 3;
    bw[g & 1][k / PL][k % PL] = ldsq[w_lane + (((dy * 3 + dx) * NT) * PL + k) * 64];
  };
  // MFMA t (0 .. 6 NT - 1) of group g: term t / NT of cout tile t % NT (smallest terms first; the accumulators alternate)
  auto mma1 = [&](int g, int t) {
    const int s = g_block(g) & 1, o = g_out(g), nt = t % NT, term = t / NT;
    if constexpr (PL == 2) {  // al bh, ah bl -> acl; ah bh -> acc
      const f16x8 a = __builtin_bit_cast(f16x8, pl[s][term == 0 ? 1 : 0]), b = __builtin_bit_cast(f16x8, bw[g & 1][nt][term == 1 ? 1 : 0]);
      if (term < 2) acl[o][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acl[o][nt], 0, 0, 0);
      else acc[o][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[o][nt], 0, 0, 0);
      return;
    }
    constexpr int ia[6] = {2, 0, 1, 1, 0, 0}, ib[6] = {0, 2, 1, 0, 1, 0};  // al bh, ah bl, am bm, am bh, ah bm, ah bh
#ifdef DRBA_EXP_NOMFMA  // experiment (wrong results): operands consumed by one VALU instruction instead of the MFMA
    acc[o][nt][term & 3] += __uint_as_float(pl[s][ia[term]][term & 3] ^ bw[g & 1][nt][ib[term]][term & 3]);
    return;
#endif
    acc[o][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pl[s][ia[term]]),
                                                         __builtin_bit_cast(bf16x8, bw[g & 1][nt][ib[term]]), acc[o][nt], 0, 0, 0);
  };

  int ab = 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the weights (its only loads)
  while (true) {
    DRBA_CLK(t0);
    __builtin_amdgcn_s_barrier();  // barrier i (see the loader)
    const int4 pub = *reinterpret_cast<const int4 *>(lds_idx + 4 * ab);
    if (__builtin_amdgcn_readfirstlane(pub.w) < 0) break;
    Item cur;
    cur.x0 = __builtin_amdgcn_readfirstlane(pub.x), cur.y0 = __builtin_amdgcn_readfirstlane(pub.y);
    cur.n = __builtin_amdgcn_readfirstlane(pub.z);
    DRBA_CLK(t1);
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int c = 0; c < NT; ++c) acc[o][c] = acl[o][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // head: the first two blocks' reads, the first split, the first fragments
    read_raw(ab, 0, 0, 8);
    read_raw(ab, 1, 0, 8);
#pragma unroll
    for (int k = 0; k < PL * NT; ++k) read_B1(0, k);
#pragma unroll
    for (int q = 0; q < NSPLIT; ++q) split_group(0, q);
    DRBA_CLK(t2);
    // 18 groups x NM (12 / 6) slots: [MFMA, 4 split instructions of the next block, one LDS read]
    static_for<18>([&](auto G) {
      constexpr int g = decltype(G)::value;
      constexpr int b = g < 3 ? g : (g < 9 ? 3 + (g - 3) / 2 : (g < 15 ? 6 + (g - 9) / 2 : 9 + (g - 15)));
      constexpr bool two = b >= 3 && b < 9;                       // the block has two groups
      constexpr bool first = !two || ((g - 3) & 1) == 0;          // g is the first group of its block
#ifndef DRBA_EXP_NOPRIO
      if constexpr (g == 9) {
        if (wave >= 4) __builtin_amdgcn_s_setprio(2);
      }
      if constexpr (g == 0) {
        if (wave >= 4) __builtin_amdgcn_s_setprio(0);
      }
#endif
      static_for<NM>([&](auto T) {
        constexpr int t = decltype(T)::value;
        __builtin_amdgcn_sched_barrier(0);
        mma1(g, t);
        // split of block b + 1: its NSPLIT groups over the slots of a one-group block, over the even slots of a two-group one
        if constexpr (b + 1 < 12) {
          if constexpr (!two) {
            if constexpr (t < NSPLIT) split_group(b + 1, t);
          } else if constexpr ((t & 1) == 0) {
            constexpr int q = (first ? 0 : NM / 2) + (t >> 1);
            if constexpr (q < NSPLIT) split_group(b + 1, q);
          }
        }
        // LDS reads: the raw dwords of block b + 2 (two per slot, slots 0..3 of the block's first group) ...
        if constexpr (first && b + 2 < 12 && t < 4) read_raw(ab, b + 2, 2 * t, 2 * t + 2);
        // ... and the next group's weight fragments (slots B0 .. B0 + PL NT - 1)
        constexpr int B0 = PL == 3 ? 4 : 2;
        if constexpr (g + 1 < 18 && t >= B0 && t - B0 < PL * NT) read_B1(g + 1, t - B0);
      });
    });
    __builtin_amdgcn_sched_barrier(0);
    DRBA_CLK(t3);

    // ---- epilogue: y = acc + bias; ResConv: y = y * beta + x; otherwise y += res (+ res2); activation (0 none,
    // 1 LeakyReLU(0.2), 2 PReLU(post_slope), 3 ReLU, 4 tanh * 10).  Lane (m, kq) holds cout nt*16 + m of pixels
    // x0 + 16 mw + 4 kq .. + 3 of rows y0 + 2 rp + o: one 16-byte store per (o, nt); lanes outside the image or past Cout
    // get an offset beyond num_records (dropped).
    {
      const size_t img = (size_t)cur.n * Cout * HW;
      const unsigned obytes = (unsigned)Cout * plane_bytes;
      const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(out + img), 0, obytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rrsrc =
          __builtin_amdgcn_make_buffer_rsrc((void *)((res ? res : out) + img), 0, res ? obytes : 0u, 0x00020000);
      const __amdgpu_buffer_rsrc_t r2rsrc =
          __builtin_amdgcn_make_buffer_rsrc((void *)((res2 ? res2 : out) + img), 0, res2 ? obytes : 0u, 0x00020000);
      const int xb = cur.x0 + 16 * mw + 4 * kq;
      auto epilogue = [&](auto post) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int co = nt * 16 + m, y = cur.y0 + 2 * rp + o;
            const unsigned off = (y < H && xb < W && co < Cout) ? (unsigned)(((co * H + y) * W + xb) * 4) : 0xffffffffu;
            f32x4 v = acc[o][nt], x1 = (f32x4){0.f, 0.f, 0.f, 0.f}, x2 = x1;
            if constexpr (PL == 2) {
              v = (v + acl[o][nt] * (1.f / 2048.f)) * (float)(1 << kSplitActShift);  // exact powers of two
#pragma unroll
              for (int k = 0; k < 4; ++k) nf = nf_fold(nf, v[k]);  // the family's overflow report
            }
            if (RL) {  // the layer's input IS the residual: channel co of the window, row 2 rp + o + 1, columns 4 + 16 mw + 4 kq ..
              x1 = *reinterpret_cast<const f32x4 *>(ldsf + ab * (A_BYTES / 4) + co * CS + (2 * rp + o + 1) * WC + 4 + 16 * mw + 4 * kq);
            } else if (res) {
              x1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off, 0, 0));
            }
            if (!RL && res2) x2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2rsrc, off, 0, 0));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float u = v[k] + bs[nt];
              if (beta) u = u * bt[nt] + x1[k];
              else {
                if (RL || res) u = u + x1[k];
                if (!RL && res2) u = u + x2[k];
              }
              v[k] = post(u);
            }
#ifdef DRBA_EXP_NOSTORE  // experiment: the stores are compiled in but never executed (H is never negative)
            if (H >= 0) continue;
#endif
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, off, 0, 0);
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
    DRBA_CLK(t4);
#ifdef DRBA_EXP_CLOCKS
    if (clk_acc[4] == 0) first_barrier = t1 - t0;
    else
#endif
    DRBA_CLK_ADD(0, t0, t1);
    DRBA_CLK_ADD(1, t1, t2);
    DRBA_CLK_ADD(2, t2, t3);
    DRBA_CLK_ADD(3, t3, t4);
    DRBA_CLK_ADD(4, 0, 1);
    ab ^= 1;
  }
  if constexpr (PL == 2) nf_report(status, DRBA_STATUS_CONV_DMA, nf);
#ifdef DRBA_EXP_CLOCKS
  if ((blockIdx.x == 131 || blockIdx.x == 7) && lane == 0 && clk_acc[4] && (wave == 0 || wave == 5))
    printf("wg %d wave %d: items %lld  barrier %lld  head %lld  blocks %lld  epilogue %lld (clocks per item)  first barrier %lld  total %lld\n",
           (int)blockIdx.x, wave, clk_acc[4], clk_acc[0] / clk_acc[4], clk_acc[1] / clk_acc[4], clk_acc[2] / clk_acc[4],
           clk_acc[3] / clk_acc[4], first_barrier, (long long)__builtin_readcyclecounter() - k_start);
#endif
#endif
}

// ------------------------------------------------------------------------------------------ host side
constexpr int kNum = 1;  // ids 0 .. kNum-1: three bf16 terms; kNum .. 2 kNum-1: the same kernel on two fp16 terms

template <bool PRE, bool RL, int PL>
hipError_t lds_limit() {
  return max_dynamic_lds(reinterpret_cast<const void *>(conv_dma1<PRE, RL, PL>), lds_bytes(PL));
}

// The work counters of a launch: 8 ints, one per XCD band, zero when the launch starts.  Two sets per (device, stream), used
// in turn -- launches on one stream run one after the other, and each zeroes the set of its successor.  Allocated on a
// stream's first launch on its device (torch's default stream is handle 0 on EVERY device: the key holds the device, and the
// counters live in that device's memory); the table is mutex-protected.  mine: the set this launch counts in; next: the set it
// clears.  The sets change roles only after the launch was issued (counters_commit): a launch that fails before it reaches
// the queue (the LDS attribute, a bad configuration) leaves the stream's next launch on a zeroed set.
// Not capturable: the first launch on a stream allocates and synchronises (drba_hip.h, "one exception").
struct CounterSlot {
  int *base;
  unsigned parity;
};
static std::mutex g_counter_mu;
static std::map<std::pair<int, hipStream_t>, CounterSlot> g_counter_tab;

CounterSlot *counters_for(hipStream_t s, int *&mine, int *&next) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_counter_mu);
  const auto key = std::make_pair(dev, s);
  auto it = g_counter_tab.find(key);
  if (it == g_counter_tab.end()) {
    int *p = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&p), 64) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 64) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      (void)hipFree(p);
      return nullptr;
    }
    it = g_counter_tab.emplace(key, CounterSlot{p, 0u}).first;
  }
  CounterSlot &sl = it->second;  // (std::map: the address of an element is stable)
  mine = sl.base + 8 * (sl.parity & 1u);
  next = sl.base + 8 * ((sl.parity & 1u) ^ 1u);
  return &sl;
}
// drba_conv_state_reset (include/drba_hip.h): the stream's entry is made if it is missing (that part allocates), both sets are
// zeroed ON THE STREAM and the roles start over -- the state every launch sequence can begin from, whatever ran before.
int counters_reset(hipStream_t s) {
  int *mine = nullptr, *next = nullptr;
  CounterSlot *sl = counters_for(s, mine, next);
  if (!sl) return DRBA_ELAUNCH;
  if (hipMemsetAsync(sl->base, 0, 64, s) != hipSuccess) return DRBA_ELAUNCH;
  std::lock_guard<std::mutex> lock(g_counter_mu);
  sl->parity = 0u;
  return DRBA_OK;
}
void counters_commit(CounterSlot *sl) {
  std::lock_guard<std::mutex> lock(g_counter_mu);
  sl->parity ^= 1u;
}

template <int PL>
int launch(const float *in, const float *wpk, const float *bias, const float *beta, const float *res, const float *res2,
           float *out, int N, int H, int W, int Cout, int act, float post_slope, int pre_act, float pre_slope, hipStream_t s) {
  const int nbx = (W + TW - 1) / TW, nby = (H + TH - 1) / TH;
  const long long total = (long long)nbx * nby * N;
  if (total >= (1ll << 31)) return DRBA_EUNSUPPORTED;
  long long grid = 256;  // one persistent workgroup per CU
  if (grid > total) grid = (total + 7) / 8 * 8;
  dim3 g((unsigned)grid);
  const u32x4 *wf = reinterpret_cast<const u32x4 *>(wpk);
  int *counters = nullptr, *counters_next = nullptr;
  CounterSlot *slot = counters_for(s, counters, counters_next);
  if (!slot) return DRBA_ELAUNCH;
  auto go = [&](auto kernel, hipError_t lds_ok) -> int {
    if (lds_ok != hipSuccess) return DRBA_ELAUNCH;
    DRBA_LAUNCH(kernel, g, dim3(NTHREADS), lds_bytes(PL), s, in, wf, bias, beta, res, res2, out, H, W, Cout, act, post_slope,
                pre_slope, nbx, nby, (int)total, counters, counters_next, PL == 2 ? status_bytes() : nullptr);
    if (hipPeekAtLastError() != hipSuccess) return DRBA_ELAUNCH;  // not issued: the sets keep their roles
    counters_commit(slot);
    return DRBA_OK;
  };
  const bool rl = res && res == in && !res2 && !pre_act && Cout == CK;
  const int rc = rl ? go(conv_dma1<false, true, PL>, lds_limit<false, true, PL>())
                    : (pre_act ? go(conv_dma1<true, false, PL>, lds_limit<true, false, PL>())
                               : go(conv_dma1<false, false, PL>, lds_limit<false, false, PL>()));
  if (rc != DRBA_OK) return rc;
  DRBA_CHECK_LAUNCH();
  return DRBA_OK;
}

}  // namespace drba_conv_dma

namespace drba {

int conv_dma_num_cfgs() { return drba_conv_dma::kNum; }
int conv_dma_f16_first() { return drba_conv_dma::kNum; }
static int planes_of(int id) { return id < drba_conv_dma::kNum ? 3 : 2; }

bool conv_dma_supports(int Cin, int Cout, int id) {
  return id >= 0 && id < 2 * drba_conv_dma::kNum && Cin == drba_conv_dma::CK && Cout > 0 && Cout <= drba_conv_dma::NTC;
}

size_t conv_dma_packed_floats(int Cin, int Cout, int id) {
  return conv_dma_supports(Cin, Cout, id) ? (size_t)drba_conv_dma::w_bytes(planes_of(id)) / 4 : 0;
}

// packed (16-byte units): [dy][dx][nt][plane h/m/l or h/l][lane] = 8 x 16 bit (split_weight_terms), element i =
//   w[nt*16 + (lane & 15)][4*i + (lane >> 4)][3*dy + dx], zero outside Cout
int conv_dma_pack(const float *w, float *packed, int Cin, int Cout, int id) {
  using namespace drba_conv_dma;
  if (!w || !packed || !conv_dma_supports(Cin, Cout, id)) return DRBA_EINVAL;
  const int PL = planes_of(id);
  if (PL == 2 && !two_term_weights_ok(w, (size_t)Cout * Cin * 9)) return DRBA_EUNSUPPORTED;
  memset(packed, 0, w_bytes(PL));
  unsigned short *dst = reinterpret_cast<unsigned short *>(packed);
  for (int tap = 0; tap < 9; ++tap)
    for (int nt = 0; nt < NT; ++nt)
      for (int lane = 0; lane < 64; ++lane) {
        const int co = nt * 16 + (lane & 15);
        if (co >= Cout) continue;
        for (int i = 0; i < 8; ++i) {
          const int ci = 4 * i + (lane >> 4);
          unsigned short term[3];
          split_weight_terms(w[((size_t)co * Cin + ci) * 9 + tap], PL, term);
          for (int pl = 0; pl < PL; ++pl) {
            const size_t unit = ((size_t)tap * NT + nt) * PL + pl;
            dst[(unit * 64 + lane) * 8 + i] = term[pl];
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
  return planes_of(id) == 3 ? launch<3>(in, packed_w, bias, beta, residual, residual2, out, N, H, W, Cout, act, post_slope, pre_act,
                                        pre_slope, (hipStream_t)stream)
                            : launch<2>(in, packed_w, bias, beta, residual, residual2, out, N, H, W, Cout, act, post_slope, pre_act,
                                        pre_slope, (hipStream_t)stream);
}

}  // namespace drba

extern "C" int drba_conv_state_reset(void *stream) { return drba_conv_dma::counters_reset((hipStream_t)stream); }
