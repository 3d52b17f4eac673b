// Where does a split-bf16 convolution tile spend its clocks?  Builds conv_split.hip with per-phase cycle counters
// (DRBA_PHASE_CLOCKS) and runs one ResConv layer shape.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics tools/exp/conv_split_phases.hip drba_amd/csrc/api_misc.hip -o /tmp/csp
//   /tmp/csp <cfg 0-4> N C H W
#define DRBA_PHASE_CLOCKS 1
#include "../../drba_amd/csrc/conv_split.hip"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

int main(int argc, char **argv) {
  const int cfg = argc > 1 ? atoi(argv[1]) : 4, N = argc > 2 ? atoi(argv[2]) : 2, C = argc > 3 ? atoi(argv[3]) : 32;
  const int H = argc > 4 ? atoi(argv[4]) : 272, W = argc > 5 ? atoi(argv[5]) : 480;
  const size_t n = (size_t)N * C * H * W;
  std::vector<float> hx(n), hw((size_t)C * C * 9), hb(C, 0.1f), hbeta(C, 1.0f);
  srand(1);
  for (auto &v : hx) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto &v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  const size_t pf = drba::conv_split_packed_floats(C, C, cfg);
  std::vector<float> hp(pf);
  if (drba::conv_split_pack(hw.data(), hp.data(), C, C, cfg) != DRBA_OK) return 1;
  float *x, *y, *p, *b, *beta;
  hipMalloc(&x, n * 4), hipMalloc(&y, n * 4), hipMalloc(&p, pf * 4), hipMalloc(&b, C * 4), hipMalloc(&beta, C * 4);
  hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(p, hp.data(), pf * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), C * 4, hipMemcpyHostToDevice);
  hipMemcpy(beta, hbeta.data(), C * 4, hipMemcpyHostToDevice);
  // argv[6] = output channels of a TRANSPOSED convolution 4x4 s2 + PixelShuffle (deconv_split ids: cfg counts from the first two-term id)
  const int dcout = argc > 6 ? atoi(argv[6]) : 0;
  float *dp = nullptr, *dy = nullptr;
  int dcfg = 0;
  if (dcout) {
    dcfg = drba::deconv_split_f16_first() + cfg;
    const size_t dpf = drba::deconv_split_packed_floats(C, dcout, dcfg);
    if (!dpf) return 3;
    std::vector<float> dw((size_t)C * dcout * 16), dpk(dpf);
    for (auto &v : dw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    if (drba::deconv_split_pack(dw.data(), dpk.data(), C, dcout, dcfg) != DRBA_OK) return 4;
    hipMalloc(&dp, dpf * 4), hipMalloc(&dy, (size_t)N * dcout * 4 * H * W * 4);
    hipMemcpy(dp, dpk.data(), dpf * 4, hipMemcpyHostToDevice);
  }
  auto run = [&]() {
    if (dcout) return drba::deconv_split_launch(dcfg, x, dp, b, dy, N, C, H, W, dcout, 1, 0, 0.f, nullptr);
    if (getenv("DRBA_PHASE_NORES")) return drba::conv_split_launch(cfg, x, p, b, nullptr, nullptr, nullptr, y, N, C, H, W, C, 1, 0.f, 0, 0.f, nullptr);
    return drba::conv_split_launch(cfg, x, p, b, beta, x, nullptr, y, N, C, H, W, C, 1, 0.f, 0, 0.f, nullptr);
  };
  for (int i = 0; i < 3; ++i)
    if (run() != DRBA_OK) return 2;
  hipDeviceSynchronize();
  static long long buf[1024 * 4 * 4];
  hipMemcpyToSymbol(HIP_SYMBOL(drba_conv_split::g_phase), buf, sizeof(buf));
#ifdef DRBA_HAVE_LOADER_CLOCKS  // (tools/exp/conv_split_loader: the loader-wave variants keep a second table)
  hipMemcpyToSymbol(HIP_SYMBOL(drba_conv_split::g_loader), buf, sizeof(long long) * 1024 * 4);
#endif
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int reps = 10;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) run();
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpyFromSymbol(buf, HIP_SYMBOL(drba_conv_split::g_phase), sizeof(buf));
  long long ph[4] = {0, 0, 0, 0};
  for (int i = 0; i < 1024 * 4; ++i)
    for (int k = 0; k < 4; ++k) ph[k] += buf[i * 4 + k];
  const double tiles = (double)ph[3] / 4.0;  // 4 waves count each tile
  printf("cfg %d N%d C%d %dx%d: %.1f us/launch (with counters), %.0f tiles/launch\n", cfg, N, C, H, W, ms / reps * 1e3, tiles / reps);
  printf("  per tile per wave: stage(+wait) %.0f clk, mfma phase %.0f clk, epilogue %.0f clk\n", ph[0] / (double)ph[3],
         ph[1] / (double)ph[3], ph[2] / (double)ph[3]);
#ifdef DRBA_HAVE_LOADER_CLOCKS
  {
    static long long lb[1024 * 4];
    hipMemcpyFromSymbol(lb, HIP_SYMBOL(drba_conv_split::g_loader), sizeof(lb));
    long long l[3] = {0, 0, 0};
    for (int i = 0; i < 1024; ++i)
      for (int k = 0; k < 3; ++k) l[k] += lb[i * 4 + k];
    if (l[0] + l[1] + l[2])
      printf("  loader wave per tile: fetch issue %.0f clk, wait + split + stage %.0f clk, barrier wait %.0f clk\n", l[0] / tiles, l[1] / tiles, l[2] / tiles);
  }
#endif
  // per workgroup (its four waves averaged): how the phases are distributed over the co-resident workgroups of a CU
  if (getenv("DRBA_PHASE_DIST")) {
    std::vector<double> m;
    for (int b = 0; b < 1024; ++b) {
      long long t = 0, mf = 0;
      for (int w = 0; w < 4; ++w) t += buf[(b * 4 + w) * 4 + 3], mf += buf[(b * 4 + w) * 4 + 1];
      if (t) m.push_back((double)mf / t);
    }
    std::vector<double> srt = m;
    std::sort(srt.begin(), srt.end());
    const size_t n_ = srt.size();
    if (n_) printf("  mfma phase per tile over %zu workgroups: min %.0f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f\n", n_, srt[0], srt[n_ / 10],
                   srt[n_ / 2], srt[n_ * 9 / 10], srt[n_ - 1]);
    printf("  first 16 workgroups and workgroups 256..271:");
    for (size_t b = 0; b < 16 && b < n_; ++b) printf(" %.0f", m[b]);
    printf(" |");
    for (size_t b = 256; b < 272 && b < n_; ++b) printf(" %.0f", m[b]);
    printf("\n");
  }
  return 0;
}
