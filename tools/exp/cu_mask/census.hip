// Which CUs does a hipExtStreamCreateWithCUMask stream run on?  (round 6: the CU-partitioned side streams)
//   hipcc --offload-arch=gfx950 -O2 tools/exp/cu_mask/census.hip -o /tmp/census && /tmp/census
// Launches a grid of short spinning workgroups on streams with various masks and prints, per mask, how many distinct
// (XCC, SE, CU) triples ran workgroups and how they spread over the XCDs -- i.e. how mask bit i maps to a physical CU.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <vector>

__global__ void census(unsigned *out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
}

static void run(const char *name, const std::vector<unsigned> &mask) {
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data()) != hipSuccess) {
    printf("%s: stream creation failed\n", name);
    return;
  }
  const int n = 4096;
  unsigned *d, h[n];
  hipMalloc(&d, n * 4);
  hipMemsetAsync(d, 0xff, n * 4, s);
  hipLaunchKernelGGL(census, dim3(n), dim3(64), 0, s, d, 20000);
  hipMemcpyAsync(h, d, n * 4, hipMemcpyDeviceToHost, s);
  hipStreamSynchronize(s);
  std::map<int, std::set<int>> per_xcc;
  for (int i = 0; i < n; ++i) {
    const int xcc = h[i] >> 16, se = (h[i] >> 13) & 7, sh = (h[i] >> 12) & 1, cu = (h[i] >> 8) & 15;
    per_xcc[xcc].insert(se * 64 + sh * 16 + cu);
  }
  int total = 0;
  printf("%-28s", name);
  for (auto &kv : per_xcc) {
    printf(" xcc%d:%zu", kv.first, kv.second.size());
    total += (int)kv.second.size();
  }
  printf("  = %d CUs\n", total);
  if (total <= 16) {
    for (auto &kv : per_xcc) {
      printf("    xcc%d:", kv.first);
      for (int c : kv.second) printf(" se%d.cu%d", c / 64, c % 16);
      printf("\n");
    }
  }
  hipFree(d);
  hipStreamDestroy(s);
}

int main() {
  auto bits = [](std::initializer_list<std::pair<int, int>> ranges) {
    std::vector<unsigned> m(8, 0u);
    for (auto r : ranges)
      for (int i = r.first; i < r.second; ++i) m[i >> 5] |= 1u << (i & 31);
    return m;
  };
  run("all 256", bits({{0, 256}}));
  run("bits 0..7", bits({{0, 8}}));
  run("bits 0..31", bits({{0, 32}}));
  run("bits 224..255", bits({{224, 256}}));
  run("bits 0..127", bits({{0, 128}}));
  run("bit 0", bits({{0, 1}}));
  run("bit 1", bits({{1, 2}}));
  run("bit 8", bits({{8, 9}}));
  run("bits 0,8,16,24", [&] { auto m = bits({}); m[0] = 0x01010101u; return m; }());
  run("bits 32..63", bits({{32, 64}}));
  return 0;
}
