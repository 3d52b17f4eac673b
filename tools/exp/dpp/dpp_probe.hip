// What do the whole-wave DPP shifts do on gfx950?  out[ctrl][lane] = update_dpp(old = -1, src = lane, ctrl)
//   hipcc --offload-arch=gfx950 -O2 tools/exp/dpp/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL>
__device__ int probe(int v) { return __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xf, 0xf, false); }
__global__ void k(int *out) {
  const int l = threadIdx.x;
  out[0 * 64 + l] = probe<0x130>(l);  // wave_shl:1
  out[1 * 64 + l] = probe<0x138>(l);  // wave_shr:1
  out[2 * 64 + l] = probe<0x101>(l);  // row_shl:1
  out[3 * 64 + l] = probe<0x111>(l);  // row_shr:1
  out[4 * 64 + l] = probe<0x134>(l);  // wave_rol:1
  out[5 * 64 + l] = probe<0x13C>(l);  // wave_ror:1
}
int main() {
  int *d, h[6 * 64];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[6] = {"wave_shl:1", "wave_shr:1", "row_shl:1", "row_shr:1", "wave_rol:1", "wave_ror:1"};
  for (int c = 0; c < 6; ++c) {
    printf("%-10s:", names[c]);
    for (int l = 0; l < 64; ++l) printf(" %d", h[c * 64 + l]);
    printf("\n");
  }
  return 0;
}
