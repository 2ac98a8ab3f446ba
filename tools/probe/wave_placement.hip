// Where do the four wavefronts of a 256-thread workgroup land?  Records HW_ID / XCC_ID of every wave of a K1-shaped launch
// (1024 workgroups x 4 waves, 40 KB LDS each -> 4 workgroups per CU) and prints, per SIMD, which in-workgroup wave indices share it.
// Build: hipcc --offload-arch=gfx950 -O2 -o wave_placement tools/probe/wave_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
#include <tuple>
__global__ __launch_bounds__(256, 4) void probe(uint32_t *out, int spin) {
  extern __shared__ uint8_t smem[];
  smem[threadIdx.x] = (uint8_t)threadIdx.x;
  __syncthreads();
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) { __builtin_amdgcn_s_sleep(8); }
  if ((threadIdx.x & 63) == 0) { const int i = blockIdx.x * 4 + (threadIdx.x >> 6); out[2 * i] = hw; out[2 * i + 1] = xcc + smem[5] * 0; }
}
int main() {
  const int NWG = 1024;
  uint32_t *d; hipMalloc(&d, NWG * 4 * 8);
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
  probe<<<NWG, 256, 40960>>>(d, 20000);   // 200 us at 100 MHz: every workgroup is resident at the same time
  std::vector<uint32_t> h(NWG * 8); hipMemcpy(h.data(), d, NWG * 4 * 8, hipMemcpyDeviceToHost);
  std::map<std::tuple<int,int,int,int,int>, std::vector<std::pair<int,int>>> simd;   // (xcc, se, sh, cu, simd) -> (wg, wave)
  std::map<std::tuple<int,int,int,int>, std::vector<int>> cu;
  for (int i = 0; i < NWG * 4; i++) {
    const uint32_t hw = h[2 * i], x = h[2 * i + 1] & 15;
    const int sid = (hw >> 4) & 3, cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    simd[{(int)x, se, sh, cuid, sid}].push_back({i / 4, i % 4});
    if (i % 4 == 0) cu[{(int)x, se, sh, cuid}].push_back(i / 4);
  }
  printf("CUs seen %zu, SIMDs seen %zu\n", cu.size(), simd.size());
  int hist[5] = {0}, shown = 0;
  for (auto &kv : simd) {
    int mask = 0; for (auto &p : kv.second) mask |= 1 << p.second;
    hist[__builtin_popcount(mask)]++;
    if (shown < 12) { shown++; printf("xcc%d se%d sh%d cu%d simd%d:", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first), std::get<4>(kv.first)); for (auto &p : kv.second) printf(" wg%d.w%d", p.first, p.second); printf("\n"); }
  }
  printf("distinct in-workgroup wave indices per SIMD: 1:%d 2:%d 3:%d 4:%d\n", hist[1], hist[2], hist[3], hist[4]);
  shown = 0;
  for (auto &kv : cu) if (shown++ < 6) { printf("xcc%d se%d sh%d cu%d: workgroups", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first)); for (int w : kv.second) printf(" %d", w); printf("\n"); }
  return 0;
}
