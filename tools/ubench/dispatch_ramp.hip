// How long does the dispatcher take to START the waves of a short launch?  (development aid)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dispatch_ramp tools/ubench/dispatch_ramp.hip && tools/ubench/dispatch_ramp
// Every wave stamps s_memrealtime (100 MHz, chip-wide) at entry, spins ~2 us so that no CU slot is recycled inside the
// ramp, and exits.  Reported: first -> last wave entry, for several grid shapes, LDS sizes and register footprints.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int BIGREGS>
__global__ void __launch_bounds__(1024) ramp_kernel(long long* out, int spin) {
  extern __shared__ float lds[];
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  if (BIGREGS) asm volatile("v_mov_b32 v120, 0" ::: "v120");
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    if (lds) lds[wave] = 1.f;
    out[(long long)blockIdx.x * (blockDim.x >> 6) + wave] = r0;
  }
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(4);
}

template <int BIGREGS>
static int run(int blocks, int threads, int lds, long long* out_d) {
  const int waves = blocks * (threads / 64);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(ramp_kernel<BIGREGS>, dim3(blocks), dim3(threads), lds, 0, out_d, 4000);
  CK(hipDeviceSynchronize());
  std::vector<long long> h(waves);
  double spread[5];
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(ramp_kernel<BIGREGS>, dim3(blocks), dim3(threads), lds, 0, out_d, 4000);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), out_d, waves * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    spread[rep] = (double)(h[waves - 1] - h[0]) * 0.01;
  }
  std::sort(spread, spread + 5);
  printf("%5d blocks x %4d threads (%5d waves), LDS %6d B, %s: first -> last wave entry %.2f us (median of 5; min %.2f, max %.2f) "
         "= %.1f ns per wave chip-wide\n", blocks, threads, waves, lds, BIGREGS ? ">=121 VGPRs" : "few VGPRs ", spread[2], spread[0],
         spread[4], spread[2] * 1e3 / waves);
  return 0;
}

int main() {
  long long* out;
  CK(hipMalloc(&out, 8 * 65536));
  const int shapes[][2] = {{256, 256}, {512, 256}, {1024, 256}, {1024, 128}, {2048, 64}, {1024, 64}, {256, 512}, {256, 1024}, {512, 512}, {2048, 256}, {4096, 256}};
  for (auto& s : shapes) if (run<0>(s[0], s[1], 0, out)) return 1;
  if (run<0>(512, 256, 40960, out)) return 1;
  if (run<1>(512, 256, 0, out)) return 1;
  if (run<1>(512, 256, 40960, out)) return 1;
  if (run<1>(1024, 128, 20480, out)) return 1;
  return 0;
}
