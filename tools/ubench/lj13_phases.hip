// Phase timeline of the LJ13 logp+force kernel at the metric's batch (development aid; results in DESIGN.md 4.2).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/ubench/lj13_phases tools/ubench/lj13_phases.hip \
//         -Lpita_amd -lpita_hip -Wl,-rpath,$PWD/pita_amd && tools/ubench/lj13_phases [walkers]
// Re-uses the product's tile routine (lj13_body of csrc/energy_kernels.hip, included as source) inside a copy of
// lj13_kernel<2> that stamps s_memtime per block: entry, coordinates staged, pair loop done, stores issued, stores
// retired.  Variants: the product's 256-thread blocks (128 walkers) and 128-thread blocks (64 walkers), and the same
// kernels with the pair loop removed (memory phases only) or the global traffic removed (compute only).
#include "../../pita_amd/csrc/energy_kernels.hip"

#include <algorithm>
#include <vector>

namespace pita {

template <int THREADS, int MODE>  // MODE 0: full, 1: no pair loop (memory only), 2: no global traffic (compute only)
__global__ void __launch_bounds__(THREADS, 1024 / THREADS) lj13_prof_kernel(const float* __restrict__ x, float* __restrict__ logp,
                                                                           float* __restrict__ force, long long B, PairParams p,
                                                                           long long* __restrict__ prof) {
  constexpr int D = 39, P = 2, WPB = THREADS / P;
  __shared__ __attribute__((aligned(16))) float fb[P][WPB * D];
  __shared__ float es[P][WPB];
  const int tid = threadIdx.x;
  const int half = tid / WPB, wl = tid - half * WPB;
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long w0 = (long long)blockIdx.x * WPB;
  const int nw = (int)((B - w0) < WPB ? (B - w0) : WPB);
  const int nfl = nw * D;
  if (MODE != 2) {
    const float4* src4 = reinterpret_cast<const float4*>(x + w0 * D);
    float4* dst4 = reinterpret_cast<float4*>(&fb[0][0]);
    for (int q = tid; q < nfl / 4; q += THREADS) dst4[q] = src4[q];
  } else {
    for (int q = tid; q < nfl; q += THREADS) fb[0][q] = 0.37f * (float)((q * 7 + blockIdx.x) % 41) - 7.f;
  }
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memtime();
  const bool act = wl < nw;
  const int row = (act ? wl : 0) * D;
  if (MODE != 1) {
    if (half == 0) lj13_body<P, 1, true>(&fb[0][row], &fb[0][row], &es[0][wl], p, act);
    else lj13_body<P, 2, true>(&fb[0][row], &fb[P - 1][row], &es[P - 1][wl], p, act);
  } else {
    __syncthreads();
    if (act) { fb[P - 1][row + half] = fb[0][row + half]; es[half][wl] = 1.f; }
  }
  __syncthreads();
  const long long t2 = __builtin_amdgcn_s_memtime();
  if (MODE != 2) {
    float4* dst4 = reinterpret_cast<float4*>(force + w0 * D);
    for (int q = tid; q < nfl / 4; q += THREADS) {
      float4 v = reinterpret_cast<const float4*>(&fb[0][0])[q];
      const float4 u = reinterpret_cast<const float4*>(&fb[P - 1][0])[q];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      dst4[q] = v;
    }
    if (tid < nw) logp[w0 + tid] = -p.inv_T * (es[0][tid] + es[P - 1][tid]);
  } else if (tid == 0 && fb[0][1] + fb[1][2] == 12345.f) {
    logp[0] = es[0][0];
  }
  const long long t3 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t4 = __builtin_amdgcn_s_memtime();
  const long long r4 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0 && prof) {
    long long* o = prof + (long long)blockIdx.x * 7;
    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = r0; o[6] = r4;
  }
}

}  // namespace pita

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int THREADS, int MODE>
static int run(const char* name, const float* x, float* logp, float* force, long long B, long long* prof_d) {
  using namespace pita;
  PairParams p{};
  p.inv_T = 1.f; p.energy_factor = 1.f; p.dist_eps = 1e-6f; p.eps = 1.f; p.rm2 = 1.f; p.osc_scale = 1.f;
  p.cw = -24.f; p.co = -1.f;
  const int WPB = THREADS / 2;
  const unsigned grid = (unsigned)((B + WPB - 1) / WPB);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL((lj13_prof_kernel<THREADS, MODE>), dim3(grid), dim3(THREADS), 0, 0, x, logp, force, B, p, (long long*)nullptr);
  CK(hipEventRecord(e0));
  const int K = 500;
  for (int i = 0; i < K; ++i) hipLaunchKernelGGL((lj13_prof_kernel<THREADS, MODE>), dim3(grid), dim3(THREADS), 0, 0, x, logp, force, B, p, (long long*)nullptr);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / K;
  hipLaunchKernelGGL((lj13_prof_kernel<THREADS, MODE>), dim3(grid), dim3(THREADS), 0, 0, x, logp, force, B, p, prof_d);
  CK(hipDeviceSynchronize());
  std::vector<long long> h((size_t)grid * 7);
  CK(hipMemcpy(h.data(), prof_d, h.size() * 8, hipMemcpyDeviceToHost));
  // s_memtime (shader clocks) has a different origin on every XCD: only differences inside a block are used.  The
  // cross-block timeline is s_memrealtime (100 MHz, chip-wide); a block's shader-clock stamps are placed on it by
  // scaling its own (t - t0) with its own realtime / shader-clock ratio.
  double ticks = 0, real = 0;
  long long r_org = h[5], r_end = 0;
  for (unsigned b = 0; b < grid; ++b) {
    ticks += (double)(h[b * 7 + 4] - h[b * 7]);
    real += (double)(h[b * 7 + 6] - h[b * 7 + 5]);
    r_org = std::min(r_org, h[b * 7 + 5]);
    r_end = std::max(r_end, h[b * 7 + 6]);
  }
  const double ghz = ticks / (real * 10.0);  // shader clocks per ns
  printf("%-30s %6.2f us/launch (%.2f TB/s); first entry -> last retire %.2f us; shader clock %.2f GHz\n", name, us,
         B * 316.0 / us / 1e6, (double)(r_end - r_org) * 0.01, ghz);
  const char* ph[5] = {"entry", "staged", "pairs done", "stores issued", "stores retired"};
  for (int k = 0; k < 5; ++k) {
    std::vector<double> v(grid);
    for (unsigned b = 0; b < grid; ++b)
      v[b] = (double)(h[b * 7 + 5] - r_org) * 0.01 + (double)(h[b * 7 + k] - h[b * 7]) / ghz * 1e-3;
    std::sort(v.begin(), v.end());
    printf("      %-15s min %5.2f  5%% %5.2f  50%% %5.2f  95%% %5.2f  max %5.2f us\n", ph[k], v[0], v[grid / 20], v[grid / 2],
           v[grid - 1 - grid / 20], v[grid - 1]);
  }
  double d[4] = {0, 0, 0, 0};
  for (unsigned b = 0; b < grid; ++b) for (int k = 0; k < 4; ++k) d[k] += (double)(h[b * 7 + k + 1] - h[b * 7 + k]);
  printf("      mean per block: stage-in %.0f, pair loop %.0f, store issue %.0f, store retire %.0f shader clocks "
         "(%.2f, %.2f, %.2f, %.2f us)\n", d[0] / grid, d[1] / grid, d[2] / grid, d[3] / grid, d[0] / grid / ghz * 1e-3,
         d[1] / grid / ghz * 1e-3, d[2] / grid / ghz * 1e-3, d[3] / grid / ghz * 1e-3);
  return 0;
}

int main(int argc, char** argv) {
  const long long B = argc > 1 ? atoll(argv[1]) : 65536;
  float *x, *logp, *force;
  long long* prof;
  CK(hipMalloc(&x, B * 39 * 4)); CK(hipMalloc(&logp, B * 4)); CK(hipMalloc(&force, B * 39 * 4));
  CK(hipMalloc(&prof, (B / 64 + 8) * 7 * 8));
  std::vector<float> h((size_t)B * 39);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.0f; }
  CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  if (run<256, 0>("256 threads (product), full", x, logp, force, B, prof)) return 1;
  if (run<256, 1>("256 threads, memory only", x, logp, force, B, prof)) return 1;
  if (run<256, 2>("256 threads, compute only", x, logp, force, B, prof)) return 1;
  if (run<128, 0>("128 threads, full", x, logp, force, B, prof)) return 1;
  if (run<128, 1>("128 threads, memory only", x, logp, force, B, prof)) return 1;
  if (run<128, 2>("128 threads, compute only", x, logp, force, B, prof)) return 1;
  return 0;
}
