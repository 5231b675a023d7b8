// How many wait states does gfx950 need between v_mfma_f32_16x16x32_f16 and a vector instruction that reads its result?
// hipcc (ROCm 7.2) pads 8; egnn_div_walker_kernel.hip saw stale last-pass rows (lanes 48..63) with two waves per SIMD.
// Development aid: the sequence is written in inline asm with physical registers so that the number of wait states is
// exactly what the source says.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_read_hazard tools/ubench/mfma_read_hazard.hip && tools/ubench/mfma_read_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// CHAIN dependent MFMAs into v[100:103], then NOPS wait states, then four v_mov reading the result.
// The expected value is computed by the same MFMAs followed by a long wait.
template <int NOPS, int CHAIN>
__device__ __forceinline__ void probe(f16x8 a, f16x8 b, float (&got)[4]) {
  asm volatile(
      "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\ts_nop 7\n\t"
      "v_mfma_f32_16x16x32_f16 v[100:103], %4, %5, v[100:103]\n\t"
      ".if %7 > 1\n\tv_mfma_f32_16x16x32_f16 v[100:103], %4, %5, v[100:103]\n\t.endif\n\t"
      ".if %7 > 2\n\tv_mfma_f32_16x16x32_f16 v[100:103], %4, %5, v[100:103]\n\t.endif\n\t"
      ".if %6 > 0\n\ts_nop %6 - 1\n\t.endif\n\t"
      "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103\n\t"
      "s_nop 15\n\ts_nop 15"
      : "=&v"(got[0]), "=&v"(got[1]), "=&v"(got[2]), "=&v"(got[3])
      : "v"(a), "v"(b), "i"(NOPS), "i"(CHAIN)
      : "v100", "v101", "v102", "v103");
}

// Second probe: a dependent MFMA whose destination is NOT its C operand (the compiler produces this when an accumulator is
// updated conditionally): v[100:103] = A B + v[100:103]; NOPS wait states; v[104:107] = A B + v[100:103].
template <int NOPS>
__device__ __forceinline__ void probe_srcc(f16x8 a, f16x8 b, float (&got)[4]) {
  asm volatile(
      "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\ts_nop 7\n\t"
      "v_mfma_f32_16x16x32_f16 v[100:103], %4, %5, v[100:103]\n\t"
      ".if %6 > 0\n\ts_nop %6 - 1\n\t.endif\n\t"
      "v_mfma_f32_16x16x32_f16 v[104:107], %4, %5, v[100:103]\n\t"
      "s_nop 15\n\ts_nop 15\n\t"
      "v_mov_b32 %0, v104\n\tv_mov_b32 %1, v105\n\tv_mov_b32 %2, v106\n\tv_mov_b32 %3, v107\n\t"
      "s_nop 15"
      : "=&v"(got[0]), "=&v"(got[1]), "=&v"(got[2]), "=&v"(got[3])
      : "v"(a), "v"(b), "i"(NOPS)
      : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
}
template <int NOPS>
__global__ void __launch_bounds__(512) kern_srcc(const float* seed, int iters, int busy_partner, unsigned* bad_by_group) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(1.0f + 0.125f * ((lane + e) & 7)); b[e] = (_Float16)(0.5f + 0.0625f * ((lane * 3 + e) & 15)); }
  if (wave >= 4 && busy_partner) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters * 6; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[q], 0, 0, 0);
    }
    if (acc[0].x + acc[1].y + acc[2].z + acc[3].w == 12345.f) bad_by_group[7] = 1;
    return;
  }
  if (wave >= 4) return;
  unsigned bad[4] = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    b[it & 7] = (_Float16)(seed[it & 1023] + (float)(lane & 15) * 0.03125f);
    float want[4], got[4];
    probe_srcc<32>(a, b, want);
    probe_srcc<NOPS>(a, b, got);
#pragma unroll
    for (int q = 0; q < 4; ++q) bad[q] += (got[q] != want[q]) ? 1u : 0u;
  }
  const unsigned n = bad[0] + bad[1] + bad[2] + bad[3];
  if (n) atomicAdd(&bad_by_group[(lane >> 4)], n);
}
template <int NOPS>
static void run_srcc(const float* d_seed, unsigned* d_bad, int busy, int blocks) {
  unsigned h[8] = {0};
  CHECK(hipMemset(d_bad, 0, sizeof(h)));
  hipLaunchKernelGGL((kern_srcc<NOPS>), dim3(blocks), dim3(512), 0, 0, d_seed, 200, busy, d_bad);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost));
  printf("  MFMA -> %2d wait states -> MFMA reading it as C into ANOTHER destination, partner %s: wrong results by lane group g=0..3: %u %u %u %u\n",
         NOPS, busy ? "busy" : "idle", h[0], h[1], h[2], h[3]);
}

// Third probe: M back-to-back INDEPENDENT MFMAs (v[100:103] ... ), the LAST one's result read after NOPS wait states.
template <int NOPS, int M>
__device__ __forceinline__ void probe_burst(f16x8 a, f16x8 b, float (&got)[4]) {
  asm volatile(
      "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
      "v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\t"
      "v_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\t"
      "v_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\ts_nop 7\n\t"
      ".rept %7\n\t"
      "v_mfma_f32_16x16x32_f16 v[104:107], %4, %5, v[104:107]\n\t"
      "v_mfma_f32_16x16x32_f16 v[108:111], %4, %5, v[108:111]\n\t"
      "v_mfma_f32_16x16x32_f16 v[112:115], %4, %5, v[112:115]\n\t"
      ".endr\n\t"
      "v_mfma_f32_16x16x32_f16 v[100:103], %4, %5, v[100:103]\n\t"
      ".if %6 > 0\n\ts_nop %6 - 1\n\t.endif\n\t"
      "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103\n\t"
      "s_nop 15\n\ts_nop 15"
      : "=&v"(got[0]), "=&v"(got[1]), "=&v"(got[2]), "=&v"(got[3])
      : "v"(a), "v"(b), "i"(NOPS), "i"(M)
      : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115");
}
template <int NOPS, int M>
__global__ void __launch_bounds__(512) kern_burst(const float* seed, int iters, int busy_partner, unsigned* bad_by_group) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(1.0f + 0.125f * ((lane + e) & 7)); b[e] = (_Float16)(0.5f + 0.0625f * ((lane * 3 + e) & 15)); }
  if (wave >= 4 && busy_partner) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters * 12; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[q], 0, 0, 0);
    }
    if (acc[0].x + acc[1].y + acc[2].z + acc[3].w == 12345.f) bad_by_group[7] = 1;
    return;
  }
  if (wave >= 4) return;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    b[it & 7] = (_Float16)(seed[it & 1023] + (float)(lane & 15) * 0.03125f);
    float want[4], got[4];
    probe_burst<64, M>(a, b, want);
    probe_burst<NOPS, M>(a, b, got);
#pragma unroll
    for (int q = 0; q < 4; ++q) bad += (got[q] != want[q]) ? 1u : 0u;
  }
  if (bad) atomicAdd(&bad_by_group[(lane >> 4)], bad);
}
template <int NOPS, int M>
static void run_burst(const float* d_seed, unsigned* d_bad, int busy, int blocks) {
  unsigned h[8] = {0};
  CHECK(hipMemset(d_bad, 0, sizeof(h)));
  hipLaunchKernelGGL((kern_burst<NOPS, M>), dim3(blocks), dim3(512), 0, 0, d_seed, 200, busy, d_bad);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost));
  printf("  burst of %2d independent MFMAs, the last one read after %2d wait states, partner %s: stale by lane group g=0..3: %u %u %u %u\n",
         3 * M + 1, NOPS, busy ? "busy" : "idle", h[0], h[1], h[2], h[3]);
}

// Fourth probe (write after read): the MFMA's A operand registers are overwritten by vector instructions NOPS wait states
// after the MFMA has been issued.  A correct machine has latched (or read) the operand by then.
template <int NOPS, int PRE>
__device__ __forceinline__ void probe_war(f16x8 a, f16x8 b, f16x8 junk, float (&got)[4]) {
  asm volatile(
      "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
      "v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\t"
      "v_mov_b32 v120, %4\n\tv_mov_b32 v121, %5\n\tv_mov_b32 v122, %6\n\tv_mov_b32 v123, %7\n\ts_nop 7\n\t"
      ".rept %14\n\tv_mfma_f32_16x16x32_f16 v[104:107], %8, %9, v[104:107]\n\t.endr\n\t"
      "v_mfma_f32_16x16x32_f16 v[100:103], v[120:123], %9, v[100:103]\n\t"
      ".if %15 > 0\n\ts_nop %15 - 1\n\t.endif\n\t"
      "v_mov_b32 v120, %10\n\tv_mov_b32 v121, %11\n\tv_mov_b32 v122, %12\n\tv_mov_b32 v123, %13\n\t"
      "s_nop 15\n\ts_nop 15\n\t"
      "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103\n\t"
      "s_nop 15"
      : "=&v"(got[0]), "=&v"(got[1]), "=&v"(got[2]), "=&v"(got[3])
      : "v"(((unsigned*)&a)[0]), "v"(((unsigned*)&a)[1]), "v"(((unsigned*)&a)[2]), "v"(((unsigned*)&a)[3]), "v"(a), "v"(b),
        "v"(((unsigned*)&junk)[0]), "v"(((unsigned*)&junk)[1]), "v"(((unsigned*)&junk)[2]), "v"(((unsigned*)&junk)[3]), "i"(PRE), "i"(NOPS)
      : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v120", "v121", "v122", "v123");
}
template <int NOPS, int PRE>
__global__ void __launch_bounds__(512) kern_war(const float* seed, int iters, int busy_partner, unsigned* bad_by_group) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x8 a, b, junk;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(1.0f + 0.125f * ((lane + e) & 7)); b[e] = (_Float16)(0.5f + 0.0625f * ((lane * 3 + e) & 15)); junk[e] = (_Float16)(-3.0f - e); }
  if (wave >= 4 && busy_partner) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters * 12; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[q], 0, 0, 0);
    }
    if (acc[0].x + acc[1].y + acc[2].z + acc[3].w == 12345.f) bad_by_group[7] = 1;
    return;
  }
  if (wave >= 4) return;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    b[it & 7] = (_Float16)(seed[it & 1023] + (float)(lane & 15) * 0.03125f);
    float want[4], got[4];
    probe_war<64, PRE>(a, b, junk, want);
    probe_war<NOPS, PRE>(a, b, junk, got);
#pragma unroll
    for (int q = 0; q < 4; ++q) bad += (got[q] != want[q]) ? 1u : 0u;
  }
  if (bad) atomicAdd(&bad_by_group[(lane >> 4)], bad);
}
template <int NOPS, int PRE>
static void run_war(const float* d_seed, unsigned* d_bad, int busy, int blocks) {
  unsigned h[8] = {0};
  CHECK(hipMemset(d_bad, 0, sizeof(h)));
  hipLaunchKernelGGL((kern_war<NOPS, PRE>), dim3(blocks), dim3(512), 0, 0, d_seed, 200, busy, d_bad);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost));
  printf("  %2d MFMAs in front, then MFMA whose A operand is overwritten %2d wait states later, partner %s: wrong by lane group: %u %u %u %u\n",
         PRE, NOPS, busy ? "busy" : "idle", h[0], h[1], h[2], h[3]);
}

// Fifth probe: the destination OVERLAPS an input operand of the same instruction (hipcc allocates it that way for
// v_mfma_f32_16x16x32_f16: no early-clobber on the 4-register destination).  WHICH: 0 = vdst is srcA, 1 = vdst is srcB.
template <int WHICH, int PRE>
__device__ __forceinline__ void probe_overlap(f16x8 a, f16x8 b, float (&got)[4], float (&want)[4]) {
  // reference: distinct registers
  asm volatile(
      "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\ts_nop 7\n\t"
      "v_mfma_f32_16x16x32_f16 v[100:103], %4, %5, v[100:103]\n\t"
      "s_nop 15\n\ts_nop 15\n\t"
      "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103\n\ts_nop 15"
      : "=&v"(want[0]), "=&v"(want[1]), "=&v"(want[2]), "=&v"(want[3]) : "v"(a), "v"(b) : "v100", "v101", "v102", "v103");
  // overlapping: operand copied into v[120:123], which is also the destination (C = 0 through v[100:103])
  asm volatile(
      "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
      "v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\t"
      "v_mov_b32 v120, %4\n\tv_mov_b32 v121, %5\n\tv_mov_b32 v122, %6\n\tv_mov_b32 v123, %7\n\ts_nop 7\n\t"
      ".rept %11\n\tv_mfma_f32_16x16x32_f16 v[104:107], %8, %9, v[104:107]\n\t.endr\n\t"
      ".if %10 == 0\n\tv_mfma_f32_16x16x32_f16 v[120:123], v[120:123], %9, v[100:103]\n\t"
      ".else\n\tv_mfma_f32_16x16x32_f16 v[120:123], %8, v[120:123], v[100:103]\n\t.endif\n\t"
      "s_nop 15\n\ts_nop 15\n\t"
      "v_mov_b32 %0, v120\n\tv_mov_b32 %1, v121\n\tv_mov_b32 %2, v122\n\tv_mov_b32 %3, v123\n\ts_nop 15"
      : "=&v"(got[0]), "=&v"(got[1]), "=&v"(got[2]), "=&v"(got[3])
      : "v"(((unsigned*)(WHICH == 0 ? &a : &b))[0]), "v"(((unsigned*)(WHICH == 0 ? &a : &b))[1]),
        "v"(((unsigned*)(WHICH == 0 ? &a : &b))[2]), "v"(((unsigned*)(WHICH == 0 ? &a : &b))[3]), "v"(a), "v"(b), "i"(WHICH), "i"(PRE)
      : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v120", "v121", "v122", "v123");
}
template <int WHICH, int PRE>
__global__ void __launch_bounds__(512) kern_overlap(const float* seed, int iters, int busy_partner, unsigned* bad_by_group) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(1.0f + 0.125f * ((lane + e) & 7)); b[e] = (_Float16)(0.5f + 0.0625f * ((lane * 3 + e) & 15)); }
  if (wave >= 4 && busy_partner) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters * 12; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[q], 0, 0, 0);
    }
    if (acc[0].x + acc[1].y + acc[2].z + acc[3].w == 12345.f) bad_by_group[7] = 1;
    return;
  }
  if (wave >= 4) return;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    b[it & 7] = (_Float16)(seed[it & 1023] + (float)(lane & 15) * 0.03125f);
    a[(it + 3) & 7] = (_Float16)(seed[(it * 7) & 1023] - (float)(lane >> 4) * 0.0625f);
    float want[4], got[4];
    probe_overlap<WHICH, PRE>(a, b, got, want);
#pragma unroll
    for (int q = 0; q < 4; ++q) bad += (got[q] != want[q]) ? 1u : 0u;
  }
  if (bad) atomicAdd(&bad_by_group[(lane >> 4)], bad);
}
template <int WHICH, int PRE>
static void run_overlap(const float* d_seed, unsigned* d_bad, int busy, int blocks) {
  unsigned h[8] = {0};
  CHECK(hipMemset(d_bad, 0, sizeof(h)));
  hipLaunchKernelGGL((kern_overlap<WHICH, PRE>), dim3(blocks), dim3(512), 0, 0, d_seed, 200, busy, d_bad);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost));
  printf("  vdst IS %s, %d MFMAs in front, partner %s: wrong results by lane group (rows 4g..4g+3) g=0..3: %u %u %u %u\n",
         WHICH == 0 ? "srcA" : "srcB", PRE, busy ? "busy" : "idle", h[0], h[1], h[2], h[3]);
}

template <int NOPS, int CHAIN>
__global__ void __launch_bounds__(512) kern(const float* seed, int iters, int busy_partner, unsigned* bad_by_group) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(1.0f + 0.125f * ((lane + e) & 7)); b[e] = (_Float16)(0.5f + 0.0625f * ((lane * 3 + e) & 15)); }
  unsigned bad = 0;
  if (wave >= 4 && busy_partner) {  // partner waves (same SIMDs as waves 0..3): a dense stream of independent MFMAs
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters * 6; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[q], 0, 0, 0);
    }
    if (acc[0].x + acc[1].y + acc[2].z + acc[3].w == 12345.f) bad_by_group[7] = 1;
    return;
  }
  if (wave >= 4) return;
  for (int it = 0; it < iters; ++it) {
    // a fresh operand every iteration: a stale read returns the previous iteration's (different) result or zero
    b[it & 7] = (_Float16)(seed[it & 1023] + (float)(lane & 15) * 0.03125f);
    float want[4], got[4];
    probe<64, CHAIN>(a, b, want);
    probe<NOPS, CHAIN>(a, b, got);
#pragma unroll
    for (int q = 0; q < 4; ++q) bad += (got[q] != want[q]) ? (1u << (8 * q)) : 0u;
  }
  // per lane group g = lane >> 4 (the rows 4g..4g+3 of the tile = the instruction's pass g), per register
  for (int q = 0; q < 4; ++q) {
    const unsigned n = (bad >> (8 * q)) & 0xff;
    if (n) atomicAdd(&bad_by_group[(lane >> 4)], n);
  }
}

template <int NOPS, int CHAIN>
static void run(const float* d_seed, unsigned* d_bad, int busy, int blocks) {
  unsigned h[8] = {0};
  CHECK(hipMemset(d_bad, 0, sizeof(h)));
  hipLaunchKernelGGL((kern<NOPS, CHAIN>), dim3(blocks), dim3(512), 0, 0, d_seed, 200, busy, d_bad);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost));
  printf("  chain %d, %2d wait states, partner %s: stale reads by lane group (pass) g=0..3: %u %u %u %u\n", CHAIN, NOPS,
         busy ? "busy" : "idle", h[0], h[1], h[2], h[3]);
}

int main() {
  std::vector<float> seed(1024);
  srand(3);
  for (auto& v : seed) v = 0.25f + (float)(rand() & 1023) / 1024.0f;
  float* d_seed;
  unsigned* d_bad;
  CHECK(hipMalloc(&d_seed, seed.size() * 4));
  CHECK(hipMalloc(&d_bad, 64));
  CHECK(hipMemcpy(d_seed, seed.data(), seed.size() * 4, hipMemcpyHostToDevice));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount;
  printf("v_mfma_f32_16x16x32_f16 -> v_mov of its result, %d blocks of 8 waves (two per SIMD), 200 probes per lane\n", blocks);
  for (int busy = 0; busy < 2; ++busy) {
    run<0, 1>(d_seed, d_bad, busy, blocks);
    run<2, 1>(d_seed, d_bad, busy, blocks);
    run<4, 1>(d_seed, d_bad, busy, blocks);
    run<5, 1>(d_seed, d_bad, busy, blocks);
    run<6, 1>(d_seed, d_bad, busy, blocks);
    run<7, 1>(d_seed, d_bad, busy, blocks);
    run<8, 1>(d_seed, d_bad, busy, blocks);
    run<9, 1>(d_seed, d_bad, busy, blocks);
    run<10, 1>(d_seed, d_bad, busy, blocks);
    run<12, 1>(d_seed, d_bad, busy, blocks);
    run<16, 1>(d_seed, d_bad, busy, blocks);
    run<6, 3>(d_seed, d_bad, busy, blocks);
    run<8, 3>(d_seed, d_bad, busy, blocks);
    run<10, 3>(d_seed, d_bad, busy, blocks);
    run<12, 3>(d_seed, d_bad, busy, blocks);
    run<16, 3>(d_seed, d_bad, busy, blocks);
    run<24, 3>(d_seed, d_bad, busy, blocks);
    run_burst<6, 4>(d_seed, d_bad, busy, blocks);
    run_burst<7, 4>(d_seed, d_bad, busy, blocks);
    run_burst<8, 4>(d_seed, d_bad, busy, blocks);
    run_burst<9, 4>(d_seed, d_bad, busy, blocks);
    run_burst<10, 4>(d_seed, d_bad, busy, blocks);
    run_burst<12, 4>(d_seed, d_bad, busy, blocks);
    run_burst<16, 4>(d_seed, d_bad, busy, blocks);
    run_burst<8, 9>(d_seed, d_bad, busy, blocks);
    run_burst<12, 9>(d_seed, d_bad, busy, blocks);
    run_burst<16, 9>(d_seed, d_bad, busy, blocks);
    run_burst<24, 9>(d_seed, d_bad, busy, blocks);
    run_war<0, 0>(d_seed, d_bad, busy, blocks);
    run_war<1, 0>(d_seed, d_bad, busy, blocks);
    run_war<2, 0>(d_seed, d_bad, busy, blocks);
    run_war<4, 0>(d_seed, d_bad, busy, blocks);
    run_war<0, 6>(d_seed, d_bad, busy, blocks);
    run_war<1, 6>(d_seed, d_bad, busy, blocks);
    run_war<2, 6>(d_seed, d_bad, busy, blocks);
    run_war<4, 6>(d_seed, d_bad, busy, blocks);
    run_war<8, 6>(d_seed, d_bad, busy, blocks);
    run_overlap<0, 0>(d_seed, d_bad, busy, blocks);
    run_overlap<1, 0>(d_seed, d_bad, busy, blocks);
    run_overlap<0, 3>(d_seed, d_bad, busy, blocks);
    run_overlap<1, 3>(d_seed, d_bad, busy, blocks);
    run_srcc<0>(d_seed, d_bad, busy, blocks);
    run_srcc<1>(d_seed, d_bad, busy, blocks);
    run_srcc<2>(d_seed, d_bad, busy, blocks);
    run_srcc<3>(d_seed, d_bad, busy, blocks);
    run_srcc<4>(d_seed, d_bad, busy, blocks);
    run_srcc<5>(d_seed, d_bad, busy, blocks);
    run_srcc<6>(d_seed, d_bad, busy, blocks);
    run_srcc<8>(d_seed, d_bad, busy, blocks);
  }
  return 0;
}
