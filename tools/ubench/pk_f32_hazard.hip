// Stand-alone reproducer attempt (round 6) for the run-to-run differences of round 5's walker-resident trace kernel
// (profiles/r05_walker_packed_fp32_hazard.txt): the LOW half of a packed fp32 result wrong in lanes 48..63 with intact inputs,
// in the hipcc-generated sequence
//     v_pk_mul_f32 v[0:1], v[170:171], v[206:207] op_sel:[0,1]             ; (x8[0] m', x8[1] m')
//     v_pk_fma_f32 v[0:1], v[52:53], s[96:97], v[0:1] op_sel_hi:[1,0,1]    ; (a zr, a ze) / 16 + ...
//     v_pk_mul_f32 v[0:1], v[0:1], s[98:99] op_sel_hi:[1,0]                ; * 64
// whose inputs were an LDS read (x8: ds_read_b128 -> v[170:173]), two fresh matrix-instruction results (zr, ze) and an
// SGPR pair.  The sequence is rebuilt here in inline asm with the SAME physical registers and operand modifiers, fed the same
// way (LDS read + two v_mfma_f32_16x16x32_f16 results), with the partner waves of the SIMD idle / issuing
// v_mfma_f32_16x16x32_f16 / issuing v_mfma_f32_32x32x16_f16 / issuing packed fp32 + transcendentals, and compared bit for
// bit with the same arithmetic done by unpacked instructions behind long waits.  Variants:
//   A<n>  n wait states between the matrix instructions and the vector reads of their results (hipcc pads 8)
//   W<n>  v[0:3] is ALSO the destination of a third matrix instruction issued just before the sequence (write-after-write
//         from the matrix pipe's late write-back; hipcc keeps 8+ wait states there), n wait states
//   L     the LDS read is waited for with s_waitcnt lgkmcnt(0) and consumed at once by the packed multiply
// Mismatches are counted per lane group g = lane / 16 (the fault was in g = 3 only).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pk_f32_hazard tools/ubench/pk_f32_hazard.hip && tools/ubench/pk_f32_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CLOB "v0", "v1", "v2", "v3", "v52", "v53", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", \
             "v170", "v171", "v172", "v173", "v206", "v207", "s96", "s97", "s98", "s99"

// the reference: same arithmetic, unpacked instructions, long waits everywhere
__device__ __forceinline__ void probe_ref(unsigned lds_addr, f16x8 a, f16x8 b, f16x8 b2, float m, float aa, float (&got)[2]) {
  asm volatile(
      "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
      "v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\ts_nop 7\n\t"
      "ds_read_b128 v[170:173], %2\n\t"
      "v_mfma_f32_16x16x32_f16 v[100:103], %3, %4, v[100:103]\n\t"
      "v_mfma_f32_16x16x32_f16 v[104:107], %3, %5, v[104:107]\n\t"
      "s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 7\n\t"
      "v_mul_f32 v52, %7, v100\n\tv_mul_f32 v53, %7, v104\n\ts_nop 7\n\t"
      "v_mul_f32 v0, v170, %6\n\tv_mul_f32 v1, v171, %6\n\ts_nop 7\n\t"
      "s_mov_b32 s96, 0x3d800000\n\ts_mov_b32 s98, 0x42800000\n\ts_nop 3\n\t"
      "v_fma_f32 v0, v52, s96, v0\n\tv_fma_f32 v1, v53, s96, v1\n\ts_nop 7\n\t"
      "v_mul_f32 v0, s98, v0\n\tv_mul_f32 v1, s98, v1\n\ts_nop 7\n\t"
      "v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1\n\ts_nop 7"
      : "=&v"(got[0]), "=&v"(got[1])
      : "v"(lds_addr), "v"(a), "v"(b), "v"(b2), "v"(m), "v"(aa)
      : CLOB, "memory");
}

// KIND 0: A<NOPS>; 1: W<NOPS>; 2: L
template <int KIND, int NOPS>
__device__ __forceinline__ void probe(unsigned lds_addr, f16x8 a, f16x8 b, f16x8 b2, float m, float aa, float (&got)[2]) {
  asm volatile(
      "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
      "v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\t"
      "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\t"
      "s_mov_b32 s96, 0x3d800000\n\ts_mov_b32 s97, 0x7fc00000\n\ts_mov_b32 s98, 0x42800000\n\ts_mov_b32 s99, 0x7fc00000\n\t"
      "v_mov_b32 v206, 0x7fc00000\n\tv_mov_b32 v207, %6\n\ts_nop 7\n\t"
      ".if %8 == 2\n\t"
      "v_mfma_f32_16x16x32_f16 v[100:103], %3, %4, v[100:103]\n\t"
      "v_mfma_f32_16x16x32_f16 v[104:107], %3, %5, v[104:107]\n\t"
      "s_nop 15\n\t"
      "v_mul_f32 v52, %7, v100\n\tv_mul_f32 v53, %7, v104\n\t"
      "ds_read_b128 v[170:173], %2\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      ".else\n\t"
      "ds_read_b128 v[170:173], %2\n\t"
      "v_mfma_f32_16x16x32_f16 v[100:103], %3, %4, v[100:103]\n\t"
      "v_mfma_f32_16x16x32_f16 v[104:107], %3, %5, v[104:107]\n\t"
      ".if %8 == 1\n\t"
      "s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\t"
      "v_mul_f32 v52, %7, v100\n\tv_mul_f32 v53, %7, v104\n\t"
      "v_mfma_f32_16x16x32_f16 v[0:3], %3, %4, v[100:103]\n\t"     // late write-back into v[0:3]
      ".if %9 > 0\n\ts_nop %9 - 1\n\t.endif\n\t"
      ".else\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      ".if %9 > 0\n\ts_nop %9 - 1\n\t.endif\n\t"
      "v_mul_f32 v52, %7, v100\n\tv_mul_f32 v53, %7, v104\n\t"
      ".endif\n\t"
      ".endif\n\t"
      "v_pk_mul_f32 v[0:1], v[170:171], v[206:207] op_sel:[0,1]\n\t"
      "v_pk_fma_f32 v[0:1], v[52:53], s[96:97], v[0:1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_mul_f32 v[0:1], v[0:1], s[98:99] op_sel_hi:[1,0]\n\t"
      "s_nop 15\n\ts_nop 15\n\t"
      "v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1\n\ts_nop 7"
      : "=&v"(got[0]), "=&v"(got[1])
      : "v"(lds_addr), "v"(a), "v"(b), "v"(b2), "v"(m), "v"(aa), "i"(KIND), "i"(NOPS)
      : CLOB, "memory");
}

// PARTNER 0: idle; 1: v_mfma_f32_16x16x32_f16; 2: v_mfma_f32_32x32x16_f16; 3: packed fp32 + transcendentals
template <int KIND, int NOPS, int PARTNER>
__global__ void __launch_bounds__(512) kern(const float* seed, int iters, unsigned* bad_by_group) {
  __shared__ float tab[8][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x8 a, b, b2;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(1.0f + 0.125f * ((lane + e) & 7));
    b[e] = (_Float16)(0.5f + 0.0625f * ((lane * 3 + e) & 15));
    b2[e] = (_Float16)(0.25f + 0.03125f * ((lane * 5 + e) & 15));
  }
  if (wave >= 4) {
    if (PARTNER == 1) {
      f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[q], 0, 0, 0);
      }
      if (acc[0].x + acc[1].y + acc[2].z + acc[3].w == 12345.f) bad_by_group[7] = 1;
    } else if (PARTNER == 2) {
      f32x16 acc[2];
      for (int q = 0; q < 2; ++q)
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
      for (int it = 0; it < iters * 6; ++it) {
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
      }
      if (acc[0][0] + acc[1][5] == 12345.f) bad_by_group[7] = 1;
    } else if (PARTNER == 3) {
      float u = seed[lane], v = seed[lane + 64], s = 0.f;
      for (int it = 0; it < iters * 40; ++it) {
        u = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u)) + v * 0.5f;
        v = fmaf(u, 0.25f, v * 0.75f);
        s += u * v;
      }
      if (s == 12345.f) bad_by_group[7] = 1;
    }
    return;
  }
  unsigned bad_lo = 0, bad_hi = 0;
  const unsigned addr = (unsigned)(size_t)(&tab[wave][lane][0]);
  for (int it = 0; it < iters; ++it) {
    const float s0 = seed[(it * 7 + lane) & 1023], s1 = seed[(it * 13 + 3 * lane + 1) & 1023];
    tab[wave][lane][0] = s0 - 0.5f; tab[wave][lane][1] = s1 + 0.25f; tab[wave][lane][2] = s0; tab[wave][lane][3] = s1;
    b[it & 7] = (_Float16)(s0 + (float)(lane & 15) * 0.03125f);
    b2[(it + 3) & 7] = (_Float16)(s1 - (float)(lane & 7) * 0.0625f);
    const float m = 0.551548f + 0.001f * (float)(it & 15), aa = 0.531705f;
    float want[2], got[2];
    __builtin_amdgcn_s_waitcnt(0);
    probe_ref(addr, a, b, b2, m, aa, want);
    probe<KIND, NOPS>(addr, a, b, b2, m, aa, got);
    bad_lo += (__float_as_uint(got[0]) != __float_as_uint(want[0])) ? 1u : 0u;
    bad_hi += (__float_as_uint(got[1]) != __float_as_uint(want[1])) ? 1u : 0u;
  }
  if (bad_lo) atomicAdd(&bad_by_group[lane >> 4], bad_lo);
  if (bad_hi) atomicAdd(&bad_by_group[8 + (lane >> 4)], bad_hi);
}

static const char* PN[4] = {"idle", "v_mfma 16x16x32", "v_mfma 32x32x16", "pk fp32 + exp/rcp"};
static unsigned long long total_probes = 0, total_bad = 0;

template <int KIND, int NOPS, int PARTNER>
static void run(const float* d_seed, unsigned* d_bad, int blocks, int iters) {
  unsigned h[16] = {0};
  CHECK(hipMemset(d_bad, 0, sizeof(h)));
  hipLaunchKernelGGL((kern<KIND, NOPS, PARTNER>), dim3(blocks), dim3(512), 0, 0, d_seed, iters, d_bad);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost));
  const char* kn = KIND == 0 ? "A (MFMA results read after n wait states)" : KIND == 1 ? "W (v[0:3] also an MFMA destination, n wait states)"
                                                                                        : "L (LDS read consumed at once)";
  const unsigned long long probes = (unsigned long long)blocks * 256 * iters;
  unsigned nb = 0;
  for (int g = 0; g < 4; ++g) nb += h[g] + h[8 + g];
  total_probes += probes; total_bad += nb;
  printf("  %-52s n=%2d partner %-18s %9llu lane-probes: low half wrong by lane group %u %u %u %u | high half %u %u %u %u\n", kn, NOPS,
         PN[PARTNER], probes, h[0], h[1], h[2], h[3], h[8], h[9], h[10], h[11]);
}

template <int KIND, int NOPS>
static void all_partners(const float* d_seed, unsigned* d_bad, int blocks, int iters) {
  run<KIND, NOPS, 0>(d_seed, d_bad, blocks, iters);
  run<KIND, NOPS, 1>(d_seed, d_bad, blocks, iters);
  run<KIND, NOPS, 2>(d_seed, d_bad, blocks, iters);
  run<KIND, NOPS, 3>(d_seed, d_bad, blocks, iters);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  std::vector<float> seed(1024);
  unsigned s = 12345u;
  for (auto& v : seed) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 16777216.0f; }
  float* d_seed; unsigned* d_bad;
  CHECK(hipMalloc(&d_seed, 4096)); CHECK(hipMalloc(&d_bad, 64));
  CHECK(hipMemcpy(d_seed, seed.data(), 4096, hipMemcpyHostToDevice));
  printf("packed-fp32 sequence of profiles/r05_walker_packed_fp32_hazard.txt, physical registers as in the failing kernel;\n"
         "8 waves per workgroup (waves 0-3 probe, 4-7 = partner on the same SIMDs), %d iterations per lane\n", iters);
  for (int blocks : {8, 256, 2048}) {
    printf("grid of %d workgroups\n", blocks);
    all_partners<0, 8>(d_seed, d_bad, blocks, iters);   // hipcc's padding
    all_partners<0, 7>(d_seed, d_bad, blocks, iters);
    all_partners<1, 11>(d_seed, d_bad, blocks, iters);
    all_partners<1, 8>(d_seed, d_bad, blocks, iters);
    all_partners<2, 0>(d_seed, d_bad, blocks, iters);
  }
  printf("below the documented wait states (expected to fail: shows the probe can see a hazard)\n");
  all_partners<0, 2>(d_seed, d_bad, 256, iters);
  all_partners<1, 0>(d_seed, d_bad, 256, iters);
  printf("total (all lines): %llu lane-probes, %llu wrong\n", total_probes, total_bad);
  return 0;
}
