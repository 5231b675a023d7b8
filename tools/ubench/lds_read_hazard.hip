// Two questions about LDS reads on gfx950 that came up while chasing a run-to-run difference in egnn_div_walker_kernel.hip
// (one dword of a ds_read_b128 result read as zero in lanes 48..63, now and then):
//   T1  is it safe to overwrite the ADDRESS register of a ds_read_b128 in the very next instruction?
//   T2  does `s_waitcnt lgkmcnt(1)` behind seven reads guarantee the data of the sixth, used by the next instruction?
// Inline asm with physical registers: the instruction sequence is exactly what the source says.  Eight waves per block.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lds_read_hazard tools/ubench/lds_read_hazard.hip && tools/ubench/lds_read_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int TEST, int PARTNER>
__global__ void __launch_bounds__(512) kern(unsigned* bad, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // table: 16 columns x 176 floats (the record layout of the kernel), float f at (col, k) = 1000 col + k + 1
  for (int i = threadIdx.x; i < 16 * 176; i += 512) lds[i] = 1000.0f * (i / 176) + (i % 176) + 1.0f;
  __syncthreads();
  if (wave >= 4) {
    if (PARTNER == 1) {  // partner: LDS read traffic + vector work
      float s = 0.f;
      for (int it = 0; it < iters * 8; ++it) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((lane + it) & 15) * 176 + 4 * ((it >> 4) & 31));
        s += v.x + v.y + v.z + v.w;
      }
      if (s == 12345.f) bad[15] = 1;
    } else if (PARTNER == 2) {  // partner: matrix instructions
      f16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(1.0f + lane); b[e] = (_Float16)(0.5f + e); }
      f32x4 acc = {0, 0, 0, 0};
      for (int it = 0; it < iters * 8; ++it) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
      if (acc.x == 12345.f) bad[15] = 1;
    }
    return;
  }
  const int col = lane & 15;
  unsigned nbad[4] = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const unsigned addr = (unsigned)((col * 176 + 4 * (it & 31)) * 4);
    float r0, r1, r2, r3;
    const float w0 = 1000.0f * col + 4 * (it & 31) + 1.0f;
    if (TEST == 1) {
      asm volatile(
          "v_mov_b32 v120, %4\n\t"
          "s_nop 3\n\t"
          "ds_read_b128 v[100:103], v120\n\t"
          "v_mov_b32 v120, 0x3f800000\n\t"   // the address register is overwritten at once
          "v_mov_b32 v121, 0x3f800000\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103\n\ts_nop 3"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "v100", "v101", "v102", "v103", "v120", "v121", "memory");
    } else {
      asm volatile(
          "v_mov_b32 v120, %4\n\t"
          "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
          "s_nop 3\n\t"
          "ds_read_b128 v[104:107], v120 offset:256\n\t"
          "ds_read_b128 v[108:111], v120 offset:272\n\t"
          "ds_read_b128 v[112:115], v120 offset:384\n\t"
          "ds_read_b128 v[116:119], v120 offset:400\n\t"
          "ds_read_b32 v124, v120 offset:652\n\t"
          "ds_read_b128 v[100:103], v120\n\t"            // the sixth read
          "ds_read_b128 v[126:129], v120 offset:16\n\t"  // the seventh
          "v_mov_b32 v120, 0x3f800000\n\t"
          "s_waitcnt lgkmcnt(1)\n\t"
          "v_pk_mul_f32 v[130:131], v[100:101], v[100:101]\n\t"  // consumer straight behind the wait
          "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103\n\t"
          "s_waitcnt lgkmcnt(0)\n\ts_nop 3"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr)
          : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114",
            "v115", "v116", "v117", "v118", "v119", "v120", "v124", "v126", "v127", "v128", "v129", "v130", "v131", "memory");
    }
    nbad[0] += (r0 != w0); nbad[1] += (r1 != w0 + 1); nbad[2] += (r2 != w0 + 2); nbad[3] += (r3 != w0 + 3);
  }
  for (int q = 0; q < 4; ++q)
    if (nbad[q]) atomicAdd(&bad[(lane >> 4) * 4 + q], nbad[q]);
}

template <int TEST, int PARTNER>
static void run(unsigned* d_bad, int blocks) {
  unsigned h[16] = {0};
  CHECK(hipMemset(d_bad, 0, sizeof(h)));
  hipLaunchKernelGGL((kern<TEST, PARTNER>), dim3(blocks), dim3(512), 16 * 176 * 4, 0, d_bad, 4000);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost));
  printf("  T%d, partner %s: wrong dwords [lane group g][dword]:", TEST, PARTNER == 0 ? "idle" : (PARTNER == 1 ? "reading LDS" : "issuing MFMAs"));
  for (int g = 0; g < 4; ++g) printf("  g%d: %u %u %u %u", g, h[4 * g], h[4 * g + 1], h[4 * g + 2], h[4 * g + 3]);
  printf("\n");
}

int main() {
  unsigned* d_bad;
  CHECK(hipMalloc(&d_bad, 64));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount;
  printf("T1: address register overwritten right behind ds_read_b128;  T2: lgkmcnt(1) behind seven reads, sixth consumed at once\n");
  for (int rep = 0; rep < 2; ++rep) {
    run<1, 0>(d_bad, blocks); run<1, 1>(d_bad, blocks); run<1, 2>(d_bad, blocks);
    run<2, 0>(d_bad, blocks); run<2, 1>(d_bad, blocks); run<2, 2>(d_bad, blocks);
  }
  return 0;
}
