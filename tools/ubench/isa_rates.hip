// Micro-benchmark of VALU / MFMA issue rates on gfx950 (development aid; results quoted in DESIGN.md).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/isa_rates tools/ubench/isa_rates.hip && tools/ubench/isa_rates
// Each kernel runs REPS x 32 copies of one instruction on independent registers in every wave of ONE workgroup per CU
// slot and reports shader cycles (s_memtime) per wave-instruction for 1 and 2 waves per SIMD.  Also probes how the f16
// MFMA treats denormal inputs (needed by the f16 two-piece operand split).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int REPS = 256;

#define BODY8(INSTR)                                                            \
  asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
               : "v"(a), "v"(b));

#define KERNEL(NAME, INSTR)                                                     \
  __global__ void __launch_bounds__(512) NAME(float* out, long long* cyc) {     \
    float r[8];                                                                 \
    for (int i = 0; i < 8; ++i) r[i] = 1.0f + threadIdx.x * 1e-3f + i;          \
    float a = 1.0001f, b = 0.5f;                                                \
    __syncthreads();                                                            \
    long long t0 = __builtin_amdgcn_s_memtime();                                \
    for (int it = 0; it < REPS; ++it) { BODY8(INSTR) BODY8(INSTR) BODY8(INSTR) BODY8(INSTR) } \
    long long t1 = __builtin_amdgcn_s_memtime();                                \
    float s = 0;                                                                \
    for (int i = 0; i < 8; ++i) s += r[i];                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                             \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
  }

#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_ADD(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define I_MUL(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define I_AND(k) "v_and_b32 %" #k ", 0xffff0000, %" #k "\n"
#define I_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define I_EXP(k) "v_exp_f32 %" #k ", %" #k "\n"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define I_CVTPK(k) "v_cvt_pk_f16_f32 %" #k ", %" #k ", %8\n"
#define I_CVTPKBF(k) "v_cvt_pk_bf16_f32 %" #k ", %" #k ", %8\n"
#define I_CVTRTZ(k) "v_cvt_pkrtz_f16_f32 %" #k ", %" #k ", %8\n"
#define I_FMAMIX(k) "v_fma_mix_f32 %" #k ", %" #k ", -1.0, %8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"
#define I_FMAMIXHI(k) "v_fma_mix_f32 %" #k ", %" #k ", -1.0, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define I_SHFL(k) "ds_swizzle_b32 %" #k ", %" #k " offset:swizzle(SWAP,16)\n"
#define I_DPP(k) "v_add_f32_dpp %" #k ", %" #k ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"

KERNEL(k_fma, I_FMA)
KERNEL(k_add, I_ADD)
KERNEL(k_mul, I_MUL)
KERNEL(k_and, I_AND)
KERNEL(k_perm, I_PERM)
KERNEL(k_exp, I_EXP)
KERNEL(k_rcp, I_RCP)
KERNEL(k_cvtpk, I_CVTPK)
KERNEL(k_cvtpkbf, I_CVTPKBF)
KERNEL(k_cvtrtz, I_CVTRTZ)
KERNEL(k_fmamix, I_FMAMIX)
KERNEL(k_fmamixhi, I_FMAMIXHI)
KERNEL(k_mov, I_MOV)
KERNEL(k_dpp, I_DPP)

// three DISTINCT vector-register operands per instruction (the kernels' usual case; the rows above share two of them)
__global__ void __launch_bounds__(1024) k_fma3(float* out, long long* cyc) {
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = 1.0f + threadIdx.x * 1e-3f + i * 1e-4f;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < REPS * 4; ++it) {
    asm volatile(
        "v_fma_f32 %0, %1, %2, %0\nv_fma_f32 %1, %2, %3, %1\nv_fma_f32 %2, %3, %4, %2\nv_fma_f32 %3, %4, %5, %3\n"
        "v_fma_f32 %4, %5, %6, %4\nv_fma_f32 %5, %6, %7, %5\nv_fma_f32 %6, %7, %0, %6\nv_fma_f32 %7, %0, %1, %7\n"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
// the same with VOP2 v_fmac (dst = accumulator) and v_mul / v_sub on distinct registers
__global__ void __launch_bounds__(1024) k_vop2mix(float* out, long long* cyc) {
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = 1.0f + threadIdx.x * 1e-3f + i * 1e-4f;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < REPS * 4; ++it) {
    asm volatile(
        "v_fmac_f32 %0, %1, %2\nv_sub_f32 %3, %4, %5\nv_mul_f32 %6, %7, %1\nv_fmac_f32 %2, %3, %4\n"
        "v_sub_f32 %5, %6, %7\nv_mul_f32 %1, %0, %2\nv_fmac_f32 %4, %5, %6\nv_sub_f32 %7, %1, %3\n"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// straight-line code: the same v_fma_f32 stream with a loop body of 2 048 instructions (16 KB of code) instead of 32 --
// does the instruction fetch keep up when nothing is re-used from the instruction buffers?
#define B32(I) BODY8(I) BODY8(I) BODY8(I) BODY8(I)
#define B256(I) B32(I) B32(I) B32(I) B32(I) B32(I) B32(I) B32(I) B32(I)
#define B2048(I) B256(I) B256(I) B256(I) B256(I) B256(I) B256(I) B256(I) B256(I)
__global__ void __launch_bounds__(1024) k_fma_long(float* out, long long* cyc) {
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = 1.0f + threadIdx.x * 1e-3f + i;
  float a = 1.0001f, b = 0.5f;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < REPS / 64; ++it) { B2048(I_FMA) }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// operand banks: three source registers per FMA from the SAME register bank (index mod 4 equal) or from three different
// banks; explicit registers (v16 .. v63 are initialised and clobbered)
#define FMA_SAME(d, a) "v_fma_f32 v" #d ", v" #a ", v" #a "+16, v" #a "+32\n"
template <int SAME>
__global__ void __launch_bounds__(1024) k_bank(float* out, long long* cyc) {
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < REPS * 4; ++it) {
    if (SAME)
      asm volatile(
          "v_fma_f32 v0, v16, v32, v48\nv_fma_f32 v1, v20, v36, v52\nv_fma_f32 v2, v24, v40, v56\nv_fma_f32 v3, v28, v44, v60\n"
          "v_fma_f32 v4, v17, v33, v49\nv_fma_f32 v5, v21, v37, v53\nv_fma_f32 v6, v25, v41, v57\nv_fma_f32 v7, v29, v45, v61\n"
          ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7");
    else
      asm volatile(
          "v_fma_f32 v0, v16, v33, v50\nv_fma_f32 v1, v20, v37, v54\nv_fma_f32 v2, v24, v41, v58\nv_fma_f32 v3, v28, v45, v62\n"
          "v_fma_f32 v4, v17, v34, v51\nv_fma_f32 v5, v21, v38, v55\nv_fma_f32 v6, v25, v42, v59\nv_fma_f32 v7, v29, v46, v63\n"
          ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7");
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s;
  asm volatile("v_add_f32 %0, v0, v7" : "=v"(s));
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// packed f32 on register pairs
__global__ void __launch_bounds__(512) k_pkfma(float* out, long long* cyc) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 r[8];
  for (int i = 0; i < 8; ++i) r[i] = f2{1.0f + threadIdx.x * 1e-3f + i, 2.0f};
  f2 a = {1.0001f, 0.9999f}, b = {0.5f, 0.25f};
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < REPS * 4; ++it) {
    asm volatile(
        "v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\n"
        "v_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
        : "v"(a), "v"(b));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// MFMA chains: DEP = one accumulator (dependent chain), else 2 independent accumulators
template <int KIND, bool DEP>
__global__ void __launch_bounds__(512) k_mfma(float* out, long long* cyc) {
  f32x16 acc0 = {0}, acc1 = {0};
  f16x8 ah, bh;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(0.01f * (threadIdx.x + i)); bh[i] = (_Float16)(0.02f * i); ab[i] = (short)(0x3c00 + i); bb[i] = (short)(0x3d00 + threadIdx.x); }
  float af = 1.0f + threadIdx.x, bf = 0.5f;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < REPS; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (KIND == 0) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
        if (DEP) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
      } else if (KIND == 1) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc0, 0, 0, 0);
        if (DEP) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc0, 0, 0, 0);
        if (DEP) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc1, 0, 0, 0);
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// MFMA + VALU co-issue: per MFMA, NV independent v_fma_f32 (same wave)
template <int KIND, int NV>
__global__ void __launch_bounds__(512) k_mix(float* out, long long* cyc) {
  f32x16 acc0 = {0};
  f16x8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(0.01f * (threadIdx.x + i)); bh[i] = (_Float16)(0.02f * i); }
  float af = 1.0f + threadIdx.x, bf = 0.5f;
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = 1.0f + i;
  float a = 1.0001f, b = 0.5f;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < REPS; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (KIND == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc0, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[v & 7]) : "v"(a), "v"(b));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i];
  for (int i = 0; i < 8; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// Two waves per SIMD running DIFFERENT instruction kinds: waves 0-3 one kind, waves 4-7 (the SIMD partners) the other.
// Tells whether the transcendental unit (or the half-rate conversions) issue beside plain VALU work of the partner wave.
#define KERNEL_PAIR(NAME, INSTR_A, INSTR_B)                                     \
  __global__ void __launch_bounds__(512) NAME(float* out, long long* cyc) {     \
    float r[8];                                                                 \
    for (int i = 0; i < 8; ++i) r[i] = 1.0f + threadIdx.x * 1e-3f + i;          \
    float a = 1.0001f, b = 0.5f;                                                \
    const bool first = (threadIdx.x >> 6) < 4;                                  \
    __syncthreads();                                                            \
    long long t0 = __builtin_amdgcn_s_memtime();                                \
    if (first) { for (int it = 0; it < REPS; ++it) { BODY8(INSTR_A) BODY8(INSTR_A) BODY8(INSTR_A) BODY8(INSTR_A) } } \
    else { for (int it = 0; it < REPS; ++it) { BODY8(INSTR_B) BODY8(INSTR_B) BODY8(INSTR_B) BODY8(INSTR_B) } } \
    long long t1 = __builtin_amdgcn_s_memtime();                                \
    float s = 0;                                                                \
    for (int i = 0; i < 8; ++i) s += r[i];                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                             \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
  }
KERNEL_PAIR(k_pair_exp_fma, I_EXP, I_FMA)
KERNEL_PAIR(k_pair_exp_exp, I_EXP, I_EXP)
KERNEL_PAIR(k_pair_fma_fma, I_FMA, I_FMA)
KERNEL_PAIR(k_pair_cvt_fma, I_CVTPK, I_FMA)
KERNEL_PAIR(k_pair_exp_cvt, I_EXP, I_CVTPK)

template <typename K>
static void run_pair(const char* name, K kern) {
  float* out; long long* cyc;
  CHECK(hipMalloc(&out, 512 * 4 * 256)); CHECK(hipMalloc(&cyc, 8 * 8 * 256));
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(1), dim3(512), 0, 0, out, cyc); CHECK(hipDeviceSynchronize()); }
  long long h[8];
  CHECK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
  double a = 0, b = 0;
  for (int w = 0; w < 4; ++w) { a = h[w] > a ? h[w] : a; b = h[w + 4] > b ? h[w + 4] : b; }
  printf("%-44s A-waves %7.2f, B-waves %7.2f cyc per wave-instruction (two waves per SIMD, one of each kind)\n", name,
         a / (32.0 * REPS), b / (32.0 * REPS));
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

// f16 MFMA denormal probe: A = 1 (k=0), B = tiny -> C should be tiny if denormals are honoured
__global__ void k_denorm(float* out) {
  const float vals[8] = {6.2e-5f /*just above min normal 6.1e-5*/, 3.0e-5f, 1.0e-6f, 6.0e-8f /*smallest denormal 5.96e-8*/, 2.0e-8f, 0.f, 0.f, 0.f};
  for (int t = 0; t < 5; ++t) {
    f16x8 a = {0}, b = {0};
    f32x16 acc = {0};
    if (threadIdx.x < 32) { a[0] = (_Float16)1.0f; b[0] = (_Float16)vals[t]; }  // k = 0 lives in lanes 0..31, element 0
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[2 * t] = (float)(_Float16)vals[t]; out[2 * t + 1] = acc[0]; }
  }
  // product of a denormal-range result: 2^-10 * 2^-10 accumulates exactly in f32
  {
    f16x8 a = {0}, b = {0};
    f32x16 acc = {0};
    if (threadIdx.x < 32) { a[0] = (_Float16)9.765625e-4f; b[0] = (_Float16)9.765625e-4f; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[10] = 9.5367431640625e-07f; out[11] = acc[0]; }
  }
}

// the same measurement with the WHOLE chip busy (one workgroup per CU, or several): does a wave-instruction cost the same
// when all 256 CUs issue at once (power management) as when one CU runs alone?
template <typename K>
static void run_chip(const char* name, K kern, int instr_per_iter, int iters, int blocks, int waves) {
  float* out; long long* cyc;
  CHECK(hipMalloc(&out, (size_t)blocks * 1024 * 4)); CHECK(hipMalloc(&cyc, (size_t)blocks * 16 * 8));
  for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), 0, 0, out, cyc); CHECK(hipDeviceSynchronize()); }
  std::vector<long long> h((size_t)blocks * waves);
  CHECK(hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost));
  double mx = 0, sum = 0;
  for (auto v : h) { mx = v > mx ? v : mx; sum += (double)v; }
  const double per = (sum / h.size()) / ((double)instr_per_iter * iters);
  printf("%-44s %4d blocks x %2d waves: mean %6.2f (max %6.2f) cyc per wave-instruction, %5.2f cyc/SIMD per instruction\n", name,
         blocks, waves, per, mx / ((double)instr_per_iter * iters), per / (waves / 4.0));
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

template <typename K>
static void run(const char* name, K kern, int instr_per_iter, int iters, bool four = false) {
  float* out; long long* cyc;
  CHECK(hipMalloc(&out, 1024 * 4 * 256)); CHECK(hipMalloc(&cyc, 8 * 16 * 256));
  for (int waves : {4, 8, 16}) {  // per workgroup = per CU: 1, 2 or (kernels built for 1024 threads) 4 waves per SIMD
    if (waves == 16 && !four) continue;
    hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 0, 0, out, cyc);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 0, 0, out, cyc);
    CHECK(hipDeviceSynchronize());
    std::vector<long long> h(waves);
    CHECK(hipMemcpy(h.data(), cyc, 8 * waves, hipMemcpyDeviceToHost));
    double mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    const double per = mx / ((double)instr_per_iter * iters);
    printf("%-44s %d waves/SIMD: %7.2f cyc per wave-instruction, %7.2f cyc/SIMD per instruction\n", name, waves / 4, per,
           per / (waves / 4));
  }
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
  run("v_fma_f32", k_fma, 32, REPS);
  run("v_add_f32", k_add, 32, REPS);
  run("v_mul_f32", k_mul, 32, REPS);
  run("v_and_b32 (literal)", k_and, 32, REPS);
  run("v_perm_b32", k_perm, 32, REPS);
  run("v_mov_b32", k_mov, 32, REPS);
  run("v_add_f32 dpp row_shr", k_dpp, 32, REPS);
  run("v_exp_f32", k_exp, 32, REPS);
  run("v_rcp_f32", k_rcp, 32, REPS);
  run("v_cvt_pk_f16_f32", k_cvtpk, 32, REPS);
  run("v_cvt_pk_bf16_f32", k_cvtpkbf, 32, REPS);
  run("v_cvt_pkrtz_f16_f32", k_cvtrtz, 32, REPS);
  run("v_fma_mix_f32 (f16 lo)", k_fmamix, 32, REPS);
  run("v_fma_mix_f32 (f16 hi)", k_fmamixhi, 32, REPS);
  run("v_pk_fma_f32", k_pkfma, 8, REPS * 4);
  run("v_fma_f32, 2 048-instruction loop body", k_fma_long, 2048, REPS / 64, true);
  run("v_fma_f32, sources in ONE register bank", k_bank<1>, 8, REPS * 4, true);
  run("v_fma_f32, sources in three banks", k_bank<0>, 8, REPS * 4, true);
  run("v_fma_f32, 3 distinct VGPR operands", k_fma3, 8, REPS * 4, true);
  for (int blocks : {1, 32, 256, 512})
    for (int waves : {8, 16}) run_chip("v_fma_f32 (3 operands), chip-wide", k_fma3, 8, REPS * 4, blocks, waves);
  for (int blocks : {1, 256}) run_chip("v_fmac/v_sub/v_mul mix, chip-wide", k_vop2mix, 8, REPS * 4, blocks, 16);
  run("v_fmac/v_sub/v_mul mix, distinct VGPRs", k_vop2mix, 8, REPS * 4, true);
  run_pair("pair: v_exp_f32 | v_fma_f32", k_pair_exp_fma);
  run_pair("pair: v_exp_f32 | v_exp_f32", k_pair_exp_exp);
  run_pair("pair: v_fma_f32 | v_fma_f32", k_pair_fma_fma);
  run_pair("pair: v_cvt_pk_f16_f32 | v_fma_f32", k_pair_cvt_fma);
  run_pair("pair: v_exp_f32 | v_cvt_pk_f16_f32", k_pair_exp_cvt);
  run("mfma_f32_32x32x16_f16 dependent chain", k_mfma<0, true>, 16, REPS);
  run("mfma_f32_32x32x16_f16 2 accumulators", k_mfma<0, false>, 16, REPS);
  run("mfma_f32_32x32x16_bf16 dependent chain", k_mfma<1, true>, 16, REPS);
  run("mfma_f32_32x32x2_f32 dependent chain", k_mfma<2, true>, 16, REPS);
  run("mfma f16 + 0 v_fma per MFMA (per MFMA)", k_mix<0, 0>, 8, REPS);
  run("mfma f16 + 4 v_fma per MFMA (per MFMA)", k_mix<0, 4>, 8, REPS);
  run("mfma f16 + 8 v_fma per MFMA (per MFMA)", k_mix<0, 8>, 8, REPS);
  run("mfma f16 + 16 v_fma per MFMA (per MFMA)", k_mix<0, 16>, 8, REPS);
  run("mfma f32x2 + 0 v_fma per MFMA (per MFMA)", k_mix<1, 0>, 8, REPS);
  run("mfma f32x2 + 8 v_fma per MFMA (per MFMA)", k_mix<1, 8>, 8, REPS);
  run("mfma f32x2 + 16 v_fma per MFMA (per MFMA)", k_mix<1, 16>, 8, REPS);
  float* d; CHECK(hipMalloc(&d, 64));
  hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, d);
  float h[12]; CHECK(hipMemcpy(h, d, 48, hipMemcpyDeviceToHost));
  for (int t = 0; t < 6; ++t) printf("f16 MFMA denormal probe: input %.9g -> 1*x through the MFMA = %.9g\n", h[2 * t], h[2 * t + 1]);
  return 0;
}
