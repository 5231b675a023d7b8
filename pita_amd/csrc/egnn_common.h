// Device helpers shared by the EGNN kernels (forward / fused sampler: egnn_kernel.hip; forward-mode
// derivative: egnn_jvp_kernel.hip) and the native handle behind pita_egnn_t.
#pragma once
#include <cstdlib>

#include "common.h"

namespace pita {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int EH = 32;       // hidden_nf
constexpr int PBS = 36;      // LDS row stride (floats) of the partner table
constexpr int MAT_F = 1024;  // floats per packed 32x32 matrix
// forward matrices (SiLU pre-scale folded in, see pita_egnn_create) followed by the TRANSPOSES of the unscaled
// matrices, which the reverse-mode kernel (egnn_vjp_kernel.hip) multiplies adjoints with
enum { M_WA = 0, M_WB, M_W2, M_WC1, M_WN1A, M_WN1B, M_WN2,
       M_WAT, M_WBT, M_W2T, M_WC1T, M_WN1AT, M_WN1BT, M_WN2T, M_COUNT };
enum { V_WRE = 0 /* 64 floats: w_r[out] | w_e[out], natural order */, V_B1 = 2, V_B2, V_WATT, V_BC1, V_WC2, V_BN1,
       V_BN2, V_WRF /* unscaled w_r, fragment order */, V_WEF /* unscaled w_e, fragment order */, V_COUNT };
constexpr int VEC_EMB_F = 96;                    // emb_w0, emb_w1, emb_b
constexpr int VEC_LAYER_F = V_COUNT * EH + 4;    // vectors (fragment order unless noted) + b_att (+pad)


__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in order; this only stops the compiler from moving
  // LDS accesses across the hand-off and drains outstanding reads.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void load_frag(const float* __restrict__ pack, int lane, float (&wf)[16]) {
  const f32x4* p = reinterpret_cast<const f32x4*>(pack) + lane;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = p[q * 64];
    wf[4 * q + 0] = v.x; wf[4 * q + 1] = v.y; wf[4 * q + 2] = v.z; wf[4 * q + 3] = v.w;
  }
}

__device__ __forceinline__ f32x16 lds_vec16(const float* v) {
  const f32x4* p = reinterpret_cast<const f32x4*>(v);
  f32x4 a = p[0], b = p[1], c = p[2], d = p[3];
  f32x16 r;
  r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
  r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
  r[8] = c.x; r[9] = c.y; r[10] = c.z; r[11] = c.w;
  r[12] = d.x; r[13] = d.y; r[14] = d.z; r[15] = d.w;
  return r;
}

// ---- 32x32 dense layer  out[o][col] = acc[o][col] + sum_k W[o][k] in[k][col]  in two arithmetic modes.
//
// PREC 0: 16 chained v_mfma_f32_32x32x2_f32 -- bit-exact fp32 (k-ordered fmaf chain).  On gfx950 the
//   f32-input MFMA runs at the fp32 VECTOR rate and (measured: SQ_VALU_MFMA_COEXEC_CYCLES = 0,
//   MFMA-busy + VALU-active = wave residency) does not overlap with VALU work on the same SIMD.
// PREC 1: the bf16 matrix pipe with an EXACT three-way split.  Every fp32 operand is cut by
//   truncation into three bf16 pieces x = x1 + x2 + x3 (8+8+8 significand bits, no rounding); the six
//   products W1X1, W1X2, W2X1, W1X3, W3X1, W2X2 (each exact in the fp32 accumulator) are summed by
//   12 v_mfma_f32_32x32x16_bf16; the dropped terms are <= 2^-24 |w||x|, i.e. the result is
//   fp32-equivalent (same error level as PREC 0, different rounding).  16x the MAC rate of PREC 0 and it
//   co-executes with the VALU (activations) of the partner wave.  Weight pieces are cut on the host.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int MAT_W = 1536;  // 32-bit words per bf16-split packed matrix: [piece 3][kstep 2][lane 64][4]

template <int PREC>
struct WFrag;

template <>
struct WFrag<0> {
  float wf[16];
  __device__ __forceinline__ void load(const float* __restrict__ m32, const unsigned* __restrict__, int mat, int lane) {
    const f32x4* p = reinterpret_cast<const f32x4*>(m32 + (size_t)mat * MAT_F) + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = p[q * 64];
      wf[4 * q + 0] = v.x; wf[4 * q + 1] = v.y; wf[4 * q + 2] = v.z; wf[4 * q + 3] = v.w;
    }
  }
  __device__ __forceinline__ f32x16 mul(const f32x16& in, f32x16 acc) const {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[r], in[r], acc, 0, 0, 0);
    return acc;
  }
};

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

template <>
struct WFrag<1> {
  u32x4 w[3][2];  // [piece][k-step], 8 bf16 each
  __device__ __forceinline__ void load(const float* __restrict__, const unsigned* __restrict__ m16, int mat, int lane) {
    const u32x4* p = reinterpret_cast<const u32x4*>(m16 + (size_t)mat * MAT_W) + lane;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
#pragma unroll
      for (int st = 0; st < 2; ++st) w[pc][st] = p[(pc * 2 + st) * 64];
  }
  // the operand split on its own, for callers that multiply one input by several weight blocks
  static __device__ __forceinline__ void split(const f32x16& in, u32x4 (&x)[3][2]) {
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const unsigned a = __float_as_uint(in[8 * st + 2 * q]), b = __float_as_uint(in[8 * st + 2 * q + 1]);
        const unsigned a1 = a & 0xFFFF0000u, b1 = b & 0xFFFF0000u;
        // the two remainders of a pair as ONE packed subtraction (exact either way)
        const f32x2_t r1 = f32x2_t{__uint_as_float(a), __uint_as_float(b)} - f32x2_t{__uint_as_float(a1), __uint_as_float(b1)};
        const unsigned ar = __float_as_uint(r1.x), br = __float_as_uint(r1.y);
        const unsigned a2 = ar & 0xFFFF0000u, b2 = br & 0xFFFF0000u;
        const f32x2_t r2 = r1 - f32x2_t{__uint_as_float(a2), __uint_as_float(b2)};  // exact; truncated on packing
        const unsigned a3 = __float_as_uint(r2.x), b3 = __float_as_uint(r2.y);
        x[0][st][q] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);  // {hi16(b), hi16(a)}
        x[1][st][q] = __builtin_amdgcn_perm(b2, a2, 0x07060302u);
        x[2][st][q] = __builtin_amdgcn_perm(b3, a3, 0x07060302u);
      }
  }
  __device__ __forceinline__ f32x16 mul_split(const u32x4 (&x)[3][2], f32x16 acc) const {
#pragma unroll
    for (int st = 0; st < 2; ++st) {  // smallest terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w[2][st]), as_bf16x8(x[0][st]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w[0][st]), as_bf16x8(x[2][st]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w[1][st]), as_bf16x8(x[1][st]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w[1][st]), as_bf16x8(x[0][st]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w[0][st]), as_bf16x8(x[1][st]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w[0][st]), as_bf16x8(x[0][st]), acc, 0, 0, 0);
    }
    return acc;
  }
  __device__ __forceinline__ f32x16 mul(const f32x16& in, f32x16 acc) const {
    u32x4 x[3][2];
    split(in, x);
    return mul_split(x, acc);
  }
};

// PREC 2: the f16 matrix pipe with a TWO-piece round-to-nearest split.
//   x1 = f16(x), x2 = f16(x - x1) (the remainder is exact in fp32), so |x - x1 - x2| <= 2^-24 |x|: the size of one
//   fp32 rounding.  Weights are split the same way on the host; the three products W2X1, W1X2, W1X1 (each exact in the
//   fp32 accumulator) leave out W2X2 <= 2^-24 |w||x|.  Half the MFMAs of PREC 1, and 4 VALU instructions per operand
//   pair instead of 11 (v_cvt_pk_f16_f32, two v_fma_mix_f32, v_cvt_pk_f16_f32).
//   f16 has a 5-bit exponent.  Low end: the second piece of |x| < 2^-2 falls into the f16 denormals (honoured by the
//   MFMA: tools/ubench/isa_rates.hip), absolute granularity 2^-24; weights (|w| ~ 0.1) therefore travel scaled by
//   F16_SW = 16 (exact power of two; their low piece is then normal down to |w| = 2^-6), activations by F16_SX, and the
//   accumulator holds F16_SX F16_SW times the true sum.  High end: F16_SX |activation| must stay below 65504; beyond
//   that the conversion gives inf, the products NaN, and the launch wrapper recomputes the affected walkers on the
//   PREC 1 path (egnn_kernel.hip: repair launch), so the limit costs time, never correctness.
constexpr float F16_SX = 1.0f, F16_SW = 16.0f;
constexpr bool F16_NODE_LAYERS = true;  // false: Wa, Wb, Wn1a, Wn1b (inputs: node features, message aggregate) stay on PREC 1
constexpr float F16_UNSCALE = 1.0f / (F16_SX * F16_SW);
constexpr float DIV_ST = 32.0f;  // scale carried by the feature tangents of the fast divergence kernel (egnn_div_kernel.hip)
constexpr float DIV_SV = 2048.0f;  // scale of the coordinate-head adjoint vc (its inputs gc o w_c2 are ~1e-4 for a fresh net)
constexpr int VEC_DIV_F = 4 * EH;
constexpr int MAT_WH = 1024;  // 32-bit words per f16-split packed matrix: [piece 2][kstep 2][lane 64][4]
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
// x - float(low / high half of pk) in ONE instruction (the compiler would emit v_cvt_f32_f16 + v_sub_f32); exact
__device__ __forceinline__ float f16_rem_lo(unsigned pk, float x) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x));
  return r;
}
__device__ __forceinline__ float f16_rem_hi(unsigned pk, float x) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x));
  return r;
}

template <>
struct WFrag<2> {
  u32x4 w[2][2];  // [piece][k-step], 8 f16 each
  __device__ __forceinline__ void load(const float* __restrict__, const unsigned* __restrict__ m16, int mat, int lane) {
    const u32x4* p = reinterpret_cast<const u32x4*>(m16 + (size_t)mat * MAT_WH) + lane;
#pragma unroll
    for (int pc = 0; pc < 2; ++pc)
#pragma unroll
      for (int st = 0; st < 2; ++st) w[pc][st] = p[(pc * 2 + st) * 64];
  }
  // `in` carries F16_SX times the true activations
  static __device__ __forceinline__ void split(const f32x16& in, u32x4 (&x)[2][2]) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float a = in[8 * st + 2 * q], b = in[8 * st + 2 * q + 1];
        const unsigned p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, f16x2));
        const float ra = f16_rem_lo(p1, a), rb = f16_rem_hi(p1, b);
        x[0][st][q] = p1;
        x[1][st][q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{ra, rb}, f16x2));
      }
  }
  __device__ __forceinline__ f32x16 mul_split(const u32x4 (&x)[2][2], f32x16 acc) const {
#pragma unroll
    for (int st = 0; st < 2; ++st) {  // smallest terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(w[1][st]), as_f16x8(x[0][st]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(w[0][st]), as_f16x8(x[1][st]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(w[0][st]), as_f16x8(x[0][st]), acc, 0, 0, 0);
    }
    return acc;
  }
  __device__ __forceinline__ f32x16 mul(const f32x16& in, f32x16 acc) const {
    u32x4 x[2][2];
    split(in, x);
    return mul_split(x, acc);
  }
};

// ---- packed fp32 helpers.  Measured on gfx950: a wave64 VALU instruction occupies its SIMD's issue for 4 cycles
// (8 for v_exp/v_rcp) whether it is v_mul_f32 or v_pk_mul_f32, so packed math (2 lanes-worth of fp32 per
// instruction) halves the issue cost of every mul/add/fma around the transcendentals.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// SiLU on PRE-SCALED pre-activations: the host folds kS = -log2(e) into the weights that produce a SiLU input
// (egnn_kernel.hip: pita_egnn_create), so v = kS z, exp(-z) = exp2(v) and the result v / (1 + exp2(v)) = kS silu(z)
// (the 1/kS is folded into the consumers).  Saves one multiply per activation in the VALU-bound edge loop.
constexpr float SILU_PRESCALE = -1.44269504088896341f;
__device__ __forceinline__ f32x2 silu2(f32x2 v) {
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(v.x);
  e.y = __builtin_amdgcn_exp2f(v.y);
  const f32x2 d = e + 1.0f;
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x);
  r.y = __builtin_amdgcn_rcpf(d.y);
  return v * r;
}
__device__ __forceinline__ void silu16(f32x16& m) {  // staged like silu16_staged below: same values
  f32x2 e[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    e[q].x = __builtin_amdgcn_exp2f(m[2 * q]);
    e[q].y = __builtin_amdgcn_exp2f(m[2 * q + 1]);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) e[q] = e[q] + 1.0f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    e[q].x = __builtin_amdgcn_rcpf(e[q].x);
    e[q].y = __builtin_amdgcn_rcpf(e[q].y);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x2 y = f32x2{m[2 * q], m[2 * q + 1]} * e[q];
    m[2 * q] = y.x;
    m[2 * q + 1] = y.y;
  }
}
// PREC 2 forms.  silu16_out: true-scale pre-activation in, F16_SX-scaled activation out (the scale rides on the
// reciprocal's argument: no extra instruction).  silu16_acc: the MFMA accumulator (F16_SX F16_SW times the true
// pre-activation) in, F16_SX-scaled activation out: one multiply more per element.
__device__ __forceinline__ f32x2 silu2_scaled(f32x2 v) {
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(v.x);
  e.y = __builtin_amdgcn_exp2f(v.y);
  const f32x2 c = {1.0f / F16_SX, 1.0f / F16_SX};
  const f32x2 d = __builtin_elementwise_fma(e, c, c);
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x);
  r.y = __builtin_amdgcn_rcpf(d.y);
  return v * r;
}
// staged over SILU_BATCH pairs at a time: the exponentials of several pairs are issued back to back, then the adds,
// the reciprocals, the products -- independent work between a transcendental and its consumer instead of hazard nops
#ifndef PITA_SILU_BATCH
#define PITA_SILU_BATCH 8
#endif
template <bool UNSCALE>
__device__ __forceinline__ void silu16_staged(f32x16& m) {
  constexpr int Q = PITA_SILU_BATCH;
  const f32x2 c = {1.0f / F16_SX, 1.0f / F16_SX};
#pragma unroll
  for (int q0 = 0; q0 < 8; q0 += Q) {
    f32x2 v[Q], e[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      v[q] = f32x2{m[2 * (q0 + q)], m[2 * (q0 + q) + 1]};
      if (UNSCALE) v[q] = v[q] * F16_UNSCALE;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      e[q].x = __builtin_amdgcn_exp2f(v[q].x);
      e[q].y = __builtin_amdgcn_exp2f(v[q].y);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) e[q] = __builtin_elementwise_fma(e[q], c, c);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      e[q].x = __builtin_amdgcn_rcpf(e[q].x);
      e[q].y = __builtin_amdgcn_rcpf(e[q].y);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const f32x2 y = v[q] * e[q];
      m[2 * (q0 + q)] = y.x;
      m[2 * (q0 + q) + 1] = y.y;
    }
  }
}
__device__ __forceinline__ void silu16_out(f32x16& m) { silu16_staged<false>(m); }
__device__ __forceinline__ void silu16_acc(f32x16& m) { silu16_staged<true>(m); }

__device__ __forceinline__ float dot16(const f32x16& w, const f32x16& m) {
  f32x2 acc = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 8; ++q)
    acc = __builtin_elementwise_fma(f32x2{w[2 * q], w[2 * q + 1]}, f32x2{m[2 * q], m[2 * q + 1]}, acc);
  return acc.x + acc.y;
}

__device__ __forceinline__ float xhalf_sum(float v) {
  // add the value held by the partner lane (l ^ 32): the other 16 features of the same column.  v_permlane32_swap_b32
  // (gfx950) exchanges the upper half of one register with the lower half of another inside the VALU: after swapping two
  // copies of v, one holds (lo, lo) and the other (hi, hi) -- no LDS round trip (ds_bpermute) on the edge loop's critical
  // path (attention gate, coordinate head).  lo + hi in every lane: the same bits as v + shfl_xor(v, 32).
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Park a resident weight fragment in accumulation registers (one-wave-per-SIMD kernels with 512 registers): the MFMA
// reads its A operand from there directly, so the resident matrices cost no VALU-visible registers.  Left to itself
// the allocator keeps them in VGPRs and spills other values around them.
__device__ __forceinline__ void frag_to_agpr(WFrag<2>& f) {
#pragma unroll
  for (int pc = 0; pc < 2; ++pc)
#pragma unroll
    for (int st = 0; st < 2; ++st) asm volatile("" : "+a"(f.w[pc][st]));
}
__device__ __forceinline__ void frag_to_agpr(WFrag<1>& f) {
#pragma unroll
  for (int pc = 0; pc < 3; ++pc)
#pragma unroll
    for (int st = 0; st < 2; ++st) asm volatile("" : "+a"(f.w[pc][st]));
}

// accurate_tanh (common.h) with both branches pinned: left alone the compiler turns the select back into a divergent
// branch (the exp / rcp side is "expensive"), which splits the edge loop's basic block.  Same values.
__device__ __forceinline__ float tanh_select(float v) {
  const float a = fabsf(v);
  const float v2 = v * v;
  float p = 62.0f / 2835.0f;
  p = fmaf(p, v2, -17.0f / 315.0f);
  p = fmaf(p, v2, 2.0f / 15.0f);
  p = fmaf(p, v2, -1.0f / 3.0f);
  p = fmaf(p, v2, 1.0f);
  float small = v * p;
  const float e = __builtin_amdgcn_exp2f(2.88539008177792681f * a);
  float big = copysignf(1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e), v);
  asm volatile("" : "+v"(small), "+v"(big));
  return a < 0.25f ? small : big;
}

}  // namespace pita

// makes `device` current for the lifetime of the guard (no-op when it already is)
struct PitaDeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit PitaDeviceGuard(int device) {
    if (device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~PitaDeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  PitaDeviceGuard(const PitaDeviceGuard&) = delete;
  PitaDeviceGuard& operator=(const PitaDeviceGuard&) = delete;
};

struct pita_egnn {
  pita_egnn_config cfg;
  unsigned* d_mats16 = nullptr;  // [L][M_COUNT][3][2][64][4]  bf16-split fragments
  unsigned* d_mats16h = nullptr; // [L][M_COUNT][2][2][64][4]  f16-split fragments of F16_SW x the matrices (forward only)
  float* d_mats = nullptr;       // [L][M_COUNT][4][64][4]     f32 fragments
  float* d_vecs = nullptr;       // [VEC_EMB_F + L*VEC_LAYER_F]
  float* d_vecs_h = nullptr;     // the same vectors with the PREC 2 scale factors folded in
  float* d_vecs_div = nullptr;   // [L][4][32] fragment order: DIV_ST kS (w_r + w_e), kS w_r, kS w_e, DIV_SV w_c2 / kS
  int* d_vjp_mark = nullptr;     // [B + 16] marks + flag of the reverse-mode kernel's f16 path (pita_egnn_vjp)
  size_t vjp_mark_bytes = 0;
  const void* shape = nullptr;   // pita::EgnnShape of egnn_kernel.hip
  const void* shape_small = nullptr;  // optional mapping with fewer walkers per wave, for batches that underfill the GPU
  int n_cu = 256;
  int device = -1;               // the device the handle's memory lives on: every entry point runs under it (a handle
                                 // may be called while another device is current; its lazily grown scratch buffers --
                                 // ws, divcache, mark, bk -- must land on ITS device).  One handle serves ONE stream at a
                                 // time: those buffers are shared by its launches (include/pita_hip.h, threading)
  float* d_ws = nullptr;         // reverse-mode checkpoint scratch (egnn_vjp_kernel.hip), grown on demand
  size_t ws_bytes = 0;
  float* d_divcache = nullptr;   // precision 2: per-edge primal factors of one trace (egnn_div_kernel.hip, DivCache)
  size_t divcache_bytes = 0;
  int* d_mark = nullptr;         // precision 2 divergence kernel: walkers left to the bf16x3 kernel (egnn_div_kernel.hip)
  size_t mark_bytes = 0;         // capacity of the marks; one flag word sits behind them (egnn_div_kernel.hip: bad_flag)
  int div_seq = 0;               // sequence number of the last launch that could mark walkers
  void* d_bk = nullptr;          // precision 2 sampler: walker backup + owed-moments markers for the repair launch
  size_t bk_bytes = 0;
};

