// Shared pieces of the matrix-pipe kernels of the wide EGNN backbone (egnn_wide_mfma_kernel.hip: forward / sampler;
// egnn_wide_mfma_jvp_kernel.hip: forward-mode derivative): LDS / fragment layout constants, the parameter block, the
// 64 x 64 dense layer as 2 x 2 blocks of the f16 two-piece MFMA tile.
#pragma once
#include "egnn_common.h"
#include "egnn_wide_common.h"

namespace pita {

constexpr int W64_PBS = 68;  // LDS row stride (floats) of the partner table: 64 + 4, conflict-free ds_read_b128
enum { WM_WA = 0, WM_WB, WM_W2, WM_WC1, WM_WN1A, WM_WN1B, WM_WN2, WM_COUNT };
constexpr int W64_MAT_W = 4 * MAT_WH;  // words per 64 x 64 matrix: blocks [out block][in block], each a WFrag<2> fragment
// per-layer vectors, 64 floats each in fragment order [block][hh][r] unless noted
enum { WV_WRE = 0 /* 128 floats: [block][w_r 32 | w_e 32], natural order (A operand of the f32 k-step) */, WV_B1 = 2, WV_B2,
       WV_WATT, WV_BC1, WV_WC2, WV_BN1, WV_BN2, WV_COUNT };
constexpr int W64_HEAD_F = 128;                     // emb_t, emb_beta (fragment order)
constexpr int W64_LAYER_F = WV_COUNT * 64 + 4;      // + b_att

struct Wide64Params {
  const unsigned* m16h;
  const float* vecs;
  const float* est;
  int L, attention, tanh_on, has_beta;
  float coord_scale;
  long long B;
  int mode;  // 0 backbone forward (t = its time input), 1 denoiser, 2 score (t = h = sigma^2)
  const float* x;
  const float* t;
  const float* beta;
  float* out;
  // mode 3: n_steps Euler-Maruyama steps of the not-debiased reverse SDE in one launch (pita_egnn_wide_sampler_run)
  float* xs;               // [B, N*DIM] walkers, in place
  const float* step_tab;   // [n_steps][PITA_STEP_STRIDE]
  int n_steps;
  const float* noise;      // nullable [n_steps, B, N*DIM]
  unsigned long long seed, walker_offset;
  long long step0;
  int remove_mean;
  double* stats_out;       // nullable [n_steps][4]
  int* bad_from;           // [B*N]: first step whose moments this launch left out for the particle (INT_MAX: none)
  int* flag;               // nullable: set to 1 when a result of this launch is not finite (the repair pass has work)
};

template <int N, int DIM, int G, int WAVES>
struct Wide64Cfg {
  static constexpr int NCOL = G * N;
  static constexpr int NT = (NCOL + 31) / 32;
  static constexpr int NCOLP = NT * 32;
  static constexpr int PB_F = NCOLP * W64_PBS;
  static constexpr int POS_F = NCOLP * DIM;
  static constexpr int WAVE_F = PB_F + 4 * POS_F;  // partner table, pos[2], pos0, the walkers' unscaled coordinates
  static __host__ __device__ constexpr int vec_f(int L) { return ((W64_HEAD_F + L * W64_LAYER_F) + 3) & ~3; }
  static __host__ __device__ constexpr size_t lds_bytes(int L) {
    return sizeof(float) * (size_t)(vec_f(L) + N * 64 + WAVES * WAVE_F);
  }
};

#ifndef PITA_WIDE64_AGPR_WEIGHTS
#define PITA_WIDE64_AGPR_WEIGHTS 1
#endif
// the four 32 x 32 blocks of a 64 x 64 matrix, resident
struct W64Mat {
  WFrag<2> b[2][2];
  __device__ __forceinline__ void load(const unsigned* __restrict__ layer, int mat, int lane) {
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) b[ob][kb].load(nullptr, layer, mat * 4 + ob * 2 + kb, lane);
  }
  // park the fragments in accumulation registers: the MFMA reads its A operand from there directly, so the resident
  // matrices cost no VALU-visible registers (left to itself the allocator keeps them in VGPRs and spills around them)
  __device__ __forceinline__ void to_agpr() {
#if PITA_WIDE64_AGPR_WEIGHTS
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
          for (int st = 0; st < 2; ++st) asm volatile("" : "+a"(b[ob][kb].w[pc][st]));
#endif
  }
  __device__ __forceinline__ void mul(const f32x16 (&in)[2], f32x16 (&acc)[2]) const {
    u32x4 xs[2][2][2];
    WFrag<2>::split(in[0], xs[0]);
    WFrag<2>::split(in[1], xs[1]);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) acc[ob] = b[ob][kb].mul_split(xs[kb], acc[ob]);
  }
};
// the same product with the blocks streamed from memory one at a time (per-node layers: used once per tile and layer)
__device__ __forceinline__ void w64_mul_stream(const unsigned* __restrict__ layer, int mat, int lane, const f32x16 (&in)[2],
                                               f32x16 (&acc)[2]) {
  u32x4 xs[2][2][2];
  WFrag<2>::split(in[0], xs[0]);
  WFrag<2>::split(in[1], xs[1]);
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      WFrag<2> w;
      w.load(nullptr, layer, mat * 4 + ob * 2 + kb, lane);
      acc[ob] = w.mul_split(xs[kb], acc[ob]);
    }
}

__device__ __forceinline__ void lds_store16(float* dst, const f32x16& v) {
  f32x4* d = reinterpret_cast<f32x4*>(dst);
#pragma unroll
  for (int q = 0; q < 4; ++q) d[q] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}

}  // namespace pita
