// Forward-mode derivative of the EDM denoiser around the wide EGNN backbone on the MATRIX pipe of gfx950 (MI355X).
//
// Replaces, for EGNN_dynamics_AD2_cat / egnn_aldp.EGNN_dynamics (paths relative to /root/reference/pita/src/models/components/):
//   utils.py:30-51          compute_divergence_exact (vmap(jacrev) of the score net)
//   energy_net.py:51-62     grad_x E_theta by autograd
//   sdes.py:218             dE_theta/dt by autograd through h(t)
// all of which are sums of directional derivatives  dD = J_x D(h, x) . vx + dD/dh . vh  of
// D(h, x) = c_s x + c_out F(c_noise(h), c_in(h) x, beta)  (score_net.py:13-43 around egnn_dynamics_ad2_cat.py:157-203).
//
// Mapping: egnn_wide64_kernel's one-walker-per-wave instantiation (one column tile: lane = (column, half of the 64 hidden
// features), dense layers as 2 x 2 blocks of the f16 two-piece MFMA tile) with ONE tangent direction carried beside the
// primal: every linear map is applied to the tangent with the same weight fragments (the tangent of a GEMM is the GEMM of
// the tangent), every SiLU passes its derivative sigma + y (1 - sigma) / kS on from the primal's own sigmoid (no further
// transcendental), tangents travel in the scaled units of their primals (egnn_common.h: kS, F16_SX, F16_SW), partner
// terms Wb dh_j and tangent positions in LDS tables beside the primal ones.  A walker whose primal or tangent leaves
// the f16 range is flagged and left to the fp32 vector-pipe kernel (egnn_wide_kernel.hip: egnn_wide_jvp_kernel).
#include "egnn_wide_mfma_common.h"

namespace pita {

struct Wide64JvpParams {
  const unsigned* m16h;
  const float* vecs;
  const float* est;
  int L, has_beta;
  float coord_scale;
  long long B;
  const float* x;
  const float* h;
  const float* beta;
  const float* vx;   // nullable [B, N*DIM]
  const float* vh;   // nullable [B]
  int dir;           // unit direction when vx is null (-1: none)
  float* out;        // nullable: D
  float* dout;       // nullable: dD
  float* dot_out;    // nullable: dot_out[b * dot_stride + dot_off] = <x_b, dD_b>
  long long dot_stride, dot_off;
  float* diag_acc;   // nullable: diag_acc[b] += dD[b, dir]
  int* bad;          // [B], zeroed by the launch wrapper: 1 = walker left to the vector-pipe kernel
};

template <int N, int DIM, int WAVES>
struct Wide64JvpCfg {
  static constexpr int NCOLP = 32 * ((N + 31) / 32);
  static constexpr int NT = NCOLP / 32;
  static constexpr int PB_F = NCOLP * W64_PBS;
  static constexpr int POS_F = NCOLP * DIM;
  static constexpr int WAVE_F = 2 * PB_F + 6 * POS_F;  // partner table + its tangent, pos[2], pos0 + their tangents
  static __host__ __device__ constexpr int vec_f(int L) { return ((W64_HEAD_F + L * W64_LAYER_F) + 3) & ~3; }
  static __host__ __device__ constexpr size_t lds_bytes(int L) {
    return sizeof(float) * (size_t)(vec_f(L) + N * 64 + WAVES * WAVE_F);
  }
};

// SiLU of egnn_common.h's PREC 2 forms with the derivative: in v = kS z (UNSCALE: the accumulator 16 kS z), out
// y = kS silu(z) in place and g = d silu / dz = s + (y / kS)(1 - s) with s the sigmoid the primal computed anyway
template <bool UNSCALE>
__device__ __forceinline__ void silu16_d(f32x16& m, f32x16& g) {
  const f32x2 c = {1.0f / F16_SX, 1.0f / F16_SX};
  constexpr float kSi = 1.0f / SILU_PRESCALE;
  f32x2 v[8], e[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    v[q] = f32x2{m[2 * q], m[2 * q + 1]};
    if (UNSCALE) v[q] = v[q] * F16_UNSCALE;
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    e[q].x = __builtin_amdgcn_exp2f(v[q].x);
    e[q].y = __builtin_amdgcn_exp2f(v[q].y);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) e[q] = __builtin_elementwise_fma(e[q], c, c);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    e[q].x = __builtin_amdgcn_rcpf(e[q].x);
    e[q].y = __builtin_amdgcn_rcpf(e[q].y);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x2 y = v[q] * e[q];
    const f32x2 one = {1.0f, 1.0f};
    const f32x2 gq = __builtin_elementwise_fma(y * kSi, one - e[q], e[q]);
    m[2 * q] = y.x; m[2 * q + 1] = y.y;
    g[2 * q] = gq.x; g[2 * q + 1] = gq.y;
  }
}

template <int N, int DIM, int WAVES, bool ATT, bool TANH>
__global__ void __launch_bounds__(WAVES * 64, 1) egnn_wide64_jvp_kernel(Wide64JvpParams p) {
  using C = Wide64JvpCfg<N, DIM, WAVES>;
  static_assert(C::NT == 1, "one walker per wave, one column tile");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = p.L;
  const int vec_f = C::vec_f(L);
  for (int i = threadIdx.x; i < W64_HEAD_F + L * W64_LAYER_F; i += WAVES * 64) lds[i] = p.vecs[i];
  float* est = lds + vec_f;
  for (int i = threadIdx.x; i < N * 64; i += WAVES * 64) est[i] = p.est[i];
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane & 31, hh = lane >> 5;
  float* PB = est + N * 64 + wave * C::WAVE_F;
  float* dPB = PB + C::PB_F;
  float* posbuf0 = dPB + C::PB_F;
  float* posbuf1 = posbuf0 + C::POS_F;
  float* pos0 = posbuf1 + C::POS_F;
  float* dposbuf0 = pos0 + C::POS_F;
  float* dposbuf1 = dposbuf0 + C::POS_F;
  float* dpos0 = dposbuf1 + C::POS_F;
  const f32x16 zero16 = {0};
  const int col = cl, nodei = cl < N ? cl : 0;
  const bool valid = cl < N;
  const int live = valid ? 1 : 0;

  for (long long w = (long long)blockIdx.x * WAVES + wave; w < p.B; w += (long long)gridDim.x * WAVES) {
    const float hv = p.h[w];
    const float bet = p.has_beta ? p.beta[w] : 0.f;
    const float vh = p.vh ? p.vh[w] : 0.f;
    // score_net.py:26-29 and their h-derivatives
    const float c_s = 1.0f / (1.0f + hv), c_in = 1.0f / sqrtf(1.0f + hv), sh = sqrtf(hv);
    const float c_out = sh * c_in, tfeat = 0.125f * logf(hv);
    const float dc_s = -c_s * c_s, dc_in = -0.5f * c_in * c_s, dc_out = 0.5f * c_in / sh + sh * dc_in;
    const float dtf = vh * (0.125f / hv);
    float xin[DIM], dxin[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      const long long e = (w * N + col) * DIM + k;
      xin[k] = valid ? p.x[e] : 0.f;
      dxin[k] = valid ? (p.vx ? p.vx[e] : ((col * DIM + k) == p.dir ? 1.0f : 0.0f)) : 0.f;
      const float ps = c_in * xin[k], dps = fmaf(c_in, dxin[k], (vh * dc_in) * xin[k]);
      if (hh == 0) {
        pos0[col * DIM + k] = ps; posbuf0[col * DIM + k] = ps;
        dpos0[col * DIM + k] = dps; dposbuf0[col * DIM + k] = dps;
      }
    }
    f32x16 hfeat[2], dhfeat[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const f32x16 wt = lds_vec16(lds + b * 32 + hh * 16), wb = lds_vec16(lds + 64 + b * 32 + hh * 16);
      const f32x16 es = lds_vec16(est + nodei * 64 + b * 32 + hh * 16);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        hfeat[b][r] = fmaf(wt[r], tfeat, fmaf(wb[r], bet, es[r]));
        dhfeat[b][r] = wt[r] * dtf;
      }
    }
    wave_lds_fence();

    float* poscur = posbuf0; float* posnext = posbuf1;
    float* dposcur = dposbuf0; float* dposnext = dposbuf1;
    for (int l = 0; l < L; ++l) {
      const unsigned* ml = p.m16h + (size_t)l * WM_COUNT * W64_MAT_W;
      const float* vbase = lds + W64_HEAD_F + l * W64_LAYER_F;
      const float* vl = vbase + hh * 16;
      const bool last = (l == L - 1);
      const float aggw = last ? 0.0f : 1.0f;
      {  // partner tables Wb h_col, Wb dh_col
        f32x16 pb[2] = {zero16, zero16}, dpb[2] = {zero16, zero16};
        w64_mul_stream(ml, WM_WB, lane, hfeat, pb);
        w64_mul_stream(ml, WM_WB, lane, dhfeat, dpb);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          pb[b] *= F16_UNSCALE; dpb[b] *= F16_UNSCALE;
          lds_store16(PB + col * W64_PBS + b * 32 + hh * 16, pb[b]);
          lds_store16(dPB + col * W64_PBS + b * 32 + hh * 16, dpb[b]);
        }
      }
      wave_lds_fence();

      W64Mat w2f, wc1f;
      w2f.load(ml, WM_W2, lane);
      wc1f.load(ml, WM_WC1, lane);
      w2f.to_agpr();
      wc1f.to_agpr();
      const float a_re0 = vbase[WV_WRE * 64 + lane], a_re1 = vbase[WV_WRE * 64 + 64 + lane];
      const float b_att = vbase[WV_COUNT * 64];

      f32x16 Ai[2] = {lds_vec16(vl + WV_B1 * 64), lds_vec16(vl + WV_B1 * 64 + 32)}, dAi[2] = {zero16, zero16};
      w64_mul_stream(ml, WM_WA, lane, hfeat, Ai);
      w64_mul_stream(ml, WM_WA, lane, dhfeat, dAi);
      Ai[0] *= F16_UNSCALE; Ai[1] *= F16_UNSCALE; dAi[0] *= F16_UNSCALE; dAi[1] *= F16_UNSCALE;
      f32x16 agg[2] = {zero16, zero16}, dagg[2] = {zero16, zero16};
      float xacc[DIM], dxacc[DIM], pown[DIM], p0own[DIM], dpown[DIM], dp0own[DIM];
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        xacc[k] = 0.f; dxacc[k] = 0.f;
        pown[k] = poscur[col * DIM + k]; p0own[k] = pos0[col * DIM + k];
        dpown[k] = dposcur[col * DIM + k]; dp0own[k] = dpos0[col * DIM + k];
      }
      for (int dd = 1; dd < N; ++dd) {
        asm volatile("" ::: "memory");
        int j = nodei + dd * live;
        j = (j >= N) ? j - N : j;
        const int cj = valid ? j : col;
        float df[DIM], ddf[DIM], radial = 0.f, ea = 0.f, dradial = 0.f, dea = 0.f;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          df[k] = pown[k] - poscur[cj * DIM + k];
          ddf[k] = dpown[k] - dposcur[cj * DIM + k];
          radial = fmaf(df[k], df[k], radial);
          dradial = fmaf(df[k], ddf[k], dradial);
          const float e0 = p0own[k] - pos0[cj * DIM + k];
          ea = fmaf(e0, e0, ea);
          dea = fmaf(e0, dp0own[k] - dpos0[cj * DIM + k], dea);
        }
        dradial *= 2.0f; dea *= 2.0f;
        const float geo = hh ? ea : radial, dgeo = hh ? dea : dradial;
        f32x16 m[2], dm[2], g[2];
        m[0] = Ai[0] + lds_vec16(PB + cj * W64_PBS + hh * 16);
        m[1] = Ai[1] + lds_vec16(PB + cj * W64_PBS + 32 + hh * 16);
        dm[0] = dAi[0] + lds_vec16(dPB + cj * W64_PBS + hh * 16);
        dm[1] = dAi[1] + lds_vec16(dPB + cj * W64_PBS + 32 + hh * 16);
        m[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re0, geo, m[0], 0, 0, 0);
        m[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re1, geo, m[1], 0, 0, 0);
        dm[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re0, dgeo, dm[0], 0, 0, 0);
        dm[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re1, dgeo, dm[1], 0, 0, 0);
        silu16_d<false>(m[0], g[0]);
        silu16_d<false>(m[1], g[1]);
        dm[0] *= g[0]; dm[1] *= g[1];
        f32x16 z[2] = {lds_vec16(vl + WV_B2 * 64), lds_vec16(vl + WV_B2 * 64 + 32)}, dz[2] = {zero16, zero16};
        w2f.mul(m, z);
        w2f.mul(dm, dz);
        silu16_d<true>(z[0], g[0]);
        silu16_d<true>(z[1], g[1]);
        dz[0] *= g[0] * F16_UNSCALE; dz[1] *= g[1] * F16_UNSCALE;
        if (ATT) {
          const f32x16 wa0 = lds_vec16(vl + WV_WATT * 64), wa1 = lds_vec16(vl + WV_WATT * 64 + 32);
          const float s = xhalf_sum(dot16(wa0, z[0]) + dot16(wa1, z[1])) + b_att;
          const float ds = xhalf_sum(dot16(wa0, dz[0]) + dot16(wa1, dz[1]));
          const float att = fast_sigmoid(s);
          const float datt = att * (1.0f - att) * ds;
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              dz[b][r] = fmaf(dz[b][r], att, z[b][r] * datt);
              z[b][r] *= att;
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            agg[b][r] = fmaf(z[b][r], aggw, agg[b][r]);
            dagg[b][r] = fmaf(dz[b][r], aggw, dagg[b][r]);
          }
        f32x16 c1[2] = {lds_vec16(vl + WV_BC1 * 64), lds_vec16(vl + WV_BC1 * 64 + 32)}, dc1[2] = {zero16, zero16};
        wc1f.mul(z, c1);
        wc1f.mul(dz, dc1);
        silu16_d<true>(c1[0], g[0]);
        silu16_d<true>(c1[1], g[1]);
        dc1[0] *= g[0] * F16_UNSCALE; dc1[1] *= g[1] * F16_UNSCALE;
        const f32x16 wc0 = lds_vec16(vl + WV_WC2 * 64), wc1v = lds_vec16(vl + WV_WC2 * 64 + 32);
        float cs = xhalf_sum(dot16(wc0, c1[0]) + dot16(wc1v, c1[1]));
        float dcs = xhalf_sum(dot16(wc0, dc1[0]) + dot16(wc1v, dc1[1]));
        if (TANH) {
          const float th = tanh_select(cs);
          dcs = (1.0f - th * th) * p.coord_scale * dcs;
          cs = th * p.coord_scale;
        }
        const float sq = __builtin_amdgcn_sqrtf(radial + 1e-8f);
        const float inrm = __builtin_amdgcn_rcpf(sq + 1.0f);
        const float dinrm = -inrm * inrm * (0.5f * dradial * __builtin_amdgcn_rcpf(sq));
        const float dsc = fmaf(dinrm, cs, inrm * dcs);
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          xacc[k] = fmaf(df[k] * inrm, cs, xacc[k]);
          dxacc[k] = fmaf(ddf[k], inrm * cs, fmaf(df[k], dsc, dxacc[k]));
        }
      }
#pragma unroll
      for (int k = 0; k < DIM; ++k)
        if (hh == 0) {
          posnext[col * DIM + k] = pown[k] + xacc[k];
          dposnext[col * DIM + k] = dpown[k] + dxacc[k];
        }
      if (!last) {  // node model, recurrent
        f32x16 n1[2] = {lds_vec16(vl + WV_BN1 * 64), lds_vec16(vl + WV_BN1 * 64 + 32)}, dn1[2] = {zero16, zero16}, g[2];
        w64_mul_stream(ml, WM_WN1A, lane, hfeat, n1);
        w64_mul_stream(ml, WM_WN1B, lane, agg, n1);
        w64_mul_stream(ml, WM_WN1A, lane, dhfeat, dn1);
        w64_mul_stream(ml, WM_WN1B, lane, dagg, dn1);
        silu16_d<true>(n1[0], g[0]);
        silu16_d<true>(n1[1], g[1]);
        dn1[0] *= g[0] * F16_UNSCALE; dn1[1] *= g[1] * F16_UNSCALE;
        f32x16 o[2] = {lds_vec16(vl + WV_BN2 * 64), lds_vec16(vl + WV_BN2 * 64 + 32)}, d_o[2] = {zero16, zero16};
        w64_mul_stream(ml, WM_WN2, lane, n1, o);
        w64_mul_stream(ml, WM_WN2, lane, dn1, d_o);
        hfeat[0] += o[0] * F16_UNSCALE; hfeat[1] += o[1] * F16_UNSCALE;
        dhfeat[0] += d_o[0] * F16_UNSCALE; dhfeat[1] += d_o[1] * F16_UNSCALE;
      }
      wave_lds_fence();
      float* tmp = poscur; poscur = posnext; posnext = tmp;
      tmp = dposcur; dposcur = dposnext; dposnext = tmp;
    }

    // ---- F = x_final - x (mean-free), D = c_s x + c_out F and their tangents; reductions over the walker
    float* scr = PB;        // [NCOLP][DIM] F, then per-column <x, dD> and the non-finite flags
    float* dscr = dPB;      // [NCOLP][DIM] dF
    float F[DIM], dF[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      F[k] = poscur[col * DIM + k] - pos0[col * DIM + k];
      dF[k] = dposcur[col * DIM + k] - dpos0[col * DIM + k];
      if (hh == 0) { scr[col * DIM + k] = F[k]; dscr[col * DIM + k] = dF[k]; }
    }
    wave_lds_fence();
    float Dv[DIM], dD[DIM], part = 0.f;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      float s = 0.f, ds = 0.f;
      for (int q = 0; q < N; ++q) { s += scr[q * DIM + k]; ds += dscr[q * DIM + k]; }
      F[k] -= s / (float)N;
      dF[k] -= ds / (float)N;
      Dv[k] = fmaf(c_s, xin[k], c_out * F[k]);
      dD[k] = fmaf(c_s, dxin[k], fmaf(c_out, dF[k], vh * fmaf(dc_s, xin[k], dc_out * F[k])));
      part = fmaf(xin[k], dD[k], part);
      ok = ok && __builtin_isfinite(Dv[k]) && __builtin_isfinite(dD[k]);
    }
    wave_lds_fence();
    float* red = scr;       // [NCOLP][2]: <x, dD> of the column, its non-finite flag
    if (hh == 0) { red[col * 2] = valid ? part : 0.f; red[col * 2 + 1] = (valid && !ok) ? 1.0f : 0.0f; }
    wave_lds_fence();
    float dot = 0.f, nbad = 0.f;
    for (int q = 0; q < N; ++q) { dot += red[q * 2]; nbad += red[q * 2 + 1]; }
    if (nbad != 0.f) {  // wave-uniform: the walker is one wave's
      if (lane == 0) p.bad[w] = 1;
    } else if (valid && hh == 0) {
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        const long long e = (w * N + col) * DIM + k;
        if (p.out) p.out[e] = Dv[k];
        if (p.dout) p.dout[e] = dD[k];
        if (p.diag_acc && !p.vx && (col * DIM + k) == p.dir) p.diag_acc[w] += dD[k];
      }
      if (p.dot_out && col == 0) p.dot_out[w * p.dot_stride + p.dot_off] = dot;
    }
    wave_lds_fence();
  }
}

struct Wide64JvpShape {
  int n, dim, waves;
  void (*kernel[2][2])(Wide64JvpParams);  // [attention][tanh]
  size_t (*lds_bytes)(int);
};
template <int N, int DIM, int WAVES>
static size_t wide64_jvp_lds_of(int L) { return Wide64JvpCfg<N, DIM, WAVES>::lds_bytes(L); }
#define PITA_WIDE64_JVP_SHAPE(N, DIM, WAVES)                                                                               \
  Wide64JvpShape { N, DIM, WAVES,                                                                                          \
                   {{egnn_wide64_jvp_kernel<N, DIM, WAVES, false, false>, egnn_wide64_jvp_kernel<N, DIM, WAVES, false, true>}, \
                    {egnn_wide64_jvp_kernel<N, DIM, WAVES, true, false>, egnn_wide64_jvp_kernel<N, DIM, WAVES, true, true>}},  \
                   wide64_jvp_lds_of<N, DIM, WAVES> }
// alanine dipeptide (22 atoms); other particle counts take the vector-pipe kernel
static const Wide64JvpShape kWide64JvpShapes[] = {PITA_WIDE64_JVP_SHAPE(22, 3, 4)};

// returns PITA_OK when the matrix-pipe kernel took the launch, 1 when the particle system has no instantiation
int wide64_jvp(pita_egnn_wide* net, const float* h, const float* x, const float* beta, const float* vx, int dir,
               const float* vh, float* out, float* dout, float* dot_out, long long dot_stride, long long dot_off,
               float* diag_acc, int* bad, long long B, hipStream_t stream) {
  if (!net->shape64) return 1;
  const Wide64JvpShape* s = nullptr;
  for (const auto& t : kWide64JvpShapes)
    if (t.n == net->cfg.n_particles && t.dim == net->cfg.n_dim) s = &t;
  if (!s || s->lds_bytes(net->cfg.n_layers) > 160 * 1024) return 1;
  auto kernel = s->kernel[net->cfg.attention ? 1 : 0][net->cfg.tanh ? 1 : 0];
  const size_t lds = s->lds_bytes(net->cfg.n_layers);
  PITA_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds));
  Wide64JvpParams p{};
  p.m16h = net->d_m16h; p.vecs = net->d_vecs64; p.est = net->d_est64;
  p.L = net->cfg.n_layers; p.has_beta = net->cfg.condition_beta;
  p.coord_scale = net->cfg.coords_range / (float)net->cfg.n_layers;
  p.B = B; p.x = x; p.h = h; p.beta = beta; p.vx = vx; p.vh = vh; p.dir = vx ? -1 : dir;
  p.out = out; p.dout = dout; p.dot_out = dot_out; p.dot_stride = dot_stride; p.dot_off = dot_off; p.diag_acc = diag_acc;
  p.bad = bad;
  const long long want = (B + s->waves - 1) / s->waves, cap = net->n_cu;  // one 4-wave block per CU
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(s->waves * 64), lds, stream, p);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

}  // namespace pita
