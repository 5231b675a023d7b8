// Forward-mode derivative (JVP) of the EDM-preconditioned EGNN denoiser for gfx950.
//
// Foundation of the debiased (Feynman-Kac) regime of the reference sampler
// (pita/src/models/components/sdes.py:151-239), which needs, per walker and step,
//   * div_x s_theta            (utils.py:30-51: vmap(jacrev) -> trace of the Jacobian),
//   * grad_x E_theta            (energy_net.py:51-62, autograd),
//   * d E_theta / dt            (sdes.py:218, autograd through h(t)).
// All three are linear in JVPs of the denoiser D(h, x) = c_s x + c_out F(c_noise, c_in x, beta)
// (score_net.py:21-33): with d_k = J_x D e_k and d_h = dD/dh,
//   div s      = (sum_k (d_k)_k - dim) / h,
//   grad_x E   = ((1 + c_s) x - D - [<x, d_k>]_k) / h          (E = (1+c_s)|x|^2/(2h) - <D, x>/h),
//   dE/dh      = d/dh[(1+c_s)/(2h)] |x|^2 + <D, x>/h^2 - <d_h, x>/h.
// This kernel evaluates D and ONE tangent direction (vx, vh) per launch for all walkers; the host loops
// over the dim + 1 directions (pita_amd/sdes.py).
//
// Mapping: identical to egnn_kernel.hip (wave = up to G walkers = dense 32-column tiles, lane = column x 16
// features, bf16 matrix pipe with the exact 3-way operand split); the tangent of every quantity lives in the
// SAME lane/register position as its primal, so each stage's derivative is plain in-lane arithmetic:
//   linear layers     d(Wx + b) = W dx             (a second MFMA chain with the same weight fragments)
//   SiLU              d silu(z) = sigma(z) (1 + z (1 - sigma(z))) dz
//   attention gate    d(att m2) = att (1 - att) dlogit m2 + att dm2
//   tanh head         d tanh(c) = (1 - tanh^2) dc
//   geometry          d|d|^2 = 2 <d, dd>,  d(d / (|d| + 1)) = (dd - u dnrm) / nrm
// One wave per SIMD (512 VGPRs: primal + tangent state of up to three column tiles) and primal + tangent partner
// tables in LDS (35 KB per wave); LJ13 runs one tile per wave at two waves per SIMD (see kJvpShapes).
#include <cstdlib>

#include "egnn_common.h"

namespace pita {

struct JvpParams {
  const unsigned* mats16;
  const float* vecs;
  int n_layers, in_nf, attention, tanh_on, feature_layout;
  float coord_scale;
  long long B;
  const float* h;      // [B] sigma^2
  const float* x;      // [B, D]
  const float* beta;   // [B] or null
  const float* vx;     // [B, D] tangent of x, or null
  int dir;             // used when vx == null: unit direction e_dir (0 <= dir < D), or -1 for vx = 0
  const float* vh;     // [B] tangent of h, or null (= 0)
  float* out;          // [B, D] denoiser D (nullable)
  float* dout;         // [B, D] JVP of D (nullable)
  float* dot_out;      // [B] <x, dD> (nullable; strided by dot_stride, offset by dot_off: writes dot_out[b*stride+off])
  long long dot_stride, dot_off;
  float* diag_acc;     // [B] += dD[b, dir] (nullable; only for unit directions)
};

template <int N, int DIM, int G, int WAVES>
struct JvpCfg {
  static constexpr int NCOL = G * N;
  static constexpr int NT = (NCOL + 31) / 32;
  static constexpr int NCOLP = NT * 32;
  static constexpr int PB_F = NCOLP * PBS;
  static constexpr int POS_F = NCOLP * DIM;
  static constexpr int WAVE_F = 2 * PB_F + 6 * POS_F;  // PB, dPB, pos[2], dpos[2], pos0, dpos0
  static __host__ __device__ constexpr int vec_f(int L) { return ((VEC_EMB_F + L * VEC_LAYER_F) + 3) & ~3; }
  static __host__ __device__ constexpr size_t lds_bytes(int L) {
    return sizeof(float) * (size_t)(vec_f(L) + WAVES * WAVE_F);
  }
};

// silu on pre-scaled pre-activations (v = kS z, see egnn_common.h) and the derivative factor of the scaled map:
// y' = kS silu(z) = v s, dy' = g dv with g = silu'(z) = s (1 + z (1 - s)), s = sigmoid(z) = 1/(1 + exp2(v)), z = v / kS
__device__ __forceinline__ void silu_dsilu(float v, float& y, float& g) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v));
  y = v * s;
  g = s * fmaf(v * (1.0f / SILU_PRESCALE), 1.0f - s, 1.0f);
}

template <int N, int DIM, int G, int WAVES, int OCC>
__global__ void __launch_bounds__(WAVES * 64, OCC) egnn_jvp_kernel(JvpParams p) {
  using C = JvpCfg<N, DIM, G, WAVES>;
  constexpr int NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = p.n_layers;
  const int vec_f = C::vec_f(L);
  for (int i = threadIdx.x; i < VEC_EMB_F + L * VEC_LAYER_F; i += WAVES * 64) lds[i] = p.vecs[i];
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane & 31, hh = lane >> 5;
  float* PB = lds + vec_f + wave * C::WAVE_F;
  float* dPB = PB + C::PB_F;
  float* posb[2] = {dPB + C::PB_F, dPB + C::PB_F + C::POS_F};
  float* dposb[2] = {posb[1] + C::POS_F, posb[1] + 2 * C::POS_F};
  float* pos0 = dposb[1] + C::POS_F;
  float* dpos0 = pos0 + C::POS_F;
  const float* vemb = lds;

  const long long total_waves = (long long)gridDim.x * WAVES;
  const long long quota = (p.B + total_waves - 1) / total_waves;
  const long long wbeg = ((long long)blockIdx.x * WAVES + wave) * quota;
  const long long wend = (wbeg + quota < p.B) ? wbeg + quota : p.B;
  for (long long walker0 = wbeg; walker0 < wend; walker0 += G) {
    const int nwalk = (int)((wend - walker0) < G ? (wend - walker0) : G);
    const int ncol = nwalk * N;
    const int ntile = (ncol + 31) >> 5;
    int col[NT], nodei[NT];
    bool valid[NT];
    float xin[NT][DIM], vxin[NT][DIM];
    float c_s[NT], c_in[NT], c_out[NT], dc_s[NT], dc_in[NT], dc_out[NT], vhv[NT], hval[NT];
    float posi[NT][DIM], dposi[NT][DIM], p0i[NT][DIM], dp0i[NT][DIM];
    f32x16 hf[NT], dhf[NT];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      col[T] = T * 32 + cl;
      const int w = col[T] / N;
      nodei[T] = col[T] - w * N;
      valid[T] = col[T] < ncol;
      const long long wid = valid[T] ? walker0 + w : p.B - 1;
      const float hv = p.h[wid];
      const float bet = p.beta ? p.beta[wid] : 0.f;
      vhv[T] = p.vh ? p.vh[wid] : 0.f;
      hval[T] = hv;
      // EDM coefficients and their h-derivatives (score_net.py:26-29)
      const float op = 1.0f + hv, rs = 1.0f / sqrtf(op), sh = sqrtf(hv);
      c_s[T] = 1.0f / op;
      dc_s[T] = -c_s[T] * c_s[T];
      c_in[T] = rs;
      dc_in[T] = -0.5f * rs / op;
      c_out[T] = sh * rs;
      dc_out[T] = 0.5f * rs / sh + sh * dc_in[T];
      const float tfeat = 0.125f * logf(hv), dtfeat = 0.125f / hv * vhv[T];
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        const long long gi = (walker0 * N + col[T]) * DIM + k;
        xin[T][k] = valid[T] ? p.x[gi] : 0.f;
        float v = 0.f;
        if (p.vx) v = valid[T] ? p.vx[gi] : 0.f;
        else if (valid[T] && nodei[T] * DIM + k == p.dir) v = 1.0f;
        vxin[T][k] = v;
        posi[T][k] = c_in[T] * xin[T][k];
        dposi[T][k] = fmaf(c_in[T], v, dc_in[T] * vhv[T] * xin[T][k]);
        p0i[T][k] = posi[T][k];
        dp0i[T][k] = dposi[T][k];
        if (hh == 0) {
          pos0[col[T] * DIM + k] = posi[T][k];
          posb[0][col[T] * DIM + k] = posi[T][k];
          dpos0[col[T] * DIM + k] = dposi[T][k];
          dposb[0][col[T] * DIM + k] = dposi[T][k];
        }
      }
      // node features (egnn_temp_conditioned.py:63-78): slots holding the time feature carry its tangent
      float a0, a1, da0, da1;
      if (p.in_nf == 1) { a0 = tfeat; a1 = 0.f; da0 = dtfeat; da1 = 0.f; }
      else if (p.feature_layout == 0) {
        const bool t0 = 2 * nodei[T] < N, t1 = 2 * nodei[T] + 1 < N;
        a0 = t0 ? tfeat : bet; a1 = t1 ? tfeat : bet; da0 = t0 ? dtfeat : 0.f; da1 = t1 ? dtfeat : 0.f;
      } else { a0 = tfeat; a1 = bet; da0 = dtfeat; da1 = 0.f; }
      const f32x16 w0 = lds_vec16(vemb + hh * 16), w1 = lds_vec16(vemb + 32 + hh * 16), eb = lds_vec16(vemb + 64 + hh * 16);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        hf[T][r] = fmaf(w0[r], a0, fmaf(w1[r], a1, eb[r]));
        dhf[T][r] = fmaf(w0[r], da0, w1[r] * da1);
      }
    }
    wave_lds_fence();

    int cur = 0;
    for (int l = 0; l < L; ++l) {
      const unsigned* mats16 = p.mats16 + (size_t)l * M_COUNT * MAT_W;
      const float* vl = lds + VEC_EMB_F + l * VEC_LAYER_F + hh * 16;
      const bool last = (l == L - 1);
      const float* poscur = posb[cur];
      const float* dposcur = dposb[cur];
      float* posnext = posb[cur ^ 1];
      float* dposnext = dposb[cur ^ 1];
      {
        WFrag<1> wb;
        wb.load(nullptr, mats16, M_WB, lane);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          if (T >= ntile) continue;
          const f32x16 z = {0};
          const f32x16 pb = wb.mul(hf[T], z), dpb = wb.mul(dhf[T], z);
          f32x4* dst = reinterpret_cast<f32x4*>(PB + col[T] * PBS + hh * 16);
          f32x4* ddst = reinterpret_cast<f32x4*>(dPB + col[T] * PBS + hh * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            dst[q] = f32x4{pb[4 * q], pb[4 * q + 1], pb[4 * q + 2], pb[4 * q + 3]};
            ddst[q] = f32x4{dpb[4 * q], dpb[4 * q + 1], dpb[4 * q + 2], dpb[4 * q + 3]};
          }
        }
      }
      wave_lds_fence();
      WFrag<1> w2f, wc1f;
      w2f.load(nullptr, mats16, M_W2, lane);
      wc1f.load(nullptr, mats16, M_WC1, lane);
      const float a_re = lds[VEC_EMB_F + l * VEC_LAYER_F + V_WRE * EH + lane];
      const float b_att = lds[VEC_EMB_F + l * VEC_LAYER_F + V_COUNT * EH];
      const f32x16 zero16 = {0};
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        if (T >= ntile) continue;
        f32x16 Ai, dAi;
        {
          WFrag<1> wa;
          wa.load(nullptr, mats16, M_WA, lane);
          Ai = wa.mul(hf[T], lds_vec16(vl + V_B1 * EH));
          dAi = wa.mul(dhf[T], zero16);
        }
        f32x16 agg = {0}, dagg = {0};
        float xacc[DIM], dxacc[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) { xacc[k] = 0.f; dxacc[k] = 0.f; }
        const int cbase = col[T] - nodei[T];
        for (int dd = 1; dd < N; ++dd) {
          asm volatile("" ::: "memory");
          int j = nodei[T] + dd;
          j = (j >= N) ? j - N : j;
          const int cj = (col[T] < ncol) ? cbase + j : col[T];
          float df[DIM], ddf[DIM], radial = 0.f, dradial = 0.f, ea = 0.f, dea = 0.f;
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            df[k] = posi[T][k] - poscur[cj * DIM + k];
            ddf[k] = dposi[T][k] - dposcur[cj * DIM + k];
            radial = fmaf(df[k], df[k], radial);
            dradial = fmaf(df[k], ddf[k], dradial);
            const float e0 = p0i[T][k] - pos0[cj * DIM + k], de0 = dp0i[T][k] - dpos0[cj * DIM + k];
            ea = fmaf(e0, e0, ea);
            dea = fmaf(e0, de0, dea);
          }
          dradial *= 2.0f;
          dea *= 2.0f;
          // edge MLP layer 1 and its tangent
          f32x16 z = Ai + lds_vec16(PB + cj * PBS + hh * 16);
          f32x16 dz = dAi + lds_vec16(dPB + cj * PBS + hh * 16);
          z = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re, hh ? ea : radial, z, 0, 0, 0);
          dz = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re, hh ? dea : dradial, dz, 0, 0, 0);
          f32x16 m, dm;
#pragma unroll
          for (int r = 0; r < 16; ++r) { float y, g; silu_dsilu(z[r], y, g); m[r] = y; dm[r] = g * dz[r]; }
          z = w2f.mul(m, lds_vec16(vl + V_B2 * EH));
          dz = w2f.mul(dm, zero16);
#pragma unroll
          for (int r = 0; r < 16; ++r) { float y, g; silu_dsilu(z[r], y, g); m[r] = y; dm[r] = g * dz[r]; }
          if (p.attention) {
            const f32x16 v_watt = lds_vec16(vl + V_WATT * EH);
            const float att = fast_sigmoid(xhalf_sum(dot16(v_watt, m)) + b_att);
            const float datt = att * (1.0f - att) * xhalf_sum(dot16(v_watt, dm));
#pragma unroll
            for (int r = 0; r < 16; ++r) { dm[r] = fmaf(datt, m[r], att * dm[r]); m[r] *= att; }
          }
          if (!last) { agg += m; dagg += dm; }
          // coordinate head and its tangent
          z = wc1f.mul(m, lds_vec16(vl + V_BC1 * EH));
          dz = wc1f.mul(dm, zero16);
          f32x16 c1, dc1;
#pragma unroll
          for (int r = 0; r < 16; ++r) { float y, g; silu_dsilu(z[r], y, g); c1[r] = y; dc1[r] = g * dz[r]; }
          const f32x16 v_wc2 = lds_vec16(vl + V_WC2 * EH);
          float cs = xhalf_sum(dot16(v_wc2, c1)), dcs = xhalf_sum(dot16(v_wc2, dc1));
          if (p.tanh_on) {
            const float th = accurate_tanh(cs);
            dcs = p.coord_scale * fmaf(-th, th, 1.0f) * dcs;
            cs = th * p.coord_scale;
          }
          const float sq = sqrtf(radial + 1e-8f), nrm = sq + 1.0f, dnrm = dradial / (2.0f * sq);
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            const float u = df[k] / nrm;
            const float du = (ddf[k] - u * dnrm) / nrm;
            xacc[k] = fmaf(u, cs, xacc[k]);
            dxacc[k] = fmaf(du, cs, fmaf(u, dcs, dxacc[k]));
          }
        }
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          posi[T][k] += xacc[k];
          dposi[T][k] += dxacc[k];
          if (hh == 0) { posnext[col[T] * DIM + k] = posi[T][k]; dposnext[col[T] * DIM + k] = dposi[T][k]; }
        }
        if (!last) {
          WFrag<1> wn;
          wn.load(nullptr, mats16, M_WN1A, lane);
          f32x16 z = wn.mul(hf[T], lds_vec16(vl + V_BN1 * EH)), dz = wn.mul(dhf[T], zero16);
          wn.load(nullptr, mats16, M_WN1B, lane);
          z = wn.mul(agg, z);
          dz = wn.mul(dagg, dz);
          f32x16 n1, dn1;
#pragma unroll
          for (int r = 0; r < 16; ++r) { float y, g; silu_dsilu(z[r], y, g); n1[r] = y; dn1[r] = g * dz[r]; }
          wn.load(nullptr, mats16, M_WN2, lane);
          hf[T] += wn.mul(n1, lds_vec16(vl + V_BN2 * EH));
          dhf[T] += wn.mul(dn1, zero16);
        }
      }
      wave_lds_fence();
      cur ^= 1;
    }

    // F = mean-free (x_final - x_in) and its tangent; then D = c_s x + c_out F and dD
    float* scr = PB;
    float* dscr = dPB;
    float F[NT][DIM], dF[NT][DIM];
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        F[T][k] = posi[T][k] - p0i[T][k];
        dF[T][k] = dposi[T][k] - dp0i[T][k];
        if (hh == 0) { scr[col[T] * DIM + k] = F[T][k]; dscr[col[T] * DIM + k] = dF[T][k]; }
      }
    wave_lds_fence();
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const int cb = (col[T] < ncol) ? col[T] - nodei[T] : 0;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        float s = 0.f, ds = 0.f;
        for (int q = 0; q < N; ++q) { s += scr[(cb + q) * DIM + k]; ds += dscr[(cb + q) * DIM + k]; }
        F[T][k] -= s / (float)N;
        dF[T][k] -= ds / (float)N;
        const float Dv = fmaf(c_s[T], xin[T][k], c_out[T] * F[T][k]);
        const float dDv = fmaf(c_s[T], vxin[T][k],
                               fmaf(vhv[T], fmaf(dc_s[T], xin[T][k], dc_out[T] * F[T][k]), c_out[T] * dF[T][k]));
        if (valid[T] && hh == 0) {
          const long long gi = (walker0 * N + col[T]) * DIM + k;
          if (p.out) p.out[gi] = Dv;
          if (p.dout) p.dout[gi] = dDv;
          if (p.diag_acc && !p.vx && nodei[T] * DIM + k == p.dir) p.diag_acc[walker0 + col[T] / N] += dDv;
        }
        dF[T][k] = dDv;  // keep for the <x, dD> reduction
      }
    }
    wave_lds_fence();
    if (p.dot_out) {  // per-walker <x, dD>: column partials through LDS, first column of each walker sums
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < DIM; ++k) part = fmaf(xin[T][k], dF[T][k], part);
        if (hh == 0) scr[col[T]] = part;
      }
      wave_lds_fence();
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        if (valid[T] && hh == 0 && nodei[T] == 0) {
          float s = 0.f;
          for (int q = 0; q < N; ++q) s += scr[col[T] + q];
          p.dot_out[(walker0 + col[T] / N) * p.dot_stride + p.dot_off] = s;
        }
      }
      wave_lds_fence();
    }
  }
}

struct JvpShape {
  int n, dim, G, waves, occ;
  void (*kernel)(JvpParams);
  size_t (*lds_bytes)(int);
};
template <int N, int DIM, int G, int WAVES>
static size_t jvp_lds_bytes_of(int L) { return JvpCfg<N, DIM, G, WAVES>::lds_bytes(L); }
#define PITA_JVP_SHAPE(N, DIM, G, WAVES, OCC) \
  JvpShape { N, DIM, G, WAVES, OCC, egnn_jvp_kernel<N, DIM, G, WAVES, OCC>, jvp_lds_bytes_of<N, DIM, G, WAVES> }
static const JvpShape kJvpShapes[] = {
    PITA_JVP_SHAPE(4, 2, 8, 4, 1),
    PITA_JVP_SHAPE(13, 3, 7, 4, 1),
    PITA_JVP_SHAPE(22, 3, 4, 4, 1),
    PITA_JVP_SHAPE(55, 3, 1, 4, 1),
    // LJ13: one column tile per wave (2 walkers, 26 of 32 columns) at two waves per SIMD measured 8 % faster than
    // three dense tiles at one wave per SIMD (3.28 vs 3.57 ms per launch at 65 536 walkers); PITA_JVP_ALT=0 selects
    // the dense shape above
    PITA_JVP_SHAPE(13, 3, 2, 4, 2),
};

}  // namespace pita

using namespace pita;

extern "C" int pita_egnn_jvp(pita_egnn_t* net, const float* h, const float* x, const float* beta, const float* vx,
                             int dir, const float* vh, float* out, float* dout, float* dot_out, int64_t dot_stride,
                             int64_t dot_off, float* diag_acc, int64_t B, void* stream) {
  PITA_REQUIRE(net && B >= 0, "pita_egnn_jvp: bad argument");
  if (B == 0) return PITA_OK;
  PitaDeviceGuard guard(net->device);
  PITA_REQUIRE(h && x && (dout || dot_out || diag_acc || out), "pita_egnn_jvp: null argument");
  PITA_REQUIRE(beta || net->cfg.in_node_nf == 1, "pita_egnn_jvp: beta required for in_node_nf=2");
  const int D = net->cfg.n_particles * net->cfg.n_dim;
  PITA_REQUIRE(vx || (dir >= -1 && dir < D), "pita_egnn_jvp: dir out of range");
  const JvpShape* s = nullptr;
  static const int alt = getenv("PITA_JVP_ALT") ? atoi(getenv("PITA_JVP_ALT")) : 1;
  for (const auto& c : kJvpShapes)
    if (c.n == net->cfg.n_particles && c.dim == net->cfg.n_dim && (c.occ == 1 || alt)) s = &c;
  if (!s) return fail(PITA_EUNSUPPORTED, "pita_egnn_jvp: no kernel for this particle system");
  JvpParams p{};
  p.mats16 = net->d_mats16; p.vecs = net->d_vecs; p.n_layers = net->cfg.n_layers; p.in_nf = net->cfg.in_node_nf;
  p.attention = net->cfg.attention; p.tanh_on = net->cfg.tanh; p.feature_layout = net->cfg.feature_layout;
  p.coord_scale = net->cfg.coords_range / (float)net->cfg.n_layers;
  p.B = B; p.h = h; p.x = x; p.beta = beta; p.vx = vx; p.dir = dir; p.vh = vh; p.out = out; p.dout = dout;
  p.dot_out = dot_out; p.dot_stride = dot_stride > 0 ? dot_stride : 1; p.dot_off = dot_off; p.diag_acc = diag_acc;
  const size_t lds = s->lds_bytes(p.n_layers);
  PITA_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(s->kernel), lds));
  const long long ngroups = (B + s->G - 1) / s->G;
  long long want = (ngroups + s->waves - 1) / s->waves;
  const long long cap = (long long)net->n_cu * s->occ;  // occ 4-wave blocks per CU
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  hipLaunchKernelGGL(s->kernel, dim3(grid), dim3(s->waves * 64), lds, (hipStream_t)stream, p);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}
