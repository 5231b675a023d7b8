// Exact divergence of the EDM-preconditioned EGNN denoiser for gfx950: K unit tangent directions per launch sharing
// ONE primal evaluation.
//
// The debiased (Feynman-Kac) weights of the reference sampler (pita/src/models/components/sdes.py:151-239) need
// div_x s_theta = (trace(J_x D) - dim) / h, which the reference computes exactly with vmap(jacrev)
// (pita/src/models/components/utils.py:30-51).  trace(J_x D) = sum_d (J_x D e_d)_d takes dim forward-mode passes;
// pita_egnn_jvp spends half of every pass recomputing the primal network.  This kernel pushes K directions
// e_{dir0}, ..., e_{dir0+K-1} through the network together: per edge the primal MLP (activations and their derivative
// factors) is evaluated once and the K tangent chains -- independent MFMA chains, so they also give the single wave
// per SIMD the instruction-level parallelism it lacks in the one-direction kernel -- reuse it.  (Only 256 of the
// wave's 512 registers are visible to VALU instructions, so per-node tangent state that the edge loop merely reads --
// Wa dh_i -- is parked in LDS next to the partner tables.)  Output:
// diag_acc[b] += sum_k (J_x D e_{dir0+k})_{dir0+k}.
//
// Mapping: as egnn_jvp_kernel.hip (wave = G walkers = dense 32-column tiles, lane = column x 16 features, exact 3-way
// bf16 split on the matrix pipe, tangents in the lane/register position of their primals); one wave per SIMD.
#include <type_traits>

#include "egnn_common.h"

namespace pita {

struct DivParams {
  const unsigned* mats16;   // bf16 three-piece fragments (this kernel; the transposed matrices of the fast kernel)
  const unsigned* mats16h;  // f16 two-piece fragments (fast kernel)
  const float* vecs;
  const float* vecs_h;      // vectors with the f16-path scale factors folded in (fast kernel)
  const float* vecs_div;    // [L][3][32] k-step weights in fragment order (fast kernel)
  float* cache;             // fast kernel: nullable, receives the per-edge primal factors the tangent chains consume
                            // (DivCache layout); egnn_div_tangent_kernel: reads them
  int* mark;                // [B] fast kernel: 1 = this launch's contribution of the walker was non-finite and NOT added;
                            // this kernel with repair != 0: recompute and add exactly the marked walkers
  int repair;
  int nchunk;               // repair pass: > 1 = the ndir directions are processed in this many chunks of K in one launch
  int* bad_flag;            // one word behind the marks: the launch that marked a walker leaves its sequence number
  int bad_seq;              // here, and the repair launches behind a launch that marked nobody return at once
  int n_layers, in_nf, attention, tanh_on, feature_layout;
  float coord_scale;
  long long B;
  const float* h;     // [B] sigma^2
  const float* x;     // [B, D]
  const float* beta;  // [B] or null
  int dir0, ndir;     // unit directions dir0 .. dir0 + ndir - 1 (ndir <= K; block-shared tangent kernel: <= NW K)
  long long cache_waves;  // block-shared tangent kernel: waves of the launch that wrote the cache (its group layout)
  float* diag_acc;    // [B] += sum_k dD_k[b, dir0 + k]
  float* out;         // [B, D] denoiser D of the primal (nullable: the caller asks for it with the first launch only)
  int no_mean;        // != 0: the launch is part of a FULL trace (pita_egnn_jacobian_trace) and leaves out the mean-free
                      // projection's share of each diagonal entry, -(1/N) sum_i d pos^L_{i,k} / d x_{i0,k}: summed over all
                      // N dim directions these shares cancel exactly -- the network sees positions through differences
                      // only, so moving every particle along k moves every pos^L_{i,k} by the same amount,
                      // sum_{i0} d pos^L_{i,k} / d x_{i0,k} = 1, and sum_d (sum_i d pos^L_{i,k_d} - 1) = dim N - N dim = 0.
                      // What is left of a direction's term needs the tangent of ONE output coordinate only.
};

template <int N, int DIM, int G, int WAVES, int K, int DAL = 1>  // DAL: Wa dh_i tables in LDS (1) or in registers (0)
struct DivCfg {
  static constexpr int NCOL = G * N;
  static constexpr int NT = (NCOL + 31) / 32;
  static constexpr int NCOLP = NT * 32;
  static constexpr int PB_F = NCOLP * PBS;
  static constexpr int POS_F = NCOLP * DIM;
  // PB, dPB[K], dA[K]; pos[2], pos0; dpos[K][2], dpos0[K]
  static constexpr int WAVE_F = (1 + K + DAL * K) * PB_F + 3 * POS_F + 3 * K * POS_F;
  static __host__ __device__ constexpr int vec_f(int L) { return ((VEC_EMB_F + L * VEC_LAYER_F) + 3) & ~3; }
  static __host__ __device__ constexpr size_t lds_bytes(int L) {
    return sizeof(float) * (size_t)(vec_f(L) + WAVES * WAVE_F + L * VEC_DIV_F);
  }
};

__device__ __forceinline__ void silu_dsilu2(float v, float& y, float& g) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v));
  y = v * s;
  g = s * fmaf(v * (1.0f / SILU_PRESCALE), 1.0f - s, 1.0f);
}

template <int N, int DIM, int G, int WAVES, int K>
__global__ void __launch_bounds__(WAVES * 64, 1) egnn_div_kernel(DivParams p) {
  using C = DivCfg<N, DIM, G, WAVES, K>;
  constexpr int NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (p.repair && p.bad_flag && *p.bad_flag != p.bad_seq) return;  // the fast launch marked nobody
  const int L = p.n_layers;
  const int vec_f = C::vec_f(L);
  for (int i = threadIdx.x; i < VEC_EMB_F + L * VEC_LAYER_F; i += WAVES * 64) lds[i] = p.vecs[i];
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane & 31, hh = lane >> 5;
  float* PB = lds + vec_f + wave * C::WAVE_F;
  float* dPB = PB + C::PB_F;                       // [K][PB_F]
  float* dA = dPB + K * C::PB_F;                   // [K][PB_F]  Wa dh_i (own node): parked in LDS, not in registers
  float* posb = dA + K * C::PB_F;                  // [2][POS_F]
  float* pos0 = posb + 2 * C::POS_F;
  float* dposb = pos0 + C::POS_F;                  // [K][2][POS_F]
  float* dpos0 = dposb + 2 * K * C::POS_F;         // [K][POS_F]
  const float* vemb = lds;
  const f32x16 zero16 = {0};

  const long long total_waves = (long long)gridDim.x * WAVES;
  const long long quota = (p.B + total_waves - 1) / total_waves;
  const long long wbeg = ((long long)blockIdx.x * WAVES + wave) * quota;
  const long long wend = (wbeg + quota < p.B) ? wbeg + quota : p.B;
  for (long long walker0 = wbeg; walker0 < wend; walker0 += G) {
    const int nwalk = (int)((wend - walker0) < G ? (wend - walker0) : G);
    const int ncol = nwalk * N;
    const int ntile = (ncol + 31) >> 5;
    int col[NT], nodei[NT];
    bool valid[NT], mine[NT];
    float c_s[NT], c_in[NT], c_out[NT];
    float posi[NT][DIM], p0i[NT][DIM], dposi[NT][K][DIM], dp0i[NT][K][DIM];
    f32x16 hf[NT], dhf[NT][K];
    if (p.repair) {  // second launch behind the f16 kernel: only groups with a marked walker are recomputed
      bool any = false;
      for (int w = 0; w < nwalk; ++w) any = any || p.mark[walker0 + w] != 0;
      if (!any) continue;
    }
    // repair pass with `nchunk` > 1: the directions dir0 .. dir0 + ndir - 1 in chunks of K inside ONE launch (one launch
    // per tangent-only launch instead of one per K directions: these launches return at once almost always)
    const int nch = (p.repair && p.nchunk > 1) ? p.nchunk : 1;
    for (int ch = 0; ch < nch; ++ch) {
    const int dir0 = p.dir0 + ch * K;
    const int ndir = nch > 1 ? ((p.ndir - ch * K) < K ? (p.ndir - ch * K) : K) : p.ndir;
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      col[T] = T * 32 + cl;
      const int w = col[T] / N;
      nodei[T] = col[T] - w * N;
      valid[T] = col[T] < ncol;
      mine[T] = valid[T] && (!p.repair || p.mark[walker0 + w] != 0);
      const long long wid = valid[T] ? walker0 + w : p.B - 1;
      const float hv = p.h[wid];
      const float bet = p.beta ? p.beta[wid] : 0.f;
      const float op = 1.0f + hv, rs = 1.0f / sqrtf(op);
      c_s[T] = 1.0f / op;
      c_in[T] = rs;
      c_out[T] = sqrtf(hv) * rs;
      const float tfeat = 0.125f * logf(hv);
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        const long long gi = (walker0 * N + col[T]) * DIM + k;
        const float xv = valid[T] ? p.x[gi] : 0.f;
        posi[T][k] = c_in[T] * xv;
        p0i[T][k] = posi[T][k];
        if (hh == 0) {
          pos0[col[T] * DIM + k] = posi[T][k];
          posb[col[T] * DIM + k] = posi[T][k];
        }
#pragma unroll
        for (int q = 0; q < K; ++q) {
          const float v = (valid[T] && q < ndir && nodei[T] * DIM + k == dir0 + q) ? c_in[T] : 0.f;  // d(c_in x)
          dposi[T][q][k] = v;
          dp0i[T][q][k] = v;
          if (hh == 0) {
            dpos0[q * C::POS_F + col[T] * DIM + k] = v;
            dposb[(2 * q) * C::POS_F + col[T] * DIM + k] = v;
          }
        }
      }
      float a0, a1;
      if (p.in_nf == 1) { a0 = tfeat; a1 = 0.f; }
      else if (p.feature_layout == 0) {
        a0 = (2 * nodei[T] < N) ? tfeat : bet;
        a1 = (2 * nodei[T] + 1 < N) ? tfeat : bet;
      } else { a0 = tfeat; a1 = bet; }
      const f32x16 w0 = lds_vec16(vemb + hh * 16), w1 = lds_vec16(vemb + 32 + hh * 16), eb = lds_vec16(vemb + 64 + hh * 16);
#pragma unroll
      for (int r = 0; r < 16; ++r) hf[T][r] = fmaf(w0[r], a0, fmaf(w1[r], a1, eb[r]));
#pragma unroll
      for (int q = 0; q < K; ++q) dhf[T][q] = zero16;  // node features do not depend on x
    }
    wave_lds_fence();

    int cur = 0;
    for (int l = 0; l < L; ++l) {
      const unsigned* mats16 = p.mats16 + (size_t)l * M_COUNT * MAT_W;
      const float* vl = lds + VEC_EMB_F + l * VEC_LAYER_F + hh * 16;
      const bool last = (l == L - 1);
      const float* poscur = posb + cur * C::POS_F;
      {
        WFrag<1> wb;
        wb.load(nullptr, mats16, M_WB, lane);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          if (T >= ntile) continue;
          const f32x16 pb = wb.mul(hf[T], zero16);
          f32x4* dst = reinterpret_cast<f32x4*>(PB + col[T] * PBS + hh * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[q] = f32x4{pb[4 * q], pb[4 * q + 1], pb[4 * q + 2], pb[4 * q + 3]};
#pragma unroll
          for (int d = 0; d < K; ++d) {
            f32x16 dpb = zero16;
            if (l > 0) dpb = wb.mul(dhf[T][d], zero16);  // dh = 0 in the first layer
            f32x4* ddst = reinterpret_cast<f32x4*>(dPB + d * C::PB_F + col[T] * PBS + hh * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) ddst[q] = f32x4{dpb[4 * q], dpb[4 * q + 1], dpb[4 * q + 2], dpb[4 * q + 3]};
          }
        }
      }
      wave_lds_fence();
      WFrag<1> w2f, wc1f;
      w2f.load(nullptr, mats16, M_W2, lane);
      wc1f.load(nullptr, mats16, M_WC1, lane);
      const float a_re = lds[VEC_EMB_F + l * VEC_LAYER_F + V_WRE * EH + lane];
      const float b_att = lds[VEC_EMB_F + l * VEC_LAYER_F + V_COUNT * EH];
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        if (T >= ntile) continue;
        f32x16 Ai;
        {
          WFrag<1> wa;
          wa.load(nullptr, mats16, M_WA, lane);
          Ai = wa.mul(hf[T], lds_vec16(vl + V_B1 * EH));
          if (l > 0) {
#pragma unroll
            for (int d = 0; d < K; ++d) {
              const f32x16 da = wa.mul(dhf[T][d], zero16);
              f32x4* dst = reinterpret_cast<f32x4*>(dA + d * C::PB_F + col[T] * PBS + hh * 16);
#pragma unroll
              for (int q = 0; q < 4; ++q) dst[q] = f32x4{da[4 * q], da[4 * q + 1], da[4 * q + 2], da[4 * q + 3]};
            }
          }
        }
        wave_lds_fence();
        f32x16 agg = {0}, dagg[K];
        float xacc[DIM], dxacc[K][DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) xacc[k] = 0.f;
#pragma unroll
        for (int d = 0; d < K; ++d) {
          dagg[d] = zero16;
#pragma unroll
          for (int k = 0; k < DIM; ++k) dxacc[d][k] = 0.f;
        }
        const int cbase = col[T] - nodei[T];
        for (int dd = 1; dd < N; ++dd) {
          asm volatile("" ::: "memory");
          int j = nodei[T] + dd;
          j = (j >= N) ? j - N : j;
          const int cj = (col[T] < ncol) ? cbase + j : col[T];
          float df[DIM], e0[DIM], radial = 0.f, ea = 0.f;
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            df[k] = posi[T][k] - poscur[cj * DIM + k];
            radial = fmaf(df[k], df[k], radial);
            e0[k] = p0i[T][k] - pos0[cj * DIM + k];
            ea = fmaf(e0[k], e0[k], ea);
          }
          // ---- primal edge MLP with derivative factors (once for all K directions)
          f32x16 z = Ai + lds_vec16(PB + cj * PBS + hh * 16);
          z = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re, hh ? ea : radial, z, 0, 0, 0);
          f32x16 g1, g2, gc, m2, m;
#pragma unroll
          for (int r = 0; r < 16; ++r) { float y, g; silu_dsilu2(z[r], y, g); z[r] = y; g1[r] = g; }
          z = w2f.mul(z, lds_vec16(vl + V_B2 * EH));
#pragma unroll
          for (int r = 0; r < 16; ++r) { float y, g; silu_dsilu2(z[r], y, g); m2[r] = y; g2[r] = g; }
          float att = 1.0f;
          m = m2;
          const f32x16 v_watt = lds_vec16(vl + V_WATT * EH);
          if (p.attention) {
            att = fast_sigmoid(xhalf_sum(dot16(v_watt, m2)) + b_att);
            m *= att;
          }
          if (!last) agg += m;
          z = wc1f.mul(m, lds_vec16(vl + V_BC1 * EH));
#pragma unroll
          for (int r = 0; r < 16; ++r) { float y, g; silu_dsilu2(z[r], y, g); z[r] = y; gc[r] = g; }
          const f32x16 v_wc2 = lds_vec16(vl + V_WC2 * EH);
          float cs = xhalf_sum(dot16(v_wc2, z)), dcs_f = 1.0f;
          if (p.tanh_on) {
            const float th = accurate_tanh(cs);
            dcs_f = p.coord_scale * fmaf(-th, th, 1.0f);
            cs = th * p.coord_scale;
          }
          const float sq = sqrtf(radial + 1e-8f), inv = 1.0f / (sq + 1.0f), hsq = 0.5f / sq;
          float u[DIM];
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            u[k] = df[k] * inv;
            xacc[k] = fmaf(u[k], cs, xacc[k]);
          }
          // ---- K tangent chains
#pragma unroll
          for (int d = 0; d < K; ++d) {
            const float* dposcur = dposb + (2 * d + cur) * C::POS_F;
            float ddf[DIM], dradial = 0.f, dea = 0.f;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              ddf[k] = dposi[T][d][k] - dposcur[cj * DIM + k];
              dradial = fmaf(df[k], ddf[k], dradial);
              dea = fmaf(e0[k], dp0i[T][d][k] - dpos0[d * C::POS_F + cj * DIM + k], dea);
            }
            dradial *= 2.0f;
            dea *= 2.0f;
            f32x16 dz = zero16;  // dh = 0 in the first layer
            if (l > 0)
              dz = lds_vec16(dA + d * C::PB_F + col[T] * PBS + hh * 16) + lds_vec16(dPB + d * C::PB_F + cj * PBS + hh * 16);
            dz = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re, hh ? dea : dradial, dz, 0, 0, 0);
            dz *= g1;
            dz = w2f.mul(dz, zero16);
            dz *= g2;  // dm2
            if (p.attention) {
              const float datt = att * (1.0f - att) * xhalf_sum(dot16(v_watt, dz));
#pragma unroll
              for (int r = 0; r < 16; ++r) dz[r] = fmaf(datt, m2[r], att * dz[r]);
            }
            if (!last) dagg[d] += dz;
            dz = wc1f.mul(dz, zero16);
            dz *= gc;
            const float dcs = dcs_f * xhalf_sum(dot16(v_wc2, dz));
            const float dnrm = dradial * hsq;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              const float du = (ddf[k] - u[k] * dnrm) * inv;
              dxacc[d][k] = fmaf(du, cs, fmaf(u[k], dcs, dxacc[d][k]));
            }
          }
        }
        float* posnext = posb + (cur ^ 1) * C::POS_F;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          posi[T][k] += xacc[k];
          if (hh == 0) posnext[col[T] * DIM + k] = posi[T][k];
#pragma unroll
          for (int d = 0; d < K; ++d) {
            dposi[T][d][k] += dxacc[d][k];
            if (hh == 0) dposb[(2 * d + (cur ^ 1)) * C::POS_F + col[T] * DIM + k] = dposi[T][d][k];
          }
        }
        if (!last) {
          WFrag<1> wn;
          wn.load(nullptr, mats16, M_WN1A, lane);
          f32x16 zn = wn.mul(hf[T], lds_vec16(vl + V_BN1 * EH));
          f32x16 dzn[K];
#pragma unroll
          for (int d = 0; d < K; ++d) dzn[d] = (l > 0) ? wn.mul(dhf[T][d], zero16) : zero16;
          wn.load(nullptr, mats16, M_WN1B, lane);
          zn = wn.mul(agg, zn);
#pragma unroll
          for (int d = 0; d < K; ++d) dzn[d] = wn.mul(dagg[d], dzn[d]);
          f32x16 gn;
#pragma unroll
          for (int r = 0; r < 16; ++r) { float y, g; silu_dsilu2(zn[r], y, g); zn[r] = y; gn[r] = g; }
          wn.load(nullptr, mats16, M_WN2, lane);
          hf[T] += wn.mul(zn, lds_vec16(vl + V_BN2 * EH));
#pragma unroll
          for (int d = 0; d < K; ++d) {
            dzn[d] *= gn;
            dhf[T][d] += wn.mul(dzn[d], zero16);
          }
        }
      }
      wave_lds_fence();
      cur ^= 1;
    }

    // dD = c_s e_d + c_out (dF - mean dF);  only the component [dir] of each direction is needed
    float* dscr = dPB;  // [K][NCOLP*DIM] scratch (dPB is free now)
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int d = 0; d < K; ++d)
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          dposi[T][d][k] -= dp0i[T][d][k];  // dF
          if (hh == 0) dscr[d * C::POS_F + col[T] * DIM + k] = dposi[T][d][k];
        }
    wave_lds_fence();
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const int cb = (col[T] < ncol) ? col[T] - nodei[T] : 0;
#pragma unroll
      for (int d = 0; d < K; ++d)
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          if (!(mine[T] && hh == 0 && d < ndir && nodei[T] * DIM + k == dir0 + d)) continue;
          float ds = 0.f;
          for (int q = 0; q < N; ++q) ds += dscr[d * C::POS_F + (cb + q) * DIM + k];
          const float dF = p.no_mean ? dposi[T][d][k] : dposi[T][d][k] - ds / (float)N;
          // the K owners of one walker are different lanes: the read-modify-writes below would race
          atomicAdd(&p.diag_acc[walker0 + col[T] / N], fmaf(c_out[T], dF, c_s[T]));
        }
    }
    wave_lds_fence();
    if (p.out && ch == 0) {  // primal denoiser D = c_s x + c_out (F - mean F), F = pos^L - pos^0 (x = pos^0 / c_in)
      float* scr = PB;
#pragma unroll
      for (int T = 0; T < NT; ++T)
        if (hh == 0) {
#pragma unroll
          for (int k = 0; k < DIM; ++k) scr[col[T] * DIM + k] = posi[T][k] - p0i[T][k];
        }
      wave_lds_fence();
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        if (!(mine[T] && hh == 0)) continue;
        const int cb = col[T] - nodei[T];
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          float sum = 0.f;
          for (int q = 0; q < N; ++q) sum += scr[(cb + q) * DIM + k];
          const float F = (posi[T][k] - p0i[T][k]) - sum / (float)N;
          const long long gi = (walker0 * N + col[T]) * DIM + k;
          p.out[gi] = fmaf(c_s[T], p.x[gi], c_out[T] * F);
        }
      }
      wave_lds_fence();
    }
    }  // chunks of directions
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Primal cache.  The primal network is the same for all D directions of a trace; the fast kernel recomputes it in each
// of its D / K launches (61 % of its vector work).  With `cache` set, the FIRST launch stores what the tangent chains
// consume from the primal -- per layer the positions entering it, per node the scaled SiLU derivative of the node model,
// per edge one (first / last layer) or four (middle layers) 16-float-per-lane vectors and a few scalars -- and
// the tangent-only kernels (wave-owned: K directions per launch; block-shared: NW K) run the remaining directions from
// that cache without touching the primal:
// ≈180 KB per walker (12 GB at 65 536 walkers: HBM capacity is what this GPU has to spare), streamed once per launch in
// 1 KB wave-loads.  Layout per walker group g (a wave's G walkers) and layer l, in floats:
//   [pos: POSF] then per tile T: [edge dd = 1..N-1: nv(l) x 1024 vectors | 8 x 64 scalars] [gn: 1024]
// scalar slots: 0 cs, 1 dcs_f (incl. the tangent scales), 2 att, 3 att (1 - att), 4 vcdmu | qr, 5 qe.  Middle-layer
// records (round 4): vector 1 is att (g2 o .) -- the gate folded into the cached SiLU derivative --, slot 3 is 1 - att,
// and slots 4..7 hold the edge's geometry as [column 32][df[3], inv, e0[3], hsq] so that no tangent wave recomputes it;
// first / last layer records keep it as [column 32][df[3], inv] in slots 2-3 and [column 32][e0[3], hsq] in slots 6-7.  The position
// block carries, behind the NT x 32 x DIM positions, c_skip and c_out c_in of every column (layer 0's block is the one
// read).  Every item is a whole number of 1 KB chunks and a group's items lie in the order the tangent sweep consumes
// them, so the block-shared tangent kernel can stream a group as plain 1 KB pieces.
template <int N, int DIM, int NT>
struct DivCache {
  static constexpr int POSX = NT * 32 * DIM;                          // offset of the per-column c_skip | c_out c_in
  static constexpr int POSF = ((POSX + 2 * NT * 32 + 255) / 256) * 256;
  static __host__ __device__ int nvec(int l, int L) { return (l == 0 || l == L - 1) ? 1 : 4; }
  static __host__ __device__ size_t edge_f(int l, int L) { return (size_t)nvec(l, L) * 1024 + 512; }
  static __host__ __device__ size_t gn_off(int l, int L) { return (size_t)(N - 1) * edge_f(l, L); }
  static __host__ __device__ size_t tile_f(int l, int L) { return 1024 + (size_t)(N - 1) * edge_f(l, L); }
  static __host__ __device__ size_t layer_f(int l, int L) { return POSF + (size_t)NT * tile_f(l, L); }
  static __host__ __device__ size_t group_f(int L) {
    size_t t = 0;
    for (int l = 0; l < L; ++l) t += layer_f(l, L);
    return t;
  }
  static __host__ __device__ size_t layer_off(int l, int L) {
    size_t t = 0;
    for (int q = 0; q < l; ++q) t += layer_f(q, L);
    return t;
  }
};
// (non-temporal stores: 12 GB written once and read back by later launches, far beyond the L2 / MALL; 18.45 -> 18.29 ms per
// trace on one box)
__device__ __forceinline__ void cache_store16(float* base, int lane, const f32x16& v) {
  f32x4* d = reinterpret_cast<f32x4*>(base) + lane;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    __builtin_nontemporal_store(f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}, d + q * 64);
}
__device__ __forceinline__ f32x16 cache_load16(const float* base, int lane) {
  const f32x4* d = reinterpret_cast<const f32x4*>(base) + lane;
  f32x16 r;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = d[q * 64];
    r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
  }
  return r;
}

// ------------------------------------------------------------------------------------------------------------------
// Fast variant (handles built with precision 2).  Same mapping and LDS tables; two changes in the arithmetic:
//
// (1) dense layers on the f16 two-piece path (egnn_common.h, PREC 2).  Feature tangents (dh, dz, dm, dagg and the LDS
//     tables dPB / dA) travel scaled by DIV_ST = 32 so that their low f16 pieces stay normal; position tangents keep
//     their true scale and start from 1 (the factor c_in of d(c_in x)/dx is applied to the final sum instead).
//     Out-of-range operands end as NaN in the walker's trace term: the term is then NOT added, the walker is marked, and
//     the launch wrapper re-runs the bf16x3 kernel above for the marked walkers (same protocol as egnn_kernel.hip).
// (2) adjoints inside the edge, computed once per edge and shared by the K directions, remove tangent GEMMs:
//     * coordinate head: w_c2 . (gc o Wc1 dm) = vc . dm with vc = Wc1^T (gc o w_c2): ONE transposed GEMM per edge
//       instead of one Wc1 GEMM per direction;
//     * last layer (aggregate dead, only the scalar head is consumed):  vc . dm = qv . dz1  with
//       qv = g1 o W2^T (g2 o (att vc + att (1 - att) (vc . m2) w_att)): no tangent GEMM at all, dz1 = dA_i + dPB_j + k-step
//       is only dotted with qv;
//     * first layer (dh = 0, pos = pos0, so radial = edge_attr): every direction's dz1 is the SAME vector
//       g1 o (w_r + w_e) times the scalar dradial_d: one W2 GEMM per edge for all directions.
//     Per edge over three layers: 6 primal + 4 adjoint + 1 + K tangent GEMMs instead of 6 + 6 K.
__device__ __forceinline__ void silu_dsilu16(const f32x16& vin, float pre, f32x16& y, f32x16& g) {
  // staged: all exponentials, then all reciprocals, then the products (no consumer directly behind its transcendental)
  f32x16 v = vin * pre, e, sg;
#pragma unroll
  for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(v[r]);
#pragma unroll
  for (int r = 0; r < 16; ++r) sg[r] = __builtin_amdgcn_rcpf(1.0f + e[r]);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float yy = v[r] * sg[r];
    y[r] = yy;
    g[r] = fmaf(yy * (1.0f / SILU_PRESCALE), e[r] * sg[r], sg[r]);  // s (1 + z (1 - s)), z = v / kS, 1 - s = e s
  }
}

template <int N, int DIM, int G, int WAVES, int K, int DAL, int OCC = 1>
__global__ void __launch_bounds__(WAVES * 64, OCC) egnn_div_fast_kernel(DivParams p) {
  using C = DivCfg<N, DIM, G, WAVES, K, DAL>;
  constexpr int NT = C::NT;
  constexpr int KA = K > 0 ? K : 1;  // array extents (K = 0, the cache writer: no direction, the loops over d vanish)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = p.n_layers;
  const int vec_f = C::vec_f(L);
  for (int i = threadIdx.x; i < VEC_EMB_F + L * VEC_LAYER_F; i += WAVES * 64) lds[i] = p.vecs_h[i];
  float* vdiv = lds + vec_f + WAVES * C::WAVE_F;   // [L][3][32] behind the per-wave tables
  for (int i = threadIdx.x; i < L * VEC_DIV_F; i += WAVES * 64) vdiv[i] = p.vecs_div[i];
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane & 31, hh = lane >> 5;
  float* PB = lds + vec_f + wave * C::WAVE_F;
  float* dPB = PB + C::PB_F;                       // [KA][PB_F]  DIV_ST x Wb dh_j
  float* dA = dPB + K * C::PB_F;                   // [KA][PB_F]  DIV_ST x Wa dh_i (DAL = 1; else registers)
  float* posb = dA + DAL * K * C::PB_F;            // [2][POS_F]
  float* pos0 = posb + 2 * C::POS_F;
  float* dposb = pos0 + C::POS_F;                  // [KA][2][POS_F]
  float* dpos0 = dposb + 2 * K * C::POS_F;         // [KA][POS_F]
  const float* vemb = lds;
  const f32x16 zero16 = {0};

  const long long total_waves = (long long)gridDim.x * WAVES;
  const long long quota = (p.B + total_waves - 1) / total_waves;
  const long long wbeg = ((long long)blockIdx.x * WAVES + wave) * quota;
  const long long wend = (wbeg + quota < p.B) ? wbeg + quota : p.B;
  using CA = DivCache<N, DIM, NT>;
  const long long groups_per_wave = (quota + G - 1) / G;
  for (long long walker0 = wbeg; walker0 < wend; walker0 += G) {
    const int nwalk = (int)((wend - walker0) < G ? (wend - walker0) : G);
    const int ncol = nwalk * N;
    const int ntile = (ncol + 31) >> 5;
    float* cgrp = p.cache ? p.cache + (size_t)(((long long)blockIdx.x * WAVES + wave) * groups_per_wave +
                                                 (walker0 - wbeg) / G) * CA::group_f(L) : nullptr;
    int col[NT], nodei[NT];
    bool valid[NT];
    float c_s[NT], c_in[NT], c_out[NT];
    float posi[NT][DIM], p0i[NT][DIM], dposi[NT][KA][DIM], dp0i[NT][KA][DIM];
    f32x16 hf[NT], dhf[NT][KA];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      col[T] = T * 32 + cl;
      const int w = col[T] / N;
      nodei[T] = col[T] - w * N;
      valid[T] = col[T] < ncol;
      const long long wid = valid[T] ? walker0 + w : p.B - 1;
      const float hv = p.h[wid];
      const float bet = p.beta ? p.beta[wid] : 0.f;
      const float op = 1.0f + hv, rs = 1.0f / sqrtf(op);
      c_s[T] = 1.0f / op;
      c_in[T] = rs;
      c_out[T] = sqrtf(hv) * rs;
      const float tfeat = 0.125f * logf(hv);
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        const long long gi = (walker0 * N + col[T]) * DIM + k;
        const float xv = valid[T] ? p.x[gi] : 0.f;
        posi[T][k] = c_in[T] * xv;
        p0i[T][k] = posi[T][k];
        if (hh == 0) {
          pos0[col[T] * DIM + k] = posi[T][k];
          posb[col[T] * DIM + k] = posi[T][k];
        }
#pragma unroll
        for (int q = 0; q < K; ++q) {
          const float v = (valid[T] && q < p.ndir && nodei[T] * DIM + k == p.dir0 + q) ? 1.0f : 0.f;  // unit direction
          dposi[T][q][k] = v;
          dp0i[T][q][k] = v;
          if (hh == 0) {
            dpos0[q * C::POS_F + col[T] * DIM + k] = v;
            dposb[(2 * q) * C::POS_F + col[T] * DIM + k] = v;
          }
        }
      }
      float a0, a1;
      if (p.in_nf == 1) { a0 = tfeat; a1 = 0.f; }
      else if (p.feature_layout == 0) {
        a0 = (2 * nodei[T] < N) ? tfeat : bet;
        a1 = (2 * nodei[T] + 1 < N) ? tfeat : bet;
      } else { a0 = tfeat; a1 = bet; }
      const f32x16 w0 = lds_vec16(vemb + hh * 16), w1 = lds_vec16(vemb + 32 + hh * 16), eb = lds_vec16(vemb + 64 + hh * 16);
#pragma unroll
      for (int r = 0; r < 16; ++r) hf[T][r] = fmaf(w0[r], a0, fmaf(w1[r], a1, eb[r]));
#pragma unroll
      for (int q = 0; q < K; ++q) dhf[T][q] = zero16;
    }
    wave_lds_fence();

    int cur = 0;
    for (int l = 0; l < L; ++l) {
      const unsigned* mats16h = p.mats16h + (size_t)l * M_COUNT * MAT_WH;  // f16x2: the forward matrices
      const float* vl = lds + VEC_EMB_F + l * VEC_LAYER_F + hh * 16;
      const bool first = (l == 0), last = (l == L - 1) && !first;
      const float* poscur = posb + cur * C::POS_F;
      float* clay = cgrp ? cgrp + CA::layer_off(l, L) : nullptr;
      if (clay) {
        for (int i = lane; i < C::POS_F; i += 64) clay[i] = poscur[i];
        if (first && hh == 0)
#pragma unroll
          for (int T = 0; T < NT; ++T) {
            clay[CA::POSX + col[T]] = c_s[T];
            clay[CA::POSX + NT * 32 + col[T]] = c_out[T] * c_in[T];
          }
      }
      {
        WFrag<2> wb;
        wb.load(nullptr, mats16h, M_WB, lane);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          if (T >= ntile) continue;
          f32x16 pb = wb.mul(hf[T], zero16);
          pb *= F16_UNSCALE;
          f32x4* dst = reinterpret_cast<f32x4*>(PB + col[T] * PBS + hh * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[q] = f32x4{pb[4 * q], pb[4 * q + 1], pb[4 * q + 2], pb[4 * q + 3]};
          if (!first) {
#pragma unroll
            for (int d = 0; d < K; ++d) {
              f32x16 dpb = wb.mul(dhf[T][d], zero16);
              dpb *= F16_UNSCALE;
              f32x4* ddst = reinterpret_cast<f32x4*>(dPB + d * C::PB_F + col[T] * PBS + hh * 16);
#pragma unroll
              for (int q = 0; q < 4; ++q) ddst[q] = f32x4{dpb[4 * q], dpb[4 * q + 1], dpb[4 * q + 2], dpb[4 * q + 3]};
            }
          }
        }
      }
      wave_lds_fence();
      WFrag<2> w2f, wc1f;
      w2f.load(nullptr, mats16h, M_W2, lane);
      wc1f.load(nullptr, mats16h, M_WC1, lane);
      WFrag<2> wc1t, w2t;  // adjoints: transposed matrices
      wc1t.load(nullptr, mats16h, M_WC1T, lane);
      if (last) w2t.load(nullptr, mats16h, M_W2T, lane);
      const float a_re = lds[VEC_EMB_F + l * VEC_LAYER_F + V_WRE * EH + lane];
      const float b_att = lds[VEC_EMB_F + l * VEC_LAYER_F + V_COUNT * EH];
      // k-step weights in fragment order, pre-scaled like the k-step's A operand: [0] DIV_ST (w_r + w_e), [1] w_r, [2] w_e
      const float* vd = vdiv + l * VEC_DIV_F + hh * 16;
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        if (T >= ntile) continue;
        f32x16 Ai, dAr[DAL ? 1 : KA];
        {
          WFrag<2> wa;
          wa.load(nullptr, mats16h, M_WA, lane);
          Ai = wa.mul(hf[T], lds_vec16(vl + V_B1 * EH));
          Ai *= F16_UNSCALE;
          if (!first) {
#pragma unroll
            for (int d = 0; d < K; ++d) {
              f32x16 da = wa.mul(dhf[T][d], zero16);
              da *= F16_UNSCALE;
              if (DAL) {
                f32x4* dst = reinterpret_cast<f32x4*>(dA + d * C::PB_F + col[T] * PBS + hh * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = f32x4{da[4 * q], da[4 * q + 1], da[4 * q + 2], da[4 * q + 3]};
              } else {
                dAr[d] = da;
              }
            }
          }
        }
        wave_lds_fence();
        f32x16 agg = {0}, dagg[KA];
        float xacc[DIM], dxacc[KA][DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) xacc[k] = 0.f;
#pragma unroll
        for (int d = 0; d < K; ++d) {
          dagg[d] = zero16;
#pragma unroll
          for (int k = 0; k < DIM; ++k) dxacc[d][k] = 0.f;
        }
        const int cbase = col[T] - nodei[T];
        const float aggw = (l == L - 1) ? 0.0f : 1.0f;
        for (int dd = 1; dd < N; ++dd) {
          asm volatile("" ::: "memory");
          int j = nodei[T] + dd;
          j = (j >= N) ? j - N : j;
          const int cj = (col[T] < ncol) ? cbase + j : col[T];
          float df[DIM], e0[DIM], radial = 0.f, ea = 0.f;
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            df[k] = posi[T][k] - poscur[cj * DIM + k];
            radial = fmaf(df[k], df[k], radial);
            e0[k] = p0i[T][k] - pos0[cj * DIM + k];
            ea = fmaf(e0[k], e0[k], ea);
          }
          // ---- primal edge MLP with derivative factors (once for all K directions)
          f32x16 z = Ai + lds_vec16(PB + cj * PBS + hh * 16);
          z = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re, hh ? ea : radial, z, 0, 0, 0);
          f32x16 y1, g1, m2, g2, yc, gc;
          silu_dsilu16(z, 1.0f, y1, g1);
          z = w2f.mul(y1, lds_vec16(vl + V_B2 * EH));
          silu_dsilu16(z, F16_UNSCALE, m2, g2);
          float att = 1.0f, datt_f = 0.0f;
          const f32x16 v_watt = lds_vec16(vl + V_WATT * EH);
          if (p.attention) {
            att = fast_sigmoid(xhalf_sum(dot16(v_watt, m2)) + b_att);
            datt_f = att * (1.0f - att);
          }
          const f32x16 m = m2 * att;
#pragma unroll
          for (int r = 0; r < 16; ++r) agg[r] = fmaf(m[r], aggw, agg[r]);
          z = wc1f.mul(m, lds_vec16(vl + V_BC1 * EH));
          silu_dsilu16(z, F16_UNSCALE, yc, gc);
          float cs = xhalf_sum(dot16(lds_vec16(vl + V_WC2 * EH), yc)), dcs_f = 1.0f;
          if (p.tanh_on) {
            const float th = accurate_tanh(cs);
            dcs_f = p.coord_scale * fmaf(-th, th, 1.0f);
            cs = th * p.coord_scale;
          }
          dcs_f *= 1.0f / (DIV_ST * DIV_SV);  // the feature tangents carry DIV_ST, the head adjoint DIV_SV
          // adjoint of the coordinate head: d(w_c2 . silu(Wc1 m + b)) = vc . dm;  vc carries DIV_SV
          f32x16 vc = wc1t.mul(gc * lds_vec16(vd + 3 * EH), zero16);
          vc *= F16_UNSCALE;
          const float sq = __builtin_amdgcn_sqrtf(radial + 1e-8f), inv = __builtin_amdgcn_rcpf(sq + 1.0f);
          const float hsq = 0.5f * __builtin_amdgcn_rcpf(sq);
          float u[DIM];
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            u[k] = df[k] * inv;
            xacc[k] = fmaf(u[k], cs, xacc[k]);
          }
          // per-direction geometry
          float ddf[KA][DIM], dradial[KA], dea[KA];
#pragma unroll
          for (int d = 0; d < K; ++d) {
            const float* dposcur = dposb + (2 * d + cur) * C::POS_F;
            dradial[d] = 0.f;
            dea[d] = 0.f;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              ddf[d][k] = dposi[T][d][k] - dposcur[cj * DIM + k];
              dradial[d] = fmaf(df[k], ddf[d][k], dradial[d]);
              dea[d] = fmaf(e0[k], dp0i[T][d][k] - dpos0[d * C::POS_F + cj * DIM + k], dea[d]);
            }
            dradial[d] *= 2.0f;
            dea[d] *= 2.0f;
          }
          float* crec = clay ? clay + CA::POSF + (size_t)T * CA::tile_f(l, L) + (size_t)(dd - 1) * CA::edge_f(l, L)
                             : nullptr;
          float* cscal = crec ? crec + (size_t)CA::nvec(l, L) * 1024 : nullptr;
          if (cscal) {
            cscal[lane] = cs;
            cscal[64 + lane] = dcs_f;
            if (first || last) {
              // first / last layer records: the edge's geometry [column][df | inv] in slots 2-3, [column][e0 | hsq] in
              // slots 6-7 (their tangent sweeps read neither the gate nor its derivative)
              f32x4 gq = {0.f, 0.f, 0.f, hh ? hsq : inv};
              gq.x = hh ? e0[0] : df[0];
              gq.y = hh ? e0[1] : df[1];
              if (DIM > 2) gq.z = hh ? e0[DIM > 2 ? 2 : 0] : df[DIM > 2 ? 2 : 0];
              *reinterpret_cast<f32x4*>(cscal + (hh ? 384 : 128) + cl * 4) = gq;
            } else {
              cscal[128 + lane] = att;
              cscal[192 + lane] = p.attention ? 1.0f - att : 0.0f;  // the gate itself rides in g2
            }
          }
          float dcs[KA];
          if (first) {
            // every direction: dz1 = dradial_d (w_r + w_e) (radial == edge_attr, dh == 0): one shared tangent chain
            f32x16 du_ = w2f.mul(g1 * lds_vec16(vd), zero16);
            du_ *= g2 * F16_UNSCALE;  // dm2 per unit dradial, DIV_ST-scaled
            f32x16 dmu = du_ * att;
            if (p.attention) {
              const float dattu = datt_f * xhalf_sum(dot16(v_watt, du_));
#pragma unroll
              for (int r = 0; r < 16; ++r) dmu[r] = fmaf(dattu, m2[r], dmu[r]);
            }
            const float vcdmu = xhalf_sum(dot16(vc, dmu));
            if (crec) {
              cache_store16(crec, lane, dmu);
              cscal[256 + lane] = vcdmu;
            }
#pragma unroll
            for (int d = 0; d < K; ++d) {
              const float w_ = dradial[d] * aggw;
#pragma unroll
              for (int r = 0; r < 16; ++r) dagg[d][r] = fmaf(dmu[r], w_, dagg[d][r]);
              dcs[d] = dcs_f * (dradial[d] * vcdmu);
            }
          } else if (last) {
            // only the scalar head consumes this layer's tangents: vc . dm = qv . dz1
            const float svm = p.attention ? xhalf_sum(dot16(vc, m2)) : 0.0f;
            f32x16 uvec = vc * att;
            if (p.attention) {
              const float c1_ = datt_f * svm;
#pragma unroll
              for (int r = 0; r < 16; ++r) uvec[r] = fmaf(c1_, v_watt[r], uvec[r]);
            }
            f32x16 qv = w2t.mul(g2 * uvec, zero16);
            qv *= g1 * F16_UNSCALE;
            const float qr = DIV_ST * xhalf_sum(dot16(qv, lds_vec16(vd + EH)));
            const float qe = DIV_ST * xhalf_sum(dot16(qv, lds_vec16(vd + 2 * EH)));
            if (crec) {
              cache_store16(crec, lane, qv);
              cscal[256 + lane] = qr;
              cscal[320 + lane] = qe;
            }
#pragma unroll
            for (int d = 0; d < K; ++d) {
              const f32x16 dz1 = (DAL ? lds_vec16(dA + d * C::PB_F + col[T] * PBS + hh * 16) : dAr[DAL ? 0 : d]) +
                                 lds_vec16(dPB + d * C::PB_F + cj * PBS + hh * 16);
              const float sdot = xhalf_sum(dot16(qv, dz1));
              dcs[d] = dcs_f * fmaf(qr, dradial[d], fmaf(qe, dea[d], sdot));
            }
          } else {
            const f32x16 g2s = g2 * F16_UNSCALE;
            if (crec) {
              cache_store16(crec, lane, g1);
              cache_store16(crec + 1024, lane, g2s * att);  // dm2 arrives gated: att (g2 o W2 dm1)
              cache_store16(crec + 2048, lane, m2);
              cache_store16(crec + 3072, lane, vc);
              // the edge's geometry, once for every wave and direction that streams this record: [column][df | inv ||
              // e0 | hsq] in the four scalar slots the middle layers leave free
              f32x4 gq = {0.f, 0.f, 0.f, hh ? hsq : inv};
              gq.x = hh ? e0[0] : df[0];
              gq.y = hh ? e0[1] : df[1];
              if (DIM > 2) gq.z = hh ? e0[DIM > 2 ? 2 : 0] : df[DIM > 2 ? 2 : 0];
              *reinterpret_cast<f32x4*>(cscal + 256 + cl * 8 + hh * 4) = gq;
            }
#pragma unroll
            for (int d = 0; d < K; ++d) {
              f32x16 dz = (DAL ? lds_vec16(dA + d * C::PB_F + col[T] * PBS + hh * 16) : dAr[DAL ? 0 : d]) +
                          lds_vec16(dPB + d * C::PB_F + cj * PBS + hh * 16);
              dz = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re, DIV_ST * (hh ? dea[d] : dradial[d]), dz, 0, 0, 0);
              dz *= g1;
              dz = w2f.mul(dz, zero16);
              dz *= g2s;  // dm2
              f32x16 dm = dz * att;
              if (p.attention) {
                const float datt = datt_f * xhalf_sum(dot16(v_watt, dz));
#pragma unroll
                for (int r = 0; r < 16; ++r) dm[r] = fmaf(datt, m2[r], dm[r]);
              }
#pragma unroll
              for (int r = 0; r < 16; ++r) dagg[d][r] = fmaf(dm[r], aggw, dagg[d][r]);
              dcs[d] = dcs_f * xhalf_sum(dot16(vc, dm));
            }
          }
#pragma unroll
          for (int d = 0; d < K; ++d) {
            const float dnrm = dradial[d] * hsq;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              const float du = (ddf[d][k] - u[k] * dnrm) * inv;
              dxacc[d][k] = fmaf(du, cs, fmaf(u[k], dcs[d], dxacc[d][k]));
            }
          }
        }
        float* posnext = posb + (cur ^ 1) * C::POS_F;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          posi[T][k] += xacc[k];
          if (hh == 0) posnext[col[T] * DIM + k] = posi[T][k];
#pragma unroll
          for (int d = 0; d < K; ++d) {
            dposi[T][d][k] += dxacc[d][k];
            if (hh == 0) dposb[(2 * d + (cur ^ 1)) * C::POS_F + col[T] * DIM + k] = dposi[T][d][k];
          }
        }
        if (l != L - 1) {
          WFrag<2> wn;
          wn.load(nullptr, mats16h, M_WN1A, lane);
          f32x16 zn = wn.mul(hf[T], lds_vec16(vl + V_BN1 * EH));
          f32x16 dzn[KA];
#pragma unroll
          for (int d = 0; d < K; ++d) dzn[d] = first ? zero16 : wn.mul(dhf[T][d], zero16);
          wn.load(nullptr, mats16h, M_WN1B, lane);
          zn = wn.mul(agg, zn);
#pragma unroll
          for (int d = 0; d < K; ++d) dzn[d] = wn.mul(dagg[d], dzn[d]);
          f32x16 yn, gn;
          silu_dsilu16(zn, F16_UNSCALE, yn, gn);
          gn *= F16_UNSCALE;
          if (clay) cache_store16(clay + CA::POSF + (size_t)T * CA::tile_f(l, L) + CA::gn_off(l, L), lane, gn);
          wn.load(nullptr, mats16h, M_WN2, lane);
          f32x16 o = wn.mul(yn, lds_vec16(vl + V_BN2 * EH));
          o *= F16_UNSCALE;
          hf[T] += o;
#pragma unroll
          for (int d = 0; d < K; ++d) {
            dzn[d] *= gn;
            f32x16 dho = wn.mul(dzn[d], zero16);
            dho *= F16_UNSCALE;
            dhf[T][d] += dho;
          }
        }
      }
      wave_lds_fence();
      cur ^= 1;
    }

    // dD_d = c_s + c_out c_in (dF - mean dF)_d  (unit position tangents); the K terms of a walker are summed by ONE lane,
    // added only when finite, otherwise the walker is marked for the bf16x3 kernel
    float* dscr = dPB;          // [KA][NCOLP*DIM]
    float* tsl = PB;            // [G][KA] terms (the partner table is free now)
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int d = 0; d < K; ++d)
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          dposi[T][d][k] -= dp0i[T][d][k];  // dF
          if (hh == 0) dscr[d * C::POS_F + col[T] * DIM + k] = dposi[T][d][k];
        }
    if (lane < G * K) tsl[lane] = 0.f;
    wave_lds_fence();
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const int cb = (col[T] < ncol) ? col[T] - nodei[T] : 0;
#pragma unroll
      for (int d = 0; d < K; ++d)
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          if (!(valid[T] && hh == 0 && d < p.ndir && nodei[T] * DIM + k == p.dir0 + d)) continue;
          float ds = 0.f;
          for (int q = 0; q < N; ++q) ds += dscr[d * C::POS_F + (cb + q) * DIM + k];
          const float dF = p.no_mean ? dposi[T][d][k] : dposi[T][d][k] - ds / (float)N;
          tsl[(col[T] / N) * K + d] = fmaf(c_out[T] * c_in[T], dF, c_s[T]);
        }
    }
    wave_lds_fence();
    bool bad[NT];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      bad[T] = false;
      if constexpr (K > 0) {
        const int w = col[T] / N;
        float sum = 0.f;
#pragma unroll
        for (int d = 0; d < K; ++d) sum += tsl[(valid[T] ? w : 0) * K + d];
        bad[T] = valid[T] && !__builtin_isfinite(sum);
        if (valid[T] && hh == 0 && nodei[T] == 0) {
          if (!bad[T]) p.diag_acc[walker0 + w] += sum;
          p.mark[walker0 + w] = bad[T] ? 1 : 0;
          if (bad[T] && p.bad_flag) *p.bad_flag = p.bad_seq;
        }
      }
    }
    wave_lds_fence();
    // primal denoiser D = c_s x + c_out (F - mean F), F = pos^L - pos^0 (x = pos^0 / c_in).  A launch without directions
    // (K = 0: the cache writer) marks by the primal itself: a non-finite position anywhere in the walker reaches every
    // column through the mean
    if (p.out || K == 0) {
      float* scr = PB;
#pragma unroll
      for (int T = 0; T < NT; ++T)
        if (hh == 0) {
#pragma unroll
          for (int k = 0; k < DIM; ++k) scr[col[T] * DIM + k] = posi[T][k] - p0i[T][k];
        }
      wave_lds_fence();
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        if (!(valid[T] && hh == 0) || bad[T]) continue;
        const int cb = col[T] - nodei[T];
        float F[DIM];
        bool fin = true;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          float sum = 0.f;
          for (int q = 0; q < N; ++q) sum += scr[(cb + q) * DIM + k];
          F[k] = (posi[T][k] - p0i[T][k]) - sum / (float)N;
          fin = fin && __builtin_isfinite(F[k]);
        }
        if constexpr (K == 0) {
          if (nodei[T] == 0) {
            p.mark[walker0 + col[T] / N] = fin ? 0 : 1;
            if (!fin && p.bad_flag) *p.bad_flag = p.bad_seq;
          }
          if (!fin) continue;
        }
        if (p.out) {
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            const long long gi = (walker0 * N + col[T]) * DIM + k;
            p.out[gi] = fmaf(c_s[T], p.x[gi], c_out[T] * F[k]);
          }
        }
      }
      wave_lds_fence();
    }
  }
}


// Tangent-only kernel: K unit directions per launch from the primal cache written by egnn_div_fast_kernel (same grid,
// same walker quota, hence the same walker groups).  Per edge it recomputes the geometry from the cached layer
// positions (same expressions as the fast kernel: same bits), loads the edge record and runs exactly the tangent
// arithmetic of the fast kernel; no primal GEMM, no activation.  Wa dh_i lives in registers (no primal state competes
// for them), so the LDS budget goes to the K partner tables.
template <int N, int DIM, int G, int WAVES, int K>
struct DivTanCfg {
  static constexpr int NCOL = G * N;
  static constexpr int NT = (NCOL + 31) / 32;
  static constexpr int NCOLP = NT * 32;
  static constexpr int PB_F = NCOLP * PBS;
  static constexpr int POS_F = NCOLP * DIM;
  static constexpr int WAVE_F = K * PB_F + 2 * POS_F + 3 * K * POS_F;  // dPB[K]; pos, pos0; dpos[K][2], dpos0[K]
  static __host__ __device__ constexpr int vec_f(int L) { return ((VEC_EMB_F + L * VEC_LAYER_F) + 3) & ~3; }
  static __host__ __device__ constexpr size_t lds_bytes(int L) {
    return sizeof(float) * (size_t)(vec_f(L) + WAVES * WAVE_F);
  }
};

template <int N, int DIM, int G, int WAVES, int K>
__global__ void __launch_bounds__(WAVES * 64, 1) egnn_div_tangent_kernel(DivParams p) {
  using C = DivTanCfg<N, DIM, G, WAVES, K>;
  constexpr int NT = C::NT;
  using CA = DivCache<N, DIM, NT>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = p.n_layers;
  const int vec_f = C::vec_f(L);
  for (int i = threadIdx.x; i < VEC_EMB_F + L * VEC_LAYER_F; i += WAVES * 64) lds[i] = p.vecs_h[i];
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane & 31, hh = lane >> 5;
  float* dPB = lds + vec_f + wave * C::WAVE_F;     // [K][PB_F]  DIV_ST x Wb dh_j
  float* posc = dPB + K * C::PB_F;                 // positions entering the current layer
  float* pos0 = posc + C::POS_F;
  float* dposb = pos0 + C::POS_F;                  // [K][2][POS_F]
  float* dpos0 = dposb + 2 * K * C::POS_F;         // [K][POS_F]
  const f32x16 zero16 = {0};

  const long long total_waves = (long long)gridDim.x * WAVES;
  const long long quota = (p.B + total_waves - 1) / total_waves;
  const long long wbeg = ((long long)blockIdx.x * WAVES + wave) * quota;
  const long long wend = (wbeg + quota < p.B) ? wbeg + quota : p.B;
  const long long groups_per_wave = (quota + G - 1) / G;
  for (long long walker0 = wbeg; walker0 < wend; walker0 += G) {
    const int nwalk = (int)((wend - walker0) < G ? (wend - walker0) : G);
    const int ncol = nwalk * N;
    const int ntile = (ncol + 31) >> 5;
    const float* cgrp = p.cache + (size_t)(((long long)blockIdx.x * WAVES + wave) * groups_per_wave + (walker0 - wbeg) / G) *
                                      CA::group_f(L);
    int col[NT], nodei[NT];
    bool valid[NT];
    float c_s[NT], cc[NT];
    float dposi[NT][K][DIM], dp0i[NT][K][DIM];
    f32x16 dhf[NT][K];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      col[T] = T * 32 + cl;
      const int w = col[T] / N;
      nodei[T] = col[T] - w * N;
      valid[T] = col[T] < ncol;
      const long long wid = valid[T] ? walker0 + w : p.B - 1;
      const float hv = p.h[wid];
      const float op = 1.0f + hv, rs = 1.0f / sqrtf(op);
      c_s[T] = 1.0f / op;
      cc[T] = (sqrtf(hv) * rs) * rs;  // c_out c_in
#pragma unroll
      for (int k = 0; k < DIM; ++k)
#pragma unroll
        for (int q = 0; q < K; ++q) {
          const float v = (valid[T] && q < p.ndir && nodei[T] * DIM + k == p.dir0 + q) ? 1.0f : 0.f;
          dposi[T][q][k] = v;
          dp0i[T][q][k] = v;
          if (hh == 0) {
            dpos0[q * C::POS_F + col[T] * DIM + k] = v;
            dposb[(2 * q) * C::POS_F + col[T] * DIM + k] = v;
          }
        }
#pragma unroll
      for (int q = 0; q < K; ++q) dhf[T][q] = zero16;
    }
    wave_lds_fence();

    int cur = 0;
    for (int l = 0; l < L; ++l) {
      const unsigned* mats16h = p.mats16h + (size_t)l * M_COUNT * MAT_WH;
      const bool first = (l == 0), last = (l == L - 1) && !first;
      const float* clay = cgrp + CA::layer_off(l, L);
      if (!first) {  // (the edges' geometry comes with their records: the layer's position block is not read here)
        WFrag<2> wb;
        wb.load(nullptr, mats16h, M_WB, lane);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          if (T >= ntile) continue;
#pragma unroll
          for (int d = 0; d < K; ++d) {
            f32x16 dpb = wb.mul(dhf[T][d], zero16);
            dpb *= F16_UNSCALE;
            f32x4* ddst = reinterpret_cast<f32x4*>(dPB + d * C::PB_F + col[T] * PBS + hh * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) ddst[q] = f32x4{dpb[4 * q], dpb[4 * q + 1], dpb[4 * q + 2], dpb[4 * q + 3]};
          }
        }
      }
      wave_lds_fence();
      WFrag<2> w2f;
      if (!first && !last) w2f.load(nullptr, mats16h, M_W2, lane);
      const float a_re = lds[VEC_EMB_F + l * VEC_LAYER_F + V_WRE * EH + lane];
      const float aggw = (l == L - 1) ? 0.0f : 1.0f;
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        if (T >= ntile) continue;
        f32x16 dAr[K];
        if (!first) {
          WFrag<2> wa;
          wa.load(nullptr, mats16h, M_WA, lane);
#pragma unroll
          for (int d = 0; d < K; ++d) {
            dAr[d] = wa.mul(dhf[T][d], zero16);
            dAr[d] *= F16_UNSCALE;
          }
        }
        f32x16 dagg[K];
        float dxacc[K][DIM];
#pragma unroll
        for (int d = 0; d < K; ++d) {
          dagg[d] = zero16;
#pragma unroll
          for (int k = 0; k < DIM; ++k) dxacc[d][k] = 0.f;
        }
        const int cbase = col[T] - nodei[T];
        const float* ctile = clay + CA::POSF + (size_t)T * CA::tile_f(l, L);
        for (int dd = 1; dd < N; ++dd) {
          asm volatile("" ::: "memory");
          int j = nodei[T] + dd;
          j = (j >= N) ? j - N : j;
          const int cj = (col[T] < ncol) ? cbase + j : col[T];
          const float* crec = ctile + (size_t)(dd - 1) * CA::edge_f(l, L);
          const float* cscal = crec + (size_t)CA::nvec(l, L) * 1024;
          const float cs = cscal[lane], dcs_f = cscal[64 + lane];
          float df[DIM], e0[DIM], inv, hsq;
          if (!first && !last) {  // middle-layer records carry the edge's geometry
            const f32x4 ga = *reinterpret_cast<const f32x4*>(cscal + 256 + cl * 8);
            const f32x4 gb = *reinterpret_cast<const f32x4*>(cscal + 256 + cl * 8 + 4);
            df[0] = ga.x; e0[0] = gb.x;
            df[1] = ga.y; e0[1] = gb.y;
            if constexpr (DIM > 2) { df[DIM - 1] = ga.z; e0[DIM - 1] = gb.z; }
            inv = ga.w;
            hsq = gb.w;
          } else {  // first / last layer records: slots 2-3 and 6-7
            const f32x4 ga = *reinterpret_cast<const f32x4*>(cscal + 128 + cl * 4);
            const f32x4 gb = *reinterpret_cast<const f32x4*>(cscal + 384 + cl * 4);
            df[0] = ga.x; e0[0] = gb.x;
            df[1] = ga.y; e0[1] = gb.y;
            if constexpr (DIM > 2) { df[DIM - 1] = ga.z; e0[DIM - 1] = gb.z; }
            inv = ga.w;
            hsq = gb.w;
          }
          float u[DIM];
#pragma unroll
          for (int k = 0; k < DIM; ++k) u[k] = df[k] * inv;
          float ddf[K][DIM], dradial[K], dea[K];
#pragma unroll
          for (int d = 0; d < K; ++d) {
            const float* dposcur = dposb + (2 * d + cur) * C::POS_F;
            dradial[d] = 0.f;
            dea[d] = 0.f;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              ddf[d][k] = dposi[T][d][k] - dposcur[cj * DIM + k];
              dradial[d] = fmaf(df[k], ddf[d][k], dradial[d]);
              dea[d] = fmaf(e0[k], dp0i[T][d][k] - dpos0[d * C::POS_F + cj * DIM + k], dea[d]);
            }
            dradial[d] *= 2.0f;
            dea[d] *= 2.0f;
          }
          float dcs[K];
          if (first) {
            const f32x16 dmu = cache_load16(crec, lane);
            const float vcdmu = cscal[256 + lane];
#pragma unroll
            for (int d = 0; d < K; ++d) {
              const float w_ = dradial[d] * aggw;
#pragma unroll
              for (int r = 0; r < 16; ++r) dagg[d][r] = fmaf(dmu[r], w_, dagg[d][r]);
              dcs[d] = dcs_f * (dradial[d] * vcdmu);
            }
          } else if (last) {
            const f32x16 qv = cache_load16(crec, lane);
            const float qr = cscal[256 + lane], qe = cscal[320 + lane];
#pragma unroll
            for (int d = 0; d < K; ++d) {
              const f32x16 dz1 = dAr[d] + lds_vec16(dPB + d * C::PB_F + cj * PBS + hh * 16);
              const float sdot = xhalf_sum(dot16(qv, dz1));
              dcs[d] = dcs_f * fmaf(qr, dradial[d], fmaf(qe, dea[d], sdot));
            }
          } else {
            const f32x16 g1 = cache_load16(crec, lane), g2s = cache_load16(crec + 1024, lane);
            const f32x16 m2 = cache_load16(crec + 2048, lane), vc = cache_load16(crec + 3072, lane);
            const float datt_f = cscal[192 + lane];  // 1 - att: the record's g2 carries the gate
            const f32x16 v_watt = lds_vec16(lds + VEC_EMB_F + l * VEC_LAYER_F + hh * 16 + V_WATT * EH);
#pragma unroll
            for (int d = 0; d < K; ++d) {
              f32x16 dz = dAr[d] + lds_vec16(dPB + d * C::PB_F + cj * PBS + hh * 16);
              dz = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re, DIV_ST * (hh ? dea[d] : dradial[d]), dz, 0, 0, 0);
              dz *= g1;
              dz = w2f.mul(dz, zero16);
              dz *= g2s;  // att dm2
              f32x16 dm = dz;
              if (p.attention) {
                const float datt = datt_f * xhalf_sum(dot16(v_watt, dz));
#pragma unroll
                for (int r = 0; r < 16; ++r) dm[r] = fmaf(datt, m2[r], dm[r]);
              }
#pragma unroll
              for (int r = 0; r < 16; ++r) dagg[d][r] = fmaf(dm[r], aggw, dagg[d][r]);
              dcs[d] = dcs_f * xhalf_sum(dot16(vc, dm));
            }
          }
#pragma unroll
          for (int d = 0; d < K; ++d) {
            const float dnrm = dradial[d] * hsq;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              const float du = (ddf[d][k] - u[k] * dnrm) * inv;
              dxacc[d][k] = fmaf(du, cs, fmaf(u[k], dcs[d], dxacc[d][k]));
            }
          }
        }
#pragma unroll
        for (int k = 0; k < DIM; ++k)
#pragma unroll
          for (int d = 0; d < K; ++d) {
            dposi[T][d][k] += dxacc[d][k];
            if (hh == 0) dposb[(2 * d + (cur ^ 1)) * C::POS_F + col[T] * DIM + k] = dposi[T][d][k];
          }
        if (l != L - 1) {
          const f32x16 gn = cache_load16(ctile + CA::gn_off(l, L), lane);
          WFrag<2> wn;
          f32x16 dzn[K];
          if (!first) {
            wn.load(nullptr, mats16h, M_WN1A, lane);
#pragma unroll
            for (int d = 0; d < K; ++d) dzn[d] = wn.mul(dhf[T][d], zero16);
          } else {
#pragma unroll
            for (int d = 0; d < K; ++d) dzn[d] = zero16;
          }
          wn.load(nullptr, mats16h, M_WN1B, lane);
#pragma unroll
          for (int d = 0; d < K; ++d) dzn[d] = wn.mul(dagg[d], dzn[d]);
          wn.load(nullptr, mats16h, M_WN2, lane);
#pragma unroll
          for (int d = 0; d < K; ++d) {
            dzn[d] *= gn;
            f32x16 dho = wn.mul(dzn[d], zero16);
            dho *= F16_UNSCALE;
            dhf[T][d] += dho;
          }
        }
      }
      wave_lds_fence();
      cur ^= 1;
    }

    // epilogue: as egnn_div_fast_kernel
    float* dscr = dPB;  // [K][NCOLP*DIM]
    float* tsl = dPB + K * C::POS_F;  // [G][K]
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int d = 0; d < K; ++d)
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          dposi[T][d][k] -= dp0i[T][d][k];  // dF
          if (hh == 0) dscr[d * C::POS_F + col[T] * DIM + k] = dposi[T][d][k];
        }
    if (lane < G * K) tsl[lane] = 0.f;
    wave_lds_fence();
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const int cb = (col[T] < ncol) ? col[T] - nodei[T] : 0;
#pragma unroll
      for (int d = 0; d < K; ++d)
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          if (!(valid[T] && hh == 0 && d < p.ndir && nodei[T] * DIM + k == p.dir0 + d)) continue;
          float ds = 0.f;
          for (int q = 0; q < N; ++q) ds += dscr[d * C::POS_F + (cb + q) * DIM + k];
          const float dF = p.no_mean ? dposi[T][d][k] : dposi[T][d][k] - ds / (float)N;
          tsl[(col[T] / N) * K + d] = fmaf(cc[T], dF, c_s[T]);
        }
    }
    wave_lds_fence();
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const int w = col[T] / N;
      float sum = 0.f;
#pragma unroll
      for (int d = 0; d < K; ++d) sum += tsl[(valid[T] ? w : 0) * K + d];
      if (valid[T] && hh == 0 && nodei[T] == 0) {
        const bool bad = !__builtin_isfinite(sum);
        if (!bad) p.diag_acc[walker0 + w] += sum;
        p.mark[walker0 + w] = bad ? 1 : 0;
        if (bad && p.bad_flag) *p.bad_flag = p.bad_seq;
      }
    }
    wave_lds_fence();
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Block-shared tangent kernel.  The wave-owned kernel above runs one wave per SIMD (K = 4 directions need all 512
// registers, a quarter of its instructions are v_accvgpr moves, and a lone wave issues at half the vector rate) and
// every launch re-reads the whole cache for four directions.  Here the NW waves of a block (two per SIMD, K = 2
// directions each: everything fits the 256 VALU-visible registers) work on the SAME walker group, so a group's cache
// is streamed from HBM once per NW K directions.  The block moves it through an LDS ring with LDS-DMA
// (global_load_lds_dwordx4: no registers, no staging instructions) in 1 KB pieces; the sweep consumes it in ITEMS of
// at most SHR_S = 18 pieces (one middle-layer edge record; three first/last-layer records; a layer's [Wb Wa W2 |
// positions]; a tile's [Wn1a Wn1b Wn2 | gn]), every item padded to SHR_S pieces in the piece numbering, so that an
// item is one contiguous ring slot (compile-time LDS offsets) and piece F of the sequence is issued by wave F mod NW
// into ring position F mod (NSLOT SHR_S).  Per item ONE workgroup barrier:
//     wait for my own pieces of the item (counted s_waitcnt vmcnt) -> barrier -> refill the slots of the items already
//     consumed by every wave -> compute from LDS.
// No other vector-memory instruction runs inside the stream -- loads return in order, so any register load (or a
// register spill: scratch is vector memory) would wait behind the whole prefetch window: the layer's weight
// fragments travel through the ring as well (4 KB lane-linear blocks, L2-resident, spliced into the group's sequence
// where the sweep needs them; padding pieces re-read the first weight block), c_skip / c_out c_in arrive with the
// position block, results are parked in LDS and flushed every RES_G groups behind a full drain -- so the hand-counted
// vmcnt is exact.
constexpr int SHR_LMAX = 4;  // layers the block-shared kernel's LDS budget is sized for
constexpr int SHR_S = 18;  // pieces per item slot: one middle-layer edge record (4 vectors + 2 KB of scalars)
template <int N, int DIM, int G, int NW, int K>
struct DivShrCfg {
  static constexpr int NCOL = G * N;
  static constexpr int NT = (NCOL + 31) / 32;
  static constexpr int NCOLP = NT * 32;
  static constexpr int PB_F = NCOLP * PBS;
  static constexpr int POS_F = NCOLP * DIM;
  static constexpr int WAVE_F = K * PB_F + 2 * K * POS_F;   // dPB[K]; dpos[K][2]
  static constexpr int RES_G = 32;                           // groups between result flushes
  // piece table of one group's sequence: one entry per piece, or -- systems whose sequence is longer (LJ55: 189 items)
  // -- two words per ITEM (every item is [weight pieces | cache pieces | padding]: first piece and count of each run)
  static constexpr bool COMPACT = NT * (N - 1) > 48;
  static constexpr int MAX_P = COMPACT ? 768 : 128 * NW;
  // pos, pos0; c_s, cc; comb; res, walker0; piece table (+ count)
  static constexpr int SHARED_F = 2 * POS_F + 2 * NCOLP + NW * G + RES_G * G + RES_G + MAX_P + 4;
  static constexpr int LMAX = SHR_LMAX;
  static __host__ __device__ constexpr int vec_f(int L) { return ((VEC_EMB_F + L * VEC_LAYER_F) + 3) & ~3; }
  static constexpr int FIXED_F = ((vec_f(LMAX) + SHARED_F + NW * WAVE_F) + 255) & ~255;
  // ring slots: what the 160 KB leave (the ring sits at the bottom of the allocation)
  static constexpr int NSLOT = (160 * 1024 - 4 * FIXED_F) / (SHR_S * 1024);
  // an item is loaded by D waves (SHR_S / D pieces each), item j by waves (j D) mod NW ...; a wave's turns are NW / D
  // items apart, at least the NSLOT - 1 items of prefetch distance, so it never has two batches in flight and "my part
  // of this item has landed" is vmcnt(0)
  static constexpr int D = (NW % 3 == 0 && NW / 3 >= NSLOT) ? 3 : ((NW % 2 == 0 && NW / 2 >= NSLOT) ? 2 : 1);
  static constexpr int PPW = SHR_S / D;
  static_assert(NSLOT >= 3 && NW / D >= NSLOT && SHR_S % D == 0, "ring / duty split");
  static __host__ __device__ constexpr size_t lds_bytes(int) { return (size_t)NSLOT * SHR_S * 1024 + sizeof(float) * (size_t)FIXED_F; }
};

// workgroup barrier that leaves vector-memory operations in flight (__syncthreads() drains them)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// LDS-DMA of one 1 KB piece: lane i's 16 bytes from sbase + voff land at LDS byte lds_byte + 16 i
__device__ __forceinline__ void glds16s(unsigned lds_byte, const float* sbase, unsigned voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}
// a run of N <= 4 consecutive pieces from one source: ONE m0 / address set-up, the instruction offset advances the
// global address and the LDS address together (LDS address = m0 + offset + 16 lane)
template <int N>
__device__ __forceinline__ void glds16s_run(unsigned lds_byte, const float* sbase, unsigned voff) {
  unsigned keep;
  if constexpr (N == 4)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
  else if constexpr (N == 3)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
  else if constexpr (N == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ f32x16 slot_vec16(const float* rec, int v, int lane) {  // vector v of a record (cache_store16 layout)
  const f32x4* d = reinterpret_cast<const f32x4*>(rec + v * 1024) + lane;
  f32x16 r;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 t = d[q * 64];
    r[4 * q] = t.x; r[4 * q + 1] = t.y; r[4 * q + 2] = t.z; r[4 * q + 3] = t.w;
  }
  return r;
}
__device__ __forceinline__ WFrag<2> slot_wfrag(const float* item, int m, int lane) {  // matrix block m of an item (WFrag<2>::load layout)
  const u32x4* d = reinterpret_cast<const u32x4*>(item + m * MAT_WH) + lane;
  WFrag<2> f;
#pragma unroll
  for (int pc = 0; pc < 2; ++pc)
#pragma unroll
    for (int st = 0; st < 2; ++st) f.w[pc][st] = d[(pc * 2 + st) * 64];
  return f;
}

template <int N, int DIM, int G, int NW, int K>
__global__ void __launch_bounds__(NW * 64, 1) egnn_div_tangent_shared_kernel(DivParams p) {
  using C = DivShrCfg<N, DIM, G, NW, K>;
  constexpr int NT = C::NT;
  constexpr int S = SHR_S;
  using CA = DivCache<N, DIM, NT>;
  static_assert(12 + CA::POSF / 256 <= S && MAT_WH == 1024, "an item must fit a ring slot");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = p.n_layers;
  constexpr int NSLOT = C::NSLOT, RC = NSLOT * S;
  float* ring = lds;                                 // [NSLOT][S][256]
  float* vecs = lds + RC * 256;
  for (int i = threadIdx.x; i < VEC_EMB_F + L * VEC_LAYER_F; i += NW * 64) vecs[i] = p.vecs_h[i];
  float* posc = vecs + C::vec_f(C::LMAX);            // block-shared: positions entering the current layer
  float* pos0 = posc + C::POS_F;
  float* cstab = pos0 + C::POS_F;                    // [NCOLP] c_skip, [NCOLP] c_out c_in
  float* comb = cstab + 2 * C::NCOLP;                // [NW][G]
  float* res = comb + NW * G;                        // [RES_G][G]
  int* resw = reinterpret_cast<int*>(res + C::RES_G * G);  // [RES_G] first walker of the group
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, cl = lane & 31, hh = lane >> 5;
  int* ptab = resw + C::RES_G;                       // [MAX_P] source of every piece of a group's sequence: bit 31 weights / cache, piece offset; then the count
  float* dPB = reinterpret_cast<float*>(ptab + C::MAX_P + 4) + wave * C::WAVE_F;  // [K][PB_F]  DIV_ST x Wb dh_j
  float* dposb = dPB + K * C::PB_F;                  // [K][2][POS_F]
  const unsigned ring_byte = (unsigned)(size_t)ring;
  const f32x16 zero16 = {0};
  // this wave's directions
  const int mydir0 = p.dir0 + wave * K;
  const int myndir = (p.ndir - wave * K) < 0 ? 0 : ((p.ndir - wave * K) > K ? K : (p.ndir - wave * K));
  const bool active = myndir > 0;

  // walker groups as the fast kernel laid them out: cache_waves waves of `quota` walkers, gpw groups each
  const long long cwaves = p.cache_waves;
  const long long quota = (p.B + cwaves - 1) / cwaves;
  const long long gpw = (quota + G - 1) / G;
  const long long total_groups = cwaves * gpw;
  const size_t group_f = CA::group_f(L);
  if (threadIdx.x == 0) {  // the pieces of one group's sweep, item by item (every item padded to S pieces)
    int np = 0;
    auto add = [&](int kind, long long first_piece, int n) {  // kind 0: cache, 1: weight blocks, 2: padding (re-reads weight piece 0)
      if constexpr (C::COMPACT) {
        if (kind == 0 && n > 0 && np + 1 < C::MAX_P) ptab[np + 1] = (int)(((unsigned)n << 26) | (unsigned)first_piece);
        if (kind == 1 && n > 0 && np < C::MAX_P) ptab[np] = (int)(((unsigned)n << 24) | (unsigned)first_piece);
      } else {
        for (int q = 0; q < n && np < C::MAX_P; ++q) ptab[np++] = (int)((kind ? 0x80000000u : 0u) | (unsigned)(first_piece + q));
      }
    };
    auto open_item = [&]() {
      if constexpr (C::COMPACT)
        if (np + 1 < C::MAX_P) { ptab[np] = 0; ptab[np + 1] = 0; }
    };
    auto close_item = [&]() {
      if constexpr (C::COMPACT) np += 2;
    };
    for (int l = 0; l < L; ++l) {
      const bool first = (l == 0), last = (l == L - 1) && !first;
      int n = 0;
      open_item();
      if (!first) {  // Wa, Wb[, W2]: consecutive blocks of the layer's matrices
        n = last ? 8 : 12;
        add(1, ((long long)l * M_COUNT + M_WA) * 4, n);
      }
      const long long lo = (long long)(CA::layer_off(l, L) / 256);
      add(0, lo, CA::POSF / 256);
      add(2, 0, S - n - CA::POSF / 256);
      close_item();
      const int ep = (int)(CA::edge_f(l, L) / 256), epi = S / ep;
      for (int T = 0; T < NT; ++T) {
        const long long to = lo + CA::POSF / 256 + (long long)T * (long long)(CA::tile_f(l, L) / 256);
        for (int q = 0; q < N - 1; q += epi) {
          const int nrec = (N - 1 - q) < epi ? (N - 1 - q) : epi;
          open_item();
          add(0, to + (long long)q * ep, nrec * ep);
          add(2, 0, S - nrec * ep);
          close_item();
        }
        n = 0;
        open_item();
        if (l != L - 1) {  // [Wn1a,] Wn1b, Wn2
          n = first ? 8 : 12;
          add(1, ((long long)l * M_COUNT + (first ? M_WN1B : M_WN1A)) * 4, n);
        }
        add(0, to + (long long)(CA::gn_off(l, L) / 256), 4);
        add(2, 0, S - n - 4);
        close_item();
      }
    }
    ptab[C::MAX_P] = np;
  }
  __syncthreads();
  const int IPG = ptab[C::MAX_P] / (C::COMPACT ? 2 : S);  // items per group (host: fits MAX_P)
  const int ngroups_i = (int)total_groups, gpw_i = (int)gpw, nblk = (int)gridDim.x;  // host: total_groups < 2^31
  auto group_walkers = [&](int g, long long& w0) -> int {
    const int wg = g / gpw_i, lg = g - wg * gpw_i;
    w0 = (long long)wg * quota + (long long)lg * G;
    long long we = (long long)(wg + 1) * quota;
    we = we < p.B ? we : p.B;
    const long long n = we - w0;
    return (int)(n < 0 ? 0 : (n > G ? G : n));
  };
  auto next_group = [&](int g) -> int {
    long long w0;
    while (g < ngroups_i && group_walkers(g, w0) == 0) g += nblk;
    return g;
  };

  // refill cursor (block-uniform): the next item to load is item ri of group rgrp, into ring slot rs; its D loading
  // waves start at wave rt
  constexpr int D = C::D, PPW = C::PPW;
  int ri = 0, rs = 0, rt = 0;
  int rgrp = next_group(blockIdx.x);
  const float* wsrc = reinterpret_cast<const float*>(p.mats16h);
  const float* gsrc = p.cache + (size_t)(rgrp < ngroups_i ? rgrp : 0) * group_f;
  const unsigned lane16 = lane * 16;
  auto load_next_item = [&]() {
    if (rgrp >= ngroups_i) return;
    const int d = wave - rt;
    if ((unsigned)d < (unsigned)D) {
      int ents;
      if constexpr (C::COMPACT) {
        const unsigned ew = (unsigned)ptab[2 * ri], ec = (unsigned)ptab[2 * ri + 1];
        const int nw_ = (int)(ew >> 24), nc_ = (int)(ec >> 26), q = d * PPW + (lane < PPW ? lane : 0);
        ents = q < nw_ ? (int)(0x80000000u | ((ew & 0xffffffu) + (unsigned)q))
                       : (q < nw_ + nc_ ? (int)((ec & 0x3ffffffu) + (unsigned)(q - nw_)) : (int)0x80000000u);
      } else {
        ents = ptab[ri * S + d * PPW + (lane < PPW ? lane : 0)];
      }
      const unsigned dst = ring_byte + (unsigned)(rs * S + d * PPW) * 1024u;
      // edge items (and the weight runs of the others) are PPW consecutive pieces of one source: one address set-up per
      // four pieces instead of one per piece -- the loading waves' scalar work is what every other wave waits for
      const int e0 = __builtin_amdgcn_readlane(ents, 0);
      if (__builtin_amdgcn_ballot_w64(lane < PPW && ents - lane != e0) == 0ull) {
        const float* sb = (e0 < 0 ? wsrc : gsrc) + (size_t)(e0 & 0x7fffffff) * 256;
#pragma unroll
        for (int q = 0; q < PPW; q += 4) {
          if (PPW - q >= 4) glds16s_run<4>(dst + q * 1024u, sb + q * 256, lane16);
          else if (PPW - q == 3) glds16s_run<3>(dst + q * 1024u, sb + q * 256, lane16);
          else if (PPW - q == 2) glds16s_run<2>(dst + q * 1024u, sb + q * 256, lane16);
          else glds16s_run<1>(dst + q * 1024u, sb + q * 256, lane16);
        }
      } else {
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
          const int ent = __builtin_amdgcn_readlane(ents, q);
          glds16s(dst + q * 1024u, (ent < 0 ? wsrc : gsrc) + (size_t)(ent & 0x7fffffff) * 256, lane16);
        }
      }
    }
    rs = (rs + 1 == NSLOT) ? 0 : rs + 1;
    rt = (rt + D == NW) ? 0 : rt + D;
    if (++ri == IPG) {
      ri = 0;
      rgrp = next_group(rgrp + nblk);
      if (rgrp < ngroups_i) gsrc = p.cache + (size_t)rgrp * group_f;
    }
  };
  // consumer state: the current item sits in ring slot cslot and was loaded by waves ct .. ct + D - 1
  int cslot = 0, ct = 0;
  const float* item = ring;
  auto begin_item = [&]() {
    if ((unsigned)(wave - ct) < (unsigned)D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my part of it has landed
    lds_barrier();
    load_next_item();  // into the slot of the item every wave has just left
  };
  auto end_item = [&]() {
    cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
    ct = (ct + D == NW) ? 0 : ct + D;
    item = ring + cslot * (S * 256);
  };
  __syncthreads();
  for (int q = 0; q < NSLOT - 1; ++q) load_next_item();

  int nres = 0;
  auto flush = [&]() {  // wave 0: add the parked results to the trace; a full drain keeps the piece count exact
    if (wave == 0) {
      for (int i = lane; i < nres * G; i += 64) {
        const int gi = i / G, w = i - gi * G;
        const float v = res[i];
        if (!(v == 0.0f && __builtin_signbit(v))) {  // -0.0 marks "no such walker"
          const long long wid = (long long)resw[gi] + w;
          const bool bad = !__builtin_isfinite(v);
          if (!bad) p.diag_acc[wid] += v;
          p.mark[wid] = bad ? 1 : 0;
          if (bad && p.bad_flag) *p.bad_flag = p.bad_seq;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    nres = 0;
  };

  for (int gid = next_group(blockIdx.x); gid < ngroups_i; gid = next_group(gid + nblk)) {
    long long walker0;
    const int nwalk = group_walkers(gid, walker0);
    const int ncol = nwalk * N;
    const int ntile = (ncol + 31) >> 5;
    int col[NT], nodei[NT];
    bool valid[NT];
    float dposi[NT][K][DIM];
    auto unit = [&](int T, int q, int k) -> float {  // component k of column col[T] of this wave's unit direction q
      return (valid[T] && q < myndir && nodei[T] * DIM + k == mydir0 + q) ? 1.0f : 0.f;
    };
    f32x16 dhf[NT][K];
    int dnode[K], dk[K];  // this wave's directions as (node, component); node -1: no such direction
    float si[NT][K];      // 2 [column's node == dnode]
#pragma unroll
    for (int q = 0; q < K; ++q) {
      const int dq = mydir0 + q;
      dnode[q] = (q < myndir) ? dq / DIM : -1;
      dk[q] = dq - (dq / DIM) * DIM;
    }
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      col[T] = T * 32 + cl;
      const int w = col[T] / N;
      nodei[T] = col[T] - w * N;
      valid[T] = col[T] < ncol;
#pragma unroll
      for (int q = 0; q < K; ++q) si[T][q] = (valid[T] && nodei[T] == dnode[q]) ? 2.0f : 0.0f;
#pragma unroll
      for (int k = 0; k < DIM; ++k)
#pragma unroll
        for (int q = 0; q < K; ++q) {
          const float v = unit(T, q, k);
          dposi[T][q][k] = v;
          if (hh == 0) {
            dposb[(2 * q) * C::POS_F + col[T] * DIM + k] = v;
            dposb[(2 * q + 1) * C::POS_F + col[T] * DIM + k] = 0.f;  // filled by the re-mapped first layer
          }
        }
#pragma unroll
      for (int q = 0; q < K; ++q) dhf[T][q] = zero16;
    }

    // Last layer of a FULL trace (DivParams::no_mean): a direction's term needs the tangent of ONE output coordinate,
    // (dnode, dk) of every walker -- the edges (dnode, j) only.  Lanes are re-mapped for that layer: lane = (virtual
    // column, feature half), virtual column = (walker of the group, direction of this wave, record of the item); one
    // pass per item instead of [records] x [directions] passes over all columns.
    constexpr int EPI2 = S / 6;  // records per first- / last-layer item
    static_assert(G * K * EPI2 <= 32, "virtual columns of the last layer");
    // (the kernel is launched for full traces of networks with at least two layers only: divshr_fits)
    // (the virtual column's coordinates are recomputed where they are used -- the last layer and the epilogue -- so that
    // nothing of them stays live across the other layers)
    auto vcol = [&](int& v_e, int& v_d, int& v_w, int& v_i0, int& v_k0, bool& v_on, int& v_col0, int& v_c0) {
      v_e = cl % EPI2; v_d = (cl / EPI2) % K; v_w = cl / (EPI2 * K);
      v_i0 = 0; v_k0 = 0;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        v_i0 = (v_d == q) ? dnode[q] : v_i0;
        v_k0 = (v_d == q) ? dk[q] : v_k0;
      }
      v_on = active && v_w < nwalk && v_w < G && v_i0 >= 0;
      v_col0 = v_on ? v_w * N + v_i0 : 0;
      v_c0 = v_col0 & 31;
    };
    float vacc[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) vacc[k] = 0.f;

    int cur = 0;
    // the layer sweep as a generic lambda: the first layer is its own instantiation, called ahead of the loop, so that
    // nothing of the later layers' state (feature tangents, Wa dh_i, W2 fragments) is live while it runs
    auto layer = [&](const int l, auto kind_tag) {  // kind 0: first layer, 1: middle layers, 2: last layer
      constexpr bool first = decltype(kind_tag)::value == 0;
      constexpr bool last = decltype(kind_tag)::value == 2;
      // first item of the layer: [Wb Wa W2 |] position block; the positions are copied to the block's tables (the
      // ring moves on), the fragments to registers
      const int nwm = first ? 0 : (last ? 2 : 3);  // matrices ahead of the positions
      begin_item();
      {
        const float* ipos = item + nwm * MAT_WH;  // (positions: the edges' geometry comes with their records)
        if (first)
          for (int i = threadIdx.x; i < 2 * C::NCOLP; i += NW * 64) cstab[i] = ipos[CA::POSX + i];
      }
      WFrag<2> w2f;
      f32x16 dArT[NT][K];  // DIV_ST x Wa dh_i of every tile, formed while the item is here: Wa itself is not kept
      if (!first && active) {
        {
          const WFrag<2> wa = slot_wfrag(item, 0, lane);
#pragma unroll
          for (int T = 0; T < NT; ++T) {
            if (T >= ntile) continue;
#pragma unroll
            for (int d = 0; d < K; ++d) {
              dArT[T][d] = wa.mul(dhf[T][d], zero16);
              dArT[T][d] *= F16_UNSCALE;
            }
          }
        }
        const WFrag<2> wb = slot_wfrag(item, 1, lane);
        if (!last) w2f = slot_wfrag(item, 2, lane);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          if (T >= ntile) continue;
#pragma unroll
          for (int d = 0; d < K; ++d) {
            f32x16 dpb = wb.mul(dhf[T][d], zero16);
            dpb *= F16_UNSCALE;
            f32x4* ddst = reinterpret_cast<f32x4*>(dPB + d * C::PB_F + col[T] * PBS + hh * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) ddst[q] = f32x4{dpb[4 * q], dpb[4 * q + 1], dpb[4 * q + 2], dpb[4 * q + 3]};
          }
        }
      }
      end_item();
      const float a_re = vecs[VEC_EMB_F + l * VEC_LAYER_F + V_WRE * EH + lane];
      const float aggw = (l == L - 1) ? 0.0f : 1.0f;
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        const bool tile_on = active && T < ntile;
        f32x16 dAr[K];
        if (!first && tile_on) {
#pragma unroll
          for (int d = 0; d < K; ++d) dAr[d] = dArT[T][d];
        }
        f32x16 dagg[K];
        float dxacc[K][DIM];
#pragma unroll
        for (int d = 0; d < K; ++d) {
          dagg[d] = zero16;
#pragma unroll
          for (int k = 0; k < DIM; ++k) dxacc[d][k] = 0.f;
        }
        const int cbase = col[T] - nodei[T];
        // the edge sweep of one tile, specialised per layer kind (0 first, 1 middle, 2 last): record size, records per
        // item and the tangent arithmetic are compile-time in each copy
        auto run_edges = [&](auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
        constexpr bool first = KIND == 0, last = KIND == 2;
        constexpr int nv = (KIND == 1) ? 4 : 1, erec = nv * 1024 + 512, epi = S / (nv * 4 + 2);
        for (int dd0 = 1; dd0 < N; dd0 += epi) {
          begin_item();  // the barrier also orders this layer's posc / pos0 / dPB writes before their first use
#pragma unroll
          for (int e = 0; e < epi; ++e) {
            const int dd = dd0 + e;
            if (!tile_on || dd >= N) break;
            const float* rec = item + e * erec;
            int j = nodei[T] + dd;
            j = (j >= N) ? j - N : j;
            const int cj = (col[T] < ncol) ? cbase + j : col[T];
            const float* sc = rec + nv * 1024;
            const float cs = sc[lane], dcs_f = sc[64 + lane];
            float df[DIM], e0[DIM], inv, hsq;
            if constexpr (KIND == 1) {  // middle-layer records carry the edge's geometry (written once by the primal launch)
              const f32x4 ga = *reinterpret_cast<const f32x4*>(sc + 256 + cl * 8);
              const f32x4 gb = *reinterpret_cast<const f32x4*>(sc + 256 + cl * 8 + 4);
              df[0] = ga.x; e0[0] = gb.x;
              df[1] = ga.y; e0[1] = gb.y;
              if constexpr (DIM > 2) { df[DIM - 1] = ga.z; e0[DIM - 1] = gb.z; }
              inv = ga.w;
              hsq = gb.w;
            } else {  // first / last layer records: slots 2-3 and 6-7
              const f32x4 ga = *reinterpret_cast<const f32x4*>(sc + 128 + cl * 4);
              const f32x4 gb = *reinterpret_cast<const f32x4*>(sc + 384 + cl * 4);
              df[0] = ga.x; e0[0] = gb.x;
              df[1] = ga.y; e0[1] = gb.y;
              if constexpr (DIM > 2) { df[DIM - 1] = ga.z; e0[DIM - 1] = gb.z; }
              inv = ga.w;
              hsq = gb.w;
            }
            float u[DIM];
#pragma unroll
            for (int k = 0; k < DIM; ++k) u[k] = df[k] * inv;
            float ddf[K][DIM], dradial[K], dea[K];
#pragma unroll
            for (int d = 0; d < K; ++d) {
              // 2 ([i == dnode] - [j == dnode]): twice the input displacement of this edge along the unit direction
              const float sij = si[T][d] - ((valid[T] && j == dnode[d]) ? 2.0f : 0.0f);
              if constexpr (first) {  // the positions entering the first layer ARE the input: no table read
                float dfk = df[0];
#pragma unroll
                for (int k = 1; k < DIM; ++k) dfk = (dk[d] == k) ? df[k] : dfk;
                dradial[d] = dfk * sij;
#pragma unroll
                for (int k = 0; k < DIM; ++k) ddf[d][k] = (dk[d] == k) ? 0.5f * sij : 0.0f;
              } else {
                const float* dposcur = dposb + (2 * d + cur) * C::POS_F;
                dradial[d] = 0.f;
#pragma unroll
                for (int k = 0; k < DIM; ++k) {
                  ddf[d][k] = dposi[T][d][k] - dposcur[cj * DIM + k];
                  dradial[d] = fmaf(df[k], ddf[d][k], dradial[d]);
                }
                dradial[d] *= 2.0f;
              }
              // tangent of the frozen edge attribute |x_i - x_j|^2 along the unit direction (node dnode, component dk):
              // 2 e0[dk] ([i == dnode] - [j == dnode]) -- no per-direction table of the input displacement
              float e0k = e0[0];
#pragma unroll
              for (int k = 1; k < DIM; ++k) e0k = (dk[d] == k) ? e0[k] : e0k;
              dea[d] = e0k * sij;
            }
            float dcs[K];
            if constexpr (first) {
              const f32x16 dmu = slot_vec16(rec, 0, lane);
              const float vcdmu = sc[256 + lane];
#pragma unroll
              for (int d = 0; d < K; ++d) {
                const float w_ = dradial[d] * aggw;
#pragma unroll
                for (int r = 0; r < 16; ++r) dagg[d][r] = fmaf(dmu[r], w_, dagg[d][r]);
                dcs[d] = dcs_f * (dradial[d] * vcdmu);
              }
            } else if constexpr (last) {
              const f32x16 qv = slot_vec16(rec, 0, lane);
              const float qr = sc[256 + lane], qe = sc[320 + lane];
#pragma unroll
              for (int d = 0; d < K; ++d) {
                const f32x16 dz1 = dAr[d] + lds_vec16(dPB + d * C::PB_F + cj * PBS + hh * 16);
                const float sdot = xhalf_sum(dot16(qv, dz1));
                dcs[d] = dcs_f * fmaf(qr, dradial[d], fmaf(qe, dea[d], sdot));
              }
            } else {
              const float datt_f = sc[192 + lane];  // 1 - att: the record's g2 carries the gate
              const float* wattp = vecs + VEC_EMB_F + l * VEC_LAYER_F + hh * 16 + V_WATT * EH;
              // the K chains side by side: each record vector is read from the ring once, the chains' matrix
              // instructions and LDS latencies overlap
              f32x16 dz[K];
              {
                const f32x16 g1 = slot_vec16(rec, 0, lane);
#pragma unroll
                for (int d = 0; d < K; ++d) {
                  dz[d] = dAr[d] + lds_vec16(dPB + d * C::PB_F + cj * PBS + hh * 16);
                  dz[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re, DIV_ST * (hh ? dea[d] : dradial[d]), dz[d], 0, 0, 0);
                  dz[d] *= g1;
                }
              }
              u32x4 xs[K][2][2];
              // issue priority (round 5, as in egnn_kernel.hip's edge): the operand split is 32 K independent half-rate
              // instructions, everything around it waits for single instructions to become ready (ring reads, the matrix
              // chain, the dot products).  At the lower priority it fills the slots the SIMD partner's dependent phases
              // leave instead of taking every other one: 19.94 -> 19.05 ms per trace, same bits
              // (profiles/r05_tangent_issue_priority.txt)
              __builtin_amdgcn_s_setprio(0);
#pragma unroll
              for (int d = 0; d < K; ++d) WFrag<2>::split(dz[d], xs[d]);
              __builtin_amdgcn_s_setprio(1);
#pragma unroll
              for (int d = 0; d < K; ++d) dz[d] = w2f.mul_split(xs[d], zero16);
              {
                const f32x16 g2s = slot_vec16(rec, 1, lane);
#pragma unroll
                for (int d = 0; d < K; ++d) dz[d] *= g2s;  // att dm2: dm without the gate's own tangent
              }
              float datt[K];
              if (p.attention) {
                const f32x16 v_watt = lds_vec16(wattp);
#pragma unroll
                for (int d = 0; d < K; ++d) datt[d] = datt_f * xhalf_sum(dot16(v_watt, dz[d]));
              }
              if (p.attention) {
                const f32x16 m2 = slot_vec16(rec, 2, lane);
#pragma unroll
                for (int d = 0; d < K; ++d)
#pragma unroll
                  for (int r = 0; r < 16; ++r) dz[d][r] = fmaf(datt[d], m2[r], dz[d][r]);
              }
              {
                const f32x16 vc = slot_vec16(rec, 3, lane);
#pragma unroll
                for (int d = 0; d < K; ++d) {
#pragma unroll
                  for (int r = 0; r < 16; ++r) dagg[d][r] = fmaf(dz[d][r], aggw, dagg[d][r]);
                  dcs[d] = dcs_f * xhalf_sum(dot16(vc, dz[d]));
                }
              }
            }
#pragma unroll
            for (int d = 0; d < K; ++d) {
              const float dnrm = dradial[d] * hsq;
#pragma unroll
              for (int k = 0; k < DIM; ++k) {
                const float du = (ddf[d][k] - u[k] * dnrm) * inv;
                dxacc[d][k] = fmaf(du, cs, fmaf(u[k], dcs[d], dxacc[d][k]));
              }
            }
          }
          end_item();
        }
        };
        // (its item loop must stay a loop: unrolled four times -- the compiler's choice -- the kernel grew from 5 600 to
        // 9 200 instructions and lost 0.7 ms; `#pragma unroll 1` below)
        constexpr bool VFIRST = true;
        if (first && VFIRST) {
          // First layer of a unit direction (dh = 0, d pos = the unit vector of node i0): only the edges that touch i0
          // have a tangent -- the row (i0, j) and the column (j, i0), 2 (N - 1) of N (N - 1).  Lanes re-mapped as in the
          // last layer: virtual column = (walker, direction, record of the item, side); side 0 = edge (i0, i0 + dd),
          // side 1 = edge (i0 - dd, i0).  What an edge gives its node a -- the aggregate's tangent (16 features per
          // half) and the position tangent -- goes to per-wave tables (the partner table is idle in this layer, the
          // "next" position table is this layer's output anyway): side 1 hits every node a != i0 exactly once, side 0
          // sums its records in registers and writes node i0 at the end.  Then the columns read their rows back.
          const int f_e = cl % EPI2, f_s = (cl / EPI2) & 1, f_d = (cl / (2 * EPI2)) % K, f_w = cl / (2 * EPI2 * K);
          static_assert(!VFIRST || G * K * 2 * EPI2 <= 32, "virtual columns of the first layer");
          int f_i0 = 0, f_k0 = 0;
#pragma unroll
          for (int q = 0; q < K; ++q) {
            f_i0 = (f_d == q) ? dnode[q] : f_i0;
            f_k0 = (f_d == q) ? dk[q] : f_k0;
          }
          const bool f_on = tile_on && f_w < nwalk && f_w < G && f_i0 >= 0;
          float* aggT = dPB + f_d * C::PB_F + hh * 16;                   // [column][PBS]: tangent of the aggregate
          float* posT = dposb + (2 * f_d + (cur ^ 1)) * C::POS_F;       // [column][DIM]: tangent positions leaving the layer
          f32x16 accA = zero16;
          float accX[DIM];
#pragma unroll
          for (int k = 0; k < DIM; ++k) accX[k] = 0.f;
#pragma unroll 1
          for (int dd0 = 1; dd0 < N; dd0 += EPI2) {
            begin_item();
            const int dd = dd0 + f_e;
            int na = f_s ? f_i0 - dd : f_i0;  // the edge's own node
            na = (na < 0) ? na + N : na;
            const int ca = f_w * N + na;
            const bool on = f_on && dd < N && (ca >> 5) == T;
            const int c = on ? (ca & 31) : 0;
            const float* rec = item + f_e * (1024 + 512);
            const float* sc = rec + 1024;
            const float cs = sc[c], dcs_f = sc[64 + c], vcdmu = sc[256 + c];
            const f32x4 ga = *reinterpret_cast<const f32x4*>(sc + 128 + c * 4);
            const float hsq = sc[384 + c * 4 + 3];
            float df[DIM];
            df[0] = ga.x;
            df[1] = ga.y;
            if constexpr (DIM > 2) df[DIM - 1] = ga.z;
            const float inv = ga.w;
            float dfk = df[0];
#pragma unroll
            for (int k = 1; k < DIM; ++k) dfk = (f_k0 == k) ? df[k] : dfk;
            const float sg = f_s ? -1.0f : 1.0f;  // d(pos_a - pos_b) along the direction: +1 if a is its node, -1 if b is
            const float dradial = 2.0f * sg * dfk;
            f32x16 v;
            {
              const f32x4* qp = reinterpret_cast<const f32x4*>(rec) + (hh * 32 + c);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4 t = qp[q * 64];
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
              }
            }
            v *= dradial;
            const float dcs = dcs_f * (dradial * vcdmu);
            const float dnrm = dradial * hsq;
            float x[DIM];
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              const float u = df[k] * inv;
              const float du = (((f_k0 == k) ? sg : 0.0f) - u * dnrm) * inv;
              x[k] = fmaf(du, cs, u * dcs);
            }
            if (on && f_s) {  // node a != i0: its one and only contribution
              f32x4* dst = reinterpret_cast<f32x4*>(aggT + ca * PBS);
#pragma unroll
              for (int q = 0; q < 4; ++q) dst[q] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
              if (hh == 0) {
#pragma unroll
                for (int k = 0; k < DIM; ++k) posT[ca * DIM + k] = x[k];
              }
            }
            if (on && !f_s) {
              accA += v;
#pragma unroll
              for (int k = 0; k < DIM; ++k) accX[k] += x[k];
            }
            end_item();
          }
          {  // node i0: the sum over its records (EPI2 neighbouring lanes)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float t = accA[r];
#pragma unroll
              for (int e = 1; e < EPI2; ++e) t += __shfl_down(accA[r], e, 64);
              accA[r] = t;
            }
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              float t = accX[k];
#pragma unroll
              for (int e = 1; e < EPI2; ++e) t += __shfl_down(accX[k], e, 64);
              accX[k] = t;
            }
            const int c0g = f_w * N + f_i0;
            if (f_on && !f_s && f_e == 0 && (c0g >> 5) == T) {
              f32x4* dst = reinterpret_cast<f32x4*>(aggT + c0g * PBS);
#pragma unroll
              for (int q = 0; q < 4; ++q) dst[q] = f32x4{accA[4 * q], accA[4 * q + 1], accA[4 * q + 2], accA[4 * q + 3]};
              if (hh == 0) {
#pragma unroll
                for (int k = 0; k < DIM; ++k) posT[c0g * DIM + k] = ((f_k0 == k) ? 1.0f : 0.0f) + accX[k];
              }
            }
          }
          wave_lds_fence();
          if (tile_on) {  // back in the column mapping
#pragma unroll
            for (int d = 0; d < K; ++d) {
              const bool has = valid[T] && d < myndir;
              dagg[d] = has ? lds_vec16(dPB + d * C::PB_F + col[T] * PBS + hh * 16) : zero16;
#pragma unroll
              for (int k = 0; k < DIM; ++k)
                dposi[T][d][k] = has ? dposb[(2 * d + (cur ^ 1)) * C::POS_F + col[T] * DIM + k] : 0.0f;
            }
          }
        } else if (last) {
          int v_e, v_d, v_w, v_i0, v_k0, v_col0, v_c0;
          bool v_on;
          vcol(v_e, v_d, v_w, v_i0, v_k0, v_on, v_col0, v_c0);
          // the virtual column's own Wa dh_i (held by lane (c0, half) of its direction's registers)
          const bool mine = v_on && (v_col0 >> 5) == T && tile_on;
          f32x16 dAv = zero16;
#pragma unroll
          for (int q = 0; q < K; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float t = __shfl(dAr[q][r], v_c0 + 32 * hh, 64);  // (ds_bpermute)
              dAv[r] = (v_d == q) ? t : dAv[r];
            }
          const float* vdpos = dposb + (2 * v_d + cur) * C::POS_F;
          const float* vdPB = dPB + v_d * C::PB_F + hh * 16;
#pragma unroll 1
          for (int dd0 = 1; dd0 < N; dd0 += EPI2) {
            begin_item();
            const int dd = dd0 + v_e;
            const bool on = mine && dd < N;
            int j = v_i0 + (on ? dd : 0);
            j = (j >= N) ? j - N : j;
            const int cj = on ? v_w * N + j : 0;
            const float* rec = item + v_e * (1024 + 512);
            const float* sc = rec + 1024;
            const float cs = sc[v_c0], dcs_f = sc[64 + v_c0], qr = sc[256 + v_c0], qe = sc[320 + v_c0];
            const f32x4 ga = *reinterpret_cast<const f32x4*>(sc + 128 + v_c0 * 4);
            const f32x4 gb = *reinterpret_cast<const f32x4*>(sc + 384 + v_c0 * 4);
            float df[DIM], e0[DIM];
            df[0] = ga.x; e0[0] = gb.x;
            df[1] = ga.y; e0[1] = gb.y;
            if constexpr (DIM > 2) { df[DIM - 1] = ga.z; e0[DIM - 1] = gb.z; }
            const float inv = ga.w, hsq = gb.w;
            float ddf[DIM], dradial = 0.f, e0k = e0[0];
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              ddf[k] = vdpos[v_col0 * DIM + k] - vdpos[cj * DIM + k];
              dradial = fmaf(df[k], ddf[k], dradial);
              if (k > 0) e0k = (v_k0 == k) ? e0[k] : e0k;
            }
            dradial *= 2.0f;
            const float dea = 2.0f * e0k;  // i is the direction's node, j is not
            f32x16 qv;
            {
              const f32x4* qp = reinterpret_cast<const f32x4*>(rec) + (hh * 32 + v_c0);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4 t = qp[q * 64];
                qv[4 * q] = t.x; qv[4 * q + 1] = t.y; qv[4 * q + 2] = t.z; qv[4 * q + 3] = t.w;
              }
            }
            const f32x16 dz1 = dAv + lds_vec16(vdPB + cj * PBS);
            const float sdot = xhalf_sum(dot16(qv, dz1));
            const float dcs = dcs_f * fmaf(qr, dradial, fmaf(qe, dea, sdot));
            const float dnrm = dradial * hsq;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              const float u = df[k] * inv;
              const float du = (ddf[k] - u * dnrm) * inv;
              const float t = fmaf(du, cs, u * dcs);
              vacc[k] += on ? t : 0.0f;
            }
            end_item();
          }
        } else if (first) run_edges(std::integral_constant<int, 0>{});
        else run_edges(std::integral_constant<int, 1>{});
        if (tile_on && !last && !(first && VFIRST)) {
#pragma unroll
          for (int k = 0; k < DIM; ++k)
#pragma unroll
            for (int d = 0; d < K; ++d) {
              dposi[T][d][k] += dxacc[d][k];
              if (hh == 0) dposb[(2 * d + (cur ^ 1)) * C::POS_F + col[T] * DIM + k] = dposi[T][d][k];
            }
        }
        const int nnm = (l == L - 1) ? 0 : (first ? 2 : 3);  // node-model matrices ahead of gn
        begin_item();
        if (tile_on && l != L - 1) {
          const f32x16 gn = slot_vec16(item, nnm, lane);
          f32x16 dzn[K];
          if (!first) {
            const WFrag<2> wn = slot_wfrag(item, 0, lane);
#pragma unroll
            for (int d = 0; d < K; ++d) dzn[d] = wn.mul(dhf[T][d], zero16);
          } else {
#pragma unroll
            for (int d = 0; d < K; ++d) dzn[d] = zero16;
          }
          {
            const WFrag<2> wn = slot_wfrag(item, nnm - 2, lane);
#pragma unroll
            for (int d = 0; d < K; ++d) dzn[d] = wn.mul(dagg[d], dzn[d]);
          }
          const WFrag<2> wn = slot_wfrag(item, nnm - 1, lane);
#pragma unroll
          for (int d = 0; d < K; ++d) {
            dzn[d] *= gn;
            f32x16 dho = wn.mul(dzn[d], zero16);
            dho *= F16_UNSCALE;
            dhf[T][d] += dho;
          }
        }
        end_item();
      }
      cur ^= 1;
    };
    layer(0, std::integral_constant<int, 0>{});
    for (int l = 1; l < L - 1; ++l) layer(l, std::integral_constant<int, 1>{});
    layer(L - 1, std::integral_constant<int, 2>{});  // (L >= 2: divshr_fits)

    // epilogue: as egnn_div_fast_kernel, per wave for its own directions, then the block's waves are summed in order
    float* tsl = dPB + K * C::POS_F;   // [G][K]
    wave_lds_fence();
    {
      int v_e, v_d, v_w, v_i0, v_k0, v_col0, v_c0;
      bool v_on;
      vcol(v_e, v_d, v_w, v_i0, v_k0, v_on, v_col0, v_c0);
      // re-mapped last layer: the records of a virtual column's (walker, direction) sit in EPI2 neighbouring lanes; the
      // output tangent is what entered the last layer (table of the previous layer, `cur` has moved on) + their sum
      float tot[DIM];
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        tot[k] = vacc[k];
#pragma unroll
        for (int e = 1; e < EPI2; ++e) tot[k] += __shfl_down(vacc[k], e, 64);
      }
      if (lane < G * K) tsl[lane] = 0.f;
      wave_lds_fence();
      if (v_on && hh == 0 && v_e == 0) {
        const float* vdpos = dposb + (2 * v_d + (cur ^ 1)) * C::POS_F;
        float tk = tot[0], pk = vdpos[v_col0 * DIM];
#pragma unroll
        for (int k = 1; k < DIM; ++k) {
          tk = (v_k0 == k) ? tot[k] : tk;
          pk = (v_k0 == k) ? vdpos[v_col0 * DIM + k] : pk;
        }
        const float dF = (pk + tk) - 1.0f;  // d(pos^L - pos^0) of the direction's own coordinate
        tsl[v_w * K + v_d] = fmaf(cstab[C::NCOLP + v_col0], dF, cstab[v_col0]);
      }
    }
    wave_lds_fence();
    if (lane < G) {
      float sum = 0.f;
#pragma unroll
      for (int d = 0; d < K; ++d) sum += tsl[lane * K + d];
      comb[wave * G + lane] = sum;
    }
    lds_barrier();
    if (wave == 0 && lane < G) {
      float sum = 0.f;
      for (int w = 0; w < NW; ++w) sum += comb[w * G + lane];
      res[nres * G + lane] = (lane < nwalk) ? sum : -0.0f;
      if (lane == 0) resw[nres] = (int)walker0;
    }
    ++nres;
    if (nres == C::RES_G) {
      wave_lds_fence();
      flush();
    }
  }
  wave_lds_fence();
  flush();
}

struct DivShape {
  int n, dim, G, waves, K;
  void (*kernel)(DivParams);
  void (*fast)(DivParams);
  size_t (*lds_bytes)(int);
  int occ = 1;  // blocks per CU the fast kernel is built for (grid cap)
};
template <int N, int DIM, int G, int WAVES, int K, int DAL>
static size_t div_lds_bytes_of(int L) { return DivCfg<N, DIM, G, WAVES, K, DAL>::lds_bytes(L); }
#define PITA_DIV_SHAPE(N, DIM, G, WAVES, K) \
  DivShape { N, DIM, G, WAVES, K, egnn_div_kernel<N, DIM, G, WAVES, K>, egnn_div_fast_kernel<N, DIM, G, WAVES, K, 1>, \
             div_lds_bytes_of<N, DIM, G, WAVES, K, 1> }
// fast kernel only, Wa dh_i in registers (frees LDS for a fourth direction); the repair kernel is the K-direction
// kernel of kDivShapes, which accepts any ndir <= its own K -- so these are used for the fast launch only
#define PITA_DIV_FAST_SHAPE(N, DIM, G, WAVES, K) \
  DivShape { N, DIM, G, WAVES, K, nullptr, egnn_div_fast_kernel<N, DIM, G, WAVES, K, 0>, \
             div_lds_bytes_of<N, DIM, G, WAVES, K, 0> }
static const DivShape kDivShapes[] = {
    PITA_DIV_SHAPE(4, 2, 8, 4, 3),
    PITA_DIV_SHAPE(13, 3, 2, 4, 3),
    PITA_DIV_SHAPE(22, 3, 1, 4, 3),
    PITA_DIV_SHAPE(55, 3, 1, 4, 1),
};
// measured for LJ13 at 65 536 walkers, all 39 directions: K = 2: 117.9 ms, K = 3: 100.6 ms (93.3 ms with Wa dh_i parked
// in LDS), K = 4: 102.8 ms (more register shuffling; no longer fits in LDS), 39 single-direction JVP launches: 126 ms.
// PITA_DIV_K selects an alternative for experiments.
static const DivShape kDivAlt[] = {PITA_DIV_SHAPE(13, 3, 2, 4, 2)};
// cache writers for the block-shared tangent kernel: the first launch of a trace carries NO direction (round 4; a
// direction costs 0.9 ms in this one-wave-per-SIMD launch and 0.17 ms in a tangent-only launch, which take 16 each: 0 + 13
// + 13 + 13 for LJ13; rounds 2-3: 1 + 13 + 13 + 12) and marks out-of-range walkers by their primal
// (measured with two blocks per CU, i.e. two waves per SIMD at 256 registers: 976 B/lane of scratch, the launch takes
// 14 ms instead of 5 -- the primal's adjoint factors plus one tangent chain need the 492 registers it uses)
#define PITA_DIV_WRITER_SHAPE(N, DIM, G, WAVES) \
  DivShape { N, DIM, G, WAVES, 0, nullptr, egnn_div_fast_kernel<N, DIM, G, WAVES, 0, 0, 1>, \
             div_lds_bytes_of<N, DIM, G, WAVES, 0, 0>, 1 }
static const DivShape kDivWriters[] = {PITA_DIV_WRITER_SHAPE(13, 3, 2, 4), PITA_DIV_WRITER_SHAPE(22, 3, 1, 4)};
static const DivShape* find_div_writer(int n, int dim) {
  static const bool off = getenv("PITA_DIV_NOWRITER") != nullptr;  // development aid
  if (off) return nullptr;
  for (const auto& c : kDivWriters)
    if (c.n == n && c.dim == dim) return &c;
  return nullptr;
}
static const DivShape* find_div_shape(int n, int dim) {
  static const int altk = getenv("PITA_DIV_K") ? atoi(getenv("PITA_DIV_K")) : 0;
  if (altk)
    for (const auto& c : kDivAlt)
      if (c.n == n && c.dim == dim && c.K == altk) return &c;
  for (const auto& c : kDivShapes)
    if (c.n == n && c.dim == dim) return &c;
  return nullptr;
}
// f16 kernel with more directions per launch than the bf16x3 kernel's LDS budget allows (Wa dh_i in registers);
// PITA_DIV_FAST_K=0 falls back to the bf16x3 kernel's K (development aid)
// tangent-only kernels (primal cache): same G / WAVES as the fast kernel of the particle system
struct DivTanShape {
  int n, dim, G, waves, K;
  int shared;              // 1: block-shared kernel (`waves` waves on one walker group, waves x K directions per launch)
  void (*kernel)(DivParams);
  size_t (*lds_bytes)(int);
  size_t (*group_f)(int);  // cache floats per walker group
  bool (*fits)(int);       // block-shared kernel: the network depth fits its LDS budget and piece table
};
template <int N, int DIM, int G, int WAVES, int K>
static size_t divtan_lds_of(int L) { return DivTanCfg<N, DIM, G, WAVES, K>::lds_bytes(L); }
template <int N, int DIM, int G>
static size_t divcache_group_f(int L) { return DivCache<N, DIM, (G * N + 31) / 32>::group_f(L); }
#define PITA_DIVTAN_SHAPE(N, DIM, G, WAVES, K) \
  DivTanShape { N, DIM, G, WAVES, K, 0, egnn_div_tangent_kernel<N, DIM, G, WAVES, K>, divtan_lds_of<N, DIM, G, WAVES, K>, \
                divcache_group_f<N, DIM, G>, nullptr }
// items of one group's sweep (as the kernel's piece table lays them out) against the table's capacity
template <int N, int DIM, int G, int NW, int K>
static bool divshr_fits(int L) {
  using C = DivShrCfg<N, DIM, G, NW, K>;
  using CA = DivCache<N, DIM, C::NT>;
  if (L > SHR_LMAX || L < 2) return false;  // (a one-layer network has no "last" layer of the re-mapped kind)
  long long items = 0;
  for (int l = 0; l < L; ++l) {
    const int epi = SHR_S / (int)(CA::edge_f(l, L) / 256);
    items += 1 + (long long)C::NT * ((N - 1 + epi - 1) / epi + 1);
  }
  return items * (C::COMPACT ? 2 : SHR_S) <= C::MAX_P;
}
template <int N, int DIM, int G, int NW, int K>
static size_t divshr_lds_of(int L) { return DivShrCfg<N, DIM, G, NW, K>::lds_bytes(L); }
#define PITA_DIVSHR_SHAPE(N, DIM, G, NW, K) \
  DivTanShape { N, DIM, G, NW, K, 1, egnn_div_tangent_shared_kernel<N, DIM, G, NW, K>, divshr_lds_of<N, DIM, G, NW, K>, \
                divcache_group_f<N, DIM, G>, divshr_fits<N, DIM, G, NW, K> }
// LJ13, all 39 directions at 65 536 walkers (first launch 5.7 ms incl. the 12 GB cache write): K = 3: 34.9 ms, K = 4:
// 32.9 ms (9 launches of 3.0 ms = 4 TB/s of cache reads), K = 5: 39.6 ms, K = 6: 44.0 ms (528 / 860 B/lane of scratch);
// without the cache (13 fast launches): 59-62 ms.  Also measured: the four waves of a block sharing ONE walker group with
// K = 2 directions each at two waves per SIMD, every wave streaming the same records and counting on the L2 for the
// repeats: 46.4 ms -- each wave's stream goes to HBM (5 launches of 8 ms = 4 x 12 GB at 5.5 TB/s); sharing would have to
// be explicit (records staged once per block in LDS).
// Block-shared kernel (LDS-DMA ring), same batch: (8 waves, K = 2) 26.3 ms, (12 waves, K = 1) 25.6-26.3 ms, (6 waves, K = 2)
// 28.5 ms: three launches of 6.3-6.8 ms that stream the cache once per 16 / 12 directions.  Per launch the LDS pipe is
// busy ~3 ms (every wave reads the records it shares: 24 KB per direction and middle-layer edge), the vector ALUs ~2 ms,
// the scalar ALU (one per CU: the ring bookkeeping of all waves) ~1 ms, the stream alone takes 2.1-2.5 ms; the waves
// run in lock step (one barrier per item), so these overlap only partly.
static const DivTanShape kDivTan[] = {
    PITA_DIVTAN_SHAPE(4, 2, 8, 4, 5),
    PITA_DIVSHR_SHAPE(13, 3, 2, 8, 2),
    PITA_DIVSHR_SHAPE(22, 3, 1, 8, 2),
    PITA_DIVSHR_SHAPE(55, 3, 1, 8, 1),
};
// wave-owned kernels for the systems above that default to the block-shared one (networks deeper than its LDS budget
// is sized for; PITA_DIV_TAN_ALT=1 selects them for A/B runs), then experiments (PITA_DIV_TAN_ALT=<index + 1>)
constexpr int kDivTanOwned = 3;  // the first kDivTanOwned entries are the wave-owned fallbacks
static const DivTanShape kDivTanAlt[] = {PITA_DIVTAN_SHAPE(13, 3, 2, 4, 4), PITA_DIVTAN_SHAPE(22, 3, 1, 4, 4),
                                         PITA_DIVTAN_SHAPE(55, 3, 1, 4, 3), PITA_DIVSHR_SHAPE(13, 3, 2, 12, 1)};
// (Round 6, measured and removed: seven / six waves per block -- every wave's tangent tables are 10.75 KB of LDS, so fewer waves
// leave FOUR ring slots instead of three -- LJ55 trace 610 -> 855 / 903 ms at 32 768 walkers, LJ13 18.4 -> 25.0 ms with seven
// waves: the launches are bound by the waves' own work, not by the ring's depth; profiles/r06_tangent_waves_per_block_ab.txt.)
static const DivTanShape* find_div_tan_shape(int n, int dim, int n_layers) {
  static const bool off = getenv("PITA_DIV_NOCACHE") != nullptr;  // development aid: A/B against the cache-free path
  if (off) return nullptr;
  static const int alt = getenv("PITA_DIV_TAN_ALT") ? atoi(getenv("PITA_DIV_TAN_ALT")) : 0;
  const int nalt = (int)(sizeof(kDivTanAlt) / sizeof(kDivTanAlt[0]));
  if (alt == 1) {
    for (int i = 0; i < kDivTanOwned; ++i)
      if (kDivTanAlt[i].n == n && kDivTanAlt[i].dim == dim) return &kDivTanAlt[i];
  } else if (alt > kDivTanOwned && alt <= nalt) {
    const auto& c = kDivTanAlt[alt - 1];
    if (c.n == n && c.dim == dim && (!c.shared || c.fits(n_layers))) return &c;
  }
  for (const auto& c : kDivTan)
    if (c.n == n && c.dim == dim) {
      if (!c.shared || c.fits(n_layers)) return &c;
      for (int i = 0; i < kDivTanOwned; ++i)
        if (kDivTanAlt[i].n == n && kDivTanAlt[i].dim == dim) return &kDivTanAlt[i];
      return nullptr;
    }
  return nullptr;
}

// measured: K = 4 for LJ13 (PITA_DIV_FAST_SHAPE(13, 3, 2, 4, 4)): 65.3 ms for the 39 directions against 61.9 ms with K = 3
// (800 B/lane of scratch: the fourth direction's state no longer fits the 512 registers) -- not instantiated
static const DivShape* find_div_fast_shape(int n, int dim) { return find_div_shape(n, dim); }
static bool div_fast_enabled(const pita_egnn* net) {
  static const bool force_slow = getenv("PITA_DIV_SLOW") != nullptr;  // development aid: A/B against the bf16x3 kernel
  return net->cfg.precision == 2 && !force_slow;
}

}  // namespace pita

using namespace pita;

extern "C" int pita_egnn_div_directions(const pita_egnn_t* net) {
  if (!net) return PITA_EINVAL;
  const DivShape* s = div_fast_enabled(net) ? find_div_fast_shape(net->cfg.n_particles, net->cfg.n_dim)
                                            : find_div_shape(net->cfg.n_particles, net->cfg.n_dim);
  return s ? s->K : PITA_EUNSUPPORTED;
}

// Matrix-core wave-instructions per walker for one FULL trace (all D directions) on the path pita_egnn_jacobian_trace takes
// with this handle, counted on the kernels' loop structure (bench.py: roofline of the debiased leg).  Fast kernel per column tile and launch: primal
// 6 (Wb) + 6 (Wa) + (N-1)(18 f16 + 1 f32 k-step) per layer, + 6 x 3 node-model GEMMs except in the last layer; tangents:
// first layer one shared W2 chain per edge (6), middle layers K (6 f16 + 1 f32) per edge, last layer one W2^T chain (6)
// per edge; per direction 6 x (Wb, Wa, Wn1a) from the second layer on and 6 x (Wn1b, Wn2) in every layer but the last.
extern "C" int pita_egnn_div_work(const pita_egnn_t* net, double* mfma16_per_walker, double* mfma32_per_walker) {
  PITA_REQUIRE(net && mfma16_per_walker && mfma32_per_walker, "pita_egnn_div_work: null argument");
  const DivShape* s = find_div_shape(net->cfg.n_particles, net->cfg.n_dim);
  if (!s) return fail(PITA_EUNSUPPORTED, "pita_egnn_div_work: no kernel for this particle system");
  const int N = s->n, L = net->cfg.n_layers, D = s->n * s->dim;
  int K = s->K;
  {  // the cached path's first launch may carry fewer directions (cache writer)
    const DivTanShape* ts0 = div_fast_enabled(net) ? find_div_tan_shape(N, s->dim, L) : nullptr;
    const DivShape* wr = (ts0 && ts0->shared) ? find_div_writer(N, s->dim) : nullptr;
    if (wr && wr->G == s->G && wr->waves == s->waves && D > s->K) K = wr->K;
  }
  double launches = K > 0 ? (D + K - 1) / K : 0;
  const double tiles_per_walker = (double)((s->G * N + 31) / 32) / s->G;
  double m16 = 0, m32 = 0, t16 = 0, t32 = 0;  // per tile: one K-direction launch with primal; one tangent-only direction
  const bool fast = div_fast_enabled(net);
  const bool cached = fast && find_div_tan_shape(N, s->dim, L) != nullptr && D > K;
  if (fast) {
    for (int l = 0; l < L; ++l) {
      const bool first = l == 0, lastl = l == L - 1;
      m16 += 12 + (N - 1) * 18.0 + (lastl ? 0 : 18);
      m32 += N - 1;
      if (first) m16 += (N - 1) * 6.0;
      else if (lastl) m16 += (N - 1) * 6.0;
      else { m16 += (N - 1) * 6.0 * K; m32 += (N - 1) * K; }
      if (!first) m16 += 12.0 * K + (lastl ? 0 : 6.0 * K);
      if (!lastl) m16 += 12.0 * K;
      // tangent-only kernel, per direction: Wb, Wa from the second layer on; W2 per edge in middle layers; node model
      if (!first) t16 += 12.0 + (lastl ? 0 : 6.0);
      if (!first && !lastl) { t16 += (N - 1) * 6.0; t32 += N - 1; }
      if (!lastl) t16 += 12.0;
    }
  } else {  // bf16x3 kernel: 12 MFMAs per dense layer, every tangent repeats the primal's GEMMs
    for (int l = 0; l < L; ++l) {
      const bool first = l == 0, lastl = l == L - 1;
      m16 += 24 + (N - 1) * 24.0 + (lastl ? 0 : 36);
      m32 += N - 1;
      m16 += K * ((N - 1) * 24.0 + (first ? 0 : 24) + (lastl ? 0 : (first ? 24 : 36)));
      m32 += K * (N - 1);
    }
  }
  if (cached) {  // pita_egnn_jacobian_trace: one launch with the primal, the other D - K directions from the cache
    *mfma16_per_walker = (m16 + (D - K) * t16) * tiles_per_walker;
    *mfma32_per_walker = (m32 + (D - K) * t32) * tiles_per_walker;
    return PITA_OK;
  }
  *mfma16_per_walker = m16 * tiles_per_walker * launches;
  *mfma32_per_walker = m32 * tiles_per_walker * launches;
  return PITA_OK;
}

// marks [B] + the flag word behind them (DivParams::bad_flag)
static int ensure_marks(pita_egnn_t* net, size_t B, hipStream_t st) {
  if (sizeof(int) * B > net->mark_bytes) {
    PITA_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(net->d_mark);
    net->d_mark = nullptr;
    net->mark_bytes = 0;
    PITA_HIP_CHECK(hipMalloc(&net->d_mark, sizeof(int) * B + 64));
    net->mark_bytes = sizeof(int) * B;
    PITA_HIP_CHECK(hipMemsetAsync(net->d_mark + B, 0, 64, st));
  }
  return PITA_OK;
}

static int div_launch(const DivShape* s, void (*kernel)(DivParams), pita_egnn_t* net, const DivParams& p, void* stream) {
  const size_t lds = s->lds_bytes(p.n_layers);
  PITA_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds));
  const long long ngroups = (p.B + s->G - 1) / s->G;
  long long want = (ngroups + s->waves - 1) / s->waves;
  const long long cap = (long long)net->n_cu * s->occ;  // one 4-wave block per CU (one wave per SIMD) unless built for more
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(s->waves * 64), lds, (hipStream_t)stream, p);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

extern "C" int pita_egnn_div_accumulate(pita_egnn_t* net, const float* h, const float* x, const float* beta, int dir0,
                                        int ndir, float* diag_acc, float* out, int64_t B, void* stream) {
  PITA_REQUIRE(net && B >= 0, "pita_egnn_div_accumulate: bad argument");
  if (B == 0) return PITA_OK;
  PitaDeviceGuard guard(net->device);
  PITA_REQUIRE(h && x && diag_acc, "pita_egnn_div_accumulate: null argument");
  PITA_REQUIRE(beta || net->cfg.in_node_nf == 1, "pita_egnn_div_accumulate: beta required for in_node_nf=2");
  const int D = net->cfg.n_particles * net->cfg.n_dim;
  const DivShape* s = find_div_shape(net->cfg.n_particles, net->cfg.n_dim);
  if (!s) return fail(PITA_EUNSUPPORTED, "pita_egnn_div_accumulate: no kernel for this particle system");
  const bool fast = div_fast_enabled(net);
  const DivShape* sf = fast ? find_div_fast_shape(net->cfg.n_particles, net->cfg.n_dim) : s;
  PITA_REQUIRE(dir0 >= 0 && ndir >= 1 && ndir <= sf->K && dir0 + ndir <= D, "pita_egnn_div_accumulate: directions out of range");
  DivParams p{};
  p.mats16 = net->d_mats16; p.mats16h = net->d_mats16h; p.vecs = net->d_vecs; p.vecs_h = net->d_vecs_h;
  p.vecs_div = net->d_vecs_div;
  p.n_layers = net->cfg.n_layers; p.in_nf = net->cfg.in_node_nf;
  p.attention = net->cfg.attention; p.tanh_on = net->cfg.tanh; p.feature_layout = net->cfg.feature_layout;
  p.coord_scale = net->cfg.coords_range / (float)net->cfg.n_layers;
  p.B = B; p.h = h; p.x = x; p.beta = beta; p.dir0 = dir0; p.ndir = ndir; p.diag_acc = diag_acc; p.out = out;
  if (!fast) return div_launch(s, s->kernel, net, p, stream);
  // f16 kernel first: it adds the finite terms and marks the walkers whose term was not; then the bf16x3 kernel
  // recomputes exactly the marked ones (in chunks of its own K directions)
  {
    const int rc0 = ensure_marks(net, (size_t)B, (hipStream_t)stream);
    if (rc0 != PITA_OK) return rc0;
  }
  p.mark = net->d_mark;
  p.bad_flag = net->d_mark + net->mark_bytes / sizeof(int);
  p.bad_seq = ++net->div_seq;
  int rc = div_launch(sf, sf->fast, net, p, stream);
  if (rc != PITA_OK) return rc;
  p.repair = 1;
  for (int d0 = 0; d0 < ndir; d0 += s->K) {
    p.dir0 = dir0 + d0;
    p.ndir = (ndir - d0) < s->K ? (ndir - d0) : s->K;
    p.out = (d0 == 0) ? out : nullptr;
    rc = div_launch(s, s->kernel, net, p, stream);
    if (rc != PITA_OK) return rc;
  }
  return PITA_OK;
}


// Exact trace of the denoiser Jacobian over ALL directions: trace[b] = sum_d (J_x D(h, x) e_d)_d (overwritten), optionally
// the denoiser itself.  Precision-2 handles with a tangent-only kernel take the primal-cache path: ONE launch of the fast
// kernel (its own directions, writes the per-edge primal factors), then tangent-only launches
// that stream the cache -- each followed by the bf16x3 repair pass for marked walkers.  The cache (≈180 KB per walker for
// LJ13) is owned by the handle; batches whose cache would exceed PITA_DIV_CACHE_GB (default 24) are processed in chunks
// of walkers.  Other handles loop pita_egnn_div_accumulate.
extern "C" int pita_egnn_jacobian_trace(pita_egnn_t* net, const float* h, const float* x, const float* beta, float* trace,
                                        float* denoiser_out, int64_t B, void* stream) {
  PITA_REQUIRE(net && B >= 0, "pita_egnn_jacobian_trace: bad argument");
  if (B == 0) return PITA_OK;
  PitaDeviceGuard guard(net->device);
  PITA_REQUIRE(h && x && trace, "pita_egnn_jacobian_trace: null argument");
  PITA_REQUIRE(beta || net->cfg.in_node_nf == 1, "pita_egnn_jacobian_trace: beta required for in_node_nf=2");
  const int n = net->cfg.n_particles, dim = net->cfg.n_dim, D = n * dim, L = net->cfg.n_layers;
  const DivShape* s = find_div_shape(n, dim);
  if (!s) return fail(PITA_EUNSUPPORTED, "pita_egnn_jacobian_trace: no kernel for this particle system");
  hipStream_t st = (hipStream_t)stream;
  PITA_HIP_CHECK(hipMemsetAsync(trace, 0, sizeof(float) * (size_t)B, st));
  const DivTanShape* ts = div_fast_enabled(net) ? find_div_tan_shape(n, dim, L) : nullptr;
  if (!ts || ts->G != s->G || (!ts->shared && ts->waves != s->waves) || D <= s->K) {
    const int K = pita_egnn_div_directions(net);
    for (int d0 = 0; d0 < D; d0 += K) {
      const int rc = pita_egnn_div_accumulate(net, h, x, beta, d0, (D - d0) < K ? (D - d0) : K, trace,
                                              d0 == 0 ? denoiser_out : nullptr, B, stream);
      if (rc != PITA_OK) return rc;
    }
    return PITA_OK;
  }
  // walkers per chunk: the cache of a chunk must fit the budget
  // budget of the primal cache: PITA_DIV_CACHE_GB, else 24 GB but never more than 60 % of what the device has free
  // right now plus what this handle already holds (other handles, the caller's tensors and a second process keep theirs)
  double budget_gb = 24.0;
  if (getenv("PITA_DIV_CACHE_GB")) {
    budget_gb = atof(getenv("PITA_DIV_CACHE_GB"));
  } else {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const double avail = 0.6 * ((double)free_b + (double)net->divcache_bytes) / 1e9;
      if (avail < budget_gb) budget_gb = avail;
    } else {
      (void)hipGetLastError();
    }
  }
  const double per_walker = 4.0 * (double)ts->group_f(L) / s->G;
  long long chunk = (long long)(budget_gb * 1e9 / per_walker);
  chunk = chunk / 1024 * 1024;
  if (chunk < 1024) chunk = 1024;
  if (chunk > B) chunk = B;
  // cache bytes a chunk of Bc walkers needs (grid of the fast kernel: the cache is indexed by wave and group)
  // the launch that writes the cache: the system's fast kernel, or a one-direction writer for the block-shared stream
  const DivShape* wr = ts->shared ? find_div_writer(n, dim) : nullptr;
  if (wr && (wr->G != s->G || wr->waves != s->waves)) wr = nullptr;
  const long long grid_cap = (long long)net->n_cu * (wr ? wr->occ : s->occ);
  auto cache_need = [&](long long Bc) -> size_t {
    const long long ngroups = (Bc + s->G - 1) / s->G;
    const long long want = (ngroups + s->waves - 1) / s->waves;
    const long long grid = want < grid_cap ? want : grid_cap;
    const long long total_waves = grid * s->waves;
    const long long quota = (Bc + total_waves - 1) / total_waves;
    return sizeof(float) * ts->group_f(L) * (size_t)(total_waves * ((quota + s->G - 1) / s->G));
  };
  // the buffer is allocated for the largest (= first) chunk; when the device cannot give that much, smaller chunks are
  // tried, and a handle that cannot even cache 1024 walkers falls back to the cache-free launches
  while (cache_need(chunk) > net->divcache_bytes) {
    PITA_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(net->d_divcache);
    net->d_divcache = nullptr;
    net->divcache_bytes = 0;
    const size_t need = cache_need(chunk);
    if (hipMalloc(&net->d_divcache, need) == hipSuccess) {
      net->divcache_bytes = need;
      break;
    }
    (void)hipGetLastError();  // out of memory: not sticky
    net->d_divcache = nullptr;
    if (chunk <= 1024) {
      const int K = pita_egnn_div_directions(net);
      for (int d0 = 0; d0 < D; d0 += K) {
        const int rc = pita_egnn_div_accumulate(net, h, x, beta, d0, (D - d0) < K ? (D - d0) : K, trace,
                                                d0 == 0 ? denoiser_out : nullptr, B, stream);
        if (rc != PITA_OK) return rc;
      }
      return PITA_OK;
    }
    chunk = (chunk / 2 + 1023) / 1024 * 1024;
  }
  for (long long b0 = 0; b0 < B; b0 += chunk) {
    const long long Bc = (B - b0) < chunk ? (B - b0) : chunk;
    DivParams p{};
    p.mats16 = net->d_mats16; p.mats16h = net->d_mats16h; p.vecs = net->d_vecs; p.vecs_h = net->d_vecs_h;
    p.vecs_div = net->d_vecs_div;
    p.n_layers = L; p.in_nf = net->cfg.in_node_nf;
    p.attention = net->cfg.attention; p.tanh_on = net->cfg.tanh; p.feature_layout = net->cfg.feature_layout;
    p.coord_scale = net->cfg.coords_range / (float)L;
    p.B = Bc; p.h = h + b0; p.x = x + b0 * D; p.beta = beta ? beta + b0 : nullptr; p.diag_acc = trace + b0;
    p.no_mean = 1;  // all D directions are summed: the mean-free projection's shares cancel (DivParams::no_mean)
    // grid of the fast / tangent kernels (identical: the cache is indexed by wave and group)
    const long long ngroups = (Bc + s->G - 1) / s->G;
    long long want = (ngroups + s->waves - 1) / s->waves;
    const long long grid = want < grid_cap ? want : grid_cap;
    const long long total_waves = grid * s->waves;
    const long long quota = (Bc + total_waves - 1) / total_waves;
    const long long groups_per_wave = (quota + s->G - 1) / s->G;
    PITA_REQUIRE(sizeof(float) * ts->group_f(L) * (size_t)(total_waves * groups_per_wave) <= net->divcache_bytes,
                 "pita_egnn_jacobian_trace: cache smaller than a chunk");
    {
      const int rc0 = ensure_marks(net, (size_t)Bc, st);
      if (rc0 != PITA_OK) return rc0;
    }
    p.mark = net->d_mark;
    p.bad_flag = net->d_mark + net->mark_bytes / sizeof(int);
    p.cache = net->d_divcache;
    p.cache_waves = total_waves;
    PITA_REQUIRE(total_waves * groups_per_wave < 0x7fffffffLL && Bc < 0x7fffffffLL,
                 "pita_egnn_jacobian_trace: too many walkers in one chunk");
    auto repair = [&](int dir0, int ndir, float* out) -> int {  // bf16x3 kernel for the marked walkers, its own K at a time
      DivParams r = p;
      r.repair = 1;
      r.cache = nullptr;
      if (ndir == 0) {  // a launch without directions: only the denoiser of the marked walkers is recomputed
        if (!out) return PITA_OK;
        r.dir0 = 0; r.ndir = 0; r.out = out;
        return div_launch(s, s->kernel, net, r, stream);
      }
      r.dir0 = dir0;
      r.ndir = ndir;
      r.nchunk = (ndir + s->K - 1) / s->K;
      r.out = out;
      return div_launch(s, s->kernel, net, r, stream);
    };
    // first launch: primal + its own directions, cache written
    const int first_k = wr ? wr->K : s->K;
    p.dir0 = 0; p.ndir = first_k < D ? first_k : D; p.out = denoiser_out ? denoiser_out + b0 * D : nullptr;
    p.bad_seq = ++net->div_seq;
    int rc = div_launch(wr ? wr : s, wr ? wr->fast : s->fast, net, p, stream);
    if (rc != PITA_OK) return rc;
    rc = repair(0, p.ndir, p.out);
    if (rc != PITA_OK) return rc;
    // remaining directions from the cache
    p.out = nullptr;
    const size_t lds = ts->lds_bytes(L);
    PITA_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(ts->kernel), lds));
    // full launches first, the remainder last (LJ13: 16, 16, 7).  Round 4 dealt the directions out in equal shares (13, 13,
    // 13); on the round-5 kernel full launches are 2 % faster per trace for LJ13 (18.35 vs 18.73 ms, four runs each on one
    // box) and the same for 22 and 55 particles (profiles/r05_tangent_issue_priority.txt)
    const int per_launch = ts->shared ? ts->K * ts->waves : ts->K;
    const long long tgrid = !ts->shared ? grid : (total_waves * groups_per_wave < (long long)net->n_cu
                                                      ? total_waves * groups_per_wave : (long long)net->n_cu);
    for (int d0 = first_k; d0 < D; d0 += per_launch) {
      p.dir0 = d0;
      p.ndir = (D - d0) < per_launch ? (D - d0) : per_launch;
      p.bad_seq = ++net->div_seq;
      hipLaunchKernelGGL(ts->kernel, dim3((unsigned)tgrid), dim3(ts->waves * 64), lds, st, p);
      PITA_LAUNCH_CHECK();
      rc = repair(d0, p.ndir, nullptr);
      if (rc != PITA_OK) return rc;
    }
  }
  return PITA_OK;
}
