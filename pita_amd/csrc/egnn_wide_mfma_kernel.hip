// EGNN backbone with hidden_nf <= 64 and static per-node features on the MATRIX pipe of gfx950 (MI355X).
//
// Replaces the same reference code as egnn_wide_kernel.hip (paths relative to /root/reference/pita/src/models/components/):
//   egnn_dynamics_ad2_cat.py:11-203   EGNN_dynamics_AD2_cat (configs/model/net/egnn_dynamics_ad2_cat.yaml: hidden 64 x 5
//                                     layers, one-hot atom-type node features concatenated with t and beta)
//   egnn.py:108-184, 187-346          EGNN.forward, E_GCL
//   score_net.py:13-43                EDM preconditioning (modes 1, 2)
//
// Mapping: egnn_kernel.hip's, with the hidden width doubled.  A wave owns G walkers = G N graph nodes = columns, packed
// into NT tiles of 32; lane l works on column (l & 31) and holds 32 of the 64 hidden features as two 16-register
// fragments (block b, feature 32 b + kfeat(r, l >> 5)): the C layout of v_mfma_f32_32x32x16_f16 for each 32-row block
// of the output, and -- with the weight blocks' k order permuted on the host -- the B layout of the next dense layer,
// so a 64 x 64 dense layer is 2 (output blocks) x 2 (input blocks) chains of 6 MFMAs on the f16 two-piece path of
// egnn_common.h (fp32-equivalent: WFrag<2>) with no data movement between layers.  Edges j = (i + dd) mod N, per-node
// sums in registers, partner terms Wb h_j and partner coordinates from a per-wave LDS table (row stride 68 floats).
// One wave per SIMD: the two per-edge matrices (W2, coordinate head: 128 registers of fragments) stay resident.
// An activation beyond the f16 range turns the walker's output non-finite; pita_egnn_wide_eval then recomputes exactly
// those walkers with the vector-pipe kernel (fp32 FMA chains), so the range costs time, never correctness.
#include "egnn_wide_mfma_common.h"

namespace pita {

// ATT / TANH: the network's attention gate and tanh-bounded coordinate head as compile-time switches: the edge loop is
// one basic block (with the branch-free tanh_select of egnn_common.h and the branch-free partner index below: 14.8 ->
// 13.9 ms per 65 536 forwards).
// Measured and dropped: the edge loop as a software pipeline (two edges in flight, the 24 MFMAs of a dense layer dealt
// out between the vector stage of the neighbouring edge with sched_group_barrier, 1 / 2 / 4 / 8 regions per stage):
// 14.7-15.4 ms -- a lone wave's stalls are dependent-instruction latencies, which the extra live state makes worse.
template <int N, int DIM, int G, int WAVES, bool ATT, bool TANH>
__global__ void __launch_bounds__(WAVES * 64, 1) egnn_wide64_kernel(Wide64Params p) {
  using C = Wide64Cfg<N, DIM, G, WAVES>;
  constexpr int NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = p.L;
  const int vec_f = C::vec_f(L);
  for (int i = threadIdx.x; i < W64_HEAD_F + L * W64_LAYER_F; i += WAVES * 64) lds[i] = p.vecs[i];
  float* est = lds + vec_f;  // [N][block][hh][r]
  for (int i = threadIdx.x; i < N * 64; i += WAVES * 64) est[i] = p.est[i];
  __syncthreads();  // the only workgroup barrier: vectors and the static embedding are shared by the block's waves

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane & 31, hh = lane >> 5;
  float* PB = est + N * 64 + wave * C::WAVE_F;
  float* posbuf0 = PB + C::PB_F;
  float* posbuf1 = posbuf0 + C::POS_F;
  float* pos0 = posbuf1 + C::POS_F;
  const f32x16 zero16 = {0};

  float* xbuf = pos0 + C::POS_F;  // the walkers' unscaled coordinates (parked in LDS across the layers and the steps)
  const bool smp = p.mode == 3;
  const long long ngroups = (p.B + G - 1) / G;
  for (long long g = (long long)blockIdx.x * WAVES + wave; g < ngroups; g += (long long)gridDim.x * WAVES) {
    const long long walker0 = g * G;
    const int nwalk = (int)((p.B - walker0) < G ? (p.B - walker0) : G);
    const int ncol = nwalk * N;
    const int ntile = (ncol + 31) >> 5;
    int col[NT], nodei[NT];
    bool valid[NT];
    int bad_from[NT];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      col[T] = T * 32 + cl;
      const int w = col[T] / N;
      nodei[T] = col[T] - w * N;
      valid[T] = col[T] < ncol;
      bad_from[T] = 0x7fffffff;
      const float* src = smp ? p.xs : p.x;
#pragma unroll
      for (int k = 0; k < DIM; ++k)
        if (hh == 0) xbuf[col[T] * DIM + k] = valid[T] ? src[(walker0 * N + col[T]) * DIM + k] : 0.f;
    }
    wave_lds_fence();
    const int nsteps = smp ? p.n_steps : 1;
    for (int step = 0; step < nsteps; ++step) {
    float c_s[NT], c_out[NT], hval[NT];
    f32x16 hfeat[NT][2];
    float g2 = 0.f, gamma = 0.f, dt = 0.f, noise_scale = 0.f, sqrt_dt = 0.f;
    if (smp) {  // wave-uniform per-step scalars through SGPRs (sde_integration.py: build_step_table)
      const float* st = p.step_tab + (size_t)step * PITA_STEP_STRIDE;
      auto uni = [&](int i) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, st[i]))); };
      g2 = uni(PITA_ST_G2); gamma = uni(PITA_ST_GAMMA); dt = uni(PITA_ST_DT);
      noise_scale = uni(PITA_ST_NOISE_SCALE); sqrt_dt = uni(PITA_ST_SQRT_DT);
    }
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      float c_in = 1.f, tfeat, bet;
      c_s[T] = 0.f; c_out[T] = 1.f; hval[T] = 1.f;
      if (smp) {
        const float* st = p.step_tab + (size_t)step * PITA_STEP_STRIDE;
        auto uni = [&](int i) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, st[i]))); };
        c_s[T] = uni(PITA_ST_CS); c_in = uni(PITA_ST_CIN); c_out[T] = uni(PITA_ST_COUT);
        tfeat = uni(PITA_ST_CNOISE); hval[T] = uni(PITA_ST_H); bet = p.has_beta ? uni(PITA_ST_BETA) : 0.f;
      } else {
        const int w = col[T] / N;
        const long long wid = valid[T] ? walker0 + w : p.B - 1;
        const float tv = p.t[wid];
        bet = p.has_beta ? p.beta[wid] : 0.f;
        tfeat = tv;
        if (p.mode != 0) {  // score_net.py:26-29
          hval[T] = tv;
          c_s[T] = 1.0f / (1.0f + tv);
          c_in = 1.0f / sqrtf(1.0f + tv);
          c_out[T] = sqrtf(tv) * c_in;
          tfeat = 0.125f * logf(tv);
        }
      }
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        const float ps = c_in * xbuf[col[T] * DIM + k];
        if (hh == 0) {
          pos0[col[T] * DIM + k] = ps;
          posbuf0[col[T] * DIM + k] = ps;
        }
      }
      // node features: embedding of [static one-hot features, t, beta] (egnn_dynamics_ad2_cat.py:157-184, egnn.py:179)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const f32x16 wt = lds_vec16(lds + b * 32 + hh * 16), wb = lds_vec16(lds + 64 + b * 32 + hh * 16);
        const f32x16 es = lds_vec16(est + nodei[T] * 64 + b * 32 + hh * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) hfeat[T][b][r] = fmaf(wt[r], tfeat, fmaf(wb[r], bet, es[r]));
      }
    }
    wave_lds_fence();

    float* poscur = posbuf0;
    float* posnext = posbuf1;
    for (int l = 0; l < L; ++l) {
      const unsigned* ml = p.m16h + (size_t)l * WM_COUNT * W64_MAT_W;
      const float* vbase = lds + W64_HEAD_F + l * W64_LAYER_F;
      const float* vl = vbase + hh * 16;
      const bool last = (l == L - 1);
      const float aggw = last ? 0.0f : 1.0f;  // the last layer's aggregate is dead (h_final is discarded)
      // ---- partner table PB[col] = Wb h_col
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        if (T >= ntile) continue;
        f32x16 pb[2] = {zero16, zero16};
        w64_mul_stream(ml, WM_WB, lane, hfeat[T], pb);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          pb[b] *= F16_UNSCALE;
          lds_store16(PB + col[T] * W64_PBS + b * 32 + hh * 16, pb[b]);
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

#pragma unroll
      for (int T = 0; T < NT; ++T) {
        if (T >= ntile) continue;
        f32x16 Ai[2] = {lds_vec16(vl + WV_B1 * 64), lds_vec16(vl + WV_B1 * 64 + 32)};
        w64_mul_stream(ml, WM_WA, lane, hfeat[T], Ai);
        Ai[0] *= F16_UNSCALE;
        Ai[1] *= F16_UNSCALE;
        f32x16 agg[2] = {zero16, zero16};
        float xacc[DIM], pown[DIM], p0own[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          xacc[k] = 0.f;
          pown[k] = poscur[col[T] * DIM + k];
          p0own[k] = pos0[col[T] * DIM + k];
        }
        const int cbase = col[T] - nodei[T];
        const int live = valid[T] ? 1 : 0;  // columns beyond the group's walkers pair with themselves (no select per edge)
        for (int dd = 1; dd < N; ++dd) {
          asm volatile("" ::: "memory");  // keep the per-edge LDS vector loads inside the loop
          int j = nodei[T] + dd * live;
          j = (j >= N) ? j - N : j;
          const int cj = cbase + j;
          float df[DIM], radial = 0.f, ea = 0.f;  // coord2radial (egnn.py E_GCL), frozen edge attribute (ad2_cat.py:186)
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            df[k] = pown[k] - poscur[cj * DIM + k];
            radial = fmaf(df[k], df[k], radial);
            const float e0 = p0own[k] - pos0[cj * DIM + k];
            ea = fmaf(e0, e0, ea);
          }
          const float geo = hh ? ea : radial;
          f32x16 m[2];
          m[0] = Ai[0] + lds_vec16(PB + cj * W64_PBS + hh * 16);
          m[1] = Ai[1] + lds_vec16(PB + cj * W64_PBS + 32 + hh * 16);
          m[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re0, geo, m[0], 0, 0, 0);
          m[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re1, geo, m[1], 0, 0, 0);
          silu16_out(m[0]);
          silu16_out(m[1]);
          f32x16 z[2] = {lds_vec16(vl + WV_B2 * 64), lds_vec16(vl + WV_B2 * 64 + 32)};
          w2f.mul(m, z);
          silu16_acc(z[0]);
          silu16_acc(z[1]);
          if (ATT) {
            const float s = dot16(lds_vec16(vl + WV_WATT * 64), z[0]) + dot16(lds_vec16(vl + WV_WATT * 64 + 32), z[1]);
            const float att = fast_sigmoid(xhalf_sum(s) + b_att);
            z[0] *= att;
            z[1] *= att;
          }
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) agg[b][r] = fmaf(z[b][r], aggw, agg[b][r]);
          f32x16 c1[2] = {lds_vec16(vl + WV_BC1 * 64), lds_vec16(vl + WV_BC1 * 64 + 32)};
          wc1f.mul(z, c1);
          silu16_acc(c1[0]);
          silu16_acc(c1[1]);
          float cs = xhalf_sum(dot16(lds_vec16(vl + WV_WC2 * 64), c1[0]) + dot16(lds_vec16(vl + WV_WC2 * 64 + 32), c1[1]));
          if (TANH) cs = tanh_select(cs) * p.coord_scale;
          const float inrm = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(radial + 1e-8f) + 1.0f);
#pragma unroll
          for (int k = 0; k < DIM; ++k) xacc[k] = fmaf(df[k] * inrm, cs, xacc[k]);
        }
#pragma unroll
        for (int k = 0; k < DIM; ++k)
          if (hh == 0) posnext[col[T] * DIM + k] = pown[k] + xacc[k];
        if (!last) {  // node model, recurrent
          f32x16 n1[2] = {lds_vec16(vl + WV_BN1 * 64), lds_vec16(vl + WV_BN1 * 64 + 32)};
          w64_mul_stream(ml, WM_WN1A, lane, hfeat[T], n1);
          w64_mul_stream(ml, WM_WN1B, lane, agg, n1);
          silu16_acc(n1[0]);
          silu16_acc(n1[1]);
          f32x16 o[2] = {lds_vec16(vl + WV_BN2 * 64), lds_vec16(vl + WV_BN2 * 64 + 32)};
          w64_mul_stream(ml, WM_WN2, lane, n1, o);
          hfeat[T][0] += o[0] * F16_UNSCALE;
          hfeat[T][1] += o[1] * F16_UNSCALE;
        }
      }
      wave_lds_fence();
      float* tmp = poscur; poscur = posnext; posnext = tmp;
    }

    // ---- vel = x_final - x, mean-free over the walker's particles; EDM combination for modes 1, 2
    float* scr = PB;
    float F[NT][DIM];
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        F[T][k] = poscur[col[T] * DIM + k] - pos0[col[T] * DIM + k];
        if (hh == 0) scr[col[T] * DIM + k] = F[T][k];
      }
    wave_lds_fence();
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const int cb = (col[T] < ncol) ? col[T] - nodei[T] : 0;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        float s = 0.f;
        for (int q = 0; q < N; ++q) s += scr[(cb + q) * DIM + k];
        F[T][k] -= s / (float)N;
      }
    }
    wave_lds_fence();
    if (!smp) {
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          float o = F[T][k];
          if (p.mode != 0) {
            const float xc = xbuf[col[T] * DIM + k];
            o = c_s[T] * xc + c_out[T] * o;             // denoiser (score_net.py:31-33)
            if (p.mode == 2) o = (o - xc) / hval[T];    // score (:19)
          }
          if (valid[T] && hh == 0) {
            p.out[(walker0 * N + col[T]) * DIM + k] = o;
            if (p.flag && !__builtin_isfinite(o)) *p.flag = 1;
          }
        }
    } else {
      // ---- reverse-SDE Euler-Maruyama update (sdes.py:119-122,250; sde_integration.py:347-348): the arithmetic of
      // ScoreNet's score + the integrator's drift + pita_em_step, so the fused and the per-step path agree to rounding
      float xn[NT][DIM];
      float st_d = 0.f, st_d2 = 0.f, st_n = 0.f, st_n2 = 0.f;
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        float xi[4] = {0.f, 0.f, 0.f, 0.f};
        const int w = col[T] / N;
        if (p.noise) {
#pragma unroll
          for (int k = 0; k < DIM; ++k)
            xi[k] = valid[T] ? p.noise[((size_t)step * p.B * N + (size_t)(walker0 * N + col[T])) * DIM + k] : 0.f;
        } else {
          philox_normal4(p.seed, p.walker_offset + (unsigned long long)(valid[T] ? walker0 + w : p.B - 1), p.step0 + step,
                         (uint32_t)nodei[T], xi);
        }
        // moments: a particle whose drift is not finite (an activation beyond the f16 range) is left to the repair
        // launch from this step on -- decided per particle, all components together (as egnn_kernel does)
        bool take = true;
        float drift[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const float xc = xbuf[col[T] * DIM + k];
          const float Dth = c_s[T] * xc + c_out[T] * F[T][k];
          drift[k] = gamma * (((Dth - xc) / hval[T]) * g2);
          take = take && __builtin_isfinite(drift[k]);
        }
        take = take && bad_from[T] == 0x7fffffff;
        if (!take && bad_from[T] == 0x7fffffff) bad_from[T] = step;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const float xc = xbuf[col[T] * DIM + k];
          const float dif = noise_scale * xi[k];
          if (p.stats_out && valid[T] && hh == 0 && take) {
            st_d += drift[k]; st_d2 = fmaf(drift[k], drift[k], st_d2);
            st_n += dif; st_n2 = fmaf(dif, dif, st_n2);
          }
          xn[T][k] = xc + (drift[k] * dt + (dif * sqrt_dt));
          if (hh == 0) scr[col[T] * DIM + k] = xn[T][k];
        }
      }
      if (p.stats_out) {
        double m4[4] = {(double)st_d, (double)st_d2, (double)st_n, (double)st_n2};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          for (int o = 32; o > 0; o >>= 1) m4[q] += __shfl_xor(m4[q], o, 64);
          if (lane == 0) atomicAdd(p.stats_out + (size_t)step * 4 + q, m4[q]);
        }
      }
      if (p.remove_mean) {
        wave_lds_fence();
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          const int cb = (col[T] < ncol) ? col[T] - nodei[T] : 0;
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            float s = 0.f;
            for (int q = 0; q < N; ++q) s += scr[(cb + q) * DIM + k];
            xn[T][k] -= s / (float)N;
          }
        }
        wave_lds_fence();
      }
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int k = 0; k < DIM; ++k)
          if (hh == 0) xbuf[col[T] * DIM + k] = xn[T][k];
    }
    wave_lds_fence();
    }  // steps
    if (smp) {
#pragma unroll
      for (int T = 0; T < NT; ++T) {
#pragma unroll
        for (int k = 0; k < DIM; ++k)
          if (valid[T] && hh == 0) {
            const float v = xbuf[col[T] * DIM + k];
            p.xs[(walker0 * N + col[T]) * DIM + k] = v;
            if (p.flag && !__builtin_isfinite(v)) *p.flag = 1;
          }
        if (p.bad_from && valid[T] && hh == 0) p.bad_from[walker0 * N + col[T]] = bad_from[T];
      }
    }
    wave_lds_fence();
  }
}

struct Wide64Shape {
  int n, dim, G, waves;
  void (*kernel[2][2])(Wide64Params);  // [attention][tanh]
  size_t (*lds_bytes)(int);
};
template <int N, int DIM, int G, int WAVES>
static size_t wide64_lds_of(int L) { return Wide64Cfg<N, DIM, G, WAVES>::lds_bytes(L); }
#define PITA_WIDE64_SHAPE(N, DIM, G, WAVES)                                                                         \
  Wide64Shape { N, DIM, G, WAVES,                                                                                  \
                {{egnn_wide64_kernel<N, DIM, G, WAVES, false, false>, egnn_wide64_kernel<N, DIM, G, WAVES, false, true>}, \
                 {egnn_wide64_kernel<N, DIM, G, WAVES, true, false>, egnn_wide64_kernel<N, DIM, G, WAVES, true, true>}}, \
                wide64_lds_of<N, DIM, G, WAVES> }
// the particle counts EGNN_dynamics_AD2_cat.get_h_initial knows (egnn_dynamics_ad2_cat.py:66-92): alanine dipeptide (22
// atoms: 4 walkers = 88 of 96 columns), tri- / tetra-alanine (33, 42 atoms: 2 walkers = 66 / 84 of 96), LJ13 (7 walkers =
// 91 of 96), LJ55 (1 walker = 55 of 64)
static const Wide64Shape kWide64Shapes[] = {PITA_WIDE64_SHAPE(22, 3, 4, 4), PITA_WIDE64_SHAPE(33, 3, 2, 4),
                                            PITA_WIDE64_SHAPE(42, 3, 2, 4), PITA_WIDE64_SHAPE(13, 3, 7, 4),
                                            PITA_WIDE64_SHAPE(55, 3, 1, 4)};
// Fewer walkers per wave for batches that leave SIMDs empty with the mapping above (4 096 alanine-dipeptide walkers are
// 1 024 groups of four = one wave per SIMD; below that, one walker per wave -- 22 of 32 columns -- fills the chip
// sooner).  Results do not depend on the grouping (columns are independent; tested bitwise).
static const Wide64Shape kWide64Small[] = {PITA_WIDE64_SHAPE(22, 3, 1, 4)};

static inline int kfeat64(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

int wide64_prepare(pita_egnn_wide* net, const float* w, const float* he) {
  const pita_egnn_wide_config& cfg = net->cfg;
  const Wide64Shape* shape = nullptr;
  for (const auto& s : kWide64Shapes)
    if (s.n == cfg.n_particles && s.dim == cfg.n_dim) shape = &s;
  if (!shape) return PITA_OK;
  const int H = cfg.hidden_nf, L = cfg.n_layers, ns = cfg.n_static, n = cfg.n_particles;
  const int nf = ns + 1 + (cfg.condition_beta ? 1 : 0);
  if (shape->lds_bytes(L) > 160 * 1024) return PITA_OK;  // deeper than the LDS holds: the vector-pipe kernel serves it
  const size_t n_m = (size_t)L * WM_COUNT * W64_MAT_W, n_v = W64_HEAD_F + (size_t)L * W64_LAYER_F;
  unsigned* hm = new unsigned[n_m]();
  float* hv = new float[n_v]();
  float* hes = new float[(size_t)n * 64]();
  const float kS = SILU_PRESCALE, kSi = 1.0f / SILU_PRESCALE, up = F16_SX * F16_SW, dn = 1.0f / F16_SX;
  auto f16_bits = [](float v) { _Float16 h = (_Float16)v; unsigned short u; memcpy(&u, &h, 2); return (unsigned)u; };
  // block (ob, kb) of F16_SW sc M[:, col0 : col0 + H] as a WFrag<2> fragment; rows / columns beyond H are zero
  auto pack_block = [&](unsigned* dst, const float* M, int ld, int col0, int ob, int kb, float sc) {
    for (int lane = 0; lane < 64; ++lane)
      for (int st = 0; st < 2; ++st)
        for (int qd = 0; qd < 4; ++qd) {
          unsigned pcs[2][2];
          for (int e = 0; e < 2; ++e) {
            const int row = ob * 32 + (lane & 31), kin = kb * 32 + kfeat64(8 * st + 2 * qd + e, lane >> 5);
            const float wv = (row < H && kin < H) ? F16_SW * (sc * M[(size_t)row * ld + col0 + kin]) : 0.f;
            const _Float16 w1 = (_Float16)wv;
            pcs[e][0] = f16_bits((float)w1);
            pcs[e][1] = f16_bits(wv - (float)w1);
          }
          for (int pc = 0; pc < 2; ++pc)
            dst[(((size_t)pc * 2 + st) * 64 + lane) * 4 + qd] = pcs[0][pc] | (pcs[1][pc] << 16);
        }
  };
  auto pack_mat = [&](unsigned* layer, int mat, const float* M, int ld, int col0, float sc) {
    for (int ob = 0; ob < 2; ++ob)
      for (int kb = 0; kb < 2; ++kb) pack_block(layer + ((size_t)mat * 4 + ob * 2 + kb) * MAT_WH, M, ld, col0, ob, kb, sc);
  };
  auto pack_vec = [&](float* dst, const float* v, int stride, float sc) {  // fragment order [block][hh][r]
    for (int b = 0; b < 2; ++b)
      for (int hh = 0; hh < 2; ++hh)
        for (int r = 0; r < 16; ++r) {
          const int f = b * 32 + kfeat64(r, hh);
          dst[b * 32 + hh * 16 + r] = f < H ? sc * v[(size_t)f * stride] : 0.f;
        }
  };
  const float* q = w;
  const float* emb_w = q; q += H * nf;
  q += H;             // embedding bias: inside he
  q += nf * H + nf;   // embedding_out: dead (h_final is discarded, egnn_dynamics_ad2_cat.py:187)
  pack_vec(hv, emb_w + ns, nf, 1.0f);
  if (cfg.condition_beta) pack_vec(hv + 64, emb_w + ns + 1, nf, 1.0f);
  for (int i = 0; i < n; ++i) pack_vec(hes + (size_t)i * 64, he + (size_t)i * 64, 1, 1.0f);
  for (int l = 0; l < L; ++l) {
    unsigned* ml = hm + (size_t)l * WM_COUNT * W64_MAT_W;
    float* vl = hv + W64_HEAD_F + (size_t)l * W64_LAYER_F;
    const float* e0w = q; q += H * (2 * H + 2);
    const float* e0b = q; q += H;
    const float* e2w = q; q += H * H;
    const float* e2b = q; q += H;
    const float* n0w = q; q += H * 2 * H;
    const float* n0b = q; q += H;
    const float* n2w = q; q += H * H;
    const float* n2b = q; q += H;
    const float* c0w = q; q += H * H;
    const float* c0b = q; q += H;
    const float* c2w = q; q += H;
    const float* aw = nullptr; const float* ab = nullptr;
    if (cfg.attention) { aw = q; q += H; ab = q; q += 1; }
    // SiLU pre-scale kS and the f16-path scales are folded in as pita_egnn_create does for precision 2
    pack_mat(ml, WM_WA, e0w, 2 * H + 2, 0, kS);
    pack_mat(ml, WM_WB, e0w, 2 * H + 2, H, kS);
    pack_mat(ml, WM_W2, e2w, H, 0, 1.0f);
    pack_mat(ml, WM_WC1, c0w, H, 0, 1.0f);
    pack_mat(ml, WM_WN1A, n0w, 2 * H, 0, kS);
    pack_mat(ml, WM_WN1B, n0w, 2 * H, H, 1.0f);
    pack_mat(ml, WM_WN2, n2w, H, 0, kSi);
    for (int b = 0; b < 2; ++b)
      for (int o = 0; o < 32; ++o) {
        const int f = b * 32 + o;
        vl[WV_WRE * 64 + b * 64 + o] = f < H ? kS * e0w[(size_t)f * (2 * H + 2) + 2 * H] : 0.f;
        vl[WV_WRE * 64 + b * 64 + 32 + o] = f < H ? kS * e0w[(size_t)f * (2 * H + 2) + 2 * H + 1] : 0.f;
      }
    pack_vec(vl + WV_B1 * 64, e0b, 1, kS * up);
    pack_vec(vl + WV_B2 * 64, e2b, 1, kS * up);
    if (aw) pack_vec(vl + WV_WATT * 64, aw, 1, kSi * dn);
    pack_vec(vl + WV_BC1 * 64, c0b, 1, kS * up);
    pack_vec(vl + WV_WC2 * 64, c2w, 1, kSi * dn);
    pack_vec(vl + WV_BN1 * 64, n0b, 1, kS * up);
    pack_vec(vl + WV_BN2 * 64, n2b, 1, up);
    vl[WV_COUNT * 64] = ab ? ab[0] : 0.f;
  }
  hipError_t e = hipMalloc(&net->d_m16h, n_m * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&net->d_vecs64, n_v * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&net->d_est64, (size_t)n * 64 * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&net->d_flag, sizeof(int));
  if (e == hipSuccess) e = hipMemcpy(net->d_m16h, hm, n_m * sizeof(unsigned), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(net->d_vecs64, hv, n_v * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(net->d_est64, hes, (size_t)n * 64 * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = ensure_dynamic_lds(reinterpret_cast<const void*>(shape->kernel[cfg.attention ? 1 : 0][cfg.tanh ? 1 : 0]),
                           shape->lds_bytes(L));
  for (const auto& t : kWide64Small)
    if (e == hipSuccess && t.n == shape->n && t.dim == shape->dim)
      e = ensure_dynamic_lds(reinterpret_cast<const void*>(t.kernel[cfg.attention ? 1 : 0][cfg.tanh ? 1 : 0]), t.lds_bytes(L));
  delete[] hm;
  delete[] hv;
  delete[] hes;
  if (e != hipSuccess) {
    wide64_release(net);
    return fail(PITA_EHIP, "pita_egnn_wide_create: matrix-pipe weights: %s", hipGetErrorString(e));
  }
  net->shape64 = shape;
  return PITA_OK;
}

void wide64_release(pita_egnn_wide* net) {
  (void)hipFree(net->d_m16h);
  (void)hipFree(net->d_vecs64);
  (void)hipFree(net->d_est64);
  (void)hipFree(net->d_bk);
  (void)hipFree(net->d_flag);
  (void)hipFree(net->d_jbad);
  net->d_jbad = nullptr;
  net->jbad_bytes = 0;
  net->d_flag = nullptr;
  net->d_bk = nullptr;
  net->bk_bytes = 0;
  net->d_m16h = nullptr;
  net->d_vecs64 = nullptr;
  net->d_est64 = nullptr;
  net->shape64 = nullptr;
}

static int wide64_run(pita_egnn_wide* net, Wide64Params& p, hipStream_t stream);

int wide64_launch(pita_egnn_wide* net, int what, const float* t, const float* x, const float* beta, float* out,
                  long long B, hipStream_t stream) {
  Wide64Params p{};
  p.B = B; p.mode = what; p.x = x; p.t = t; p.beta = beta; p.out = out;
  return wide64_run(net, p, stream);
}

int wide64_sampler(pita_egnn_wide* net, float* x, long long B, const float* step_tab, int n_steps, const float* noise,
                   unsigned long long seed, unsigned long long walker_offset, long long step0, int remove_mean,
                   double* stats_out, int* bad_from, hipStream_t stream) {
  Wide64Params p{};
  p.B = B; p.mode = 3; p.xs = x; p.step_tab = step_tab; p.n_steps = n_steps; p.noise = noise; p.seed = seed;
  p.walker_offset = walker_offset; p.step0 = step0; p.remove_mean = remove_mean; p.stats_out = stats_out;
  p.bad_from = bad_from;
  return wide64_run(net, p, stream);
}

static int wide64_run(pita_egnn_wide* net, Wide64Params& p, hipStream_t stream) {
  const long long B = p.B;
  const Wide64Shape* s = static_cast<const Wide64Shape*>(net->shape64);
  for (const auto& t : kWide64Small)  // small batch: the one-walker mapping when the regular one fills under 3/4 of the SIMDs
    if (t.n == s->n && t.dim == s->dim && (B + s->G - 1) / s->G < (long long)net->n_cu * 3) s = &t;
  p.m16h = net->d_m16h; p.vecs = net->d_vecs64; p.est = net->d_est64;
  p.L = net->cfg.n_layers; p.attention = net->cfg.attention; p.tanh_on = net->cfg.tanh; p.has_beta = net->cfg.condition_beta;
  p.coord_scale = net->cfg.coords_range / (float)net->cfg.n_layers;
  p.flag = net->d_flag;
  if (p.flag) PITA_HIP_CHECK(hipMemsetAsync(p.flag, 0, sizeof(int), stream));
  const long long ngroups = (B + s->G - 1) / s->G;
  const long long want = (ngroups + s->waves - 1) / s->waves, cap = net->n_cu;  // one 4-wave block per CU
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  hipLaunchKernelGGL(s->kernel[p.attention ? 1 : 0][p.tanh_on ? 1 : 0], dim3(grid), dim3(s->waves * 64), s->lds_bytes(p.L), stream, p);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

}  // namespace pita
