// EGNN score network + fused Euler-Maruyama sampler step for gfx950 (MI355X).
//
// Replaces (reference paths relative to /root/reference/pita/src/models/components/):
//   egnn_temp_conditioned.py:56-93   EGNN_dynamics.forward  (egnn.py:50-80 without beta)
//   egnn_temp_conditioned.py:172-194 EGNN.forward
//   egnn_temp_conditioned.py:321-356 E_GCL.forward / edge_model :265 / coord_model :294 /
//                                    node_model :281 / coord2radial :348
//   score_net.py:13-43               ScoreNet.forward / denoiser (EDM preconditioning)
//   sdes.py:117-128,245-251          VEReverseSDE.f_not_debiased / diffusion
//   sde_integration.py:299-351,148   euler_maruyama_step + remove_mean
//
// Mapping to the hardware (one wavefront = one independent walker group, no barriers):
//   * a wave owns G walkers = G*N graph nodes = "columns"; columns are packed densely into NT
//     tiles of 32.  Lane l works on column (l & 31) of the current tile and holds 16 of the 32
//     hidden features of that column: feature kfeat(r,hh) = (r&3) + 8(r>>2) + 4hh, hh = l>>5.
//     This is exactly the C/D layout of v_mfma_f32_32x32x2_f32, and -- with the weight
//     fragment W[out = l&31][in = kfeat(r,hh)] as the A operand -- also its B layout, so every
//     32x32 dense layer is a chain of 16 MFMAs whose result feeds the next layer's MFMA with no
//     data movement at all (exact fp32: the f32 MFMA is a k-ordered fmaf chain).
//   * the directed edges (i -> j) of the fully connected graph are enumerated as
//     j = (i + dd) mod N, dd = 1..N-1: for a fixed dd every lane (= node i) processes ONE of its
//     own outgoing edges, so the per-node sums over j (message aggregation, coordinate update)
//     are plain in-register accumulations over the dd loop -- no scatter, no atomics, no
//     cross-lane reduction, and a deterministic summation order.
//   * the first edge-MLP layer is split algebraically: W1 [h_i,h_j,r,e] = Wa h_i + Wb h_j +
//     w_r r + w_e e.  Wa h_i, Wb h_j are per-node GEMMs (2 MFMA chains per tile per layer); the
//     partner term Wb h_j and the partner coordinates are gathered from a per-wave LDS table
//     (row stride 36 floats -> conflict-free ds_read_b128 for consecutive columns).
//   * walkers stay in registers across all SDE steps of a launch (fused sampler mode): HBM
//     traffic per walker-step is ~0 and there is no per-step tail/launch cost.
//   * the last layer's node update and the embedding_out head are dead in the reference
//     (egnn_temp_conditioned.py:80,189: h_final is discarded) and are skipped.
//   * two waves share a SIMD (OCC = 2, 256 registers each).  Inside an edge the three 16-wide SiLUs -- the only phases that
//     can issue every cycle -- run at issue priority 0, the dependent phases (matrix chains with their operand splits, dot
//     chains, gathers) at 1 (s_setprio): the partner's scarce instruction goes first, the SiLU fills what is left.
#include "egnn_common.h"

namespace pita {

struct EgnnParams {
  const unsigned* mats16;  // [L][M_COUNT][3][2][64][4]  bf16-split fragments (PREC 1, and the per-node layers of PREC 2)
  const unsigned* mats16h; // [L][M_COUNT][2][2][64][4]  f16-split fragments (PREC 2: W2, Wc1, Wn2)
  const float* mats;  // [L][M_COUNT][4][64][4]
  const float* vecs;  // [VEC_EMB_F + L*VEC_LAYER_F]
  int n_layers, in_nf, attention, tanh_on, feature_layout;
  float coord_scale;  // coords_range / n_layers
  int mode;           // 0 forward, 1 denoiser, 2 score, 3 fused sampler steps
  long long B;
  // modes 0-2
  const float* x_in;
  const float* t;     // mode 0: backbone time input (c_noise); modes 1,2: h = sigma^2
  const float* beta;  // nullable
  float* out;
  // mode 3
  float* x;
  const float* step_tab;
  int n_steps;
  const float* noise;
  unsigned long long seed, walker_offset;
  long long step0;
  int remove_mean;
  float* drift_out;
  double* stats_out;  // nullable [n_steps][4]: += sum / sum of squares of drift_X and of the diffusion term
  // PREC 2 runs as two launches (egnn_launch): the f16 kernel, then the PREC 1 kernel with repair = 1, which recomputes
  // exactly those walker groups whose results came out non-finite (an activation beyond the f16 range) and exits at
  // once everywhere else
  int repair;
  const float* x_backup;  // mode 3: the walkers as they were before the f16 launch
  int* bad_from;          // mode 3 with stats_out: [B*N] first step whose moments the f16 launch left out (INT_MAX: none)
};

#ifndef PITA_EGNN_RECOMPUTE_COLS
#define PITA_EGNN_RECOMPUTE_COLS 1
#endif

template <int N, int DIM, int G, int WAVES>
struct EgnnCfg {
  static constexpr int NCOL = G * N;
  static constexpr int NT = (NCOL + 31) / 32;
  static constexpr int NCOLP = NT * 32;
  static constexpr int PB_F = NCOLP * PBS;
  static constexpr int POS_F = NCOLP * DIM;
  static constexpr int WAVE_F = PB_F + 4 * POS_F;  // partner table, pos[2], pos0, walkers (x of the current step)
  static __host__ __device__ constexpr int vec_f(int L) { return ((VEC_EMB_F + L * VEC_LAYER_F) + 3) & ~3; }
  static __host__ __device__ constexpr size_t lds_bytes(int L) {
    return sizeof(float) * (size_t)(vec_f(L) + WAVES * WAVE_F);
  }
};

// SAMPLER = true: mode 3 only (per-step scalars are wave-uniform -> SGPRs); false: modes 0-2 (per-walker t / h / beta).
// Separate instantiations keep the fused sampler's register budget free of the forward modes' per-column scalars.
template <int N, int DIM, int G, int WAVES, int PREC, bool SAMPLER, int OCC = 2>
__global__ void __launch_bounds__(WAVES * 64, OCC) egnn_kernel(EgnnParams p) {
  const int mode = SAMPLER ? 3 : p.mode;
  using C = EgnnCfg<N, DIM, G, WAVES>;
  constexpr int NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int L = p.n_layers;
  const int vec_f = C::vec_f(L);
  for (int i = threadIdx.x; i < VEC_EMB_F + L * VEC_LAYER_F; i += WAVES * 64) lds[i] = p.vecs[i];
  __syncthreads();  // the only workgroup barrier: weight vectors are shared by the block's waves

  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int cl = lane & 31;
  const int hh = lane >> 5;
  float* PB = lds + vec_f + wave * C::WAVE_F;
  float* posbuf0 = PB + C::PB_F;
  float* posbuf1 = posbuf0 + C::POS_F;
  float* pos0 = posbuf1 + C::POS_F;
  float* xbuf = pos0 + C::POS_F;  // the walkers' unscaled coordinates: parked in LDS across the layers, not in registers
  const float* vemb = lds;

  // Work split: every wave gets the same contiguous quota of walkers (65 536 walkers on 2 048 resident waves =
  // 32 each) and cuts it into groups of at most G walkers (7,7,7,7,4): all waves finish together instead of a
  // ragged last round of whole groups.  Results do not depend on the grouping (tested bitwise).
  const long long total_waves = (long long)gridDim.x * WAVES;
  const long long quota = (p.B + total_waves - 1) / total_waves;
  const long long wbeg = ((long long)blockIdx.x * WAVES + wave) * quota;
  const long long wend = (wbeg + quota < p.B) ? wbeg + quota : p.B;
  for (long long walker0 = wbeg; walker0 < wend; walker0 += G) {
    const int nwalk = (int)((wend - walker0) < G ? (wend - walker0) : G);
    const int ncol = nwalk * N;              // live columns of this group
    const int ntile = (ncol + 31) >> 5;      // live column tiles (uniform)
    // ---- per-column bookkeeping
    int col[NT], nodei[NT];
    bool valid[NT];
    long long wid[NT];
    float xcur[NT][DIM];  // transient: the walkers as loaded; they live in xbuf (LDS) from here on
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      col[T] = T * 32 + cl;
      int w = col[T] / N;
      nodei[T] = col[T] - w * N;
      wid[T] = walker0 + w;
      valid[T] = (col[T] < ncol);
      if (!valid[T]) wid[T] = p.B - 1;  // clamp for safe (unused) parameter loads
      const float* src = (mode == 3 ? p.x : p.x_in);
#pragma unroll
      for (int k = 0; k < DIM; ++k) xcur[T][k] = valid[T] ? src[(walker0 * N + col[T]) * DIM + k] : 0.0f;
    }
    int bad_from[NT];  // per (walker, particle): first step whose moments are still owed (see EgnnParams::repair)
#pragma unroll
    for (int T = 0; T < NT; ++T) bad_from[T] = 0x7fffffff;
    bool mine[NT];  // this column's walker takes the results of this launch (repair launch: only the non-finite ones)
#pragma unroll
    for (int T = 0; T < NT; ++T) mine[T] = valid[T];
    if (PREC != 2 && p.repair) {
      bool bad = false;
      float* flag = PB;  // per-column flag table (the partner table is not live yet)
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        bool b = false;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const float v = (mode == 3) ? xcur[T][k] : (valid[T] ? p.out[(walker0 * N + col[T]) * DIM + k] : 0.0f);
          b = b || !__builtin_isfinite(v);
        }
        if (hh == 0) flag[col[T]] = b ? 1.0f : 0.0f;
        bad = bad || b;
      }
      if (!__any(bad)) continue;  // wave-uniform: this group's f16 result stands
      // walker granularity: the group is recomputed, but only walkers with a non-finite particle are overwritten, so
      // a walker's result never depends on its neighbours in the batch
      wave_lds_fence();
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        const int cb = (col[T] < ncol) ? col[T] - nodei[T] : 0;
        float any = 0.f;
        for (int q = 0; q < N; ++q) any += flag[cb + q];
        mine[T] = valid[T] && any != 0.f;
      }
      wave_lds_fence();
      if (mode == 3) {
#pragma unroll
        for (int T = 0; T < NT; ++T) {
#pragma unroll
          for (int k = 0; k < DIM; ++k)
            xcur[T][k] = valid[T] ? p.x_backup[(walker0 * N + col[T]) * DIM + k] : 0.0f;
          if (p.stats_out && p.bad_from && valid[T]) bad_from[T] = p.bad_from[walker0 * N + col[T]];
        }
      }
    }

#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int k = 0; k < DIM; ++k)
        if (hh == 0) xbuf[col[T] * DIM + k] = xcur[T][k];
    wave_lds_fence();

    const int nsteps = (mode == 3) ? p.n_steps : 1;
    for (int step = 0; step < nsteps; ++step) {
#if PITA_EGNN_RECOMPUTE_COLS
      // the per-column bookkeeping is recomputed from the (opaque) lane id at every step instead of being carried
      // across the step loop: carried, the register allocator spills it once per walker group (~25 dwords per lane,
      // the bulk of the kernel's HBM writes); recomputed it costs ~30 instructions per wave and step
      int col[NT], nodei[NT];
      bool valid[NT];
      long long wid[NT];
      {
        int cl_o = cl;
        asm volatile("" : "+v"(cl_o));
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          col[T] = T * 32 + cl_o;
          const int w = col[T] / N;
          nodei[T] = col[T] - w * N;
          wid[T] = walker0 + w;
          valid[T] = (col[T] < ncol);
          if (!valid[T]) wid[T] = p.B - 1;
        }
      }
#endif
      // ---- per-column scalars of this evaluation
      float c_s[NT], c_in[NT], c_out[NT], tfeat[NT], hval[NT], bfeat[NT];
      float g2 = 0.f, gamma = 0.f, dt = 0.f, noise_scale = 0.f, sqrt_dt = 0.f;
      if (mode == 3) {
        // wave-uniform per-step scalars: through v_readfirstlane into SGPRs (a plain load of the table would occupy a
        // dozen VGPRs per lane for the whole step: the pointer is not known to be read-only, so no scalar load)
        const float* st = p.step_tab + (size_t)step * PITA_STEP_STRIDE;
        auto uni = [&](int i) {
          return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, st[i])));
        };
        const float u_cs = uni(PITA_ST_CS), u_cin = uni(PITA_ST_CIN), u_cout = uni(PITA_ST_COUT),
                    u_cn = uni(PITA_ST_CNOISE), u_h = uni(PITA_ST_H), u_b = uni(PITA_ST_BETA);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          c_s[T] = u_cs; c_in[T] = u_cin; c_out[T] = u_cout;
          tfeat[T] = u_cn; hval[T] = u_h; bfeat[T] = u_b;
        }
        g2 = uni(PITA_ST_G2); gamma = uni(PITA_ST_GAMMA); dt = uni(PITA_ST_DT);
        noise_scale = uni(PITA_ST_NOISE_SCALE); sqrt_dt = uni(PITA_ST_SQRT_DT);
      } else {
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          float tv = p.t[wid[T]];
          bfeat[T] = p.beta ? p.beta[wid[T]] : 0.0f;
          if (mode == 0) {
            c_s[T] = 0.f; c_in[T] = 1.f; c_out[T] = 1.f; tfeat[T] = tv; hval[T] = 1.f;
          } else {  // score_net.py:26-29
            hval[T] = tv;
            c_s[T] = 1.0f / (1.0f + tv);
            c_in[T] = 1.0f / sqrtf(1.0f + tv);
            c_out[T] = sqrtf(tv) * c_in[T];
            tfeat[T] = 0.125f * logf(tv);
          }
        }
      }

      // ---- scaled input coordinates -> registers + LDS (pos0 = input geometry, frozen edge_attr)
      // (positions live in the LDS tables pos0 / poscur between uses: each tile reloads its own column)
      f32x16 hfeat[NT];
#pragma unroll
      for (int T = 0; T < NT; ++T) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const float xc = xbuf[col[T] * DIM + k];
          const float ps = (mode == 0) ? xc : c_in[T] * xc;
          if (hh == 0) {
            pos0[col[T] * DIM + k] = ps;
            posbuf0[col[T] * DIM + k] = ps;
          }
        }
        // initial node features (egnn_temp_conditioned.py:63-78) and embedding (:179)
        float a0, a1;
        if (p.in_nf == 1) {
          a0 = tfeat[T]; a1 = 0.f;
        } else if (p.feature_layout == 0) {  // quirk Q1: rows of reshape([t]*N + [beta]*N, (N,2))
          a0 = (2 * nodei[T] < N) ? tfeat[T] : bfeat[T];
          a1 = (2 * nodei[T] + 1 < N) ? tfeat[T] : bfeat[T];
        } else {
          a0 = tfeat[T]; a1 = bfeat[T];
        }
        f32x16 w0 = lds_vec16(vemb + hh * 16), w1 = lds_vec16(vemb + 32 + hh * 16), eb = lds_vec16(vemb + 64 + hh * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) hfeat[T][r] = fmaf(w0[r], a0, fmaf(w1[r], a1, eb[r]));
      }
      wave_lds_fence();

      float* poscur = posbuf0;
      float* posnext = posbuf1;
      for (int l = 0; l < L; ++l) {
        const float* mats = p.mats + (size_t)l * M_COUNT * MAT_F;
        const unsigned* mats16 = p.mats16 + (size_t)l * M_COUNT * MAT_W;    // bf16 three-piece fragments
        const unsigned* mats16h = p.mats16h + (size_t)l * M_COUNT * MAT_WH;  // f16 two-piece fragments (PREC 2)
        constexpr int NPREC = (PREC == 2 && !F16_NODE_LAYERS) ? 1 : PREC;    // arithmetic of the per-node layers
        constexpr bool NODE16 = (PREC == 2 && F16_NODE_LAYERS);
        const unsigned* matsn = NODE16 ? mats16h : mats16;
        const float* vl = lds + VEC_EMB_F + l * VEC_LAYER_F + hh * 16;
        const bool last = (l == L - 1);
        // the last layer's aggregate is dead: a multiplier instead of 16 selects (and it undoes the PREC 2 scale)
        const float aggw = last ? 0.0f : (PREC == 2 ? 1.0f / F16_SX : 1.0f);
        // PREC 2: SiLU outputs travel F16_SX-scaled and the accumulators of W2 / Wc1 / Wn2 hold F16_SX F16_SW x their
        // true value (egnn_common.h); the bias / gate / head vectors arrive pre-scaled from the host (pita_egnn_create)
        // ---- partner table PB[col] = Wb h_col
        {
          WFrag<NPREC> wb;
          wb.load(mats, matsn, M_WB, lane);
#pragma unroll
          for (int T = 0; T < NT; ++T) {
            if (T >= ntile) continue;
            f32x16 z = {0};
            f32x16 pb = wb.mul(hfeat[T], z);
            if (NODE16) pb *= F16_UNSCALE;
            f32x4* dst = reinterpret_cast<f32x4*>(PB + col[T] * PBS + hh * 16);
            dst[0] = f32x4{pb[0], pb[1], pb[2], pb[3]};
            dst[1] = f32x4{pb[4], pb[5], pb[6], pb[7]};
            dst[2] = f32x4{pb[8], pb[9], pb[10], pb[11]};
            dst[3] = f32x4{pb[12], pb[13], pb[14], pb[15]};
          }
        }
        wave_lds_fence();

        WFrag<PREC> w2f, wc1f;
        w2f.load(mats, PREC == 2 ? mats16h : mats16, M_W2, lane);
        wc1f.load(mats, PREC == 2 ? mats16h : mats16, M_WC1, lane);
        // A operand of the extra k-step that adds w_r*radial + w_e*edge_attr: A[out][k] = (w_r | w_e)
        const float a_re = lds[VEC_EMB_F + l * VEC_LAYER_F + V_WRE * EH + lane];
        const float b_att = lds[VEC_EMB_F + l * VEC_LAYER_F + V_COUNT * EH];

#pragma unroll
        for (int T = 0; T < NT; ++T) {
          if (T >= ntile) continue;
          // own first-layer term  Ai = Wa h_i + b1
          f32x16 Ai;
          {
            WFrag<NPREC> wa;
            wa.load(mats, matsn, M_WA, lane);
            Ai = wa.mul(hfeat[T], lds_vec16(vl + V_B1 * EH));
            if (NODE16) Ai *= F16_UNSCALE;
          }
          f32x16 agg = {0};
          float xacc[DIM], pown[DIM], p0own[DIM];
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            xacc[k] = 0.f;
            pown[k] = poscur[col[T] * DIM + k];
            p0own[k] = pos0[col[T] * DIM + k];
          }
          const int cbase = col[T] - nodei[T];

          // Issue priority inside the edge (round 5).  Two waves share a SIMD and one vector issue port.  An edge alternates
          // between phases that can issue every cycle -- the three 16-wide SiLUs: 16 exp, 16 rcp and their packed adds /
          // multiplies, all independent -- and phases whose next instruction is rarely ready: the two dense layers (six
          // dependent matrix instructions with the operand split threaded between them), the two 8-link dot chains, the
          // gathers.  With equal priorities a wave in a SiLU takes every other issue slot from a partner that is in a
          // dependent phase, and the partner's matrix chain -- and with it the matrix pipe -- waits.  The SiLUs run at
          // priority 0, everything else at 1: the wave with the scarce instruction goes first, the SiLU fills what is left.
          // Same instructions, same bits; 108.3 -> 102.5 ms per 100 steps on one box (profiles/r05_sampler_issue_priority.txt).
          for (int dd = 1; dd < N; ++dd) {
            asm volatile("" ::: "memory");  // keep the per-edge LDS vector loads inside the loop (VGPR budget)
            int j = nodei[T] + dd;
            j = (j >= N) ? j - N : j;
            const int cj = (col[T] < ncol) ? cbase + j : col[T];
            // geometry (coord2radial :348-356; edge_attr :79)
            float df[DIM], radial = 0.f, ea = 0.f;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              df[k] = pown[k] - poscur[cj * DIM + k];
              radial = fmaf(df[k], df[k], radial);
              const float e0 = p0own[k] - pos0[cj * DIM + k];
              ea = fmaf(e0, e0, ea);
            }
            // edge MLP layer 1 (:232-237,:270-271): Wa h_i + Wb h_j + b1, then one k-step [w_r|w_e]·[radial;ea]
            f32x16 m = Ai + lds_vec16(PB + cj * PBS + hh * 16);
            m = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re, hh ? ea : radial, m, 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if (PREC == 2) silu16_out(m); else silu16(m);
            __builtin_amdgcn_s_setprio(1);
            m = w2f.mul(m, lds_vec16(vl + V_B2 * EH));
            __builtin_amdgcn_s_setprio(0);
            if (PREC == 2) silu16_acc(m); else silu16(m);
            __builtin_amdgcn_s_setprio(1);
            if (p.attention) {  // :259-260,:273-275
              const float att = fast_sigmoid(xhalf_sum(dot16(lds_vec16(vl + V_WATT * EH), m)) + b_att);
              m *= att;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) agg[r] = fmaf(m[r], aggw, agg[r]);  // node_model aggregation (:284)
            // coordinate head (:245-256,:297-298)
            f32x16 c1 = wc1f.mul(m, lds_vec16(vl + V_BC1 * EH));
            __builtin_amdgcn_s_setprio(0);
            if (PREC == 2) silu16_acc(c1); else silu16(c1);
            __builtin_amdgcn_s_setprio(1);
            float cs = xhalf_sum(dot16(lds_vec16(vl + V_WC2 * EH), c1));
            if (p.tanh_on) cs = accurate_tanh(cs) * p.coord_scale;
            // 1 / (|d| + 1) on the transcendental unit (v_sqrt_f32, v_rcp_f32: ~1 ulp each) instead of the IEEE
            // division / square-root expansions (~45 instructions per edge)
            const float inrm = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(radial + 1e-8f) + 1.0f);
#pragma unroll
            for (int k = 0; k < DIM; ++k) xacc[k] = fmaf(df[k] * inrm, cs, xacc[k]);
          }
          // coordinate update (:306,:318): other tiles still read the old coordinates
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            if (hh == 0) posnext[col[T] * DIM + k] = pown[k] + xacc[k];
          }
          if (!last) {  // node model (:239-243,:284-291), recurrent
            WFrag<NPREC> wn;
            wn.load(mats, matsn, M_WN1A, lane);
            f32x16 n1 = wn.mul(hfeat[T], lds_vec16(vl + V_BN1 * EH));
            wn.load(mats, matsn, M_WN1B, lane);
            n1 = wn.mul(agg, n1);
            if (NODE16) silu16_acc(n1); else if (PREC == 2) silu16_out(n1); else silu16(n1);
            WFrag<PREC> wo;
            wo.load(mats, PREC == 2 ? mats16h : mats16, M_WN2, lane);
            f32x16 o = wo.mul(n1, lds_vec16(vl + V_BN2 * EH));
            if (PREC == 2) o *= F16_UNSCALE;
            hfeat[T] += o;
          }
        }
        wave_lds_fence();
        float* tmp = poscur; poscur = posnext; posnext = tmp;
      }

      // ---- vel = x_final - x, mean-free over the walker's particles (:81-84)
      float* scr = PB;  // partner table is free now
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

      if (mode != 3) {
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            float o = F[T][k];
            if (mode >= 1) {
              const float xc = xbuf[col[T] * DIM + k];
              o = c_s[T] * xc + c_out[T] * F[T][k];           // denoiser (score_net.py:31-33)
              if (mode == 2) o = (o - xc) / hval[T];        // score (:19)
            }
            if (mine[T] && hh == 0) p.out[(walker0 * N + col[T]) * DIM + k] = o;
          }
      } else {
        // ---- reverse-SDE Euler-Maruyama update (sdes.py:119-122,250; sde_integration.py:347-348)
        float xn[NT][DIM];
        float st_d = 0.f, st_d2 = 0.f, st_n = 0.f, st_n2 = 0.f;  // per-lane partial moments (<= NT*DIM terms each)
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          float xi[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.noise) {
#pragma unroll
            for (int k = 0; k < DIM; ++k)
              xi[k] = valid[T] ? p.noise[((size_t)step * p.B * N + (size_t)(walker0 * N + col[T])) * DIM + k] : 0.f;
          } else {
            philox_normal4(p.seed, p.walker_offset + (unsigned long long)wid[T], p.step0 + step, (uint32_t)nodei[T], xi);
          }
          // whose launch owns this column's moments of this step: decided per COLUMN (all DIM components together), so the
          // f16 launch and the repair launch partition the moments exactly even when the components of one particle
          // differ in finiteness at the step where an overflow first shows
          bool take_col = true;
          if (p.stats_out && valid[T] && hh == 0) {
            if (PREC == 2) {  // leave non-finite values to the repair launch and remember from which step on
#pragma unroll
              for (int k = 0; k < DIM; ++k) {
                const float xc = xbuf[col[T] * DIM + k];
                const float Dth = c_s[T] * xc + c_out[T] * F[T][k];
                take_col = take_col && __builtin_isfinite(gamma * (((Dth - xc) / hval[T]) * g2));
              }
              take_col = take_col && bad_from[T] == 0x7fffffff;  // once owed, always owed (the walker is recomputed)
              if (!take_col && bad_from[T] == 0x7fffffff) bad_from[T] = step;
            } else if (p.repair) {
              take_col = step >= bad_from[T];
            }
          }
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            const float xc = xbuf[col[T] * DIM + k];
            const float Dth = c_s[T] * xc + c_out[T] * F[T][k];
            const float sc = (Dth - xc) / hval[T];
            const float drift = gamma * (sc * g2);
            if (p.drift_out && step == nsteps - 1 && mine[T] && hh == 0)
              p.drift_out[(walker0 * N + col[T]) * DIM + k] = drift;
            const float dif = noise_scale * xi[k];  // sdes.py:250
            if (p.stats_out && valid[T] && hh == 0) {
              const bool take = take_col;
              if (take) {
                st_d += drift; st_d2 = fmaf(drift, drift, st_d2);
                st_n += dif; st_n2 = fmaf(dif, dif, st_n2);
              }
            }
            xn[T][k] = xc + (drift * dt + (dif * sqrt_dt));
            if (hh == 0) scr[col[T] * DIM + k] = xn[T][k];
          }
        }
        if (p.stats_out) {  // per-step moments of the SDETerms the reference returns (sde_integration.py:150,289)
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
        wave_lds_fence();
      }
    }  // steps

    if (mode == 3) {
#pragma unroll
      for (int T = 0; T < NT; ++T) {
#pragma unroll
        for (int k = 0; k < DIM; ++k)
          if (mine[T] && hh == 0) p.x[(walker0 * N + col[T]) * DIM + k] = xbuf[col[T] * DIM + k];
        if (PREC == 2 && p.stats_out && p.bad_from && valid[T] && hh == 0) p.bad_from[walker0 * N + col[T]] = bad_from[T];
      }
    }
    wave_lds_fence();
  }
}

// ------------------------------------------------------------------------------------ host side

struct EgnnShape {
  int n, dim, G, waves, occ;  // occ: resident workgroups per CU the kernels were compiled for
  void (*kernel[3][2])(EgnnParams);  // [PREC][SAMPLER]
  size_t (*lds_bytes)(int);
};

template <int N, int DIM, int G, int WAVES>
static size_t lds_bytes_of(int L) { return EgnnCfg<N, DIM, G, WAVES>::lds_bytes(L); }

#define PITA_EGNN_SHAPE(N, DIM, G, WAVES, OCC) \
  EgnnShape { N, DIM, G, WAVES, OCC, \
              {{egnn_kernel<N, DIM, G, WAVES, 0, false, OCC>, egnn_kernel<N, DIM, G, WAVES, 0, true, OCC>}, \
               {egnn_kernel<N, DIM, G, WAVES, 1, false, OCC>, egnn_kernel<N, DIM, G, WAVES, 1, true, OCC>}, \
               {egnn_kernel<N, DIM, G, WAVES, 2, false, OCC>, egnn_kernel<N, DIM, G, WAVES, 2, true, OCC>}}, \
              lds_bytes_of<N, DIM, G, WAVES> }

// Instantiated (n_particles, n_dim) shapes: DW4, LJ13, alanine dipeptide (22 atoms), LJ55.
static const EgnnShape kShapes[] = {
    PITA_EGNN_SHAPE(4, 2, 8, 4, 2),
    PITA_EGNN_SHAPE(13, 3, 7, 4, 2),
    PITA_EGNN_SHAPE(22, 3, 4, 4, 2),
    PITA_EGNN_SHAPE(55, 3, 1, 4, 2),
};
// Fewer walkers per wave for batches that cannot give every SIMD two waves with the mapping above (a lone wave issues a
// vector instruction every ~5 cycles, two waves one every ~2.5: tools/ubench/isa_rates.hip): more, shorter waves at a
// lower column fill.  Results do not depend on the grouping (tested bitwise).
static const EgnnShape kShapesSmall[] = {
    PITA_EGNN_SHAPE(13, 3, 2, 4, 2),
    PITA_EGNN_SHAPE(22, 3, 2, 4, 2),
};

}  // namespace pita


using namespace pita;

extern "C" int64_t pita_egnn_num_weights(const pita_egnn_config* c) {
  if (!c) return PITA_EINVAL;
  const int64_t H = c->hidden_nf;
  int64_t per_layer = (H * (2 * H + 2) + H) + (H * H + H) + (H * 2 * H + H) + (H * H + H) + (H * H + H) + H;
  if (c->attention) per_layer += H + 1;
  return (H * c->in_node_nf + H) + (c->in_node_nf * H + c->in_node_nf) + c->n_layers * per_layer;
}

static inline int kfeat(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

extern "C" int pita_egnn_create(pita_egnn_t** out, const pita_egnn_config* cfg, const float* w, int64_t n_weights) {
  PITA_REQUIRE(out && cfg && w, "pita_egnn_create: null argument");
  if (cfg->hidden_nf != EH)
    return fail(PITA_EUNSUPPORTED, "pita_egnn_create: hidden_nf=%d (the HIP kernel implements 32)", cfg->hidden_nf);
  PITA_REQUIRE(cfg->in_node_nf == 1 || cfg->in_node_nf == 2, "in_node_nf must be 1 or 2");
  PITA_REQUIRE(cfg->precision >= 0 && cfg->precision <= 2,
               "precision must be 0 (f32 MFMA), 1 (bf16 three-piece split) or 2 (f16 two-piece split)");
  PITA_REQUIRE(cfg->n_layers >= 1 && cfg->n_layers <= 16, "n_layers out of range");
  PITA_REQUIRE(n_weights == pita_egnn_num_weights(cfg), "pita_egnn_create: got %lld weights, expected %lld",
               (long long)n_weights, (long long)pita_egnn_num_weights(cfg));
  const EgnnShape* shape = nullptr;
  const EgnnShape* shape_small = nullptr;
  for (const auto& s : kShapes)
    if (s.n == cfg->n_particles && s.dim == cfg->n_dim) shape = &s;
  for (const auto& s : kShapesSmall)
    if (s.n == cfg->n_particles && s.dim == cfg->n_dim) shape_small = &s;
  const int H = EH, L = cfg->n_layers, nf = cfg->in_node_nf;
  const size_t n_mats = (size_t)L * M_COUNT * MAT_F, n_vecs = VEC_EMB_F + (size_t)L * VEC_LAYER_F;
  float* h_mats = new float[n_mats];
  float* h_vecs = new float[n_vecs]();
  const size_t n_mats16 = (size_t)L * M_COUNT * MAT_W;
  unsigned* h_mats16 = new unsigned[n_mats16];
  const size_t n_mats16h = (size_t)L * M_COUNT * MAT_WH;
  unsigned* h_mats16h = new unsigned[n_mats16h]();
  float* h_vecs_h = new float[n_vecs]();
  float* h_vecs_div = new float[(size_t)L * VEC_DIV_F]();
  // walk the state_dict order
  const float* q = w;
  const float* emb_w = q; q += H * nf;
  const float* emb_b = q; q += H;
  q += nf * H + nf;  // embedding_out: dead (h_final is discarded, egnn_temp_conditioned.py:80)
  for (int hh = 0; hh < 2; ++hh)
    for (int r = 0; r < 16; ++r) {
      const int f = kfeat(r, hh);
      h_vecs[hh * 16 + r] = emb_w[f * nf + 0];
      h_vecs[32 + hh * 16 + r] = (nf == 2) ? emb_w[f * nf + 1] : 0.f;
      h_vecs[64 + hh * 16 + r] = emb_b[f];
    }
  // SiLU pre-activations are kept in the scaled form z' = kS z (kS = -log2 e): then exp(-z) = exp2(z') needs no
  // multiply and y' = z' / (1 + exp2(z')) = kS silu(z).  The scale is folded into the weights once, here:
  //   producers of a SiLU input (Wa, Wb, w_r, w_e, b1, b2, bc1, Wn1a, bn1)      x kS
  //   linear consumers of a SiLU output that do NOT feed another SiLU (w_att, w_c2, Wn2)   x 1/kS
  //   W2, Wc1, Wn1b (SiLU output -> SiLU input) and the node features h stay unscaled.
  const float kS = SILU_PRESCALE, kSi = 1.0f / SILU_PRESCALE;
  auto pack_mat = [&](float* dst, const float* M, int ld, int col0, float sc) {
    for (int qd = 0; qd < 4; ++qd)
      for (int lane = 0; lane < 64; ++lane)
        for (int s = 0; s < 4; ++s)
          dst[(qd * 64 + lane) * 4 + s] = sc * M[(lane & 31) * ld + col0 + kfeat(4 * qd + s, lane >> 5)];
  };
  // bf16 three-way truncation split of the same fragments: word q of (piece, kstep st, lane) packs the
  // pieces of elements r = 8 st + 2q (low half) and r + 1 (high half)
  auto trunc16 = [](float v) { unsigned u; memcpy(&u, &v, 4); u &= 0xFFFF0000u; float o; memcpy(&o, &u, 4); return o; };
  auto hi16 = [](float v) { unsigned u; memcpy(&u, &v, 4); return u >> 16; };
  auto pack_mat16 = [&](unsigned* dst, const float* M, int ld, int col0, float sc) {
    for (int lane = 0; lane < 64; ++lane)
      for (int st = 0; st < 2; ++st)
        for (int qd = 0; qd < 4; ++qd) {
          unsigned pcs[2][3];
          for (int e = 0; e < 2; ++e) {
            const float w = sc * M[(lane & 31) * ld + col0 + kfeat(8 * st + 2 * qd + e, lane >> 5)];
            const float w1 = trunc16(w), r1 = w - w1, w2 = trunc16(r1), r2 = r1 - w2;
            pcs[e][0] = hi16(w1); pcs[e][1] = hi16(w2); pcs[e][2] = hi16(r2);
          }
          for (int pc = 0; pc < 3; ++pc)
            dst[(((size_t)pc * 2 + st) * 64 + lane) * 4 + qd] = pcs[0][pc] | (pcs[1][pc] << 16);
        }
  };
  // f16 two-piece round-to-nearest split of F16_SW x the same fragments (PREC 2): [piece][kstep][lane][4]
  auto f16_bits = [](float v) { _Float16 h = (_Float16)v; unsigned short u; memcpy(&u, &h, 2); return (unsigned)u; };
  auto pack_mat16h = [&](unsigned* dst, const float* M, int ld, int col0, float sc) {
    for (int lane = 0; lane < 64; ++lane)
      for (int st = 0; st < 2; ++st)
        for (int qd = 0; qd < 4; ++qd) {
          unsigned pcs[2][2];
          for (int e = 0; e < 2; ++e) {
            const float w = F16_SW * (sc * M[(lane & 31) * ld + col0 + kfeat(8 * st + 2 * qd + e, lane >> 5)]);
            const _Float16 w1 = (_Float16)w;
            pcs[e][0] = f16_bits((float)w1);
            pcs[e][1] = f16_bits(w - (float)w1);
          }
          for (int pc = 0; pc < 2; ++pc)
            dst[(((size_t)pc * 2 + st) * 64 + lane) * 4 + qd] = pcs[0][pc] | (pcs[1][pc] << 16);
        }
  };
  auto pack_vec = [&](float* dst, const float* v, int stride, float sc) {
    for (int hh = 0; hh < 2; ++hh)
      for (int r = 0; r < 16; ++r) dst[hh * 16 + r] = sc * v[kfeat(r, hh) * stride];
  };
  for (int l = 0; l < L; ++l) {
    float* mats = h_mats + (size_t)l * M_COUNT * MAT_F;
    float* vecs = h_vecs + VEC_EMB_F + (size_t)l * VEC_LAYER_F;
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
    if (cfg->attention) { aw = q; q += H; ab = q; q += 1; }
    unsigned* m16 = h_mats16 + (size_t)l * M_COUNT * MAT_W;
    pack_mat16(m16 + M_WA * MAT_W, e0w, 2 * H + 2, 0, kS);
    pack_mat16(m16 + M_WB * MAT_W, e0w, 2 * H + 2, H, kS);
    pack_mat16(m16 + M_W2 * MAT_W, e2w, H, 0, 1.0f);
    pack_mat16(m16 + M_WC1 * MAT_W, c0w, H, 0, 1.0f);
    pack_mat16(m16 + M_WN1A * MAT_W, n0w, 2 * H, 0, kS);
    pack_mat16(m16 + M_WN1B * MAT_W, n0w, 2 * H, H, 1.0f);
    pack_mat16(m16 + M_WN2 * MAT_W, n2w, H, 0, kSi);
    unsigned* m16h = h_mats16h + (size_t)l * M_COUNT * MAT_WH;
    pack_mat16h(m16h + M_WA * MAT_WH, e0w, 2 * H + 2, 0, kS);
    pack_mat16h(m16h + M_WB * MAT_WH, e0w, 2 * H + 2, H, kS);
    pack_mat16h(m16h + M_W2 * MAT_WH, e2w, H, 0, 1.0f);
    pack_mat16h(m16h + M_WC1 * MAT_WH, c0w, H, 0, 1.0f);
    pack_mat16h(m16h + M_WN1A * MAT_WH, n0w, 2 * H, 0, kS);
    pack_mat16h(m16h + M_WN1B * MAT_WH, n0w, 2 * H, H, 1.0f);
    pack_mat16h(m16h + M_WN2 * MAT_WH, n2w, H, 0, kSi);
    pack_mat(mats + M_WA * MAT_F, e0w, 2 * H + 2, 0, kS);
    pack_mat(mats + M_WB * MAT_F, e0w, 2 * H + 2, H, kS);
    pack_mat(mats + M_W2 * MAT_F, e2w, H, 0, 1.0f);
    pack_mat(mats + M_WC1 * MAT_F, c0w, H, 0, 1.0f);
    pack_mat(mats + M_WN1A * MAT_F, n0w, 2 * H, 0, kS);
    pack_mat(mats + M_WN1B * MAT_F, n0w, 2 * H, H, 1.0f);
    pack_mat(mats + M_WN2 * MAT_F, n2w, H, 0, kSi);
    {  // transposes of the unscaled matrices (reverse mode): T[k][o] = W[o][col0 + k]
      float T[EH * EH];
      auto pack_t = [&](int slot, const float* M, int ld, int col0) {
        for (int k = 0; k < H; ++k)
          for (int o = 0; o < H; ++o) T[k * H + o] = M[o * ld + col0 + k];
        pack_mat16(m16 + slot * MAT_W, T, H, 0, 1.0f);
        pack_mat(mats + slot * MAT_F, T, H, 0, 1.0f);
      };
      auto pack_th = [&](int slot, const float* M, int ld, int col0) {  // f16 fragments (fast divergence kernel)
        for (int k = 0; k < H; ++k)
          for (int o = 0; o < H; ++o) T[k * H + o] = M[o * ld + col0 + k];
        pack_mat16h(m16h + slot * MAT_WH, T, H, 0, 1.0f);
      };
      pack_th(M_W2T, e2w, H, 0);
      pack_th(M_WC1T, c0w, H, 0);
      pack_t(M_WAT, e0w, 2 * H + 2, 0);
      pack_t(M_WBT, e0w, 2 * H + 2, H);
      pack_t(M_W2T, e2w, H, 0);
      pack_t(M_WC1T, c0w, H, 0);
      pack_t(M_WN1AT, n0w, 2 * H, 0);
      pack_t(M_WN1BT, n0w, 2 * H, H);
      pack_t(M_WN2T, n2w, H, 0);
    }
    for (int o = 0; o < H; ++o) {
      vecs[V_WRE * EH + o] = kS * e0w[o * (2 * H + 2) + 2 * H];          // w_r[out]  (k = 0: radial)
      vecs[V_WRE * EH + H + o] = kS * e0w[o * (2 * H + 2) + 2 * H + 1];  // w_e[out]  (k = 1: edge_attr)
    }
    pack_vec(vecs + V_B1 * EH, e0b, 1, kS);
    pack_vec(vecs + V_B2 * EH, e2b, 1, kS);
    if (aw) pack_vec(vecs + V_WATT * EH, aw, 1, kSi);
    pack_vec(vecs + V_BC1 * EH, c0b, 1, kS);
    pack_vec(vecs + V_WC2 * EH, c2w, 1, kSi);
    pack_vec(vecs + V_BN1 * EH, n0b, 1, kS);
    pack_vec(vecs + V_BN2 * EH, n2b, 1, 1.0f);
    pack_vec(vecs + V_WRF * EH, e0w + 2 * H, 2 * H + 2, 1.0f);
    pack_vec(vecs + V_WEF * EH, e0w + 2 * H + 1, 2 * H + 2, 1.0f);
    {
      float* vd = h_vecs_div + (size_t)l * VEC_DIV_F;
      pack_vec(vd + EH, e0w + 2 * H, 2 * H + 2, kS);
      pack_vec(vd + 2 * EH, e0w + 2 * H + 1, 2 * H + 2, kS);
      for (int o = 0; o < H; ++o) vd[o] = DIV_ST * (vd[EH + o] + vd[2 * EH + o]);
      pack_vec(vd + 3 * EH, c2w, 1, DIV_SV * kSi);
    }
    vecs[V_COUNT * EH] = ab ? ab[0] : 0.f;
    {  // PREC 2 copy: biases that initialise an f16-path accumulator (W2, Wc1, Wn2) x F16_SX F16_SW, vectors that
       // consume F16_SX-scaled SiLU outputs (attention gate, coordinate head) / F16_SX
      float* vh = h_vecs_h + VEC_EMB_F + (size_t)l * VEC_LAYER_F;
      memcpy(vh, vecs, sizeof(float) * VEC_LAYER_F);
      const float up = F16_SX * F16_SW, dn = 1.0f / F16_SX;
      for (int o = 0; o < H; ++o) {
        vh[V_B2 * EH + o] *= up; vh[V_BC1 * EH + o] *= up; vh[V_BN2 * EH + o] *= up;
        if (F16_NODE_LAYERS) { vh[V_B1 * EH + o] *= up; vh[V_BN1 * EH + o] *= up; }
        vh[V_WATT * EH + o] *= dn; vh[V_WC2 * EH + o] *= dn;
      }
    }
  }
  memcpy(h_vecs_h, h_vecs, sizeof(float) * VEC_EMB_F);
  pita_egnn* net = new pita_egnn();
  net->cfg = *cfg;
  net->shape = shape;
  net->shape_small = shape_small;
  hipError_t e1 = hipMalloc(&net->d_mats, n_mats * sizeof(float));
  hipError_t e2 = hipMalloc(&net->d_vecs, n_vecs * sizeof(float));
  hipError_t e0 = hipMalloc(&net->d_mats16, n_mats16 * sizeof(unsigned));
  if (e0 == hipSuccess) e0 = hipMalloc(&net->d_mats16h, n_mats16h * sizeof(unsigned));
  if (e0 == hipSuccess) e0 = hipMalloc(&net->d_vecs_h, n_vecs * sizeof(float));
  if (e0 == hipSuccess) e0 = hipMalloc(&net->d_vecs_div, (size_t)L * VEC_DIV_F * sizeof(float));
  if (e0 == hipSuccess)
    e0 = hipMemcpy(net->d_vecs_div, h_vecs_div, (size_t)L * VEC_DIV_F * sizeof(float), hipMemcpyHostToDevice);
  if (e0 == hipSuccess) e0 = hipMemcpy(net->d_mats16h, h_mats16h, n_mats16h * sizeof(unsigned), hipMemcpyHostToDevice);
  if (e0 == hipSuccess) e0 = hipMemcpy(net->d_vecs_h, h_vecs_h, n_vecs * sizeof(float), hipMemcpyHostToDevice);
  if (e1 == hipSuccess && e2 == hipSuccess && e0 == hipSuccess) {
    e1 = hipMemcpy(net->d_mats, h_mats, n_mats * sizeof(float), hipMemcpyHostToDevice);
    e2 = hipMemcpy(net->d_vecs, h_vecs, n_vecs * sizeof(float), hipMemcpyHostToDevice);
    e0 = hipMemcpy(net->d_mats16, h_mats16, n_mats16 * sizeof(unsigned), hipMemcpyHostToDevice);
  }
  if (e0 != hipSuccess && e1 == hipSuccess) e1 = e0;
  delete[] h_mats;
  delete[] h_vecs;
  delete[] h_mats16;
  delete[] h_mats16h;
  delete[] h_vecs_h;
  delete[] h_vecs_div;
  if (e1 != hipSuccess || e2 != hipSuccess) {
    (void)hipFree(net->d_mats);
    (void)hipFree(net->d_vecs);
    (void)hipFree(net->d_mats16);
    (void)hipFree(net->d_mats16h);
    (void)hipFree(net->d_vecs_h);
    (void)hipFree(net->d_vecs_div);
    delete net;
    return fail(PITA_EHIP, "pita_egnn_create: device upload failed: %s",
                hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  }
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess) {
    net->device = dev;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) net->n_cu = prop.multiProcessorCount;
  }
  // opt in to the LDS the kernel needs (static limit is 64 KiB)
  size_t lds = shape->lds_bytes(L);
  hipError_t e3 = hipSuccess;
  for (int a = 0; a < 3 && e3 == hipSuccess; ++a)
    for (int b = 0; b < 2 && e3 == hipSuccess; ++b) {
      e3 = ensure_dynamic_lds(reinterpret_cast<const void*>(shape->kernel[a][b]), lds);
      if (e3 == hipSuccess && shape_small)
        e3 = ensure_dynamic_lds(reinterpret_cast<const void*>(shape_small->kernel[a][b]), shape_small->lds_bytes(L));
    }
  if (e3 != hipSuccess) {
    pita_egnn_destroy(net);
    return fail(PITA_EHIP, "pita_egnn_create: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e3));
  }
  *out = net;
  return PITA_OK;
}

extern "C" int pita_egnn_destroy(pita_egnn_t* net) {
  if (!net) return PITA_OK;
  PitaDeviceGuard guard(net->device);
  (void)hipFree(net->d_mats);
  (void)hipFree(net->d_vecs);
  (void)hipFree(net->d_mats16);
  (void)hipFree(net->d_vjp_mark);
  (void)hipFree(net->d_mats16h);
  (void)hipFree(net->d_vecs_h);
  (void)hipFree(net->d_vecs_div);
  (void)hipFree(net->d_ws);
  (void)hipFree(net->d_bk);
  (void)hipFree(net->d_mark);
  (void)hipFree(net->d_divcache);
  delete net;
  return PITA_OK;
}

// the mapping for a batch of B walkers: the small-group shape when the regular one leaves SIMDs with a single wave
static const EgnnShape* shape_for(const pita_egnn_t* net, long long B) {
  const EgnnShape* s = static_cast<const EgnnShape*>(net->shape);
  const EgnnShape* t = static_cast<const EgnnShape*>(net->shape_small);
  if (!t) return s;
  const long long two_per_simd = (long long)net->n_cu * 8;  // waves
  const long long waves_s = (B + s->G - 1) / s->G;
  return waves_s < two_per_simd ? t : s;
}

static int egnn_launch(pita_egnn_t* net, EgnnParams& p, void* stream) {
  PitaDeviceGuard guard(net->device);
  const EgnnShape* s = shape_for(net, p.B);
  const int prec = net->cfg.precision;
  p.mats = net->d_mats;
  p.mats16 = net->d_mats16;
  p.mats16h = net->d_mats16h;
  p.vecs = prec == 2 ? net->d_vecs_h : net->d_vecs;
  p.n_layers = net->cfg.n_layers;
  p.in_nf = net->cfg.in_node_nf;
  p.attention = net->cfg.attention;
  p.tanh_on = net->cfg.tanh;
  p.feature_layout = net->cfg.feature_layout;
  p.coord_scale = net->cfg.coords_range / (float)net->cfg.n_layers;
  if (p.B == 0) return PITA_OK;
  const long long ngroups = (p.B + s->G - 1) / s->G;
  const size_t lds = s->lds_bytes(p.n_layers);
  // resident blocks per CU: limited by LDS and by the 256-VGPR budget (2 waves per SIMD = two 4-wave blocks)
  int blocks_per_cu = (int)((160 * 1024) / lds);
  blocks_per_cu = blocks_per_cu < 1 ? 1 : (blocks_per_cu > s->occ ? s->occ : blocks_per_cu);
  long long want = (ngroups + s->waves - 1) / s->waves;
  long long cap = (long long)net->n_cu * blocks_per_cu;
  // forward modes: one group per wave (plain grid); sampler mode: persistent grid-stride
  unsigned grid = (unsigned)(want < cap ? want : cap);
  if (p.mode != 3) grid = (unsigned)want;
  const int smp = p.mode == 3 ? 1 : 0;
  if (prec == 2 && smp) {  // keep the walkers: the repair launch restarts the affected groups from them
    const size_t need = sizeof(float) * (size_t)p.B * s->n * s->dim + sizeof(int) * (size_t)p.B * s->n;
    if (need > net->bk_bytes) {
      PITA_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));  // an earlier launch may still use the old buffer
      (void)hipFree(net->d_bk);
      net->d_bk = nullptr;
      net->bk_bytes = 0;
      PITA_HIP_CHECK(hipMalloc(&net->d_bk, need));
      net->bk_bytes = need;
    }
    float* xb = static_cast<float*>(net->d_bk);
    PITA_HIP_CHECK(hipMemcpyAsync(xb, p.x, sizeof(float) * (size_t)p.B * s->n * s->dim, hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream));
    p.x_backup = xb;
    p.bad_from = reinterpret_cast<int*>(xb + (size_t)p.B * s->n * s->dim);
  }
  hipLaunchKernelGGL(s->kernel[prec][smp], dim3(grid), dim3(s->waves * 64), lds, (hipStream_t)stream, p);
  PITA_LAUNCH_CHECK();
  if (prec == 2) {
    p.repair = 1;
    p.vecs = net->d_vecs;  // the PREC 1 kernel takes the unscaled vectors
    hipLaunchKernelGGL(s->kernel[1][smp], dim3(grid), dim3(s->waves * 64), lds, (hipStream_t)stream, p);
    PITA_LAUNCH_CHECK();
  }
  return PITA_OK;
}

// MFMA wave-instructions the fused sampler executes per walker-step, counted on the kernel's own loop structure (same
// walker quota / grouping arithmetic as the launch above): per live column tile and layer one partner-table GEMM, one
// own-term GEMM, (N-1) edges x (one f32 k-step + two dense layers) and -- except in the last layer -- three node-model
// GEMMs.  A dense 32x32 layer is 16 f32 MFMAs (precision 0), 12 bf16 MFMAs (precision 1) or -- the per-edge layers and
// Wn2 in precision 2 -- 6 f16 MFMAs.
extern "C" int pita_egnn_sampler_work(const pita_egnn_t* net, int64_t B, double* mfma16_per_walker_step,
                                      double* mfma32_per_walker_step) {
  PITA_REQUIRE(net && B > 0 && mfma16_per_walker_step && mfma32_per_walker_step, "pita_egnn_sampler_work: bad argument");
  const EgnnShape* s = shape_for(net, B);
  const int L = net->cfg.n_layers, N = s->n, G = s->G;
  const size_t lds = s->lds_bytes(L);
  int blocks_per_cu = (int)((160 * 1024) / lds);
  blocks_per_cu = blocks_per_cu < 1 ? 1 : (blocks_per_cu > s->occ ? s->occ : blocks_per_cu);
  const long long ngroups = (B + G - 1) / G;
  long long want = (ngroups + s->waves - 1) / s->waves, cap = (long long)net->n_cu * blocks_per_cu;
  const long long total_waves = (want < cap ? want : cap) * s->waves;
  const long long quota = (B + total_waves - 1) / total_waves;
  double tiles = 0.0;  // live column tiles per step, summed over all waves
  for (long long wbeg = 0; wbeg < B; wbeg += quota) {
    const long long wend = wbeg + quota < B ? wbeg + quota : B;
    for (long long w0 = wbeg; w0 < wend; w0 += G) {
      const long long nw = (wend - w0) < G ? (wend - w0) : G;
      tiles += (double)((nw * N + 31) / 32);
    }
  }
  const double node_dense = tiles * (L * 2.0 + (L - 1) * 2.0);                // Wb, Wa per layer; Wn1a, Wn1b
  const double edge_dense = tiles * (L * 2.0 * (N - 1) + (L - 1) * 1.0);      // W2, Wc1 per edge; Wn2
  const double kstep = tiles * L * (N - 1);                                   // f32 (radial, edge_attr) k-steps
  if (net->cfg.precision == 0) {
    *mfma16_per_walker_step = 0.0;
    *mfma32_per_walker_step = (16.0 * (node_dense + edge_dense) + kstep) / (double)B;
  } else {
    const double f16d = net->cfg.precision == 2 ? 6.0 : 12.0;
    *mfma16_per_walker_step = ((F16_NODE_LAYERS ? f16d : 12.0) * node_dense + f16d * edge_dense) / (double)B;
    *mfma32_per_walker_step = kstep / (double)B;
  }
  return PITA_OK;
}

extern "C" int pita_egnn_sampler_mapping(const pita_egnn_t* net, int64_t B, int* walkers_per_group, int64_t* waves,
                                         int64_t* wave_slots) {
  PITA_REQUIRE(net && B > 0 && walkers_per_group && waves && wave_slots, "pita_egnn_sampler_mapping: bad argument");
  const EgnnShape* s = shape_for(net, B);  // the same arithmetic as egnn_launch (mode 3)
  const size_t lds = s->lds_bytes(net->cfg.n_layers);
  int blocks_per_cu = (int)((160 * 1024) / lds);
  blocks_per_cu = blocks_per_cu < 1 ? 1 : (blocks_per_cu > s->occ ? s->occ : blocks_per_cu);
  const long long ngroups = (B + s->G - 1) / s->G;
  const long long want = (ngroups + s->waves - 1) / s->waves, cap = (long long)net->n_cu * blocks_per_cu;
  *walkers_per_group = s->G;
  *waves = (want < cap ? want : cap) * s->waves;
  *wave_slots = cap * s->waves;
  return PITA_OK;
}

extern "C" int pita_egnn_forward(pita_egnn_t* net, const float* t, const float* x, const float* beta, float* out,
                                 int64_t B, void* stream) {
  PITA_REQUIRE(net && t && x && out && B >= 0, "pita_egnn_forward: null argument");
  PITA_REQUIRE(beta || net->cfg.in_node_nf == 1, "pita_egnn_forward: beta required for in_node_nf=2");
  EgnnParams p{};
  p.mode = 0; p.B = B; p.x_in = x; p.t = t; p.beta = beta; p.out = out;
  return egnn_launch(net, p, stream);
}

extern "C" int pita_egnn_edm(pita_egnn_t* net, int what, const float* h, const float* x, const float* beta, float* out,
                             int64_t B, void* stream) {
  PITA_REQUIRE(net && h && x && out && B >= 0, "pita_egnn_edm: null argument");
  PITA_REQUIRE(what == 1 || what == 2, "pita_egnn_edm: what must be 1 (denoiser) or 2 (score)");
  PITA_REQUIRE(beta || net->cfg.in_node_nf == 1, "pita_egnn_edm: beta required for in_node_nf=2");
  EgnnParams p{};
  p.mode = what; p.B = B; p.x_in = x; p.t = h; p.beta = beta; p.out = out;
  return egnn_launch(net, p, stream);
}

extern "C" int pita_egnn_sampler_run(pita_egnn_t* net, float* x, int64_t B, const float* step_tab, int n_steps,
                                     const float* noise, uint64_t seed, uint64_t walker_offset, int64_t step0,
                                     int remove_mean, float* drift_out, double* stats_out, void* stream) {
  PITA_REQUIRE(net && x && step_tab && B >= 0 && n_steps >= 0, "pita_egnn_sampler_run: bad argument");
  if (n_steps == 0) return PITA_OK;
  EgnnParams p{};
  p.mode = 3; p.B = B; p.x = x; p.step_tab = step_tab; p.n_steps = n_steps; p.noise = noise;
  p.seed = seed; p.walker_offset = walker_offset; p.step0 = step0; p.remove_mean = remove_mean;
  p.drift_out = drift_out; p.stats_out = stats_out;
  return egnn_launch(net, p, stream);
}
