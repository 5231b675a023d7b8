// EGNN backbone with hidden_nf up to 64 and static per-node input features, for gfx950 (MI355X).
//
// Replaces (paths relative to /root/reference/pita/src/models/components/):
//   egnn_dynamics_ad2_cat.py:11-203   EGNN_dynamics_AD2_cat (the alanine-dipeptide backbone of
//                                     configs/model/net/egnn_dynamics_ad2_cat.yaml: hidden 64 x 5 layers, one-hot atom-type
//                                     node features concatenated with t and beta)
//   egnn.py:108-184, 187-346          EGNN.forward, E_GCL (edge_model, coord_model, node_model, coord2radial; the same
//                                     layer as egnn_temp_conditioned.py)
//   score_net.py:13-43                EDM preconditioning (modes 1, 2)
//
// Why a second EGNN kernel: egnn_kernel.hip ties the hidden width to the 32 x 32 MFMA tile (lane = column x 16 of 32
// features).  This one trades speed for generality -- any hidden_nf <= 64, any number of static node features -- with
// the plainest possible mapping: ONE wavefront = one walker, LANE = hidden feature.  A dense layer y = W v is
//     y_f = b_f + sum_k W[f][k] v_k      lane f, k = 0 .. H-1:  v_k by v_readlane_b32 (lane k holds v_k), W[f][.] in registers
// i.e. 2 vector instructions per k; an edge costs two such layers (W2, coord head) + three SiLUs + two wave reductions
// (attention gate, coordinate scalar: DPP trees), ~300 instructions; the per-node layers (Wa, Wb, node MLP) read their
// transposed weight rows from the L2.  fp32 FMA chains throughout (no reduced-precision operands).  Edges are swept
// i-major, j ascending: the reference's index_add order, so per-node sums accumulate in the same order.
// Throughput is that of the vector pipe (~8e5 walker-forwards/s for 22 atoms, 64 x 5): the LJ / DW4 configurations
// stay on the MFMA kernel.
#include <cstdlib>

#include "egnn_wide_common.h"

namespace pita {

constexpr int WIDE_HP = 64;  // lanes of a wave = padded hidden width

// per-layer block of the packed weights (floats); matrices are stored TRANSPOSED, [k][64]: lane f reads W[f][k] at k*64+f
struct WideLayer {
  static constexpr int MAT = WIDE_HP * WIDE_HP;
  static constexpr int WA = 0, WB = MAT, W2 = 2 * MAT, WC1 = 3 * MAT, WN1A = 4 * MAT, WN1B = 5 * MAT, WN2 = 6 * MAT;
  static constexpr int VEC = 7 * MAT;  // then vectors of 64: wr, we, b1, b2, watt, batt (lane 0), bc1, wc2, bn1, bn2
  static constexpr int WR = VEC, WE = VEC + 64, B1 = VEC + 128, B2 = VEC + 192, WATT = VEC + 256, BATT = VEC + 320,
                       BC1 = VEC + 384, WC2 = VEC + 448, BN1 = VEC + 512, BN2 = VEC + 576;
  // behind the vectors: the same seven matrices in NATURAL order, [f][64]: lane k reads W[f][k] at f*64+k -- the operand
  // layout of the TRANSPOSED products y_k = sum_f W[f][k] v_f of the reverse-mode kernel (egnn_wide_vjp_kernel)
  static constexpr int NAT = VEC + 640;
  static constexpr int WAN = NAT, WBN = NAT + MAT, W2N = NAT + 2 * MAT, WC1N = NAT + 3 * MAT, WN1AN = NAT + 4 * MAT,
                       WN1BN = NAT + 5 * MAT, WN2N = NAT + 6 * MAT;
  static constexpr int SIZE = NAT + 7 * MAT;
};
constexpr int WIDE_HEAD = 128;  // emb_t[64], emb_beta[64] in front of the layers

struct WideParams {
  const float* w;
  const float* estatic;
  int n, dim, H, L, attention, tanh_on, has_beta;
  float coord_scale;
  long long B;
  int mode;  // 0 backbone forward (t = its time input), 1 denoiser, 2 score (t = h = sigma^2)
  const float* x;
  const float* t;
  const float* beta;
  float* out;
  int only_bad;  // recompute only the walkers whose `out` holds a non-finite value (repair pass behind the matrix-pipe kernel)
  // mode 3: n_steps Euler-Maruyama steps of the not-debiased reverse SDE in one launch (pita_egnn_wide_sampler_run)
  float* xs;               // [B, n*dim] walkers, in place
  const float* x_backup;   // only_bad: the walkers as they were before the matrix-pipe launch
  const float* step_tab;   // [n_steps][PITA_STEP_STRIDE]
  int n_steps;
  const float* noise;      // nullable [n_steps, B, n*dim]
  unsigned long long seed, walker_offset;
  long long step0;
  int remove_mean;
  double* stats_out;       // nullable [n_steps][4]
  const int* bad_from;     // only_bad: [B*n] first step whose moments the matrix-pipe launch left to this one
  const int* bad_flag;     // only_bad, nullable: 0 = the matrix-pipe launch left nothing non-finite, return at once
};

namespace {

template <int CTRL>
__device__ __forceinline__ float wdpp_add(float v) {
  const int m = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
  return v + __builtin_bit_cast(float, m);
}
// sum over the 64 lanes, fixed tree order, returned to every lane
__device__ __forceinline__ float wwave_sum(float v) {
  v = wdpp_add<0xb1>(v);
  v = wdpp_add<0x4e>(v);
  v = wdpp_add<0x124>(v);
  v = wdpp_add<0x128>(v);
  v = wdpp_add<0x142>(v);
  v = wdpp_add<0x143>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wlane(float v, int k) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}
__device__ __forceinline__ float wsilu(float z) { return z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * z)); }
__device__ __forceinline__ void wfence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// y_f = acc_f + sum_k row[k] v_k  with the weight row of this lane in registers
template <int HK>
__device__ __forceinline__ float dense_reg(const float (&row)[WIDE_HP], float v, float acc) {
#pragma unroll
  for (int k = 0; k < HK; ++k) acc = fmaf(row[k], wlane(v, k), acc);
  return acc;
}
// the same with the transposed matrix in memory ([k][64] floats)
template <int HK>
__device__ __forceinline__ float dense_mem(const float* __restrict__ wt, int lane, float v, float acc) {
#pragma unroll 8
  for (int k = 0; k < HK; ++k) acc = fmaf(wt[k * WIDE_HP + lane], wlane(v, k), acc);
  return acc;
}

// Broadcast through LDS instead of v_readlane_b32 (PITA_WIDE_READLANE=1 at compile time restores the register form):
// lane k parks v_k in a 64-float per-wave slot and every lane reads the vector back four k at a time (all lanes the same
// address: a broadcast ds_read_b128), so a dense layer is 1 store + 16 loads + 64 v_fma instead of 64 v_readlane + 64
// v_fma.  v_readlane writes a scalar register, and those writes turned out to be the bound of the register form: the
// kernels issued ~0.5 instructions per cycle and COMPUTE UNIT whatever the number of resident waves.  Same operands in
// the same k order: bit-identical results.
#ifndef PITA_WIDE_READLANE
#define PITA_WIDE_READLANE 0
#endif
typedef float wf32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bc_put(float* bc, int lane, float v) {
  bc[lane] = v;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <int HK>
__device__ __forceinline__ float dense_reg_b(const float (&row)[WIDE_HP], float* bc, int lane, float v, float acc) {
#if PITA_WIDE_READLANE
  return dense_reg<HK>(row, v, acc);
#else
  bc_put(bc, lane, v);
  float a1 = 0.f, a2 = 0.f, a3 = 0.f;  // four partial sums: a 64-long dependent FMA chain per layer was the kernel's bound
#pragma unroll
  for (int k = 0; k < HK; k += 4) {
    const wf32x4 a = *reinterpret_cast<const wf32x4*>(bc + k);
    acc = fmaf(row[k], a[0], acc); a1 = fmaf(row[k + 1], a[1], a1);
    a2 = fmaf(row[k + 2], a[2], a2); a3 = fmaf(row[k + 3], a[3], a3);
  }
  asm volatile("" ::: "memory");
  return (acc + a1) + (a2 + a3);
#endif
}
template <int HK>
__device__ __forceinline__ float dense_mem_b(const float* __restrict__ wt, float* bc, int lane, float v, float acc) {
#if PITA_WIDE_READLANE
  return dense_mem<HK>(wt, lane, v, acc);
#else
  bc_put(bc, lane, v);
  float a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
  for (int k = 0; k < HK; k += 4) {
    const wf32x4 a = *reinterpret_cast<const wf32x4*>(bc + k);
    acc = fmaf(wt[k * WIDE_HP + lane], a[0], acc); a1 = fmaf(wt[(k + 1) * WIDE_HP + lane], a[1], a1);
    a2 = fmaf(wt[(k + 2) * WIDE_HP + lane], a[2], a2); a3 = fmaf(wt[(k + 3) * WIDE_HP + lane], a[3], a3);
  }
  asm volatile("" ::: "memory");
  return (acc + a1) + (a2 + a3);
#endif
}

// HK = hidden width rounded up to 32 (the k extent of every dense layer); SMP: the all-steps sampler (mode 3) as its own
// instantiation, so that the evaluation modes do not carry the step loop's state through the layers (it spilled there)
template <int HK, bool SMP>
__global__ void __launch_bounds__(256, 2) egnn_wide_kernel(WideParams p) {  // two waves per SIMD: at most 256 registers
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = p.n, DIM = p.dim;
  const int per_wave = 3 * n * WIDE_HP + (SMP ? 4 : 3) * n * 4 + WIDE_HP;
  float* hf = lds + wave * per_wave;     // [n][64] node features
  float* At = hf + n * WIDE_HP;          // [n][64] Wa h_i + b1
  float* Bt = At + n * WIDE_HP;          // [n][64] Wb h_j
  float* pos = Bt + n * WIDE_HP;         // [n][4] positions entering the layer
  float* pos0 = pos + n * 4;             // [n][4] input geometry (frozen edge attribute)
  float* posn = pos0 + n * 4;            // [n][4] positions leaving the layer
  float* xw = posn + n * 4;              // [n][4] the walker's unscaled coordinates across the steps (sampler mode only)
  float* bc = SMP ? xw + n * 4 : xw;     // [64] broadcast slot of the dense layers
  constexpr bool smp = SMP;
  if (p.only_bad && p.bad_flag && *p.bad_flag == 0) return;  // nothing to repair: no pass over the results at all
  const long long nw = (long long)gridDim.x * waves;
  for (long long w = (long long)blockIdx.x * waves + wave; w < p.B; w += nw) {
    if (p.only_bad) {
      bool bad = false;
      const float* res = smp ? p.xs : p.out;
      for (int q = lane; q < n * DIM; q += 64) bad = bad || !__builtin_isfinite(res[w * n * DIM + q]);
      if (!__any(bad)) continue;  // wave-uniform
    }
    if (smp) {  // (the evaluation modes read the walker from memory where they need it: no table, no extra live state)
      const float* src = p.only_bad ? p.x_backup : p.xs;
      for (int q = lane; q < n * DIM; q += 64) {
        const int i = q / DIM, k = q - i * DIM;
        xw[i * 4 + k] = src[w * n * DIM + q];
      }
      wfence();
    }
    const int nsteps = smp ? p.n_steps : 1;
    for (int step = 0; step < nsteps; ++step) {
    float tv, bet, c_s = 0.f, c_in = 1.f, c_out = 1.f, tfeat, hval = 1.f;
    float g2 = 0.f, gamma = 0.f, dt = 0.f, noise_scale = 0.f, sqrt_dt = 0.f;
    if (smp) {  // per-step scalars of the host table (sde_integration.py: build_step_table), as the matrix-pipe kernel
      const float* st = p.step_tab + (size_t)step * PITA_STEP_STRIDE;
      c_s = st[PITA_ST_CS]; c_in = st[PITA_ST_CIN]; c_out = st[PITA_ST_COUT]; tfeat = st[PITA_ST_CNOISE];
      hval = st[PITA_ST_H]; bet = p.has_beta ? st[PITA_ST_BETA] : 0.f; tv = hval;
      g2 = st[PITA_ST_G2]; gamma = st[PITA_ST_GAMMA]; dt = st[PITA_ST_DT];
      noise_scale = st[PITA_ST_NOISE_SCALE]; sqrt_dt = st[PITA_ST_SQRT_DT];
    } else {
      tv = p.t[w];
      bet = p.has_beta ? p.beta[w] : 0.f;
      tfeat = tv;
      if (p.mode != 0) {  // score_net.py:26-29
        hval = tv;
        c_s = 1.0f / (1.0f + tv);
        c_in = 1.0f / sqrtf(1.0f + tv);
        c_out = sqrtf(tv) * c_in;
        tfeat = 0.125f * logf(tv);
      }
    }
    // scaled input coordinates
    for (int q = lane; q < n * DIM; q += 64) {
      const int i = q / DIM, k = q - i * DIM;
      const float v = c_in * (smp ? xw[i * 4 + k] : p.x[w * n * DIM + q]);
      pos[i * 4 + k] = v;
      pos0[i * 4 + k] = v;
    }
    // node features: embedding of [static one-hot features, t, beta] (egnn_dynamics_ad2_cat.py:157-184, egnn.py:179)
    {
      const float et = p.w[lane], eb = p.w[64 + lane];
      for (int i = 0; i < n; ++i) hf[i * WIDE_HP + lane] = fmaf(et, tfeat, fmaf(eb, bet, p.estatic[i * WIDE_HP + lane]));
    }
    wfence();
    for (int l = 0; l < p.L; ++l) {
      const float* wl = p.w + WIDE_HEAD + (size_t)l * WideLayer::SIZE;
      // per-node first-layer terms of the edge MLP: W1 [h_i, h_j, r, e] = Wa h_i + Wb h_j + w_r r + w_e e
      for (int i = 0; i < n; ++i) {
        const float hv = hf[i * WIDE_HP + lane];
        At[i * WIDE_HP + lane] = dense_mem_b<HK>(wl + WideLayer::WA, bc, lane, hv, wl[WideLayer::B1 + lane]);
        Bt[i * WIDE_HP + lane] = dense_mem_b<HK>(wl + WideLayer::WB, bc, lane, hv, 0.f);
      }
      wfence();
      float w2[WIDE_HP], wc1[WIDE_HP];
#pragma unroll
      for (int k = 0; k < HK; ++k) {
        w2[k] = wl[WideLayer::W2 + k * WIDE_HP + lane];
        wc1[k] = wl[WideLayer::WC1 + k * WIDE_HP + lane];
      }
      const float wr = wl[WideLayer::WR + lane], we = wl[WideLayer::WE + lane], b2 = wl[WideLayer::B2 + lane];
      const float watt = wl[WideLayer::WATT + lane], batt = wl[WideLayer::BATT], bc1 = wl[WideLayer::BC1 + lane];
      const float wc2 = wl[WideLayer::WC2 + lane];
      const bool last = (l == p.L - 1);
      for (int i = 0; i < n; ++i) {
        const float Ai = At[i * WIDE_HP + lane];
        float pi[3] = {0.f, 0.f, 0.f}, p0i[3] = {0.f, 0.f, 0.f}, xacc[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < DIM; ++k) { pi[k] = pos[i * 4 + k]; p0i[k] = pos0[i * 4 + k]; }
        float agg = 0.f;
        for (int j = 0; j < n; ++j) {
          if (j == i) continue;
          float df[3] = {0.f, 0.f, 0.f}, radial = 0.f, ea = 0.f;
          for (int k = 0; k < DIM; ++k) {  // coord2radial (egnn.py E_GCL), frozen edge attribute (ad2_cat.py:186)
            df[k] = pi[k] - pos[j * 4 + k];
            radial = fmaf(df[k], df[k], radial);
            const float e0 = p0i[k] - pos0[j * 4 + k];
            ea = fmaf(e0, e0, ea);
          }
          float m = wsilu(fmaf(we, ea, fmaf(wr, radial, Ai + Bt[j * WIDE_HP + lane])));
          m = wsilu(dense_reg_b<HK>(w2, bc, lane, m, b2));
          if (p.attention) m *= fast_sigmoid(wwave_sum(watt * m) + batt);
          agg += m;
          const float c1 = wsilu(dense_reg_b<HK>(wc1, bc, lane, m, bc1));
          float cs = wwave_sum(wc2 * c1);
          if (p.tanh_on) cs = accurate_tanh(cs) * p.coord_scale;
          const float inv = 1.0f / (sqrtf(radial + 1e-8f) + 1.0f);
          for (int k = 0; k < DIM; ++k) xacc[k] = fmaf(df[k] * inv, cs, xacc[k]);
        }
        if (lane < DIM) posn[i * 4 + lane] = pi[lane < 3 ? lane : 0] + xacc[lane < 3 ? lane : 0];
        if (!last) {  // node model, recurrent (the last layer's node update is dead: h_final is discarded)
          const float hv = hf[i * WIDE_HP + lane];
          float z = dense_mem_b<HK>(wl + WideLayer::WN1A, bc, lane, hv, wl[WideLayer::BN1 + lane]);
          z = dense_mem_b<HK>(wl + WideLayer::WN1B, bc, lane, agg, z);
          const float o = dense_mem_b<HK>(wl + WideLayer::WN2, bc, lane, wsilu(z), wl[WideLayer::BN2 + lane]);
          hf[i * WIDE_HP + lane] = hv + o;  // A / B tables of this layer were built from the old features
        }
      }
      wfence();
      for (int q = lane; q < n * 4; q += 64) pos[q] = posn[q];
      wfence();
    }
    // vel = x_final - x, mean-free over the particles; EDM combination for modes 1, 2
    float mean[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < DIM; ++k) {
      float s = 0.f;
      for (int i = 0; i < n; ++i) s += pos[i * 4 + k] - pos0[i * 4 + k];
      mean[k] = s / (float)n;
    }
    if (!smp) {
      for (int q = lane; q < n * DIM; q += 64) {
        const int i = q / DIM, k = q - i * DIM;
        const float F = (pos[i * 4 + k] - pos0[i * 4 + k]) - mean[k];
        float o = F;
        if (p.mode != 0) {
          const float xc = p.x[w * n * DIM + q];
          o = c_s * xc + c_out * F;
          if (p.mode == 2) o = (o - xc) / hval;
        }
        p.out[w * n * DIM + q] = o;
      }
    } else {
      // reverse-SDE Euler-Maruyama update + remove_mean: the arithmetic of egnn_wide64_kernel's sampler mode
      float st_d = 0.f, st_d2 = 0.f, st_n = 0.f, st_n2 = 0.f;
      for (int i = lane; i < n; i += 64) {
        float xi[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.noise) {
          for (int k = 0; k < DIM; ++k) xi[k] = p.noise[((size_t)step * p.B * n + (size_t)(w * n + i)) * DIM + k];
        } else {
          philox_normal4(p.seed, p.walker_offset + (unsigned long long)w, p.step0 + step, (uint32_t)i, xi);
        }
        const bool take = !p.only_bad || !p.bad_from || step >= p.bad_from[w * n + i];
        for (int k = 0; k < DIM; ++k) {
          const float F = (pos[i * 4 + k] - pos0[i * 4 + k]) - mean[k];
          const float xc = xw[i * 4 + k];
          const float Dth = c_s * xc + c_out * F;
          const float drift = gamma * (((Dth - xc) / hval) * g2);
          const float dif = noise_scale * xi[k];
          if (p.stats_out && take) {
            st_d += drift; st_d2 = fmaf(drift, drift, st_d2);
            st_n += dif; st_n2 = fmaf(dif, dif, st_n2);
          }
          posn[i * 4 + k] = xc + (drift * dt + (dif * sqrt_dt));
        }
      }
      if (p.stats_out) {
        double m4[4] = {(double)st_d, (double)st_d2, (double)st_n, (double)st_n2};
        for (int qq = 0; qq < 4; ++qq) {
          for (int o = 32; o > 0; o >>= 1) m4[qq] += __shfl_xor(m4[qq], o, 64);
          if (lane == 0) atomicAdd(p.stats_out + (size_t)step * 4 + qq, m4[qq]);
        }
      }
      wfence();
      float mu[3] = {0.f, 0.f, 0.f};
      if (p.remove_mean) {
        for (int k = 0; k < DIM; ++k) {
          float sum = 0.f;
          for (int i = 0; i < n; ++i) sum += posn[i * 4 + k];
          mu[k] = sum / (float)n;
        }
      }
      for (int q = lane; q < n * DIM; q += 64) {
        const int i = q / DIM, k = q - i * DIM;
        xw[i * 4 + k] = posn[i * 4 + k] - mu[k];
      }
    }
    wfence();
    }  // steps
    if (smp) {
      for (int q = lane; q < n * DIM; q += 64) {
        const int i = q / DIM, k = q - i * DIM;
        p.xs[w * n * DIM + q] = xw[i * 4 + k];
      }
    }
    wfence();
  }
}

}  // namespace

// ---------------------------------------------------------------------------- forward-mode derivative (vector pipe)
// dD = J_x D(h, x) . vx + dD/dh . vh of the EDM denoiser D(h, x) = c_s x + c_out F(c_noise(h), c_in(h) x, beta) around
// this backbone, ONE tangent direction per launch beside the primal (the building block the debiased Feynman-Kac
// regime needs, sdes.py:151-239: trace J_x D, J_x D^T x and <x, dD/dh> are sums of such directions; the reference takes
// them from vmap(jacrev) / autograd).  Same mapping as egnn_wide_kernel -- one wavefront = one walker, lane = hidden
// feature -- with a tangent beside every primal quantity: the node features, the partner table Wb dh_j, the positions.
// A tangent dense layer costs what the primal one does (v_readlane + v_fma per k on the same weight row); the SiLU
// derivative sigma (1 + z (1 - sigma)) re-uses the primal's sigmoid.  fp32 FMA chains like the primal.
struct WideJvpParams {
  WideParams base;       // mode is ignored: the denoiser (mode 1) is differentiated
  const float* vx;       // nullable [B, n*dim]: position direction; null -> unit vector e_dir (dir >= 0) or zero (dir < 0)
  const float* vh;       // nullable [B]: direction in h
  int dir;
  float* dout;           // nullable [B, n*dim]
  float* dot_out;        // nullable: dot_out[b * dot_stride + dot_off] = <x_b, dD_b>
  long long dot_stride, dot_off;
  float* diag_acc;       // nullable: diag_acc[b] += dD[b, dir]
  const int* only_bad;   // nullable [B]: process only the walkers the matrix-pipe kernel flagged (egnn_wide_mfma_jvp_kernel.hip)
};

namespace {

// silu(z) and d silu / dz from one sigmoid
__device__ __forceinline__ float wsilu_d(float z, float& dz) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * z));
  dz = sg * fmaf(z, 1.0f - sg, 1.0f);
  return z * sg;
}
// primal and tangent of one dense layer on the same weight row (registers)
template <int HK>
__device__ __forceinline__ void dense_reg2(const float (&row)[WIDE_HP], float* bc, int lane, float v, float dv, float& acc,
                                           float& dacc) {
#if PITA_WIDE_READLANE
#pragma unroll
  for (int k = 0; k < HK; ++k) {
    acc = fmaf(row[k], wlane(v, k), acc);
    dacc = fmaf(row[k], wlane(dv, k), dacc);
  }
#else
  bc[lane] = v;
  bc_put(bc + WIDE_HP, lane, dv);
  float pa[4] = {acc, 0.f, 0.f, 0.f}, pd[4] = {dacc, 0.f, 0.f, 0.f};  // same partial sums as dense_reg_b for the primal
#pragma unroll
  for (int k = 0; k < HK; k += 4) {
    const wf32x4 a = *reinterpret_cast<const wf32x4*>(bc + k), b = *reinterpret_cast<const wf32x4*>(bc + WIDE_HP + k);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pa[e] = fmaf(row[k + e], a[e], pa[e]);
      pd[e] = fmaf(row[k + e], b[e], pd[e]);
    }
  }
  asm volatile("" ::: "memory");
  acc = (pa[0] + pa[1]) + (pa[2] + pa[3]);
  dacc = (pd[0] + pd[1]) + (pd[2] + pd[3]);
#endif
}
template <int HK>
__device__ __forceinline__ void dense_mem2(const float* __restrict__ wt, float* bc, int lane, float v, float dv, float& acc,
                                           float& dacc) {
#if PITA_WIDE_READLANE
#pragma unroll 8
  for (int k = 0; k < HK; ++k) {
    const float w = wt[k * WIDE_HP + lane];
    acc = fmaf(w, wlane(v, k), acc);
    dacc = fmaf(w, wlane(dv, k), dacc);
  }
#else
  bc[lane] = v;
  bc_put(bc + WIDE_HP, lane, dv);
  float pa[4] = {acc, 0.f, 0.f, 0.f}, pd[4] = {dacc, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int k = 0; k < HK; k += 4) {
    const wf32x4 a = *reinterpret_cast<const wf32x4*>(bc + k), b = *reinterpret_cast<const wf32x4*>(bc + WIDE_HP + k);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float w = wt[(k + e) * WIDE_HP + lane];
      pa[e] = fmaf(w, a[e], pa[e]);
      pd[e] = fmaf(w, b[e], pd[e]);
    }
  }
  asm volatile("" ::: "memory");
  acc = (pa[0] + pa[1]) + (pa[2] + pa[3]);
  dacc = (pd[0] + pd[1]) + (pd[2] + pd[3]);
#endif
}

template <int HK>
__global__ void __launch_bounds__(256) egnn_wide_jvp_kernel(WideJvpParams q) {
  const WideParams& p = q.base;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = p.n, DIM = p.dim;
  const int per_wave = 5 * n * WIDE_HP + 6 * n * 4 + 2 * WIDE_HP;
  float* hf = lds + wave * per_wave;     // [n][64] node features
  float* At = hf + n * WIDE_HP;          // [n][64] Wa h_i + b1
  float* Bt = At + n * WIDE_HP;          // [n][64] Wb h_j
  float* dhf = Bt + n * WIDE_HP;         // tangents of hf and Bt (Wa dh_i is formed per node on the fly)
  float* dBt = dhf + n * WIDE_HP;
  float* pos = dBt + n * WIDE_HP;        // [n][4] each: pos, pos0, posn and their tangents
  float* pos0 = pos + n * 4;
  float* posn = pos0 + n * 4;
  float* dpos = posn + n * 4;
  float* dpos0 = dpos + n * 4;
  float* dposn = dpos0 + n * 4;
  float* bc = dposn + n * 4;             // [2][64] broadcast slots (primal, tangent) of the dense layers
  const long long nw = (long long)gridDim.x * waves;
  for (long long w = (long long)blockIdx.x * waves + wave; w < p.B; w += nw) {
    if (q.only_bad && q.only_bad[w] == 0) continue;  // wave-uniform
    const float hval = p.t[w];
    const float bet = p.has_beta ? p.beta[w] : 0.f;
    const float vh = q.vh ? q.vh[w] : 0.f;
    // score_net.py:26-29 and their h-derivatives
    const float c_s = 1.0f / (1.0f + hval), c_in = 1.0f / sqrtf(1.0f + hval), sh = sqrtf(hval);
    const float c_out = sh * c_in, tfeat = 0.125f * logf(hval);
    const float dc_s = -c_s * c_s, dc_in = -0.5f * c_in * c_s, dc_out = 0.5f * c_in / sh + sh * dc_in;
    const float dtf = vh * (0.125f / hval);
    for (int qd = lane; qd < n * DIM; qd += 64) {
      const int i = qd / DIM, k = qd - i * DIM;
      const float xv = p.x[w * n * DIM + qd];
      const float dx = q.vx ? q.vx[w * n * DIM + qd] : (qd == q.dir ? 1.0f : 0.0f);
      const float v = c_in * xv, dv = fmaf(c_in, dx, (vh * dc_in) * xv);
      pos[i * 4 + k] = v; pos0[i * 4 + k] = v;
      dpos[i * 4 + k] = dv; dpos0[i * 4 + k] = dv;
    }
    {
      const float et = p.w[lane], eb = p.w[64 + lane];
      for (int i = 0; i < n; ++i) {
        hf[i * WIDE_HP + lane] = fmaf(et, tfeat, fmaf(eb, bet, p.estatic[i * WIDE_HP + lane]));
        dhf[i * WIDE_HP + lane] = et * dtf;
      }
    }
    wfence();
    for (int l = 0; l < p.L; ++l) {
      const float* wl = p.w + WIDE_HEAD + (size_t)l * WideLayer::SIZE;
      for (int i = 0; i < n; ++i) {
        const float hv = hf[i * WIDE_HP + lane], dhv = dhf[i * WIDE_HP + lane];
        At[i * WIDE_HP + lane] = dense_mem_b<HK>(wl + WideLayer::WA, bc, lane, hv, wl[WideLayer::B1 + lane]);
        float b = 0.f, db = 0.f;
        dense_mem2<HK>(wl + WideLayer::WB, bc, lane, hv, dhv, b, db);
        Bt[i * WIDE_HP + lane] = b;
        dBt[i * WIDE_HP + lane] = db;
      }
      wfence();
      float w2[WIDE_HP], wc1[WIDE_HP];
#pragma unroll
      for (int k = 0; k < HK; ++k) {
        w2[k] = wl[WideLayer::W2 + k * WIDE_HP + lane];
        wc1[k] = wl[WideLayer::WC1 + k * WIDE_HP + lane];
      }
      const float wr = wl[WideLayer::WR + lane], we = wl[WideLayer::WE + lane], b2 = wl[WideLayer::B2 + lane];
      const float watt = wl[WideLayer::WATT + lane], batt = wl[WideLayer::BATT], bc1 = wl[WideLayer::BC1 + lane];
      const float wc2 = wl[WideLayer::WC2 + lane];
      const bool last = (l == p.L - 1);
      for (int i = 0; i < n; ++i) {
        const float Ai = At[i * WIDE_HP + lane];
        const float dAi = dense_mem_b<HK>(wl + WideLayer::WA, bc, lane, dhf[i * WIDE_HP + lane], 0.f);
        float pi[3] = {0.f, 0.f, 0.f}, p0i[3] = {0.f, 0.f, 0.f}, xacc[3] = {0.f, 0.f, 0.f};
        float dpi[3] = {0.f, 0.f, 0.f}, dp0i[3] = {0.f, 0.f, 0.f}, dxacc[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < DIM; ++k) {
          pi[k] = pos[i * 4 + k]; p0i[k] = pos0[i * 4 + k];
          dpi[k] = dpos[i * 4 + k]; dp0i[k] = dpos0[i * 4 + k];
        }
        float agg = 0.f, dagg = 0.f;
        for (int j = 0; j < n; ++j) {
          if (j == i) continue;
          float df[3] = {0.f, 0.f, 0.f}, ddf[3] = {0.f, 0.f, 0.f}, radial = 0.f, ea = 0.f, dradial = 0.f, dea = 0.f;
          for (int k = 0; k < DIM; ++k) {
            df[k] = pi[k] - pos[j * 4 + k];
            ddf[k] = dpi[k] - dpos[j * 4 + k];
            radial = fmaf(df[k], df[k], radial);
            dradial = fmaf(df[k], ddf[k], dradial);
            const float e0 = p0i[k] - pos0[j * 4 + k];
            ea = fmaf(e0, e0, ea);
            dea = fmaf(e0, dp0i[k] - dpos0[j * 4 + k], dea);
          }
          dradial *= 2.0f; dea *= 2.0f;
          float g1, g2, gc;
          const float z1 = fmaf(we, ea, fmaf(wr, radial, Ai + Bt[j * WIDE_HP + lane]));
          const float dz1 = fmaf(we, dea, fmaf(wr, dradial, dAi + dBt[j * WIDE_HP + lane]));
          const float m1 = wsilu_d(z1, g1), dm1 = g1 * dz1;
          float z2 = b2, dz2 = 0.f;
          dense_reg2<HK>(w2, bc, lane, m1, dm1, z2, dz2);
          float m = wsilu_d(z2, g2), dm = g2 * dz2;
          if (p.attention) {
            const float a = fast_sigmoid(wwave_sum(watt * m) + batt);
            const float da = a * (1.0f - a) * wwave_sum(watt * dm);
            dm = fmaf(dm, a, m * da);
            m *= a;
          }
          agg += m; dagg += dm;
          float zc = bc1, dzc = 0.f;
          dense_reg2<HK>(wc1, bc, lane, m, dm, zc, dzc);
          const float c1 = wsilu_d(zc, gc), dc1 = gc * dzc;
          float cs = wwave_sum(wc2 * c1), dcs = wwave_sum(wc2 * dc1);
          if (p.tanh_on) {
            const float th = accurate_tanh(cs);
            dcs = (1.0f - th * th) * p.coord_scale * dcs;
            cs = th * p.coord_scale;
          }
          const float sq = sqrtf(radial + 1e-8f), inv = 1.0f / (sq + 1.0f);
          const float dinv = -inv * inv * (0.5f * dradial / sq);
          const float dsc = fmaf(dinv, cs, inv * dcs);  // d(inv cs)
          for (int k = 0; k < DIM; ++k) {
            xacc[k] = fmaf(df[k] * inv, cs, xacc[k]);
            dxacc[k] = fmaf(ddf[k], inv * cs, fmaf(df[k], dsc, dxacc[k]));
          }
        }
        if (lane < DIM) {
          const int k = lane < 3 ? lane : 0;
          posn[i * 4 + lane] = pi[k] + xacc[k];
          dposn[i * 4 + lane] = dpi[k] + dxacc[k];
        }
        if (!last) {
          const float hv = hf[i * WIDE_HP + lane], dhv = dhf[i * WIDE_HP + lane];
          float z = wl[WideLayer::BN1 + lane], dz = 0.f, gz;
          dense_mem2<HK>(wl + WideLayer::WN1A, bc, lane, hv, dhv, z, dz);
          dense_mem2<HK>(wl + WideLayer::WN1B, bc, lane, agg, dagg, z, dz);
          const float sz = wsilu_d(z, gz);
          float o = wl[WideLayer::BN2 + lane], d_o = 0.f;
          dense_mem2<HK>(wl + WideLayer::WN2, bc, lane, sz, gz * dz, o, d_o);
          hf[i * WIDE_HP + lane] = hv + o;
          dhf[i * WIDE_HP + lane] = dhv + d_o;
        }
      }
      wfence();
      for (int qd = lane; qd < n * 4; qd += 64) { pos[qd] = posn[qd]; dpos[qd] = dposn[qd]; }
      wfence();
    }
    float mean[3] = {0.f, 0.f, 0.f}, dmean[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < DIM; ++k) {
      float s = 0.f, ds = 0.f;
      for (int i = 0; i < n; ++i) { s += pos[i * 4 + k] - pos0[i * 4 + k]; ds += dpos[i * 4 + k] - dpos0[i * 4 + k]; }
      mean[k] = s / (float)n; dmean[k] = ds / (float)n;
    }
    float dot = 0.f, diag = 0.f;
    for (int qd = lane; qd < n * DIM; qd += 64) {
      const int i = qd / DIM, k = qd - i * DIM;
      const float F = (pos[i * 4 + k] - pos0[i * 4 + k]) - mean[k];
      const float dF = (dpos[i * 4 + k] - dpos0[i * 4 + k]) - dmean[k];
      const float xc = p.x[w * n * DIM + qd];
      const float dx = q.vx ? q.vx[w * n * DIM + qd] : (qd == q.dir ? 1.0f : 0.0f);
      const float dD = fmaf(c_s, dx, fmaf(c_out, dF, vh * fmaf(dc_s, xc, dc_out * F)));
      if (p.out) p.out[w * n * DIM + qd] = fmaf(c_s, xc, c_out * F);
      if (q.dout) q.dout[w * n * DIM + qd] = dD;
      dot = fmaf(xc, dD, dot);
      if (qd == q.dir) diag = dD;
    }
    if (q.dot_out) {
      const float s = wwave_sum(dot);
      if (lane == 0) q.dot_out[w * q.dot_stride + q.dot_off] = s;
    }
    if (q.diag_acc && q.dir >= 0) {
      const float s = wwave_sum(diag);
      if (lane == 0) q.diag_acc[w] += s;
    }
    wfence();
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Reverse mode: vjp = J_x D^T cot for a per-walker cotangent (default: x itself, what grad_x E_theta needs --
// energy_net.py:51-62 gets it from autograd) and <cot, dD/dh> (the h-derivative term of dE_theta/dt, sdes.py:218) from ONE
// sweep, instead of dim + 1 forward-mode launches.  Same mapping as the kernels above (one wavefront = one walker, lane =
// hidden feature).  Forward sweep with per-layer checkpoints (features and positions entering the layer, the node
// model's pre-activation) in a per-wave global scratch; backward sweep layer by layer: a node's own adjoint sums stay in
// registers, what an edge sends to its partner j (the W_b path, the position and edge-attribute adjoints) is added into
// per-wave LDS tables -- edges are swept one at a time by the whole wave, so those read-modify-writes never collide.
// Every edge is recomputed with its SiLU derivatives; the transposed products read the natural-order weight copies
// (W2^T and Wc1^T rows in registers beside the forward rows: one wave per SIMD).
struct WideVjpParams {
  WideParams base;   // x, t (= h), beta, out (nullable: the denoiser)
  const float* cot;  // nullable [B, n*dim]: cotangent (null: x)
  float* vjp;        // [B, n*dim]
  float* dot_h;      // nullable [B]
  float* ws;         // checkpoints: [wave slot][L][2 n 64 + n 4]
};

namespace {

template <int HK>
__global__ void __launch_bounds__(256, 1) egnn_wide_vjp_kernel(WideVjpParams q) {
  const WideParams& p = q.base;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = p.n, DIM = p.dim;
  const int per_wave = 5 * n * WIDE_HP + 6 * n * 4 + WIDE_HP;
  float* hf = lds + wave * per_wave;     // [n][64] features entering the layer
  float* At = hf + n * WIDE_HP;          // [n][64] Wa h_i + b1
  float* Bt = At + n * WIDE_HP;          // [n][64] Wb h_j
  float* hb = Bt + n * WIDE_HP;          // [n][64] adjoint of the features leaving the layer (then: entering it)
  float* TB = hb + n * WIDE_HP;          // [n][64] adjoint of Bt, summed over the edges that read row j
  float* pos = TB + n * WIDE_HP;         // [n][4] positions entering the layer
  float* pos0 = pos + n * 4;             // [n][4] input geometry
  float* posn = pos0 + n * 4;            // [n][4] positions leaving the layer (forward sweep)
  float* pb = posn + n * 4;              // [n][4] adjoint of the positions leaving the layer
  float* pbn = pb + n * 4;               // [n][4] adjoint of the positions entering it (being summed)
  float* p0b = pbn + n * 4;              // [n][4] adjoint of the input geometry (edge attribute of every layer, and -u)
  float* bc = p0b + n * 4;               // [64] broadcast slot of the dense layers
  const size_t ck_layer = (size_t)2 * n * WIDE_HP + (size_t)n * 4;
  float* ws = q.ws + (size_t)(blockIdx.x * waves + wave) * p.L * ck_layer;
  const bool want_h = q.dot_h != nullptr;
  const long long nw = (long long)gridDim.x * waves;
  for (long long w = (long long)blockIdx.x * waves + wave; w < p.B; w += nw) {
    const float hval = p.t[w];
    const float bet = p.has_beta ? p.beta[w] : 0.f;
    const float c_s = 1.0f / (1.0f + hval), c_in = 1.0f / sqrtf(1.0f + hval), sh = sqrtf(hval);
    const float c_out = sh * c_in, tfeat = 0.125f * logf(hval);
    const float dc_s = -c_s * c_s, dc_in = -0.5f * c_in * c_s, dc_out = 0.5f * c_in / sh + sh * dc_in;
    for (int qd = lane; qd < n * 4; qd += 64) { pos[qd] = 0.f; pos0[qd] = 0.f; pb[qd] = 0.f; pbn[qd] = 0.f; p0b[qd] = 0.f; }
    wfence();
    for (int qd = lane; qd < n * DIM; qd += 64) {
      const int i = qd / DIM, k = qd - i * DIM;
      const float v = c_in * p.x[w * n * DIM + qd];
      pos[i * 4 + k] = v;
      pos0[i * 4 + k] = v;
    }
    const float et = p.w[lane], eb = p.w[64 + lane];
    for (int i = 0; i < n; ++i) {
      hf[i * WIDE_HP + lane] = fmaf(et, tfeat, fmaf(eb, bet, p.estatic[i * WIDE_HP + lane]));
      hb[i * WIDE_HP + lane] = 0.f;
    }
    wfence();
    // ------------------------------------------------------------------ forward sweep, checkpoints
    for (int l = 0; l < p.L; ++l) {
      const float* wl = p.w + WIDE_HEAD + (size_t)l * WideLayer::SIZE;
      float* ck = ws + (size_t)l * ck_layer;
      for (int i = 0; i < n; ++i) {
        const float hv = hf[i * WIDE_HP + lane];
        ck[i * WIDE_HP + lane] = hv;
        At[i * WIDE_HP + lane] = dense_mem_b<HK>(wl + WideLayer::WA, bc, lane, hv, wl[WideLayer::B1 + lane]);
        Bt[i * WIDE_HP + lane] = dense_mem_b<HK>(wl + WideLayer::WB, bc, lane, hv, 0.f);
      }
      for (int qd = lane; qd < n * 4; qd += 64) ck[2 * n * WIDE_HP + qd] = pos[qd];
      wfence();
      float w2[WIDE_HP], wc1[WIDE_HP];
#pragma unroll
      for (int k = 0; k < HK; ++k) {
        w2[k] = wl[WideLayer::W2 + k * WIDE_HP + lane];
        wc1[k] = wl[WideLayer::WC1 + k * WIDE_HP + lane];
      }
      const float wr = wl[WideLayer::WR + lane], we = wl[WideLayer::WE + lane], b2 = wl[WideLayer::B2 + lane];
      const float watt = wl[WideLayer::WATT + lane], batt = wl[WideLayer::BATT], bc1 = wl[WideLayer::BC1 + lane];
      const float wc2 = wl[WideLayer::WC2 + lane];
      const bool last = (l == p.L - 1);
      for (int i = 0; i < n; ++i) {
        const float Ai = At[i * WIDE_HP + lane];
        float pi[3] = {0.f, 0.f, 0.f}, p0i[3] = {0.f, 0.f, 0.f}, xacc[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < DIM; ++k) { pi[k] = pos[i * 4 + k]; p0i[k] = pos0[i * 4 + k]; }
        float agg = 0.f;
        for (int j = 0; j < n; ++j) {
          if (j == i) continue;
          float df[3] = {0.f, 0.f, 0.f}, radial = 0.f, ea = 0.f;
          for (int k = 0; k < DIM; ++k) {
            df[k] = pi[k] - pos[j * 4 + k];
            radial = fmaf(df[k], df[k], radial);
            const float e0 = p0i[k] - pos0[j * 4 + k];
            ea = fmaf(e0, e0, ea);
          }
          float m = wsilu(fmaf(we, ea, fmaf(wr, radial, Ai + Bt[j * WIDE_HP + lane])));
          m = wsilu(dense_reg_b<HK>(w2, bc, lane, m, b2));
          if (p.attention) m *= fast_sigmoid(wwave_sum(watt * m) + batt);
          agg += m;
          const float c1 = wsilu(dense_reg_b<HK>(wc1, bc, lane, m, bc1));
          float cs = wwave_sum(wc2 * c1);
          if (p.tanh_on) cs = accurate_tanh(cs) * p.coord_scale;
          const float inv = 1.0f / (sqrtf(radial + 1e-8f) + 1.0f);
          for (int k = 0; k < DIM; ++k) xacc[k] = fmaf(df[k] * inv, cs, xacc[k]);
        }
        if (lane < DIM) posn[i * 4 + lane] = pi[lane < 3 ? lane : 0] + xacc[lane < 3 ? lane : 0];
        if (!last) {
          const float hv = hf[i * WIDE_HP + lane];
          float z = dense_mem_b<HK>(wl + WideLayer::WN1A, bc, lane, hv, wl[WideLayer::BN1 + lane]);
          z = dense_mem_b<HK>(wl + WideLayer::WN1B, bc, lane, agg, z);
          ck[(n + i) * WIDE_HP + lane] = z;
          const float o = dense_mem_b<HK>(wl + WideLayer::WN2, bc, lane, wsilu(z), wl[WideLayer::BN2 + lane]);
          hf[i * WIDE_HP + lane] = hv + o;
        }
      }
      wfence();
      for (int qd = lane; qd < n * 4; qd += 64) pos[qd] = posn[qd];
      wfence();
    }
    // F = (pos^L - pos0) - mean;  D = c_s x + c_out F;  adjoint of u = pos^L - pos0:  c_out (cot - mean_i cot)
    float mean[3] = {0.f, 0.f, 0.f}, cmean[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < DIM; ++k) {
      float s = 0.f, sc = 0.f;
      for (int i = 0; i < n; ++i) {
        s += pos[i * 4 + k] - pos0[i * 4 + k];
        sc += q.cot ? q.cot[(w * n + i) * DIM + k] : p.x[(w * n + i) * DIM + k];
      }
      mean[k] = s / (float)n;
      cmean[k] = sc / (float)n;
    }
    float hpart = 0.f;  // this lane's share of <cot, dD/dh>
    for (int qd = lane; qd < n * DIM; qd += 64) {
      const int i = qd / DIM, k = qd - i * DIM;
      const float F = (pos[i * 4 + k] - pos0[i * 4 + k]) - mean[k];
      const float xc = p.x[w * n * DIM + qd];
      const float ct = q.cot ? q.cot[w * n * DIM + qd] : xc;
      if (p.out) p.out[w * n * DIM + qd] = fmaf(c_s, xc, c_out * F);
      hpart = fmaf(ct, fmaf(dc_s, xc, dc_out * F), hpart);
      const float u = c_out * (ct - cmean[k]);
      pb[i * 4 + k] = u;
      p0b[i * 4 + k] = -u;
    }
    wfence();
    // ------------------------------------------------------------------ backward sweep
    for (int l = p.L - 1; l >= 0; --l) {
      const float* wl = p.w + WIDE_HEAD + (size_t)l * WideLayer::SIZE;
      const float* ck = ws + (size_t)l * ck_layer;
      const bool last = (l == p.L - 1);
      const bool need_h = (l > 0) || want_h;  // h^0 does not depend on x (but on h, through the time feature)
      for (int qd = lane; qd < n * 4; qd += 64) { pos[qd] = ck[2 * n * WIDE_HP + qd]; pbn[qd] = 0.f; }
      for (int i = 0; i < n; ++i) hf[i * WIDE_HP + lane] = ck[i * WIDE_HP + lane];
      wfence();
      for (int i = 0; i < n; ++i) {
        const float hv = hf[i * WIDE_HP + lane];
        At[i * WIDE_HP + lane] = dense_mem_b<HK>(wl + WideLayer::WA, bc, lane, hv, wl[WideLayer::B1 + lane]);
        Bt[i * WIDE_HP + lane] = dense_mem_b<HK>(wl + WideLayer::WB, bc, lane, hv, 0.f);
        TB[i * WIDE_HP + lane] = 0.f;
      }
      wfence();
      float w2[WIDE_HP], wc1[WIDE_HP], w2t[WIDE_HP], wc1t[WIDE_HP];
#pragma unroll
      for (int k = 0; k < HK; ++k) {
        w2[k] = wl[WideLayer::W2 + k * WIDE_HP + lane];
        wc1[k] = wl[WideLayer::WC1 + k * WIDE_HP + lane];
        w2t[k] = wl[WideLayer::W2N + k * WIDE_HP + lane];
        wc1t[k] = wl[WideLayer::WC1N + k * WIDE_HP + lane];
      }
      const float wr = wl[WideLayer::WR + lane], we = wl[WideLayer::WE + lane], b2 = wl[WideLayer::B2 + lane];
      const float watt = wl[WideLayer::WATT + lane], batt = wl[WideLayer::BATT], bc1 = wl[WideLayer::BC1 + lane];
      const float wc2 = wl[WideLayer::WC2 + lane];
      for (int i = 0; i < n; ++i) {
        const float Ai = At[i * WIDE_HP + lane];
        // node model backward: h' = h + Wn2 silu(zn) + bn2, zn = Wn1a h + Wn1b agg + bn1
        float hbi = hb[i * WIDE_HP + lane], aggb = 0.f;
        if (!last) {
          float gz;
          (void)wsilu_d(ck[(n + i) * WIDE_HP + lane], gz);
          const float znb = gz * dense_mem_b<HK>(wl + WideLayer::WN2N, bc, lane, hbi, 0.f);
          aggb = dense_mem_b<HK>(wl + WideLayer::WN1BN, bc, lane, znb, 0.f);
          if (need_h) hbi = dense_mem_b<HK>(wl + WideLayer::WN1AN, bc, lane, znb, hbi);
        }
        float pi[3] = {0.f, 0.f, 0.f}, p0i[3] = {0.f, 0.f, 0.f}, X[3] = {0.f, 0.f, 0.f};
        float pacc[3] = {0.f, 0.f, 0.f}, p0acc[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < DIM; ++k) { pi[k] = pos[i * 4 + k]; p0i[k] = pos0[i * 4 + k]; X[k] = pb[i * 4 + k]; }
        float S = 0.f;
        for (int j = 0; j < n; ++j) {
          if (j == i) continue;
          float df[3] = {0.f, 0.f, 0.f}, e0[3] = {0.f, 0.f, 0.f}, radial = 0.f, ea = 0.f, t = 0.f;
          for (int k = 0; k < DIM; ++k) {
            df[k] = pi[k] - pos[j * 4 + k];
            radial = fmaf(df[k], df[k], radial);
            e0[k] = p0i[k] - pos0[j * 4 + k];
            ea = fmaf(e0[k], e0[k], ea);
            t = fmaf(df[k], X[k], t);
          }
          // the edge again, with derivative factors
          float g1, g2, gc;
          const float m1 = wsilu_d(fmaf(we, ea, fmaf(wr, radial, Ai + Bt[j * WIDE_HP + lane])), g1);
          const float m2 = wsilu_d(dense_reg_b<HK>(w2, bc, lane, m1, b2), g2);
          float a = 1.0f, m = m2;
          if (p.attention) {
            a = fast_sigmoid(wwave_sum(watt * m2) + batt);
            m = m2 * a;
          }
          const float c1 = wsilu_d(dense_reg_b<HK>(wc1, bc, lane, m, bc1), gc);
          float cs = wwave_sum(wc2 * c1), dcs = 1.0f;
          if (p.tanh_on) {
            const float th = accurate_tanh(cs);
            dcs = p.coord_scale * fmaf(-th, th, 1.0f);
            cs = th * p.coord_scale;
          }
          const float sq = sqrtf(radial + 1e-8f), inv = 1.0f / (sq + 1.0f);
          // backward: pos'_i += df inv cs
          const float csb = (inv * t) * dcs, invb = cs * t;
          float mb = dense_reg_b<HK>(wc1t, bc, lane, gc * (wc2 * csb), aggb);  // adjoint of the gated message
          if (p.attention) {
            const float sb = (a * (1.0f - a)) * wwave_sum(mb * m2);
            mb = fmaf(mb, a, watt * sb);
          }
          const float z1b = g1 * dense_reg_b<HK>(w2t, bc, lane, g2 * mb, 0.f);
          if (need_h) {
            S += z1b;
            TB[j * WIDE_HP + lane] += z1b;
          }
          const float radb = fmaf(invb, -(inv * inv) * (0.5f / sq), wwave_sum(wr * z1b));
          const float eab = wwave_sum(we * z1b);
          const float ic = inv * cs;
          for (int k = 0; k < DIM; ++k) {
            const float dfb = fmaf(ic, X[k], 2.0f * radb * df[k]);
            const float e0b = 2.0f * eab * e0[k];
            pacc[k] += dfb;
            p0acc[k] += e0b;
            if (lane == 0) {
              pbn[j * 4 + k] -= dfb;
              p0b[j * 4 + k] -= e0b;
            }
          }
        }
        hb[i * WIDE_HP + lane] = need_h ? dense_mem_b<HK>(wl + WideLayer::WAN, bc, lane, S, hbi) : hbi;
        if (lane == 0)
          for (int k = 0; k < DIM; ++k) {
            pbn[i * 4 + k] += X[k] + pacc[k];
            p0b[i * 4 + k] += p0acc[k];
          }
      }
      wfence();
      if (need_h)
        for (int j = 0; j < n; ++j)
          hb[j * WIDE_HP + lane] = dense_mem_b<HK>(wl + WideLayer::WBN, bc, lane, TB[j * WIDE_HP + lane], hb[j * WIDE_HP + lane]);
      for (int qd = lane; qd < n * 4; qd += 64) pb[qd] = pbn[qd];
      wfence();
    }
    // pos^0 = pos0 = c_in x
    for (int qd = lane; qd < n * DIM; qd += 64) {
      const int i = qd / DIM, k = qd - i * DIM;
      const float yb = pb[i * 4 + k] + p0b[i * 4 + k];
      const float xc = p.x[w * n * DIM + qd];
      const float ct = q.cot ? q.cot[w * n * DIM + qd] : xc;
      q.vjp[w * n * DIM + qd] = fmaf(c_s, ct, c_in * yb);
      hpart = fmaf(dc_in * yb, xc, hpart);
    }
    if (want_h) {
      float tb = 0.f;  // through the time feature ln(h)/8 of every node's embedding
      for (int i = 0; i < n; ++i) tb = fmaf(hb[i * WIDE_HP + lane], et, tb);
      const float s = wwave_sum(fmaf(tb, 0.125f / hval, hpart));
      if (lane == 0) q.dot_h[w] = s;
    }
    wfence();
  }
}

}  // namespace
}  // namespace pita

using namespace pita;

extern "C" int64_t pita_egnn_wide_num_weights(const pita_egnn_wide_config* c) {
  if (!c) return PITA_EINVAL;
  const int64_t H = c->hidden_nf, nf = c->n_static + 1 + (c->condition_beta ? 1 : 0);
  int64_t per_layer = (H * (2 * H + 2) + H) + (H * H + H) + (H * 2 * H + H) + (H * H + H) + (H * H + H) + H;
  if (c->attention) per_layer += H + 1;
  return (H * nf + H) + (nf * H + nf) + c->n_layers * per_layer;
}

extern "C" int pita_egnn_wide_create(pita_egnn_wide_t** out, const pita_egnn_wide_config* cfg, const float* w,
                                     int64_t n_weights, const float* h_initial) {
  PITA_REQUIRE(out && cfg && w, "pita_egnn_wide_create: null argument");
  PITA_REQUIRE(cfg->hidden_nf >= 1 && cfg->hidden_nf <= WIDE_HP, "pita_egnn_wide_create: hidden_nf must be in [1, 64]");
  PITA_REQUIRE(cfg->n_particles >= 2 && cfg->n_particles <= 64 && cfg->n_dim >= 1 && cfg->n_dim <= 3,
               "pita_egnn_wide_create: n_particles in [2, 64], n_dim in [1, 3]");
  PITA_REQUIRE(cfg->n_layers >= 1 && cfg->n_layers <= 16 && cfg->n_static >= 0, "pita_egnn_wide_create: bad layer / feature count");
  PITA_REQUIRE(cfg->n_static == 0 || h_initial, "pita_egnn_wide_create: h_initial missing");
  PITA_REQUIRE(n_weights == pita_egnn_wide_num_weights(cfg), "pita_egnn_wide_create: got %lld weights, expected %lld",
               (long long)n_weights, (long long)pita_egnn_wide_num_weights(cfg));
  const int H = cfg->hidden_nf, L = cfg->n_layers, ns = cfg->n_static, n = cfg->n_particles;
  const int nf = ns + 1 + (cfg->condition_beta ? 1 : 0);
  const size_t n_w = WIDE_HEAD + (size_t)L * WideLayer::SIZE;
  float* hw = new float[n_w]();
  float* he = new float[(size_t)n * WIDE_HP]();
  const float* q = w;
  const float* emb_w = q; q += H * nf;
  const float* emb_b = q; q += H;
  q += nf * H + nf;  // embedding_out: dead (h_final is discarded, egnn_dynamics_ad2_cat.py:187)
  for (int f = 0; f < H; ++f) {
    hw[f] = emb_w[f * nf + ns];
    hw[64 + f] = cfg->condition_beta ? emb_w[f * nf + ns + 1] : 0.f;
    for (int i = 0; i < n; ++i) {
      double s = emb_b[f];
      for (int k = 0; k < ns; ++k) s += (double)emb_w[f * nf + k] * (double)h_initial[i * ns + k];
      he[i * WIDE_HP + f] = (float)s;
    }
  }
  auto put_t = [&](float* dst, const float* M, int ld, int col0) {  // dst[k][f] = M[f][col0 + k]
    for (int k = 0; k < H; ++k)
      for (int f = 0; f < H; ++f) dst[k * WIDE_HP + f] = M[f * ld + col0 + k];
  };
  auto put_n = [&](float* dst, const float* M, int ld, int col0) {  // dst[f][k] = M[f][col0 + k]
    for (int f = 0; f < H; ++f)
      for (int k = 0; k < H; ++k) dst[f * WIDE_HP + k] = M[f * ld + col0 + k];
  };
  for (int l = 0; l < L; ++l) {
    float* wl = hw + WIDE_HEAD + (size_t)l * WideLayer::SIZE;
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
    put_t(wl + WideLayer::WA, e0w, 2 * H + 2, 0);
    put_t(wl + WideLayer::WB, e0w, 2 * H + 2, H);
    put_t(wl + WideLayer::W2, e2w, H, 0);
    put_t(wl + WideLayer::WC1, c0w, H, 0);
    put_t(wl + WideLayer::WN1A, n0w, 2 * H, 0);
    put_t(wl + WideLayer::WN1B, n0w, 2 * H, H);
    put_t(wl + WideLayer::WN2, n2w, H, 0);
    put_n(wl + WideLayer::WAN, e0w, 2 * H + 2, 0);
    put_n(wl + WideLayer::WBN, e0w, 2 * H + 2, H);
    put_n(wl + WideLayer::W2N, e2w, H, 0);
    put_n(wl + WideLayer::WC1N, c0w, H, 0);
    put_n(wl + WideLayer::WN1AN, n0w, 2 * H, 0);
    put_n(wl + WideLayer::WN1BN, n0w, 2 * H, H);
    put_n(wl + WideLayer::WN2N, n2w, H, 0);
    for (int f = 0; f < H; ++f) {
      wl[WideLayer::WR + f] = e0w[f * (2 * H + 2) + 2 * H];
      wl[WideLayer::WE + f] = e0w[f * (2 * H + 2) + 2 * H + 1];
      wl[WideLayer::B1 + f] = e0b[f];
      wl[WideLayer::B2 + f] = e2b[f];
      wl[WideLayer::WATT + f] = aw ? aw[f] : 0.f;
      wl[WideLayer::BC1 + f] = c0b[f];
      wl[WideLayer::WC2 + f] = c2w[f];
      wl[WideLayer::BN1 + f] = n0b[f];
      wl[WideLayer::BN2 + f] = n2b[f];
    }
    wl[WideLayer::BATT] = ab ? ab[0] : 0.f;
  }
  pita_egnn_wide* net = new pita_egnn_wide();
  net->cfg = *cfg;
  hipError_t e = hipMalloc(&net->d_w, n_w * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&net->d_estatic, (size_t)n * WIDE_HP * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(net->d_w, hw, n_w * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(net->d_estatic, he, (size_t)n * WIDE_HP * sizeof(float), hipMemcpyHostToDevice);
  delete[] hw;
  if (e != hipSuccess) {
    delete[] he;
    (void)hipFree(net->d_w);
    (void)hipFree(net->d_estatic);
    delete net;
    return fail(PITA_EHIP, "pita_egnn_wide_create: device upload failed: %s", hipGetErrorString(e));
  }
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess) {
    net->device = dev;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) net->n_cu = prop.multiProcessorCount;
  }
  const int rc64 = wide64_prepare(net, w, he);
  delete[] he;
  if (rc64 != PITA_OK) {
    (void)hipFree(net->d_w);
    (void)hipFree(net->d_estatic);
    delete net;
    return rc64;
  }
  *out = net;
  return PITA_OK;
}

extern "C" int pita_egnn_wide_uses_matrix_pipe(const pita_egnn_wide_t* net) {
  // PITA_WIDE_NO_MFMA (read at every call): A/B against the vector-pipe kernel
  return (net && net->shape64 && getenv("PITA_WIDE_NO_MFMA") == nullptr) ? 1 : 0;
}

extern "C" int pita_egnn_wide_destroy(pita_egnn_wide_t* net) {
  if (!net) return PITA_OK;
  wide64_release(net);
  (void)hipFree(net->d_vjp_ws);
  (void)hipFree(net->d_w);
  (void)hipFree(net->d_estatic);
  delete net;
  return PITA_OK;
}

// what: 0 backbone forward (t = its time input), 1 denoiser D_theta, 2 score (t = h = sigma^2)
extern "C" int pita_egnn_wide_eval(pita_egnn_wide_t* net, int what, const float* t, const float* x, const float* beta,
                                   float* out, int64_t B, void* stream) {
  PITA_REQUIRE(net && B >= 0 && what >= 0 && what <= 2, "pita_egnn_wide_eval: bad argument");
  if (B == 0) return PITA_OK;
  PITA_REQUIRE(t && x && out, "pita_egnn_wide_eval: null argument");
  PITA_REQUIRE(beta || !net->cfg.condition_beta, "pita_egnn_wide_eval: beta required (condition_beta)");
  int prev = -1;
  bool switched = false;
  if (net->device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != net->device)
    switched = hipSetDevice(net->device) == hipSuccess;
  WideParams p{};
  p.w = net->d_w; p.estatic = net->d_estatic;
  p.n = net->cfg.n_particles; p.dim = net->cfg.n_dim; p.H = net->cfg.hidden_nf; p.L = net->cfg.n_layers;
  p.attention = net->cfg.attention; p.tanh_on = net->cfg.tanh; p.has_beta = net->cfg.condition_beta;
  p.coord_scale = net->cfg.coords_range / (float)net->cfg.n_layers;
  p.B = B; p.mode = what; p.x = x; p.t = t; p.beta = beta; p.out = out;
  int rc = PITA_OK;
  // matrix-pipe kernel first where the particle system has one; the vector-pipe kernel then recomputes the walkers whose
  // result came out non-finite (an activation beyond the f16 range) and returns at once for all others
  if (pita_egnn_wide_uses_matrix_pipe(net)) {
    rc = wide64_launch(net, what, t, x, beta, out, B, (hipStream_t)stream);
    p.only_bad = 1;
    p.bad_flag = net->d_flag;
  }
  const size_t per_wave = sizeof(float) * (size_t)(3 * p.n * WIDE_HP + 3 * p.n * 4 + WIDE_HP);
  int waves = 4;
  while (waves > 1 && per_wave * waves > 72 * 1024) waves >>= 1;  // two blocks per CU inside the 160 KB
  auto kernel = p.H <= 32 ? egnn_wide_kernel<32, false> : egnn_wide_kernel<64, false>;
  if (rc != PITA_OK) {
  } else if (per_wave * waves > 150 * 1024) {
    rc = fail(PITA_EUNSUPPORTED, "pita_egnn_wide_eval: %d particles need %zu B of LDS per wave", p.n, per_wave);
  } else if (ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), per_wave * waves) != hipSuccess) {
    rc = fail(PITA_EHIP, "pita_egnn_wide_eval: cannot reserve %zu B of LDS", per_wave * waves);
  } else {
    const long long want = (B + waves - 1) / waves, cap = (long long)net->n_cu * 8;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(waves * 64), per_wave * waves, (hipStream_t)stream, p);
    if (hipGetLastError() != hipSuccess) rc = fail(PITA_EHIP, "pita_egnn_wide_eval: launch failed");
  }
  if (switched) (void)hipSetDevice(prev);
  return rc;
}

// Forward-mode derivative of the denoiser around the wide backbone (vector-pipe kernel; see egnn_wide_jvp_kernel)
extern "C" int pita_egnn_wide_jvp(pita_egnn_wide_t* net, const float* h, const float* x, const float* beta, const float* vx,
                                  int dir, const float* vh, float* out, float* dout, float* dot_out, int64_t dot_stride,
                                  int64_t dot_off, float* diag_acc, int64_t B, void* stream) {
  PITA_REQUIRE(net && B >= 0, "pita_egnn_wide_jvp: bad argument");
  if (B == 0) return PITA_OK;
  PITA_REQUIRE(h && x, "pita_egnn_wide_jvp: null argument");
  PITA_REQUIRE(beta || !net->cfg.condition_beta, "pita_egnn_wide_jvp: beta required (condition_beta)");
  PITA_REQUIRE(dir < net->cfg.n_particles * net->cfg.n_dim, "pita_egnn_wide_jvp: direction %d out of range", dir);
  PITA_REQUIRE(!diag_acc || (dir >= 0 && !vx), "pita_egnn_wide_jvp: diag_acc needs a unit direction");
  int prev = -1;
  bool switched = false;
  if (net->device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != net->device)
    switched = hipSetDevice(net->device) == hipSuccess;
  WideJvpParams q{};
  WideParams& p = q.base;
  p.w = net->d_w; p.estatic = net->d_estatic;
  p.n = net->cfg.n_particles; p.dim = net->cfg.n_dim; p.H = net->cfg.hidden_nf; p.L = net->cfg.n_layers;
  p.attention = net->cfg.attention; p.tanh_on = net->cfg.tanh; p.has_beta = net->cfg.condition_beta;
  p.coord_scale = net->cfg.coords_range / (float)net->cfg.n_layers;
  p.B = B; p.mode = 1; p.x = x; p.t = h; p.beta = beta; p.out = out;
  q.vx = vx; q.vh = vh; q.dir = vx ? -1 : dir; q.dout = dout; q.dot_out = dot_out; q.dot_stride = dot_stride;
  q.dot_off = dot_off; q.diag_acc = diag_acc;
  int rc = PITA_OK;
  // matrix-pipe kernel first where the particle system has one (PITA_WIDE_NO_MFMA: the vector-pipe kernel alone); it
  // flags the walkers whose primal or tangent left the f16 range and the vector-pipe kernel below computes exactly those
  if (pita_egnn_wide_uses_matrix_pipe(net)) {
    const size_t need = sizeof(int) * (size_t)B;
    if (need > net->jbad_bytes) {
      hipError_t e = hipStreamSynchronize((hipStream_t)stream);
      (void)hipFree(net->d_jbad);
      net->d_jbad = nullptr;
      net->jbad_bytes = 0;
      if (e == hipSuccess) e = hipMalloc(&net->d_jbad, need);
      if (e != hipSuccess) rc = fail(PITA_EHIP, "pita_egnn_wide_jvp: flag buffer: %s", hipGetErrorString(e));
      else net->jbad_bytes = need;
    }
    if (rc == PITA_OK && hipMemsetAsync(net->d_jbad, 0, need, (hipStream_t)stream) != hipSuccess)
      rc = fail(PITA_EHIP, "pita_egnn_wide_jvp: memset failed");
    if (rc == PITA_OK) {
      const int r64 = wide64_jvp(net, h, x, beta, vx, dir, vh, out, dout, dot_out, dot_stride, dot_off, diag_acc, net->d_jbad,
                                 B, (hipStream_t)stream);
      if (r64 == PITA_OK) q.only_bad = net->d_jbad;
      else if (r64 != 1) rc = r64;
    }
  }
  const size_t per_wave = sizeof(float) * (size_t)(5 * p.n * WIDE_HP + 6 * p.n * 4 + 2 * WIDE_HP);
  int waves = 4;
  while (waves > 1 && per_wave * waves > 150 * 1024) waves >>= 1;
  auto kernel = p.H <= 32 ? egnn_wide_jvp_kernel<32> : egnn_wide_jvp_kernel<64>;
  
  if (rc != PITA_OK) {
  } else if (per_wave * waves > 150 * 1024) {
    rc = fail(PITA_EUNSUPPORTED, "pita_egnn_wide_jvp: %d particles need %zu B of LDS per wave", p.n, per_wave);
  } else if (ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), per_wave * waves) != hipSuccess) {
    rc = fail(PITA_EHIP, "pita_egnn_wide_jvp: cannot reserve %zu B of LDS", per_wave * waves);
  } else {
    const long long want = (B + waves - 1) / waves, cap = (long long)net->n_cu * 2;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(waves * 64), per_wave * waves, (hipStream_t)stream, q);
    if (hipGetLastError() != hipSuccess) rc = fail(PITA_EHIP, "pita_egnn_wide_jvp: launch failed");
  }
  if (switched) (void)hipSetDevice(prev);
  return rc;
}

// Fused sampler on the wide backbone: n_steps Euler-Maruyama steps of the NOT-debiased reverse VE-SDE in one launch
// (arguments and step table as pita_egnn_sampler_run).  Matrix-pipe kernel where the particle system has one, followed by
// the vector-pipe kernel on exactly the walkers that came out non-finite (restarted from a backup of the walkers, all
// steps in fp32; the per-step moments are partitioned between the two launches per particle); the vector-pipe kernel
// alone for every other shape.
extern "C" int pita_egnn_wide_sampler_run(pita_egnn_wide_t* net, float* x, int64_t B, const float* step_tab, int n_steps,
                                          const float* noise, uint64_t seed, uint64_t walker_offset, int64_t step0,
                                          int remove_mean, double* stats_out, void* stream) {
  PITA_REQUIRE(net && B >= 0 && n_steps >= 0, "pita_egnn_wide_sampler_run: bad argument");
  if (B == 0 || n_steps == 0) return PITA_OK;
  PITA_REQUIRE(x && step_tab, "pita_egnn_wide_sampler_run: null argument");
  int prev = -1;
  bool switched = false;
  if (net->device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != net->device)
    switched = hipSetDevice(net->device) == hipSuccess;
  hipStream_t st = (hipStream_t)stream;
  WideParams p{};
  p.w = net->d_w; p.estatic = net->d_estatic;
  p.n = net->cfg.n_particles; p.dim = net->cfg.n_dim; p.H = net->cfg.hidden_nf; p.L = net->cfg.n_layers;
  p.attention = net->cfg.attention; p.tanh_on = net->cfg.tanh; p.has_beta = net->cfg.condition_beta;
  p.coord_scale = net->cfg.coords_range / (float)net->cfg.n_layers;
  p.B = B; p.mode = 3; p.xs = x; p.step_tab = step_tab; p.n_steps = n_steps; p.noise = noise; p.seed = seed;
  p.walker_offset = walker_offset; p.step0 = step0; p.remove_mean = remove_mean; p.stats_out = stats_out;
  int rc = PITA_OK;
  const size_t nx = (size_t)B * p.n * p.dim;
  if (pita_egnn_wide_uses_matrix_pipe(net)) {
    const size_t need = sizeof(float) * nx + sizeof(int) * (size_t)B * p.n;
    if (need > net->bk_bytes) {
      hipError_t e = hipStreamSynchronize(st);  // an earlier launch may still use the old buffer
      (void)hipFree(net->d_bk);
      net->d_bk = nullptr;
      net->bk_bytes = 0;
      if (e == hipSuccess) e = hipMalloc(&net->d_bk, need);
      if (e != hipSuccess) rc = fail(PITA_EHIP, "pita_egnn_wide_sampler_run: backup buffer: %s", hipGetErrorString(e));
      else net->bk_bytes = need;
    }
    if (rc == PITA_OK) {
      float* xb = static_cast<float*>(net->d_bk);
      int* bad_from = reinterpret_cast<int*>(xb + nx);
      if (hipMemcpyAsync(xb, x, sizeof(float) * nx, hipMemcpyDeviceToDevice, st) != hipSuccess)
        rc = fail(PITA_EHIP, "pita_egnn_wide_sampler_run: backup copy failed");
      if (rc == PITA_OK)
        rc = wide64_sampler(net, x, B, step_tab, n_steps, noise, seed, walker_offset, step0, remove_mean, stats_out,
                            bad_from, st);
      p.only_bad = 1;
      p.x_backup = xb;
      p.bad_from = bad_from;
      p.bad_flag = net->d_flag;
    }
  }
  const size_t per_wave = sizeof(float) * (size_t)(3 * p.n * WIDE_HP + 4 * p.n * 4 + WIDE_HP);
  int waves = 4;
  while (waves > 1 && per_wave * waves > 72 * 1024) waves >>= 1;
  auto kernel = p.H <= 32 ? egnn_wide_kernel<32, true> : egnn_wide_kernel<64, true>;
  if (rc != PITA_OK) {
  } else if (per_wave * waves > 150 * 1024) {
    rc = fail(PITA_EUNSUPPORTED, "pita_egnn_wide_sampler_run: %d particles need %zu B of LDS per wave", p.n, per_wave);
  } else if (ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), per_wave * waves) != hipSuccess) {
    rc = fail(PITA_EHIP, "pita_egnn_wide_sampler_run: cannot reserve %zu B of LDS", per_wave * waves);
  } else {
    const long long want = (B + waves - 1) / waves, cap = (long long)net->n_cu * 8;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(waves * 64), per_wave * waves, st, p);
    if (hipGetLastError() != hipSuccess) rc = fail(PITA_EHIP, "pita_egnn_wide_sampler_run: launch failed");
  }
  if (switched) (void)hipSetDevice(prev);
  return rc;
}

// Reverse-mode derivative of the denoiser around the wide backbone (see egnn_wide_vjp_kernel); arguments as pita_egnn_vjp
extern "C" int pita_egnn_wide_vjp(pita_egnn_wide_t* net, const float* h, const float* x, const float* beta, const float* cot,
                                  float* out, float* vjp, float* dot_h, int64_t B, void* stream) {
  PITA_REQUIRE(net && B >= 0, "pita_egnn_wide_vjp: bad argument");
  if (B == 0) return PITA_OK;
  PITA_REQUIRE(h && x && vjp, "pita_egnn_wide_vjp: null argument");
  PITA_REQUIRE(beta || !net->cfg.condition_beta, "pita_egnn_wide_vjp: beta required (condition_beta)");
  int prev = -1;
  bool switched = false;
  if (net->device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != net->device)
    switched = hipSetDevice(net->device) == hipSuccess;
  WideVjpParams q{};
  WideParams& p = q.base;
  p.w = net->d_w; p.estatic = net->d_estatic;
  p.n = net->cfg.n_particles; p.dim = net->cfg.n_dim; p.H = net->cfg.hidden_nf; p.L = net->cfg.n_layers;
  p.attention = net->cfg.attention; p.tanh_on = net->cfg.tanh; p.has_beta = net->cfg.condition_beta;
  p.coord_scale = net->cfg.coords_range / (float)net->cfg.n_layers;
  p.B = B; p.mode = 1; p.x = x; p.t = h; p.beta = beta; p.out = out;
  q.cot = cot; q.vjp = vjp; q.dot_h = dot_h;
  int rc = PITA_OK;
  const size_t per_wave = sizeof(float) * (size_t)(5 * p.n * WIDE_HP + 6 * p.n * 4 + WIDE_HP);
  int waves = 4;
  while (waves > 1 && per_wave * waves > 150 * 1024) waves >>= 1;
  auto kernel = p.H <= 32 ? egnn_wide_vjp_kernel<32> : egnn_wide_vjp_kernel<64>;
  const long long want = (B + waves - 1) / waves, cap = (long long)net->n_cu;  // one wave per SIMD
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  const size_t ws_need = sizeof(float) * (size_t)grid * waves * p.L * ((size_t)2 * p.n * WIDE_HP + (size_t)p.n * 4);
  if (per_wave * waves > 150 * 1024) {
    rc = fail(PITA_EUNSUPPORTED, "pita_egnn_wide_vjp: %d particles need %zu B of LDS per wave", p.n, per_wave);
  } else if (ws_need > net->vjp_ws_bytes) {
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);  // an earlier launch may still use the old buffer
    (void)hipFree(net->d_vjp_ws);
    net->d_vjp_ws = nullptr;
    net->vjp_ws_bytes = 0;
    if (e == hipSuccess) e = hipMalloc(&net->d_vjp_ws, ws_need);
    if (e != hipSuccess) rc = fail(PITA_EHIP, "pita_egnn_wide_vjp: checkpoint buffer: %s", hipGetErrorString(e));
    else net->vjp_ws_bytes = ws_need;
  }
  if (rc == PITA_OK && ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), per_wave * waves) != hipSuccess)
    rc = fail(PITA_EHIP, "pita_egnn_wide_vjp: cannot reserve %zu B of LDS", per_wave * waves);
  if (rc == PITA_OK) {
    q.ws = net->d_vjp_ws;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(waves * 64), per_wave * waves, (hipStream_t)stream, q);
    if (hipGetLastError() != hipSuccess) rc = fail(PITA_EHIP, "pita_egnn_wide_vjp: launch failed");
  }
  if (switched) (void)hipSetDevice(prev);
  return rc;
}
