// Reverse-mode derivative (VJP) of the EDM-preconditioned EGNN denoiser for gfx950.
//
// The debiased (Feynman-Kac) drift of the reference sampler (pita/src/models/components/sdes.py:151-239) needs
// grad_x E_theta (energy_net.py:51-62, torch.autograd.grad).  With E = (1+c_s)|x|^2/(2h) - <D(h,x), x>/h
// (energy_net.py:33-49) that is  ((1+c_s) x - D - J_x D^T x) / h : ONE vector-Jacobian product of the denoiser
// D(h, x) = c_s x + c_out F(c_noise, c_in x, beta) (score_net.py:21-33) instead of dim forward-mode launches.
//
// This kernel returns D and  vjp = J_x D(h,x)^T cot  for a per-walker cotangent `cot` (cot = x for grad E).
//
// Mapping: as egnn_kernel.hip / egnn_jvp_kernel.hip (wave = up to G walkers = dense 32-column tiles, lane =
// column x 16 features, exact 3-way bf16 split on the matrix pipe).  Structure per group of walkers:
//   forward sweep   layers 0..L-1, checkpointing each layer's input (h^l, pos^l) and node pre-activation zn^l in a
//                   per-wave global scratch (83 KB per wave for LJ13, L2/MALL resident);
//   backward sweep  layers L-1..0: every edge's MLP is recomputed from the checkpoint (activations AND their
//                   derivatives), then back-propagated with the transposed weight fragments:
//                     m_bar  = agg_bar_i + Wc1^T (c1_bar . silu'(zc))            coordinate head + node aggregate
//                     m2_bar = att m_bar + w_att <m_bar, m2> att (1 - att)       attention gate
//                     z1_bar = (W2^T (m2_bar . silu'(z2))) . silu'(z1)
//                   z1_bar feeds h_i (through Wa), h_j (through Wb), |d|^2 and the frozen edge attribute; the geometry
//                   d / (|d| + 1) feeds pos_i and pos_j.  Sums over a node's own edges stay in registers (the edge
//                   enumeration j = i + dd keeps i fixed per lane); contributions to the PARTNER j go through LDS
//                   tables -- for a fixed dd the map i -> j is a permutation, so the read-add-write is conflict free --
//                   and Wa^T / Wb^T are applied once per node to the summed z1_bar, not per edge.
// One wave per SIMD (512 VGPRs), like the JVP kernel.
#include "egnn_common.h"

namespace pita {

struct VjpParams {
  const unsigned* mats16;
  const unsigned* mats16h;  // f16 two-piece fragments (PE = 2: the per-edge recompute GEMMs W2, Wc1)
  const float* vecs;
  int n_layers, in_nf, attention, tanh_on, feature_layout;
  float coord_scale;
  long long B;
  const float* h;     // [B] sigma^2
  const float* x;     // [B, D]
  const float* beta;  // [B] or null
  const float* cot;   // [B, D] cotangent (null: cot = x)
  float* out;         // [B, D] denoiser D (nullable)
  float* vjp;         // [B, D] J_x D^T cot
  float* dot_h;       // nullable [B]: <cot, dD/dh> -- the reverse sweep also reaches the inputs that depend on h (time
                      // feature, c_in scaling) and the explicit c_s(h), c_out(h): no forward-mode launch needed
  float* dot_parts;   // nullable [B, 2] (needs dot_h): { c_out <cot, F>,  <cot, d(c_out F)/dh> = dot_h - c_s'(h) <cot, x> }.
                      // E_theta and dE_theta/dh assembled from these carry no 1/h^2-sized cancellation (fk_kernels.hip)
  float* ws;          // checkpoint scratch: total_waves * ws_f floats
  const int* mark;    // repair pass (nullable): only the walkers marked by vjp_mark_kernel are recomputed and written
  const int* flag;    // repair pass: 0 = nothing was marked, return at once
};

template <int N, int DIM, int G, int WAVES>
struct VjpCfg {
  static constexpr int NCOL = G * N;
  static constexpr int NT = (NCOL + 31) / 32;
  static constexpr int NCOLP = NT * 32;
  static constexpr int PB_F = NCOLP * PBS;
  static constexpr int POS_F = NCOLP * DIM;
  static constexpr int WAVE_F = 2 * PB_F + 4 * POS_F;  // PB, TB (z1_bar scatter), pos, pos0, pos_bar scatter, pos0_bar scatter
  static constexpr int CK_F = (16 + 16 + 4) * 64;      // per (layer, tile): h (16/lane), zn (16/lane), pos (<= 4/lane)
  static __host__ __device__ constexpr int vec_f(int L) { return ((VEC_EMB_F + L * VEC_LAYER_F) + 3) & ~3; }
  static __host__ __device__ constexpr size_t lds_bytes(int L) {
    return sizeof(float) * (size_t)(vec_f(L) + WAVES * WAVE_F);
  }
  static __host__ __device__ constexpr size_t ws_f(int L) { return (size_t)L * NT * CK_F; }
};

__device__ __forceinline__ void ck_store16(float* base, int lane, const f32x16& v) {
  f32x4* p = reinterpret_cast<f32x4*>(base) + lane;
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q * 64] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}
__device__ __forceinline__ f32x16 ck_load16(const float* base, int lane) {
  const f32x4* p = reinterpret_cast<const f32x4*>(base) + lane;
  f32x16 r;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = p[q * 64];
    r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
  }
  return r;
}

// silu on pre-scaled pre-activations v = kS z (egnn_common.h): y' = kS silu(z) and the TRUE derivative silu'(z)
__device__ __forceinline__ void silu_grad(float v, float& y, float& g) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v));
  y = v * s;
  g = s * fmaf(v * (1.0f / SILU_PRESCALE), 1.0f - s, 1.0f);
}

// the same for a 16-element vector, staged (exponentials, reciprocals, products): same values, no transcendental with
// its consumer directly behind it
__device__ __forceinline__ void silu_grad16(const f32x16& v, f32x16& y, f32x16& g) {
  // element pairs as explicit two-vectors: the adds, multiplies and the fma become packed instructions (a lone wave
  // issues one instruction of any kind per ~5 cycles, so halving their count is what counts); same operations per
  // element, same values
  f32x2 s[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    s[q].x = __builtin_amdgcn_exp2f(v[2 * q]);
    s[q].y = __builtin_amdgcn_exp2f(v[2 * q + 1]);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) s[q] = s[q] + 1.0f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    s[q].x = __builtin_amdgcn_rcpf(s[q].x);
    s[q].y = __builtin_amdgcn_rcpf(s[q].y);
  }
  const f32x2 one = {1.0f, 1.0f};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x2 vv = {v[2 * q], v[2 * q + 1]};
    const f32x2 yy = vv * s[q];
    const f32x2 w = __builtin_elementwise_fma(vv * (1.0f / SILU_PRESCALE), one - s[q], one);
    const f32x2 gg = s[q] * w;
    y[2 * q] = yy.x; y[2 * q + 1] = yy.y;
    g[2 * q] = gg.x; g[2 * q + 1] = gg.y;
  }
}

__device__ __forceinline__ void lds_add16(float* dst, const f32x16& v) {
  f32x4* p = reinterpret_cast<f32x4*>(dst);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 a = p[q];
    a.x += v[4 * q]; a.y += v[4 * q + 1]; a.z += v[4 * q + 2]; a.w += v[4 * q + 3];
    p[q] = a;
  }
}

// a per-edge primal GEMM of the sweeps: bf16x3 (PE = 1), or the f16 two-piece path of the forward kernels (PE = 2: half
// the matrix instructions, a two-instruction operand split; the accumulator holds F16_SX F16_SW x the pre-activation)
template <int PE>
__device__ __forceinline__ f32x16 edge_gemm(const WFrag<PE>& w, const f32x16& in, const f32x16& bias) {
  if constexpr (PE == 2) {
    f32x16 r = w.mul(in, bias * (F16_SX * F16_SW));
    r *= F16_UNSCALE;
    return r;
  } else {
    return w.mul(in, bias);
  }
}

// FIXED: attention gate, tanh-bounded coordinate head and the h-derivative output are compile-time "on" (every reference
// configuration of the debiased regime): the edge loops carry no run-time branch.  FIXED = false keeps them run-time.
// PE: arithmetic of the four per-edge primal GEMMs (W2, Wc1 in the forward sweep and again in the recompute of the
// backward sweep); the adjoint GEMMs and the per-node layers stay bf16x3 (adjoints have no fixed range to scale for).
// A walker whose activations leave the f16 range ends non-finite, is marked by vjp_mark_kernel and recomputed by the
// PE = 1 instantiation (`mark`), like the forward and divergence kernels do it.
template <int N, int DIM, int G, int WAVES, bool FIXED, int PE>
__global__ void __launch_bounds__(WAVES * 64, 1) egnn_vjp_kernel(VjpParams p) {
  using C = VjpCfg<N, DIM, G, WAVES>;
  if (p.mark && *p.flag == 0) return;
  const bool att_on = FIXED ? true : (p.attention != 0);
  const bool tanh_on = FIXED ? true : (p.tanh_on != 0);
  const bool want_h = FIXED ? true : (p.dot_h != nullptr);
  constexpr int NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = p.n_layers;
  const int vec_f = C::vec_f(L);
  for (int i = threadIdx.x; i < VEC_EMB_F + L * VEC_LAYER_F; i += WAVES * 64) lds[i] = p.vecs[i];
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane & 31, hh = lane >> 5;
  float* PB = lds + vec_f + wave * C::WAVE_F;
  float* TB = PB + C::PB_F;
  float* posb = TB + C::PB_F;
  float* pos0 = posb + C::POS_F;
  float* pbsc = pos0 + C::POS_F;
  float* p0sc = pbsc + C::POS_F;
  const float* vemb = lds;
  float* ws = p.ws + ((size_t)blockIdx.x * WAVES + wave) * C::ws_f(L);
  const f32x16 zero16 = {0};
  const float kS = SILU_PRESCALE, kSi = 1.0f / SILU_PRESCALE;

  const long long total_waves = (long long)gridDim.x * WAVES;
  const long long quota = (p.B + total_waves - 1) / total_waves;
  const long long wbeg = ((long long)blockIdx.x * WAVES + wave) * quota;
  const long long wend = (wbeg + quota < p.B) ? wbeg + quota : p.B;
  for (long long walker0 = wbeg; walker0 < wend; walker0 += G) {
    const int nwalk = (int)((wend - walker0) < G ? (wend - walker0) : G);
    if (p.mark) {  // repair pass: groups without a marked walker are skipped (wave-uniform)
      const int m = (lane < nwalk) ? p.mark[walker0 + lane] : 0;
      if (!__any(m != 0)) continue;
    }
    const int ncol = nwalk * N;
    const int ntile = (ncol + 31) >> 5;
    int col[NT], nodei[NT];
    bool valid[NT];
    float xin[NT][DIM], cot[NT][DIM], c_s[NT], c_in[NT], c_out[NT], hvv[NT], dhacc[NT], facc[NT], cxacc[NT];
    bool a0t[NT], a1t[NT];  // which embedding inputs of this node are the time feature (quirk Q1 layout)
    float posi[NT][DIM], p0i[NT][DIM];
    f32x16 hf[NT];
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
      hvv[T] = hv;
      dhacc[T] = 0.f; facc[T] = 0.f; cxacc[T] = 0.f;
      const float tfeat = 0.125f * logf(hv);
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        const long long gi = (walker0 * N + col[T]) * DIM + k;
        xin[T][k] = valid[T] ? p.x[gi] : 0.f;
        cot[T][k] = valid[T] ? (p.cot ? p.cot[gi] : xin[T][k]) : 0.f;
        posi[T][k] = c_in[T] * xin[T][k];
        p0i[T][k] = posi[T][k];
        if (hh == 0) {
          pos0[col[T] * DIM + k] = posi[T][k];
          posb[col[T] * DIM + k] = posi[T][k];
          p0sc[col[T] * DIM + k] = 0.f;
        }
      }
      float a0, a1;
      if (p.in_nf == 1) { a0 = tfeat; a1 = 0.f; a0t[T] = true; a1t[T] = false; }
      else if (p.feature_layout == 0) {
        a0t[T] = 2 * nodei[T] < N;
        a1t[T] = 2 * nodei[T] + 1 < N;
        a0 = a0t[T] ? tfeat : bet;
        a1 = a1t[T] ? tfeat : bet;
      } else { a0 = tfeat; a1 = bet; a0t[T] = true; a1t[T] = false; }
      const f32x16 w0 = lds_vec16(vemb + hh * 16), w1 = lds_vec16(vemb + 32 + hh * 16), eb = lds_vec16(vemb + 64 + hh * 16);
#pragma unroll
      for (int r = 0; r < 16; ++r) hf[T][r] = fmaf(w0[r], a0, fmaf(w1[r], a1, eb[r]));
    }
    wave_lds_fence();

    // ------------------------------------------------------------------ forward sweep with checkpoints
    for (int l = 0; l < L; ++l) {
      const unsigned* mats16 = p.mats16 + (size_t)l * M_COUNT * MAT_W;
      const float* vl = lds + VEC_EMB_F + l * VEC_LAYER_F + hh * 16;
      const bool last = (l == L - 1);
      {
        WFrag<1> wb;
        wb.load(nullptr, mats16, M_WB, lane);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          if (T >= ntile) continue;
          float* ck = ws + ((size_t)l * NT + T) * C::CK_F;
          ck_store16(ck, lane, hf[T]);
#pragma unroll
          for (int k = 0; k < DIM; ++k) ck[32 * 64 + k * 64 + lane] = posi[T][k];
          const f32x16 pb = wb.mul(hf[T], zero16);
          f32x4* dst = reinterpret_cast<f32x4*>(PB + col[T] * PBS + hh * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[q] = f32x4{pb[4 * q], pb[4 * q + 1], pb[4 * q + 2], pb[4 * q + 3]};
        }
      }
      wave_lds_fence();
      const unsigned* matse = PE == 2 ? p.mats16h + (size_t)l * M_COUNT * MAT_WH : mats16;
      WFrag<PE> w2f, wc1f;
      w2f.load(nullptr, matse, M_W2, lane);
      wc1f.load(nullptr, matse, M_WC1, lane);
      frag_to_agpr(w2f);
      frag_to_agpr(wc1f);
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
        }
        f32x16 agg = {0};
        float xacc[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) xacc[k] = 0.f;
        const int cbase = col[T] - nodei[T];
        const int live = valid[T] ? 1 : 0;  // columns beyond the group's walkers pair with themselves
        for (int dd = 1; dd < N; ++dd) {
          asm volatile("" ::: "memory");
          int j = nodei[T] + dd * live;
          j = (j >= N) ? j - N : j;
          const int cj = cbase + j;
          float df[DIM], radial = 0.f, ea = 0.f;
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            df[k] = posi[T][k] - posb[cj * DIM + k];
            radial = fmaf(df[k], df[k], radial);
            const float e0 = p0i[T][k] - pos0[cj * DIM + k];
            ea = fmaf(e0, e0, ea);
          }
          f32x16 z = Ai + lds_vec16(PB + cj * PBS + hh * 16);
          z = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re, hh ? ea : radial, z, 0, 0, 0);
          silu16(z);
          z = edge_gemm<PE>(w2f, z, lds_vec16(vl + V_B2 * EH));
          silu16(z);
          if (att_on) {
            const float att = fast_sigmoid(xhalf_sum(dot16(lds_vec16(vl + V_WATT * EH), z)) + b_att);
            z *= att;
          }
          if (!last) agg += z;
          f32x16 c1 = edge_gemm<PE>(wc1f, z, lds_vec16(vl + V_BC1 * EH));
          silu16(c1);
          float cs = xhalf_sum(dot16(lds_vec16(vl + V_WC2 * EH), c1));
          if (tanh_on) cs = tanh_select(cs) * p.coord_scale;
          const float inv = 1.0f / (sqrtf(radial + 1e-8f) + 1.0f);
#pragma unroll
          for (int k = 0; k < DIM; ++k) xacc[k] = fmaf(df[k] * inv, cs, xacc[k]);
        }
#pragma unroll
        for (int k = 0; k < DIM; ++k) posi[T][k] += xacc[k];
        if (!last) {
          WFrag<1> wn;
          wn.load(nullptr, mats16, M_WN1A, lane);
          f32x16 z = wn.mul(hf[T], lds_vec16(vl + V_BN1 * EH));
          wn.load(nullptr, mats16, M_WN1B, lane);
          z = wn.mul(agg, z);
          ck_store16(ws + ((size_t)l * NT + T) * C::CK_F + 16 * 64, lane, z);
          silu16(z);
          wn.load(nullptr, mats16, M_WN2, lane);
          hf[T] += wn.mul(z, lds_vec16(vl + V_BN2 * EH));
        }
      }
      wave_lds_fence();  // every tile has read this layer's partner positions
#pragma unroll
      for (int T = 0; T < NT; ++T)
        if (hh == 0) {
#pragma unroll
          for (int k = 0; k < DIM; ++k) posb[col[T] * DIM + k] = posi[T][k];
        }
      wave_lds_fence();
    }

    // ------------------------------------------------------------------ D = c_s x + c_out F, F mean-free
    float* scr = TB;
    float pb[NT][DIM];  // adjoint of the positions entering the next layer; starts as d<cot, D>/d pos^L
#pragma unroll
    for (int T = 0; T < NT; ++T)
      if (hh == 0) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          scr[col[T] * DIM + k] = posi[T][k] - p0i[T][k];
          scr[C::POS_F + col[T] * DIM + k] = cot[T][k];
        }
      }
    wave_lds_fence();
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const int cb = (col[T] < ncol) ? col[T] - nodei[T] : 0;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        float s = 0.f, sc = 0.f;
        for (int q = 0; q < N; ++q) { s += scr[(cb + q) * DIM + k]; sc += scr[C::POS_F + (cb + q) * DIM + k]; }
        const float F = (posi[T][k] - p0i[T][k]) - s / (float)N;
        if (p.out && valid[T] && hh == 0 && (!p.mark || p.mark[walker0 + col[T] / N]))
          p.out[(walker0 * N + col[T]) * DIM + k] = fmaf(c_s[T], xin[T][k], c_out[T] * F);
        if (want_h && valid[T] && hh == 0) {  // explicit h-dependence of D = c_s(h) x + c_out(h) F
          const float op = 1.0f + hvv[T];
          const float dcs = -c_s[T] * c_s[T];                               // d/dh 1/(1+h)
          const float dcin = -0.5f * c_in[T] / op;                          // d/dh (1+h)^-1/2
          const float dcout = 0.5f * c_in[T] / sqrtf(hvv[T]) + sqrtf(hvv[T]) * dcin;
          // the c_s'(h) <cot, x> share is kept apart: it is ~1/h times larger than the rest at small h, and the
          // assembly of dE/dh cancels it against a closed form (fk_kernels.hip)
          dhacc[T] = fmaf(cot[T][k], dcout * F, dhacc[T]);
          facc[T] = fmaf(cot[T][k], c_out[T] * F, facc[T]);
          cxacc[T] = fmaf(cot[T][k] * dcs, xin[T][k], cxacc[T]);
        }
        pb[T][k] = valid[T] ? c_out[T] * (cot[T][k] - sc / (float)N) : 0.f;  // remove_mean is self-adjoint
      }
    }
    wave_lds_fence();

    // ------------------------------------------------------------------ backward sweep
    f32x16 hb[NT];
    float vfin[NT][DIM], p0acc[NT][DIM];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      hb[T] = zero16;
#pragma unroll
      for (int k = 0; k < DIM; ++k) { vfin[T][k] = pb[T][k]; p0acc[T][k] = 0.f; }
    }
    for (int l = L - 1; l >= 0; --l) {
      const unsigned* mats16 = p.mats16 + (size_t)l * M_COUNT * MAT_W;
      const float* vl = lds + VEC_EMB_F + l * VEC_LAYER_F + hh * 16;
      const bool last = (l == L - 1);
      {  // partner tables of layer l from the checkpoint; clear the scatter tables
        WFrag<1> wb;
        wb.load(nullptr, mats16, M_WB, lane);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          if (T >= ntile) continue;
          const float* ck = ws + ((size_t)l * NT + T) * C::CK_F;
          const f32x16 hl = ck_load16(ck, lane);
          const f32x16 pbv = wb.mul(hl, zero16);
          f32x4* dst = reinterpret_cast<f32x4*>(PB + col[T] * PBS + hh * 16);
          f32x4* tdst = reinterpret_cast<f32x4*>(TB + col[T] * PBS + hh * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            dst[q] = f32x4{pbv[4 * q], pbv[4 * q + 1], pbv[4 * q + 2], pbv[4 * q + 3]};
            tdst[q] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          if (hh == 0) {
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
              posb[col[T] * DIM + k] = ck[32 * 64 + k * 64 + lane];
              pbsc[col[T] * DIM + k] = 0.f;
            }
          }
        }
      }
      wave_lds_fence();
      const unsigned* matse = PE == 2 ? p.mats16h + (size_t)l * M_COUNT * MAT_WH : mats16;
      WFrag<PE> w2f, wc1f;
      WFrag<1> w2t, wc1t;
      w2f.load(nullptr, matse, M_W2, lane);
      wc1f.load(nullptr, matse, M_WC1, lane);
      w2t.load(nullptr, mats16, M_W2T, lane);
      wc1t.load(nullptr, mats16, M_WC1T, lane);
      frag_to_agpr(w2f);
      frag_to_agpr(wc1f);
      frag_to_agpr(w2t);
      frag_to_agpr(wc1t);
      const float a_re = lds[VEC_EMB_F + l * VEC_LAYER_F + V_WRE * EH + lane];
      const float b_att = lds[VEC_EMB_F + l * VEC_LAYER_F + V_COUNT * EH];
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        if (T >= ntile) continue;
        const float* ck = ws + ((size_t)l * NT + T) * C::CK_F;
        const f32x16 hl = ck_load16(ck, lane);
        float pl[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) pl[k] = ck[32 * 64 + k * 64 + lane];
        f32x16 Ai;
        {
          WFrag<1> wa;
          wa.load(nullptr, mats16, M_WA, lane);
          Ai = wa.mul(hl, lds_vec16(vl + V_B1 * EH));
        }
        // node MLP backward: h^{l+1} = h^l + Wn2 silu(zn) + bn2,  zn = Wn1a h^l + Wn1b agg + bn1
        f32x16 aggb = zero16;
        if (!last) {
          const f32x16 zn = ck_load16(ck + 16 * 64, lane);
          WFrag<1> wt;
          wt.load(nullptr, mats16, M_WN2T, lane);
          f32x16 znb = wt.mul(hb[T], zero16);
          { f32x16 yv, gv; silu_grad16(zn, yv, gv); znb *= gv; }
          wt.load(nullptr, mats16, M_WN1BT, lane);
          aggb = wt.mul(znb, zero16);
          if (l > 0 || want_h) {
            wt.load(nullptr, mats16, M_WN1AT, lane);
            hb[T] = wt.mul(znb, hb[T]);
          }
        }
        f32x16 S = zero16;
        float pacc[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) pacc[k] = 0.f;
        const int cbase = col[T] - nodei[T];
        const int live = valid[T] ? 1 : 0;  // columns beyond the group's walkers pair with themselves
        for (int dd = 1; dd < N; ++dd) {
          asm volatile("" ::: "memory");
          int j = nodei[T] + dd * live;
          j = (j >= N) ? j - N : j;
          const int cj = cbase + j;
          float df[DIM], e0[DIM], radial = 0.f, ea = 0.f;
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            df[k] = pl[k] - posb[cj * DIM + k];
            radial = fmaf(df[k], df[k], radial);
            e0[k] = p0i[T][k] - pos0[cj * DIM + k];
            ea = fmaf(e0[k], e0[k], ea);
          }
          // recompute the edge with derivative factors
          f32x16 z = Ai + lds_vec16(PB + cj * PBS + hh * 16);
          z = __builtin_amdgcn_mfma_f32_32x32x2f32(a_re, hh ? ea : radial, z, 0, 0, 0);
          f32x16 g1, g2, m2, gc;
          { f32x16 yv; silu_grad16(z, yv, g1); z = yv; }
          z = edge_gemm<PE>(w2f, z, lds_vec16(vl + V_B2 * EH));
          silu_grad16(z, m2, g2);
          float att = 1.0f;
          f32x16 m = m2;
          if (att_on) {
            att = fast_sigmoid(xhalf_sum(dot16(lds_vec16(vl + V_WATT * EH), m2)) + b_att);
            m *= att;
          }
          z = edge_gemm<PE>(wc1f, m, lds_vec16(vl + V_BC1 * EH));
          { f32x16 yv; silu_grad16(z, yv, gc); z = yv; }
          const f32x16 v_wc2 = lds_vec16(vl + V_WC2 * EH);
          float cs = xhalf_sum(dot16(v_wc2, z)), dcs_raw = 1.0f;
          if (tanh_on) {
            const float th = tanh_select(cs);
            dcs_raw = p.coord_scale * fmaf(-th, th, 1.0f);
            cs = th * p.coord_scale;
          }
          const float sq = sqrtf(radial + 1e-8f), inv = 1.0f / (sq + 1.0f);
          // backward: coordinate update pos_i += u cs, u = df / (sq + 1)
          float csb = 0.f, ub[DIM], udot = 0.f;
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            const float u = df[k] * inv;
            csb = fmaf(pb[T][k], u, csb);
            ub[k] = pb[T][k] * cs;
            udot = fmaf(ub[k], u, udot);
          }
          const float c1w = (kS * dcs_raw) * csb;  // true w_c2 = kS * packed w_c2
          z = (v_wc2 * c1w) * gc;  // zc_bar
          f32x16 mb = wc1t.mul(z, aggb);
          if (att_on) {
            const float attb = xhalf_sum(dot16(mb, m2)) * kSi;  // <m_bar, m2>, m2 held as kS m2
            const float lw = kS * (attb * att * (1.0f - att));   // true w_att = kS * packed w_att
            const f32x16 v_watt = lds_vec16(vl + V_WATT * EH);
#pragma unroll
            for (int r = 0; r < 16; ++r) mb[r] = fmaf(att, mb[r], lw * v_watt[r]);
          }
          mb *= g2;  // z2_bar
          f32x16 z1b = w2t.mul(mb, zero16);
          z1b *= g1;
          if (l > 0 || want_h) {  // h^0 does not depend on x (but on h, through the time feature)
            S += z1b;
            lds_add16(TB + cj * PBS + hh * 16, z1b);
          }
          float radb = xhalf_sum(dot16(lds_vec16(vl + V_WRF * EH), z1b));
          const float eab = xhalf_sum(dot16(lds_vec16(vl + V_WEF * EH), z1b));
          radb = fmaf(-udot * inv, 0.5f / sq, radb);  // through nrm = sqrt(radial + eps) + 1
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            const float dfb = fmaf(ub[k], inv, 2.0f * radb * df[k]);
            const float e0b = 2.0f * eab * e0[k];
            pacc[k] += dfb;
            p0acc[T][k] += e0b;
            if (hh == 0) {
              pbsc[cj * DIM + k] -= dfb;
              p0sc[cj * DIM + k] -= e0b;
            }
          }
        }
        if (l > 0 || want_h) {
          WFrag<1> wt;
          wt.load(nullptr, mats16, M_WAT, lane);
          hb[T] = wt.mul(S, hb[T]);
        }
#pragma unroll
        for (int k = 0; k < DIM; ++k) pb[T][k] += pacc[k];
      }
      wave_lds_fence();  // all scatter contributions of this layer are in LDS
      {
        WFrag<1> wt;
        if (l > 0 || want_h) wt.load(nullptr, mats16, M_WBT, lane);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          if (T >= ntile) continue;
#pragma unroll
          for (int k = 0; k < DIM; ++k) pb[T][k] += pbsc[col[T] * DIM + k];
          if (l > 0 || want_h) hb[T] = wt.mul(lds_vec16(TB + col[T] * PBS + hh * 16), hb[T]);
        }
      }
      wave_lds_fence();
    }

    // pos^0 = pos0 = c_in x:  y_bar = pos_bar^0 + (edge-attribute path) - v   (F = pos^L - pos^0)
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      if (!(valid[T] && hh == 0)) continue;
      const bool wr = !p.mark || p.mark[walker0 + col[T] / N] != 0;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        const float yb = pb[T][k] + p0acc[T][k] + p0sc[col[T] * DIM + k] - vfin[T][k];
        if (wr) p.vjp[(walker0 * N + col[T]) * DIM + k] = fmaf(c_s[T], cot[T][k], c_in[T] * yb);
        if (want_h) dhacc[T] = fmaf((-0.5f * c_in[T] / (1.0f + hvv[T])) * yb, xin[T][k], dhacc[T]);  // through c_in(h) x
      }
    }
    wave_lds_fence();
    if (want_h) {
      // through the time feature c_noise = ln(h)/8: t_bar = sum_nodes <h_bar^0, d h^0 / dt>; then one lane per walker
      // adds up its particles' partial sums (both feature halves) in a fixed order
      float* red = TB;  // [4][NCOLP]: the sweep's sum per feature half, c_out <cot, F>, c_s' <cot, x>
      const f32x16 w0 = lds_vec16(vemb + hh * 16), w1 = lds_vec16(vemb + 32 + hh * 16);
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        float tb = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) tb = fmaf(hb[T][r], (a0t[T] ? w0[r] : 0.f) + (a1t[T] ? w1[r] : 0.f), tb);
        const float part = valid[T] ? fmaf(tb, 0.125f / hvv[T], hh == 0 ? dhacc[T] : 0.f) : 0.f;
        red[hh * C::NCOLP + col[T]] = part;
        if (hh == 0) {
          red[2 * C::NCOLP + col[T]] = valid[T] ? facc[T] : 0.f;
          red[3 * C::NCOLP + col[T]] = valid[T] ? cxacc[T] : 0.f;
        }
      }
      wave_lds_fence();
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        if (!(valid[T] && hh == 0 && nodei[T] == 0)) continue;
        float sum = 0.f, sum_f = 0.f, sum_x = 0.f;
        for (int q = 0; q < N; ++q) {
          sum += red[col[T] + q] + red[C::NCOLP + col[T] + q];
          sum_f += red[2 * C::NCOLP + col[T] + q];
          sum_x += red[3 * C::NCOLP + col[T] + q];
        }
        if (!p.mark || p.mark[walker0 + col[T] / N]) {
          p.dot_h[walker0 + col[T] / N] = sum + sum_x;
          if (p.dot_parts) {
            p.dot_parts[2 * (walker0 + col[T] / N)] = sum_f;
            p.dot_parts[2 * (walker0 + col[T] / N) + 1] = sum;
          }
        }
      }
      wave_lds_fence();
    }
  }
}

// marks the walkers whose results of the f16-path launch are not finite (one wavefront per walker: coalesced rows)
__global__ void __launch_bounds__(256) vjp_mark_kernel(const float* __restrict__ vjp, const float* __restrict__ dot_h,
                                                       const float* __restrict__ out, long long B, int D,
                                                       int* __restrict__ mark, int* __restrict__ flag) {
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= B) return;
  bool bad = lane == 0 && dot_h && !__builtin_isfinite(dot_h[w]);
  for (int k = lane; k < D; k += 64) {
    bad = bad || !__builtin_isfinite(vjp[w * D + k]);
    if (out) bad = bad || !__builtin_isfinite(out[w * D + k]);
  }
  const bool any = __builtin_amdgcn_ballot_w64(bad) != 0ull;
  if (lane == 0) {
    mark[w] = any ? 1 : 0;
    if (any) *flag = 1;
  }
}

struct VjpShape {
  int n, dim, G, waves;
  void (*kernel[2])(VjpParams);  // [FIXED], bf16x3 edge GEMMs (also the repair pass)
  void (*kernel16[2])(VjpParams);  // [FIXED], f16 two-piece edge GEMMs
  size_t (*lds_bytes)(int);
  size_t (*ws_f)(int);
};
template <int N, int DIM, int G, int WAVES>
static size_t vjp_lds_bytes_of(int L) { return VjpCfg<N, DIM, G, WAVES>::lds_bytes(L); }
template <int N, int DIM, int G, int WAVES>
static size_t vjp_ws_f_of(int L) { return VjpCfg<N, DIM, G, WAVES>::ws_f(L); }
#define PITA_VJP_SHAPE(N, DIM, G, WAVES)                                                         \
  VjpShape { N, DIM, G, WAVES, {egnn_vjp_kernel<N, DIM, G, WAVES, false, 1>, egnn_vjp_kernel<N, DIM, G, WAVES, true, 1>}, \
             {egnn_vjp_kernel<N, DIM, G, WAVES, false, 2>, egnn_vjp_kernel<N, DIM, G, WAVES, true, 2>}, \
             vjp_lds_bytes_of<N, DIM, G, WAVES>, \
             vjp_ws_f_of<N, DIM, G, WAVES> }
static const VjpShape kVjpShapes[] = {
    PITA_VJP_SHAPE(4, 2, 8, 4),
    PITA_VJP_SHAPE(13, 3, 7, 4),
    PITA_VJP_SHAPE(22, 3, 4, 4),
    PITA_VJP_SHAPE(55, 3, 1, 4),
};
// Latency mapping for batches that underfill the chip (round 6).  The reference's own operating point is 2 048 walkers
// (configs/experiment/lj13.yaml:32): seven walkers per wave make 293 waves on 1 024 SIMDs, each walking three column tiles
// -- 492 us per launch, 40 % of a debiased step there.  Two walkers per wave (one column tile) fill every SIMD once.
static const VjpShape kVjpShapesSmall[] = {
    PITA_VJP_SHAPE(13, 3, 2, 4),
};
// column-tile passes of the busiest wave: (groups per wave) x (tiles per group)
static long long vjp_passes(const VjpShape& s, long long B, long long wave_slots) {
  const long long ngroups = (B + s.G - 1) / s.G;
  return ((ngroups + wave_slots - 1) / wave_slots) * ((s.G * s.n + 31) / 32);
}

}  // namespace pita

using namespace pita;

extern "C" int pita_egnn_vjp(pita_egnn_t* net, const float* h, const float* x, const float* beta, const float* cot,
                             float* out, float* vjp, float* dot_h, float* dot_parts, int64_t B, void* stream) {
  PITA_REQUIRE(net && B >= 0, "pita_egnn_vjp: bad argument");
  PITA_REQUIRE(dot_h || !dot_parts, "pita_egnn_vjp: dot_parts needs dot_h");
  if (B == 0) return PITA_OK;
  PitaDeviceGuard guard(net->device);
  PITA_REQUIRE(h && x && vjp, "pita_egnn_vjp: null argument");
  PITA_REQUIRE(beta || net->cfg.in_node_nf == 1, "pita_egnn_vjp: beta required for in_node_nf=2");
  const VjpShape* s = nullptr;
  for (const auto& c : kVjpShapes)
    if (c.n == net->cfg.n_particles && c.dim == net->cfg.n_dim) s = &c;
  if (!s) return fail(PITA_EUNSUPPORTED, "pita_egnn_vjp: no kernel for this particle system");
  for (const auto& c : kVjpShapesSmall)  // fewer tile passes on the busiest wave with the small groups: take them
    if (c.n == s->n && c.dim == s->dim &&
        vjp_passes(c, B, (long long)net->n_cu * c.waves) < vjp_passes(*s, B, (long long)net->n_cu * s->waves))
      s = &c;
  VjpParams p{};
  p.mats16 = net->d_mats16; p.vecs = net->d_vecs; p.n_layers = net->cfg.n_layers; p.in_nf = net->cfg.in_node_nf;
  p.attention = net->cfg.attention; p.tanh_on = net->cfg.tanh; p.feature_layout = net->cfg.feature_layout;
  p.coord_scale = net->cfg.coords_range / (float)net->cfg.n_layers;
  p.B = B; p.h = h; p.x = x; p.beta = beta; p.cot = cot; p.out = out; p.vjp = vjp; p.dot_h = dot_h;
  p.dot_parts = dot_parts;
  const size_t lds = s->lds_bytes(p.n_layers);
  const int fixed = (p.attention && p.tanh_on && dot_h) ? 1 : 0;
  static const bool force_bf16 = getenv("PITA_VJP_BF16") != nullptr;  // development aid: A/B against the bf16x3 edge GEMMs
  const bool f16 = net->cfg.precision == 2 && !force_bf16;
  p.mats16h = net->d_mats16h;
  hipStream_t st = (hipStream_t)stream;
  auto configure = [&](const void* k) -> int {
    PITA_HIP_CHECK(ensure_dynamic_lds(k, lds));
    return PITA_OK;
  };
  const long long ngroups = (B + s->G - 1) / s->G;
  long long want = (ngroups + s->waves - 1) / s->waves;
  const long long cap = net->n_cu;  // one 4-wave block per CU (one wave per SIMD)
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  const size_t need = sizeof(float) * s->ws_f(p.n_layers) * (size_t)grid * s->waves;
  if (net->ws_bytes < need) {  // checkpoint scratch, owned by the handle (one stream at a time, see pita_hip.h)
    PITA_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(net->d_ws);
    net->d_ws = nullptr; net->ws_bytes = 0;
    PITA_HIP_CHECK(hipMalloc(&net->d_ws, need));
    net->ws_bytes = need;
  }
  p.ws = net->d_ws;
  const auto kernel = s->kernel[fixed];
  if (!f16) {
    const int rc = configure(reinterpret_cast<const void*>(kernel));
    if (rc != PITA_OK) return rc;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(s->waves * 64), lds, st, p);
    PITA_LAUNCH_CHECK();
    return PITA_OK;
  }
  // f16 edge GEMMs first; walkers that came out non-finite are marked and recomputed by the bf16x3 kernel, which returns
  // at once when nothing was marked
  if (sizeof(int) * ((size_t)B + 16) > net->vjp_mark_bytes) {
    PITA_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(net->d_vjp_mark);
    net->d_vjp_mark = nullptr; net->vjp_mark_bytes = 0;
    PITA_HIP_CHECK(hipMalloc(&net->d_vjp_mark, sizeof(int) * ((size_t)B + 16)));
    net->vjp_mark_bytes = sizeof(int) * ((size_t)B + 16);
  }
  int* mark = net->d_vjp_mark;
  int* flag = mark + B;
  PITA_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), st));
  const auto kernel16 = s->kernel16[fixed];
  int rc = configure(reinterpret_cast<const void*>(kernel16));
  if (rc != PITA_OK) return rc;
  hipLaunchKernelGGL(kernel16, dim3(grid), dim3(s->waves * 64), lds, st, p);
  PITA_LAUNCH_CHECK();
  hipLaunchKernelGGL(vjp_mark_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, vjp, dot_h, out, (long long)B,
                     net->cfg.n_particles * net->cfg.n_dim, mark, flag);
  PITA_LAUNCH_CHECK();
  p.mark = mark;
  p.flag = flag;
  rc = configure(reinterpret_cast<const void*>(kernel));
  if (rc != PITA_OK) return rc;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(s->waves * 64), lds, st, p);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}
