// MLP score backbone (MyMLP / MyMLPTemperature) for gfx950 on the bf16 matrix pipe with the exact 3-way operand split
// (fp32-equivalent, see egnn_common.h; the first version used v_mfma_f32_32x32x2_f32, which shares the VALU datapath
// with the sin/cos/erf work and ran 2x slower).
//
// Replaces (paths relative to /root/reference/pita/src/models/components/):
//   mlp.py:11-24    SinusoidalEmbedding  (per-coordinate scale 25, time / beta scale 1)
//   mlp.py:100-118  Block                (x + GELU(Linear(x)))
//   mlp.py:244-267  MyMLP.forward ; mlp.py:501-524 MyMLPTemperature.forward
//
// Mapping: one wavefront = 32 walkers = the 32 columns of a 32x32 MFMA tile.  Activations
// live in registers in the MFMA C/D layout (lane = walker column, 16 of every 32 features per
// lane) -- the same chaining trick as the EGNN kernel, so no layer ever leaves the register
// file.  The sinusoidal embedding is generated on the fly, 32 features at a time, straight into
// the B-operand layout of the first GEMM (the [B, 384] embedding tensor is never materialised).
// Weights are pre-packed into per-lane fragment order and stream from L2.
#include "egnn_common.h"

namespace pita {

struct MlpParams {
  const unsigned* w0;   // [NB][KC] blocks of MAT_W words: bf16 three-way split fragments ([piece][k-step][lane][4])
  const unsigned* wl;   // [L][NB][NB] blocks
  const unsigned* wf;   // [NBO][NB] blocks
  const unsigned* stream;  // all blocks once more in the order mlp_tile consumes them (null: emb_size/2 % 32 != 0)
  int S;
  const float* b0;   // [NB][32]  fragment order
  const float* bl;   // [L][NB][32]
  const float* bf;   // [NBO][32]
  const float* freqs;  // [E/2]
  int input_dim, out_dim, n_layers, emb, temp, KC, NBO;
  long long B;
  const float* t;
  const float* x;
  const float* beta;
  float* out;
};

__device__ __forceinline__ f32x16 mlp_bias(const float* b, int hh) {
  const f32x4* p = reinterpret_cast<const f32x4*>(b + hh * 16);
  f32x4 a = p[0], bq = p[1], c = p[2], d = p[3];
  f32x16 r;
  r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = bq.x; r[5] = bq.y; r[6] = bq.z; r[7] = bq.w;
  r[8] = c.x; r[9] = c.y; r[10] = c.z; r[11] = c.w; r[12] = d.x; r[13] = d.y; r[14] = d.z; r[15] = d.w;
  return r;
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 rounding level) on v_rcp / v_exp: the library erff
// and sincosf (range reduction of angles up to ~1e4 rad) were 3/4 of the kernel's instructions
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float pl = 1.061405429f;
  pl = fmaf(pl, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float r = 1.0f - (pl * t) * __builtin_amdgcn_exp2f(-1.44269504088896341f * ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752f)); }

// sin and cos of an fp32 angle in radians: the angle itself is the reference's fp32 value; its reduction to one
// revolution is done in fp64 (exact to 1e-13 rev for |angle| < 1e5), then the transcendental unit evaluates
// sin / cos of revolutions (absolute error ~1e-6, two orders below the sensitivity of sin to the fp32 rounding of such
// angles)
__device__ __forceinline__ void sincos_rev(float ang, float& sn, float& cs) {
  const double r = (double)ang * 0.15915494309189535;
  const float f = (float)(r - floor(r));
  sn = __builtin_amdgcn_sinf(f);
  cs = __builtin_amdgcn_cosf(f);
}

// ---- where a tile's weight blocks come from
// GlobalWeights: every wave streams its own copy of each block from L2 (600 KB per 32-walker tile-pass for the
//   128-wide net: the limit of the first version at large batches).
// StreamWeights: the four waves of a workgroup walk the same block sequence in lock step; each block is fetched once
//   per workgroup -- global -> registers while the previous block's MFMAs run, -> LDS, one barrier -- and read from LDS
//   by all four waves (double buffered).  The host packs the blocks in consumption order (`stream`).
struct GlobalWeights {
  const MlpParams& p;
  int lane;
  __device__ __forceinline__ WFrag<1> fetch(const unsigned* base, int idx) {
    WFrag<1> w;
    w.load(nullptr, base, idx, lane);
    return w;
  }
  __device__ __forceinline__ void done() {}
};

struct StreamWeights {
  const unsigned* stream;  // [S][MAT_W]
  unsigned* buf;           // LDS [2][MAT_W]
  int S, s, cur, tid, lane;
  uint2 pre[3];
  __device__ __forceinline__ void prime() {
    const uint2* g = reinterpret_cast<const uint2*>(stream);
    uint2* l = reinterpret_cast<uint2*>(buf);
#pragma unroll
    for (int q = 0; q < 3; ++q) l[tid + 256 * q] = g[tid + 256 * q];
    s = 0;
    cur = 0;
    __syncthreads();
  }
  __device__ __forceinline__ WFrag<1> fetch(const unsigned*, int) {
    const int nxt = (s + 1 == S) ? 0 : s + 1;
    const uint2* g = reinterpret_cast<const uint2*>(stream + (size_t)nxt * MAT_W);
#pragma unroll
    for (int q = 0; q < 3; ++q) pre[q] = g[tid + 256 * q];  // in flight while this block's MFMAs run
    WFrag<1> w;
    const u32x4* pl = reinterpret_cast<const u32x4*>(buf + cur * MAT_W) + lane;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
#pragma unroll
      for (int st = 0; st < 2; ++st) w.w[pc][st] = pl[(pc * 2 + st) * 64];
    return w;
  }
  __device__ __forceinline__ void done() {
    uint2* l = reinterpret_cast<uint2*>(buf + (cur ^ 1) * MAT_W);
#pragma unroll
    for (int q = 0; q < 3; ++q) l[tid + 256 * q] = pre[q];
    __syncthreads();  // next block visible; everybody has read the current one
    cur ^= 1;
    s = (s + 1 == S) ? 0 : s + 1;
  }
};

// One 32-walker tile through the network.  xrow: this lane's walker coordinates [input_dim] (global or LDS), used as
// xrow[var] * xscale (xscale = c_in of the EDM preconditioning in the fused sampler, 1 in the plain forward: x * 1.0f
// is exact); emit(row, value) receives output row `row` of this lane's walker from the lane that holds it.
template <int NB, typename Weights, typename Emit>
__device__ __forceinline__ void mlp_tile(const MlpParams& p, Weights& W, int lane, int hh, const float* xrow, float xscale,
                                         float tv, float bv, Emit&& emit) {
  const int half = p.emb >> 1;
  // ---- layer 0: GELU(W0 . [emb(x_0) .. emb(x_{D-1}), emb(t), (emb(beta))] + b0)
  f32x16 z[NB];
#pragma unroll
  for (int ob = 0; ob < NB; ++ob) z[ob] = mlp_bias(p.b0 + ob * 32, hh);
  auto feed = [&](const f32x16& e, int kc) {
    u32x4 es[3][2];
    WFrag<1>::split(e, es);  // one split per chunk, shared by the NB output blocks
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      const WFrag<1> w = W.fetch(p.w0, ob * p.KC + kc);
      z[ob] = w.mul_split(es, z[ob]);
      W.done();
    }
  };
  auto input_of = [&](int var, float& v, float& scale) {
    if (var < p.input_dim) { v = xrow[var] * xscale; scale = 25.0f; }
    else if (var == p.input_dim) { v = tv; scale = 1.0f; }
    else { v = bv; scale = 1.0f; }
  };
  if ((half & 31) == 0) {
    // a 32-feature chunk is all-sine or all-cosine of one variable, and the cosine chunk `half/32` chunks later uses the
    // same angles: one sincosf serves both (the precise range reduction is the expensive part of either)
    const int hc = half >> 5, per_var = 2 * hc;
    for (int kc = 0; kc < p.KC; ++kc) {
      const int var = kc / per_var, c = kc - var * per_var;
      if (c >= hc) continue;  // cosine chunks are emitted together with their sine chunk
      float v, scale;
      input_of(var, v, scale);
      f32x16 es_, ec_;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int idx = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        float sn, cs;
        sincos_rev((v * scale) * p.freqs[idx], sn, cs);
        es_[r] = sn;
        ec_[r] = cs;
      }
      feed(es_, kc);
      feed(ec_, kc + hc);
    }
  } else {
    for (int kc = 0; kc < p.KC; ++kc) {
      f32x16 e;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = kc * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int var = f / p.emb, idx = f - var * p.emb;
        float v, scale;
        input_of(var, v, scale);
        float sn, cs;
        sincos_rev((v * scale) * p.freqs[idx < half ? idx : idx - half], sn, cs);
        e[r] = idx < half ? sn : cs;
      }
      feed(e, kc);
    }
  }
#pragma unroll
  for (int ob = 0; ob < NB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) z[ob][r] = gelu_erf(z[ob][r]);
  // ---- residual blocks: z += GELU(W_l z + b_l)
  for (int l = 0; l < p.n_layers; ++l) {
    f32x16 nz[NB];
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) nz[ob] = mlp_bias(p.bl + ((size_t)l * NB + ob) * 32, hh);
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      u32x4 zs[3][2];
      WFrag<1>::split(z[kb], zs);
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) {
        const WFrag<1> w = W.fetch(p.wl, (l * NB + ob) * NB + kb);
        nz[ob] = w.mul_split(zs, nz[ob]);
        W.done();
      }
    }
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) z[ob][r] += gelu_erf(nz[ob][r]);
  }
  // ---- output head
  for (int ob = 0; ob < p.NBO; ++ob) {
    f32x16 o = mlp_bias(p.bf + ob * 32, hh);
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      const WFrag<1> w = W.fetch(p.wf, ob * NB + kb);
      o = w.mul(z[kb], o);
      W.done();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (row < p.out_dim) emit(row, o[r]);
    }
  }
}

template <int NB, bool STREAM>
__global__ void __launch_bounds__(256, 2) mlp_kernel(MlpParams p) {
  __shared__ __attribute__((aligned(16))) unsigned wbuf[STREAM ? 2 * MAT_W : 4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane & 31, hh = lane >> 5;
  const long long ntile = (p.B + 31) / 32, ngroup = (ntile + 3) / 4;
  StreamWeights SW{p.stream, wbuf, p.S, 0, 0, (int)threadIdx.x, lane, {}};
  GlobalWeights GW{p, lane};
  if (STREAM) SW.prime();
  for (long long grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {  // block-uniform trip count (barriers inside)
    const long long tile = grp * 4 + wave;
    const long long wid = tile * 32 + cl;
    const bool valid = wid < p.B;
    const long long wl = valid ? wid : p.B - 1;
    auto emit = [&](int row, float v) { if (valid) p.out[wid * p.out_dim + row] = v; };
    if (STREAM) mlp_tile<NB>(p, SW, lane, hh, p.x + wl * p.input_dim, 1.0f, p.t[wl], p.beta ? p.beta[wl] : 0.f, emit);
    else mlp_tile<NB>(p, GW, lane, hh, p.x + wl * p.input_dim, 1.0f, p.t[wl], p.beta ? p.beta[wl] : 0.f, emit);
  }
}

// ---- fused sampler: all Euler-Maruyama steps of the not-debiased reverse SDE in one launch (the MLP counterpart of
// egnn_kernel's mode 3; sde_integration.py:299-351 + sdes.py:117-128,245-251 + score_net.py:13-43).  A wave keeps its
// 32 walkers in LDS for the whole launch: per step the network sees c_in x and c_noise, the lanes holding the output
// rows drop F into LDS, and lane (walker, hh) updates the coordinates hh, hh + 2, ... of its walker with the same
// arithmetic as the EGNN sampler / pita_em_step.
struct MlpSamplerParams {
  MlpParams m;
  float* x;
  const float* step_tab;
  const float* noise;
  int n_steps, remove_mean, n_particles, n_dim;
  unsigned long long seed, walker_offset;
  long long step0;
  double* stats_out;  // nullable [n_steps][4], as pita_egnn_sampler_run
};

__device__ __forceinline__ void mlp_wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int NB, bool STREAM>
__global__ void __launch_bounds__(256, 2) mlp_sampler_kernel(MlpSamplerParams q) {
  extern __shared__ float sm[];
  __shared__ __attribute__((aligned(16))) unsigned wbuf[STREAM ? 2 * MAT_W : 4];
  const MlpParams& p = q.m;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane & 31, hh = lane >> 5;
  const int D = p.input_dim;
  float* xs = sm + wave * 2 * 32 * D;  // [32][D] walkers of this wave
  float* fs = xs + 32 * D;             // [32][D] network output F
  const long long ntile = (p.B + 31) / 32, ngroup = (ntile + 3) / 4;
  StreamWeights SW{p.stream, wbuf, p.S, 0, 0, (int)threadIdx.x, lane, {}};
  GlobalWeights GW{p, lane};
  if (STREAM) SW.prime();
  for (long long grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {  // block-uniform trip count (barriers inside)
    const long long tile = grp * 4 + wave;
    const long long w0 = tile * 32;
    const int nw = (w0 >= p.B) ? 0 : (int)((p.B - w0) < 32 ? (p.B - w0) : 32);
    for (int i = lane; i < 32 * D; i += 64) xs[i] = (i < nw * D) ? q.x[w0 * D + i] : 0.f;
    mlp_wave_fence();
    const long long wid = w0 + cl;
    float* xrow = xs + cl * D;
    float* frow = fs + cl * D;
    for (int s = 0; s < q.n_steps; ++s) {
      const float* st = q.step_tab + (size_t)s * PITA_STEP_STRIDE;
      const float c_s = st[PITA_ST_CS], c_in = st[PITA_ST_CIN], c_out = st[PITA_ST_COUT], hv = st[PITA_ST_H];
      const float g2 = st[PITA_ST_G2], gamma = st[PITA_ST_GAMMA], dt = st[PITA_ST_DT];
      const float noise_scale = st[PITA_ST_NOISE_SCALE], sqrt_dt = st[PITA_ST_SQRT_DT];
      auto emit = [&](int row, float v) { frow[row] = v; };
      if (STREAM) mlp_tile<NB>(p, SW, lane, hh, xrow, c_in, st[PITA_ST_CNOISE], st[PITA_ST_BETA], emit);
      else mlp_tile<NB>(p, GW, lane, hh, xrow, c_in, st[PITA_ST_CNOISE], st[PITA_ST_BETA], emit);
      mlp_wave_fence();
      float st_d = 0.f, st_d2 = 0.f, st_n = 0.f, st_n2 = 0.f;
      for (int var = hh; var < D; var += 2) {
        const float xv = xrow[var];
        float xi;
        if (q.noise) {
          xi = (cl < nw) ? q.noise[((long long)s * p.B + wid) * D + var] : 0.f;
        } else {
          float z4[4];
          philox_normal4(q.seed, q.walker_offset + (unsigned long long)wid, q.step0 + s, (uint32_t)(var / q.n_dim), z4);
          const int c = var - (var / q.n_dim) * q.n_dim;
          xi = c == 0 ? z4[0] : (c == 1 ? z4[1] : (c == 2 ? z4[2] : z4[3]));
        }
        const float Dth = c_s * xv + c_out * frow[var];
        const float sc = (Dth - xv) / hv;
        const float drift = gamma * (sc * g2);
        const float dif = noise_scale * xi;
        if (q.stats_out && cl < nw) {
          st_d += drift; st_d2 = fmaf(drift, drift, st_d2);
          st_n += dif; st_n2 = fmaf(dif, dif, st_n2);
        }
        xrow[var] = xv + (drift * dt + (dif * sqrt_dt));
      }
      if (q.stats_out) {
        double m4[4] = {(double)st_d, (double)st_d2, (double)st_n, (double)st_n2};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          for (int o = 32; o > 0; o >>= 1) m4[k] += __shfl_xor(m4[k], o, 64);
          if (lane == 0) atomicAdd(q.stats_out + (size_t)s * 4 + k, m4[k]);
        }
      }
      mlp_wave_fence();
      if (q.remove_mean) {  // per-dimension particle means, staged through the (now free) F rows
        for (int var = hh; var < D; var += 2) {
          const int c = var % q.n_dim;
          float sum = 0.f;
          for (int i = 0; i < q.n_particles; ++i) sum += xrow[i * q.n_dim + c];
          frow[var] = sum / (float)q.n_particles;
        }
        mlp_wave_fence();
        for (int var = hh; var < D; var += 2) xrow[var] -= frow[var];
        mlp_wave_fence();
      }
    }
    for (int i = lane; i < nw * D; i += 64) q.x[w0 * D + i] = xs[i];
    mlp_wave_fence();
  }
}

}  // namespace pita

struct pita_mlp {
  pita_mlp_config cfg;
  float* d_all = nullptr;
  pita::MlpParams p{};
};

using namespace pita;

static inline int mlp_kfeat(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

extern "C" int64_t pita_mlp_num_weights(const pita_mlp_config* c) {
  if (!c) return PITA_EINVAL;
  const int64_t C = (int64_t)c->emb_size * (c->input_dim + 1 + (c->temperature_conditioned ? 1 : 0));
  const int64_t H = c->hidden_size;
  // the reference sizes the head by emb_size (mlp.py:238-239), valid only when emb_size == hidden_size
  return (H * C + H) + (int64_t)c->hidden_layers * (H * H + H) + ((int64_t)c->out_dim * H + c->out_dim);
}

extern "C" int pita_mlp_create(pita_mlp_t** out, const pita_mlp_config* cfg, const float* w, int64_t n_weights,
                               const float* freqs) {
  PITA_REQUIRE(out && cfg && w && freqs, "pita_mlp_create: null argument");
  const int H = cfg->hidden_size, E = cfg->emb_size, L = cfg->hidden_layers;
  const int C = E * (cfg->input_dim + 1 + (cfg->temperature_conditioned ? 1 : 0));
  if (!(H == 32 || H == 64 || H == 128))
    return fail(PITA_EUNSUPPORTED, "pita_mlp_create: hidden_size=%d (32, 64, 128 implemented)", H);
  if (E != H) return fail(PITA_EUNSUPPORTED, "pita_mlp_create: emb_size must equal hidden_size (reference head sizing)");
  PITA_REQUIRE(E % 2 == 0 && C % 32 == 0, "pita_mlp_create: emb_size*(inputs) must be a multiple of 32");
  PITA_REQUIRE(cfg->out_dim >= 1 && cfg->input_dim >= 1 && L >= 0, "pita_mlp_create: bad dims");
  PITA_REQUIRE(n_weights == pita_mlp_num_weights(cfg), "pita_mlp_create: got %lld weights, expected %lld",
               (long long)n_weights, (long long)pita_mlp_num_weights(cfg));
  const int NB = H / 32, KC = C / 32, NBO = (cfg->out_dim + 31) / 32;
  const size_t n_w0 = (size_t)NB * KC * MAT_W, n_wl = (size_t)L * NB * NB * MAT_W, n_wf = (size_t)NBO * NB * MAT_W;
  const size_t n_b0 = (size_t)NB * 32, n_bl = (size_t)L * NB * 32, n_bf = (size_t)NBO * 32, n_fr = E / 2;
  const bool streamable = ((E / 2) % 32) == 0;  // pure sine / cosine chunks: the order mlp_tile consumes blocks is fixed
  const int S = NB * KC + L * NB * NB + NBO * NB;
  const size_t n_st = streamable ? (size_t)S * MAT_W : 0;
  const size_t total = n_w0 + n_wl + n_wf + n_b0 + n_bl + n_bf + n_fr + n_st;
  float* h = new float[total]();
  float* h_w0 = h; float* h_wl = h_w0 + n_w0; float* h_wf = h_wl + n_wl;
  float* h_b0 = h_wf + n_wf; float* h_bl = h_b0 + n_b0; float* h_bf = h_bl + n_bl; float* h_fr = h_bf + n_bf;
  float* h_st = h_fr + n_fr;
  // one 32x32 block as bf16 three-way truncation-split MFMA fragments: word q of (piece, k-step st, lane) packs the
  // pieces of elements r = 8 st + 2 q (low half) and r + 1 (high half); same layout as the EGNN weights
  auto trunc16 = [](float v) { unsigned u; memcpy(&u, &v, 4); u &= 0xFFFF0000u; float o; memcpy(&o, &u, 4); return o; };
  auto hi16 = [](float v) { unsigned u; memcpy(&u, &v, 4); return u >> 16; };
  auto pack_block = [&](float* dstf, const float* M, int rows, int ld, int ob, int kb) {
    unsigned* dst = reinterpret_cast<unsigned*>(dstf);
    for (int lane = 0; lane < 64; ++lane)
      for (int st = 0; st < 2; ++st)
        for (int qd = 0; qd < 4; ++qd) {
          unsigned pcs[2][3];
          for (int e = 0; e < 2; ++e) {
            const int row = ob * 32 + (lane & 31), col = kb * 32 + mlp_kfeat(8 * st + 2 * qd + e, lane >> 5);
            const float w = (row < rows) ? M[(size_t)row * ld + col] : 0.f;
            const float w1 = trunc16(w), r1 = w - w1, w2 = trunc16(r1), r2 = r1 - w2;
            pcs[e][0] = hi16(w1); pcs[e][1] = hi16(w2); pcs[e][2] = hi16(r2);
          }
          for (int pc = 0; pc < 3; ++pc)
            dst[(((size_t)pc * 2 + st) * 64 + lane) * 4 + qd] = pcs[0][pc] | (pcs[1][pc] << 16);
        }
  };
  auto pack_bias = [&](float* dst, const float* b, int rows, int ob) {
    for (int hh = 0; hh < 2; ++hh)
      for (int r = 0; r < 16; ++r) {
        const int row = ob * 32 + mlp_kfeat(r, hh);
        dst[hh * 16 + r] = row < rows ? b[row] : 0.f;
      }
  };
  const float* q = w;
  const float* W0 = q; q += (size_t)H * C;
  const float* B0 = q; q += H;
  for (int ob = 0; ob < NB; ++ob) {
    for (int kc = 0; kc < KC; ++kc) pack_block(h_w0 + ((size_t)ob * KC + kc) * MAT_W, W0, H, C, ob, kc);
    pack_bias(h_b0 + ob * 32, B0, H, ob);
  }
  for (int l = 0; l < L; ++l) {
    const float* Wl = q; q += (size_t)H * H;
    const float* Bl = q; q += H;
    for (int ob = 0; ob < NB; ++ob) {
      for (int kb = 0; kb < NB; ++kb) pack_block(h_wl + (((size_t)l * NB + ob) * NB + kb) * MAT_W, Wl, H, H, ob, kb);
      pack_bias(h_bl + ((size_t)l * NB + ob) * 32, Bl, H, ob);
    }
  }
  const float* Wf = q; q += (size_t)cfg->out_dim * H;
  const float* Bf = q;
  for (int ob = 0; ob < NBO; ++ob) {
    for (int kb = 0; kb < NB; ++kb) pack_block(h_wf + ((size_t)ob * NB + kb) * MAT_W, Wf, cfg->out_dim, H, ob, kb);
    pack_bias(h_bf + ob * 32, Bf, cfg->out_dim, ob);
  }
  for (size_t i = 0; i < n_fr; ++i) h_fr[i] = freqs[i];
  if (streamable) {  // the blocks once more, in mlp_tile's consumption order (see StreamWeights)
    float* dst = h_st;
    auto put = [&](const float* blk) { memcpy(dst, blk, sizeof(float) * MAT_W); dst += MAT_W; };
    const int hc = (E / 2) / 32, per_var = 2 * hc;
    for (int kc = 0; kc < KC; ++kc) {
      if (kc % per_var >= hc) continue;
      for (int ob = 0; ob < NB; ++ob) put(h_w0 + ((size_t)ob * KC + kc) * MAT_W);
      for (int ob = 0; ob < NB; ++ob) put(h_w0 + ((size_t)ob * KC + kc + hc) * MAT_W);
    }
    for (int l = 0; l < L; ++l)
      for (int kb = 0; kb < NB; ++kb)
        for (int ob = 0; ob < NB; ++ob) put(h_wl + (((size_t)l * NB + ob) * NB + kb) * MAT_W);
    for (int ob = 0; ob < NBO; ++ob)
      for (int kb = 0; kb < NB; ++kb) put(h_wf + ((size_t)ob * NB + kb) * MAT_W);
  }
  pita_mlp* net = new pita_mlp();
  net->cfg = *cfg;
  hipError_t e = hipMalloc(&net->d_all, total * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(net->d_all, h, total * sizeof(float), hipMemcpyHostToDevice);
  delete[] h;
  if (e != hipSuccess) {
    (void)hipFree(net->d_all);
    delete net;
    return fail(PITA_EHIP, "pita_mlp_create: device upload failed: %s", hipGetErrorString(e));
  }
  MlpParams& p = net->p;
  p.w0 = reinterpret_cast<const unsigned*>(net->d_all); p.wl = p.w0 + n_w0; p.wf = p.wl + n_wl;
  p.b0 = net->d_all + n_w0 + n_wl + n_wf; p.bl = p.b0 + n_b0;
  p.stream = streamable ? reinterpret_cast<const unsigned*>(net->d_all + n_w0 + n_wl + n_wf + n_b0 + n_bl + n_bf + n_fr) : nullptr;
  p.S = S;
  p.bf = p.bl + n_bl; p.freqs = p.bf + n_bf;
  p.input_dim = cfg->input_dim; p.out_dim = cfg->out_dim; p.n_layers = L; p.emb = E;
  p.temp = cfg->temperature_conditioned; p.KC = KC; p.NBO = NBO;
  *out = net;
  return PITA_OK;
}

extern "C" int pita_mlp_destroy(pita_mlp_t* net) {
  if (!net) return PITA_OK;
  (void)hipFree(net->d_all);
  delete net;
  return PITA_OK;
}

extern "C" int pita_mlp_forward(pita_mlp_t* net, const float* t, const float* x, const float* beta, float* out,
                                int64_t B, void* stream) {
  PITA_REQUIRE(net && t && x && out && B >= 0, "pita_mlp_forward: null argument");
  PITA_REQUIRE(beta || !net->cfg.temperature_conditioned, "pita_mlp_forward: beta required (temperature_conditioned)");
  if (B == 0) return PITA_OK;
  MlpParams p = net->p;
  p.B = B; p.t = t; p.x = x; p.beta = beta; p.out = out;
  const long long nblk = ((B + 31) / 32 + 3) / 4;
  const unsigned grid = (unsigned)(nblk < 4096 ? nblk : 4096);
  hipStream_t s = (hipStream_t)stream;
  const int nb = net->cfg.hidden_size / 32;
  if (p.stream) {
    if (nb == 1) hipLaunchKernelGGL((mlp_kernel<1, true>), dim3(grid), dim3(256), 0, s, p);
    else if (nb == 2) hipLaunchKernelGGL((mlp_kernel<2, true>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((mlp_kernel<4, true>), dim3(grid), dim3(256), 0, s, p);
  } else {
    if (nb == 1) hipLaunchKernelGGL((mlp_kernel<1, false>), dim3(grid), dim3(256), 0, s, p);
    else if (nb == 2) hipLaunchKernelGGL((mlp_kernel<2, false>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((mlp_kernel<4, false>), dim3(grid), dim3(256), 0, s, p);
  }
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

extern "C" int pita_mlp_sampler_run(pita_mlp_t* net, float* x, int64_t B, const float* step_tab, int n_steps,
                                    const float* noise, uint64_t seed, uint64_t walker_offset, int64_t step0,
                                    int remove_mean, int n_particles, int n_dim, double* stats_out, void* stream) {
  PITA_REQUIRE(net && x && step_tab && B >= 0 && n_steps >= 0, "pita_mlp_sampler_run: bad argument");
  if (n_steps == 0 || B == 0) return PITA_OK;
  const int D = net->cfg.input_dim;
  PITA_REQUIRE(net->cfg.out_dim == D, "pita_mlp_sampler_run: the score net must map R^D to R^D");
  PITA_REQUIRE(D <= 64, "pita_mlp_sampler_run: input_dim <= 64");
  PITA_REQUIRE(n_particles >= 1 && n_dim >= 1 && n_dim <= 4 && n_particles * n_dim == D,
               "pita_mlp_sampler_run: n_particles * n_dim must equal input_dim (n_dim <= 4)");
  MlpSamplerParams q{};
  q.m = net->p; q.m.B = B;
  q.x = x; q.step_tab = step_tab; q.noise = noise; q.n_steps = n_steps; q.remove_mean = remove_mean;
  q.n_particles = n_particles; q.n_dim = n_dim; q.seed = seed; q.walker_offset = walker_offset; q.step0 = step0;
  q.stats_out = stats_out;
  const long long nblk = ((B + 31) / 32 + 3) / 4;
  const unsigned grid = (unsigned)(nblk < 4096 ? nblk : 4096);
  const size_t lds = sizeof(float) * 4 * 2 * 32 * (size_t)D;
  hipStream_t s = (hipStream_t)stream;
  const int nb = net->cfg.hidden_size / 32;
  if (q.m.stream) {
    if (nb == 1) hipLaunchKernelGGL((mlp_sampler_kernel<1, true>), dim3(grid), dim3(256), lds, s, q);
    else if (nb == 2) hipLaunchKernelGGL((mlp_sampler_kernel<2, true>), dim3(grid), dim3(256), lds, s, q);
    else hipLaunchKernelGGL((mlp_sampler_kernel<4, true>), dim3(grid), dim3(256), lds, s, q);
  } else {
    if (nb == 1) hipLaunchKernelGGL((mlp_sampler_kernel<1, false>), dim3(grid), dim3(256), lds, s, q);
    else if (nb == 2) hipLaunchKernelGGL((mlp_sampler_kernel<2, false>), dim3(grid), dim3(256), lds, s, q);
    else hipLaunchKernelGGL((mlp_sampler_kernel<4, false>), dim3(grid), dim3(256), lds, s, q);
  }
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}
