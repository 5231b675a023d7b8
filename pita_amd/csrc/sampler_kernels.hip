// Elementwise pieces of the sampler and systematic resampling for gfx950.
//
// Replaces (paths relative to /root/reference/pita/src/):
//   models/components/sde_integration.py:347-349   x += drift*dt + diffusion*sqrt(dt)
//   models/components/sdes.py:245-251              diffusion = scale * g(t) * randn_like(x)
//   utils/data_utils.py:4-26                       remove_mean
//   energies/base_prior.py:77-83                   MeanFreePrior.sample
//   models/components/utils.py:111-120             sample_cat_sys (systematic resampling)
//   models/components/sde_integration.py:293       x = x[choice]
//   models/components/sde_integration.py:28-45     mala_proposal
//   models/components/sde_integration.py:379-398, 430-461   MALA accept/reject, step-size adaptation
#include "common.h"

namespace pita {

// One thread per (walker, particle); WB = floor(256/n) walkers per block staged in LDS so the
// per-walker mean is a broadcast read and every global access is one coalesced span.
enum { OP_EM = 0, OP_PRIOR = 1, OP_RMEAN = 2, OP_NORMAL = 3 };

struct ElemParams {
  float dt, noise_scale, sqrt_dt, scale;
  unsigned long long seed, walker_offset;
  long long step;
  int remove_mean;
  const long long* walker_ids;  // optional: Philox walker key of row w is walker_ids[w] instead of walker_offset + w
  double* stats_out;            // OP_EM, nullable [4]: += sum / sum of squares of drift and of noise_scale * xi
};

template <int DIM, int OP>
__global__ void __launch_bounds__(256) elem_kernel(float* __restrict__ x, const float* __restrict__ drift,
                                                   const float* __restrict__ noise, long long B, int n, int WB,
                                                   ElemParams p) {
  extern __shared__ float sm[];  // [WB*n*DIM]
  __shared__ double red[4][4];   // OP_EM moments: one commit per BLOCK at the end of its grid-stride loop (an atomic per
                                 // wave and batch of WB walkers -- 55 000 double atomics on four addresses at 65 536 LJ13
                                 // walkers -- made this the slowest small kernel of a debiased step: 0.67 ms)
  const int tid = threadIdx.x;
  const long long nblk = (B + WB - 1) / WB;
  double momd[4] = {0.0, 0.0, 0.0, 0.0};
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long w0 = blk * WB;
    const int nw = (int)((B - w0) < WB ? (B - w0) : WB);
    const int w = tid / n, i = tid - w * n;
    const bool act = w < nw;
    float v[DIM];
    float mom[4] = {0.f, 0.f, 0.f, 0.f};
    if (act) {
      const long long base = ((w0 + w) * n + i) * DIM;
      float xi[4] = {0.f, 0.f, 0.f, 0.f};
      if (OP == OP_EM || OP == OP_PRIOR || OP == OP_NORMAL) {
        if (noise) {
#pragma unroll
          for (int k = 0; k < DIM; ++k) xi[k] = noise[base + k];
        } else {
          philox_normal4(p.seed, p.walker_offset + (unsigned long long)(w0 + w), p.step, (uint32_t)i, xi);
        }
      }
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        if (OP == OP_EM) {
          const float dr = drift[base + k], dif = p.noise_scale * xi[k];
          v[k] = x[base + k] + (dr * p.dt + (dif * p.sqrt_dt));
          if (p.stats_out) {
            mom[0] += dr; mom[1] = fmaf(dr, dr, mom[1]);
            mom[2] += dif; mom[3] = fmaf(dif, dif, mom[3]);
          }
        }
        if (OP == OP_PRIOR) v[k] = xi[k] * p.scale;
        if (OP == OP_RMEAN) v[k] = x[base + k];
        if (OP == OP_NORMAL) v[k] = xi[k];
        sm[(w * n + i) * DIM + k] = v[k];
      }
    }
    if (OP == OP_EM && p.stats_out) {
#pragma unroll
      for (int q = 0; q < 4; ++q) momd[q] += (double)mom[q];
    }
    __syncthreads();
    if (act) {
      const long long base = ((w0 + w) * n + i) * DIM;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        if (p.remove_mean) {
          float s = 0.f;
          for (int q = 0; q < n; ++q) s += sm[(w * n + q) * DIM + k];
          v[k] -= s / (float)n;
        }
        x[base + k] = v[k];
      }
    }
    __syncthreads();
  }
  if (OP == OP_EM && p.stats_out) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double vq = momd[q];
      for (int o = 32; o > 0; o >>= 1) vq += __shfl_xor(vq, o, 64);
      if ((tid & 63) == 0) red[tid >> 6][q] = vq;
    }
    __syncthreads();
    if (tid < 4) atomicAdd(p.stats_out + tid, (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
  }
}

template <int OP>
static int launch_elem(float* x, const float* drift, const float* noise, int64_t B, int n, int d, const ElemParams& p,
                       void* stream) {
  PITA_REQUIRE(B >= 0, "elementwise: negative batch");
  if (B == 0) return PITA_OK;
  PITA_REQUIRE(x, "elementwise: null argument");
  PITA_REQUIRE(n >= 1 && n <= 256, "elementwise: n_particles must be in [1,256]");
  PITA_REQUIRE(d >= 1 && d <= 3, "elementwise: n_dim must be 1, 2 or 3");
  if (B == 0) return PITA_OK;
  const int WB = 256 / n;
  const long long nblk = (B + WB - 1) / WB;
  // with moments: a grid-stride loop over fewer blocks (one set of four atomics per block)
  const long long cap = (OP == OP_EM && p.stats_out) ? 256LL * 4 : 256LL * 16;
  const unsigned grid = (unsigned)(nblk < cap ? nblk : cap);
  const size_t lds = sizeof(float) * (size_t)(WB * n * d);
  hipStream_t s = (hipStream_t)stream;
  switch (d) {
    case 1: hipLaunchKernelGGL((elem_kernel<1, OP>), dim3(grid), dim3(256), lds, s, x, drift, noise, B, n, WB, p); break;
    case 2: hipLaunchKernelGGL((elem_kernel<2, OP>), dim3(grid), dim3(256), lds, s, x, drift, noise, B, n, WB, p); break;
    default: hipLaunchKernelGGL((elem_kernel<3, OP>), dim3(grid), dim3(256), lds, s, x, drift, noise, B, n, WB, p); break;
  }
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

// sum and sum of squares of a device vector (per-step statistics of SDETerms fields, sde_integration.py:150)
__global__ void __launch_bounds__(256) moments_kernel(const float* __restrict__ v, long long n, double* __restrict__ out) {
  double s = 0.0, s2 = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const double a = (double)v[e];
    s += a;
    s2 += a * a;
  }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(out, s); atomicAdd(out + 1, s2); }
}

// ---------------------------------------------------------------------------- EDM preconditioning around a foreign backbone
// score_net.py:13-43 for backbones that are not fused HIP kernels (any module with forward(t, x, beta)):
//   scale : x_in = c_in x, t_in = c_noise = ln(h)/8                       (inputs of the backbone)
//   combine: D = c_s x + c_out F; optional beta preconditioning (:36-38); score = (D - x)/h
__global__ void __launch_bounds__(256) edm_scale_kernel(const float* __restrict__ h, const float* __restrict__ x,
                                                        float* __restrict__ xs, float* __restrict__ cn, long long B, int D) {
  const long long total = B * D;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long b = e / D;
    const float hv = h[b];
    xs[e] = (1.0f / sqrtf(1.0f + hv)) * x[e];
    if (e - b * D == 0) cn[b] = 0.125f * logf(hv);
  }
}

__global__ void __launch_bounds__(256) edm_combine_kernel(const float* __restrict__ h, const float* __restrict__ x,
                                                          const float* __restrict__ F, const float* __restrict__ beta,
                                                          float* __restrict__ Dout, float* __restrict__ score, long long B,
                                                          int D) {
  const long long total = B * D;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long b = e / D;
    const float hv = h[b], xv = x[e];
    const float c_s = 1.0f / (1.0f + hv), c_in = 1.0f / sqrtf(1.0f + hv), c_out = sqrtf(hv) * c_in;
    float Dv = c_s * xv + c_out * F[e];
    float sc = (Dv - xv) / hv;
    if (beta) {
      const float bt = beta[b];
      Dv = Dv * bt + (1.0f - bt) * xv;
      sc = sc * bt;
    }
    if (Dout) Dout[e] = Dv;
    if (score) score[e] = sc;
  }
}

// ---------------------------------------------------------------------------- MALA
// The step size lives on the device (double, like the Python float it replaces) so the adaptive chain
// runs without a host round trip per step; the kernels round it to fp32 where torch would.
template <int DIM>
__global__ void __launch_bounds__(256) mala_propose_kernel(const float* __restrict__ x, const float* __restrict__ force,
                                                           float* __restrict__ x_prop, const float* __restrict__ noise,
                                                           long long B, int n, const double* __restrict__ dt_dev,
                                                           ElemParams p) {
  const float hdt = (float)(0.5 * dt_dev[0]), sdt = (float)sqrt(dt_dev[0]);
  const long long total = B * n;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long w = t / n;
    const int i = (int)(t - w * n);
    const long long base = t * DIM;
    float xi[4] = {0.f, 0.f, 0.f, 0.f};
    if (noise) {
#pragma unroll
      for (int k = 0; k < DIM; ++k) xi[k] = noise[base + k];
    } else {
      philox_normal4(p.seed, p.walker_ids ? (unsigned long long)p.walker_ids[w] : p.walker_offset + (unsigned long long)w,
                     p.step, (uint32_t)i, xi);
    }
#pragma unroll
    for (int k = 0; k < DIM; ++k) x_prop[base + k] = (x[base + k] + hdt * force[base + k]) + sdt * xi[k];
  }
}

template <int DIM>
__global__ void __launch_bounds__(256) mala_accept_kernel(float* __restrict__ x, float* __restrict__ logp,
                                                          const float* __restrict__ force, const float* __restrict__ x_prop,
                                                          const float* __restrict__ logp_prop,
                                                          const float* __restrict__ force_prop,
                                                          const float* __restrict__ uniforms, long long B, int n, int WB,
                                                          const double* __restrict__ dt_dev, int* __restrict__ acc_count,
                                                          ElemParams p) {
  extern __shared__ float sm[];
  float* xsel = sm;                      // [WB*n*DIM] accepted/kept coordinates
  float* qf = sm + WB * n * DIM;         // [WB*n] partial |x' - fwd_mean|^2
  float* qb = qf + WB * n;               // [WB*n] partial |x - bwd_mean|^2
  float* flag = qb + WB * n;             // [WB]   1.0 = accepted
  const float hdt = (float)(0.5 * dt_dev[0]), tdt = (float)(2.0 * dt_dev[0]);
  const int tid = threadIdx.x;
  const long long nblk = (B + WB - 1) / WB;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long w0 = blk * WB;
    const int nw = (int)((B - w0) < WB ? (B - w0) : WB);
    const int w = tid / n, i = tid - w * n;
    const bool act = w < nw;
    const long long base = ((w0 + w) * n + i) * DIM;
    float xo[DIM], xp[DIM];
    if (act) {
      float sf = 0.f, sb = 0.f;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        xo[k] = x[base + k];
        xp[k] = x_prop[base + k];
        const float df = xp[k] - (xo[k] + hdt * force[base + k]);
        const float db = xo[k] - (xp[k] + hdt * force_prop[base + k]);
        sf += df * df;
        sb += db * db;
      }
      qf[w * n + i] = sf;
      qb[w * n + i] = sb;
    }
    __syncthreads();
    if (act && i == 0) {
      float sf = 0.f, sb = 0.f;
      for (int j = 0; j < n; ++j) { sf += qf[w * n + j]; sb += qb[w * n + j]; }
      const float lqf = -sf / tdt, lqb = -sb / tdt;
      const float lp = logp[w0 + w], lpp = logp_prop[w0 + w];
      const float ratio = (lpp - lp) + (lqb - lqf);
      const float u = uniforms ? uniforms[w0 + w]
                               : philox_uniform(p.seed, p.walker_ids ? (unsigned long long)p.walker_ids[w0 + w]
                                                                     : p.walker_offset + (unsigned long long)(w0 + w),
                                                p.step, 0xFFFFFu);
      const float af = (logf(u) < ratio) ? 1.0f : 0.0f;
      logp[w0 + w] = af * lpp + (1.0f - af) * lp;
      flag[w] = af;
    }
    __syncthreads();
    float v[DIM];
    if (act) {
      const float af = flag[w];
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        v[k] = af * xp[k] + (1.0f - af) * xo[k];
        xsel[(w * n + i) * DIM + k] = v[k];
      }
    }
    if (tid == 0) {
      int c = 0;
      for (int j = 0; j < nw; ++j) c += flag[j] != 0.f;
      if (c) atomicAdd(acc_count, c);
    }
    __syncthreads();
    if (act) {
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        if (p.remove_mean) {
          float s = 0.f;
          for (int j = 0; j < n; ++j) s += xsel[(w * n + j) * DIM + k];
          v[k] -= s / (float)n;
        }
        x[base + k] = v[k];
      }
    }
    __syncthreads();
  }
}

__global__ void mala_adapt_kernel(double* dt_dev, int* acc_count, long long total, int adaptive, float* rate_out) {
  const float rate = (float)acc_count[0] / (float)total;
  if (rate_out) rate_out[0] = rate;
  if (adaptive) dt_dev[0] = ((double)rate > 0.55) ? dt_dev[0] * 1.1 : dt_dev[0] / 1.1;  // sde_integration.py:439-443
  acc_count[0] = 0;
}

// ---------------------------------------------------------------------------- resampling
// Systematic resampling (utils.py:111-120) as five short multi-block passes over the logits, all in a fixed
// summation order (bitwise reproducible), 1 024 walkers per block:
//   A  block maxima                      B  block sums of exp(l - max) in double
//   C  block sums of the clipped weights clip(exp(l - max) / sum, 1e-6, 1) in double
//   D  bins = inclusive cumsum: blocks' prefix (sequential over the block totals) + in-block scan, accumulated in
//      double and rounded to fp32 per prefix -- torch's CPU cumsum semantics (float input, double accumulator)
//   E  ids[k] = #bins < u_k (digitize right=True), one thread per k, binary search
// exp is evaluated in double and rounded once: the correctly rounded fp32 exponential, which is what torch-CPU's
// (SLEEF, <= 1 ulp) expf returns in all but a vanishing fraction of arguments; the ocml expf differs from it far
// more often and each difference can move an id whose uniform lies within an ulp of a bin edge.
constexpr int RS_T = 256, RS_E = 1024;  // threads and elements per block

__device__ __forceinline__ float rs_exp(float v) { return (float)exp((double)v); }

__device__ __forceinline__ double block_sum_d(double v, double* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  for (int k = 0; k < RS_T / 64; ++k) r += red[k];
  return r;
}

__global__ void __launch_bounds__(RS_T) rs_max_kernel(const float* __restrict__ logits, long long B, float* __restrict__ bmax) {
  __shared__ float red[RS_T / 64];
  const long long lo = (long long)blockIdx.x * RS_E;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < RS_E; i += RS_T)
    if (lo + i < B) mx = fmaxf(mx, logits[lo + i]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < RS_T / 64; ++k) mx = fmaxf(mx, red[k]);
    bmax[blockIdx.x] = mx;
  }
}

// phase 0: sums of exp(l - max);  phase 1: sums of the clipped weights
__global__ void __launch_bounds__(RS_T) rs_sum_kernel(const float* __restrict__ logits, long long B, int nblk,
                                                      const float* __restrict__ bmax, const double* __restrict__ bsum,
                                                      double* __restrict__ out, int phase) {
  __shared__ double red[RS_T / 64];
  float mx = -INFINITY;
  for (int k = 0; k < nblk; ++k) mx = fmaxf(mx, bmax[k]);
  float sm = 1.0f;
  if (phase == 1) {
    double t = 0.0;
    for (int k = 0; k < nblk; ++k) t += bsum[k];
    sm = (float)t;
  }
  const long long lo = (long long)blockIdx.x * RS_E;
  double acc = 0.0;
  for (int i = threadIdx.x; i < RS_E; i += RS_T)
    if (lo + i < B) {
      float w = rs_exp(logits[lo + i] - mx);
      if (phase == 1) w = fminf(fmaxf(w / sm, 1e-6f), 1.0f);
      acc += (double)w;
    }
  acc = block_sum_d(acc, red);
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(RS_T) rs_bins_kernel(const float* __restrict__ logits, long long B, int nblk,
                                                       const float* __restrict__ bmax, const double* __restrict__ bsum,
                                                       const double* __restrict__ bw, float* __restrict__ bins) {
  __shared__ double part[RS_T];
  float mx = -INFINITY;
  for (int k = 0; k < nblk; ++k) mx = fmaxf(mx, bmax[k]);
  double t = 0.0;
  for (int k = 0; k < nblk; ++k) t += bsum[k];
  const float sm = (float)t;
  double base = 0.0;
  for (int k = 0; k < (int)blockIdx.x; ++k) base += bw[k];
  constexpr int PER = RS_E / RS_T;  // contiguous elements per thread
  const long long lo = (long long)blockIdx.x * RS_E + (long long)threadIdx.x * PER;
  float w[PER];
  double acc = 0.0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    w[q] = (lo + q < B) ? fminf(fmaxf(rs_exp(logits[lo + q] - mx) / sm, 1e-6f), 1.0f) : 0.0f;
    acc += (double)w[q];
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double run = 0.0;
    for (int k = 0; k < RS_T; ++k) { const double v = part[k]; part[k] = run; run += v; }
  }
  __syncthreads();
  acc = base + part[threadIdx.x];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    acc += (double)w[q];
    if (lo + q < B) bins[lo + q] = (float)acc;
  }
}

__global__ void __launch_bounds__(256) rs_search_kernel(const float* __restrict__ bins, long long B, double u0,
                                                        long long* __restrict__ ids) {
  // u_k = (u0 + fp32(k * fp32(1/B))) mod 1 in double; ids = #bins < u  (digitize right=True), clamp
  const float invB = (float)(1.0 / (double)B);
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < B; k += (long long)gridDim.x * 256) {
    const double u = fmod(u0 + (double)((float)k * invB), 1.0);
    long long a = 0, b = B;  // first index with bins[idx] >= u
    while (a < b) {
      const long long m = (a + b) >> 1;
      if ((double)bins[m] < u) a = m + 1; else b = m;
    }
    ids[k] = (a >= B) ? B - 1 : a;
  }
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ src, const long long* __restrict__ ids,
                                                          float* __restrict__ out, long long B, int D) {
  const long long total = B * D;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / D;
    out[e] = src[ids[r] * D + (e - r * D)];
  }
}

// distinct parents of a systematic resampling = cyclic runs of the id vector (ids are non-decreasing up to the rotation by the
// event's uniform): one block, integer sums -- replaces roll + != + cast + sum + clamp (five launches) behind every event
__global__ void __launch_bounds__(1024) count_runs_kernel(const long long* __restrict__ ids, long long B,
                                                          long long* __restrict__ out) {
  __shared__ unsigned part[16];
  unsigned c = 0;
  for (long long i = threadIdx.x; i < B; i += 1024) c += ids[i] != ids[i == 0 ? B - 1 : i - 1] ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int w = 0; w < 16; ++w) t += part[w];
    *out = t < 1u ? 1 : (long long)t;
  }
}

}  // namespace pita

using namespace pita;

extern "C" int pita_count_runs(const int64_t* ids, int64_t B, int64_t* out, void* stream) {
  PITA_REQUIRE(ids && out && B >= 1, "pita_count_runs: bad argument");
  hipLaunchKernelGGL(count_runs_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const long long*)ids, (long long)B,
                     (long long*)out);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

extern "C" int pita_em_step(float* x, const float* drift, const float* noise, int64_t B, int n, int d, float dt,
                            float noise_scale, float sqrt_dt, uint64_t seed, uint64_t walker_offset, int64_t step,
                            int remove_mean, double* stats_out, void* stream) {
  PITA_REQUIRE(drift, "pita_em_step: drift is null");
  ElemParams p{};
  p.dt = dt; p.noise_scale = noise_scale; p.sqrt_dt = sqrt_dt; p.seed = seed; p.walker_offset = walker_offset;
  p.step = step; p.remove_mean = remove_mean; p.stats_out = stats_out;
  return launch_elem<OP_EM>(x, drift, noise, B, n, d, p, stream);
}

// the same for up to four vectors of one length in ONE launch: out[2 q], out[2 q + 1] += sum / sum of squares of v_q
// (the four per-step statistics of the debiased regime were four launches of ~27 us)
__global__ void __launch_bounds__(256) moments4_kernel(const float* __restrict__ v0, const float* __restrict__ v1,
                                                       const float* __restrict__ v2, const float* __restrict__ v3,
                                                       long long n, double* __restrict__ out) {
  const float* vs[4] = {v0, v1, v2, v3};
  __shared__ double red[4][8];
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (vs[q]) {
        const double a = (double)vs[q][e];
        acc[2 * q] += a;
        acc[2 * q + 1] += a * a;
      }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    double t = acc[q];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = t;
  }
  __syncthreads();
  if (threadIdx.x < 8 && vs[threadIdx.x >> 1])
    atomicAdd(out + threadIdx.x, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

extern "C" int pita_moments4(const float* v0, const float* v1, const float* v2, const float* v3, int64_t n, double* out,
                             void* stream) {
  PITA_REQUIRE(n >= 0 && out, "pita_moments4: bad argument");
  if (n == 0 || !(v0 || v1 || v2 || v3)) return PITA_OK;
  const long long nb = (n + 255) / 256;
  hipLaunchKernelGGL(moments4_kernel, dim3((unsigned)(nb < 256 ? nb : 256)), dim3(256), 0, (hipStream_t)stream, v0, v1, v2, v3,
                     (long long)n, out);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

// ---- histogram of a sample over given bin edges (the reference's evaluation figure: base_molecule_energy_function.py:160-254,
// matplotlib `hist` = numpy.histogram): counts[i] = #{ edges[i] <= v < edges[i+1] }, the last bin closed on the right,
// values outside [edges[0], edges[nbins]] and NaNs not counted.  One pass over the sample: bins privatised in LDS per
// block (one copy per wave: the four waves of a block do not contend), a guessed bin from the uniform spacing corrected
// against the edges (numpy's own rule, so that counts agree with numpy.histogram bin for bin), one 64-bit add per bin
// and block at the end.
constexpr int HIST_MAX_BINS = 1024;
__global__ void __launch_bounds__(256) histogram_kernel(const float* __restrict__ v, long long n, const float* __restrict__ edges,
                                                        int nbins, unsigned long long* __restrict__ counts) {
  extern __shared__ unsigned hist_lds[];  // [4 waves][nbins] counters, then [nbins + 1] edges
  unsigned* bins = hist_lds + (threadIdx.x >> 6) * nbins;
  float* e = reinterpret_cast<float*>(hist_lds + 4 * nbins);
  for (int i = threadIdx.x; i < 4 * nbins; i += 256) hist_lds[i] = 0u;
  for (int i = threadIdx.x; i <= nbins; i += 256) e[i] = edges[i];
  __syncthreads();
  const float lo = e[0], hi = e[nbins];
  const float inv = (float)nbins / (hi - lo);
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256) {
    const float x = v[k];
    if (!(x >= lo && x <= hi)) continue;  // out of range or NaN
    int i = (int)((x - lo) * inv);
    i = i < 0 ? 0 : (i > nbins - 1 ? nbins - 1 : i);
    while (i > 0 && x < e[i]) --i;
    while (i < nbins - 1 && x >= e[i + 1]) ++i;
    atomicAdd(&bins[i], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nbins; i += 256) {
    const unsigned long long t = (unsigned long long)hist_lds[i] + hist_lds[nbins + i] + hist_lds[2 * nbins + i] + hist_lds[3 * nbins + i];
    if (t) atomicAdd(counts + i, t);
  }
}

extern "C" int pita_histogram(const float* v, int64_t n, const float* edges, int nbins, unsigned long long* counts, void* stream) {
  PITA_REQUIRE(n >= 0 && edges && counts && nbins >= 1 && nbins <= HIST_MAX_BINS, "pita_histogram: bad argument (1 <= nbins <= %d)",
               HIST_MAX_BINS);
  PITA_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(unsigned long long) * (size_t)nbins, (hipStream_t)stream));
  if (n == 0) return PITA_OK;
  PITA_REQUIRE(v, "pita_histogram: null input");
  const long long nb = (n + 255) / 256;
  const size_t lds = sizeof(unsigned) * 4 * (size_t)nbins + sizeof(float) * ((size_t)nbins + 1);
  hipLaunchKernelGGL(histogram_kernel, dim3((unsigned)(nb < 1024 ? nb : 1024)), dim3(256), lds, (hipStream_t)stream, v,
                     (long long)n, edges, nbins, counts);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

extern "C" int pita_moments(const float* v, int64_t n, double* out, void* stream) {
  PITA_REQUIRE(n >= 0 && out, "pita_moments: bad argument");
  if (n == 0) return PITA_OK;
  PITA_REQUIRE(v, "pita_moments: null input");
  const long long nb = (n + 255) / 256;
  hipLaunchKernelGGL(moments_kernel, dim3((unsigned)(nb < 1024 ? nb : 1024)), dim3(256), 0, (hipStream_t)stream, v,
                     (long long)n, out);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

extern "C" int pita_prior_sample(float* x, const float* noise, int64_t B, int n, int d, float scale, uint64_t seed,
                                 uint64_t walker_offset, int mean_free, void* stream) {
  ElemParams p{};
  p.scale = scale; p.seed = seed; p.walker_offset = walker_offset; p.step = -1; p.remove_mean = mean_free;
  return launch_elem<OP_PRIOR>(x, nullptr, noise, B, n, d, p, stream);
}

extern "C" int pita_remove_mean(float* x, int64_t B, int n, int d, void* stream) {
  ElemParams p{};
  p.remove_mean = 1;
  return launch_elem<OP_RMEAN>(x, nullptr, nullptr, B, n, d, p, stream);
}

extern "C" int pita_fill_normal(float* out, int64_t B, int n, int d, uint64_t seed, uint64_t walker_offset,
                                int64_t step, void* stream) {
  ElemParams p{};
  p.seed = seed; p.walker_offset = walker_offset; p.step = step; p.remove_mean = 0;
  return launch_elem<OP_NORMAL>(out, nullptr, nullptr, B, n, d, p, stream);
}

static inline long long rs_nblk(int64_t B) { return (B + RS_E - 1) / RS_E; }

// workspace: float bins[B] | float bmax[nblk] | double bsum[nblk] | double bw[nblk]   (8-byte aligned sections)
extern "C" size_t pita_resample_workspace_bytes(int64_t B) {
  const size_t n = (size_t)(B > 0 ? B : 1), nb = (size_t)rs_nblk(B > 0 ? B : 1);
  return ((n * 4 + 7) & ~(size_t)7) + ((nb * 4 + 7) & ~(size_t)7) + 2 * nb * 8;
}

extern "C" int pita_systematic_resample(const float* logits, int64_t B, double u0, int64_t* ids, void* workspace,
                                        void* stream) {
  PITA_REQUIRE(logits && ids && workspace && B >= 0, "pita_systematic_resample: null argument");
  if (B == 0) return PITA_OK;
  PITA_REQUIRE(((uintptr_t)workspace & 7) == 0, "pita_systematic_resample: workspace must be 8-byte aligned");
  const long long nb = rs_nblk(B);
  char* w = static_cast<char*>(workspace);
  float* bins = reinterpret_cast<float*>(w);
  w += ((size_t)B * 4 + 7) & ~(size_t)7;
  float* bmax = reinterpret_cast<float*>(w);
  w += ((size_t)nb * 4 + 7) & ~(size_t)7;
  double* bsum = reinterpret_cast<double*>(w);
  double* bw = bsum + nb;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(rs_max_kernel, dim3((unsigned)nb), dim3(RS_T), 0, s, logits, (long long)B, bmax);
  hipLaunchKernelGGL(rs_sum_kernel, dim3((unsigned)nb), dim3(RS_T), 0, s, logits, (long long)B, (int)nb, bmax, bsum, bsum, 0);
  hipLaunchKernelGGL(rs_sum_kernel, dim3((unsigned)nb), dim3(RS_T), 0, s, logits, (long long)B, (int)nb, bmax, bsum, bw, 1);
  hipLaunchKernelGGL(rs_bins_kernel, dim3((unsigned)nb), dim3(RS_T), 0, s, logits, (long long)B, (int)nb, bmax, bsum, bw, bins);
  const long long sb = (B + 255) / 256;
  hipLaunchKernelGGL(rs_search_kernel, dim3((unsigned)(sb < 8192 ? sb : 8192)), dim3(256), 0, s, bins, (long long)B, u0,
                     (long long*)ids);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

extern "C" int pita_gather_rows(const float* src, const int64_t* ids, float* out, int64_t B, int D, void* stream) {
  PITA_REQUIRE(src && ids && out && B >= 0 && D >= 1, "pita_gather_rows: bad argument");
  PITA_REQUIRE(src != out, "pita_gather_rows: in-place gather is not supported");
  if (B == 0) return PITA_OK;
  const long long nb = (B * D + 255) / 256;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, (hipStream_t)stream, src,
                     (const long long*)ids, out, (long long)B, D);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

extern "C" int pita_mala_propose(const float* x, const float* force, float* x_prop, const float* noise, int64_t B, int n,
                                 int d, const double* dt_dev, uint64_t seed, uint64_t walker_offset,
                                 const int64_t* walker_ids, int64_t step, void* stream) {
  PITA_REQUIRE(B >= 0, "pita_mala_propose: negative batch");
  if (B == 0) return PITA_OK;
  PITA_REQUIRE(x && force && x_prop && dt_dev, "pita_mala_propose: null argument");
  PITA_REQUIRE(n >= 1 && n <= 256 && d >= 1 && d <= 3, "pita_mala_propose: n_particles in [1,256], n_dim in [1,3]");
  ElemParams p{};
  p.seed = seed; p.walker_offset = walker_offset; p.step = step; p.walker_ids = (const long long*)walker_ids;
  const long long nb = (B * n + 255) / 256;
  const unsigned grid = (unsigned)(nb < 8192 ? nb : 8192);
  hipStream_t s = (hipStream_t)stream;
  switch (d) {
    case 1: hipLaunchKernelGGL(mala_propose_kernel<1>, dim3(grid), dim3(256), 0, s, x, force, x_prop, noise, B, n, dt_dev, p); break;
    case 2: hipLaunchKernelGGL(mala_propose_kernel<2>, dim3(grid), dim3(256), 0, s, x, force, x_prop, noise, B, n, dt_dev, p); break;
    default: hipLaunchKernelGGL(mala_propose_kernel<3>, dim3(grid), dim3(256), 0, s, x, force, x_prop, noise, B, n, dt_dev, p); break;
  }
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

extern "C" int pita_mala_accept(float* x, float* logp, const float* force, const float* x_prop, const float* logp_prop,
                                const float* force_prop, const float* uniforms, int64_t B, int n, int d,
                                const double* dt_dev, uint64_t seed, uint64_t walker_offset, const int64_t* walker_ids,
                                int64_t step, int remove_mean, int* acc_count, void* stream) {
  PITA_REQUIRE(B >= 0, "pita_mala_accept: negative batch");
  if (B == 0) return PITA_OK;
  PITA_REQUIRE(x && logp && force && x_prop && logp_prop && force_prop && dt_dev && acc_count,
               "pita_mala_accept: null argument");
  PITA_REQUIRE(n >= 1 && n <= 256 && d >= 1 && d <= 3, "pita_mala_accept: n_particles in [1,256], n_dim in [1,3]");
  ElemParams p{};
  p.seed = seed; p.walker_offset = walker_offset; p.step = step; p.remove_mean = remove_mean;
  p.walker_ids = (const long long*)walker_ids;
  const int WB = 256 / n;
  const long long nblk = (B + WB - 1) / WB;
  const unsigned grid = (unsigned)(nblk < 256LL * 16 ? nblk : 256LL * 16);
  const size_t lds = sizeof(float) * (size_t)(WB * n * d + 2 * WB * n + WB);
  hipStream_t s = (hipStream_t)stream;
  switch (d) {
    case 1: hipLaunchKernelGGL(mala_accept_kernel<1>, dim3(grid), dim3(256), lds, s, x, logp, force, x_prop, logp_prop, force_prop, uniforms, B, n, WB, dt_dev, acc_count, p); break;
    case 2: hipLaunchKernelGGL(mala_accept_kernel<2>, dim3(grid), dim3(256), lds, s, x, logp, force, x_prop, logp_prop, force_prop, uniforms, B, n, WB, dt_dev, acc_count, p); break;
    default: hipLaunchKernelGGL(mala_accept_kernel<3>, dim3(grid), dim3(256), lds, s, x, logp, force, x_prop, logp_prop, force_prop, uniforms, B, n, WB, dt_dev, acc_count, p); break;
  }
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

extern "C" int pita_mala_adapt(double* dt_dev, int* acc_count, int64_t total, int adaptive, float* rate_out, void* stream) {
  PITA_REQUIRE(dt_dev && acc_count && total > 0, "pita_mala_adapt: bad argument");
  hipLaunchKernelGGL(mala_adapt_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dt_dev, acc_count, (long long)total,
                     adaptive, rate_out);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

extern "C" int pita_edm_scale_input(const float* h, const float* x, float* x_scaled, float* c_noise, int64_t B, int D,
                                    void* stream) {
  PITA_REQUIRE(B >= 0 && D >= 1, "pita_edm_scale_input: bad shape");
  if (B == 0) return PITA_OK;
  PITA_REQUIRE(h && x && x_scaled && c_noise, "pita_edm_scale_input: null argument");
  const long long nb = (B * D + 255) / 256;
  hipLaunchKernelGGL(edm_scale_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, (hipStream_t)stream, h, x,
                     x_scaled, c_noise, (long long)B, D);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

extern "C" int pita_edm_combine(const float* h, const float* x, const float* F, const float* beta, float* D_out,
                                float* score_out, int64_t B, int D, void* stream) {
  PITA_REQUIRE(B >= 0 && D >= 1, "pita_edm_combine: bad shape");
  if (B == 0) return PITA_OK;
  PITA_REQUIRE(h && x && F && (D_out || score_out), "pita_edm_combine: null argument");
  const long long nb = (B * D + 255) / 256;
  hipLaunchKernelGGL(edm_combine_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, (hipStream_t)stream, h, x, F,
                     beta, D_out, score_out, (long long)B, D);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}
