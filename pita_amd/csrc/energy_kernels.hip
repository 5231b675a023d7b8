// Target Boltzmann log-densities and forces for gfx950.
//
// Replaces (paths relative to /root/reference/):
//   pita/src/energies/lennardjones_energy.py:34-36,121-155,213-227  (+ bgflow distance_vectors /
//       distances_from_vectors, an un-vendored dependency: r = sqrt(|dx|^2 + 1e-6), ordered pairs)
//   pita/src/energies/gmm_energy.py:87-90 -> fab/fab/target_distributions/gmm.py:71-79,104
//   DW4: bgflow.MultiDoubleWellPotential (not in the reference tree, see oracle header)
//
// Layout / mapping: x is the reference's [B, n*d] row-major tensor.  A 256-thread block
// stages WB = floor(256/n) consecutive walkers (one contiguous, fully coalesced span of
// WB*n*d floats) in LDS; thread (w,i) owns particle i of walker w and sweeps the partners j
// from LDS (broadcast-friendly: the n lanes of a walker read the same address).  Forces are
// written back through LDS so the global store is again one coalesced span.  Pair energies are
// HBM-bound for LJ13/DW4 (316 / 68 algorithmic bytes per walker-eval).
#include <cstdlib>

#include "pair_common.h"

namespace pita {

// pair energy and e'(r)/r of the smooth-core LJ (lennardjones_energy.py:39-54,131-133): the reference evaluates
// `lj * ~filter + filter * spline(r)`, the spline clamped to its first interval for r < range_min -- one cubic in
// (r - range_min), same operation order (c0 dx**3 + c1 dx**2 + c2 dx + c3)
__device__ __forceinline__ void lj_smooth_pair(float r2, const PairParams& p, float& e, float& coef) {
  const float inv = __builtin_amdgcn_rcpf(r2);
  if (r2 < 1.001f * p.sm_min * p.sm_min) {  // cheap pre-test; the reference's filter is the exact `r < range_min` below
    const float r = sqrtf(r2);
    if (r < p.sm_min) {
      const float u = r - p.sm_min, u2 = u * u, u3 = u2 * u;
      e = ((p.sc0 * u3 + p.sc1 * u2) + p.sc2 * u) + p.sc3;
      coef = (fmaf(3.0f * p.sc0, u2, fmaf(2.0f * p.sc1, u, p.sc2))) / r;
      return;
    }
  }
  const float s2 = p.rm2 * inv, s6 = s2 * s2 * s2;
  e = p.eps * fmaf(s6, s6, -2.0f * s6);
  coef = p.eps * (12.0f * fmaf(-s6, s6, s6)) * inv;
}

// logp-force of particle i of one walker whose coordinates xw[n*DIM] sit in LDS: f = d logp / d x_i,
// e = this particle's share of the energy (ordered pairs for LJ, half of each unordered pair for DW).
template <int DIM, int KIND>
__device__ __forceinline__ void pair_force(const float* xw, int i, int n, const PairParams& p, float (&f)[DIM], float& e) {
  float xi[DIM], mean[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) { f[k] = 0.f; mean[k] = 0.f; xi[k] = xw[i * DIM + k]; }
  e = 0.f;
  for (int j = 0; j < n; ++j) {
    float d[DIM], r2 = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      const float xj = xw[j * DIM + k];
      mean[k] += xj;
      d[k] = xi[k] - xj;
      r2 = fmaf(d[k], d[k], r2);
    }
    if (j == i) continue;
    if (KIND == E_LJ) {
      r2 += p.dist_eps;
      const float inv = __builtin_amdgcn_rcpf(r2);
      const float s2 = p.rm2 * inv, s6 = s2 * s2 * s2;
      const float t6 = fmaf(s6, s6, -2.0f * s6);  // s^12 - 2 s^6
      e = fmaf(p.eps, t6, e);
      const float coef = p.eps * (12.0f * fmaf(-s6, s6, s6)) * inv;  // e'(r)/r = eps*12*(s^6 - s^12)/r^2
#pragma unroll
      for (int k = 0; k < DIM; ++k) f[k] = fmaf(coef, d[k], f[k]);
    } else if (KIND == E_LJS) {
      float ep, coef;
      lj_smooth_pair(r2 + p.dist_eps, p, ep, coef);
      e += ep;
#pragma unroll
      for (int k = 0; k < DIM; ++k) f[k] = fmaf(coef, d[k], f[k]);
    } else {
      const float dist = sqrtf(r2);
      const float u = dist - p.d0, u2 = u * u;
      e += fmaf(p.a * u2, u2, fmaf(p.b, u2, p.c));
      const float coef = (u * fmaf(4.0f * p.a, u2, 2.0f * p.b)) / dist;
#pragma unroll
      for (int k = 0; k < DIM; ++k) f[k] = fmaf(coef, d[k], f[k]);
    }
  }
  if (is_lj<KIND>()) {
    // E = ef * sum_{i != j} lj + 0.5*osc*sum |x - mean|^2 ; each unordered pair appears twice
    float osc = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      const float c = xi[k] - mean[k] / (float)n;
      osc += c * c;
      f[k] = -p.inv_T * (2.0f * p.energy_factor * f[k] + p.osc_scale * c);
    }
    e = p.energy_factor * e + 0.5f * p.osc_scale * osc;
  } else {
    e *= 0.5f;  // unordered pairs once
#pragma unroll
    for (int k = 0; k < DIM; ++k) f[k] = -p.inv_T * f[k];
  }
}

template <int DIM, int KIND>
__global__ void __launch_bounds__(256) pair_energy_kernel(const float* __restrict__ x, float* __restrict__ logp,
                                                          float* __restrict__ force, long long B, int n, int WB,
                                                          PairParams p) {
  extern __shared__ float sm[];
  float* xs = sm;                   // [WB*n*DIM] coordinates, later reused for forces
  float* es = sm + WB * n * DIM;    // [WB*n] per-particle energy partials
  const int tid = threadIdx.x;
  const long long nblk = (B + WB - 1) / WB;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long w0 = blk * WB;
    const int nw = (int)((B - w0) < WB ? (B - w0) : WB);
    const int nfl = nw * n * DIM;
    const float* src = x + w0 * n * DIM;
    for (int i = tid; i < nfl; i += 256) xs[i] = src[i];
    __syncthreads();
    const int w = tid / n, i = tid - w * n;
    const bool act = w < nw;
    float f[DIM], e = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) f[k] = 0.f;
    if (act) {
      pair_force<DIM, KIND>(xs + w * n * DIM, i, n, p, f, e);
      es[w * n + i] = e;
    }
    __syncthreads();  // all reads of xs done; es complete
    if (act) {
      if (force) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) xs[(w * n + i) * DIM + k] = f[k];
      }
      if (i == 0) {
        float s = 0.f;
        for (int q = 0; q < n; ++q) s += es[w * n + q];
        logp[w0 + w] = -s * p.inv_T;
      }
    }
    __syncthreads();
    if (force) {
      float* dst = force + w0 * n * DIM;
      for (int q = tid; q < nfl; q += 256) dst[q] = xs[q];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------- generic kernel with Newton's third law
// For n <= 64 a walker's particles are the lanes of ONE wavefront (floor(64/n) walkers per wave): every unordered pair
// is evaluated once, as the circulant (i, i + dd mod n), dd = 1 .. (n-1)/2 (+ half of dd = n/2 for even n); lane i keeps
// its own force in registers and hands -f to partner j through an LDS table -- for a fixed dd the map i -> j is a
// permutation and a wave's LDS operations execute in order, so the read-add-write needs no atomics and the summation
// order is fixed.  Halves the pair arithmetic of the ordered-pair kernel above (LJ55: 1 485 instead of 2 970 pairs).
__device__ __forceinline__ void pair_wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// per-dimension particle mean of one walker: lanes i < DIM sum, everyone reads (ms: DIM floats of this walker)
template <int DIM>
__device__ __forceinline__ void walker_mean(const float* xw, float* ms, int i, int n, float (&mean)[DIM]) {
  if (i < DIM) {
    float s = 0.f;
    for (int j = 0; j < n; ++j) s += xw[j * DIM + i];
    ms[i] = s;
  }
  pair_wave_fence();
#pragma unroll
  for (int k = 0; k < DIM; ++k) mean[k] = ms[k];
}

// f = d logp / d x_i, e = this lane's share of the energy.  fw: [n*DIM] LDS, this walker's partner-force table.
template <int DIM, int KIND>
__device__ __forceinline__ void pair_force_n3l(const float* xw, float* fw, float* ms, int i, int n, const PairParams& p,
                                               float (&f)[DIM], float& e) {
  float xi[DIM], mean[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) { f[k] = 0.f; xi[k] = xw[i * DIM + k]; fw[i * DIM + k] = 0.f; }
  if (is_lj<KIND>()) walker_mean<DIM>(xw, ms, i, n, mean);
  e = 0.f;
  const int nh = (n - 1) >> 1, npass = nh + ((n & 1) ? 0 : 1);
  for (int dd = 1; dd <= npass; ++dd) {
    asm volatile("" ::: "memory");
    const bool on = dd <= nh || i < (n >> 1);  // the antipodal distance of an even ring: each pair once
    int j = i + dd;
    j = (j >= n) ? j - n : j;
    float d[DIM], r2 = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      d[k] = xi[k] - xw[j * DIM + k];
      r2 = fmaf(d[k], d[k], r2);
    }
    float coef, ep;
    if (KIND == E_LJ) {
      r2 += p.dist_eps;
      const float inv = __builtin_amdgcn_rcpf(r2);
      const float s2 = p.rm2 * inv, s6 = s2 * s2 * s2;
      ep = 2.0f * (p.eps * fmaf(s6, s6, -2.0f * s6));              // the reference sums ordered pairs
      coef = p.eps * (12.0f * fmaf(-s6, s6, s6)) * inv;             // e'(r)/r = eps*12*(s^6 - s^12)/r^2
    } else if (KIND == E_LJS) {
      lj_smooth_pair(r2 + p.dist_eps, p, ep, coef);
      ep *= 2.0f;
    } else {
      const float dist = sqrtf(r2);
      const float u = dist - p.d0, u2 = u * u;
      ep = fmaf(p.a * u2, u2, fmaf(p.b, u2, p.c));
      coef = (u * fmaf(4.0f * p.a, u2, 2.0f * p.b)) / dist;
    }
    if (on) {
      e += ep;
#pragma unroll
      for (int k = 0; k < DIM; ++k) {
        const float v = coef * d[k];
        f[k] += v;
        fw[j * DIM + k] -= v;
      }
    }
  }
  pair_wave_fence();
#pragma unroll
  for (int k = 0; k < DIM; ++k) f[k] += fw[i * DIM + k];
  if (is_lj<KIND>()) {
    // E = ef * sum_{i != j} lj + 0.5*osc*sum |x - mean|^2
    float osc = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      const float c = xi[k] - mean[k] / (float)n;
      osc += c * c;
      f[k] = -p.inv_T * (2.0f * p.energy_factor * f[k] + p.osc_scale * c);
    }
    e = p.energy_factor * e + 0.5f * p.osc_scale * osc;
  } else {
#pragma unroll
    for (int k = 0; k < DIM; ++k) f[k] = -p.inv_T * f[k];
  }
}

struct WaveMap {  // thread -> (walker of the block, particle) with whole walkers per wavefront
  int w, i;
  bool in_range;
  __device__ WaveMap(int tid, int n) {
    const int wpw = 64 / n, lane = tid & 63, wl = lane / n;
    i = lane - wl * n;
    w = (tid >> 6) * wpw + wl;
    in_range = wl < wpw;
  }
};

template <int DIM, int KIND>
__global__ void __launch_bounds__(256) pair_energy_n3l_kernel(const float* __restrict__ x, float* __restrict__ logp,
                                                              float* __restrict__ force, long long B, int n, int WB,
                                                              PairParams p) {
  extern __shared__ float sm[];
  float* xs = sm;                    // [WB*n*DIM] coordinates
  float* fs = xs + WB * n * DIM;     // [WB*n*DIM] partner forces, then total forces
  float* es = fs + WB * n * DIM;     // [WB*n] per-particle energy partials
  float* ms = es + WB * n;           // [WB*DIM] per-walker coordinate sums
  const int tid = threadIdx.x;
  const WaveMap m(tid, n);
  const long long nblk = (B + WB - 1) / WB;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long w0 = blk * WB;
    const int nw = (int)((B - w0) < WB ? (B - w0) : WB);
    const int nfl = nw * n * DIM;
    const float* src = x + w0 * n * DIM;
    for (int q = tid; q < nfl; q += 256) xs[q] = src[q];
    __syncthreads();
    const bool act = m.in_range && m.w < nw;
    if (act) {
      float f[DIM], e;
      float* fw = fs + m.w * n * DIM;
      pair_force_n3l<DIM, KIND>(xs + m.w * n * DIM, fw, ms + m.w * DIM, m.i, n, p, f, e);
#pragma unroll
      for (int k = 0; k < DIM; ++k) fw[m.i * DIM + k] = f[k];
      es[m.w * n + m.i] = e;
    }
    __syncthreads();
    if (act && m.i == 0) {
      float s = 0.f;
      for (int q = 0; q < n; ++q) s += es[m.w * n + q];
      logp[w0 + m.w] = -s * p.inv_T;
    }
    if (force) {
      float* dst = force + w0 * n * DIM;
      for (int q = tid; q < nfl; q += 256) dst[q] = fs[q];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------- LJ13 fast path
// LJ13 (n = 13, d = 3) with one walker per lane(-pair): coordinates are staged once in LDS (row
// stride 39 words = odd, so the per-lane row reads are bank-conflict free and use immediate
// offsets), the 39 force accumulators live in VGPRs, and the 78 unordered pairs are enumerated
// at compile time as the circulant (i, i+dd mod 13), dd = 1..6, with Newton's third law (the
// generic kernel above evaluates every ordered pair).  P = 2 wave-pairs share a walker set:
// half 0 takes dd in {1,3,4}; half 1 runs the same code on the relabelled particles
// k -> 2k mod 13, which maps {1,3,4} onto {2,6,8=-5}: exactly the other three circulant distances.
// The partial force sets meet in LDS on the way to one fully coalesced float4 store.  At 65 536
// walkers this doubles the waves per SIMD and halves the dependent VALU chain per wave.
template <int P, int MULT, bool OSC, bool WANT_E, bool UNIT_RM>
__device__ __forceinline__ void lj13_partial(const float* __restrict__ xw, float (&fr)[39], float& e_out, const PairParams& p) {
  constexpr int N = 13, NDD = 6 / P;
  constexpr int DD[2][6] = {{1, 2, 3, 4, 5, 6}, {1, 3, 4, 0, 0, 0}};
#pragma unroll
  for (int q = 0; q < 39; ++q) fr[q] = 0.f;
  float e = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    // keep the walker's coordinates in LDS: without these fences the compiler hoists all 39 reads (plus
    // packed-math shuffles) to the top and spills; with them the live set is the 39 force accumulators
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const int pi = ((MULT * i) % N) * 3;
    const float xi0 = xw[pi], xi1 = xw[pi + 1], xi2 = xw[pi + 2];
#pragma unroll
    for (int t = 0; t < NDD; ++t) {
      const int j = (i + DD[P - 1][t]) % N;
      const int pj = ((MULT * j) % N) * 3;
      const float d0 = xi0 - xw[pj], d1 = xi1 - xw[pj + 1], d2 = xi2 - xw[pj + 2];
      const float r2 = fmaf(d2, d2, fmaf(d1, d1, fmaf(d0, d0, p.dist_eps)));
      const float inv = __builtin_amdgcn_rcpf(r2);
      const float s2 = UNIT_RM ? inv : p.rm2 * inv;             // (rm/r)^2  (x * 1.0f is exact: same bits)
      const float s6 = s2 * s2 * s2;
      if (WANT_E) e += fmaf(s6, s6, -2.0f * s6);                 // (rm/r)^12 - 2 (rm/r)^6, per pair (less cancellation)
      const float ts = s6 * s2;
      const float coef = fmaf(-s6, ts, ts);                      // (s^6 - s^12) s^2 = e'(r)/r * rm^2/(12 eps)
      fr[i * 3] = fmaf(coef, d0, fr[i * 3]); fr[i * 3 + 1] = fmaf(coef, d1, fr[i * 3 + 1]); fr[i * 3 + 2] = fmaf(coef, d2, fr[i * 3 + 2]);
      fr[j * 3] = fmaf(-coef, d0, fr[j * 3]); fr[j * 3 + 1] = fmaf(-coef, d1, fr[j * 3 + 1]); fr[j * 3 + 2] = fmaf(-coef, d2, fr[j * 3 + 2]);
    }
  }
  const float pair_w = 2.0f * p.energy_factor * p.eps;  // every unordered pair counts twice in the reference sum
  float osc = 0.f;
  if (OSC) {
    // harmonic oscillator about the particle mean: counted once, by the half that does not apply the update in
    // the fused descent kernel (this keeps the two halves' instruction counts level there)
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < N; ++k) { m0 += xw[k * 3]; m1 += xw[k * 3 + 1]; m2 += xw[k * 3 + 2]; }
    m0 /= (float)N; m1 /= (float)N; m2 /= (float)N;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int pk = ((MULT * k) % N) * 3;
      const float c0 = xw[pk] - m0, c1 = xw[pk + 1] - m1, c2 = xw[pk + 2] - m2;
      if (WANT_E) osc = fmaf(c0, c0, fmaf(c1, c1, fmaf(c2, c2, osc)));
      fr[k * 3] = fmaf(p.cw, fr[k * 3], p.co * c0);  // d logp / dx = -(dE_pair/dx + osc (x - mean)) / T
      fr[k * 3 + 1] = fmaf(p.cw, fr[k * 3 + 1], p.co * c1);
      fr[k * 3 + 2] = fmaf(p.cw, fr[k * 3 + 2], p.co * c2);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 39; ++q) fr[q] = p.cw * fr[q];
  }
  // pin the results: otherwise the compiler sinks ALL the arithmetic below the caller's barrier (its only
  // users are guarded stores) while the 39 coordinate loads must stay above -> everything spills
#pragma unroll
  for (int q = 0; q < 39; ++q) asm volatile("" : "+v"(fr[q]));
  if (WANT_E) {
    asm volatile("" : "+v"(e), "+v"(osc));
    e_out = fmaf(pair_w, e, OSC ? 0.5f * p.osc_scale * osc : 0.f);
  }
}

// fr is indexed by the RELABELLED particle k; the real particle is (MULT*k) mod 13
template <int P, int MULT, bool UNIT_RM>
__device__ __forceinline__ void lj13_body(const float* __restrict__ xw, float* __restrict__ fw, float* __restrict__ e_out,
                                          const PairParams& p, bool act) {
  constexpr int N = 13;
  float fr[39], e;
  lj13_partial<P, MULT, (P == 1 || MULT == 2), true, UNIT_RM>(xw, fr, e, p);
  asm volatile("" : "+v"(e));
  __syncthreads();  // all coordinate reads of the block are done: the stage may be overwritten with forces
  if (act) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int pk = ((MULT * k) % N) * 3;
      fw[pk] = fr[k * 3]; fw[pk + 1] = fr[k * 3 + 1]; fw[pk + 2] = fr[k * 3 + 2];
    }
    *e_out = e;
  }
}

template <int P, bool UNIT_RM>
__global__ void __launch_bounds__(256, 4) lj13_kernel(const float* __restrict__ x, float* __restrict__ logp,
                                                      float* __restrict__ force, long long B, PairParams p) {
  constexpr int D = 39, WPB = 256 / P;
  __shared__ __attribute__((aligned(16))) float fb[P][WPB * D];  // fb[0] doubles as the coordinate stage
  __shared__ float es[P][WPB];
  const int tid = threadIdx.x;
  const int half = tid / WPB, wl = tid - half * WPB;  // half is wave-uniform (WPB is a multiple of 64)
  const long long nblk = (B + WPB - 1) / WPB;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long w0 = blk * WPB;
    const int nw = (int)((B - w0) < WPB ? (B - w0) : WPB);
    const int nfl = nw * D;
    {  // coalesced stage-in (16 B per lane; block base is 16 B aligned because WPB % 4 == 0)
      const float4* src4 = reinterpret_cast<const float4*>(x + w0 * D);
      float4* dst4 = reinterpret_cast<float4*>(&fb[0][0]);
      for (int q = tid; q < nfl / 4; q += 256) dst4[q] = src4[q];
      for (int q = (nfl & ~3) + tid; q < nfl; q += 256) fb[0][q] = x[w0 * D + q];
    }
    __syncthreads();
    const bool act = wl < nw;
    const int row = (act ? wl : 0) * D;
    if (P == 1 || half == 0) lj13_body<P, 1, UNIT_RM>(&fb[0][row], &fb[0][row], &es[0][wl], p, act);
    else lj13_body<P, 2, UNIT_RM>(&fb[0][row], &fb[P - 1][row], &es[P - 1][wl], p, act);
    __syncthreads();
    if (force) {
      float4* dst4 = reinterpret_cast<float4*>(force + w0 * D);
      for (int q = tid; q < nfl / 4; q += 256) {
        float4 v = reinterpret_cast<const float4*>(&fb[0][0])[q];
        if (P == 2) {
          const float4 u = reinterpret_cast<const float4*>(&fb[P - 1][0])[q];
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        dst4[q] = v;
      }
      for (int q = (nfl & ~3) + tid; q < nfl; q += 256) force[w0 * D + q] = fb[0][q] + (P == 2 ? fb[P - 1][q] : 0.f);
    }
    if (tid < nw) logp[w0 + tid] = -p.inv_T * (es[0][tid] + (P == 2 ? es[P - 1][tid] : 0.f));
    __syncthreads();
  }
}

// Large batches (one lane per walker, several tiles per block): the same tile routine in a persistent block that keeps
// the NEXT tile's coordinates in flight while it works on the current one.  A tile is 256 walkers x 156 B = 39 KB; its
// ten 16-byte loads per lane are issued before the pair loop of the tile before it starts and land in registers, so every
// resident block (three per CU: LDS) has a full tile in flight all the time instead of only while it waits for it --
// the plain kernel above computes and streams at 0.46 of the HBM peak because its three blocks per CU have, on
// average, one tile in flight between them (bandwidth = bytes in flight / latency) and the vector pipe idles meanwhile.
template <bool UNIT_RM>
__global__ void __launch_bounds__(256, 3) lj13_stream_kernel(const float* __restrict__ x, float* __restrict__ logp,
                                                             float* __restrict__ force, long long B, PairParams p) {
  constexpr int D = 39, WPB = 256, NV = (WPB * D / 4 + 255) / 256;  // 16-byte pieces of a tile per lane: 10 (9.75)
  __shared__ __attribute__((aligned(16))) float fb[WPB * D];
  __shared__ float es[WPB];
  const int tid = threadIdx.x;
  const long long nblk = (B + WPB - 1) / WPB;
  float4 nxt[NV];
  // full tiles only take the register path (a last, ragged tile is loaded in place below)
#define LJ13_FETCH(BLK)                                                                   \
  {                                                                                       \
    const float4* src4_ = reinterpret_cast<const float4*>(x + (BLK) * WPB * D);           \
    _Pragma("unroll") for (int v = 0; v < NV; ++v) {                                      \
      const int q_ = tid + v * 256;                                                       \
      nxt[v] = (v < NV - 1 || q_ < WPB * D / 4) ? src4_[q_ < WPB * D / 4 ? q_ : 0] : float4{0.f, 0.f, 0.f, 0.f}; \
    }                                                                                     \
  }
  long long blk = blockIdx.x;
  bool have = false;
  if (blk < nblk && (blk + 1) * WPB <= B) { LJ13_FETCH(blk); have = true; }
  for (; blk < nblk; blk += gridDim.x) {
    const long long w0 = blk * WPB;
    const int nw = (int)((B - w0) < WPB ? (B - w0) : WPB);
    const int nfl = nw * D;
    if (have) {
      float4* dst4 = reinterpret_cast<float4*>(&fb[0]);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int q = tid + v * 256;
        if (q < WPB * D / 4) dst4[q] = nxt[v];
      }
    } else {
      const float4* src4 = reinterpret_cast<const float4*>(x + w0 * D);
      float4* dst4 = reinterpret_cast<float4*>(&fb[0]);
      for (int q = tid; q < nfl / 4; q += 256) dst4[q] = src4[q];
      for (int q = (nfl & ~3) + tid; q < nfl; q += 256) fb[q] = x[w0 * D + q];
    }
    const long long nb = blk + gridDim.x;
    have = nb < nblk && (nb + 1) * WPB <= B;
    if (have) LJ13_FETCH(nb);  // in flight during this tile's pair loop
    __syncthreads();
    const bool act = tid < nw;
    const int row = (act ? tid : 0) * D;
    lj13_body<1, 1, UNIT_RM>(&fb[row], &fb[row], &es[tid], p, act);
    __syncthreads();
    if (force) {
      float4* dst4 = reinterpret_cast<float4*>(force + w0 * D);
      for (int q = tid; q < nfl / 4; q += 256) dst4[q] = reinterpret_cast<const float4*>(&fb[0])[q];
      for (int q = (nfl & ~3) + tid; q < nfl; q += 256) force[w0 * D + q] = fb[q];
    }
    if (tid < nw) logp[w0 + tid] = -p.inv_T * es[tid];
    __syncthreads();
  }
#undef LJ13_FETCH
}

// ---------------------------------------------------------------------------- fused descent on the target
// S steps of  x <- remove_mean(x + F(x) dt + noise_scale * sqrt_dt * xi)  in ONE launch
// (sde_integration.py:353-360 negative_time_descent; ULA when noise_scale = 1).  The walkers of a block stay
// in LDS for all S steps: HBM sees one read and one write of x per launch instead of per step, and the
// per-step force array never exists.  Arithmetic and summation orders are those of pita_*_logp_force
// followed by pita_em_step, so the fused and the per-step paths agree bit for bit.

template <int DIM, int KIND>
__global__ void __launch_bounds__(256) pair_descent_kernel(float* __restrict__ x, const float* __restrict__ noise,
                                                           long long B, int n, int WB, PairParams p, DescentParams q) {
  extern __shared__ float sm[];  // [WB*n*DIM]
  const int tid = threadIdx.x;
  const long long nblk = (B + WB - 1) / WB;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long w0 = blk * WB;
    const int nw = (int)((B - w0) < WB ? (B - w0) : WB);
    const int nfl = nw * n * DIM;
    float* gx = x + w0 * n * DIM;
    for (int i = tid; i < nfl; i += 256) sm[i] = gx[i];
    __syncthreads();
    const int w = tid / n, i = tid - w * n;
    const bool act = w < nw;
    float* xw = sm + w * n * DIM;
    for (int s = 0; s < q.nsteps; ++s) {
      float f[DIM], e, v[DIM];
      if (act) {
        pair_force<DIM, KIND>(xw, i, n, p, f, e);
        float xi[4] = {0.f, 0.f, 0.f, 0.f};
        if (q.noise_scale != 0.f) {
          if (noise) {
#pragma unroll
            for (int k = 0; k < DIM; ++k) xi[k] = noise[(((long long)s * B + w0 + w) * n + i) * DIM + k];
          } else {
            philox_normal4(q.seed, q.walker_offset + (unsigned long long)(w0 + w), q.step0 + s, (uint32_t)i, xi);
          }
        }
#pragma unroll
        for (int k = 0; k < DIM; ++k) v[k] = xw[i * DIM + k] + (f[k] * q.dt + ((q.noise_scale * xi[k]) * q.sqrt_dt));
      }
      __syncthreads();  // every force of this step has read the old coordinates
      if (act) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) xw[i * DIM + k] = v[k];
      }
      __syncthreads();
      if (q.remove_mean) {
        if (act) {
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            float sum = 0.f;
            for (int j = 0; j < n; ++j) sum += xw[j * DIM + k];
            v[k] -= sum / (float)n;
          }
        }
        __syncthreads();
        if (act) {
#pragma unroll
          for (int k = 0; k < DIM; ++k) xw[i * DIM + k] = v[k];
        }
        __syncthreads();
      }
    }
    for (int i2 = tid; i2 < nfl; i2 += 256) gx[i2] = sm[i2];
    __syncthreads();
  }
}

// wave-local variant (n <= 64, pair_force_n3l): a walker never leaves its wavefront, so the step loop needs no
// workgroup barrier at all
template <int DIM, int KIND>
__global__ void __launch_bounds__(256) pair_descent_n3l_kernel(float* __restrict__ x, const float* __restrict__ noise,
                                                               long long B, int n, int WB, PairParams p, DescentParams q) {
  extern __shared__ float sm[];
  float* xs = sm;                    // [WB*n*DIM]
  float* fs = xs + WB * n * DIM;     // [WB*n*DIM]
  float* ms = fs + WB * n * DIM;     // [WB*DIM]
  const int tid = threadIdx.x;
  const WaveMap m(tid, n);
  const long long nblk = (B + WB - 1) / WB;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long w0 = blk * WB;
    const int nw = (int)((B - w0) < WB ? (B - w0) : WB);
    const int nfl = nw * n * DIM;
    float* gx = x + w0 * n * DIM;
    for (int c = tid; c < nfl; c += 256) xs[c] = gx[c];
    __syncthreads();
    const bool act = m.in_range && m.w < nw;
    if (act) {
      float* xw = xs + m.w * n * DIM;
      float* fw = fs + m.w * n * DIM;
      float* mw = ms + m.w * DIM;
      const int i = m.i;
      for (int s = 0; s < q.nsteps; ++s) {
        float f[DIM], e, v[DIM];
        pair_force_n3l<DIM, KIND>(xw, fw, mw, i, n, p, f, e);
        float xi[4] = {0.f, 0.f, 0.f, 0.f};
        if (q.noise_scale != 0.f) {
          if (noise) {
#pragma unroll
            for (int k = 0; k < DIM; ++k) xi[k] = noise[(((long long)s * B + w0 + m.w) * n + i) * DIM + k];
          } else {
            philox_normal4(q.seed, q.walker_offset + (unsigned long long)(w0 + m.w), q.step0 + s, (uint32_t)i, xi);
          }
        }
#pragma unroll
        for (int k = 0; k < DIM; ++k) v[k] = xw[i * DIM + k] + (f[k] * q.dt + ((q.noise_scale * xi[k]) * q.sqrt_dt));
        pair_wave_fence();
#pragma unroll
        for (int k = 0; k < DIM; ++k) xw[i * DIM + k] = v[k];
        pair_wave_fence();
        if (q.remove_mean) {
          float mean[DIM];
          walker_mean<DIM>(xw, mw, i, n, mean);
#pragma unroll
          for (int k = 0; k < DIM; ++k) xw[i * DIM + k] = v[k] - mean[k] / (float)n;
          pair_wave_fence();
        }
      }
    }
    __syncthreads();
    for (int c = tid; c < nfl; c += 256) gx[c] = xs[c];
    __syncthreads();
  }
}

// LJ13: 128 walkers per 256-thread block, the two halves of lj13_kernel<2>.  Half 1 hands its partial
// forces over through LDS; half 0 adds them to its own (still in registers), applies the update and the
// centring for all 13 particles of its walker and writes the new coordinates back to the LDS stage.
template <bool UNIT_RM>
__global__ void __launch_bounds__(256, 2) lj13_descent_kernel(float* __restrict__ x, const float* __restrict__ noise,
                                                              long long B, PairParams p, DescentParams q) {
  constexpr int D = 39, WPB = 128, N = 13;
  __shared__ __attribute__((aligned(16))) float xs[WPB * D];
  __shared__ __attribute__((aligned(16))) float fb[WPB * D];
  __shared__ float nzb[WPB * D];  // (noise_scale * xi) * sqrt_dt of the current step
  const int tid = threadIdx.x;
  const int half = tid / WPB, wl = tid - half * WPB;
  const bool langevin = q.noise_scale != 0.f;
  const long long nblk = (B + WPB - 1) / WPB;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long w0 = blk * WPB;
    const int nw = (int)((B - w0) < WPB ? (B - w0) : WPB);
    const int nfl = nw * D;
    {
      const float4* src4 = reinterpret_cast<const float4*>(x + w0 * D);
      float4* dst4 = reinterpret_cast<float4*>(xs);
      for (int i = tid; i < nfl / 4; i += 256) dst4[i] = src4[i];
      for (int i = (nfl & ~3) + tid; i < nfl; i += 256) xs[i] = x[w0 * D + i];
    }
    __syncthreads();
    const bool act = wl < nw;
    const int row = (act ? wl : 0) * D;
    for (int s = 0; s < q.nsteps; ++s) {
      if (langevin) {  // both halves draw the step's noise (6 / 7 particles each) into LDS
#pragma unroll 1
        for (int k = half ? 6 : 0; k < (half ? N : 6); ++k) {
          float xi[4] = {0.f, 0.f, 0.f, 0.f};
          if (noise) {
            const float* nz = noise + (((long long)s * B + w0 + (act ? wl : 0)) * D + k * 3);
            xi[0] = nz[0]; xi[1] = nz[1]; xi[2] = nz[2];
          } else {
            philox_normal4(q.seed, q.walker_offset + (unsigned long long)(w0 + wl), q.step0 + s, (uint32_t)k, xi);
          }
          if (act) {
#pragma unroll
            for (int c = 0; c < 3; ++c) nzb[row + k * 3 + c] = (q.noise_scale * xi[c]) * q.sqrt_dt;
          }
        }
      }
      float fr[39], e;
      if (half == 0) {
        lj13_partial<2, 1, false, false, UNIT_RM>(xs + row, fr, e, p);
      } else {
        lj13_partial<2, 2, true, false, UNIT_RM>(xs + row, fr, e, p);
        if (act) {
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const int pk = ((2 * k) % N) * 3;
            fb[row + pk] = fr[k * 3]; fb[row + pk + 1] = fr[k * 3 + 1]; fb[row + pk + 2] = fr[k * 3 + 2];
          }
        }
      }
      __syncthreads();  // partial forces of half 1 are in LDS; nobody reads the old coordinates any more
      if (half == 0) {
        float m0 = 0.f, m1 = 0.f, m2 = 0.f;
        if (langevin) {
#pragma unroll
          for (int k = 0; k < 39; ++k)
            fr[k] = xs[row + k] + ((fr[k] + fb[row + k]) * q.dt + nzb[row + k]);
        } else {
#pragma unroll
          for (int k = 0; k < 39; ++k) fr[k] = xs[row + k] + (fr[k] + fb[row + k]) * q.dt;
        }
#pragma unroll
        for (int k = 0; k < N; ++k) { m0 += fr[k * 3]; m1 += fr[k * 3 + 1]; m2 += fr[k * 3 + 2]; }
        if (q.remove_mean) {
          m0 /= (float)N; m1 /= (float)N; m2 /= (float)N;
#pragma unroll
          for (int k = 0; k < N; ++k) { fr[k * 3] -= m0; fr[k * 3 + 1] -= m1; fr[k * 3 + 2] -= m2; }
        }
        if (act) {
#pragma unroll
          for (int k = 0; k < 39; ++k) xs[row + k] = fr[k];
        }
      }
      __syncthreads();
    }
    {
      float4* dst4 = reinterpret_cast<float4*>(x + w0 * D);
      const float4* src4 = reinterpret_cast<const float4*>(xs);
      for (int i = tid; i < nfl / 4; i += 256) dst4[i] = src4[i];
      for (int i = (nfl & ~3) + tid; i < nfl; i += 256) x[w0 * D + i] = xs[i];
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------- fused MALA chain on the LJ13 target
// All post_mcmc_steps of metropolis_hastings_mala(_adaptive) (sde_integration.py:362-470) in ONE launch with the walkers
// resident in LDS: per step  force(x) -> proposal x' = (x + dt/2 F) + sqrt(dt) xi (:28-45) -> logp, force at x' ->
// log q_f, log q_b, accept iff log u < (logp' - logp) + (log q_b - log q_f) -> the reference's float blend of x and
// logp, optional centring, acceptance count -> step-size adaptation dt *= 1.1 / dt /= 1.1 on the GLOBAL acceptance rate
// (:439-443).  The two target evaluations per step are the two halves of lj13_kernel<2> (half 1 hands its partial forces
// and energy over through LDS), the elementwise arithmetic and its summation orders are those of mala_propose_kernel /
// mala_accept_kernel / mala_adapt_kernel (sampler_kernels.hip): the chain is bit-identical to the launch-per-kernel
// path.  The adaptive variant needs every block's acceptance count before the next step: one grid-wide barrier per step
// (all blocks co-resident: checked by the launch wrapper).  Only the counter itself crosses blocks, so the barrier is ONE
// relaxed agent-scope atomic add of (1 << 32 | accepted) per block and a relaxed polling load -- no release / acquire
// fences, which at agent scope write back and invalidate the XCD's L2 (measured: ~80 us per step with them); bounded spin.

template <bool UNIT_RM>
__global__ void __launch_bounds__(256, 2) lj13_mala_kernel(float* __restrict__ x, float* __restrict__ logp, long long B,
                                                           PairParams p, MalaParams q) {
  constexpr int D = 39, WPB = 128, N = 13;
  __shared__ __attribute__((aligned(16))) float xs[WPB * D];   // current walkers
  __shared__ __attribute__((aligned(16))) float xps[WPB * D];  // proposals
  __shared__ __attribute__((aligned(16))) float fb[WPB * D];   // half 1's partial forces
  __shared__ float es1[WPB];
  __shared__ int cnt[4];
  __shared__ int total_acc;
  const int tid = threadIdx.x;
  const int half = tid / WPB, wl = tid - half * WPB;
  const long long nblk = (B + WPB - 1) / WPB;
  // Tiles of 128 walkers.  Non-adaptive chains and adaptive chains with at most one tile per block keep a tile in LDS
  // for all steps; an adaptive chain with more tiles than resident blocks runs steps outside, tiles inside, one HBM
  // round trip of the walkers per step (the grid barrier of a step needs every tile's count first).
  const bool roundtrip = q.adaptive && nblk > (long long)gridDim.x;
  double dt = q.dt_dev[0];
  long long w0 = 0, wg = 0;
  int nw = 0, nfl = 0, row = 0;
  bool act = false;
  unsigned long long key = 0;
  float lp = 0.f;
  auto load = [&](long long blk) {
    w0 = blk * WPB;
    nw = (int)((B - w0) < WPB ? (B - w0) : WPB);
    nfl = nw * D;
    {
      const float4* src4 = reinterpret_cast<const float4*>(x + w0 * D);
      float4* dst4 = reinterpret_cast<float4*>(xs);
      for (int i = tid; i < nfl / 4; i += 256) dst4[i] = src4[i];
      for (int i = (nfl & ~3) + tid; i < nfl; i += 256) xs[i] = x[w0 * D + i];
    }
    __syncthreads();
    act = wl < nw;
    row = (act ? wl : 0) * D;
    wg = w0 + (act ? wl : 0);
    key = q.walker_ids ? (unsigned long long)q.walker_ids[wg] : q.walker_offset + (unsigned long long)wg;
    lp = (half == 0 && act) ? logp[wg] : 0.f;
  };
  auto store = [&]() {
    {
      float4* dst4 = reinterpret_cast<float4*>(x + w0 * D);
      const float4* src4 = reinterpret_cast<const float4*>(xs);
      for (int i = tid; i < nfl / 4; i += 256) dst4[i] = src4[i];
      for (int i = (nfl & ~3) + tid; i < nfl; i += 256) x[w0 * D + i] = xs[i];
    }
    if (half == 0 && act) logp[wg] = lp;
    __syncthreads();
  };
  // one MALA step of the tile in LDS; returns the tile's accepted walkers (in every thread)
  auto step = [&](int s) -> int {
      const float hdt = (float)(0.5 * dt), sdt = (float)sqrt(dt), tdt = (float)(2.0 * dt);
      float fr[39], e;
      // ---- A: target force at x
      if (half == 0) {
        lj13_partial<2, 1, false, false, UNIT_RM>(xs + row, fr, e, p);
      } else {
        lj13_partial<2, 2, true, false, UNIT_RM>(xs + row, fr, e, p);
        if (act) {
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const int pk = ((2 * k) % N) * 3;
            fb[row + pk] = fr[k * 3]; fb[row + pk + 1] = fr[k * 3 + 1]; fb[row + pk + 2] = fr[k * 3 + 2];
          }
        }
      }
      __syncthreads();
      // ---- B: proposal and log q_f (half 0 owns the walker)
      float sf = 0.f;
      if (half == 0 && act) {
#pragma unroll 1
        for (int i = 0; i < N; ++i) {
          float xi[4] = {0.f, 0.f, 0.f, 0.f};
          if (q.noise) {
            const float* nz = q.noise + (((long long)s * B + wg) * D + i * 3);
            xi[0] = nz[0]; xi[1] = nz[1]; xi[2] = nz[2];
          } else {
            philox_normal4(q.seed, key, q.step0 + s, (uint32_t)i, xi);
          }
          float sfi = 0.f;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int k = i * 3 + c;
            const float F = fr[k] + fb[row + k];
            const float t1 = xs[row + k] + hdt * F;
            const float xp = t1 + sdt * xi[c];
            xps[row + k] = xp;
            const float df = xp - t1;
            sfi += df * df;
          }
          sf += sfi;
        }
      }
      __syncthreads();
      // ---- C: target log-density and force at x'
      if (half == 0) {
        lj13_partial<2, 1, false, true, UNIT_RM>(xps + row, fr, e, p);
      } else {
        lj13_partial<2, 2, true, true, UNIT_RM>(xps + row, fr, e, p);
        if (act) {
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const int pk = ((2 * k) % N) * 3;
            fb[row + pk] = fr[k * 3]; fb[row + pk + 1] = fr[k * 3 + 1]; fb[row + pk + 2] = fr[k * 3 + 2];
          }
          es1[wl] = e;
        }
      }
      __syncthreads();
      // ---- D: accept / reject, blend, centring, acceptance count
      bool acc = false;
      if (half == 0 && act) {
        const float lpp = -p.inv_T * (e + es1[wl]);
        float sb = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) {
          float sbi = 0.f;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int k = i * 3 + c;
            const float Fp = fr[k] + fb[row + k];
            const float db = xs[row + k] - (xps[row + k] + hdt * Fp);
            sbi += db * db;
          }
          sb += sbi;
        }
        const float lqf = -sf / tdt, lqb = -sb / tdt;
        const float ratio = (lpp - lp) + (lqb - lqf);
        const float u = q.uniforms ? q.uniforms[(long long)s * B + wg] : philox_uniform(q.seed, key, q.step0 + s, 0xFFFFFu);
        const float af = (logf(u) < ratio) ? 1.0f : 0.0f;
        acc = af != 0.f;
        lp = af * lpp + (1.0f - af) * lp;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int k = 0; k < 39; ++k) fr[k] = af * xps[row + k] + (1.0f - af) * xs[row + k];
        if (q.remove_mean) {
#pragma unroll
          for (int k = 0; k < N; ++k) { m0 += fr[k * 3]; m1 += fr[k * 3 + 1]; m2 += fr[k * 3 + 2]; }
          m0 /= (float)N; m1 /= (float)N; m2 /= (float)N;
#pragma unroll
          for (int k = 0; k < N; ++k) { fr[k * 3] -= m0; fr[k * 3 + 1] -= m1; fr[k * 3 + 2] -= m2; }
        }
#pragma unroll
        for (int k = 0; k < 39; ++k) xs[row + k] = fr[k];
      }
      const unsigned long long bal = __ballot(acc);
      if ((tid & 63) == 0) cnt[tid >> 6] = __popcll(bal);
      __syncthreads();
    return cnt[0] + cnt[1] + cnt[2] + cnt[3];
  };
  // a step's count of this block goes to the grid; adaptive chains wait for everybody's and adapt dt
  auto publish = [&](int s, int c) {
    if (tid == 0) {
      __hip_atomic_fetch_add(&q.sync[s], (1ull << 32) | (unsigned long long)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (q.adaptive) {  // grid-wide barrier: wait until every block has added its count
        const unsigned long long v = mala_grid_wait(q.sync, s, q.nsteps,
                                                    (unsigned long long)gridDim.x + q.debug_missing_blocks, q.spin_limit);
        total_acc = (int)(v & 0xFFFFFFFFull);
      }
    }
    __syncthreads();
    if (q.adaptive) {
      const float rate = (float)total_acc / (float)q.total;
      dt = ((double)rate > 0.55) ? dt * 1.1 : dt / 1.1;  // sde_integration.py:439-443
    }
  };
  if (roundtrip) {
    for (int s = 0; s < q.nsteps; ++s) {
      int c = 0;
      for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        load(blk);
        c += step(s);
        store();
      }
      publish(s, c);
    }
  } else {
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
      load(blk);
      for (int s = 0; s < q.nsteps; ++s) publish(s, step(s));
      store();
    }
  }
}

// acceptance rates of all steps and the final step size from the per-step counts (same arithmetic as mala_adapt_kernel).
// A chain that raised its error flag (a grid-barrier spin ran out: the blocks were not all co-resident after all, e.g.
// because another stream or process held compute units) adapted dt from different counts in different blocks: its
// walkers are not a valid chain.  dt and every rate are then NaN, which no caller can mistake for a result
// (WeightedSDEIntegrator._mala restores its backup and reruns the launch-per-kernel chain).
__global__ void mala_finish_kernel(double* dt_dev, const unsigned long long* sync, int nsteps, long long total, int adaptive,
                                   float* rates_out) {
  const bool failed = sync[nsteps] != 0;
  double dt = dt_dev[0];
  for (int s = 0; s < nsteps; ++s) {
    const float rate = (float)(int)(sync[s] & 0xFFFFFFFFull) / (float)total;
    if (rates_out) rates_out[s] = failed ? __builtin_nanf("") : rate;
    if (adaptive) dt = ((double)rate > 0.55) ? dt * 1.1 : dt / 1.1;
  }
  dt_dev[0] = failed ? __builtin_nan("") : dt;
}

int mala_spin_limit() {
  const char* e = getenv("PITA_DEBUG_MALA_SPIN_LIMIT");
  if (e && *e) return atoi(e);
  return 1 << 22;
}

int launch_mala_finish(double* dt_dev, const unsigned long long* sync, int nsteps, long long total, int adaptive,
                       float* rates_out, void* stream) {
  hipLaunchKernelGGL(mala_finish_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dt_dev, sync, nsteps, total, adaptive,
                     rates_out);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

template <int KIND>
static int launch_descent(float* x, const float* noise, int64_t B, int n, int d, const PairParams& p, const DescentParams& q,
                          void* stream) {
  PITA_REQUIRE(B >= 0 && q.nsteps >= 0, "descent: negative batch or step count");
  if (B == 0 || q.nsteps == 0) return PITA_OK;
  PITA_REQUIRE(x, "descent: null argument");
  PITA_REQUIRE(n >= 2 && n <= 256, "descent: n_particles must be in [2,256]");
  PITA_REQUIRE(d >= 1 && d <= 3, "descent: n_dim must be 1, 2 or 3");
  hipStream_t s = (hipStream_t)stream;
  if (KIND == E_LJ && n == 13 && d == 3) {
    const long long nblk = (B + 127) / 128;
    const unsigned grid = (unsigned)(nblk < 256LL * 4 ? nblk : 256LL * 4);
    if (p.rm2 == 1.0f) hipLaunchKernelGGL(lj13_descent_kernel<true>, dim3(grid), dim3(256), 0, s, x, noise, (long long)B, p, q);
    else hipLaunchKernelGGL(lj13_descent_kernel<false>, dim3(grid), dim3(256), 0, s, x, noise, (long long)B, p, q);
    PITA_LAUNCH_CHECK();
    return PITA_OK;
  }
  {  // compile-time particle counts (LJ55, DW4): ring_kernels.hip
    const int rc = ring_launch_descent(KIND, x, noise, B, n, d, p, q, stream);
    if (rc != 1) return rc;
  }
  if (n <= 64) {
    const int WB = 4 * (64 / n);
    const long long nblk = (B + WB - 1) / WB;
    const unsigned grid = (unsigned)(nblk < 256LL * 16 ? nblk : 256LL * 16);
    const size_t lds = sizeof(float) * (size_t)(2 * WB * n * d + WB * d);
    switch (d) {
      case 1: hipLaunchKernelGGL((pair_descent_n3l_kernel<1, KIND>), dim3(grid), dim3(256), lds, s, x, noise, B, n, WB, p, q); break;
      case 2: hipLaunchKernelGGL((pair_descent_n3l_kernel<2, KIND>), dim3(grid), dim3(256), lds, s, x, noise, B, n, WB, p, q); break;
      default: hipLaunchKernelGGL((pair_descent_n3l_kernel<3, KIND>), dim3(grid), dim3(256), lds, s, x, noise, B, n, WB, p, q); break;
    }
    PITA_LAUNCH_CHECK();
    return PITA_OK;
  }
  const int WB = 256 / n;
  const long long nblk = (B + WB - 1) / WB;
  const unsigned grid = (unsigned)(nblk < 256LL * 16 ? nblk : 256LL * 16);
  const size_t lds = sizeof(float) * (size_t)(WB * n * d);
  switch (d) {
    case 1: hipLaunchKernelGGL((pair_descent_kernel<1, KIND>), dim3(grid), dim3(256), lds, s, x, noise, B, n, WB, p, q); break;
    case 2: hipLaunchKernelGGL((pair_descent_kernel<2, KIND>), dim3(grid), dim3(256), lds, s, x, noise, B, n, WB, p, q); break;
    default: hipLaunchKernelGGL((pair_descent_kernel<3, KIND>), dim3(grid), dim3(256), lds, s, x, noise, B, n, WB, p, q); break;
  }
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

template <int KIND>
static int launch_pair(const float* x, float* logp, float* force, int64_t B, int n, int d, const PairParams& p,
                       void* stream) {
  PITA_REQUIRE(B >= 0, "pair energy: negative batch");
  if (B == 0) return PITA_OK;
  PITA_REQUIRE(x && logp, "pair energy: null argument");
  PITA_REQUIRE(n >= 2 && n <= 256, "pair energy: n_particles must be in [2,256]");
  PITA_REQUIRE(d >= 1 && d <= 3, "pair energy: n_dim must be 1, 2 or 3");
  if (B == 0) return PITA_OK;
  hipStream_t s = (hipStream_t)stream;
  {  // compile-time particle counts (LJ55, DW4): ring_kernels.hip
    const int rc = ring_launch_energy(KIND, x, logp, force, B, n, d, p, stream);
    if (rc != 1) return rc;
  }
  if (n <= 64) {  // whole walkers per wavefront: unordered pairs with Newton's third law
    const int WB = 4 * (64 / n);
    const long long nblk = (B + WB - 1) / WB;
    const unsigned grid = (unsigned)(nblk < 256LL * 16 ? nblk : 256LL * 16);
    const size_t lds = sizeof(float) * (size_t)(2 * WB * n * d + WB * n + WB * d);
    switch (d) {
      case 1: hipLaunchKernelGGL((pair_energy_n3l_kernel<1, KIND>), dim3(grid), dim3(256), lds, s, x, logp, force, B, n, WB, p); break;
      case 2: hipLaunchKernelGGL((pair_energy_n3l_kernel<2, KIND>), dim3(grid), dim3(256), lds, s, x, logp, force, B, n, WB, p); break;
      default: hipLaunchKernelGGL((pair_energy_n3l_kernel<3, KIND>), dim3(grid), dim3(256), lds, s, x, logp, force, B, n, WB, p); break;
    }
    PITA_LAUNCH_CHECK();
    return PITA_OK;
  }
  const int WB = 256 / n;
  const long long nblk = (B + WB - 1) / WB;
  const unsigned grid = (unsigned)(nblk < 256LL * 16 ? nblk : 256LL * 16);
  const size_t lds = sizeof(float) * (size_t)(WB * n * d + WB * n);
  switch (d) {
    case 1: hipLaunchKernelGGL((pair_energy_kernel<1, KIND>), dim3(grid), dim3(256), lds, s, x, logp, force, B, n, WB, p); break;
    case 2: hipLaunchKernelGGL((pair_energy_kernel<2, KIND>), dim3(grid), dim3(256), lds, s, x, logp, force, B, n, WB, p); break;
    default: hipLaunchKernelGGL((pair_energy_kernel<3, KIND>), dim3(grid), dim3(256), lds, s, x, logp, force, B, n, WB, p); break;
  }
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

// ---------------------------------------------------------------------------- GMM
template <int DIM>
__global__ void __launch_bounds__(256) gmm_kernel(const float* __restrict__ x, float* __restrict__ logp,
                                                  float* __restrict__ force, long long B, const float* __restrict__ means,
                                                  const float* __restrict__ scales, int K, float inv_T) {
  extern __shared__ float sm[];
  float* mu = sm;                 // [K*DIM]
  float* isg = sm + K * DIM;      // [K*DIM] 1/scale
  float* lc = sm + 2 * K * DIM;   // [K] -sum log scale - 0.5 dim log 2pi
  for (int k = threadIdx.x; k < K; k += 256) {
    float ls = 0.f;
    for (int d = 0; d < DIM; ++d) {
      mu[k * DIM + d] = means[k * DIM + d];
      isg[k * DIM + d] = 1.0f / scales[k * DIM + d];
      ls += logf(scales[k * DIM + d]);
    }
    lc[k] = -ls - 0.5f * DIM * 1.8378770664093453f;
  }
  __syncthreads();
  for (long long b = (long long)blockIdx.x * 256 + threadIdx.x; b < B; b += (long long)gridDim.x * 256) {
    float xv[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) xv[d] = x[b * DIM + d];
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) {
      float m = 0.f;
#pragma unroll
      for (int d = 0; d < DIM; ++d) { const float z = (xv[d] - mu[k * DIM + d]) * isg[k * DIM + d]; m += z * z; }
      mx = fmaxf(mx, lc[k] - 0.5f * m);
    }
    float s = 0.f, g[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) g[d] = 0.f;
    for (int k = 0; k < K; ++k) {
      float m = 0.f, z[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d) { z[d] = (xv[d] - mu[k * DIM + d]) * isg[k * DIM + d]; m += z[d] * z[d]; }
      const float w = expf((lc[k] - 0.5f * m) - mx);
      s += w;
#pragma unroll
      for (int d = 0; d < DIM; ++d) g[d] -= w * z[d] * isg[k * DIM + d];
    }
    logp[b] = ((mx + logf(s)) - logf((float)K)) * inv_T;
    if (force) {
#pragma unroll
      for (int d = 0; d < DIM; ++d) force[b * DIM + d] = (g[d] / s) * inv_T;
    }
  }
}

}  // namespace pita

using namespace pita;

extern "C" int pita_lj_smooth_logp_force(const float* x, float* logp, float* force, int64_t B, int n, int d,
                                         float temperature, float energy_factor, float dist_eps, float eps, float rm,
                                         float osc_scale, float range_min, const float* coef4, void* stream) {
  PITA_REQUIRE(temperature > 0.f, "pita_lj_smooth_logp_force: temperature must be > 0");
  PITA_REQUIRE(coef4 && range_min > 0.f, "pita_lj_smooth_logp_force: spline coefficients / range_min");
  PairParams p{};
  p.inv_T = 1.0f / temperature; p.energy_factor = energy_factor; p.dist_eps = dist_eps; p.eps = eps;
  p.rm2 = rm * rm; p.osc_scale = osc_scale;
  p.sm_min = range_min; p.sc0 = coef4[0]; p.sc1 = coef4[1]; p.sc2 = coef4[2]; p.sc3 = coef4[3];
  return launch_pair<E_LJS>(x, logp, force, B, n, d, p, stream);
}

extern "C" int pita_lj_logp_force(const float* x, float* logp, float* force, int64_t B, int n, int d, float temperature,
                                  float energy_factor, float dist_eps, float eps, float rm, float osc_scale,
                                  void* stream) {
  PITA_REQUIRE(temperature > 0.f, "pita_lj_logp_force: temperature must be > 0");
  PairParams p{};
  p.inv_T = 1.0f / temperature; p.energy_factor = energy_factor; p.dist_eps = dist_eps; p.eps = eps;
  p.rm2 = rm * rm; p.osc_scale = osc_scale;
  p.cw = -p.inv_T * (2.0f * energy_factor * eps * 12.0f / p.rm2); p.co = -p.inv_T * osc_scale;
  if (n == 13 && d == 3 && B > 0) {
    PITA_REQUIRE(x && logp, "pita_lj_logp_force: null argument");
    if (getenv("PITA_LJ13_RING")) return ring_launch_energy(E_LJ, x, logp, force, B, n, d, p, stream);
    // 2 lanes per walker while the batch cannot fill every SIMD twice with 1 lane per walker
    const bool two = B <= 256LL * 1024;
    const int WPB = two ? 128 : 256;
    const long long nblk = (B + WPB - 1) / WPB;
    const unsigned grid = (unsigned)(nblk < 256LL * 32 ? nblk : 256LL * 32);
    const bool unit = p.rm2 == 1.0f;
    hipStream_t s = (hipStream_t)stream;
    if (!two && !getenv("PITA_LJ13_NO_STREAM")) {  // persistent blocks with the next tile in flight
      static PerDevice<int> per_cu_on, n_cu_on;
      int &per_cu = per_cu_on.get(), &n_cu = n_cu_on.get();
      if (per_cu == 0) {
        int dev = 0, v = 0;
        hipDeviceProp_t prop;
        PITA_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, lj13_stream_kernel<true>, 256, 0));
        PITA_HIP_CHECK(hipGetDevice(&dev));
        PITA_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        per_cu = v > 0 ? v : 1;
        n_cu = prop.multiProcessorCount;
      }
      const long long cap = (long long)per_cu * n_cu;
      const unsigned g2 = (unsigned)(nblk < cap ? nblk : cap);
      if (unit) hipLaunchKernelGGL(lj13_stream_kernel<true>, dim3(g2), dim3(256), 0, s, x, logp, force, (long long)B, p);
      else hipLaunchKernelGGL(lj13_stream_kernel<false>, dim3(g2), dim3(256), 0, s, x, logp, force, (long long)B, p);
      PITA_LAUNCH_CHECK();
      return PITA_OK;
    }
    if (two && unit) hipLaunchKernelGGL((lj13_kernel<2, true>), dim3(grid), dim3(256), 0, s, x, logp, force, (long long)B, p);
    else if (two) hipLaunchKernelGGL((lj13_kernel<2, false>), dim3(grid), dim3(256), 0, s, x, logp, force, (long long)B, p);
    else if (unit) hipLaunchKernelGGL((lj13_kernel<1, true>), dim3(grid), dim3(256), 0, s, x, logp, force, (long long)B, p);
    else hipLaunchKernelGGL((lj13_kernel<1, false>), dim3(grid), dim3(256), 0, s, x, logp, force, (long long)B, p);
    PITA_LAUNCH_CHECK();
    return PITA_OK;
  }
  return launch_pair<E_LJ>(x, logp, force, B, n, d, p, stream);
}

extern "C" int pita_dw_logp_force(const float* x, float* logp, float* force, int64_t B, int n, int d, float temperature,
                                  float a, float b, float c, float d0, void* stream) {
  PITA_REQUIRE(temperature > 0.f, "pita_dw_logp_force: temperature must be > 0");
  PairParams p{};
  p.inv_T = 1.0f / temperature; p.a = a; p.b = b; p.c = c; p.d0 = d0;
  return launch_pair<E_DW>(x, logp, force, B, n, d, p, stream);
}

extern "C" int pita_gmm_logp_force(const float* x, float* logp, float* force, int64_t B, int dim, const float* means,
                                   const float* scales, int K, float temperature, void* stream) {
  PITA_REQUIRE(B >= 0, "pita_gmm_logp_force: negative batch");
  if (B == 0) return PITA_OK;
  PITA_REQUIRE(x && logp && means && scales, "pita_gmm_logp_force: null argument");
  PITA_REQUIRE(K >= 1 && K <= 2048, "pita_gmm_logp_force: K out of range");
  PITA_REQUIRE(temperature > 0.f, "pita_gmm_logp_force: temperature must be > 0");
  if (dim < 1 || dim > 4) return fail(PITA_EUNSUPPORTED, "pita_gmm_logp_force: dim=%d (1..4 implemented)", dim);
  if (B == 0) return PITA_OK;
  const long long nb = (B + 255) / 256;
  const unsigned grid = (unsigned)(nb < 4096 ? nb : 4096);
  const size_t lds = sizeof(float) * (size_t)(2 * K * dim + K);
  const float inv_T = 1.0f / temperature;
  hipStream_t s = (hipStream_t)stream;
  switch (dim) {
    case 1: hipLaunchKernelGGL(gmm_kernel<1>, dim3(grid), dim3(256), lds, s, x, logp, force, B, means, scales, K, inv_T); break;
    case 2: hipLaunchKernelGGL(gmm_kernel<2>, dim3(grid), dim3(256), lds, s, x, logp, force, B, means, scales, K, inv_T); break;
    case 3: hipLaunchKernelGGL(gmm_kernel<3>, dim3(grid), dim3(256), lds, s, x, logp, force, B, means, scales, K, inv_T); break;
    default: hipLaunchKernelGGL(gmm_kernel<4>, dim3(grid), dim3(256), lds, s, x, logp, force, B, means, scales, K, inv_T); break;
  }
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

static DescentParams descent_params(int nsteps, float dt, float noise_scale, float sqrt_dt, uint64_t seed,
                                    uint64_t walker_offset, int64_t step0, int remove_mean) {
  DescentParams q{};
  q.dt = dt; q.noise_scale = noise_scale; q.sqrt_dt = sqrt_dt; q.nsteps = nsteps; q.remove_mean = remove_mean;
  q.seed = seed; q.walker_offset = walker_offset; q.step0 = step0;
  return q;
}

extern "C" int pita_lj_descent(float* x, const float* noise, int64_t B, int n, int d, float temperature, float energy_factor,
                               float dist_eps, float eps, float rm, float osc_scale, int nsteps, float dt,
                               float noise_scale, float sqrt_dt, uint64_t seed, uint64_t walker_offset, int64_t step0,
                               int remove_mean, void* stream) {
  PITA_REQUIRE(temperature > 0.f, "pita_lj_descent: temperature must be > 0");
  PairParams p{};
  p.inv_T = 1.0f / temperature; p.energy_factor = energy_factor; p.dist_eps = dist_eps; p.eps = eps;
  p.rm2 = rm * rm; p.osc_scale = osc_scale;
  p.cw = -p.inv_T * (2.0f * energy_factor * eps * 12.0f / p.rm2); p.co = -p.inv_T * osc_scale;
  return launch_descent<E_LJ>(x, noise, B, n, d, p,
                              descent_params(nsteps, dt, noise_scale, sqrt_dt, seed, walker_offset, step0, remove_mean), stream);
}

extern "C" int pita_dw_descent(float* x, const float* noise, int64_t B, int n, int d, float temperature, float a, float b,
                               float c, float d0, int nsteps, float dt, float noise_scale, float sqrt_dt, uint64_t seed,
                               uint64_t walker_offset, int64_t step0, int remove_mean, void* stream) {
  PITA_REQUIRE(temperature > 0.f, "pita_dw_descent: temperature must be > 0");
  PairParams p{};
  p.inv_T = 1.0f / temperature; p.a = a; p.b = b; p.c = c; p.d0 = d0;
  return launch_descent<E_DW>(x, noise, B, n, d, p,
                              descent_params(nsteps, dt, noise_scale, sqrt_dt, seed, walker_offset, step0, remove_mean), stream);
}

extern "C" size_t pita_lj_mala_workspace_bytes(int nsteps) { return 8 * (size_t)((nsteps > 0 ? nsteps : 0) + 1); }

// shared tail of the fused-chain entry points: zero the per-step counters, run the chain, derive rates and final dt
template <class Chain>
static int run_mala_chain(int nsteps, double* dt_dev, int adaptive, int64_t total, float* rates_out, void* workspace,
                          MalaParams& q, void* stream, Chain&& chain) {
  PITA_REQUIRE(((uintptr_t)workspace & 7) == 0, "fused MALA: workspace must be 8-byte aligned");
  unsigned long long* sync = static_cast<unsigned long long*>(workspace);
  PITA_HIP_CHECK(hipMemsetAsync(sync, 0, pita_lj_mala_workspace_bytes(nsteps), (hipStream_t)stream));
  q.sync = sync;
  q.spin_limit = mala_spin_limit();
  const char* miss = getenv("PITA_DEBUG_MALA_MISSING_BLOCKS");
  q.debug_missing_blocks = (miss && *miss) ? atoi(miss) : 0;
  const int rc = chain();
  if (rc != PITA_OK) return rc;
  return launch_mala_finish(dt_dev, sync, nsteps, (long long)total, adaptive, rates_out, stream);
}

extern "C" int pita_lj_mala(float* x, float* logp, const float* noise, const float* uniforms, int64_t B, int n, int d,
                            float temperature, float energy_factor, float dist_eps, float eps, float rm, float osc_scale,
                            int nsteps, double* dt_dev, int adaptive, int64_t total, uint64_t seed,
                            uint64_t walker_offset, const int64_t* walker_ids, int64_t step0, int remove_mean,
                            float* rates_out, void* workspace, void* stream) {
  PITA_REQUIRE(temperature > 0.f, "pita_lj_mala: temperature must be > 0");
  PITA_REQUIRE(B >= 0 && nsteps >= 0 && total > 0, "pita_lj_mala: bad argument");
  if (!((n == 13 || n == 55) && d == 3))
    return fail(PITA_EUNSUPPORTED, "pita_lj_mala: fused chains exist for LJ13 and LJ55 (n = %d, d = %d)", n, d);
  if (nsteps == 0) return PITA_OK;
  PITA_REQUIRE(dt_dev && workspace && (B == 0 || (x && logp)), "pita_lj_mala: null argument");
  PairParams p{};
  p.inv_T = 1.0f / temperature; p.energy_factor = energy_factor; p.dist_eps = dist_eps; p.eps = eps;
  p.rm2 = rm * rm; p.osc_scale = osc_scale;
  p.cw = -p.inv_T * (2.0f * energy_factor * eps * 12.0f / p.rm2); p.co = -p.inv_T * osc_scale;
  hipStream_t s = (hipStream_t)stream;
  MalaParams q{};
  q.noise = noise; q.uniforms = uniforms; q.walker_ids = (const long long*)walker_ids; q.seed = seed;
  q.walker_offset = walker_offset; q.step0 = step0; q.total = total; q.dt_dev = dt_dev; q.nsteps = nsteps;
  q.adaptive = adaptive; q.remove_mean = remove_mean;
  if (n == 55) {
    return run_mala_chain(nsteps, dt_dev, adaptive, total, rates_out, workspace, q, stream, [&]() {
      if (B == 0) return (int)PITA_OK;
      const int rc = ring_launch_mala(E_LJ, x, logp, B, n, d, p, q, stream);
      return rc == 1 ? fail(PITA_EUNSUPPORTED, "pita_lj_mala: no ring kernel for n = %d", n) : rc;
    });
  }
  const long long nblk = (B + 127) / 128;
  const bool unit = p.rm2 == 1.0f;
  static PerDevice<int> capacity_on;  // co-resident blocks of the chain kernel, per device
  int& capacity = capacity_on.get();
  if (capacity == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    PITA_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lj13_mala_kernel<true>, 256, 0));
    PITA_HIP_CHECK(hipGetDevice(&dev));
    PITA_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    capacity = per_cu * prop.multiProcessorCount;
  }
  return run_mala_chain(nsteps, dt_dev, adaptive, total, rates_out, workspace, q, stream, [&]() {
    if (nblk == 0) return (int)PITA_OK;
    const unsigned grid = (unsigned)(nblk < capacity ? nblk : capacity);
    if (unit) hipLaunchKernelGGL(lj13_mala_kernel<true>, dim3(grid), dim3(256), 0, s, x, logp, (long long)B, p, q);
    else hipLaunchKernelGGL(lj13_mala_kernel<false>, dim3(grid), dim3(256), 0, s, x, logp, (long long)B, p, q);
    PITA_LAUNCH_CHECK();
    return (int)PITA_OK;
  });
}

extern "C" int pita_dw_mala(float* x, float* logp, const float* noise, const float* uniforms, int64_t B, int n, int d,
                            float temperature, float a, float b, float c, float d0, int nsteps, double* dt_dev,
                            int adaptive, int64_t total, uint64_t seed, uint64_t walker_offset,
                            const int64_t* walker_ids, int64_t step0, int remove_mean, float* rates_out, void* workspace,
                            void* stream) {
  PITA_REQUIRE(temperature > 0.f, "pita_dw_mala: temperature must be > 0");
  PITA_REQUIRE(B >= 0 && nsteps >= 0 && total > 0, "pita_dw_mala: bad argument");
  if (!(n == 4 && d == 2))
    return fail(PITA_EUNSUPPORTED, "pita_dw_mala: the fused chain exists for DW4 (n = %d, d = %d)", n, d);
  if (nsteps == 0) return PITA_OK;
  PITA_REQUIRE(dt_dev && workspace && (B == 0 || (x && logp)), "pita_dw_mala: null argument");
  PairParams p{};
  p.inv_T = 1.0f / temperature; p.a = a; p.b = b; p.c = c; p.d0 = d0;
  MalaParams q{};
  q.noise = noise; q.uniforms = uniforms; q.walker_ids = (const long long*)walker_ids; q.seed = seed;
  q.walker_offset = walker_offset; q.step0 = step0; q.total = total; q.dt_dev = dt_dev; q.nsteps = nsteps;
  q.adaptive = adaptive; q.remove_mean = remove_mean;
  return run_mala_chain(nsteps, dt_dev, adaptive, total, rates_out, workspace, q, stream, [&]() {
    if (B == 0) return (int)PITA_OK;
    const int rc = ring_launch_mala(E_DW, x, logp, B, n, d, p, q, stream);
    return rc == 1 ? fail(PITA_EUNSUPPORTED, "pita_dw_mala: no ring kernel for n = %d", n) : rc;
  });
}
