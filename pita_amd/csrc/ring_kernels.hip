// Pair targets with a compile-time particle count: one walker's particles on the lanes of ONE wavefront ("ring"
// kernels), for LJ55 (config C5) and DW4 (config C2).
//
// Replaces, like energy_kernels.hip (paths relative to /root/reference/):
//   pita/src/energies/lennardjones_energy.py:121-155,213-227  LennardJonesPotential._energy / LennardJonesEnergy.__call__
//   DW4: bgflow.MultiDoubleWellPotential (not in the reference tree, see oracle header)
//   pita/src/models/components/sde_integration.py:28-45,353-470  negative_time_descent, mala_proposal,
//                                                                metropolis_hastings_mala(_adaptive)
//
// Mapping.  Lane l of a wavefront is particle i = l mod N of walker l / N (LJ55: one walker per wave, 55 of 64 lanes;
// DW4: 16 walkers per wave); lanes past the last whole walker shadow the first particles again and are masked out of
// sums and stores.  Every unordered pair is evaluated ONCE, as the circulant (i, i + dd mod N), dd = 1 .. (N-1)/2:
//   * lane i keeps its own coordinates and force in registers;
//   * the walker's coordinates sit in LDS twice in a row, [2N][4] floats, so particle (i + dd) mod N is the entry at
//     i + dd and particle (i - dd) mod N the entry at i + N - dd: one ds_read_b128 each at an IMMEDIATE offset from the
//     lane's own entry -- no address arithmetic in the pair loop;
//   * the pair (i, i+dd) gives lane i the scalar coef = e'(r)/r; lane j = i + dd needs the same scalar (Newton's third
//     law): ONE ds_bpermute_b32 per pair from a per-lane address table computed once per kernel, instead of three
//     read-modify-writes of a partner force table (the generic kernel, energy_kernels.hip: pair_force_n3l); the
//     receiving lane forms x_j - x_i itself -- bitwise the negative of the sender's difference, so forces sum to zero
//     exactly as with an explicit hand-over;
//   * for even N the antipodal distance N/2 pairs every particle with one partner: both lanes evaluate it (identical
//     bits), the energy is counted by the lower half.
// Per pair and lane: 2 LDS reads, 1 permute, ~23 vector instructions + v_rcp_f32 (LJ) -- 684 vector instructions per LJ55
// walker-evaluation, and that count is the bound: PMC on the final kernel gives 3 300 SIMD cycles per walker with the
// vector pipe busy 0.86 of them (one fp32 wave instruction per ~4.2 cycles); without the permutes and second reads
// (wrong results, timing only) 41 us instead of 46 at 32 768 walkers.  A packed-fp32 pair loop (two distances per
// v_pk_fma_f32 on a structure-of-arrays table, 281 packed + 87 plain instructions) measured 54 us: the packed operations
// issue at half rate on this chip, as tools/ubench/isa_rates.hip says, so the 157 TFLOP/s vector peak -- which counts
// them at full rate -- is not reachable by this arithmetic; against the plain-FMA rate the kernel sits at 0.38.  Sums over a walker's particles
// that belong to THIS file's arithmetic only (oscillator centre, energy) are DPP tree reductions; sums whose order is
// shared with the per-step kernels (remove_mean of pita_em_step, the proposal densities and centring of
// pita_mala_accept: sequential over the particles from 0.f) keep that order, carried by a few lanes that walk the LDS
// table while the others idle -- so the fused loops below are bit-identical to the launch-per-kernel chains.
#include "pair_common.h"

namespace pita {

namespace {

#ifndef RING_DD_GROUP
#define RING_DD_GROUP 2
#endif

__device__ __forceinline__ void wfence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// one table entry as ONE ds_read_b128 (4 LDS cycles per wave instruction; left to itself the compiler narrows the
// read to the three components in use, ds_read_b96: 8 cycles -- the pair loop is then bound by the LDS pipe)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 read_entry(const float* p) {
  // (volatile keeps the 16-byte read whole; the explicit LDS address space keeps it a ds_read -- address-space
  // inference skips volatile accesses and would leave a flat load)
  typedef const volatile f32x4_t __attribute__((address_space(3))) * lds_entry_ptr;
  const f32x4_t v = *(lds_entry_ptr)(p);
  float4 o;
  o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
  return o;
}

// v + (v moved across lanes by the DPP control); lanes without a source add 0
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int m = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
  return v + __builtin_bit_cast(float, m);
}
// sum over the 64 lanes in a fixed tree order, returned to every lane (through an SGPR)
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xb1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4e>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x124>(v);  // row_ror 4
  v = dpp_add<0x128>(v);  // row_ror 8
  v = dpp_add<0x142>(v);  // row_bcast 15
  v = dpp_add<0x143>(v);  // row_bcast 31 -> lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sum over each aligned group of four lanes, in every lane of the group
__device__ __forceinline__ float quad_sum(float v) {
  v = dpp_add<0xb1>(v);
  return dpp_add<0x4e>(v);
}
// sum over each DPP row (16 aligned lanes), in every lane of the row
__device__ __forceinline__ float row_sum(float v) {
  v = dpp_add<0xb1>(v);
  v = dpp_add<0x4e>(v);
  v = dpp_add<0x124>(v);  // row_ror 4
  return dpp_add<0x128>(v);  // row_ror 8
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

template <int N, int DIM>
struct Ring {
  static_assert(N >= 2 && N <= 64 && DIM >= 1 && DIM <= 3, "ring kernels: a walker must fit one wavefront");
  // a walker owns an aligned group of STRIDE lanes -- a quad, a DPP row or the whole wave -- so that the sums over
  // its particles are DPP reductions (quad_sum / row_sum / wave_sum); lanes N .. STRIDE-1 of a group shadow its first
  // particles (valid reads, identical arithmetic) and are masked out of sums and stores
  static constexpr int STRIDE = N <= 4 ? 4 : N <= 16 ? 16 : 64;
  static constexpr int WPW = 64 / STRIDE;                // walkers per wavefront
  static constexpr int NH = (N - 1) / 2;                 // circulant distances with two distinct partners
  static constexpr bool EVEN = (N % 2) == 0;             // + the antipodal distance N / 2
  static constexpr int TAB_F = WPW * 2 * N * 4;          // floats of one coordinate table of a wave
  static_assert(2 * N >= STRIDE, "a shadow lane must map onto a particle");

  int wl, i;      // walker of the wave, particle
  bool real;      // false: a shadow lane past the walker's last particle
  int bp[NH > 0 ? NH : 1];  // ds_bpermute byte address of the lane holding particle (i - dd) mod N of this walker

  __device__ explicit Ring(int lane) {
    wl = lane / STRIDE;
    i = lane - wl * STRIDE;
    real = i < N;
    if (!real) i -= N;
#pragma unroll
    for (int dd = 1; dd <= NH; ++dd) bp[dd - 1] = 4 * (wl * STRIDE + (i >= dd ? i - dd : i - dd + N));
  }
  // sum of v over the walker's lanes (callers zero the shadow lanes), in every lane of the walker
  static __device__ __forceinline__ float walker_sum(float v);
  // this lane's first-copy entry of a wave's table
  __device__ float* entry(float* tab) const { return tab + (wl * 2 * N + i) * 4; }
  __device__ float* walker_tab(float* tab) const { return tab + wl * 2 * N * 4; }

  __device__ void put(float* tab, const float (&v)[DIM], float pad = 0.f) const {
    float4 e;
    e.x = v[0]; e.y = DIM > 1 ? v[1] : 0.f; e.z = DIM > 2 ? v[2] : 0.f; e.w = pad;
    float* t = entry(tab);
    *reinterpret_cast<float4*>(t) = e;
    *reinterpret_cast<float4*>(t + N * 4) = e;
  }
};

template <int N, int DIM>
__device__ __forceinline__ float Ring<N, DIM>::walker_sum(float v) {
  return STRIDE == 4 ? quad_sum(v) : STRIDE == 16 ? row_sum(v) : wave_sum(v);
}

// f = sum over the partners j of coef_ij (x_i - x_j), coef = e'(r)/r (LJ: in units of 12 eps / rm^2);
// e = this lane's share of the pair energy: the pairs (i, i + dd), LJ in units of eps -- accumulated as
// sum s^12 - 2 sum s^6 (two instructions per pair instead of three; the two sums of a lane's <= 32 pairs are O(30) near
// equilibrium, so the subtraction costs ~1e-6 of the lane's share: inside the fp32 rounding of the 55-lane total).
template <int N, int DIM, int KIND, bool WANT_E, bool UNIT_RM>
__device__ __forceinline__ void ring_pairs(const Ring<N, DIM>& r, const float (&xi)[DIM], const float* te,
                                           const PairParams& p, float (&f)[DIM], float& e) {
  using R = Ring<N, DIM>;
#pragma unroll
  for (int k = 0; k < DIM; ++k) f[k] = 0.f;
  e = 0.f;
  float e12 = 0.f, e6 = 0.f;
#pragma unroll
  for (int dd = 1; dd <= R::NH + (R::EVEN ? 1 : 0); ++dd) {
    if ((dd - 1) % RING_DD_GROUP == 0) {
      // let the scheduler overlap a few distances, but neither hoist all N - 1 table reads nor sink the force sums
      // (its default: every difference vector stays live to the end, > 256 registers)
#pragma unroll
      for (int k = 0; k < DIM; ++k) asm volatile("" : "+v"(f[k]));
      asm volatile("" : "+v"(e), "+v"(e12), "+v"(e6)::"memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    const bool antipodal = R::EVEN && dd == N / 2;
    const float4 xj4 = read_entry(te + dd * 4);
    const float xj[3] = {xj4.x, xj4.y, xj4.z};
    float d[DIM], r2 = (KIND == E_LJ) ? p.dist_eps : 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      d[k] = xi[k] - xj[k];
      r2 = fmaf(d[k], d[k], r2);
    }
    float coef, ep = 0.f;
    if (KIND == E_LJ) {
      const float inv = __builtin_amdgcn_rcpf(r2);
      const float s2 = UNIT_RM ? inv : p.rm2 * inv;   // (rm / r)^2
      const float s6 = s2 * s2 * s2;
      if (WANT_E) {                                   // (rm/r)^12 and (rm/r)^6 sums
        const float s6e = (antipodal && r.i >= N / 2) ? 0.f : s6;
        e12 = fmaf(s6e, s6e, e12);
        e6 += s6e;
      }
      const float ts = s6 * s2;
      coef = fmaf(-s6, ts, ts);                       // (s^6 - s^12) s^2 = e'(r)/r * rm^2 / (12 eps)
    } else {
      const float dist = sqrtf(r2);
      const float u = dist - p.d0, u2 = u * u;
      ep = fmaf(p.a * u2, u2, fmaf(p.b, u2, p.c));
      coef = (u * fmaf(4.0f * p.a, u2, 2.0f * p.b)) / dist;
    }
    if (WANT_E && KIND != E_LJ) e += (antipodal && r.i >= N / 2) ? 0.f : ep;
#pragma unroll
    for (int k = 0; k < DIM; ++k) f[k] = fmaf(coef, d[k], f[k]);
    if (!antipodal) {
      // the pair (i - dd, i): its coefficient comes from the lane that evaluated it, the difference is formed here
      const float cb = __builtin_bit_cast(
          float, __builtin_amdgcn_ds_bpermute(r.bp[dd - 1], __builtin_bit_cast(int, coef)));
      const float4 xm4 = read_entry(te + (N - dd) * 4);
      const float xm[3] = {xm4.x, xm4.y, xm4.z};
#pragma unroll
      for (int k = 0; k < DIM; ++k) f[k] = fmaf(cb, xi[k] - xm[k], f[k]);
    }
  }
  if (WANT_E && KIND == E_LJ) e = fmaf(-2.0f, e6, e12);
}

// raw pair sums -> d logp / dx and log-density (lennardjones_energy.py:125-151: ordered pairs, i.e. every unordered
// pair twice, + the harmonic oscillator about the particle mean; DW: every unordered pair once)
template <int N, int DIM, int KIND, bool WANT_E>
__device__ __forceinline__ void ring_finish(const Ring<N, DIM>& r, const float (&xi)[DIM], const PairParams& p,
                                            float (&f)[DIM], float e, float& logp) {
  if (KIND == E_LJ) {
    float c[DIM], osc = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      const float mean = Ring<N, DIM>::walker_sum(r.real ? xi[k] : 0.f) * (1.0f / (float)N);
      c[k] = xi[k] - mean;
      osc = fmaf(c[k], c[k], osc);
      f[k] = fmaf(p.cw, f[k], p.co * c[k]);
    }
    if (WANT_E) {
      const float v = fmaf(2.0f * p.energy_factor * p.eps, e, 0.5f * p.osc_scale * osc);
      logp = -p.inv_T * Ring<N, DIM>::walker_sum(r.real ? v : 0.f);
    }
  } else {
#pragma unroll
    for (int k = 0; k < DIM; ++k) f[k] = -p.inv_T * f[k];
    if (WANT_E) logp = -p.inv_T * Ring<N, DIM>::walker_sum(r.real ? e : 0.f);
  }
}

template <int N, int DIM>
__device__ __forceinline__ void load_x(const Ring<N, DIM>& r, const float* x, long long w, float (&xi)[DIM]) {
  const float* src = x + (w * N + r.i) * DIM;
#pragma unroll
  for (int k = 0; k < DIM; ++k) xi[k] = src[k];
}

// sum over the particles j = 0 .. N-1 of comp k of the walker's table, sequentially from 0.f: the order of
// elem_kernel / mala_accept_kernel (sampler_kernels.hip).  One walker per wave: lanes k < DIM walk the table, the
// result reaches every lane through SGPRs; several walkers per wave (N = 4): every lane sums for itself.
template <int N, int DIM>
__device__ __forceinline__ void seq_sums(const Ring<N, DIM>& r, const float* tab, int lane, float (&s)[DIM]) {
  if (Ring<N, DIM>::WPW == 1) {
    const float* cb = tab + (lane < DIM ? lane : 0);
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) a += cb[j * 4];
#pragma unroll
    for (int k = 0; k < DIM; ++k) s[k] = lane_bcast(a, k);
  } else {
    const float* tw = tab + r.wl * 2 * N * 4;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < N; ++j) a += tw[j * 4 + k];
      s[k] = a;
    }
  }
}

// ---------------------------------------------------------------------------- log-density + force
template <int N, int DIM, int KIND, bool UNIT_RM>
__global__ void __launch_bounds__(256) ring_energy_kernel(const float* __restrict__ x, float* __restrict__ logp,
                                                          float* __restrict__ force, long long B, PairParams p) {
  using R = Ring<N, DIM>;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* tab = sm + wave * R::TAB_F;
  const R r(lane);
  const long long ngroups = (B + R::WPW - 1) / R::WPW, nwaves = (long long)gridDim.x * 4;
  // the next group's coordinates are fetched while the current one is being worked on (a wave owns several groups
  // when the batch exceeds the resident waves; without the prefetch each pays the full HBM latency in sequence)
  float xn[DIM];
  long long g = (long long)blockIdx.x * 4 + wave;
  if (g < ngroups) {
    const long long w = g * R::WPW + r.wl;
    load_x<N, DIM>(r, x, w < B ? w : B - 1, xn);
  }
  for (; g < ngroups; g += nwaves) {
    const long long w = g * R::WPW + r.wl;
    const bool act = r.real && w < B;
    float xi[DIM], f[DIM], e, lp = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) xi[k] = xn[k];
    if (g + nwaves < ngroups) {
      const long long wn = (g + nwaves) * R::WPW + r.wl;
      load_x<N, DIM>(r, x, wn < B ? wn : B - 1, xn);
    }
    r.put(tab, xi);
    wfence();
    ring_pairs<N, DIM, KIND, true, UNIT_RM>(r, xi, r.entry(tab), p, f, e);
    ring_finish<N, DIM, KIND, true>(r, xi, p, f, e, lp);
    if (act) {
      if (force) {
        float* dst = force + (w * N + r.i) * DIM;
#pragma unroll
        for (int k = 0; k < DIM; ++k) dst[k] = f[k];
      }
      if (r.i == 0) logp[w] = lp;
    }
    wfence();
  }
}

// ---------------------------------------------------------------------------- fused descent
// S steps of x <- remove_mean(x + F(x) dt + noise_scale sqrt_dt xi) in one launch (sde_integration.py:353-360); the
// walker stays in registers + its LDS table, HBM sees one read and one write of x per launch.  Arithmetic of
// ring_energy_kernel followed by pita_em_step (elem_kernel): bit-identical to the per-step path.
template <int N, int DIM, int KIND, bool UNIT_RM>
__global__ void __launch_bounds__(256) ring_descent_kernel(float* __restrict__ x, const float* __restrict__ noise,
                                                           long long B, PairParams p, DescentParams q) {
  using R = Ring<N, DIM>;
  // the accept counters and seq_sums' shadow-lane handling below are written for the quad and the whole-wave mappings;
  // the DPP-row mapping (13 particles on 16 lanes) is validated for the energy kernel only
  static_assert(R::STRIDE == 4 || R::STRIDE == 64, "descent / MALA chains: quad or whole-wave walkers only");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* tab = sm + wave * R::TAB_F;
  const R r(lane);
  const long long ngroups = (B + R::WPW - 1) / R::WPW, nwaves = (long long)gridDim.x * 4;
  for (long long g = (long long)blockIdx.x * 4 + wave; g < ngroups; g += nwaves) {
    const long long w = g * R::WPW + r.wl;
    const bool act = r.real && w < B;
    const long long ws = w < B ? w : B - 1;
    float xi[DIM];
    load_x<N, DIM>(r, x, ws, xi);
    r.put(tab, xi);
    wfence();
    for (int s = 0; s < q.nsteps; ++s) {
      float f[DIM], e, lp;
      ring_pairs<N, DIM, KIND, false, UNIT_RM>(r, xi, r.entry(tab), p, f, e);
      ring_finish<N, DIM, KIND, false>(r, xi, p, f, e, lp);
      float nz[4] = {0.f, 0.f, 0.f, 0.f};
      if (q.noise_scale != 0.f) {
        if (noise) {
#pragma unroll
          for (int k = 0; k < DIM; ++k) nz[k] = noise[(((long long)s * B + ws) * N + r.i) * DIM + k];
        } else {
          philox_normal4(q.seed, q.walker_offset + (unsigned long long)ws, q.step0 + s, (uint32_t)r.i, nz);
        }
      }
      float v[DIM];
#pragma unroll
      for (int k = 0; k < DIM; ++k) v[k] = xi[k] + (f[k] * q.dt + ((q.noise_scale * nz[k]) * q.sqrt_dt));
      if (q.remove_mean) {
        r.put(tab, v);
        wfence();
        float sum[DIM];
        seq_sums<N, DIM>(r, tab, lane, sum);
#pragma unroll
        for (int k = 0; k < DIM; ++k) v[k] -= sum[k] / (float)N;
      }
#pragma unroll
      for (int k = 0; k < DIM; ++k) xi[k] = v[k];
      r.put(tab, xi);
      wfence();
    }
    if (act) {
      float* dst = x + (w * N + r.i) * DIM;
#pragma unroll
      for (int k = 0; k < DIM; ++k) dst[k] = xi[k];
    }
  }
}

// ---------------------------------------------------------------------------- fused MALA chain
// All post_mcmc_steps of metropolis_hastings_mala(_adaptive) (sde_integration.py:362-470) in one launch; arithmetic and
// summation orders of ring_energy_kernel + mala_propose_kernel + ring_energy_kernel + mala_accept_kernel +
// mala_adapt_kernel: bit-identical to the launch-per-kernel chain.
//   * non-adaptive, or adaptive with at most one walker group per wave: a wave keeps its walkers on chip for all steps;
//   * adaptive with more groups than resident waves (LJ55 at 32 768 walkers per GPU): steps outside, groups inside,
//     the walkers make one HBM round trip per step (1.3 KB per LJ55 walker against ~2 x 1 500 pairs of arithmetic).
// The adaptive step size needs the global acceptance count of a step before the next one: one grid-wide barrier per
// step made of ONE relaxed agent-scope atomic add of (1 << 32 | accepted) per block and a relaxed polling load (see
// lj13_mala_kernel); the grid never exceeds the co-resident capacity (launch wrapper); a spin that runs out raises
// sync[nsteps] and the finish kernel poisons dt and the rates with NaN.
template <int N, int DIM, int KIND, bool UNIT_RM>
__global__ void __launch_bounds__(256) ring_mala_kernel(float* __restrict__ x, float* __restrict__ logp, long long B,
                                                        PairParams p, MalaParams q) {
  using R = Ring<N, DIM>;
  // the accept counters and seq_sums' shadow-lane handling below are written for the quad and the whole-wave mappings;
  // the DPP-row mapping (13 particles on 16 lanes) is validated for the energy kernel only
  static_assert(R::STRIDE == 4 || R::STRIDE == 64, "descent / MALA chains: quad or whole-wave walkers only");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int cnt[4];
  __shared__ int total_acc;
  // non-adaptive chains only count: per block and step in LDS, flushed once at the end (a global atomic per wave and
  // step -- 32 768 x nsteps adds on nsteps addresses for LJ55 -- serialises in the L2 and costs more than the chain)
  constexpr int STEP_CNT = 512;
  __shared__ int stepcnt[STEP_CNT];
  for (int t = threadIdx.x; t < STEP_CNT; t += 256) stepcnt[t] = 0;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* tab1 = sm + wave * 2 * R::TAB_F;  // current walkers; pad slot of an entry: this particle's |x' - fwd mean|^2
  float* tab2 = tab1 + R::TAB_F;           // proposals;       pad slot: |x - bwd mean|^2
  const R r(lane);
  const long long ngroups = (B + R::WPW - 1) / R::WPW, nwaves = (long long)gridDim.x * 4;
  const long long g0 = (long long)blockIdx.x * 4 + wave;
  const bool roundtrip = q.adaptive && ngroups > nwaves;  // steps outside, groups inside
  double dt = q.dt_dev[0];
  float xi[DIM], lp = 0.f;

  // one MALA step of the walker group in (xi, lp, tab1); returns the number of accepted walkers of this wave
  auto step = [&](long long ws, bool act, int s) -> int {
    const float hdt = (float)(0.5 * dt), sdt = (float)sqrt(dt), tdt = (float)(2.0 * dt);
    float F[DIM], Fp[DIM], e, lpp = 0.f, dummy;
    ring_pairs<N, DIM, KIND, false, UNIT_RM>(r, xi, r.entry(tab1), p, F, e);
    ring_finish<N, DIM, KIND, false>(r, xi, p, F, e, dummy);
    const unsigned long long key = q.walker_ids ? (unsigned long long)q.walker_ids[ws] : q.walker_offset + (unsigned long long)ws;
    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    if (q.noise) {
#pragma unroll
      for (int k = 0; k < DIM; ++k) nz[k] = q.noise[(((long long)s * B + ws) * N + r.i) * DIM + k];
    } else {
      philox_normal4(q.seed, key, q.step0 + s, (uint32_t)r.i, nz);
    }
    float xp[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) xp[k] = (xi[k] + hdt * F[k]) + sdt * nz[k];  // mala_propose_kernel
    r.put(tab2, xp);
    wfence();
    ring_pairs<N, DIM, KIND, true, UNIT_RM>(r, xp, r.entry(tab2), p, Fp, e);
    ring_finish<N, DIM, KIND, true>(r, xp, p, Fp, e, lpp);
    float sfi = 0.f, sbi = 0.f;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {  // mala_accept_kernel
      const float df = xp[k] - (xi[k] + hdt * F[k]);
      const float db = xi[k] - (xp[k] + hdt * Fp[k]);
      sfi += df * df;
      sbi += db * db;
    }
    r.entry(tab1)[3] = sfi;
    r.entry(tab2)[3] = sbi;
    wfence();
    // sequential sums over the particles: sf, sb, and -- while the chain lanes are at it -- the coordinate sums of both
    // candidates (the centring below needs the one the accept decision picks)
    float sf, sb, mo[DIM], mp[DIM];
    bool have_means = true;
    if (R::WPW == 1) {
      const int c = lane;
      const float* cb = c < DIM ? tab1 + c : c < 2 * DIM ? tab2 + (c - DIM) : c == 2 * DIM ? tab1 + 3 : tab2 + 3;
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < N; ++j) a += cb[j * 4];
#pragma unroll
      for (int k = 0; k < DIM; ++k) { mo[k] = lane_bcast(a, k); mp[k] = lane_bcast(a, DIM + k); }
      sf = lane_bcast(a, 2 * DIM);
      sb = lane_bcast(a, 2 * DIM + 1);
    } else {
      const float* t1 = r.walker_tab(tab1);
      const float* t2 = r.walker_tab(tab2);
      sf = 0.f; sb = 0.f;
#pragma unroll
      for (int j = 0; j < N; ++j) { sf += t1[j * 4 + 3]; sb += t2[j * 4 + 3]; }
      have_means = false;
    }
    const float lqf = -sf / tdt, lqb = -sb / tdt;
    const float ratio = (lpp - lp) + (lqb - lqf);
    const float u = q.uniforms ? q.uniforms[(long long)s * B + ws] : philox_uniform(q.seed, key, q.step0 + s, 0xFFFFFu);
    const float af = (logf(u) < ratio) ? 1.0f : 0.0f;
    lp = af * lpp + (1.0f - af) * lp;
    float v[DIM];
    bool fin = true;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      v[k] = af * xp[k] + (1.0f - af) * xi[k];
      fin = fin && __builtin_isfinite(xp[k]) && __builtin_isfinite(xi[k]);
    }
    if (q.remove_mean) {
      float sum[DIM];
      // af is 0 or 1: the blend IS one of the two candidates (up to the sign of a zero, which no sum can see) unless a
      // coordinate is not finite -- then sum the blended values themselves, like mala_accept_kernel
      if (have_means && !__any(!fin)) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) sum[k] = af != 0.f ? mp[k] : mo[k];
      } else {
        r.put(tab2, v);
        wfence();
        seq_sums<N, DIM>(r, tab2, lane, sum);
      }
#pragma unroll
      for (int k = 0; k < DIM; ++k) v[k] -= sum[k] / (float)N;
    }
#pragma unroll
    for (int k = 0; k < DIM; ++k) xi[k] = v[k];
    r.put(tab1, xi);
    wfence();
    return __popcll(__ballot(act && r.i == 0 && af != 0.f));
  };

  // grid-wide exchange of a step's acceptance count (adaptive chains)
  auto exchange = [&](int s, int acc_wave) {
    if (lane == 0) cnt[wave] = acc_wave;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long c = (unsigned long long)(cnt[0] + cnt[1] + cnt[2] + cnt[3]);
      __hip_atomic_fetch_add(&q.sync[s], (1ull << 32) | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long v = mala_grid_wait(q.sync, s, q.nsteps, (unsigned long long)gridDim.x + q.debug_missing_blocks,
                                                  q.spin_limit);
      total_acc = (int)(v & 0xFFFFFFFFull);
    }
    __syncthreads();
    const float rate = (float)total_acc / (float)q.total;
    dt = ((double)rate > 0.55) ? dt * 1.1 : dt / 1.1;  // sde_integration.py:439-443
  };

  auto load = [&](long long ws) {
    load_x<N, DIM>(r, x, ws, xi);
    lp = logp[ws];
    r.put(tab1, xi);
    wfence();
  };
  auto store = [&](long long w, bool act) {
    if (act) {
      float* dst = x + (w * N + r.i) * DIM;
#pragma unroll
      for (int k = 0; k < DIM; ++k) dst[k] = xi[k];
      if (r.i == 0) logp[w] = lp;
    }
  };

  if (roundtrip) {
    for (int s = 0; s < q.nsteps; ++s) {
      int acc_wave = 0;
      for (long long g = g0; g < ngroups; g += nwaves) {
        const long long w = g * R::WPW + r.wl;
        const bool act = r.real && w < B;
        const long long ws = w < B ? w : B - 1;
        load(ws);
        acc_wave += step(ws, act, s);
        store(w, act);
      }
      exchange(s, acc_wave);  // (a walker is re-read by the wave that wrote it: no fence needed across the barrier)
    }
  } else {
    // q.adaptive here means: one group per wave at most, every wave takes part in every step's barrier
    const long long gmax = q.adaptive ? g0 + 1 : ngroups;
    for (long long g = g0; g < gmax; g += nwaves) {
      const bool have = g < ngroups;
      const long long w = (have ? g : 0) * R::WPW + r.wl;
      const bool act = have && r.real && w < B;
      const long long ws = w < B ? w : B - 1;
      load(ws);
      for (int s = 0; s < q.nsteps; ++s) {
        const int acc = step(ws, act, s);
        if (q.adaptive) {
          exchange(s, acc);
        } else if (lane == 0 && acc) {
          if (s < STEP_CNT) atomicAdd(&stepcnt[s], acc);
          else __hip_atomic_fetch_add(&q.sync[s], (unsigned long long)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      store(w, act);
    }
    if (!q.adaptive) {
      __syncthreads();
      for (int t = threadIdx.x; t < STEP_CNT && t < q.nsteps; t += 256)
        if (stepcnt[t])
          __hip_atomic_fetch_add(&q.sync[t], (unsigned long long)stepcnt[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int N, int DIM, int KIND>
struct RingLaunch {
  using R = Ring<N, DIM>;
  static int cus() {
    static PerDevice<int> n_on;
    int& n = n_on.get();
    if (n == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
      if (n <= 0) n = 256;
    }
    return n;
  }
  static unsigned grid_for(long long B, int blocks_per_cu) {
    const long long ngroups = (B + R::WPW - 1) / R::WPW, want = (ngroups + 3) / 4, cap = (long long)cus() * blocks_per_cu;
    return (unsigned)(want < cap ? want : cap);
  }
  // resident blocks per CU of a kernel (registers and LDS decide): the persistent grids are exactly one residency
  template <class K>
  static int resident(K kernel, size_t lds, int& cache) {
    if (cache == 0) {
      int v = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kernel, 256, lds) != hipSuccess || v <= 0) v = 4;
      cache = v;
    }
    return cache;
  }
  static int energy(const float* x, float* logp, float* force, long long B, const PairParams& p, hipStream_t s) {
    const size_t lds = sizeof(float) * 4 * R::TAB_F;
    static PerDevice<int> occ_on[2];
    int* occ[2] = {&occ_on[0].get(), &occ_on[1].get()};
    if (KIND == E_LJ && p.rm2 == 1.0f) {
      const unsigned grid = grid_for(B, resident(ring_energy_kernel<N, DIM, KIND, true>, lds, *occ[1]));
      hipLaunchKernelGGL((ring_energy_kernel<N, DIM, KIND, true>), dim3(grid), dim3(256), lds, s, x, logp, force, B, p);
    } else {
      const unsigned grid = grid_for(B, resident(ring_energy_kernel<N, DIM, KIND, false>, lds, *occ[0]));
      hipLaunchKernelGGL((ring_energy_kernel<N, DIM, KIND, false>), dim3(grid), dim3(256), lds, s, x, logp, force, B, p);
    }
    PITA_LAUNCH_CHECK();
    return PITA_OK;
  }
  static int descent(float* x, const float* noise, long long B, const PairParams& p, const DescentParams& q, hipStream_t s) {
    const size_t lds = sizeof(float) * 4 * R::TAB_F;
    static PerDevice<int> occ_on[2];
    int* occ[2] = {&occ_on[0].get(), &occ_on[1].get()};
    if (KIND == E_LJ && p.rm2 == 1.0f) {
      const unsigned grid = grid_for(B, resident(ring_descent_kernel<N, DIM, KIND, true>, lds, *occ[1]));
      hipLaunchKernelGGL((ring_descent_kernel<N, DIM, KIND, true>), dim3(grid), dim3(256), lds, s, x, noise, B, p, q);
    } else {
      const unsigned grid = grid_for(B, resident(ring_descent_kernel<N, DIM, KIND, false>, lds, *occ[0]));
      hipLaunchKernelGGL((ring_descent_kernel<N, DIM, KIND, false>), dim3(grid), dim3(256), lds, s, x, noise, B, p, q);
    }
    PITA_LAUNCH_CHECK();
    return PITA_OK;
  }
  static int mala(float* x, float* logp, long long B, const PairParams& p, const MalaParams& q, hipStream_t s) {
    const size_t lds = sizeof(float) * 4 * 2 * R::TAB_F;
    const bool unit = KIND == E_LJ && p.rm2 == 1.0f;
    static PerDevice<int> per_cu_on[2];  // co-resident blocks per CU of the chain kernel, per device
    int& slot = per_cu_on[unit ? 1 : 0].get();
    if (slot == 0) {
      int v = 0;
      if (unit) PITA_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, ring_mala_kernel<N, DIM, KIND, true>, 256, lds));
      else PITA_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, ring_mala_kernel<N, DIM, KIND, false>, 256, lds));
      PITA_REQUIRE(v > 0, "ring_mala: the chain kernel does not fit a compute unit");
      slot = v;
    }
    const unsigned grid = grid_for(B, slot);
    if (unit) hipLaunchKernelGGL((ring_mala_kernel<N, DIM, KIND, true>), dim3(grid), dim3(256), lds, s, x, logp, B, p, q);
    else hipLaunchKernelGGL((ring_mala_kernel<N, DIM, KIND, false>), dim3(grid), dim3(256), lds, s, x, logp, B, p, q);
    PITA_LAUNCH_CHECK();
    return PITA_OK;
  }
};

}  // namespace

int ring_launch_energy(int kind, const float* x, float* logp, float* force, int64_t B, int n, int d, const PairParams& p,
                       void* stream) {
  if (kind == E_LJ && n == 55 && d == 3) return RingLaunch<55, 3, E_LJ>::energy(x, logp, force, B, p, (hipStream_t)stream);
  if (kind == E_LJ && n == 13 && d == 3) return RingLaunch<13, 3, E_LJ>::energy(x, logp, force, B, p, (hipStream_t)stream);
  if (kind == E_DW && n == 4 && d == 2) return RingLaunch<4, 2, E_DW>::energy(x, logp, force, B, p, (hipStream_t)stream);
  return 1;
}

int ring_launch_descent(int kind, float* x, const float* noise, int64_t B, int n, int d, const PairParams& p,
                        const DescentParams& q, void* stream) {
  if (kind == E_LJ && n == 55 && d == 3) return RingLaunch<55, 3, E_LJ>::descent(x, noise, B, p, q, (hipStream_t)stream);
  if (kind == E_DW && n == 4 && d == 2) return RingLaunch<4, 2, E_DW>::descent(x, noise, B, p, q, (hipStream_t)stream);
  return 1;
}

int ring_launch_mala(int kind, float* x, float* logp, int64_t B, int n, int d, const PairParams& p, MalaParams q,
                     void* stream) {
  if (kind == E_LJ && n == 55 && d == 3) return RingLaunch<55, 3, E_LJ>::mala(x, logp, B, p, q, (hipStream_t)stream);
  if (kind == E_DW && n == 4 && d == 2) return RingLaunch<4, 2, E_DW>::mala(x, logp, B, p, q, (hipStream_t)stream);
  return 1;
}

}  // namespace pita
