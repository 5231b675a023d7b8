// Parameter blocks shared by the pair-target kernels (energy_kernels.hip, ring_kernels.hip).
#pragma once
#include "common.h"

namespace pita {

enum { E_LJ = 0, E_DW = 1, E_LJS = 2 };  // E_LJS: LJ with the reference's cubic core below range_min (smooth=True)
template <int KIND> constexpr bool is_lj() { return KIND == E_LJ || KIND == E_LJS; }

struct PairParams {
  float inv_T, energy_factor, dist_eps, eps, rm2, osc_scale;  // LJ
  float cw, co;  // LJ fast paths: -inv_T * 24 ef eps / rm^2 (pair force weight), -inv_T * osc_scale
  float a, b, c, d0;                                          // DW
  float sm_min, sc0, sc1, sc2, sc3;  // E_LJS: below r = sm_min the pair energy is sc0 u^3 + sc1 u^2 + sc2 u + sc3, u = r - sm_min
};

struct DescentParams {
  float dt, noise_scale, sqrt_dt;
  int nsteps, remove_mean;
  unsigned long long seed, walker_offset;
  long long step0;
};

struct MalaParams {
  const float* noise;      // nullable [nsteps, B, n*d]
  const float* uniforms;   // nullable [nsteps, B]
  const long long* walker_ids;
  unsigned long long seed, walker_offset;
  long long step0, total;
  const double* dt_dev;
  int nsteps, adaptive, remove_mean;
  int spin_limit;            // bound of the adaptive chain's grid-barrier spin (polls); see mala_spin_limit()
  int debug_missing_blocks;  // tests only (PITA_DEBUG_MALA_MISSING_BLOCKS): the barrier waits for this many blocks that never come
  unsigned long long* sync;  // [nsteps + 1]: per step (blocks arrived << 32 | walkers accepted), error flag; zeroed by the wrapper
};

// Polls a block may spend in the per-step grid barrier of an adaptive chain before it raises the error flag
// sync[nsteps] (the launch then poisons dt / rates with NaN and the caller reruns the launch-per-kernel chain).
// PITA_DEBUG_MALA_SPIN_LIMIT overrides it (tests force the timeout path with 0).
int mala_spin_limit();

#ifdef __HIPCC__
// One block's bounded wait for the `nblocks` arrivals of step s (sync[s] = arrivals << 32 | accepted walkers); returns the
// counter as last read.  A wait that runs out raises the chain's error flag sync[nsteps].  The flag is read before the
// first poll and every 64 polls, so once ANY block has given up every other wait of the launch ends at once: a chain
// whose grid turned out not to be co-resident costs one timeout, not one per remaining step and block.
__device__ __forceinline__ unsigned long long mala_grid_wait(unsigned long long* sync, int s, int nsteps,
                                                             unsigned long long nblocks, int spin_limit) {
  unsigned long long v = 0;
  int spins = 0;
  while (((v = __hip_atomic_load(&sync[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) < nblocks) {
    if ((spins & 63) == 0 && __hip_atomic_load(&sync[nsteps], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    if (++spins > spin_limit) {  // never hang the device
      __hip_atomic_store(&sync[nsteps], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  return v;
}
#endif

// rates_out[s] and the final dt from the per-step counts of a fused chain; NaN everywhere when the chain flagged a
// barrier timeout (energy_kernels.hip)
int launch_mala_finish(double* dt_dev, const unsigned long long* sync, int nsteps, long long total, int adaptive,
                       float* rates_out, void* stream);

// Ring kernels (ring_kernels.hip): compile-time particle count, a walker's particles on the lanes of one wavefront,
// partner coordinates and partner forces by immediate-offset LDS reads / one cross-lane permute per pair.
// Return PITA_OK when they took the call, 1 when no instantiation covers (n, d, kind, parameters).
int ring_launch_energy(int kind, const float* x, float* logp, float* force, int64_t B, int n, int d, const PairParams& p,
                       void* stream);
int ring_launch_descent(int kind, float* x, const float* noise, int64_t B, int n, int d, const PairParams& p,
                        const DescentParams& q, void* stream);
int ring_launch_mala(int kind, float* x, float* logp, int64_t B, int n, int d, const PairParams& p, MalaParams q,
                     void* stream);

}  // namespace pita
