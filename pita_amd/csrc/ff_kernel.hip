// Table-driven classical force field (bonds, angles, periodic torsions, LJ + Coulomb with exceptions and the
// CutoffNonPeriodic reaction field) for gfx950: log-density -E/kT and force per walker.
//
// Replaces the ARITHMETIC behind ALPEnergy.__call__ (pita/src/energies/alp_energy.py:122-149), which the reference
// delegates to OpenMM (amber14-all + implicit/obc1, :93-120) through bgflow's OpenMMBridge.  Neither OpenMM nor the
// PDB / XML parameters are in the reference tree: the functional forms below are those of OpenMM's HarmonicBondForce,
// HarmonicAngleForce, PeriodicTorsionForce, NonbondedForce and GBSAOBCForce (OBC1 + ACE surface area); parameters come
// from the caller as flat tables (e.g. exported from an OpenMM System).  PARITY UNPINNED: checked only against the
// oracle's restatement of the same forms (autograd forces) on synthetic topologies.
//
// Mapping: one thread = one (walker, atom), floor(256/n) walkers per 256-thread block (so even the 4 096 walkers per
// GPU of BASELINE config C4 put a wave on every SIMD); coordinates and tables live in LDS; loads and stores of
// x / force are contiguous spans.  ~14 kflop (+ ~60 kflop with GB-OBC1) and 536 B per walker-eval for a 22-atom
// peptide: compute- rather than bandwidth-bound.
#include "common.h"

namespace pita {

struct FfParams {
  int n, nb, na, nt, np;
  // tables (inside `blob`): bond_idx[nb][2], bond_par[nb][2], angle_idx[na][3], angle_par[na][2], tors_idx[nt][4],
  // tors_par[nt][3], pair_idx[np][2], pair_par[np][4] = (ONE_4PI_EPS0*qq, sigma, 4*eps, is_exception)
  float length_scale, inv_kT, cutoff, krf, crf;
  int use_cutoff;
  int gb;                // GBSAOBCForce (OBC1) present
  // gb_par[n][4] (inside `blob`) = (offset radius rho = R - 0.009, scaled radius s*rho, R, charge)
  float gb_pf, gb_sa, gb_probe;  // -ONE_4PI_EPS0 (1/eps_solute - 1/eps_solvent); 4 pi * surface-area energy; probe radius
  const unsigned* blob;  // all tables, contiguous (each padded to 16 B); staged into LDS once per block
  int blob_words;
  int o_bond_idx, o_bond_par, o_angle_idx, o_angle_par, o_tors_idx, o_tors_par, o_pair_idx, o_pair_par, o_gb_par;  // word offsets
  int o_tors_cs;         // word offset of tors_cs[nt][2] = (cos phase, sin phase), computed on the host in double
  int o_csr_off, o_csr_ent;  // per-atom interaction lists: csr_off[3][n+1] (bonds, angles, torsions), entries (term << 2) | role
  const float* x; float* logp; float* force;
  long long B;
  // fused negative-time / Langevin descent (ff_kernel<true>): all steps of x <- remove_mean(x + F dt + noise_scale sqrt_dt xi)
  // with the walkers LDS-resident; the arithmetic of pita_em_step (sampler_kernels.hip: elem_kernel<3, OP_EM>) op for op
  float* xio;            // [B][3 n] walkers, updated in place
  const float* noise;    // [steps][B][3 n] injected normals, or null: Philox keyed by (seed, global walker, step, atom)
  int steps, remove_mean;
  float dt, noise_scale, sqrt_dt;
  unsigned long long seed, walker_offset;
  long long step0;
};

// One thread = one (walker, atom): it evaluates every interaction its atom takes part in and keeps the atom's gradient
// in registers -- no atomics, no cross-lane reduction of forces (an interaction is evaluated once per participating
// atom: 2x for pairs and bonds, 3x angles, 4x torsions; the alternative, LDS float atomics, measured 10x slower: the
// LDS executes ds_add_f32 at well under one lane per clock).  Per-atom interaction lists (CSR) are built on the host.
constexpr int FF_THREADS = 256;

template <bool DESCENT>
__global__ void __launch_bounds__(FF_THREADS) ff_kernel(FfParams p) {
  extern __shared__ float sm[];
  const int n = p.n, D = 3 * n, WPB = FF_THREADS / n;
  // interaction tables: LDS-resident for the whole (persistent) block
  unsigned* tb = reinterpret_cast<unsigned*>(sm);
  for (int i = threadIdx.x; i < p.blob_words; i += FF_THREADS) tb[i] = p.blob[i];
  const int* bond_idx = reinterpret_cast<const int*>(tb + p.o_bond_idx);
  const float* bond_par = reinterpret_cast<const float*>(tb + p.o_bond_par);
  const int* angle_idx = reinterpret_cast<const int*>(tb + p.o_angle_idx);
  const float* angle_par = reinterpret_cast<const float*>(tb + p.o_angle_par);
  const int* tors_idx = reinterpret_cast<const int*>(tb + p.o_tors_idx);
  const float* tors_par = reinterpret_cast<const float*>(tb + p.o_tors_par);
  const float* tors_cs = reinterpret_cast<const float*>(tb + p.o_tors_cs);  // [nt][2] cos, sin of the phase
  const float* pair_par = reinterpret_cast<const float*>(tb + p.o_pair_par);
  const float* gb_par = reinterpret_cast<const float*>(tb + p.o_gb_par);
  const int* csr_off = reinterpret_cast<const int*>(tb + p.o_csr_off);  // [3][n+1]: bonds, angles, torsions
  const int* csr_ent = reinterpret_cast<const int*>(tb + p.o_csr_ent);  // (term << 2) | role
  float* xs = sm + p.blob_words;   // [WPB][D] coordinates (nm)
  float* gs = xs + WPB * D;        // [WPB][D] gradient, for the coalesced store
  float* es = gs + WPB * D;        // [WPB][n] energy partials
  float* br = es + WPB * n;        // [WPB][n] Born radii
  float* bw = br + WPB * n;        // [WPB][n] dE/d(HCT sum)
  const int n_slots = 2 * p.nb + 3 * p.na + 4 * p.nt;  // gradient slots of the bonded terms: one per (term, participating atom)
  float* tg = bw + WPB * n;        // [WPB][n_slots][3] bonded-term gradients, phase 1 -> phase 2
  float* xu = tg + WPB * 3 * n_slots;  // DESCENT: [WPB][D] walkers in model units, resident over the steps
  float* vv = xu + WPB * D;        // DESCENT: [WPB][D] updated walkers before the centring
  const int tid = threadIdx.x, wl = tid / n, a = tid - wl * n;
  const bool lane_on = wl < WPB;
  const long long nblk = (p.B + WPB - 1) / WPB;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long w0 = blk * WPB;
    const int nw = (int)((p.B - w0) < WPB ? (p.B - w0) : WPB);
    if (DESCENT) {
      for (int q = tid; q < nw * D; q += FF_THREADS) xu[q] = p.xio[w0 * D + q];
      __syncthreads();
    }
    for (int step = 0; step < (DESCENT ? p.steps : 1); ++step) {
    for (int q = tid; q < nw * D; q += FF_THREADS) xs[q] = (DESCENT ? xu[q] : p.x[w0 * D + q]) * p.length_scale;
    __syncthreads();
    const bool act = lane_on && wl < nw;
    const float* xr = xs + (act ? wl : 0) * D;
    const float xa0 = xr[3 * a % D], xa1 = xr[(3 * a + 1) % D], xa2 = xr[(3 * a + 2) % D];
    float E = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
    // ---- bonded terms, two phases (round 6).  Phase 1: one thread per TERM (the walker's threads deal the bonds, angles and
    //      torsions out among themselves) evaluates it ONCE and parks the gradient of every participating atom in LDS;
    //      phase 2: one thread per ATOM adds up its slots through the per-atom lists.  Before, every participating atom
    //      re-evaluated the whole term (angles 3x, torsions 4x, with acosf / atan2f / sinf / cosf each time) and the atom
    //      sitting in the most torsions set the pace of its wave.
    {
      float* tgw = tg + (act ? wl : 0) * (3 * n_slots);
      if (act) {
        // HarmonicBondForce: 1/2 k (r - r0)^2; slots [2 t + role]
        for (int t = a; t < p.nb; t += n) {
          const int i = 3 * bond_idx[2 * t], j = 3 * bond_idx[2 * t + 1];
          const float r0 = bond_par[2 * t], k = bond_par[2 * t + 1];
          const float d0 = xr[i] - xr[j], d1 = xr[i + 1] - xr[j + 1], d2 = xr[i + 2] - xr[j + 2];
          const float r = sqrtf(fmaf(d0, d0, fmaf(d1, d1, d2 * d2)));
          const float dr = r - r0;
          E = fmaf(0.5f * k * dr, dr, E);
          const float c = k * dr / r;
          float* o = tgw + 3 * (2 * t);
          o[0] = c * d0; o[1] = c * d1; o[2] = c * d2;
          o[3] = -c * d0; o[4] = -c * d1; o[5] = -c * d2;
        }
        // HarmonicAngleForce: 1/2 k (theta - theta0)^2; slots [2 nb + 3 t + role]
        for (int t = a; t < p.na; t += n) {
          const int i = 3 * angle_idx[3 * t], j = 3 * angle_idx[3 * t + 1], k3 = 3 * angle_idx[3 * t + 2];
          const float th0 = angle_par[2 * t], k = angle_par[2 * t + 1];
          float av[3], bv[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) { av[c] = xr[i + c] - xr[j + c]; bv[c] = xr[k3 + c] - xr[j + c]; }
          const float aa = fmaf(av[0], av[0], fmaf(av[1], av[1], av[2] * av[2]));
          const float bb = fmaf(bv[0], bv[0], fmaf(bv[1], bv[1], bv[2] * bv[2]));
          const float ab = fmaf(av[0], bv[0], fmaf(av[1], bv[1], av[2] * bv[2]));
          const float inv = 1.0f / sqrtf(aa * bb);
          const float cosv = fminf(fmaxf(ab * inv, -1.0f), 1.0f);
          const float dth = acosf(cosv) - th0;
          E = fmaf(0.5f * k * dth, dth, E);
          const float sinv = fmaxf(sqrtf(fmaf(-cosv, cosv, 1.0f)), 1e-6f);
          const float dEdc = -k * dth / sinv;  // dE/dcos
          float* o = tgw + 3 * (2 * p.nb + 3 * t);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float gi = dEdc * (bv[c] * inv - cosv * av[c] / aa);
            const float gk = dEdc * (av[c] * inv - cosv * bv[c] / bb);
            o[c] = gi; o[3 + c] = -(gi + gk); o[6 + c] = gk;
          }
        }
        // PeriodicTorsionForce: k (1 + cos(n phi - phase)); slots [2 nb + 3 na + 4 t + role].  cos / sin of n phi by the
        // angle-addition recurrence from (cos phi, sin phi) = (x, y) / |(x, y)| -- no atan2f / sinf / cosf per term; the
        // periodicity is an integer in OpenMM; cos / sin of the phase come from the host (tors_cs)
        for (int t = a; t < p.nt; t += n) {
          const int i = 3 * tors_idx[4 * t], j = 3 * tors_idx[4 * t + 1], k3 = 3 * tors_idx[4 * t + 2], l = 3 * tors_idx[4 * t + 3];
          const float per = tors_par[3 * t], k = tors_par[3 * t + 2];
          const float cph = tors_cs[2 * t], sph = tors_cs[2 * t + 1];
          float b1[3], b2[3], b3[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) { b1[c] = xr[j + c] - xr[i + c]; b2[c] = xr[k3 + c] - xr[j + c]; b3[c] = xr[l + c] - xr[k3 + c]; }
          const float n1[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
          const float n2[3] = {b2[1] * b3[2] - b2[2] * b3[1], b2[2] * b3[0] - b2[0] * b3[2], b2[0] * b3[1] - b2[1] * b3[0]};
          const float b22 = fmaf(b2[0], b2[0], fmaf(b2[1], b2[1], b2[2] * b2[2]));
          const float nb2 = sqrtf(b22);
          const float yv = (b1[0] * n2[0] + b1[1] * n2[1] + b1[2] * n2[2]) * nb2;
          const float xv = n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2];
          const float nrm = sqrtf(fmaf(xv, xv, yv * yv));
          const float c1 = nrm > 0.f ? xv / nrm : 1.0f, s1 = nrm > 0.f ? yv / nrm : 0.0f;
          float cn = 1.0f, sn = 0.0f;
          const int nper = (int)per;
          for (int q = 0; q < nper; ++q) {
            const float cc = cn * c1 - sn * s1, ss = sn * c1 + cn * s1;
            cn = cc; sn = ss;
          }
          E += k * (1.0f + (cn * cph + sn * sph));
          const float dEdphi = -k * per * (sn * cph - cn * sph);
          const float n11 = fmaxf(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2], 1e-20f);
          const float n22 = fmaxf(n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2], 1e-20f);
          const float pq = (b1[0] * b2[0] + b1[1] * b2[1] + b1[2] * b2[2]) / b22;
          const float qq = (b3[0] * b2[0] + b3[1] * b2[1] + b3[2] * b2[2]) / b22;
          const float ci = -nb2 / n11, cl = nb2 / n22;
          // d phi / d r of the four atoms: i: ci n1; j: -(pq + 1) ci n1 + qq cl n2; k: pq ci n1 - (qq + 1) cl n2; l: cl n2
          const float wi[4] = {1.0f, -(pq + 1.0f), pq, 0.f}, wlc[4] = {0.f, qq, -(qq + 1.0f), 1.0f};
          float* o = tgw + 3 * (2 * p.nb + 3 * p.na + 4 * t);
#pragma unroll
          for (int role = 0; role < 4; ++role)
#pragma unroll
            for (int c = 0; c < 3; ++c) o[3 * role + c] = dEdphi * (wi[role] * ci * n1[c] + wlc[role] * cl * n2[c]);
        }
      }
      __syncthreads();
      if (act) {
        const int kind_base[3] = {0, 2 * p.nb, 2 * p.nb + 3 * p.na}, arity[3] = {2, 3, 4};
        for (int kind = 0; kind < 3; ++kind)
          for (int e = csr_off[kind * (n + 1) + a]; e < csr_off[kind * (n + 1) + a + 1]; ++e) {
            const int t = csr_ent[e] >> 2, role = csr_ent[e] & 3;
            const float* o = tgw + 3 * (kind_base[kind] + arity[kind] * t + role);
            g0 += o[0]; g1 += o[1]; g2 += o[2];
          }
      }
    }
    if (act) {
      // ---- NonbondedForce: every partner j of atom a (exceptions carry their own parameters)
      for (int j = 0; j < n; ++j) {
        if (j == a) continue;
        const int lo = a < j ? a : j, hi = a < j ? j : a;
        const int t = lo * n - lo * (lo + 1) / 2 + (hi - lo - 1);
        const float qq = pair_par[4 * t], sg = pair_par[4 * t + 1], e4 = pair_par[4 * t + 2], exc = pair_par[4 * t + 3];
        const float d0 = xa0 - xr[3 * j], d1 = xa1 - xr[3 * j + 1], d2 = xa2 - xr[3 * j + 2];
        const float r2 = fmaf(d0, d0, fmaf(d1, d1, d2 * d2));
        const float ir2 = 1.0f / r2, ir = sqrtf(ir2);
        const float s2 = sg * sg * ir2, s6 = s2 * s2 * s2;
        float e = e4 * fmaf(s6, s6, -s6);
        float g = e4 * (-12.0f * s6 * s6 + 6.0f * s6) * ir2;  // (dE/dr)/r
        if (p.use_cutoff && exc == 0.f) {
          e += qq * (ir + p.krf * r2 - p.crf);
          g += qq * (-ir * ir2 + 2.0f * p.krf);
        } else {
          e += qq * ir;
          g += -qq * ir * ir2;
        }
        const float inside = (!p.use_cutoff || r2 < p.cutoff * p.cutoff) ? 1.0f : 0.0f;
        if (a < j) E += inside * e;
        g *= inside;
        g0 = fmaf(g, d0, g0); g1 = fmaf(g, d1, g1); g2 = fmaf(g, d2, g2);
      }
    }
    // ---- GBSAOBCForce, OBC1 (Onufriev-Bashford-Case 2004, model I: alpha, beta, gamma = 0.8, 0, 2.909125) with the
    //      ACE surface-area term, as in OpenMM's reference algorithm: Born radii from the pairwise HCT integral,
    //      the generalised-Born pair sum (self terms included), then the chain rule through the Born radii.
    if (p.gb) {
      const float rc = p.use_cutoff ? p.cutoff : 3.0e38f;
      const float rho = gb_par[4 * a], sga = gb_par[4 * a + 1], R = gb_par[4 * a + 2], qa = gb_par[4 * a + 3];
      float chain = 0.f, Ba = 1.f;
      if (act) {
        float sum = 0.f;
        for (int j = 0; j < n; ++j) {  // HCT integral of atom a
          if (j == a) continue;
          const float sg = gb_par[4 * j + 1];
          const float d0 = xa0 - xr[3 * j], d1 = xa1 - xr[3 * j + 1], d2 = xa2 - xr[3 * j + 2];
          const float r = sqrtf(fmaf(d0, d0, fmaf(d1, d1, d2 * d2)));
          const float rs = r + sg, ir = 1.0f / r;
          const float l = 1.0f / fmaxf(rho, fabsf(r - sg)), u = 1.0f / rs;
          const float l2 = l * l, u2 = u * u;
          float term = l - u + 0.25f * r * (u2 - l2) + 0.5f * ir * logf(u / l) + 0.25f * sg * sg * ir * (l2 - u2);
          if (rho < sg - r) term += 2.0f * (1.0f / rho - l);
          sum += (rho < rs && r < rc) ? term : 0.f;
        }
        const float psi = 0.5f * rho * sum, psi2 = psi * psi;
        const float th = tanhf(fmaf(2.909125f * psi2, psi, 0.8f * psi));
        Ba = 1.0f / (1.0f / rho - th / R);
        chain = Ba * Ba * (1.0f - th * th) * fmaf(3.0f * 2.909125f, psi2, 0.8f) * 0.5f * rho / R;  // dB / d(sum)
        br[wl * n + a] = Ba;
      }
      __syncthreads();
      if (act) {
        // self term and ACE surface area, then the generalised-Born pair sum
        const float rr = R + p.gb_probe, q3 = (R / Ba) * (R / Ba) * (R / Ba);
        const float sa = p.gb_sa * rr * rr * q3 * q3;
        const float eself = 0.5f * p.gb_pf * qa * qa / Ba;
        E += eself + sa;
        float bfa = -(eself + 6.0f * sa) / Ba;  // dE/dB_a
        for (int j = 0; j < n; ++j) {
          if (j == a) continue;
          const float c = p.gb_pf * qa * gb_par[4 * j + 3], Bj = br[wl * n + j];
          const float d0 = xa0 - xr[3 * j], d1 = xa1 - xr[3 * j + 1], d2 = xa2 - xr[3 * j + 2];
          const float r2 = fmaf(d0, d0, fmaf(d1, d1, d2 * d2));
          const float a2 = Ba * Bj, Dv = r2 / (4.0f * a2), ex = expf(-Dv);
          const float if2 = 1.0f / fmaf(a2, ex, r2), G = c * sqrtf(if2);
          const float inside = (r2 < rc * rc) ? 1.0f : 0.f;
          if (a < j) E += inside * (p.use_cutoff ? G - c / p.cutoff : G);
          const float g = inside * (-G * if2 * (1.0f - 0.25f * ex));           // (dG/dr)/r
          bfa = fmaf(inside * (-0.5f * G * if2 * ex * (1.0f + Dv)), Bj, bfa);  // dG/d(Ba Bj) * Bj
          g0 = fmaf(g, d0, g0); g1 = fmaf(g, d1, g1); g2 = fmaf(g, d2, g2);
        }
        bw[wl * n + a] = bfa * chain;  // dE / d(sum_a)
      }
      __syncthreads();
      if (act) {
        const float wa = bw[wl * n + a];
        for (int j = 0; j < n; ++j) {  // chain rule through the Born radii of a (descreened by j) and of j (by a)
          if (j == a) continue;
          const float rhoj = gb_par[4 * j], sgj = gb_par[4 * j + 1], wj = bw[wl * n + j];
          const float d0 = xa0 - xr[3 * j], d1 = xa1 - xr[3 * j + 1], d2 = xa2 - xr[3 * j + 2];
          const float r = sqrtf(fmaf(d0, d0, fmaf(d1, d1, d2 * d2))), ir = 1.0f / r;
          float acc = 0.f;
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            const float ro = side ? rhoj : rho, sg = side ? sga : sgj, w = side ? wj : wa;
            const float rs = r + sg, dl = r - sg;
            const float l = 1.0f / fmaxf(ro, fabsf(dl)), u = 1.0f / rs;
            const float l2 = l * l, u2 = u * u, u3 = u2 * u;
            const float lp = (fabsf(dl) > ro) ? (dl > 0.f ? -l2 : l2) : 0.f;  // dl/dr
            const float lg = logf(u / l);
            float dt = lp + u2 + 0.25f * (u2 - l2) - 0.5f * r * fmaf(l, lp, u3) - 0.5f * lg * ir * ir -
                       0.5f * ir * (u + lp / l) - 0.25f * sg * sg * (l2 - u2) * ir * ir + 0.5f * sg * sg * ir * fmaf(l, lp, u3);
            if (ro < sg - r) dt -= 2.0f * lp;
            acc += (ro < rs && r < rc) ? w * dt : 0.f;
          }
          const float g = acc * ir;
          g0 = fmaf(g, d0, g0); g1 = fmaf(g, d1, g1); g2 = fmaf(g, d2, g2);
        }
      }
    }
    if (act) {
      const float sc = -p.inv_kT * p.length_scale;  // d logp / d x_model
      gs[wl * D + 3 * a] = sc * g0; gs[wl * D + 3 * a + 1] = sc * g1; gs[wl * D + 3 * a + 2] = sc * g2;
      es[wl * n + a] = E;
    }
    __syncthreads();
    if (!DESCENT) {
      if (act && a == 0) {
        float tot = 0.f;
        for (int q = 0; q < n; ++q) tot += es[wl * n + q];
        p.logp[w0 + wl] = -tot * p.inv_kT;
      }
      if (p.force)
        for (int q = tid; q < nw * D; q += FF_THREADS) p.force[w0 * D + q] = gs[q];
    } else {
      // Euler-Maruyama update + centring, exactly as pita_em_step does it from the force tensor: v = x + (F dt + (ns xi) sqrt_dt),
      // mean over the atoms summed in atom order, v -= mean
      float v[3] = {0.f, 0.f, 0.f};
      if (act) {
        float xi[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.noise) {
          const long long base = (((long long)step * p.B + (w0 + wl)) * n + a) * 3;
#pragma unroll
          for (int k = 0; k < 3; ++k) xi[k] = p.noise[base + k];
        } else {
          philox_normal4(p.seed, p.walker_offset + (unsigned long long)(w0 + wl), p.step0 + step, (uint32_t)a, xi);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float dr = gs[wl * D + 3 * a + k], dif = p.noise_scale * xi[k];
          v[k] = xu[wl * D + 3 * a + k] + (dr * p.dt + (dif * p.sqrt_dt));
          vv[wl * D + 3 * a + k] = v[k];
        }
      }
      __syncthreads();
      if (act) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (p.remove_mean) {
            float sum = 0.f;
            for (int q = 0; q < n; ++q) sum += vv[wl * D + 3 * q + k];
            v[k] -= sum / (float)n;
          }
          xu[wl * D + 3 * a + k] = v[k];
        }
      }
    }
    __syncthreads();
    }  // steps
    if (DESCENT) {
      for (int q = tid; q < nw * D; q += FF_THREADS) p.xio[w0 * D + q] = xu[q];
      __syncthreads();
    }
  }
}

}  // namespace pita

struct pita_ff {
  pita::FfParams p{};
  void* d_all = nullptr;
};

using namespace pita;

extern "C" int pita_ff_create(pita_ff_t** out, const pita_ff_config* c) {
  PITA_REQUIRE(out && c, "pita_ff_create: null argument");
  PITA_REQUIRE(c->n_atoms >= 2 && c->n_atoms <= 40, "pita_ff_create: n_atoms must be in [2,40]");
  PITA_REQUIRE(c->charge && c->sigma && c->epsilon, "pita_ff_create: per-atom nonbonded parameters missing");
  PITA_REQUIRE(c->kT > 0.f && c->length_scale > 0.f, "pita_ff_create: kT and length_scale must be > 0");
  PITA_REQUIRE((c->n_bonds == 0 || (c->bond_idx && c->bond_par)) && (c->n_angles == 0 || (c->angle_idx && c->angle_par)) &&
                   (c->n_torsions == 0 || (c->tors_idx && c->tors_par)) && (c->n_exceptions == 0 || (c->exc_idx && c->exc_par)),
               "pita_ff_create: table pointer missing");
  for (int t = 0; t < c->n_torsions; ++t) {  // cos / sin(n phi - phase) come from an angle-addition recurrence over n
    const float per = c->tors_par[3 * t];
    PITA_REQUIRE(per >= 0.f && per <= 12.f && per == (float)(int)per,
                 "pita_ff_create: torsion periodicity must be an integer in [0, 12] (OpenMM PeriodicTorsionForce)");
  }
  const int n = c->n_atoms, np = n * (n - 1) / 2;
  auto check_idx = [&](const int* idx, int cnt) {
    for (int i = 0; i < cnt; ++i)
      if (idx[i] < 0 || idx[i] >= n) return false;
    return true;
  };
  PITA_REQUIRE(check_idx(c->bond_idx, 2 * c->n_bonds) && check_idx(c->angle_idx, 3 * c->n_angles) &&
                   check_idx(c->tors_idx, 4 * c->n_torsions) && check_idx(c->exc_idx, 2 * c->n_exceptions),
               "pita_ff_create: atom index out of range");
  // dense pair table with Lorentz-Berthelot mixing, overridden by the exceptions
  int* pidx = new int[2 * np];
  float* ppar = new float[4 * np];
  const float K = 138.935456f;
  int t = 0;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j, ++t) {
      pidx[2 * t] = i; pidx[2 * t + 1] = j;
      ppar[4 * t] = K * c->charge[i] * c->charge[j];
      ppar[4 * t + 1] = 0.5f * (c->sigma[i] + c->sigma[j]);
      ppar[4 * t + 2] = 4.0f * sqrtf(c->epsilon[i] * c->epsilon[j]);
      ppar[4 * t + 3] = 0.f;
    }
  for (int e = 0; e < c->n_exceptions; ++e) {
    int a = c->exc_idx[2 * e], b = c->exc_idx[2 * e + 1];
    if (a > b) { int tmp = a; a = b; b = tmp; }
    if (a == b) { delete[] pidx; delete[] ppar; return fail(PITA_EINVAL, "pita_ff_create: exception with i == j"); }
    const int tt = a * n - a * (a + 1) / 2 + (b - a - 1);
    ppar[4 * tt] = K * c->exc_par[3 * e];
    ppar[4 * tt + 1] = c->exc_par[3 * e + 1];
    ppar[4 * tt + 2] = 4.0f * c->exc_par[3 * e + 2];
    ppar[4 * tt + 3] = 1.f;
  }
  const size_t b_bi = sizeof(int) * 2 * c->n_bonds, b_bp = sizeof(float) * 2 * c->n_bonds;
  const size_t b_ai = sizeof(int) * 3 * c->n_angles, b_ap = sizeof(float) * 2 * c->n_angles;
  const size_t b_ti = sizeof(int) * 4 * c->n_torsions, b_tp = sizeof(float) * 3 * c->n_torsions;
  const size_t b_pi = sizeof(int) * 2 * np, b_pp = sizeof(float) * 4 * np;
  const bool gb = c->gb_radius != nullptr;
  PITA_REQUIRE(!gb || c->gb_scale, "pita_ff_create: gb_scale missing");
  PITA_REQUIRE(!gb || (c->gb_solute_dielectric > 0.f && c->gb_solvent_dielectric > 0.f),
               "pita_ff_create: GB dielectric constants must be > 0");
  float* gpar = new float[4 * n];
  for (int i = 0; i < n && gb; ++i) {
    if (!(c->gb_radius[i] > 0.009f)) { delete[] pidx; delete[] ppar; delete[] gpar; return fail(PITA_EINVAL, "pita_ff_create: gb_radius must exceed the 0.009 nm dielectric offset"); }
    gpar[4 * i] = c->gb_radius[i] - 0.009f;
    gpar[4 * i + 1] = (c->gb_radius[i] - 0.009f) * c->gb_scale[i];
    gpar[4 * i + 2] = c->gb_radius[i];
    gpar[4 * i + 3] = c->charge[i];
  }
  const size_t b_gb = gb ? sizeof(float) * 4 * n : 0;
  // per-atom interaction lists
  const int n_ent = 2 * c->n_bonds + 3 * c->n_angles + 4 * c->n_torsions;
  int* csr_off = new int[3 * (n + 1)];
  int* csr_ent = new int[n_ent > 0 ? n_ent : 1];
  {
    int pos = 0;
    const int* idx[3] = {c->bond_idx, c->angle_idx, c->tors_idx};
    const int cnt[3] = {c->n_bonds, c->n_angles, c->n_torsions}, arity[3] = {2, 3, 4};
    for (int kind = 0; kind < 3; ++kind) {
      for (int a = 0; a < n; ++a) {
        csr_off[kind * (n + 1) + a] = pos;
        for (int tt = 0; tt < cnt[kind]; ++tt)
          for (int role = 0; role < arity[kind]; ++role)
            if (idx[kind][arity[kind] * tt + role] == a) csr_ent[pos++] = (tt << 2) | role;
      }
      csr_off[kind * (n + 1) + n] = pos;
    }
  }
  const size_t b_co = sizeof(int) * 3 * (n + 1), b_ce = sizeof(int) * (n_ent > 0 ? n_ent : 1);
  const size_t b_tc = sizeof(float) * 2 * c->n_torsions;
  float* tcs = new float[2 * (c->n_torsions > 0 ? c->n_torsions : 1)];
  for (int t = 0; t < c->n_torsions; ++t) {
    tcs[2 * t] = (float)cos((double)c->tors_par[3 * t + 1]);
    tcs[2 * t + 1] = (float)sin((double)c->tors_par[3 * t + 1]);
  }
  const size_t total = b_bi + b_bp + b_ai + b_ap + b_ti + b_tp + b_tc + b_pi + b_pp + b_gb + b_co + b_ce + 16 * 12;  // each table padded to 16 B
  pita_ff* ff = new pita_ff();
  hipError_t e = hipMalloc(&ff->d_all, total);
  char* base = static_cast<char*>(ff->d_all);
  size_t off = 0;
  auto put = [&](const void* src, size_t bytes) -> const void* {
    const void* dst = base + off;
    if (bytes && e == hipSuccess) e = hipMemcpy(base + off, src, bytes, hipMemcpyHostToDevice);
    off += (bytes + 15) & ~size_t(15);
    return dst;
  };
  FfParams& p = ff->p;
  auto word_off = [&](const void* q) { return (int)((static_cast<const char*>(q) - base) / 4); };
  if (e == hipSuccess) {
    p.o_bond_idx = word_off(put(c->bond_idx, b_bi)); p.o_bond_par = word_off(put(c->bond_par, b_bp));
    p.o_angle_idx = word_off(put(c->angle_idx, b_ai)); p.o_angle_par = word_off(put(c->angle_par, b_ap));
    p.o_tors_idx = word_off(put(c->tors_idx, b_ti)); p.o_tors_par = word_off(put(c->tors_par, b_tp));
    p.o_tors_cs = word_off(put(tcs, b_tc));
    p.o_pair_idx = word_off(put(pidx, b_pi)); p.o_pair_par = word_off(put(ppar, b_pp));
    p.o_gb_par = word_off(put(gpar, b_gb));
    p.o_csr_off = word_off(put(csr_off, b_co)); p.o_csr_ent = word_off(put(csr_ent, b_ce));
    p.blob = reinterpret_cast<const unsigned*>(base);
    p.blob_words = (int)(off / 4);
  }
  delete[] pidx;
  delete[] ppar;
  delete[] gpar;
  delete[] csr_off;
  delete[] csr_ent;
  delete[] tcs;
  if (e != hipSuccess) {
    (void)hipFree(ff->d_all);
    delete ff;
    return fail(PITA_EHIP, "pita_ff_create: device upload failed: %s", hipGetErrorString(e));
  }
  p.n = n; p.nb = c->n_bonds; p.na = c->n_angles; p.nt = c->n_torsions; p.np = np;
  p.length_scale = c->length_scale; p.inv_kT = 1.0f / c->kT;
  p.use_cutoff = c->use_cutoff; p.cutoff = c->cutoff;
  p.gb = gb ? 1 : 0;
  if (gb) {
    p.gb_pf = -K * (1.0f / c->gb_solute_dielectric - 1.0f / c->gb_solvent_dielectric);
    p.gb_sa = c->gb_surface_area_factor;
    p.gb_probe = 0.14f;
  }
  if (c->use_cutoff) {
    PITA_REQUIRE(c->cutoff > 0.f, "pita_ff_create: cutoff must be > 0");
    const float er = c->rf_dielectric;
    p.krf = (1.0f / (c->cutoff * c->cutoff * c->cutoff)) * (er - 1.0f) / (2.0f * er + 1.0f);
    p.crf = (1.0f / c->cutoff) * (3.0f * er) / (2.0f * er + 1.0f);
  }
  *out = ff;
  return PITA_OK;
}

extern "C" int pita_ff_destroy(pita_ff_t* ff) {
  if (!ff) return PITA_OK;
  (void)hipFree(ff->d_all);
  delete ff;
  return PITA_OK;
}

extern "C" int pita_ff_logp_force(pita_ff_t* ff, const float* x, float* logp, float* force, int64_t B, void* stream) {
  PITA_REQUIRE(ff && B >= 0, "pita_ff_logp_force: bad argument");
  if (B == 0) return PITA_OK;
  PITA_REQUIRE(x && logp, "pita_ff_logp_force: null argument");
  FfParams p = ff->p;
  p.x = x; p.logp = logp; p.force = force; p.B = B;
  const int WPB = FF_THREADS / p.n;
  const size_t lds = sizeof(float) * ((size_t)p.blob_words + (size_t)WPB * (2 * 3 * p.n + 3 * p.n + 3 * (2 * p.nb + 3 * p.na + 4 * p.nt)));
  PITA_REQUIRE(lds <= 64 * 1024, "pita_ff_logp_force: interaction tables do not fit in LDS");
  const long long nblk = (B + WPB - 1) / WPB;
  const long long cap = 256LL * 8;  // persistent blocks: the tables are staged once per block
  hipLaunchKernelGGL(ff_kernel<false>, dim3((unsigned)(nblk < cap ? nblk : cap)), dim3(FF_THREADS), lds, (hipStream_t)stream, p);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

extern "C" int pita_ff_descent(pita_ff_t* ff, float* x, const float* noise, int64_t B, int n_steps, float dt,
                               float noise_scale, float sqrt_dt, uint64_t seed, uint64_t walker_offset, int64_t step0,
                               int remove_mean, void* stream) {
  PITA_REQUIRE(ff && B >= 0 && n_steps >= 0, "pita_ff_descent: bad argument");
  if (B == 0 || n_steps == 0) return PITA_OK;
  PITA_REQUIRE(x, "pita_ff_descent: null argument");
  FfParams p = ff->p;
  p.xio = x; p.noise = noise; p.B = B; p.steps = n_steps; p.dt = dt; p.noise_scale = noise_scale; p.sqrt_dt = sqrt_dt;
  p.seed = seed; p.walker_offset = walker_offset; p.step0 = step0; p.remove_mean = remove_mean;
  const int WPB = FF_THREADS / p.n;
  const size_t lds = sizeof(float) * ((size_t)p.blob_words + (size_t)WPB * (4 * 3 * p.n + 3 * p.n + 3 * (2 * p.nb + 3 * p.na + 4 * p.nt)));
  PITA_REQUIRE(lds <= 64 * 1024, "pita_ff_descent: interaction tables do not fit in LDS");
  const long long nblk = (B + WPB - 1) / WPB;
  const long long cap = 256LL * 8;
  hipLaunchKernelGGL(ff_kernel<true>, dim3((unsigned)(nblk < cap ? nblk : cap)), dim3(FF_THREADS), lds, (hipStream_t)stream, p);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}
