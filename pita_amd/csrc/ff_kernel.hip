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
// Mapping: one lane = one walker; a wave stages its 64 walkers' coordinates in LDS (row stride odd -> the per-lane row
// accesses of a uniformly indexed atom are bank-conflict free), walks the interaction tables with wave-uniform
// (scalar) loads, and accumulates the energy gradient into a second LDS row per walker (only the owning lane touches
// its row: no atomics).  Loads and stores of x / force are contiguous spans.  ~14 kflop and 536 B per walker-eval for a
// 22-atom peptide: latency- rather than bandwidth-bound at the 16 384-walker batches of BASELINE config C4.
#include "common.h"

namespace pita {

struct FfParams {
  int n, nb, na, nt, np;
  const int* bond_idx; const float* bond_par;
  const int* angle_idx; const float* angle_par;
  const int* tors_idx; const float* tors_par;
  const int* pair_idx; const float* pair_par;  // [np][2] ; [np][4] = (ONE_4PI_EPS0*qq, sigma, 4*eps, is_exception)
  float length_scale, inv_kT, cutoff, krf, crf;
  int use_cutoff;
  int gb;                // GBSAOBCForce (OBC1) present
  const float* gb_par;   // [n][4] = (offset radius rho = R - 0.009, scaled radius s*rho, R, charge)
  float gb_pf, gb_sa, gb_probe;  // -ONE_4PI_EPS0 (1/eps_solute - 1/eps_solvent); 4 pi * surface-area energy; probe radius
  const float* x; float* logp; float* force;
  long long B;
};

__global__ void __launch_bounds__(64) ff_kernel(FfParams p) {
  extern __shared__ float sm[];
  const int D = 3 * p.n, S = D | 1, SB = p.n | 1;
  float* xs = sm;
  float* gs = sm + 64 * S;
  float* bs = gs + 64 * S;  // GB: Born radii | chain factors | dE/dB, three [64][SB] tables
  const int lane = threadIdx.x;
  const long long nblk = (p.B + 63) / 64;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long w0 = blk * 64;
    const int nw = (int)((p.B - w0) < 64 ? (p.B - w0) : 64);
    for (int q = lane; q < 64 * D; q += 64) {
      const int w = q / D, c = q - w * D;
      xs[w * S + c] = (w < nw) ? p.x[w0 * D + q] * p.length_scale : (float)(c % 7) * 0.37f;  // dummy rows stay finite
      gs[w * S + c] = 0.f;
    }
    __syncthreads();
    const float* xr = xs + lane * S;
    float* gr = gs + lane * S;
    float E = 0.f;
    // ---- HarmonicBondForce: 1/2 k (r - r0)^2
    for (int t = 0; t < p.nb; ++t) {
      const int i = 3 * p.bond_idx[2 * t], j = 3 * p.bond_idx[2 * t + 1];
      const float r0 = p.bond_par[2 * t], k = p.bond_par[2 * t + 1];
      const float d0 = xr[i] - xr[j], d1 = xr[i + 1] - xr[j + 1], d2 = xr[i + 2] - xr[j + 2];
      const float r = sqrtf(fmaf(d0, d0, fmaf(d1, d1, d2 * d2)));
      const float dr = r - r0;
      E = fmaf(0.5f * k * dr, dr, E);
      const float c = k * dr / r;
      gr[i] += c * d0; gr[i + 1] += c * d1; gr[i + 2] += c * d2;
      gr[j] -= c * d0; gr[j + 1] -= c * d1; gr[j + 2] -= c * d2;
    }
    // ---- HarmonicAngleForce: 1/2 k (theta - theta0)^2
    for (int t = 0; t < p.na; ++t) {
      const int i = 3 * p.angle_idx[3 * t], j = 3 * p.angle_idx[3 * t + 1], k3 = 3 * p.angle_idx[3 * t + 2];
      const float th0 = p.angle_par[2 * t], k = p.angle_par[2 * t + 1];
      float a[3], b[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { a[c] = xr[i + c] - xr[j + c]; b[c] = xr[k3 + c] - xr[j + c]; }
      const float aa = fmaf(a[0], a[0], fmaf(a[1], a[1], a[2] * a[2])), bb = fmaf(b[0], b[0], fmaf(b[1], b[1], b[2] * b[2]));
      const float ab = fmaf(a[0], b[0], fmaf(a[1], b[1], a[2] * b[2]));
      const float inv = 1.0f / sqrtf(aa * bb);
      const float cosv = fminf(fmaxf(ab * inv, -1.0f), 1.0f);
      const float th = acosf(cosv);
      const float dth = th - th0;
      E = fmaf(0.5f * k * dth, dth, E);
      const float sinv = fmaxf(sqrtf(fmaf(-cosv, cosv, 1.0f)), 1e-6f);
      const float dEdc = -k * dth / sinv;  // dE/dcos
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float gi = dEdc * (b[c] * inv - cosv * a[c] / aa);
        const float gk = dEdc * (a[c] * inv - cosv * b[c] / bb);
        gr[i + c] += gi; gr[k3 + c] += gk; gr[j + c] -= gi + gk;
      }
    }
    // ---- PeriodicTorsionForce: k (1 + cos(n phi - phase))
    for (int t = 0; t < p.nt; ++t) {
      const int i = 3 * p.tors_idx[4 * t], j = 3 * p.tors_idx[4 * t + 1], k3 = 3 * p.tors_idx[4 * t + 2], l = 3 * p.tors_idx[4 * t + 3];
      const float per = p.tors_par[3 * t], ph = p.tors_par[3 * t + 1], k = p.tors_par[3 * t + 2];
      float b1[3], b2[3], b3[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { b1[c] = xr[j + c] - xr[i + c]; b2[c] = xr[k3 + c] - xr[j + c]; b3[c] = xr[l + c] - xr[k3 + c]; }
      const float n1[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
      const float n2[3] = {b2[1] * b3[2] - b2[2] * b3[1], b2[2] * b3[0] - b2[0] * b3[2], b2[0] * b3[1] - b2[1] * b3[0]};
      const float b22 = fmaf(b2[0], b2[0], fmaf(b2[1], b2[1], b2[2] * b2[2]));
      const float nb2 = sqrtf(b22);
      const float yv = (b1[0] * n2[0] + b1[1] * n2[1] + b1[2] * n2[2]) * nb2;
      const float xv = n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2];
      const float phi = atan2f(yv, xv);
      const float ang = fmaf(per, phi, -ph);
      E += k * (1.0f + cosf(ang));
      const float dEdphi = -k * per * sinf(ang);
      const float n11 = fmaxf(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2], 1e-20f);
      const float n22 = fmaxf(n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2], 1e-20f);
      const float pq = (b1[0] * b2[0] + b1[1] * b2[1] + b1[2] * b2[2]) / b22;
      const float qq = (b3[0] * b2[0] + b3[1] * b2[1] + b3[2] * b2[2]) / b22;
      const float ci = -nb2 / n11, cl = nb2 / n22;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float di = ci * n1[c], dl = cl * n2[c];
        gr[i + c] += dEdphi * di;
        gr[l + c] += dEdphi * dl;
        gr[j + c] += dEdphi * (-(pq + 1.0f) * di + qq * dl);
        gr[k3 + c] += dEdphi * (-(qq + 1.0f) * dl + pq * di);
      }
    }
    // ---- NonbondedForce: all pairs i<j (exceptions carry their own parameters)
    for (int t = 0; t < p.np; ++t) {
      const int i = 3 * p.pair_idx[2 * t], j = 3 * p.pair_idx[2 * t + 1];
      const float qq = p.pair_par[4 * t], sg = p.pair_par[4 * t + 1], e4 = p.pair_par[4 * t + 2], exc = p.pair_par[4 * t + 3];
      const float d0 = xr[i] - xr[j], d1 = xr[i + 1] - xr[j + 1], d2 = xr[i + 2] - xr[j + 2];
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
      E += inside * e;
      g *= inside;
      gr[i] += g * d0; gr[i + 1] += g * d1; gr[i + 2] += g * d2;
      gr[j] -= g * d0; gr[j + 1] -= g * d1; gr[j + 2] -= g * d2;
    }
    // ---- GBSAOBCForce, OBC1 (Onufriev-Bashford-Case 2004, model I: alpha, beta, gamma = 0.8, 0, 2.909125) with the
    //      ACE surface-area term, as in OpenMM's reference algorithm: Born radii from the pairwise HCT integral,
    //      the generalised-Born pair sum (self terms included), then the chain rule through the Born radii.
    if (p.gb) {
      float* br = bs + lane * SB;
      float* bc = br + 64 * SB;
      float* bf = bc + 64 * SB;
      const float rc = p.use_cutoff ? p.cutoff : 3.0e38f;
      for (int i = 0; i < p.n; ++i) {
        const float rho = p.gb_par[4 * i], R = p.gb_par[4 * i + 2];
        float sum = 0.f;
        for (int j = 0; j < p.n; ++j) {
          if (j == i) continue;
          const float sg = p.gb_par[4 * j + 1];
          const float d0 = xr[3 * i] - xr[3 * j], d1 = xr[3 * i + 1] - xr[3 * j + 1], d2 = xr[3 * i + 2] - xr[3 * j + 2];
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
        const float B = 1.0f / (1.0f / rho - th / R);
        br[i] = B;
        bc[i] = B * B * (1.0f - th * th) * fmaf(3.0f * 2.909125f, psi2, 0.8f) * 0.5f * rho / R;  // dB / d(sum)
        bf[i] = 0.f;
      }
      for (int i = 0; i < p.n; ++i) {
        const float qi = p.gb_pf * p.gb_par[4 * i + 3], Bi = br[i], R = p.gb_par[4 * i + 2];
        // self term and ACE surface area
        const float rr = R + p.gb_probe, q3 = (R / Bi) * (R / Bi) * (R / Bi);
        const float sa = p.gb_sa * rr * rr * q3 * q3;
        const float eself = 0.5f * qi * p.gb_par[4 * i + 3] / Bi;
        E += eself + sa;
        float bfi = -(eself + 6.0f * sa) / Bi;
        for (int j = i + 1; j < p.n; ++j) {
          const float c = qi * p.gb_par[4 * j + 3], Bj = br[j];
          const float d0 = xr[3 * i] - xr[3 * j], d1 = xr[3 * i + 1] - xr[3 * j + 1], d2 = xr[3 * i + 2] - xr[3 * j + 2];
          const float r2 = fmaf(d0, d0, fmaf(d1, d1, d2 * d2));
          const float a2 = Bi * Bj, Dv = r2 / (4.0f * a2), ex = expf(-Dv);
          const float if2 = 1.0f / fmaf(a2, ex, r2), G = c * sqrtf(if2);
          const float inside = (r2 < rc * rc) ? 1.0f : 0.f;
          E += inside * (p.use_cutoff ? G - c / p.cutoff : G);
          const float g = inside * (-G * if2 * (1.0f - 0.25f * ex));        // (dG/dr)/r
          const float da = inside * (-0.5f * G * if2 * ex * (1.0f + Dv));    // dG/d(Bi Bj)
          gr[3 * i] += g * d0; gr[3 * i + 1] += g * d1; gr[3 * i + 2] += g * d2;
          gr[3 * j] -= g * d0; gr[3 * j + 1] -= g * d1; gr[3 * j + 2] -= g * d2;
          bfi = fmaf(da, Bj, bfi);
          bf[j] = fmaf(da, Bi, bf[j]);
        }
        bf[i] += bfi;
      }
      for (int i = 0; i < p.n; ++i) {
        const float rho = p.gb_par[4 * i];
        const float wi = bf[i] * bc[i];  // dE / d(sum_i)
        for (int j = 0; j < p.n; ++j) {
          if (j == i) continue;
          const float sg = p.gb_par[4 * j + 1];
          const float d0 = xr[3 * i] - xr[3 * j], d1 = xr[3 * i + 1] - xr[3 * j + 1], d2 = xr[3 * i + 2] - xr[3 * j + 2];
          const float r = sqrtf(fmaf(d0, d0, fmaf(d1, d1, d2 * d2)));
          const float rs = r + sg, ir = 1.0f / r, dl = r - sg;
          const float l = 1.0f / fmaxf(rho, fabsf(dl)), u = 1.0f / rs;
          const float l2 = l * l, u2 = u * u, u3 = u2 * u;
          const float lp = (fabsf(dl) > rho) ? (dl > 0.f ? -l2 : l2) : 0.f;  // dl/dr
          const float lg = logf(u / l);
          float dt = lp + u2 + 0.25f * (u2 - l2) - 0.5f * r * fmaf(l, lp, u3) - 0.5f * lg * ir * ir -
                     0.5f * ir * (u + lp / l) - 0.25f * sg * sg * (l2 - u2) * ir * ir + 0.5f * sg * sg * ir * fmaf(l, lp, u3);
          if (rho < sg - r) dt -= 2.0f * lp;
          const float g = (rho < rs && r < rc) ? wi * dt * ir : 0.f;
          gr[3 * i] += g * d0; gr[3 * i + 1] += g * d1; gr[3 * i + 2] += g * d2;
          gr[3 * j] -= g * d0; gr[3 * j + 1] -= g * d1; gr[3 * j + 2] -= g * d2;
        }
      }
    }
    if (lane < nw) p.logp[w0 + lane] = -E * p.inv_kT;
    __syncthreads();
    if (p.force) {
      const float sc = -p.inv_kT * p.length_scale;  // d logp / d x_model
      for (int q = lane; q < nw * D; q += 64) {
        const int w = q / D, c = q - w * D;
        p.force[w0 * D + q] = sc * gs[w * S + c];
      }
    }
    __syncthreads();
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
  PITA_REQUIRE(c->n_atoms >= 2 && c->n_atoms <= 40, "pita_ff_create: n_atoms must be in [2,40] (LDS rows of 64 walkers)");
  PITA_REQUIRE(c->charge && c->sigma && c->epsilon, "pita_ff_create: per-atom nonbonded parameters missing");
  PITA_REQUIRE(c->kT > 0.f && c->length_scale > 0.f, "pita_ff_create: kT and length_scale must be > 0");
  PITA_REQUIRE((c->n_bonds == 0 || (c->bond_idx && c->bond_par)) && (c->n_angles == 0 || (c->angle_idx && c->angle_par)) &&
                   (c->n_torsions == 0 || (c->tors_idx && c->tors_par)) && (c->n_exceptions == 0 || (c->exc_idx && c->exc_par)),
               "pita_ff_create: table pointer missing");
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
  const size_t total = b_bi + b_bp + b_ai + b_ap + b_ti + b_tp + b_pi + b_pp + b_gb + 16 * 9;  // each table padded to 16 B
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
  if (e == hipSuccess) {
    p.bond_idx = (const int*)put(c->bond_idx, b_bi); p.bond_par = (const float*)put(c->bond_par, b_bp);
    p.angle_idx = (const int*)put(c->angle_idx, b_ai); p.angle_par = (const float*)put(c->angle_par, b_ap);
    p.tors_idx = (const int*)put(c->tors_idx, b_ti); p.tors_par = (const float*)put(c->tors_par, b_tp);
    p.pair_idx = (const int*)put(pidx, b_pi); p.pair_par = (const float*)put(ppar, b_pp);
    p.gb_par = (const float*)put(gpar, b_gb);
  }
  delete[] pidx;
  delete[] ppar;
  delete[] gpar;
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
  const int S = (3 * p.n) | 1, SB = p.n | 1;
  const size_t lds = sizeof(float) * (2 * 64 * S + (p.gb ? 3 * 64 * SB : 0));
  const long long nblk = (B + 63) / 64;
  hipLaunchKernelGGL(ff_kernel, dim3((unsigned)(nblk < 8192 ? nblk : 8192)), dim3(64), lds, (hipStream_t)stream, p);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}
