/*
 * pita_hip.h -- C ABI of libpita_hip.so: MI355X (gfx950) kernels for the inference-time
 * annealed reverse-SDE sampling path of PITA (taraak/pita).
 *
 * The reference has no FFI (it is 100% Python); its "plugin API" is the Python classes
 * WeightedSDEIntegrator / VEReverseSDE / ScoreNet / EGNN_dynamics / MyMLP /
 * LennardJonesEnergy / GMM / Prior.  pita_amd/ mirrors those classes and binds the entry
 * points below through ctypes (INTEGRATION.md shows the binding).  Each entry point cites
 * the reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer named x/out/logp/force/... is a DEVICE pointer owned by the caller
 *     (torch tensors); the library never allocates per call and never synchronises;
 *   - `stream` is a hipStream_t passed as void*; all work is asynchronous on it;
 *   - walker-major row-major layout [B, D] with D = n_particles * n_dim, fp32, exactly the
 *     reference's tensor layout (no transposes at the boundary);
 *   - every function returns 0 on success or a negative PITA_E* code; pita_last_error()
 *     gives the message.  Nothing throws, nothing aborts;
 *   - handles (pita_egnn_t, pita_mlp_t, pita_ff_t) live on the device that was current when they were
 *     created; calls that take an EGNN handle make that device current for their duration (its scratch
 *     buffers -- reverse-mode checkpoints, the f16 path's walker backup, divergence marks, the primal
 *     cache -- grow on demand on THAT device) and restore the caller's.  A handle's scratch is shared by
 *     its launches: use one handle from ONE stream at a time (one handle per stream for concurrency).
 */
#ifndef PITA_HIP_H
#define PITA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PITA_ABI_VERSION 11

enum {
  PITA_OK = 0,
  PITA_EINVAL = -1,      /* bad argument (null pointer, unsupported shape ...) */
  PITA_EUNSUPPORTED = -2,/* configuration the HIP path does not implement */
  PITA_EHIP = -3,        /* a HIP runtime call failed */
  PITA_ENOMEM = -4
};

int pita_abi_version(void);
/* copies the calling thread's last error message into buf (NUL terminated); returns its length */
int pita_last_error(char* buf, size_t buflen);
/* number of visible HIP devices, or a negative error */
int pita_device_count(void);

/* ---------------------------------------------------------------- target energies (K1-K3)
 * All return log-density = -E/T (the reference convention, base_energy_function.py:149) and,
 * when force != NULL, force = d logp / dx.
 */

/* Lennard-Jones cluster + harmonic oscillator.
 * replaces LennardJonesEnergy.__call__ (pita/src/energies/lennardjones_energy.py:213-227),
 * LennardJonesPotential._energy/_log_prob (:121-155) and bgflow's distance_vectors /
 * distances_from_vectors (r = sqrt(|dx|^2 + dist_eps)); ordered pairs (each pair twice). */
int pita_lj_logp_force(const float* x, float* logp, float* force /*nullable*/, int64_t B,
                       int n_particles, int n_dim, float temperature, float energy_factor,
                       float dist_eps, float eps, float rm, float osc_scale, void* stream);

/* The same target with the reference's smooth core (LennardJonesEnergy(smooth=True),
 * pita/src/energies/lennardjones_energy.py:39-54,114-119,131-133): for r < range_min the pair energy is the first
 * interval of the cubic spline fitted to the LJ curve on [range_min, range_max] -- coef4 = its four coefficients
 * (host memory; scipy CubicSpline(...).c[:, 0] as the reference computes them), evaluated as
 * c0 u^3 + c1 u^2 + c2 u + c3 with u = r - range_min.  The force is the analytic derivative of that expression. */
int pita_lj_smooth_logp_force(const float* x, float* logp, float* force /*nullable*/, int64_t B,
                              int n_particles, int n_dim, float temperature, float energy_factor,
                              float dist_eps, float eps, float rm, float osc_scale, float range_min,
                              const float* coef4, void* stream);

/* DW4-style multi double well  E = sum_{i<j} a (d-d0)^4 + b (d-d0)^2 + c.
 * Not present in the reference tree (only a dead import, base_datamodule.py:13); formula of
 * bgflow.MultiDoubleWellPotential. */
int pita_dw_logp_force(const float* x, float* logp, float* force /*nullable*/, int64_t B,
                       int n_particles, int n_dim, float temperature, float a, float b, float c,
                       float d0, void* stream);

/* Fused descent on the target: nsteps of  x <- remove_mean(x + F(x) dt + (noise_scale xi) sqrt_dt)  in one launch,
 * walkers resident in LDS throughout.  replaces WeightedSDEIntegrator.negative_time_descent
 * (pita/src/models/components/sde_integration.py:353-360): dt = dt_negative_time, noise_scale = 1 and
 * sqrt_dt = sqrt(2 dt) when do_langevin, else noise_scale = 0.  Bit-identical to pita_*_logp_force followed by
 * pita_em_step, step after step.  noise: nullable device [nsteps, B, n*d]; NULL -> Philox keyed
 * (seed, walker_offset + walker, step0 + s, particle). */
int pita_lj_descent(float* x, const float* noise, int64_t B, int n_particles, int n_dim, float temperature,
                    float energy_factor, float dist_eps, float eps, float rm, float osc_scale, int nsteps, float dt,
                    float noise_scale, float sqrt_dt, uint64_t seed, uint64_t walker_offset, int64_t step0,
                    int remove_mean, void* stream);
int pita_dw_descent(float* x, const float* noise, int64_t B, int n_particles, int n_dim, float temperature, float a,
                    float b, float c, float d0, int nsteps, float dt, float noise_scale, float sqrt_dt, uint64_t seed,
                    uint64_t walker_offset, int64_t step0, int remove_mean, void* stream);

/* Fused MALA chain on a pair target: all nsteps of metropolis_hastings_mala / _adaptive
 * (pita/src/models/components/sde_integration.py:28-45,362-470) in one launch, walkers resident on chip; bit-identical to
 * pita_*_logp_force + pita_mala_propose + pita_*_logp_force + pita_mala_accept + pita_mala_adapt step after step.
 * pita_lj_mala: LJ13 and LJ55; pita_dw_mala: DW4.  x [B, n*d] and logp [B] (log-density of x on entry) are updated in
 * place; dt_dev holds the step size (adapted in place when adaptive != 0, against the acceptance rate over `total` walkers
 * -- this rank's B: with several ranks the global rate needs the launch-per-kernel path); rates_out [nsteps] receives the
 * acceptance rates.  noise [nsteps, B, n*d] / uniforms [nsteps, B] nullable -> Philox keyed (seed, walker key, step0 + s,
 * particle / 0xFFFFF), walker key = walker_ids[w] or walker_offset + w.  workspace: 8-byte aligned device scratch of
 * pita_lj_mala_workspace_bytes(nsteps) bytes.
 * The adaptive chain synchronises the grid once per step (the global acceptance count decides the next step size); the
 * grid is sized to the co-resident capacity of an idle device; batches beyond it make one HBM round trip of the walkers
 * per step (steps outside, tiles inside) instead of staying on chip.  If the device was NOT idle and a
 * block's bounded wait runs out, the chain is invalid: dt_dev[0] and every rates_out[s] are set to NaN (x / logp hold
 * garbage) and the caller must rerun from its own copy of the walkers -- pita_amd.WeightedSDEIntegrator does. */
size_t pita_lj_mala_workspace_bytes(int nsteps);
int pita_lj_mala(float* x, float* logp, const float* noise /*nullable*/, const float* uniforms /*nullable*/, int64_t B,
                 int n_particles, int n_dim, float temperature, float energy_factor, float dist_eps, float eps, float rm,
                 float osc_scale, int nsteps, double* dt_dev, int adaptive, int64_t total, uint64_t seed,
                 uint64_t walker_offset, const int64_t* walker_ids /*nullable*/, int64_t step0, int remove_mean,
                 float* rates_out /*nullable*/, void* workspace, void* stream);
int pita_dw_mala(float* x, float* logp, const float* noise /*nullable*/, const float* uniforms /*nullable*/, int64_t B,
                 int n_particles, int n_dim, float temperature, float a, float b, float c, float d0, int nsteps,
                 double* dt_dev, int adaptive, int64_t total, uint64_t seed, uint64_t walker_offset,
                 const int64_t* walker_ids /*nullable*/, int64_t step0, int remove_mean, float* rates_out /*nullable*/,
                 void* workspace, void* stream);

/* ---------------------------------------------------------------- wide EGNN backbone (hidden_nf <= 64, static node features)
 * replaces EGNN_dynamics_AD2_cat.forward (pita/src/models/components/egnn_dynamics_ad2_cat.py:157-203; the
 * alanine-dipeptide backbone of configs/model/net/egnn_dynamics_ad2_cat.yaml: hidden 64 x 5 layers, one-hot atom-type
 * node features concatenated with t and beta) over EGNN / E_GCL of egnn.py:108-346, and ScoreNet's EDM preconditioning
 * (score_net.py:13-43).  Two kernels serve the handle: a matrix-pipe kernel (csrc/egnn_wide_mfma_kernel.hip: 64 x 64 dense
 * layers as 2 x 2 blocks of 32 x 32 x 16 f16 MFMAs on the fp32-equivalent two-piece split) for the particle systems it is
 * instantiated for (22, 33, 42, 13, 55 particles x 3), and a vector-pipe kernel (csrc/egnn_wide_kernel.hip: lane = hidden feature, fp32 FMA
 * chains) for every other shape -- which also recomputes, behind the matrix-pipe launch, exactly those walkers whose
 * result came out non-finite (an activation beyond the f16 range).
 * weights: the module's state_dict flattened in registration order (embedding, embedding_out, gcl_0 .. gcl_{L-1}, like
 * pita_egnn_create); h_initial: host [n_particles, n_static] static node features (the reference's get_h_initial()). */
typedef struct pita_egnn_wide pita_egnn_wide_t;
typedef struct {
  int n_particles, n_dim, hidden_nf, n_layers;
  int n_static;       /* static features per node; in_node_nf = n_static + 1 (t) + condition_beta */
  int condition_beta;
  int attention, tanh;
  float coords_range; /* 15.0 in the reference; per-layer range = coords_range / n_layers */
} pita_egnn_wide_config;
int64_t pita_egnn_wide_num_weights(const pita_egnn_wide_config* cfg);
int pita_egnn_wide_create(pita_egnn_wide_t** out, const pita_egnn_wide_config* cfg, const float* weights,
                          int64_t n_weights, const float* h_initial /* host; nullable when n_static == 0 */);
int pita_egnn_wide_destroy(pita_egnn_wide_t* net);
/* 1 when pita_egnn_wide_eval runs this handle on the matrix-pipe kernel, 0 when on the vector-pipe kernel alone */
int pita_egnn_wide_uses_matrix_pipe(const pita_egnn_wide_t* net);
/* what = 0: vel[B, n*d] = backbone(t[B], x[B, n*d], beta[B]) (mean-free); 1: denoiser D_theta(h = t, x); 2: score */
int pita_egnn_wide_eval(pita_egnn_wide_t* net, int what, const float* t, const float* x, const float* beta /*nullable*/,
                        float* out, int64_t B, void* stream);

/* Fused sampler on this backbone: n_steps Euler-Maruyama steps of the NOT-debiased reverse VE-SDE in ONE launch, walkers
 * on chip between the steps -- pita_egnn_sampler_run for EGNN_dynamics_AD2_cat (same step table PITA_ST_*, same Philox
 * keying by (seed, walker_offset + walker, step0 + step, particle), same stats_out moments; replaces, per step,
 * sdes.py:117-128,245-251 + sde_integration.py:299-351,148).  Matrix-pipe kernel where the particle system has one,
 * then the fp32 vector-pipe kernel on exactly the walkers it left non-finite (an activation beyond the f16 range),
 * restarted from a handle-owned backup. */
int pita_egnn_wide_sampler_run(pita_egnn_wide_t* net, float* x, int64_t B, const float* step_tab, int n_steps,
                               const float* noise /*nullable*/, uint64_t seed, uint64_t walker_offset, int64_t step0,
                               int remove_mean, double* stats_out /*nullable*/, void* stream);

/* Forward-mode derivative of the EDM denoiser D(h, x) = c_s x + c_out F(c_noise(h), c_in(h) x, beta) around this
 * backbone, one tangent direction per launch (arguments as pita_egnn_jvp):  dout = J_x D . vx + dD/dh . vh, out = D
 * (nullable), dot_out[b * dot_stride + dot_off] = <x_b, dD_b>, diag_acc[b] += dD[b, dir].  What the debiased
 * Feynman-Kac regime (pita/src/models/components/sdes.py:151-239 with utils.py:30-51, energy_net.py:51-62) needs of
 * EGNN_dynamics_AD2_cat: trace J_x D, J_x D^T x and <x, dD/dh> are sums over such directions; the reference takes them
 * from vmap(jacrev) and autograd.  fp32 vector-pipe kernel (csrc/egnn_wide_kernel.hip: egnn_wide_jvp_kernel). */
int pita_egnn_wide_jvp(pita_egnn_wide_t* net, const float* h, const float* x, const float* beta /*nullable*/,
                       const float* vx /*nullable*/, int dir, const float* vh /*nullable*/, float* out /*nullable*/,
                       float* dout /*nullable*/, float* dot_out /*nullable*/, int64_t dot_stride, int64_t dot_off,
                       float* diag_acc /*nullable*/, int64_t B, void* stream);

/* Reverse-mode derivative of the same denoiser (arguments as pita_egnn_vjp): vjp = J_x D(h, x)^T cot for a per-walker
 * cotangent (null: x itself -- grad_x E_theta = ((1 + c_s) x - D - J_x D^T x)/h, which the reference takes from autograd,
 * pita/src/models/components/energy_net.py:51-62), out = D (nullable), dot_h (nullable) = <cot, dD/dh> (the h-derivative
 * term of dE_theta/dt, sdes.py:218) -- ONE launch instead of dim + 1 forward-mode launches.  fp32 vector-pipe kernel
 * with per-layer checkpoints in a handle-owned scratch (csrc/egnn_wide_kernel.hip: egnn_wide_vjp_kernel). */
int pita_egnn_wide_vjp(pita_egnn_wide_t* net, const float* h, const float* x, const float* beta /*nullable*/,
                       const float* cot /*nullable*/, float* out /*nullable*/, float* vjp, float* dot_h /*nullable*/,
                       int64_t B, void* stream);

/* Diagonal Gaussian mixture with equal weights.
 * replaces GMM.__call__ (pita/src/energies/gmm_energy.py:87-90) ->
 * fab GMM.log_prob (fab/fab/target_distributions/gmm.py:71-79,104).
 * means, scales: device [K, dim].  */
int pita_gmm_logp_force(const float* x, float* logp, float* force /*nullable*/, int64_t B, int dim,
                        const float* means, const float* scales, int K, float temperature,
                        void* stream);

/* Table-driven classical force field (K4): bonds, angles, periodic torsions, LJ + Coulomb over all atom pairs with
 * exceptions and the optional CutoffNonPeriodic reaction field, and the GB-OBC1 implicit solvent with its ACE
 * surface-area term -- the functional forms of OpenMM's HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce,
 * NonbondedForce and GBSAOBCForce.  Stands in for the arithmetic ALPEnergy.__call__
 * (pita/src/energies/alp_energy.py:93-149) delegates to OpenMM (amber14-all.xml + implicit/obc1.xml); the real
 * amber14 parameters are outside the reference tree (parity unpinned).  All table pointers are HOST pointers copied
 * at creation.  Units: nm, kJ/mol, elementary charges; x_model * length_scale = nm; logp = -E/kT. */
typedef struct pita_ff pita_ff_t;
typedef struct {
  int n_atoms;
  int n_bonds;      const int* bond_idx;  /* [n_bonds][2] */      const float* bond_par;  /* [n_bonds][2]: r0, k */
  int n_angles;     const int* angle_idx; /* [n_angles][3] */     const float* angle_par; /* [n_angles][2]: theta0, k */
  int n_torsions;   const int* tors_idx;  /* [n_torsions][4] */   const float* tors_par;  /* [n_torsions][3]: periodicity, phase, k */
  const float* charge; const float* sigma; const float* epsilon;  /* per atom; Lorentz-Berthelot mixing */
  int n_exceptions; const int* exc_idx;   /* [n_exceptions][2] */ const float* exc_par;   /* [n][3]: chargeProd, sigma, epsilon */
  int use_cutoff; float cutoff; float rf_dielectric;              /* CutoffNonPeriodic reaction field (78.3 in OpenMM) */
  float length_scale;                                              /* 0.1640 for the reference's normalised ALDP (energy/aldp.yaml:11) */
  float kT;                                                        /* kJ/mol */
  /* GBSAOBCForce, OBC1 (amber implicit/obc1.xml): NULL gb_radius = no implicit solvent */
  const float* gb_radius;                                          /* [n] nm */
  const float* gb_scale;                                           /* [n] HCT overlap scale factors */
  float gb_solute_dielectric, gb_solvent_dielectric;               /* OpenMM defaults 1.0, 78.5 */
  float gb_surface_area_factor;                                    /* 4 pi x 2.25936 kJ/mol/nm^2 = 28.3919551; 0 = no SA term */
} pita_ff_config;
int pita_ff_create(pita_ff_t** out, const pita_ff_config* cfg);
int pita_ff_destroy(pita_ff_t* ff);
int pita_ff_logp_force(pita_ff_t* ff, const float* x, float* logp, float* force /*nullable*/, int64_t B, void* stream);

/* Negative-time / Langevin descent on the force-field target (sde_integration.py:353-360 with the target of
 * alp_energy.py:122-149): n_steps of x <- remove_mean(x + F dt + noise_scale sqrt_dt xi) in ONE launch, walkers LDS-resident,
 * in place on x [B, 3 n_atoms]; bit-identical to n_steps x (pita_ff_logp_force + pita_em_step).  noise: [n_steps, B, 3 n]
 * injected normals or NULL (Philox keyed by seed, walker_offset + walker, step0 + step, atom). */
int pita_ff_descent(pita_ff_t* ff, float* x, const float* noise /*nullable*/, int64_t B, int n_steps, float dt,
                    float noise_scale, float sqrt_dt, uint64_t seed, uint64_t walker_offset, int64_t step0,
                    int remove_mean, void* stream);

/* ---------------------------------------------------------------- EGNN backbone (K5, K7)
 * replaces EGNN_dynamics.forward (pita/src/models/components/egnn_temp_conditioned.py:56-93,
 * egnn.py:50-80), EGNN.forward (:172-194), E_GCL (:197-356) and the EDM wrappers
 * ScoreNet.denoiser / ScoreNet.forward (score_net.py:13-43).
 */
typedef struct pita_egnn pita_egnn_t;

typedef struct {
  int n_particles;
  int n_dim;
  int hidden_nf;        /* only 32 is implemented in HIP */
  int n_layers;
  int in_node_nf;       /* 1 = time only (egnn.py), 2 = time + temperature */
  int attention;        /* att_mlp sigmoid gate present */
  int tanh;             /* tanh on the coordinate head (* coords_range / n_layers) */
  float coords_range;   /* 15.0 in the reference */
  int feature_layout;   /* 0 = "pita": the reference's t/beta interleave quirk
                           (egnn_temp_conditioned.py:68-78); 1 = per-node (t, beta) */
  int precision;        /* arithmetic of the dense layers, all fp32-accurate:
                           0 = v_mfma_f32_32x32x2_f32 (bit-exact fp32 fmaf chains),
                           1 = bf16 matrix pipe with an exact 3-way operand split (6 products,
                               error <= 2^-24 |w||x|: fp32-equivalent, not bit-identical to 0),
                           2 = f16 matrix pipe with a 2-way round-to-nearest operand split (3 products,
                               each operand within 2^-24 relative; operands travel scaled by 16, so
                               |activations| must stay below 4094; forward / fused sampler only -- the
                               derivative kernels of the same handle run mode 1) */
} pita_egnn_config;

/* `weights` is a HOST pointer to the reference state_dict flattened in its own key order:
 * embedding.{weight,bias}, embedding_out.{weight,bias}, then per layer l:
 * edge_mlp.0.{weight[H,2H+2],bias}, edge_mlp.2.{weight,bias}, node_mlp.0.{weight[H,2H],bias},
 * node_mlp.2.{weight,bias}, coord_mlp.0.{weight,bias}, coord_mlp.2.weight[1,H],
 * (att_mlp.0.{weight[1,H],bias[1]} if attention).  n_weights is checked. */
int pita_egnn_create(pita_egnn_t** out, const pita_egnn_config* cfg, const float* weights,
                     int64_t n_weights);
int pita_egnn_destroy(pita_egnn_t* net);
int64_t pita_egnn_num_weights(const pita_egnn_config* cfg);

/* backbone forward: out[B,D] = F(t[B], x[B,D], beta[B]) (beta nullable when in_node_nf==1) */
int pita_egnn_forward(pita_egnn_t* net, const float* t, const float* x, const float* beta,
                      float* out, int64_t B, void* stream);
/* EDM-preconditioned outputs, h = sigma^2 per walker (score_net.py:21-43):
 *   what = 1: denoiser D = c_s x + c_out F(c_noise, c_in x, beta)
 *   what = 2: score (D - x)/h */
int pita_egnn_edm(pita_egnn_t* net, int what, const float* h, const float* x, const float* beta,
                  float* out, int64_t B, void* stream);

/* Forward-mode derivative of the denoiser D(h, x) = c_s x + c_out F(c_noise(h), c_in(h) x, beta), one tangent
 * direction per launch:  dout = J_x D . vx + dD/dh . vh   (and out = D when out != NULL).
 * vx: device [B, D] or NULL; when NULL the direction is the unit vector e_dir of every walker (0 <= dir < D) or
 * zero (dir = -1).  vh: device [B] or NULL (= 0).
 * Building block of the debiased Feynman-Kac regime (sdes.py:151-239): div_x s_theta (utils.py:30-51),
 * grad_x E_theta (energy_net.py:51-62) and dE_theta/dt (sdes.py:218) are linear in these JVPs -- the reference
 * gets them from torch.func.jacrev / autograd. */
int pita_egnn_jvp(pita_egnn_t* net, const float* h, const float* x, const float* beta, const float* vx,
                  int dir, const float* vh, float* out /*nullable*/, float* dout /*nullable*/,
                  float* dot_out /*nullable: dot_out[b*dot_stride + dot_off] = <x_b, dD_b>*/, int64_t dot_stride,
                  int64_t dot_off, float* diag_acc /*nullable: diag_acc[b] += dD[b, dir]*/, int64_t B, void* stream);

/* Reverse-mode derivative of the denoiser:  vjp = J_x D(h, x)^T cot  (and out = D when out != NULL), one launch for
 * all walkers.  cot: device [B, D] or NULL (= x).  With cot = x this is the only derivative grad_x E_theta needs
 * (energy_net.py:33-62: E = (1+c_s)|x|^2/(2h) - <D, x>/h, so grad E = ((1+c_s) x - D - J^T x)/h), replacing the
 * reference's torch.autograd.grad through the network.  Checkpoint scratch is owned by the handle.
 * dot_h (nullable, device [B]): <cot, dD/dh> from the same reverse sweep (it also reaches the h-dependent inputs of the
 * backbone -- time feature ln(h)/8, the scaling c_in(h) -- and the explicit c_s(h), c_out(h)): with cot = x this is the
 * term of dE_theta/dt (sdes.py:218) that otherwise takes a forward-mode launch in the h direction.
 * dot_parts (nullable, device [B, 2], needs dot_h): the same derivative split the way the reference's energy is written
 * (energy_net.py:33-41, E = |x|^2 (1 - c_s)/(2h) - c_out/(c_in h) <F, c_in x>):
 *   dot_parts[b][0] = c_out <cot, F>,   dot_parts[b][1] = <cot, d(c_out F)/dh> = dot_h[b] - c_s'(h) <cot, x>.
 * pita_fk_assemble builds E_theta and dE_theta/dh from these without the |x|^2/h^2-sized cancellation that the forms
 * through <D, x> and <x, dD/dh> carry at small h. */
int pita_egnn_vjp(pita_egnn_t* net, const float* h, const float* x, const float* beta, const float* cot /*nullable*/,
                  float* out /*nullable*/, float* vjp, float* dot_h /*nullable*/, float* dot_parts /*nullable*/, int64_t B,
                  void* stream);

/* Exact trace of the denoiser Jacobian, K unit directions per launch sharing one primal evaluation:
 *   diag_acc[b] += sum_{k < ndir} (J_x D(h, x) e_{dir0+k})_{dir0+k},   1 <= ndir <= pita_egnn_div_directions(net).
 * Looping dir0 over 0, K, 2K, ... < D accumulates trace(J_x D), from which div_x s_theta = (trace - D)/h: the exact
 * divergence the reference computes with vmap(jacrev) (pita/src/models/components/utils.py:30-51). */
int pita_egnn_div_directions(const pita_egnn_t* net);
int pita_egnn_div_accumulate(pita_egnn_t* net, const float* h, const float* x, const float* beta, int dir0, int ndir,
                             float* diag_acc, float* denoiser_out /*nullable: D(h, x) of the shared primal*/, int64_t B,
                             void* stream);

/* The whole trace in one call: trace[b] = trace(J_x D(h_b, x_b)) (overwritten), optionally D itself.  Handles of
 * precision 2 compute the primal network ONCE: the first launch stores the per-edge primal factors the tangent chains
 * consume in a handle-owned cache (about 180 KB per LJ13 walker; batches beyond PITA_DIV_CACHE_GB, default 24, go in chunks
 * of walkers) and the remaining directions run as tangent-only launches, four directions each, streaming that cache. */
int pita_egnn_jacobian_trace(pita_egnn_t* net, const float* h, const float* x, const float* beta, float* trace,
                             float* denoiser_out /*nullable*/, int64_t B, void* stream);

/* Work accounting for the roofline of the debiased leg (bench.py): matrix-core wave-instructions per walker for one full
 * trace (all n_particles * n_dim directions) on the path this handle takes; units as pita_egnn_sampler_work. */
int pita_egnn_div_work(const pita_egnn_t* net, double* mfma16_per_walker, double* mfma32_per_walker);

/* Feynman-Kac drift assembly per walker from those reductions (replaces the torch/autograd expressions of
 * sdes.py:157-227): with E = be [(1+c_s)|x|^2/(2h) - <D_E,x>/h],
 *   grad E = be ((1+c_s) x - D_E - jtx_E)/h,  s = bs (D_S - x)/h,  b = s g2/2,
 *   U = E, or with pin_energy (energy_net.py:43-48)  U = w U0 + (1-w) E,  U0 = clamp(-logp_target, +-1e3),
 *   grad U = (1-w) grad E  (the reference's energy classes return log p detached: no target force enters),
 *   dU/dt = dw/dt (U0 - E) + (1-w) dE/dh dh/dt   (dh/dt from the schedule, not assumed equal to g^2),
 *   drift_X = gamma (-grad U) g2/2 + gamma b,
 *   drift_A = gamma^2 <-grad U, b> + gamma bs (trace_S - D)/h g2/2 + gamma dU/dt + dgamma U   (NOT yet clamped).
 * be / bs: per-walker inverse temperatures when the energy / score net was built with precondition_beta
 * (score_net.py:36-38, energy_net.py:40-41), NULL = 1.  pin_w = (1-t)^3 and pin_dw = -3 (1-t)^2 with logp_target [B];
 * logp_target NULL = no pinning.  All arrays are device pointers: x, D_E, jtx_E, D_S, drift_X [B,D]; the rest [B].
 * dot_parts ([B, 2] from pita_egnn_vjp, nullable): when given, E = be [|x|^2/(2(1+h)) - s1/h] and
 * dE/dh = be [-|x|^2/(2(1+h)^2) + s1/h^2 - s2/h] with (s1, s2) = dot_parts[b] -- the same quantities in the
 * reference's own well-conditioned form; NULL keeps the forms through <D_E, x> and dot_h. */
int pita_fk_assemble(const float* x, const float* h, const float* g2, const float* dhdt, const float* D_E,
                     const float* jtx_E, const float* dot_h, const float* dot_parts /*nullable*/, const float* D_S,
                     const float* trace_S, float gamma, float dgamma, const float* beta_e /*nullable*/, const float* beta_s /*nullable*/, float pin_w,
                     float pin_dw, const float* logp_target /*nullable*/, float* drift_X, float* drift_A, float* div_bt,
                     float* cross, float* dUdt, float* Ut, int64_t B, int D, void* stream);
/* K11: in place a[c] = min(a[c], quantile_q(a over its chunk)), chunks of `chunk` consecutive walkers, linear
 * interpolation like torch.quantile (sdes.py:230; sde_integration.py:179). */
int pita_quantile_clamp(float* a, int64_t B, int64_t chunk, float q, void* stream);

/* ---------------------------------------------------------------- fused sampler (K5+K7+K8)
 * Runs n_steps Euler-Maruyama steps of the NOT-debiased reverse VE-SDE in ONE launch, walkers
 * resident on chip for the whole trajectory.  replaces, per step,
 * VEReverseSDE.f_not_debiased + .diffusion (sdes.py:117-128,245-251),
 * WeightedSDEIntegrator.euler_maruyama_step (sde_integration.py:299-351) and the
 * remove_mean of integrate_sde (:148).
 *
 * step_tab: device [n_steps][PITA_STEP_STRIDE] floats computed by the host in the
 * reference's fp32 op order (see pita_amd/sde_integration.py):
 */
#define PITA_STEP_STRIDE 16
enum {
  PITA_ST_CS = 0, PITA_ST_CIN = 1, PITA_ST_COUT = 2, PITA_ST_CNOISE = 3, PITA_ST_H = 4,
  PITA_ST_G2 = 5, PITA_ST_GAMMA = 6, PITA_ST_DT = 7, PITA_ST_NOISE_SCALE = 8 /* scale*g(t) */,
  PITA_ST_SQRT_DT = 9, PITA_ST_BETA = 10
};
/* noise: nullable device [n_steps, B, D] standard normals (parity mode).  When NULL the kernel
 * draws Philox4x32-10 normals keyed by (seed, walker_offset + walker, step0 + step, particle)
 * so results do not depend on how walkers are sharded over GPUs.
 * drift_out: nullable device [B, D]; receives drift_X of the LAST step of the launch.
 * stats_out: nullable device double [n_steps][4]; step s ADDS (sum drift_X, sum drift_X^2, sum diffusion,
 * sum diffusion^2) over this call's B*D elements, diffusion = noise_scale * xi (sdes.py:250).  These are the moments
 * behind the per-step SDETerms the reference returns and only ever reduces to .mean()/.std()
 * (sde_integration.py:150,289; energytemp_module.py:1132-1143). */
int pita_egnn_sampler_run(pita_egnn_t* net, float* x, int64_t B, const float* step_tab,
                          int n_steps, const float* noise, uint64_t seed, uint64_t walker_offset,
                          int64_t step0, int remove_mean, float* drift_out, double* stats_out, void* stream);

/* Work accounting for the roofline of pita_egnn_sampler_run (bench.py): matrix-core wave-instructions executed per
 * walker-step at batch B, counted on the kernel's own loop structure -- mfma16: v_mfma_f32_32x32x16_{bf16,f16}
 * (32 768 flop each), mfma32: v_mfma_f32_32x32x2_f32 (4 096 flop each). */
int pita_egnn_sampler_work(const pita_egnn_t* net, int64_t B, double* mfma16_per_walker_step,
                           double* mfma32_per_walker_step);

/* How pita_egnn_sampler_run maps a batch of B walkers onto the device (bench.py `small_batch`: the reference generates
 * num_eval_samples = 2 048 walkers in inference chunks of 512, configs/experiment/lj13.yaml:27,32 -- far fewer than the
 * chip holds): walkers_per_group = walkers packed into one wavefront's column tiles (the small-group mapping is chosen
 * when the regular one would leave SIMDs with a single wave), waves = wavefronts the launch starts, wave_slots = wavefronts
 * of this kernel the device can hold at once (CUs x resident blocks x waves per block). */
int pita_egnn_sampler_mapping(const pita_egnn_t* net, int64_t B, int* walkers_per_group, int64_t* waves,
                              int64_t* wave_slots);

/* ---------------------------------------------------------------- MLP backbone (K6)
 * replaces MyMLP.forward (mlp.py:244-267) / MyMLPTemperature.forward (:501-524) incl. the
 * sinusoidal embeddings (:11-24) and residual GELU blocks (:100-118). */
typedef struct pita_mlp pita_mlp_t;
typedef struct {
  int input_dim;       /* number of coordinates (each embedded with scale 25) */
  int out_dim;
  int hidden_size;     /* multiple of 32; == emb_size in every reference config */
  int hidden_layers;
  int emb_size;        /* sinusoidal embedding size (even) */
  int temperature_conditioned; /* MyMLPTemperature: extra beta embedding */
} pita_mlp_config;
/* weights: HOST pointer, reference state_dict order: joint_mlp.0.{weight,bias},
 * joint_mlp.{1..L}.ff.{weight,bias}, joint_mlp.{L+1}.{weight,bias}.
 * freqs: HOST pointer to the emb_size/2 sinusoidal frequencies exp(-ln(1e4)/(half-1) * k) as the
 * caller's framework computes them in fp32 (mlp.py:20-21) -- passed in rather than recomputed so
 * that the angles are bit-identical to the reference's. */
int pita_mlp_create(pita_mlp_t** out, const pita_mlp_config* cfg, const float* weights,
                    int64_t n_weights, const float* freqs);
int pita_mlp_destroy(pita_mlp_t* net);
int64_t pita_mlp_num_weights(const pita_mlp_config* cfg);
int pita_mlp_forward(pita_mlp_t* net, const float* t, const float* x, const float* beta /*nullable*/,
                     float* out, int64_t B, void* stream);
/* Fused not-debiased sampler for the MLP backbone: the counterpart of pita_egnn_sampler_run (same step table, noise and
 * Philox conventions; sde_integration.py:299-351 + sdes.py:117-128 + score_net.py:13-43) with the walkers LDS-resident
 * for all n_steps.  Requires out_dim == input_dim == n_particles * n_dim <= 64 (GMM: n_particles = 1, n_dim = 2). */
int pita_mlp_sampler_run(pita_mlp_t* net, float* x, int64_t B, const float* step_tab, int n_steps,
                         const float* noise /*nullable*/, uint64_t seed, uint64_t walker_offset, int64_t step0,
                         int remove_mean, int n_particles, int n_dim, double* stats_out /*nullable, as above*/,
                         void* stream);

/* ---------------------------------------------------------------- elementwise sampler pieces
 * K8: x <- x + drift*dt + (noise_scale*xi)*sqrt_dt, then optional per-walker mean removal
 * (sde_integration.py:347-349,148; data_utils.py:4-26).  noise nullable -> Philox as above.
 * stats_out: nullable device double [4], += the same four moments as pita_egnn_sampler_run for this one step. */
int pita_em_step(float* x, const float* drift, const float* noise, int64_t B, int n_particles,
                 int n_dim, float dt, float noise_scale, float sqrt_dt, uint64_t seed,
                 uint64_t walker_offset, int64_t step, int remove_mean, double* stats_out, void* stream);
/* out[0] += sum v, out[1] += sum v^2 (device double[2]): moments of one SDETerms field (divergence_score, cross_term,
 * dUt_dt; sdes.py:34-41), so the integrator returns statistics instead of N x [B] host copies (sde_integration.py:289) */
int pita_moments(const float* v, int64_t n, double* out, void* stream);
/* The same for up to four vectors of one length in one launch (null pointers are skipped): out[2 q], out[2 q + 1] += the sum /
 * sum of squares of v_q -- the four per-step SDETerms statistics of the debiased regime (drift_A, divergence_score,
 * cross_term, dUt_dt; sde_integration.py:150,289). */
int pita_moments4(const float* v0, const float* v1, const float* v2, const float* v3, int64_t n, double* out /*[8]*/,
                  void* stream);
/* Histogram of n device floats over nbins + 1 ascending device bin edges, as numpy.histogram / matplotlib's `hist` count
 * them (bin i = [edges[i], edges[i+1]), the last bin closed; values outside the edges and NaNs are not counted):
 * counts[nbins] (device, overwritten).  The two 100-bin density arrays under the reference's sample figure
 * (src/energies/base_molecule_energy_function.py:160-254: bins from the test set, energy range (min - 10, max + 10)) are
 * these counts divided by (their sum x the bin widths): pita_amd.metrics.sample_histograms.  1 <= nbins <= 1024. */
int pita_histogram(const float* v, int64_t n, const float* edges, int nbins, unsigned long long* counts, void* stream);
/* K9: MeanFreePrior.sample (base_prior.py:77-83): x = scale * N(0,1) minus particle mean.
 * noise nullable -> Philox keyed (seed, walker_offset + walker, step = -1). */
int pita_prior_sample(float* x, const float* noise, int64_t B, int n_particles, int n_dim,
                      float scale, uint64_t seed, uint64_t walker_offset, int mean_free,
                      void* stream);
/* data_utils.remove_mean in place */
int pita_remove_mean(float* x, int64_t B, int n_particles, int n_dim, void* stream);
/* standard normals [B, D] from the library's Philox stream (the generator used when noise==NULL) */
int pita_fill_normal(float* out, int64_t B, int n_particles, int n_dim, uint64_t seed,
                     uint64_t walker_offset, int64_t step, void* stream);

/* EDM preconditioning around a backbone that is NOT one of the fused kernels (ScoreNet.denoiser / forward,
 * pita/src/models/components/score_net.py:13-43): scale gives the backbone inputs x_in = c_in x and c_noise = ln(h)/8;
 * combine gives D = c_s x + c_out F (times beta + (1-beta) x when beta != NULL, :36-38) and score = (D - x)/h
 * (times beta).  D_out / score_out nullable. */
int pita_edm_scale_input(const float* h, const float* x, float* x_scaled, float* c_noise, int64_t B, int D,
                         void* stream);
int pita_edm_combine(const float* h, const float* x, const float* F, const float* beta /*nullable*/, float* D_out,
                     float* score_out, int64_t B, int D, void* stream);

/* E_theta(h, x) = (1 - c_s)/(2h) |x|^2 - c_out/(c_in h) <F, c_in x> from the backbone output F = F(c_noise, c_in x, beta)
 * (EnergyNet.forward_energy, pita/src/models/components/energy_net.py:14-41; times beta when beta != NULL). */
int pita_energy_theta(const float* h, const float* x, const float* F, const float* beta /*nullable*/, float* E,
                      int64_t B, int D, void* stream);

/* ---------------------------------------------------------------- MALA (K12)
 * sde_integration.py:28-45 mala_proposal; :362-470 accept/reject and step-size adaptation.
 * dt_dev: device double holding the step size (adapted in place by pita_mala_adapt, no host round trip).
 *   propose: x_prop = (x + dt/2 * force) + sqrt(dt) * xi    (noise nullable -> Philox (seed, walker, step, particle))
 *   accept : log q_f = -|x_prop - (x + dt/2 F)|^2 / 2dt, log q_b = -|x - (x_prop + dt/2 F_prop)|^2 / 2dt,
 *            accept iff log u < (logp_prop - logp) + (log q_b - log q_f); x and logp are updated in place with the
 *            reference's float blend a*new + (1-a)*old; optional mean removal (is_molecule); *acc_count += #accepted.
 *            uniforms nullable -> Philox.
 *   adapt  : rate = acc_count / total -> rate_out (nullable); adaptive: dt *= 1.1 if rate > 0.55 else dt /= 1.1;
 *            acc_count is reset.  With several ranks all-reduce acc_count first and pass the global total.
 *   walker_ids (device int64[B], nullable): the Philox walker key of row w; NULL = walker_offset + w.  The chain runs on
 *            the walkers with a finite log-density only (quirk Q7), i.e. on a compacted batch: keying by the original
 *            global index keeps the noise of a walker independent of how many were set aside before it / on other ranks. */
int pita_mala_propose(const float* x, const float* force, float* x_prop, const float* noise, int64_t B,
                      int n_particles, int n_dim, const double* dt_dev, uint64_t seed, uint64_t walker_offset,
                      const int64_t* walker_ids /*nullable*/, int64_t step, void* stream);
int pita_mala_accept(float* x, float* logp, const float* force, const float* x_prop, const float* logp_prop,
                     const float* force_prop, const float* uniforms, int64_t B, int n_particles, int n_dim,
                     const double* dt_dev, uint64_t seed, uint64_t walker_offset, const int64_t* walker_ids /*nullable*/,
                     int64_t step, int remove_mean, int* acc_count, void* stream);
int pita_mala_adapt(double* dt_dev, int* acc_count, int64_t total, int adaptive, float* rate_out, void* stream);

/* ---------------------------------------------------------------- resampling (K10, K11)
 * K10 systematic resampling, replaces sample_cat_sys (utils.py:111-120): weights =
 * clip(softmax(logits),1e-6,1) (not renormalised), inclusive fp32 cumsum, u_k = (u0 + k/B) mod 1
 * in fp64, ids = digitize(u, bins, right=True) clamped to B-1.
 * Five short multi-block passes (block maxima, block sums, clipped-weight block sums, bins, binary search), fixed
 * summation order; exp evaluated in double and rounded once.
 * workspace: 8-byte aligned device scratch of at least pita_resample_workspace_bytes(B) bytes. */
size_t pita_resample_workspace_bytes(int64_t B);
int pita_systematic_resample(const float* logits, int64_t B, double u0, int64_t* ids,
                             void* workspace, void* stream);
/* out[k, :] = src[ids[k], :] */
int pita_gather_rows(const float* src, const int64_t* ids, float* out, int64_t B, int D,
                     void* stream);

/* Number of distinct parents of one resampling event, max(1, #{i : ids[i] != ids[(i - 1) mod B]}), written to the device
 * word `out` (no host synchronisation): the ids of a systematic resampling are non-decreasing up to the cyclic rotation by
 * the event's uniform (utils.py:111-120), so every distinct parent is one cyclic run.  Replaces
 * len(np.unique(choice)) of sde_integration.py:295. */
int pita_count_runs(const int64_t* ids, int64_t B, int64_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PITA_HIP_H */
