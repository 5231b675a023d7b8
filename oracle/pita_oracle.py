"""CPU ORACLE for the PITA inference-time annealed-SDE sampling path.

*** TEST INFRASTRUCTURE ONLY ***  Nothing under ``pita_amd/`` may import this file.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
use it, and only as the checker / the reported CPU baseline.

What it is: a functional, dtype-generic (float32 = reference arithmetic, float64 =
"truth") restatement in plain torch-CPU ops of the reference's algorithms on the hot
path.  Every function cites the reference file:line
it follows (paths relative to /root/reference/).

Parity pinning: the reference ships NO golden vectors / known-answer tests for this
path.  The oracle is therefore pinned against outputs of the reference
itself, imported in the build container with import-time shims
(tests/golden/make_golden.py -> tests/golden/*.npz, checked by tests/test_oracle_golden.py).
Third-party arithmetic absent from /root/reference:
  * bgflow (environment.yaml:56, unpinned git dep): LJ pair distances
    ``sqrt(|dx|^2 + 1e-6)`` over all ordered pairs -- restated from bgflow's published
    ``utils/geometry.py``; cross-checked against the in-tree restatement
    sampling/sample_lj13.py:24-30 (``energy2``).
  * DW4 (bgflow.MultiDoubleWellPotential, only a dead import in the reference,
    pita/src/energies/base_datamodule.py:13): PARITY UNPINNED -- defined here by the
    published formula E = sum_{i<j} a (d-d0)^4 + b (d-d0)^2 + c, a=0.9 b=-4 c=0 d0=4.
  * OpenMM/amber14 for alanine dipeptide: not restatable here (parameters outside the
    tree); out of scope this round.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------
# A8  noise schedule      pita/src/models/components/noise_schedules.py:98-138
# --------------------------------------------------------------------------------------


@dataclass
class Elucidating:
    """EDM/Karras schedule; ``h`` = sigma(t)^2, ``g`` = sqrt(dh/dt).

    Follows noise_schedules.py:99-124 including its op order (term1/term2 are python
    floats; the tensor math happens in the dtype of ``t``).
    """

    sigma_min: float
    sigma_max: float
    rho: float

    @property
    def term1(self) -> float:  # noise_schedules.py:103
        return self.sigma_max ** (1 / self.rho)

    @property
    def term2(self) -> float:  # noise_schedules.py:104
        return self.sigma_min ** (1 / self.rho) - self.sigma_max ** (1 / self.rho)

    def h(self, t: Tensor) -> Tensor:  # noise_schedules.py:114-115
        return (self.term1 + (1 - t) * self.term2) ** (2 * self.rho)

    def g(self, t: Tensor) -> Tensor:  # noise_schedules.py:108-112
        return (-2 * self.rho * (self.term1 + (1 - t) * self.term2) ** (2 * self.rho - 1) * self.term2) ** 0.5

    def t(self, ht: Tensor) -> Tensor:  # noise_schedules.py:117-119
        return 1 - ((ht ** (1 / (2 * self.rho)) - self.term1) / self.term2)

    def dh_dt(self, t: Tensor) -> Tensor:  # noise_schedules.py:121-125
        return -2 * self.rho * self.term2 * (self.term1 + (1 - t) * self.term2) ** (2 * self.rho - 1)


@dataclass
class Geometric:
    """noise_schedules.py:62-95."""

    sigma_min: float
    sigma_max: float

    def g(self, t: Tensor) -> Tensor:
        sd = self.sigma_max / self.sigma_min
        return self.sigma_min * (sd**t) * ((2 * np.log(sd)) ** 0.5)

    def h(self, t: Tensor) -> Tensor:
        sd = self.sigma_max / self.sigma_min
        return (self.sigma_min * (((sd ** (2 * t)) - 1) ** 0.5)) ** 2


# --------------------------------------------------------------------------------------
# A9  annealing-factor schedules   annealing_factor_schedules.py:20-109
# --------------------------------------------------------------------------------------


@dataclass
class GammaConstant:  # :20-32
    annealing_factor: float

    def gamma(self, t: Tensor) -> Tensor:
        return torch.ones_like(t) * self.annealing_factor

    def dgamma_dt(self, t: Tensor) -> Tensor:
        return torch.zeros_like(t)


@dataclass
class GammaLinear:  # :35-69
    annealing_factor: float
    annealing_factor_start: float
    t_start: float = 1.0
    t_end: float = 0.0

    def _slope(self) -> float:
        return (self.annealing_factor - self.annealing_factor_start) / (self.t_end - self.t_start)

    def gamma(self, t: Tensor) -> Tensor:
        lin = self._slope() * (t - self.t_start) + self.annealing_factor_start
        return torch.where(t > self.t_start, self.annealing_factor_start,
                           torch.where(t < self.t_end, self.annealing_factor, lin))

    def dgamma_dt(self, t: Tensor) -> Tensor:
        zero = torch.tensor(0.0, dtype=t.dtype)
        return torch.where(t > self.t_start, zero, torch.where(t < self.t_end, zero, self._slope()))


@dataclass
class GammaSigmoid:  # :72-109
    annealing_factor: float
    annealing_factor_start: float
    t_start: float = 1.0
    t_end: float = 0.0
    sharpness: float = 10.0

    def _smooth(self, t: Tensor) -> Tensor:
        center = (self.t_start + self.t_end) / 2
        width = self.t_start - self.t_end
        return 1 / (1 + torch.exp(-self.sharpness * ((center - t) / width)))

    def gamma(self, t: Tensor) -> Tensor:
        return self.annealing_factor_start + (self.annealing_factor - self.annealing_factor_start) * self._smooth(t)

    def dgamma_dt(self, t: Tensor) -> Tensor:
        s = self._smooth(t)
        width = self.t_start - self.t_end
        return (self.annealing_factor - self.annealing_factor_start) * ((self.sharpness / width) * s * (1 - s))


# --------------------------------------------------------------------------------------
# A11 remove_mean, A10 prior     pita/src/utils/data_utils.py:4-26, base_prior.py:77-83
# --------------------------------------------------------------------------------------


def remove_mean(x: Tensor, n_particles: int, n_dim: int) -> Tensor:
    shp = x.shape
    v = x.reshape(-1, n_particles, n_dim)
    return (v - v.mean(dim=1, keepdim=True)).reshape(shp)


def prior_from_noise(noise: Tensor, scale: float, n_particles: int, n_dim: int, mean_free: bool = True) -> Tensor:
    """MeanFreePrior.sample with the ``randn`` draw supplied by the caller (base_prior.py:77-83)."""
    s = noise * scale
    if mean_free:
        s = remove_mean(s, n_particles, n_dim)
    return s


def prior_scale(schedule, gamma_sched, t_start: float, dtype=torch.float32) -> float:
    """sqrt(h(t_start)/gamma(t_start))   energytemp_module.py:250-257."""
    t = torch.tensor(t_start, dtype=dtype)
    return float((schedule.h(t) / gamma_sched.gamma(t)) ** 0.5)


# --------------------------------------------------------------------------------------
# A12 Lennard-Jones target     lennardjones_energy.py:34-36,121-155,213-227 (+ bgflow distances)
# --------------------------------------------------------------------------------------


def lj_logp(x: Tensor, n_particles: int, n_dim: int, temperature: float = 1.0, energy_factor: float = 1.0,
            dist_eps: float = 1e-6, eps: float = 1.0, rm: float = 1.0, osc_scale: float = 1.0) -> Tensor:
    """log-density = -E/T with E = energy_factor * sum_{i != j} lj(r_ij) + 0.5*osc*|x - mean|^2.

    Same op order as the reference: ordered pairs with the diagonal removed
    (bgflow distance_vectors), r = sqrt(sum(d^2)+dist_eps), ``(rm/r)**12 - 2 (rm/r)**6``
    (lennardjones_energy.py:34-36), sum, oscillator (:139-141), /T (:153-155).
    """
    B = x.shape[0]
    v = x.reshape(B, n_particles, n_dim)
    diff = v[:, :, None, :] - v[:, None, :, :]
    mask = ~torch.eye(n_particles, dtype=torch.bool)
    diff = diff[:, mask].reshape(B, n_particles, n_particles - 1, n_dim)
    r = (diff.pow(2).sum(dim=-1) + dist_eps).sqrt()
    lj = eps * ((rm / r) ** 12 - 2 * (rm / r) ** 6)
    e = lj.reshape(B, -1).sum(dim=-1) * energy_factor
    c = v - v.mean(dim=1, keepdim=True)
    e = e + 0.5 * c.pow(2).sum(dim=(-2, -1)) * osc_scale
    return -e / temperature


def lj_logp_force(x: Tensor, n_particles: int, n_dim: int, temperature: float = 1.0, energy_factor: float = 1.0,
                  dist_eps: float = 1e-6, eps: float = 1.0, rm: float = 1.0, osc_scale: float = 1.0
                  ) -> Tuple[Tensor, Tensor]:
    """(logp, d logp / dx).  The reference gets the force by autograd
    (lennardjones_energy.py:222-225); here it is the closed form
      F_i = -(1/T) [ 2*ef * sum_{j!=i} e'(r_ij) (x_i-x_j)/r_ij + osc*(x_i - mean) ],
      e'(r) = eps*(-12 rm^12 r^-13 + 12 rm^6 r^-7).
    """
    B = x.shape[0]
    v = x.reshape(B, n_particles, n_dim)
    diff = v[:, :, None, :] - v[:, None, :, :]  # [B,i,j,d] = x_i - x_j
    r2 = diff.pow(2).sum(dim=-1) + dist_eps
    eye = torch.eye(n_particles, dtype=torch.bool)
    r2 = r2.masked_fill(eye, 1.0)
    inv_r2 = rm * rm / r2
    s6 = inv_r2**3
    # e'(r)/r = eps * (-12 s6^2 + 12 s6) / r^2
    coef = eps * (-12.0 * s6 * s6 + 12.0 * s6) / r2
    coef = coef.masked_fill(eye, 0.0)
    grad_e = 2.0 * energy_factor * (coef[..., None] * diff).sum(dim=2)
    c = v - v.mean(dim=1, keepdim=True)
    grad_e = grad_e + osc_scale * c
    force = (-grad_e / temperature).reshape(B, n_particles * n_dim)
    return lj_logp(x, n_particles, n_dim, temperature, energy_factor, dist_eps, eps, rm, osc_scale), force


def lj_smooth_coeffs(range_min: float = 0.65, range_max: float = 2.0, interpolation: int = 1000, eps: float = 1.0,
                     rm: float = 1.0) -> Tuple[Tensor, Tensor]:
    """(breakpoints x [interpolation], coefficients c [4, interpolation-1]) of the spline the reference fits in
    LennardJonesPotential.__init__ (lennardjones_energy.py:114-119): scipy CubicSpline through the float32 LJ curve."""
    from scipy.interpolate import CubicSpline
    pts = torch.linspace(range_min, range_max, interpolation)
    es = eps * ((rm / pts) ** 12 - 2 * (rm / pts) ** 6)
    c = CubicSpline(pts.numpy(), es.numpy()).c
    return pts, torch.tensor(c).float()


def lj_smooth_logp(x: Tensor, n_particles: int, n_dim: int, temperature: float = 1.0, energy_factor: float = 1.0,
                   dist_eps: float = 1e-6, eps: float = 1.0, rm: float = 1.0, osc_scale: float = 1.0,
                   range_min: float = 0.65, range_max: float = 2.0, interpolation: int = 1000) -> Tensor:
    """lj_logp with smooth=True: pair energies below range_min come from the spline (cubic_spline,
    lennardjones_energy.py:39-54: bucketize - 1 clamped to [0, len-2], so every r < range_min lands in interval 0), blended
    as ``lj * ~filter + filter * spline(r)`` (:131-133)."""
    B = x.shape[0]
    v = x.reshape(B, n_particles, n_dim)
    diff = v[:, :, None, :] - v[:, None, :, :]
    mask = ~torch.eye(n_particles, dtype=torch.bool)
    diff = diff[:, mask].reshape(B, n_particles, n_particles - 1, n_dim)
    r = (diff.pow(2).sum(dim=-1) + dist_eps).sqrt()
    lj = eps * ((rm / r) ** 12 - 2 * (rm / r) ** 6)
    xs, c = lj_smooth_coeffs(range_min, range_max, interpolation, eps, rm)
    iv = torch.clamp(torch.bucketize(r, xs) - 1, 0, len(xs) - 2)
    dx = r - xs[iv]
    spl = c[0, iv] * dx**3 + c[1, iv] * dx**2 + c[2, iv] * dx + c[3, iv]
    filt = r < range_min
    lj = lj * ~filt + filt * spl
    e = lj.reshape(B, -1).sum(dim=-1) * energy_factor
    cc = v - v.mean(dim=1, keepdim=True)
    e = e + 0.5 * cc.pow(2).sum(dim=(-2, -1)) * osc_scale
    return -e / temperature


def lj_smooth_logp_force(x: Tensor, n_particles: int, n_dim: int, **kw) -> Tuple[Tensor, Tensor]:
    """(logp, d logp / dx) by autograd, as the reference does (lennardjones_energy.py:222-225)."""
    xg = x.detach().clone().requires_grad_(True)
    lp = lj_smooth_logp(xg, n_particles, n_dim, **kw)
    (g,) = torch.autograd.grad(lp.sum(), xg)
    return lp.detach(), g


def lj_energy2(x: Tensor, n_particles: int) -> Tensor:
    """Second, in-tree restatement of the LJ log-density: sampling/sample_lj13.py:24-30
    (2*sum_{i<j} via pdist, NO distance eps)."""
    v = x.reshape(-1, n_particles, 3)
    iu = torch.triu_indices(n_particles, n_particles, offset=1)
    d = (v[:, iu[0]] - v[:, iu[1]]).pow(2).sum(-1).sqrt()
    lj = ((1.0 / d) ** 12 - 2 * (1.0 / d) ** 6).sum(dim=-1)
    osc = 0.5 * (v - v.mean(dim=1, keepdim=True)).pow(2).sum(dim=(-2, -1))
    return -(2 * lj + osc)


# --------------------------------------------------------------------------------------
# DW4 target (PARITY UNPINNED: not in the reference tree; bgflow.MultiDoubleWellPotential)
# --------------------------------------------------------------------------------------


def dw4_logp_force(x: Tensor, n_particles: int = 4, n_dim: int = 2, temperature: float = 1.0,
                   a: float = 0.9, b: float = -4.0, c: float = 0.0, offset: float = 4.0) -> Tuple[Tensor, Tensor]:
    """E = sum_{i<j} a (d_ij-offset)^4 + b (d_ij-offset)^2 + c ;  logp = -E/T ; plain
    Euclidean d_ij (no eps)."""
    B = x.shape[0]
    v = x.reshape(B, n_particles, n_dim)
    diff = v[:, :, None, :] - v[:, None, :, :]
    eye = torch.eye(n_particles, dtype=torch.bool)
    d = diff.pow(2).sum(-1).masked_fill(eye, 1.0).sqrt()
    u = d - offset
    e_pair = (a * u**4 + b * u**2 + c).masked_fill(eye, 0.0)
    e = 0.5 * e_pair.sum(dim=(1, 2))  # each unordered pair once
    de = ((4 * a * u**3 + 2 * b * u) / d).masked_fill(eye, 0.0)
    grad = (de[..., None] * diff).sum(dim=2)
    return -e / temperature, (-grad / temperature).reshape(B, n_particles * n_dim)


# --------------------------------------------------------------------------------------
# A13 40-mode GMM target    gmm_energy.py:38,87-90 ; fab/fab/target_distributions/gmm.py:40-49,71-79,104
# --------------------------------------------------------------------------------------


def gmm_params(n_mixes: int = 40, dim: int = 2, loc_scaling: float = 40.0, log_var_scaling: float = 1.0,
               seed: int = 0) -> Tuple[Tensor, Tensor]:
    """(means[K,dim], scale[K,dim]): drawn exactly like the reference -- ``torch.manual_seed(0)``
    (gmm_energy.py:38) followed by ``(rand(K,dim)-0.5)*2*loc_scaling`` (gmm.py:40-41);
    scale = softplus(log_var_scaling) (gmm.py:43-44).  Uses a private generator so the
    global RNG is left alone (same Philox/MT stream as the global one seeded with 0)."""
    gen = torch.Generator().manual_seed(seed)
    means = (torch.rand((n_mixes, dim), generator=gen) - 0.5) * 2 * loc_scaling
    scale = torch.nn.functional.softplus(torch.ones((n_mixes, dim)) * log_var_scaling)
    return means, scale


def gmm_logp(x: Tensor, means: Tensor, scale: Tensor, temperature: float = 1.0) -> Tensor:
    """MixtureSameFamily(Categorical(equal), MVN(diag scale)).log_prob(x) / T  (gmm.py:71-79,104; gmm_energy.py:87-90)."""
    K, dim = means.shape
    z = (x[:, None, :] - means[None]) / scale[None]
    comp = -0.5 * z.pow(2).sum(-1) - scale.log().sum(-1)[None] - 0.5 * dim * math.log(2 * math.pi)
    return (torch.logsumexp(comp, dim=-1) - math.log(K)) / temperature


def gmm_logp_force(x: Tensor, means: Tensor, scale: Tensor, temperature: float = 1.0) -> Tuple[Tensor, Tensor]:
    """logp and its gradient (the reference GMM has no ``return_force``; the gradient is
    what ``BaseEnergyFunction.score`` (base_energy_function.py:149-152) would give)."""
    K, dim = means.shape
    dx = x[:, None, :] - means[None]
    z = dx / scale[None]
    comp = -0.5 * z.pow(2).sum(-1) - scale.log().sum(-1)[None] - 0.5 * dim * math.log(2 * math.pi)
    w = torch.softmax(comp, dim=-1)
    grad = -(w[..., None] * dx / scale[None] ** 2).sum(dim=1)
    return (torch.logsumexp(comp, dim=-1) - math.log(K)) / temperature, grad / temperature


# --------------------------------------------------------------------------------------
# A6  EGNN backbone      egnn_temp_conditioned.py:56-93,172-194,321-356 (egnn.py = no beta)
# --------------------------------------------------------------------------------------


def egnn_edges(n: int) -> Tuple[Tensor, Tensor]:
    """Edge order of the reference: for i<j emit (i,j) then (j,i)  (egnn_temp_conditioned.py:95-103)."""
    rows, cols = [], []
    for i in range(n):
        for j in range(i + 1, n):
            rows += [i, j]
            cols += [j, i]
    return torch.tensor(rows, dtype=torch.long), torch.tensor(cols, dtype=torch.long)


def egnn_node_features(t: Tensor, beta: Optional[Tensor], n: int, layout: str = "pita") -> Tensor:
    """Initial node features ``h0[B*n, in_nf]``.

    With temperature conditioning the reference concatenates two ``[B,n]`` tensors along
    the LAST dim and reshapes ``[B,2n] -> [B*n,2]`` (egnn_temp_conditioned.py:63-78), so
    node k of a walker gets elements (2k, 2k+1) of ``[t]*n + [beta]*n``: for n=13 nodes 0-5
    see (t,t), node 6 sees (t,beta), nodes 7-12 see (beta,beta).  Kept (quirk Q1)."""
    B = t.shape[0]
    h = torch.ones(B, n, dtype=t.dtype) * t[:, None]
    if beta is None:
        return h.reshape(B * n, 1)
    hb = torch.ones(B, n, dtype=t.dtype) * beta[:, None]
    if layout == "correct":  # (t, beta) on every node -- what egnn_aldp.py:157-167 does; NOT the LJ reference behaviour
        return torch.stack([h, hb], dim=-1).reshape(B * n, 2)
    return torch.cat([h, hb], dim=-1).reshape(B * n, 2)


def _silu(v: Tensor) -> Tensor:
    return v * torch.sigmoid(v)


def egnn_forward(p: Dict[str, Tensor], t: Tensor, x: Tensor, beta: Optional[Tensor], n: int, d: int,
                 n_layers: int = 3, coords_range: float = 15.0, tanh: bool = True, attention: bool = True,
                 return_h0: bool = False, feature_layout: str = "pita", h0: Optional[Tensor] = None):
    """EGNN_dynamics.forward: velocity ``x_final - x`` made mean-free.

    ``p`` is the reference ``state_dict`` (keys ``egnn.embedding.weight`` ...).  Materialises
    per-edge tensors exactly like the reference (gather h[row], h[col], cat, Linear, SiLU,
    Linear, SiLU, sigmoid gate, coord MLP with tanh * coords_range/n_layers, index_add
    aggregation) so that fp32 results agree with it to rounding.
    """
    B = x.shape[0]
    dt = x.dtype
    P = {k: v.to(dt) for k, v in p.items()}
    row1, col1 = egnn_edges(n)
    off = (torch.arange(B) * n)[:, None]
    row = (row1[None] + off).reshape(-1)
    col = (col1[None] + off).reshape(-1)
    pos = x.reshape(B * n, d).clone()
    pos0 = pos
    # h0: node features given by the caller (EGNN_dynamics_AD2_cat: [static one-hot rows, t, beta], see egnn_ad2_cat_features)
    h = egnn_node_features(t, beta, n, feature_layout) if h0 is None else h0.to(dt)
    h0 = h
    edge_attr = ((pos[row] - pos[col]) ** 2).sum(dim=1, keepdim=True)  # :79 (frozen, quirk Q10)
    h = h @ P["egnn.embedding.weight"].T + P["egnn.embedding.bias"]  # :179
    rng = float(coords_range) / n_layers  # :143
    for l in range(n_layers):
        g = f"egnn.gcl_{l}."
        diff = pos[row] - pos[col]  # :350
        radial = (diff**2).sum(1, keepdim=True)  # :351
        diff = diff / (torch.sqrt(radial + 1e-8) + 1)  # :353-354
        e_in = torch.cat([h[row], h[col], radial, edge_attr], dim=1)  # :270
        m = _silu(e_in @ P[g + "edge_mlp.0.weight"].T + P[g + "edge_mlp.0.bias"])
        m = _silu(m @ P[g + "edge_mlp.2.weight"].T + P[g + "edge_mlp.2.bias"])  # :271
        if attention:
            att = torch.sigmoid(m @ P[g + "att_mlp.0.weight"].T + P[g + "att_mlp.0.bias"])  # :273-275
            m = m * att
        c = _silu(m @ P[g + "coord_mlp.0.weight"].T + P[g + "coord_mlp.0.bias"]) @ P[g + "coord_mlp.2.weight"].T
        if tanh:
            c = torch.tanh(c) * rng  # :252-254,:297-298
        trans = diff * c
        pos = pos + torch.zeros_like(pos).index_add(0, row, trans)  # :306,:318 (scatter_add over row)
        agg = torch.zeros(B * n, m.shape[1], dtype=dt).index_add(0, row, m)  # :284
        nin = torch.cat([h, agg], dim=1)  # :288
        out = _silu(nin @ P[g + "node_mlp.0.weight"].T + P[g + "node_mlp.0.bias"]) @ P[g + "node_mlp.2.weight"].T \
            + P[g + "node_mlp.2.bias"]
        h = h + out  # :290-291 recurrent
    vel = (pos - pos0).reshape(B, n, d)  # :81-83
    vel = vel - vel.mean(dim=1, keepdim=True)  # :84
    vel = vel.reshape(B, n * d)
    if return_h0:
        return vel, h0
    return vel


def egnn_ad2_cat_h_initial(n: int) -> Tensor:
    """Static node features of EGNN_dynamics_AD2_cat.get_h_initial (egnn_dynamics_ad2_cat.py:66-92) for the particle
    counts that need no topology file: one-hot atom types with the methyl hydrogens merged."""
    groups = {22: [([0, 2, 3], 2), ([19, 20, 21], 20), ([11, 12, 13], 12)],
              33: [([1, 2, 3], 2), ([9, 10, 11], 10), ([19, 20, 21], 18), ([29, 30, 31], 31)],
              42: [([1, 2, 3], 2), ([11, 12, 13], 12), ([21, 22, 23], 22), ([31, 32, 33], 32), ([39, 40, 41], 40)]}
    if n in (13, 55):
        return torch.zeros(n, 1)
    at = np.arange(n)
    for idx, v in groups[n]:
        at[idx] = v
    return torch.nn.functional.one_hot(torch.tensor(at)).to(torch.float32)


def egnn_aldp_h_initial(n: int) -> Tensor:
    """Static node features of ``egnn_aldp.EGNN_dynamics.get_h_initial`` (egnn_aldp.py:51-78): as
    ``egnn_ad2_cat_h_initial`` except for the methyl grouping of the 22-atom system ([1, 2, 3] instead of [0, 2, 3]).
    The module's forward (:131-158) is ``egnn_ad2_cat_forward`` with this table."""
    groups = {22: [([1, 2, 3], 2), ([19, 20, 21], 20), ([11, 12, 13], 12)],
              33: [([1, 2, 3], 2), ([9, 10, 11], 10), ([19, 20, 21], 18), ([29, 30, 31], 31)],
              42: [([1, 2, 3], 2), ([11, 12, 13], 12), ([21, 22, 23], 22), ([31, 32, 33], 32), ([39, 40, 41], 40)]}
    if n in (13, 55):
        return torch.zeros(n, 1)
    at = np.arange(n)
    for idx, v in groups[n]:
        at[idx] = v
    return torch.nn.functional.one_hot(torch.tensor(at)).to(torch.float32)


def egnn_ad2_cat_forward(p: Dict[str, Tensor], t: Tensor, x: Tensor, beta: Optional[Tensor], n: int, d: int,
                         n_layers: int = 5, tanh: bool = True, attention: bool = True,
                         h_initial: Optional[Tensor] = None) -> Tensor:
    """EGNN_dynamics_AD2_cat.forward (egnn_dynamics_ad2_cat.py:157-203): node features [h_initial, t(, beta)] per node,
    then the same EGNN / E_GCL stack as ``egnn_forward`` (egnn.py and egnn_temp_conditioned.py share the layer)."""
    B = x.shape[0]
    hi = (egnn_ad2_cat_h_initial(n) if h_initial is None else h_initial).to(x.dtype)
    cols = [hi[None].expand(B, n, hi.shape[1]), t.to(x.dtype)[:, None, None].expand(B, n, 1)]
    if beta is not None:
        cols.append(beta.to(x.dtype)[:, None, None].expand(B, n, 1))
    h0 = torch.cat(cols, dim=-1).reshape(B * n, -1)
    return egnn_forward(p, t, x, beta, n, d, n_layers=n_layers, tanh=tanh, attention=attention, h0=h0)


# --------------------------------------------------------------------------------------
# A7  MLP backbone       mlp.py:11-24 (SinusoidalEmbedding), :100-118 (Block), :244-267 / :501-524
# --------------------------------------------------------------------------------------


def sinusoidal_embedding(v: Tensor, size: int, scale: float) -> Tensor:
    """mlp.py:17-24: f_k = exp(-ln(1e4)/(half-1) * k), emb = [sin(v*scale*f), cos(v*scale*f)].
    NB: the reference builds the frequency table in float32 whatever the input dtype."""
    half = size // 2
    w = torch.log(torch.tensor([10000.0])) / (half - 1)
    f = torch.exp(-w * torch.arange(half)).to(v.dtype)
    e = (v * scale)[:, None] * f[None]
    return torch.cat([torch.sin(e), torch.cos(e)], dim=-1)


def _gelu(v: Tensor) -> Tensor:
    return torch.nn.functional.gelu(v)  # exact erf GELU (nn.GELU default)


def mlp_forward(p: Dict[str, Tensor], t: Tensor, x: Tensor, beta: Optional[Tensor] = None, emb_size: int = 128,
                hidden_layers: int = 3, temperature_conditioned: bool = False) -> Tensor:
    """MyMLP.forward (mlp.py:244-267) / MyMLPTemperature.forward (:501-524).

    ``p``: reference state_dict (``joint_mlp.0.weight`` [H, C], ``joint_mlp.{1..L}.ff.weight``,
    ``joint_mlp.{L+1}.weight``).  Input coords embed with scale 25, time (and beta) with scale 1;
    concat order: x_0.., x_{D-1}, t, (beta).  MyMLP ignores beta (it lands in ``x_self_cond``).
    """
    dt = x.dtype
    P = {k: v.to(dt) for k, v in p.items()}
    embs = [sinusoidal_embedding(x[:, i], emb_size, 25.0) for i in range(x.shape[-1])]
    embs.append(sinusoidal_embedding(t, emb_size, 1.0))
    if temperature_conditioned:
        embs.append(sinusoidal_embedding(beta, emb_size, 1.0))
    z = torch.cat(embs, dim=-1)
    z = _gelu(z @ P["joint_mlp.0.weight"].T + P["joint_mlp.0.bias"])
    for l in range(1, hidden_layers + 1):
        z = z + _gelu(z @ P[f"joint_mlp.{l}.ff.weight"].T + P[f"joint_mlp.{l}.ff.bias"])
    L = hidden_layers + 1
    return z @ P[f"joint_mlp.{L}.weight"].T + P[f"joint_mlp.{L}.bias"]


# --------------------------------------------------------------------------------------
# A5  EDM preconditioning    score_net.py:13-43, energy_net.py:14-49
# --------------------------------------------------------------------------------------

Backbone = Callable[[Tensor, Tensor, Tensor], Tensor]  # (c_noise[B], x_scaled[B,D], beta[B]) -> [B,D]


def edm_coeffs(h: Tensor):
    """c_s, c_in, c_out, c_noise  (score_net.py:26-29)."""
    c_s = 1 / (1 + h)
    c_in = 1 / (1 + h) ** 0.5
    c_out = h**0.5 * c_in
    c_noise = (1 / 8) * torch.log(h)
    return c_s, c_in, c_out, c_noise


def denoiser(backbone: Backbone, h: Tensor, x: Tensor, beta, precondition_beta: bool = False) -> Tensor:
    """ScoreNet.denoiser (score_net.py:21-43)."""
    b = beta * torch.ones(x.shape[0], dtype=x.dtype)
    c_s, c_in, c_out, c_noise = edm_coeffs(h)
    D = c_s[:, None] * x + c_out[:, None] * backbone(c_noise, c_in[:, None] * x, b)
    if precondition_beta:
        D = D * b[:, None] + (1 - b[:, None]) * x
    return D


def score(backbone: Backbone, h: Tensor, x: Tensor, beta, precondition_beta: bool = False) -> Tensor:
    """ScoreNet.forward: (D_theta - x)/h  (score_net.py:13-19).  NB forward() uses the
    (possibly beta-preconditioned) denoiser, not the ``score`` local of denoiser()."""
    return (denoiser(backbone, h, x, beta, precondition_beta) - x) / h[:, None]


def energy_theta(backbone: Backbone, h: Tensor, x: Tensor, beta, precondition_beta: bool = False) -> Tensor:
    """EnergyNet.forward_energy without pinning (energy_net.py:14-49)."""
    b = beta * torch.ones(x.shape[0], dtype=x.dtype)
    c_s, c_in, c_out, c_noise = edm_coeffs(h)
    xs = c_in[:, None] * x
    U = (backbone(c_noise, xs, b) * xs).sum(dim=1)
    E = (1 - c_s) / (2 * h) * torch.linalg.norm(x, dim=-1) ** 2 - c_out / (c_in * h) * U
    if precondition_beta:
        E = E * b
    return E


# --------------------------------------------------------------------------------------
# A15 systematic resampling, quantile clamp     utils.py:111-120 ; sdes.py:230 ; sde_integration.py:179
# --------------------------------------------------------------------------------------


def sample_cat_sys(logits: Tensor, u0: float) -> np.ndarray:
    """ids[B] for one uniform ``u0`` (float64).  utils.py:111-120: u = (u0 + arange/B) mod 1
    in float64; weights = clip(softmax(logits),1e-6,1) NOT renormalised; bins = cumsum
    (in the dtype of logits); ids = digitize(u, bins, right=True); id==B -> B-1."""
    bs = logits.shape[-1]
    u = (torch.tensor([u0], dtype=torch.float64) + 1 / bs * torch.arange(bs)) % 1.0
    w = torch.clip(torch.softmax(logits, dim=-1), 1e-6, 1.0)
    bins = torch.cumsum(w, dim=-1)
    ids = np.digitize(u, bins, right=True)
    ids[ids == bs] = bs - 1
    return ids


def quantile_clamp(a: Tensor, q: float = 0.9) -> Tensor:
    """clamp(a, max=quantile(a, q)) with torch's default linear interpolation (sdes.py:230)."""
    return torch.clamp(a, max=torch.quantile(a, q))


# --------------------------------------------------------------------------------------
# A4  reverse-SDE terms      sdes.py:117-128 (not debiased), :130-239 (debiased), :245-251
# --------------------------------------------------------------------------------------


@dataclass
class Terms:  # sdes.py:34-41 (SDETerms)
    drift_X: Tensor
    drift_A: Tensor
    divergence_score: Optional[Tensor] = None
    cross_term: Optional[Tensor] = None
    dUt_dt: Optional[Tensor] = None
    diffusion: Optional[Tensor] = None


def f_not_debiased(backbone: Backbone, sched, gamma_sched, t: Tensor, x: Tensor, beta,
                   precondition_beta: bool = False) -> Terms:
    """drift_X = gamma(t) * s_theta(h(t), x, beta) * g(t)^2 ; drift_A = 0   (sdes.py:117-128,140-149)."""
    gamma = gamma_sched.gamma(t)
    tb = t * torch.ones(x.shape[0], dtype=x.dtype)
    ht = sched.h(tb)
    s = score(backbone, ht, x, beta, precondition_beta)
    return Terms(drift_X=gamma * (s * sched.g(tb).pow(2).unsqueeze(-1)), drift_A=torch.zeros(x.shape[0], dtype=x.dtype))


def f_debiased(score_backbone: Backbone, energy_backbone: Backbone, sched, gamma_sched, t: Tensor, x: Tensor,
               beta, clamp_quantile: Optional[float] = 0.9, pin_energy: bool = False, target_logp=None,
               precondition_beta: bool = False) -> Terms:
    """Feynman-Kac corrected drift (sdes.py:151-239): needs grad_x E_theta, exact div s_theta
    (vmap(jacrev), utils.py:43-51), dE_theta/dt through h(t).  ``pin_energy`` (energy_net.py:43-48) blends the model
    energy with the clamped target energy U0 = clamp(-log p_target(x), +-1e3); ``target_logp`` is called like the
    reference's energy classes, whose result is DETACHED (lennardjones_energy.py:227), so no target force enters
    grad_x U.  ``precondition_beta`` applies to both nets (score_net.py:36-38, energy_net.py:40-41)."""
    from torch.func import jacrev, vmap

    gamma = gamma_sched.gamma(t)

    def energy(ht, xg, tb):
        E = energy_theta(energy_backbone, ht, xg, beta, precondition_beta)
        if not pin_energy:
            return E
        U0 = torch.clamp(-target_logp(xg.detach()).detach().to(E.dtype), max=1e3, min=-1e3)
        return (1 - tb) ** 3 * U0 + (1 - (1 - tb) ** 3) * E

    with torch.enable_grad():
        xg = x.detach().clone().requires_grad_(True)
        tb = (t * torch.ones(x.shape[0], dtype=x.dtype)).detach().requires_grad_(True)
        ht = sched.h(tb)
        g2 = sched.g(tb).pow(2)
        Ut = energy(ht, xg, tb)
        nabla_U = torch.autograd.grad(Ut.sum(), xg, create_graph=True)[0]
        s_t = score(score_backbone, ht, xg, beta, precondition_beta)
        bt = s_t * g2.unsqueeze(-1) / 2
        drift_X = (gamma * -nabla_U * g2.unsqueeze(-1) / 2 + gamma * bt).detach()
        Ut2 = energy(ht, xg, tb)
        dUt_dt = torch.autograd.grad(Ut2.sum(), tb)[0].detach()

        def one(h1, x1):
            return score(score_backbone, h1.unsqueeze(0), x1.unsqueeze(0), beta, precondition_beta).squeeze(0)

        jac = vmap(jacrev(one, argnums=1))(ht.detach(), x.detach())
        div_bt = jac.diagonal(dim1=-2, dim2=-1).sum(-1).detach() * g2.detach() / 2
        inner = (-nabla_U * bt).sum(-1).detach()
        drift_A = (gamma * gamma * inner + gamma * div_bt + gamma * dUt_dt
                   + gamma_sched.dgamma_dt(tb.detach()) * Ut2.detach())
        if clamp_quantile is not None:
            drift_A = quantile_clamp(drift_A, clamp_quantile)
    return Terms(drift_X=drift_X, drift_A=drift_A.detach(), divergence_score=div_bt, cross_term=inner, dUt_dt=dUt_dt)


# --------------------------------------------------------------------------------------
# A1-A3 integrator, A16 post-processing      sde_integration.py:98-212,214-297,299-351,353-470
# --------------------------------------------------------------------------------------

NoiseFn = Callable[[int, Tuple[int, ...]], Tensor]  # (draw index, shape) -> standard normal draws


@dataclass
class IntegratorConfig:
    num_integration_steps: int
    start_resampling_step: int = 0
    end_resampling_step: int = 10**9
    diffusion_scale: float = 1.0
    time_range: float = 1.0
    resampling_interval: int = -1
    batch_size: Optional[int] = None  # inference chunk (quantile clamp is per chunk)
    should_mean_free: bool = True
    debias: bool = False


def integrate_sde(cfg: IntegratorConfig, x1: Tensor, drift_fn: Callable[[Tensor, Tensor], Terms], g_fn, noise_fn: NoiseFn,
                  n_particles: int, n_dim: int, uniform_fn: Optional[Callable[[int], float]] = None,
                  record: bool = False):
    """Euler-Maruyama loop of WeightedSDEIntegrator.integrate_sde (single rank).

    times = linspace(T, 0, N+1)[:-1], dt = T/N (sde_integration.py:115-122); per step
    x += drift_X*dt + scale*g(t)*xi*sqrt(dt), a += drift_A*dt (:347-349); window gates
    (:278-282); systematic resampling when due (:283-297); remove_mean (:148).
    ``drift_fn(t, x_chunk)`` is evaluated per inference chunk like :312-343; noise is drawn
    per chunk (one ``noise_fn`` call per chunk, in order).
    """
    N = cfg.num_integration_steps
    dt_ = x1.dtype
    times = torch.linspace(cfg.time_range, 0.0, N + 1)[:-1].to(dt_)
    dt = cfg.time_range / N
    x = x1.clone()
    a = torch.zeros(x.shape[0], dtype=dt_)
    bs = cfg.batch_size or x.shape[0]
    logw, uniq, traj, drifts = [], [], [], []
    draw = 0
    for step, t in enumerate(times):
        dX, dA, dif = [], [], []
        for lo in range(0, x.shape[0], bs):
            xc = x[lo:lo + bs]
            terms = drift_fn(t, xc)
            tb = t * torch.ones(xc.shape[0], dtype=dt_)
            dif.append(cfg.diffusion_scale * g_fn(tb)[:, None] * noise_fn(draw, tuple(xc.shape)).to(dt_))
            draw += 1
            dX.append(terms.drift_X)
            dA.append(terms.drift_A)
        dX, dA, dif = torch.cat(dX), torch.cat(dA), torch.cat(dif)
        x_next = x + (dX * dt + dif * np.sqrt(dt))
        a_next = a + dA * dt
        if step < cfg.start_resampling_step:
            a_next = torch.zeros_like(a_next)
            x_next = x
        if step >= cfg.end_resampling_step:
            a_next = torch.zeros_like(a_next)
        ri = cfg.resampling_interval
        n_unique = x_next.shape[0]
        if not (ri == -1 or (step + 1) % ri != 0 or step < cfg.start_resampling_step
                or step >= cfg.end_resampling_step):
            ids = sample_cat_sys(a_next, uniform_fn(step))
            x_next = x_next[torch.from_numpy(ids)]
            a_next = torch.zeros_like(a_next)
            n_unique = len(np.unique(ids))
        x, a = x_next, a_next
        if cfg.should_mean_free:
            x = remove_mean(x, n_particles, n_dim)
        logw.append(a)
        uniq.append(n_unique)
        if record:
            traj.append(x.clone())
            drifts.append(dX)
    out = dict(x=x, logweights=torch.stack(logw), num_unique=uniq)
    if record:
        out["traj"] = torch.stack(traj)
        out["drift_X"] = torch.stack(drifts)
    return out


def resample_at_end(x: Tensor, a: Tensor, t_end: Tensor, target_logp, model_energy, gamma: float, u0: float):
    """End-of-trajectory reweighting + resampling (sde_integration.py:158-183): a_next = log p_target(x) -
    (-E_theta(h(t_end), x) gamma(t_end)) + a, clamped at its 0.9 quantile, then systematic resampling.
    ``model_energy(t, x)`` is EnergyNet.forward_energy at h(t).  Returns (x[ids], a_next, n_unique)."""
    tb = t_end * torch.ones(x.shape[0], dtype=x.dtype)
    a_next = target_logp(x) + model_energy(tb, x) * gamma + a
    a_next = quantile_clamp(a_next, 0.9)
    ids = sample_cat_sys(a_next, u0)
    return x[torch.from_numpy(ids)], a_next, len(np.unique(ids))


def negative_time_descent(x: Tensor, logp_force, n_steps: int, dt: float, n_particles: int, n_dim: int,
                          do_langevin: bool = False, noise_fn: Optional[NoiseFn] = None, mean_free: bool = True) -> Tensor:
    """sde_integration.py:353-360."""
    for k in range(n_steps):
        _, F = logp_force(x)
        x = x + F * dt
        if do_langevin:
            x = x + noise_fn(k, tuple(x.shape)).to(x.dtype) * np.sqrt(2 * dt)
        if mean_free:
            x = remove_mean(x, n_particles, n_dim)
    return x


def mala_step(x: Tensor, logp_curr: Tensor, logp_force, dt: float, noise: Tensor, log_u: Tensor):
    """One MALA proposal + accept/reject on already-valid rows (sde_integration.py:28-45,375-396).
    ``noise``: proposal normals, ``log_u``: log of the accept uniforms.  Returns
    (x_new, logp_new, accept_mask)."""
    _, grad = logp_force(x)
    x_prop = x + 0.5 * dt * grad + torch.sqrt(torch.tensor(dt)).to(x.dtype) * noise
    fwd_mean = x + 0.5 * dt * grad
    log_q_f = -((x_prop - fwd_mean) ** 2).sum(dim=1) / (2 * dt)
    logp_prop, grad_prop = logp_force(x_prop)
    bwd_mean = x_prop + 0.5 * dt * grad_prop
    log_q_b = -((x - bwd_mean) ** 2).sum(dim=1) / (2 * dt)
    ratio = (logp_prop - logp_curr) + (log_q_b - log_q_f)
    acc = log_u < ratio
    af = acc.to(x.dtype)
    x_new = af[:, None] * x_prop + (1 - af[:, None]) * x
    logp_new = af * logp_prop + (1 - af) * logp_curr
    return x_new, logp_new, acc


# --------------------------------------------------------------------------------------
# Evaluation helper (section 8(f) N4): 1-D Wasserstein-2 between energy samples
#   distribution_distances.py:13-33 uses POT ``emd2_1d`` (squared-euclid) -> sqrt.
# --------------------------------------------------------------------------------------


def w2_1d(a: np.ndarray, b: np.ndarray) -> float:
    """Exact 1-D W2 between two empirical measures with uniform weights (quantile coupling)."""
    a = np.sort(np.asarray(a, dtype=np.float64))
    b = np.sort(np.asarray(b, dtype=np.float64))
    if len(a) == len(b):
        return float(np.sqrt(np.mean((a - b) ** 2)))
    # general sizes: integrate |F_a^-1 - F_b^-1|^2 over the merged quantile grid
    qa = np.arange(1, len(a) + 1) / len(a)
    qb = np.arange(1, len(b) + 1) / len(b)
    q = np.unique(np.concatenate([qa, qb]))
    w = np.diff(np.concatenate([[0.0], q]))
    ia = np.minimum(np.searchsorted(qa, q - 1e-15), len(a) - 1)
    ib = np.minimum(np.searchsorted(qb, q - 1e-15), len(b) - 1)
    return float(np.sqrt(np.sum(w * (a[ia] - b[ib]) ** 2)))


# --------------------------------------------------------------------------------------
# A14 classical force field (PARITY UNPINNED).  The reference evaluates alanine dipeptide with OpenMM
# (amber14-all + implicit/obc1, CutoffNonPeriodic 2 nm) through bgflow (alp_energy.py:93-149); neither OpenMM nor
# the PDB / force-field XML are in the tree, so only the published functional forms of OpenMM's standard forces are
# restated here (HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce, NonbondedForce with Lorentz-Berthelot
# mixing, exceptions and the CutoffNonPeriodic reaction field; GBSAOBCForce = the OBC model I generalised-Born
# solvent of Onufriev, Bashford & Case, Proteins 55:383 (2004) with HCT pairwise descreening and the ACE
# surface-area term, as OpenMM evaluates it [restated from the published algorithm; not checkable here]).
# --------------------------------------------------------------------------------------

ONE_4PI_EPS0 = 138.935456  # kJ nm / (mol e^2), OpenMM's constant


def ff_energy(x: Tensor, ff: Dict[str, Tensor], length_scale: float = 1.0, cutoff: Optional[float] = None,
              rf_dielectric: float = 78.3) -> Tensor:
    """Potential energy [kJ/mol] per walker of a table-driven molecular force field.
    x: [B, 3n] in model units (x * length_scale = nm).  ff: index / parameter tables:
      bond_idx[nb,2], bond_par[nb,2]=(r0,k)            E = 1/2 k (r - r0)^2
      angle_idx[na,3], angle_par[na,2]=(th0,k)         E = 1/2 k (theta - th0)^2
      tors_idx[nt,4], tors_par[nt,3]=(n,phase,k)       E = k (1 + cos(n phi - phase))
      charge[n], sigma[n], epsilon[n]                  LJ 4 eps ((s/r)^12 - (s/r)^6), s=(si+sj)/2, eps=sqrt(ei ej); Coulomb
      exc_idx[ne,2], exc_par[ne,3]=(qq,sigma,eps)      pairs that replace the mixed parameters (exclusions: zeros)
    cutoff (nm): None = all pairs, plain Coulomb; else pairs beyond it are dropped and the non-exception Coulomb term is
    the reaction-field form qq (1/r + k_rf r^2 - c_rf)."""
    B = x.shape[0]
    n = ff["charge"].shape[0]
    r = (x * length_scale).reshape(B, n, 3)
    dt = r.dtype
    E = torch.zeros(B, dtype=dt)
    if ff["bond_idx"].numel():
        i, j = ff["bond_idx"][:, 0], ff["bond_idx"][:, 1]
        d = (r[:, i] - r[:, j]).norm(dim=-1)
        E = E + (0.5 * ff["bond_par"][:, 1].to(dt) * (d - ff["bond_par"][:, 0].to(dt)) ** 2).sum(-1)
    if ff["angle_idx"].numel():
        i, j, k = ff["angle_idx"].T
        a, b = r[:, i] - r[:, j], r[:, k] - r[:, j]
        cosv = (a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))
        th = torch.acos(cosv.clamp(-1.0, 1.0))
        E = E + (0.5 * ff["angle_par"][:, 1].to(dt) * (th - ff["angle_par"][:, 0].to(dt)) ** 2).sum(-1)
    if ff["tors_idx"].numel():
        i, j, k, l = ff["tors_idx"].T
        b1, b2, b3 = r[:, j] - r[:, i], r[:, k] - r[:, j], r[:, l] - r[:, k]
        n1, n2 = torch.cross(b1, b2, dim=-1), torch.cross(b2, b3, dim=-1)
        yv = (b1 * n2).sum(-1) * b2.norm(dim=-1)
        xv = (n1 * n2).sum(-1)
        phi = torch.atan2(yv, xv)
        per, ph, kk = (ff["tors_par"][:, c].to(dt) for c in range(3))
        E = E + (kk * (1 + torch.cos(per * phi - ph))).sum(-1)
    # nonbonded: all i<j pairs, exceptions override
    iu = torch.triu_indices(n, n, offset=1)
    q, sg, ep = ff["charge"].to(dt), ff["sigma"].to(dt), ff["epsilon"].to(dt)
    qq = q[iu[0]] * q[iu[1]]
    s = 0.5 * (sg[iu[0]] + sg[iu[1]])
    e = torch.sqrt(ep[iu[0]] * ep[iu[1]])
    is_exc = torch.zeros(iu.shape[1], dtype=torch.bool)
    if ff["exc_idx"].numel():
        lut = {(int(a), int(b)): t for t, (a, b) in enumerate(zip(iu[0], iu[1]))}
        for t, (a, b) in enumerate(ff["exc_idx"].tolist()):
            p = lut[(min(a, b), max(a, b))]
            is_exc[p] = True
            qq[p], s[p], e[p] = ff["exc_par"][t, 0].to(dt), ff["exc_par"][t, 1].to(dt), ff["exc_par"][t, 2].to(dt)
    d = (r[:, iu[0]] - r[:, iu[1]]).norm(dim=-1)
    sr6 = (s / d) ** 6
    lj = 4 * e * (sr6 * sr6 - sr6)
    if cutoff is None:
        coul = ONE_4PI_EPS0 * qq / d
        E = E + (lj + coul).sum(-1)
    else:
        krf = (1.0 / cutoff**3) * (rf_dielectric - 1) / (2 * rf_dielectric + 1)
        crf = (1.0 / cutoff) * (3 * rf_dielectric) / (2 * rf_dielectric + 1)
        coul_rf = ONE_4PI_EPS0 * qq * (1 / d + krf * d * d - crf)
        coul_plain = ONE_4PI_EPS0 * qq / d
        coul = torch.where(is_exc, coul_plain, coul_rf)
        inside = (d < cutoff).to(dt)
        E = E + ((lj + coul) * inside).sum(-1)
    if "gb_radius" in ff:
        E = E + gbsa_obc1_energy(r, q, ff["gb_radius"].to(dt), ff["gb_scale"].to(dt), cutoff,
                                 float(ff.get("gb_solute_dielectric", 1.0)), float(ff.get("gb_solvent_dielectric", 78.5)),
                                 float(ff.get("gb_surface_area_factor", 28.3919551)))
    return E


def gbsa_obc1_energy(r: Tensor, q: Tensor, radius: Tensor, scale: Tensor, cutoff: Optional[float] = None,
                     solute_eps: float = 1.0, solvent_eps: float = 78.5, sa_factor: float = 28.3919551,
                     probe: float = 0.14) -> Tensor:
    """GB-OBC (model I: alpha=0.8, beta=0, gamma=2.909125) + ACE surface area, kJ/mol per walker.  r: [B, n, 3] nm.
      rho_i = R_i - 0.009, sigma_j = s_j rho_j;  I_i = sum_{j != i, rho_i < r + sigma_j} H(r_ij, rho_i, sigma_j)  (HCT)
      psi_i = rho_i I_i / 2;  B_i = 1 / (1/rho_i - tanh(alpha psi - beta psi^2 + gamma psi^3) / R_i)
      E = -k_e (1/eps_in - 1/eps_out) [ sum_{i<j} q_i q_j / f_ij + 1/2 sum_i q_i^2 / B_i ] + sum_i sa (R_i + probe)^2 (R_i/B_i)^6,
      f_ij = sqrt(r^2 + B_i B_j exp(-r^2 / (4 B_i B_j))); with a cutoff, pairs beyond it are dropped everywhere and the
      pair term is shifted by -q_i q_j / cutoff."""
    Bn, n = r.shape[0], r.shape[1]
    dt = r.dtype
    eye = torch.eye(n, dtype=torch.bool)
    diff = r[:, :, None] - r[:, None]
    d = torch.sqrt((diff * diff).sum(-1) + eye.to(dt))  # diagonal padded with 1 (masked out below)
    rho = radius - 0.009
    sig = rho * scale
    rho_i, sig_j = rho[None, :, None], sig[None, None, :]
    rs = d + sig_j
    l = 1.0 / torch.maximum(rho_i.expand_as(d), (d - sig_j).abs())
    u = 1.0 / rs
    term = l - u + 0.25 * d * (u * u - l * l) + 0.5 / d * torch.log(u / l) + 0.25 * sig_j**2 / d * (l * l - u * u)
    term = term + torch.where(rho_i < sig_j - d, 2.0 * (1.0 / rho_i - l), torch.zeros_like(d))
    mask = (~eye)[None] & (rho_i < rs)
    if cutoff is not None:
        mask = mask & (d < cutoff)
    psi = 0.5 * rho * (term * mask.to(dt)).sum(-1)
    born = 1.0 / (1.0 / rho - torch.tanh(0.8 * psi + 2.909125 * psi**3) / radius)
    pf = -ONE_4PI_EPS0 * (1.0 / solute_eps - 1.0 / solvent_eps)
    iu = torch.triu_indices(n, n, offset=1)
    dij = d[:, iu[0], iu[1]]
    a2 = born[:, iu[0]] * born[:, iu[1]]
    f = torch.sqrt(dij * dij + a2 * torch.exp(-dij * dij / (4 * a2)))
    c = pf * q[iu[0]] * q[iu[1]]
    pair = c / f
    if cutoff is not None:
        pair = (pair - c / cutoff) * (dij < cutoff).to(dt)
    e_self = 0.5 * pf * (q * q / born).sum(-1)
    e_sa = (sa_factor * (radius + probe) ** 2 * (radius / born) ** 6).sum(-1)
    return pair.sum(-1) + e_self + e_sa


def ff_logp_force(x: Tensor, ff: Dict[str, Tensor], kT: float, length_scale: float = 1.0, cutoff: Optional[float] = None,
                  rf_dielectric: float = 78.3) -> Tuple[Tensor, Tensor]:
    """(logp, d logp / dx) with logp = -E/kT; forces by autograd (the test-side ground truth)."""
    xg = x.detach().clone().requires_grad_(True)
    lp = -ff_energy(xg, ff, length_scale, cutoff, rf_dielectric) / kT
    (g,) = torch.autograd.grad(lp.sum(), xg)
    return lp.detach(), g
