"""Lennard-Jones cluster target evaluated by the HIP pairwise kernel.

Mirror of ``LennardJonesEnergy`` (pita/src/energies/lennardjones_energy.py:158-227): same
constructor arguments, ``__call__(samples, return_force=False)`` returns the detached
log-density (and analytic force instead of autograd).  The arithmetic of
``LennardJonesPotential._energy`` (:121-146) + bgflow's ``distances_from_vectors`` is in
pita_amd/csrc/energy_kernels.hip (pita_lj_logp_force).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .base_energy_function import BaseMoleculeEnergy


def smooth_core_coefficients(range_min=0.65, range_max=2.0, interpolation=1000, eps=1.0, rm=1.0):
    """The four coefficients (float32, highest power first) of the cubic the reference uses below ``range_min`` when
    ``smooth=True``: first interval of scipy's CubicSpline through the LJ curve sampled on
    ``torch.linspace(range_min, range_max, interpolation)`` in float32 (LennardJonesPotential.__init__,
    lennardjones_energy.py:114-119); ``cubic_spline`` (:39-54) clamps every r < range_min to that interval."""
    from scipy.interpolate import CubicSpline  # the reference's own dependency for this option
    pts = torch.linspace(range_min, range_max, interpolation)
    es = eps * ((rm / pts) ** 12 - 2 * (rm / pts) ** 6)
    c = CubicSpline(pts.numpy(), es.numpy()).c
    return np.ascontiguousarray(torch.tensor(c).float().numpy()[:, 0]), float(pts[0])


class LennardJonesEnergy(BaseMoleculeEnergy):
    def __init__(self, dimensionality, n_particles, spatial_dim, data_path=None, device="cuda",
                 plot_samples_epoch_period=5, plotting_buffer_sample_size=512, energy_factor=1.0, is_molecule=True,
                 smooth=False, temperature=1.0, should_normalize=False, data_normalization_factor=1.0,
                 dist_eps=1e-6, *args, **kwargs):
        if n_particles not in (13, 55):  # lennardjones_energy.py:177-182
            raise NotImplementedError("LennardJonesEnergy: the reference defines LJ13 and LJ55 only")
        self.name = "LJ13_efm" if n_particles == 13 else "LJ55"
        super().__init__(dimensionality=dimensionality, n_particles=n_particles, spatial_dim=spatial_dim,
                         data_path=data_path, data_name="LJ", device=device, is_molecule=is_molecule,
                         temperature=temperature, should_normalize=should_normalize,
                         data_normalization_factor=data_normalization_factor)
        self.energy_factor = float(energy_factor)
        self.dist_eps = float(dist_eps)  # bgflow distances_from_vectors eps
        self.smooth = bool(smooth)
        if self.smooth:  # LennardJonesPotential defaults: range_min=0.65, range_max=2.0, interpolation=1000
            self._smooth_coef, self._smooth_min = smooth_core_coefficients()
        self.plot_samples_epoch_period = plot_samples_epoch_period
        self.plotting_buffer_sample_size = plotting_buffer_sample_size

    def __call__(self, samples: torch.Tensor, return_force=False):
        x = _lib.dev_tensor(samples, "samples")
        if self.should_normalize:
            x = self.unnormalize(x)
        x = x.reshape(-1, self._dimensionality)
        B = x.shape[0]
        logp = torch.empty(B, device=x.device, dtype=torch.float32)
        force = torch.empty_like(x) if return_force else None
        if self.smooth:
            _lib.check(_lib.lib().pita_lj_smooth_logp_force(
                x.data_ptr(), logp.data_ptr(), _lib.ptr(force), B, self.n_particles, self.n_spatial_dim,
                float(self.temperature), self.energy_factor, self.dist_eps, 1.0, 1.0, 1.0, self._smooth_min,
                self._smooth_coef.ctypes.data_as(ctypes.c_void_p), _lib.stream_ptr(x.device)), "pita_lj_smooth_logp_force")
            return (logp, force) if return_force else logp
        _lib.check(_lib.lib().pita_lj_logp_force(
            x.data_ptr(), logp.data_ptr(), _lib.ptr(force), B, self.n_particles, self.n_spatial_dim,
            float(self.temperature), self.energy_factor, self.dist_eps, 1.0, 1.0, 1.0, _lib.stream_ptr(x.device)),
            "pita_lj_logp_force")
        return (logp, force) if return_force else logp

    def fused_descent(self, x, num_steps, dt, noise_scale, sqrt_dt, seed=0, walker_offset=0, step0=0, remove_mean=True,
                      noise=None):
        """``num_steps`` of x <- remove_mean(x + F dt + noise_scale*sqrt_dt*xi) in ONE launch, in place
        (negative_time_descent, sde_integration.py:353-360); None when the fused kernel does not apply."""
        if self.should_normalize or self.smooth:  # the fused kernels know the plain LJ curve only
            return None
        _lib.check(_lib.lib().pita_lj_descent(
            x.data_ptr(), _lib.ptr(noise), x.shape[0], self.n_particles, self.n_spatial_dim, float(self.temperature),
            self.energy_factor, self.dist_eps, 1.0, 1.0, 1.0, int(num_steps), float(dt), float(noise_scale),
            float(sqrt_dt), seed, walker_offset, step0, int(remove_mean), _lib.stream_ptr(x.device)), "pita_lj_descent")
        return x

    def fused_mala(self, x, logp, num_steps, dt_dev, adaptive, total, noise=None, uniforms=None, seed=0, walker_offset=0,
                   walker_ids=None, step0=0, remove_mean=True, rates_out=None):
        """All ``num_steps`` MALA steps in ONE launch, in place on ``x`` / ``logp`` / ``dt_dev`` (pita_lj_mala;
        metropolis_hastings_mala(_adaptive), sde_integration.py:362-470).  Returns None when the fused kernel does not
        apply (other particle numbers, normalised coordinates, an LJ13 adaptive chain too large to be co-resident): the
        caller then runs the launch-per-kernel path, which gives the same bits.  LJ13 and LJ55 have fused chains."""
        if self.should_normalize or self.smooth or self.n_particles not in (13, 55) or self.n_spatial_dim != 3:
            return None
        L = _lib.lib()
        ws = torch.empty((int(L.pita_lj_mala_workspace_bytes(int(num_steps))) + 7) // 8, device=x.device, dtype=torch.int64)
        rc = L.pita_lj_mala(x.data_ptr(), logp.data_ptr(), _lib.ptr(noise), _lib.ptr(uniforms), x.shape[0],
                            self.n_particles, 3,
                            float(self.temperature), self.energy_factor, self.dist_eps, 1.0, 1.0, 1.0, int(num_steps),
                            dt_dev.data_ptr(), int(bool(adaptive)), int(total), int(seed) & 0xFFFFFFFFFFFFFFFF,
                            int(walker_offset), _lib.ptr(walker_ids), int(step0), int(bool(remove_mean)),
                            _lib.ptr(rates_out), ws.data_ptr(), _lib.stream_ptr(x.device))
        if rc == -2:  # PITA_EUNSUPPORTED
            return None
        _lib.check(rc, "pita_lj_mala")
        return x
