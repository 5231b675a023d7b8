"""Gaussian prior of the reverse SDE (mirror of pita/src/energies/base_prior.py:8-83).

``Prior(scale, n_particles, spatial_dim, device, should_mean_free).sample(n)`` draws
``scale * N(0, I)`` and, for particle systems, removes the per-walker particle mean
(MeanFreePrior.sample :77-83).  Draws come from the library's Philox4x32-10 stream keyed by
(seed, global walker index): the same walker gets the same draw however the batch is sharded.
Pass ``noise=`` to ``sample`` to reproduce a given randn draw (parity tests).
"""
import math

import torch

from . import _lib


class Prior:
    def __init__(self, scale, n_particles=None, spatial_dim=None, dim=None, device="cuda", should_mean_free=True,
                 seed=0):
        if n_particles is None or spatial_dim is None:
            # the reference computes n_particles*spatial_dim unconditionally (base_prior.py:29)
            raise AssertionError("Prior: n_particles and spatial_dim must be provided")
        self.n_particles, self.spatial_dim = n_particles, spatial_dim
        self.dim = n_particles * spatial_dim
        self.scale = float(scale)
        self.device = device
        self.should_mean_free = should_mean_free
        self.seed = seed
        self._calls = 0

    def sample(self, n_samples, noise=None, walker_offset=0):
        n = int(n_samples[0]) if not isinstance(n_samples, int) else n_samples
        x = torch.empty(n, self.dim, device=self.device, dtype=torch.float32)
        if noise is not None:
            noise = _lib.dev_tensor(noise, "noise")
        # a fresh stream per call: (seed, call counter) -> 64-bit key
        key = (int(self.seed) * 0x9E3779B97F4A7C15 + self._calls) & 0xFFFFFFFFFFFFFFFF
        self._calls += 1
        _lib.check(_lib.lib().pita_prior_sample(x.data_ptr(), _lib.ptr(noise), n, self.n_particles, self.spatial_dim,
                                                self.scale, key, walker_offset, int(self.should_mean_free),
                                                _lib.stream_ptr(x.device)), "pita_prior_sample")
        return x

    def log_prob(self, x):
        """base_prior.py:56-75 (mean-free) / isotropic normal; evaluation helper in torch ops."""
        r2 = (x.reshape(x.shape[0], -1) ** 2).sum(-1) / self.scale**2
        dof = (self.n_particles - 1) * self.spatial_dim if self.should_mean_free else self.dim
        return -0.5 * r2 - 0.5 * dof * math.log(2 * math.pi * self.scale**2)
