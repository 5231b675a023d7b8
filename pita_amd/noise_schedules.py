"""Noise schedules sigma^2(t) = h(t), g(t) = sqrt(dh/dt) -- host-side, scalar per SDE step.

Mirror of the reference plugin interface ``BaseNoiseSchedule.g/h``
(pita/src/models/components/noise_schedules.py:7-16) and its concrete schedules (:19-138).
These run on the HOST in fp32 torch scalar ops, in the reference's op order, because the
sampler only needs one (h, g) pair per step; the values are shipped to the GPU inside the
per-step table consumed by the fused HIP sampler (include/pita_hip.h: PITA_ST_*).
Any user object with ``h(t)`` / ``g(t)`` taking torch tensors works as a plug-in.
"""
import math

import torch


class BaseNoiseSchedule:
    def g(self, t):
        raise NotImplementedError

    def h(self, t):
        raise NotImplementedError


class _PowerLaw(BaseNoiseSchedule):
    """h = beta * t**p ; g = sqrt(beta * p * t**(p-1))  (Linear p=1, Quadratic p=2, Power p)."""

    def __init__(self, beta, power):
        self.beta, self.power = beta, power

    def h(self, t):
        return self.beta * (t**self.power)

    def g(self, t):
        return torch.sqrt(self.beta * self.power * (t ** (self.power - 1)))


class LinearNoiseSchedule(_PowerLaw):  # noise_schedules.py:19-27
    def __init__(self, beta):
        super().__init__(beta, 1)

    def h(self, t):
        return self.beta * t

    def g(self, t):
        return torch.full_like(t, self.beta**0.5)


class QuadraticNoiseSchedule(_PowerLaw):  # :30-38
    def __init__(self, beta):
        super().__init__(beta, 2)

    def h(self, t):
        return self.beta * t**2

    def g(self, t):
        return torch.sqrt(self.beta * 2 * t)


class PowerNoiseSchedule(_PowerLaw):  # :41-50
    pass


class SubLinearNoiseSchedule(BaseNoiseSchedule):  # :53-61
    def __init__(self, beta):
        self.beta = beta

    def h(self, t):
        return self.beta * t**0.5

    def g(self, t):
        return torch.sqrt(self.beta * 0.5 * 1 / (t**0.5 + 1e-3))


class _LogSigmaSampling:
    """Training-side helpers kept for interface completeness (noise_schedules.py:82-95,127-138)."""

    def _ln_sigma_range(self):
        raise NotImplementedError

    def get_ln_sigmat_bins(self, num_bins):
        import numpy as np

        lo, hi = self._ln_sigma_range()
        return np.linspace(lo, hi, num_bins + 1)


class GeometricNoiseSchedule(BaseNoiseSchedule, _LogSigmaSampling):  # :62-95
    def __init__(self, sigma_min, sigma_max):
        self.sigma_min, self.sigma_max = sigma_min, sigma_max
        self.sigma_diff = sigma_max / sigma_min

    def g(self, t):
        return self.sigma_min * (self.sigma_diff**t) * ((2 * math.log(self.sigma_diff)) ** 0.5)

    def h(self, t):
        return (self.sigma_min * (((self.sigma_diff ** (2 * t)) - 1) ** 0.5)) ** 2

    def _ln_sigma_range(self):
        return math.log(self.sigma_min), math.log(self.sigma_max)

    def sample_ln_sigma(self, num_samples, device):
        lo, hi = self._ln_sigma_range()
        return torch.rand(num_samples, device=device) * (hi - lo) + lo


class ElucidatingNoiseSchedule(BaseNoiseSchedule, _LogSigmaSampling):
    """EDM / Karras: sigma(t) = (smax^(1/rho) + (1-t)(smin^(1/rho) - smax^(1/rho)))^rho, h = sigma^2.
    noise_schedules.py:98-138; defaults of configs/model/noise_schedule/elucidating.yaml."""

    def __init__(self, sigma_min, sigma_max=80.0, rho=7, P_mean=-1.2, P_std=1.2):
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho
        self.P_mean, self.P_std = P_mean, P_std
        inv = 1 / rho
        self.term1 = sigma_max**inv
        self.term2 = sigma_min**inv - sigma_max**inv

    def _base(self, t):
        return self.term1 + (1 - t) * self.term2

    def h(self, t):
        return self._base(t) ** (2 * self.rho)

    def dh_dt(self, t):
        return -2 * self.rho * self.term2 * self._base(t) ** (2 * self.rho - 1)

    def g(self, t):
        return (-2 * self.rho * self._base(t) ** (2 * self.rho - 1) * self.term2) ** 0.5

    def t(self, ht):
        return 1 - ((ht ** (1 / (2 * self.rho)) - self.term1) / self.term2)

    def _ln_sigma_range(self):
        return self.P_mean - 2 * self.P_std, self.P_mean + 2 * self.P_std

    def sample_ln_sigma(self, num_samples, device):
        return torch.randn(num_samples, device=device) * self.P_std + self.P_mean
