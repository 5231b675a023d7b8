"""Annealing-factor schedules gamma(t) = beta_low/beta_high ramp (host-side scalars).

Mirror of ``BaseAnnealingFactorSchedule.gamma/dgamma_dt``
(pita/src/models/components/annealing_factor_schedules.py:8-17) and the three concrete
schedules (:20-109).  Evaluated once per SDE step on the host; the value travels to the GPU
in the per-step table (PITA_ST_GAMMA).
"""
import torch


def _as_tensor(t):
    return t if isinstance(t, torch.Tensor) else torch.tensor(t)


class BaseAnnealingFactorSchedule:
    """gamma(t) multiplies the score in the annealed reverse SDE (drift = gamma s_theta g^2): gamma = beta_low / beta_high
    transports samples of temperature 1/beta_high to 1/beta_low.  The sampler evaluates gamma (and gamma' in the
    Feynman-Kac weights) on the host, once per step, in fp32 tensors like the reference -- the values land in column
    PITA_ST_GAMMA of the step table consumed by the fused kernels."""

    def gamma(self, t):
        raise NotImplementedError

    def dgamma_dt(self, t):
        raise NotImplementedError


class ConstantAnnealingFactorSchedule(BaseAnnealingFactorSchedule):  # :20-32
    def __init__(self, annealing_factor):
        self.annealing_factor = annealing_factor

    def gamma(self, t):
        return torch.ones_like(_as_tensor(t)) * self.annealing_factor

    def dgamma_dt(self, t):
        return torch.zeros_like(_as_tensor(t))


class _Ramp(BaseAnnealingFactorSchedule):
    """Common state of the time-dependent schedules: gamma moves from ``annealing_factor_start`` at ``t_start`` to
    ``annealing_factor`` at ``t_end`` (reverse time: t_start > t_end)."""

    def __init__(self, annealing_factor, annealing_factor_start, t_start=1.0, t_end=0.0):
        self.annealing_factor = annealing_factor
        self.annealing_factor_start = annealing_factor_start
        self.t_start, self.t_end = t_start, t_end

    @property
    def _delta(self):
        return self.annealing_factor - self.annealing_factor_start


class LinearAnnealingFactorSchedule(_Ramp):  # :35-69 (reverse time: t runs t_start -> t_end)
    def gamma(self, t):
        t = _as_tensor(t)
        slope = self._delta / (self.t_end - self.t_start)
        ramp = slope * (t - self.t_start) + self.annealing_factor_start
        inside = torch.where(t < self.t_end, self.annealing_factor, ramp)
        return torch.where(t > self.t_start, self.annealing_factor_start, inside)

    def dgamma_dt(self, t):
        t = _as_tensor(t)
        slope = self._delta / (self.t_end - self.t_start)
        zero = torch.tensor(0.0, dtype=t.dtype)
        return torch.where(t > self.t_start, zero, torch.where(t < self.t_end, zero, slope))


class SigmoidAnnealingFactorSchedule(_Ramp):  # :72-109
    def __init__(self, annealing_factor, annealing_factor_start, t_start=1.0, t_end=0.0, sharpness=10.0):
        super().__init__(annealing_factor, annealing_factor_start, t_start, t_end)
        self.center = (t_start + t_end) / 2
        self.width = t_start - t_end
        self.sharpness = sharpness

    def _smooth(self, t):
        return 1 / (1 + torch.exp(-self.sharpness * ((self.center - t) / self.width)))

    def gamma(self, t):
        return self.annealing_factor_start + self._delta * self._smooth(_as_tensor(t))

    def dgamma_dt(self, t):
        s = self._smooth(_as_tensor(t))
        return self._delta * ((self.sharpness / self.width) * s * (1 - s))
