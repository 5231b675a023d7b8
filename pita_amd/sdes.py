"""Reverse VE-SDE terms (mirror of pita/src/models/components/sdes.py).

``VEReverseSDE.f`` returns the ``SDETerms`` of one step.  Implemented: the NOT-debiased regime
(``debias_inference=False`` -> ``f_not_debiased``, sdes.py:117-128), i.e. drift_X = gamma(t) *
s_theta(h(t), x, beta) * g(t)^2, drift_A = 0.  The debiased Feynman-Kac regime (:151-239) needs
grad_x / divergence / d/dt of the backbone and is the next tier: it raises.
The fused HIP sampler does not call ``f`` per step -- ``WeightedSDEIntegrator`` hands the whole
trajectory to pita_egnn_sampler_run; ``f`` serves the per-step (recording / plug-in) path.
"""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class SDETerms:  # sdes.py:34-92
    drift_X: torch.Tensor
    drift_A: torch.Tensor
    divergence_score: Optional[torch.Tensor] = None
    cross_term: Optional[torch.Tensor] = None
    dUt_dt: Optional[torch.Tensor] = None
    diffusion: Optional[torch.Tensor] = None

    _FIELDS = ("drift_X", "drift_A", "divergence_score", "cross_term", "dUt_dt", "diffusion")

    @staticmethod
    def cpu(data):
        return SDETerms(**{f: (getattr(data, f).cpu() if getattr(data, f) is not None else None)
                           for f in SDETerms._FIELDS})

    @staticmethod
    def concatenate(data_list):
        if not data_list:
            raise ValueError("The data_list is empty.")
        first = data_list[0]
        return SDETerms(**{f: (torch.cat([getattr(d, f) for d in data_list], dim=0)
                               if getattr(first, f) is not None else None) for f in SDETerms._FIELDS})


def _per_walker(t, x):
    return t * torch.ones(x.shape[0], device=x.device) if t.dim() == 0 else t


class VEReverseSDE:
    def __init__(self, noise_schedule, energy_net=None, score_net=None, cdf=None, pin_energy=False,
                 debias_inference=True):
        self.noise_schedule = noise_schedule
        self.energy_net, self.score_net = energy_net, score_net
        self.pin_energy = pin_energy
        self.debias_inference = debias_inference
        self.compiled_divergence_fn = cdf  # accepted for interface compatibility (unused: next tier)
        self.trainer = None  # set from outside by the reference (energytemp_module.py:1295)

    def g(self, t):
        return self.noise_schedule.g(t)

    def f_not_debiased(self, t, x, beta, gamma_energy):
        assert self.score_net is not None
        ht = self.noise_schedule.h(t)
        score = self.score_net(ht, x, beta)
        drift_X = gamma_energy * (score * self.g(t).pow(2).unsqueeze(-1))
        return SDETerms(drift_X=drift_X.detach(), drift_A=torch.zeros(x.shape[0], device=x.device))

    def f(self, t, x, beta, gamma_energy_schedule, gamma_score, energy_function, resampling_interval=-1):
        gamma_energy = gamma_energy_schedule.gamma(t)  # gamma_score is overwritten by it (sdes.py:142-143)
        t = _per_walker(t, x)
        if isinstance(gamma_energy, torch.Tensor):
            gamma_energy = gamma_energy.to(x.device)
        if not self.debias_inference:
            return self.f_not_debiased(t, x, beta, gamma_energy)
        raise NotImplementedError(
            "VEReverseSDE.f with debias_inference=True (Feynman-Kac weights, sdes.py:151-239) is not built on the "
            "HIP path yet; construct VEReverseSDE(..., debias_inference=False)")

    def diffusion(self, t, x, diffusion_scale):
        t = _per_walker(t, x)
        return diffusion_scale * self.g(t)[:, None] * torch.randn_like(x)
