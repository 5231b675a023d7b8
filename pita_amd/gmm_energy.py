"""40-mode 2-D Gaussian-mixture target on the HIP kernel.

Mirror of ``GMM`` (pita/src/energies/gmm_energy.py:16-90) and the part of fab's GMM it uses
(fab/fab/target_distributions/gmm.py:20-62,71-79,104): means ``(rand(K,dim)-0.5)*2*loc_scaling``
drawn right after ``torch.manual_seed(0)``, scale ``softplus(log_var_scaling)``, equal mixture
weights.  ``__call__`` returns log_prob/temperature.  ``return_force`` is an extension (the
reference GMM has none; the sampler never asks for it when is_molecule=False).
"""
import torch

from . import _lib
from .base_energy_function import BaseEnergyFunction


class GMM(BaseEnergyFunction):
    def __init__(self, dimensionality=2, n_mixes=40, loc_scaling=40, log_var_scaling=1.0, mean=None, scale=None,
                 cat_probs=None, device="cuda", should_unnormalize=False, data_normalization_factor=50,
                 temperature=1.0, **kwargs):
        if cat_probs is not None:
            raise NotImplementedError("GMM: non-uniform mixture weights are not built (reference default is equal)")
        # the reference seeds the GLOBAL generator (gmm_energy.py:38); a private generator seeded
        # with 0 draws the same means without that side effect
        gen = torch.Generator().manual_seed(0)
        if mean is None:
            mean = (torch.rand((n_mixes, dimensionality), generator=gen) - 0.5) * 2 * loc_scaling
        if scale is None:
            scale = torch.nn.functional.softplus(torch.ones((n_mixes, dimensionality)) * log_var_scaling)
        self.device = device
        self.locs = mean.to(device=device, dtype=torch.float32).contiguous()
        self.scales = scale.to(device=device, dtype=torch.float32).contiguous()
        self.n_mixes = n_mixes
        self.temperature = temperature
        self.should_unnormalize = should_unnormalize
        self.data_normalization_factor = data_normalization_factor
        self.name = "gmm"
        super().__init__(dimensionality=dimensionality, normalization_min=-data_normalization_factor,
                         normalization_max=data_normalization_factor)

    def __call__(self, samples: torch.Tensor, return_force=False):
        x = _lib.dev_tensor(samples, "samples")
        if self.should_unnormalize:
            x = self.unnormalize(x)
        x = x.reshape(-1, self._dimensionality)
        B = x.shape[0]
        logp = torch.empty(B, device=x.device, dtype=torch.float32)
        force = torch.empty_like(x) if return_force else None
        _lib.check(_lib.lib().pita_gmm_logp_force(
            x.data_ptr(), logp.data_ptr(), _lib.ptr(force), B, self._dimensionality, self.locs.data_ptr(),
            self.scales.data_ptr(), self.n_mixes, float(self.temperature), _lib.stream_ptr(x.device)),
            "pita_gmm_logp_force")
        return (logp, force) if return_force else logp
