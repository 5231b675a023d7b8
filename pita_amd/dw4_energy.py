"""DW4-style multi double-well target (BASELINE config C2) on the HIP pairwise kernel.

The reference tree has no DW4 energy (only the dead import
pita/src/energies/base_datamodule.py:13 of ``bgflow.MultiDoubleWellPotential``); this class gives
it the same plug-in shape as LennardJonesEnergy.  E = sum_{i<j} a (d-d0)^4 + b (d-d0)^2 + c,
defaults a=0.9, b=-4, c=0, d0=4 (the DEM/bgflow DW4 parameters).  Parity is pinned only
against the oracle's own restatement.
"""
import torch

from . import _lib
from .base_energy_function import BaseMoleculeEnergy


class MultiDoubleWellEnergy(BaseMoleculeEnergy):
    def __init__(self, dimensionality=8, n_particles=4, spatial_dim=2, data_path=None, device="cuda", a=0.9, b=-4.0,
                 c=0.0, offset=4.0, is_molecule=True, temperature=1.0, should_normalize=False,
                 data_normalization_factor=1.0, *args, **kwargs):
        self.name = f"DW{n_particles}"
        super().__init__(dimensionality=dimensionality, n_particles=n_particles, spatial_dim=spatial_dim,
                         data_path=data_path, data_name="DW", device=device, is_molecule=is_molecule,
                         temperature=temperature, should_normalize=should_normalize,
                         data_normalization_factor=data_normalization_factor)
        self.a, self.b, self.c, self.offset = float(a), float(b), float(c), float(offset)

    def __call__(self, samples: torch.Tensor, return_force=False):
        x = _lib.dev_tensor(samples, "samples")
        if self.should_normalize:
            x = self.unnormalize(x)
        x = x.reshape(-1, self._dimensionality)
        B = x.shape[0]
        logp = torch.empty(B, device=x.device, dtype=torch.float32)
        force = torch.empty_like(x) if return_force else None
        _lib.check(_lib.lib().pita_dw_logp_force(
            x.data_ptr(), logp.data_ptr(), _lib.ptr(force), B, self.n_particles, self.n_spatial_dim,
            float(self.temperature), self.a, self.b, self.c, self.offset, _lib.stream_ptr(x.device)),
            "pita_dw_logp_force")
        return (logp, force) if return_force else logp

    def fused_descent(self, x, num_steps, dt, noise_scale, sqrt_dt, seed=0, walker_offset=0, step0=0, remove_mean=True,
                      noise=None):
        """See LennardJonesEnergy.fused_descent."""
        if self.should_normalize:
            return None
        _lib.check(_lib.lib().pita_dw_descent(
            x.data_ptr(), _lib.ptr(noise), x.shape[0], self.n_particles, self.n_spatial_dim, float(self.temperature),
            self.a, self.b, self.c, self.offset, int(num_steps), float(dt), float(noise_scale), float(sqrt_dt), seed,
            walker_offset, step0, int(remove_mean), _lib.stream_ptr(x.device)), "pita_dw_descent")
        return x

    def fused_mala(self, x, logp, num_steps, dt_dev, adaptive, total, noise=None, uniforms=None, seed=0, walker_offset=0,
                   walker_ids=None, step0=0, remove_mean=True, rates_out=None):
        """See LennardJonesEnergy.fused_mala (pita_dw_mala: the DW4 ring kernel)."""
        if self.should_normalize or self.n_particles != 4 or self.n_spatial_dim != 2:
            return None
        L = _lib.lib()
        ws = torch.empty((int(L.pita_lj_mala_workspace_bytes(int(num_steps))) + 7) // 8, device=x.device, dtype=torch.int64)
        rc = L.pita_dw_mala(x.data_ptr(), logp.data_ptr(), _lib.ptr(noise), _lib.ptr(uniforms), x.shape[0], 4, 2,
                            float(self.temperature), self.a, self.b, self.c, self.offset, int(num_steps),
                            dt_dev.data_ptr(), int(bool(adaptive)), int(total), int(seed) & 0xFFFFFFFFFFFFFFFF,
                            int(walker_offset), _lib.ptr(walker_ids), int(step0), int(bool(remove_mean)),
                            _lib.ptr(rates_out), ws.data_ptr(), _lib.stream_ptr(x.device))
        if rc == -2:  # PITA_EUNSUPPORTED
            return None
        _lib.check(rc, "pita_dw_mala")
        return x
