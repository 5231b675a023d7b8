"""remove_mean on the GPU (mirror of pita/src/utils/data_utils.py:4-26)."""
import torch

from . import _lib


def remove_mean(samples: torch.Tensor, n_particles: int, n_dimensions: int) -> torch.Tensor:
    """Per-walker subtraction of the particle mean; returns a new tensor of the same shape."""
    shape = samples.shape
    x = _lib.dev_tensor(samples, "samples").reshape(-1, n_particles * n_dimensions).clone()
    _lib.check(_lib.lib().pita_remove_mean(x.data_ptr(), x.shape[0], n_particles, n_dimensions,
                                           _lib.stream_ptr(x.device)), "pita_remove_mean")
    return x.reshape(shape)


def interatomic_dist(samples: torch.Tensor) -> torch.Tensor:
    """Upper-triangle pair distances [B, n(n-1)/2] (data_utils.py:29-37); evaluation helper, torch ops."""
    n = samples.shape[-2]
    iu = torch.triu_indices(n, n, offset=1, device=samples.device)
    return torch.linalg.norm(samples[:, iu[1]] - samples[:, iu[0]], dim=-1)
