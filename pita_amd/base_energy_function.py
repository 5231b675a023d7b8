"""Energy plug-in contract of the sampler (mirror of
pita/src/energies/base_energy_function.py:24-171 and base_molecule_energy_function.py:18-118).

A target is a callable ``energy(samples[B,D]) -> logp[B]`` returning the LOG-DENSITY (-E/T);
molecule targets also accept ``return_force=True -> (logp[B], force[B,D])``, both detached.
The sampler reads ``n_particles``, ``n_spatial_dim`` and ``is_molecule``.
"""
import os
from typing import Optional

import numpy as np
import torch

from .data_utils import remove_mean


class BaseEnergyFunction:
    def __init__(self, dimensionality: int, n_particles: Optional[int] = None, spatial_dim: Optional[int] = None,
                 is_molecule: Optional[bool] = False, normalization_min: Optional[float] = None,
                 normalization_max: Optional[float] = None):
        self._dimensionality = dimensionality
        self._is_molecule = is_molecule
        self.normalization_min, self.normalization_max = normalization_min, normalization_max
        self._test_set = self.setup_test_set()
        self._val_set = self.setup_val_set()
        self._train_set = None

    # datasets are optional plumbing (absent on a fresh box)
    def setup_test_set(self):
        return None

    def setup_val_set(self):
        return None

    def setup_train_set(self):
        return None

    @property
    def dimensionality(self):
        return self._dimensionality

    @property
    def is_molecule(self):
        return self._is_molecule

    @property
    def test_set(self):
        return self._test_set

    @property
    def val_set(self):
        return self._val_set

    @property
    def train_set(self):
        if self._train_set is None:
            self._train_set = self.setup_train_set()
        return self._train_set

    # [-1,1] box normalisation for non-molecules, scale normalisation for molecules (:60-105)
    def normalize(self, x):
        if self._is_molecule:
            return remove_mean(x, self.n_particles, self.n_spatial_dim) / self.data_normalization_factor
        lo, hi = self.normalization_min, self.normalization_max
        return ((x - lo) / (hi - lo + 1e-5)) * 2 - 1

    def unnormalize(self, x):
        if self._is_molecule:
            return x * self.data_normalization_factor
        lo, hi = self.normalization_min, self.normalization_max
        return ((x + 1) / 2) * (hi - lo) + lo

    def __call__(self, samples: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class BaseMoleculeEnergy(BaseEnergyFunction):
    """Particle-system target: carries n_particles / n_spatial_dim / temperature and the optional
    ``{train,val,test}_split_<name><n>-10000.npy`` datasets (base_molecule_energy_function.py:48-94)."""

    def __init__(self, dimensionality, n_particles, spatial_dim, data_path=None, data_name="", device="cuda",
                 is_molecule=True, temperature=1.0, should_normalize=False, data_normalization_factor=1.0):
        assert spatial_dim * n_particles == dimensionality
        self.temperature = temperature
        self.n_particles, self.n_spatial_dim = n_particles, spatial_dim
        self.device = device
        self.should_normalize = should_normalize
        self.data_normalization_factor = data_normalization_factor
        self._paths = {}
        if data_path:
            fmt = "{:0.1f}" if "LJ" in data_name else "{:0.2f}"
            d = f"{data_path}{data_name}{n_particles}_temp_{fmt.format(temperature)}/"
            for split in ("train", "val", "test"):
                self._paths[split] = d + f"{split}_split_{data_name}{n_particles}-10000.npy"
        super().__init__(dimensionality=dimensionality, is_molecule=is_molecule)

    def _load(self, split):
        path = self._paths.get(split)
        if not path or not os.path.exists(path):
            return None
        data = torch.tensor(np.load(path, allow_pickle=True), device=self.device, dtype=torch.float32)
        return self.normalize(data) if self.should_normalize else remove_mean(data, self.n_particles, self.n_spatial_dim)

    def setup_test_set(self):
        return self._load("test")

    def setup_val_set(self):
        return self._load("val")

    def setup_train_set(self):
        return self._load("train")

    def interatomic_dist(self, x):
        if self.should_normalize:
            x = self.unnormalize(x)
        from .data_utils import interatomic_dist

        return interatomic_dist(x.reshape(-1, self.n_particles, self.n_spatial_dim))
