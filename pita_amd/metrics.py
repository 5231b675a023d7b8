"""Evaluation quantities the reference reports for generated samples: 1-D Wasserstein
distances between energy samples (``energy_distances``, distribution_distances.py:13-33, POT ``emd2_1d``) and between
interatomic-distance samples (energytemp_module.py:1157-1191).  Energies come from the HIP target kernels; the 1-D
optimal transport between two empirical measures is the sorted-sample coupling (device sort)."""
import math

import torch


def _w_1d(a: torch.Tensor, b: torch.Tensor, p: int):
    a, _ = torch.sort(a.double().flatten())
    b, _ = torch.sort(b.double().flatten())
    if a.numel() != b.numel():  # general sizes: merge the two quantile grids
        qa = torch.arange(1, a.numel() + 1, device=a.device, dtype=torch.float64) / a.numel()
        qb = torch.arange(1, b.numel() + 1, device=b.device, dtype=torch.float64) / b.numel()
        q = torch.unique(torch.cat([qa, qb]))
        w = torch.diff(torch.cat([q.new_zeros(1), q]))
        ia = torch.clamp(torch.searchsorted(qa, q - 1e-15), max=a.numel() - 1)
        ib = torch.clamp(torch.searchsorted(qb, q - 1e-15), max=b.numel() - 1)
        return float((w * (a[ia] - b[ib]).abs() ** p).sum())
    return float(((a - b).abs() ** p).mean())


def energy_distances(pred: torch.Tensor, true: torch.Tensor, prefix: str = "", energy_threshold: float = 1000):
    """Same dictionary as the reference's ``energy_distances`` (W2 = sqrt of the squared-euclidean 1-D OT cost,
    W1 = euclidean cost, plus the 'cropped' variants that zero out |E| > threshold entries of ``pred``)."""
    out = {f"{prefix}/energy_w2": math.sqrt(_w_1d(true, pred, 2)), f"{prefix}/energy_w1": _w_1d(true, pred, 1),
           f"{prefix}/mean_dist": float((pred.double().mean() - true.double().mean()).abs())}
    mask = (pred < -energy_threshold) | (pred > energy_threshold)
    cp, ct = (~mask) * pred, (~mask) * true if mask.shape == true.shape else true
    out[f"{prefix}/cropped_energy_w2"] = math.sqrt(_w_1d(ct, cp, 2))
    out[f"{prefix}/cropped_energy_w1"] = _w_1d(ct, cp, 1)
    out[f"{prefix}/num_cropped"] = int(mask.sum())
    return out


def interatomic_w2(energy_function, pred: torch.Tensor, true: torch.Tensor) -> float:
    """W2 between the pooled interatomic-distance samples of two walker sets."""
    return math.sqrt(_w_1d(energy_function.interatomic_dist(true), energy_function.interatomic_dist(pred), 2))
