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


def histogram(values: torch.Tensor, edges) -> torch.Tensor:
    """Counts of ``values`` (any shape, device tensor) over ``edges`` (ascending, nbins + 1) as ``numpy.histogram`` counts
    them -- half-open bins, the last one closed, out-of-range values and NaNs dropped (pita_histogram: one pass,
    LDS-privatised bins).  Returns int64 [nbins] on the device."""
    from . import _lib

    v = _lib.dev_tensor(values.reshape(-1).float().contiguous(), "values")
    e = torch.as_tensor(edges, dtype=torch.float32).to(v.device).contiguous()
    nb = e.numel() - 1
    counts = torch.empty(nb, dtype=torch.int64, device=v.device)
    _lib.check(_lib.lib().pita_histogram(v.data_ptr(), v.numel(), e.data_ptr(), nb, counts.data_ptr(), _lib.stream_ptr(v.device)),
               "pita_histogram")
    return counts


def _density(counts: torch.Tensor, edges):
    import numpy as np

    n = counts.cpu().numpy()
    db = np.array(np.diff(np.asarray(edges)), float)  # numpy.histogram(density=True): widths in the edges' own dtype first
    return n / db / n.sum()


def sample_histograms(energy_function, samples: torch.Tensor, test_set: torch.Tensor, energy_samples: torch.Tensor = None,
                      bins: int = 100):
    """The numbers under the reference's sample figure (``get_dataset_fig``, base_molecule_energy_function.py:160-254):
    two pairs of ``bins``-bin density arrays, generated samples against the test set --

    * interatomic distances: bin edges from the TEST set's distances (``hist(dist_test, bins=100, density=True)``), the
      generated distances counted on the same edges (:168-187);
    * energies ``-log p``: ``bins`` equal bins over ``(min(E_test) - 10, max(E_test) + 10)`` (:213-238).

    Bin edges are numpy's own for that range and dtype (``numpy.histogram_bin_edges``), the counting runs on the device.
    Returns a dict of numpy arrays: ``dist_edges, dist_test, dist_samples, energy_edges, energy_test, energy_samples``."""
    import numpy as np

    d_test = energy_function.interatomic_dist(test_set).reshape(-1).float()
    d_gen = energy_function.interatomic_dist(samples).reshape(-1).float()
    mm = torch.stack([d_test.min(), d_test.max()]).cpu().numpy().astype(np.float32)
    d_edges = np.histogram_bin_edges(mm, bins=bins)
    e_test = -energy_function(test_set).float()
    e_gen = -(energy_function(samples) if energy_samples is None else energy_samples).float()
    em = torch.stack([e_test.min(), e_test.max()]).cpu().numpy().astype(np.float32)
    e_edges = np.histogram_bin_edges(em, bins=bins, range=(float(em[0]) - 10, float(em[1]) + 10))
    return {"dist_edges": d_edges, "dist_test": _density(histogram(d_test, d_edges), d_edges),
            "dist_samples": _density(histogram(d_gen, d_edges), d_edges),
            "energy_edges": e_edges, "energy_test": _density(histogram(e_test, e_edges), e_edges),
            "energy_samples": _density(histogram(e_gen, e_edges), e_edges)}
