"""Helpers for tests/golden/em_traj_lj13_debias_long.npz, shared by the CPU oracle test and the GPU test (no oracle or
product import here)."""
import numpy as np


def long_fixture_draws(g):
    """Regenerate the PCG64 streams the fixture was generated on (tests/golden/make_golden.py:gen_traj_debias_long
    stores only their seed): SDE normals [N, B, 39], MALA normals [n_mala, B, 39], MALA uniforms [n_mala, B] (fp32),
    resampling uniforms [end + 1] (fp64)."""
    seed, N, B, n_mala, end = (int(g[k]) for k in ("seed", "N", "B", "n_mala", "end"))
    noise = np.random.Generator(np.random.PCG64(seed)).standard_normal((N, B, 39), dtype=np.float32)
    mala_noise = np.random.Generator(np.random.PCG64(seed + 2)).standard_normal((n_mala, B, 39), dtype=np.float32)
    mala_u = np.random.Generator(np.random.PCG64(seed + 3)).random((n_mala, B), dtype=np.float32)
    us = np.random.Generator(np.random.PCG64(seed + 4)).random(end + 1)
    return noise, mala_noise, mala_u, us


def ids_mismatch_is_bin_edge_tie(logits, u0, ids_got, ids_want, rel_tol=2e-6):
    """True when every position where two systematic-resampling id vectors differ is a +-1 neighbour whose uniform
    sits within fp32 rounding of the bin edge between the two parents (utils.py:111-120: bins are an fp32 cumsum of
    clip(softmax(logits), 1e-6, 1); a last-bit difference in logits moves an edge by ~1e-7 of the running sum).
    ``logits``: float array [B] (either side's), ``u0``: the event's float64 uniform."""
    logits = np.asarray(logits, dtype=np.float64)
    B = logits.shape[0]
    w = np.exp(logits - logits.max())
    w = np.clip(w / w.sum(), 1e-6, 1.0)
    bins = np.cumsum(w)
    u = (u0 + np.arange(B) / B) % 1.0
    bad = np.nonzero(np.asarray(ids_got) != np.asarray(ids_want))[0]
    for i in bad:
        a, b = int(ids_got[i]), int(ids_want[i])
        if abs(a - b) != 1:
            return False
        edge = bins[min(a, b)]
        if abs(u[i] - edge) > rel_tol * max(edge, 1e-3) + 1e-6 * w[max(a, b)] + 1e-6 * w[min(a, b)]:
            return False
    return True
