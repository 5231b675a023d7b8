"""Synthetic molecular topologies for the force-field tests and timing tools (no oracle import here)."""
import numpy as np


def synthetic_peptide(n=22, seed=0):
    """A 22-atom branched chain with bonded tables and nonbonded parameters of realistic magnitude (the real amber14
    parameters of alanine dipeptide are outside the reference tree)."""
    rng = np.random.default_rng(seed)
    parent = [-1] + [int(rng.integers(max(0, i - 3), i)) for i in range(1, n)]
    bonds = [(parent[i], i) for i in range(1, n)]
    nbr = {i: set() for i in range(n)}
    for a, b in bonds:
        nbr[a].add(b); nbr[b].add(a)
    angles = sorted({(a, j, c) for j in range(n) for a in nbr[j] for c in nbr[j] if a < c})
    tors = sorted({(a, j, k, d) for (j, k) in bonds + [(b, a) for a, b in bonds] if j < k for a in nbr[j] - {k} for d in nbr[k] - {j} if a != d})
    # equilibrium-ish coordinates: random walk with bond length ~0.15 nm
    pos = np.zeros((n, 3))
    for i in range(1, n):
        v = rng.normal(size=3); v /= np.linalg.norm(v)
        pos[i] = pos[parent[i]] + 0.15 * v
    t = dict(bond_idx=np.array(bonds), bond_par=np.stack([rng.uniform(0.1, 0.16, len(bonds)), rng.uniform(2e5, 4e5, len(bonds))], 1),
             angle_idx=np.array(angles), angle_par=np.stack([rng.uniform(1.7, 2.2, len(angles)), rng.uniform(300, 700, len(angles))], 1),
             tors_idx=np.array(tors), tors_par=np.stack([rng.integers(1, 4, len(tors)).astype(float), rng.choice([0.0, np.pi], len(tors)), rng.uniform(0.5, 8, len(tors))], 1),
             charge=rng.normal(0, 0.35, n), sigma=rng.uniform(0.1, 0.34, n), epsilon=rng.uniform(0.05, 0.7, n))
    exc, par = [], []
    for a, b in bonds:
        exc.append((a, b)); par.append((0, 1, 0))
    for a, j, c in angles:
        exc.append((a, c)); par.append((0, 1, 0))
    seen = set(map(tuple, map(sorted, exc)))
    for a, j, k, d in tors:
        if tuple(sorted((a, d))) not in seen:
            seen.add(tuple(sorted((a, d))))
            exc.append((a, d)); par.append((t["charge"][a] * t["charge"][d] / 1.2, 0.5 * (t["sigma"][a] + t["sigma"][d]), 0.5 * np.sqrt(t["epsilon"][a] * t["epsilon"][d])))
    t["exc_idx"], t["exc_par"] = np.array(exc), np.array(par, dtype=float)
    return t, pos
