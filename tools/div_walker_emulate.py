"""fp64 emulation of the ALGEBRA of the walker-resident exact-trace kernel (pita_amd/csrc/egnn_div_walker_kernel.hip):
directions as matrix columns, the per-edge Jacobian folded into one 35 x 32 matrix per edge.  Checks the formulation (signs,
the coordinate-head rows, the dr / de terms as one small product per node) against autograd of the oracle's denoiser
before / beside the HIP implementation.  Development aid (imports the oracle: not product code).

    python tools/div_walker_emulate.py            # trained-like weights, 4 walkers
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pita_oracle as O  # noqa: E402


def sig(v):
    return 1.0 / (1.0 + np.exp(-v))


def trace_folded(P, h, x, beta, n=13, dim=3, L=3, coords_range=15.0, attention=True, tanh=True):
    """trace(J_x D) of one walker by the folded formulation.  P: state_dict as float64 numpy arrays."""
    D = n * dim
    c_s, c_in, c_out, c_noise = 1 / (1 + h), (1 + h) ** -0.5, h ** 0.5 * (1 + h) ** -0.5, np.log(h) / 8
    pos0 = (c_in * x).reshape(n, dim)
    pos = pos0.copy()
    # node features (quirk Q1)
    feat = np.concatenate([np.full(n, c_noise), np.full(n, beta)]).reshape(n, 2)
    hf = feat @ P["egnn.embedding.weight"].T + P["egnn.embedding.bias"]  # [n, 32]
    H = hf.shape[1]
    dH = np.zeros((n, H, D))          # [node][feature][direction]
    dPos = np.zeros((n, dim, D))      # [node][coord][direction]; unit directions (c_in applied at the end)
    for i in range(n):
        for k in range(dim):
            dPos[i, k, i * dim + k] = 1.0
    dPos0 = dPos.copy()
    rng = coords_range / L
    for l in range(L):
        g = f"egnn.gcl_{l}."
        W1, b1 = P[g + "edge_mlp.0.weight"], P[g + "edge_mlp.0.bias"]
        Wa, Wb, wr, we = W1[:, :H], W1[:, H:2 * H], W1[:, 2 * H], W1[:, 2 * H + 1]
        W2, b2 = P[g + "edge_mlp.2.weight"], P[g + "edge_mlp.2.bias"]
        watt, batt = (P[g + "att_mlp.0.weight"][0], P[g + "att_mlp.0.bias"][0]) if attention else (np.zeros(H), 0.0)
        Wc1, bc1, wc2 = P[g + "coord_mlp.0.weight"], P[g + "coord_mlp.0.bias"], P[g + "coord_mlp.2.weight"][0]
        Wn1, bn1 = P[g + "node_mlp.0.weight"], P[g + "node_mlp.0.bias"]
        Wn1a, Wn1b = Wn1[:, :H], Wn1[:, H:]
        Wn2, bn2 = P[g + "node_mlp.2.weight"], P[g + "node_mlp.2.bias"]
        last = l == L - 1
        Za, Zb = hf @ Wa.T + b1, hf @ Wb.T
        ZA = np.einsum("rk,nkd->nrd", Wa, dH)   # [n][32][D]
        ZB = np.einsum("rk,nkd->nrd", Wb, dH)
        new_h, new_pos, new_dH, new_dPos = hf.copy(), pos.copy(), dH.copy(), dPos.copy()
        for i in range(n):
            R = H + dim                             # extended rows: 32 message rows + dim coordinate rows
            Acc = np.zeros((R, D))
            Abar = np.zeros((R, H))
            agg = np.zeros(H)
            trans = np.zeros(dim)
            for j in range(n):
                if j == i:
                    continue
                dlt, dlt0 = pos[i] - pos[j], pos0[i] - pos0[j]
                rad, e0 = dlt @ dlt, dlt0 @ dlt0
                z1 = Za[i] + Zb[j] + wr * rad + we * e0
                s1 = sig(z1); a1 = z1 * s1; g1 = s1 * (1 + z1 * (1 - s1))
                z2 = W2 @ a1 + b2
                s2 = sig(z2); m = z2 * s2; g2 = s2 * (1 + z2 * (1 - s2))
                att = sig(watt @ m + batt) if attention else 1.0
                ms = att * m
                zc = Wc1 @ ms + bc1
                sc = sig(zc); ac = zc * sc; gc = sc * (1 + zc * (1 - sc))
                c = wc2 @ ac
                if tanh:
                    th = np.tanh(c); phi = rng * th; tau = rng * (1 - th * th)
                else:
                    phi, tau = c, 1.0
                sq = np.sqrt(rad + 1e-8); inv = 1 / (sq + 1); hsq = 0.5 / sq
                dhat = dlt * inv
                trans += dhat * phi
                agg += ms
                # adjoints
                vc = Wc1.T @ (gc * wc2)
                a = att * g2
                mp = att * (1 - att) * m if attention else np.zeros(H)
                p = g1 * (W2.T @ (watt * g2))
                q = g1 * (W2.T @ (a * vc)) + p * (mp @ vc)
                M = a[:, None] * W2 * g1[None, :] + np.outer(mp, p)      # d ms = M d z1
                Mext = np.concatenate([M, np.outer(dhat * tau, q)], 0)    # rows 32+k: Dhat_k tau q^T  (-> Dhat_k tau dc)
                # (1) partner term through the edge product
                Acc += Mext @ ZB[j]
                Abar += Mext
                # (2) dr / de / direct position terms: coefficient vectors (what the kernel's S-product carries)
                alpha = Mext @ wr           # coefficient of dr_ij[d]
                eps = Mext @ we             # coefficient of de_ij[d]
                alpha[H:] -= phi * inv * dhat * hsq   # dDhat_k phi: -phi inv Dhat_k hsq dr
                dD = dPos[i] - dPos[j]      # [dim][D]
                dr = 2 * dlt @ dD           # [D]
                de = 2 * dlt0 @ (dPos0[i] - dPos0[j])
                Acc += np.outer(alpha, dr) + np.outer(eps, de)
                Acc[H:] += phi * inv * dD   # dDhat_k phi: + phi inv dDelta_k
            Acc += Abar @ ZA[i]
            new_pos[i] = pos[i] + trans
            new_dPos[i] = dPos[i] + Acc[H:]
            if not last:
                zn = Wn1a @ hf[i] + Wn1b @ agg + bn1
                sn = sig(zn); an = zn * sn; gn = sn * (1 + zn * (1 - sn))
                new_h[i] = hf[i] + Wn2 @ an + bn2
                dzn = Wn1a @ dH[i] + Wn1b @ Acc[:H]
                new_dH[i] = dH[i] + Wn2 @ (gn[:, None] * dzn)
        hf, pos, dH, dPos = new_h, new_pos, new_dH, new_dPos
    tr = 0.0
    for i in range(n):
        for k in range(dim):
            d = i * dim + k
            tr += c_s + c_out * c_in * (dPos[i, k, d] - 1.0)   # mean-free shares cancel over a full trace
    return tr


def main():
    torch.manual_seed(0)
    w = dict(np.load(os.path.join(ROOT, "tests", "golden", "egnn_weights_trainedlike.npz")))
    wt = {k: torch.tensor(v).double() for k, v in w.items()}
    P = {k: v.numpy() for k, v in wt.items()}
    B, n, dim = 4, 13, 3
    x = torch.randn(B, n * dim, dtype=torch.float64) * 1.5
    x = O.remove_mean(x, n, dim)
    h = torch.tensor([0.05, 0.7, 4.0, 60.0], dtype=torch.float64)
    beta = 1.0
    bb = lambda cn, xs, b: O.egnn_forward(wt, cn, xs, b, n, dim)
    for b in range(B):
        f = lambda xv: O.denoiser(bb, h[b:b + 1], xv[None], beta)[0]
        J = torch.autograd.functional.jacobian(f, x[b])
        ref = float(torch.trace(J))
        got = trace_folded(P, float(h[b]), x[b].numpy(), beta)
        print(f"walker {b}: h = {float(h[b]):6.2f}  autograd trace {ref:+.12e}  folded {got:+.12e}  rel {abs(got - ref) / abs(ref):.2e}")


if __name__ == "__main__":
    main()
