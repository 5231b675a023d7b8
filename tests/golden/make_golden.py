#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/*.npz by IMPORTING THE
REFERENCE (read-only, /root/reference) in the build container, CPU, float32.

Run:  python tests/golden/make_golden.py          (needs /root/reference; never runs on the GPU box)

Each .npz holds seeded inputs plus the outputs the reference's own code produced for
them.  The fixtures are data only; no reference source is stored.  The oracle
(oracle/pita_oracle.py) and the HIP path are both checked against them.
"""
import os
import sys
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()

from src.energies import base_prior, gmm_energy, lennardjones_energy  # noqa: E402
from src.models.components import (annealing_factor_schedules, egnn, egnn_temp_conditioned, energy_net,  # noqa: E402
                                   mlp, noise_schedules, score_net, sde_integration, sdes, utils)
from src.utils import data_utils  # noqa: E402

torch.set_num_threads(8)
F32 = np.float32


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB  ({len(arrs)} arrays)")


def sd_np(module):
    return {k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


class LJ(lennardjones_energy.LennardJonesEnergy):
    """The reference class with only its dataset loading (absent .npy files) disabled."""

    def setup_test_set(self):
        return None

    def setup_val_set(self):
        return None


def lattice_cluster(n, d, B, gen, spacing=1.08, jitter=0.07):
    """Compact, physically reasonable n-particle clusters (min distance ~0.8)."""
    r = 3
    grid = np.stack(np.meshgrid(*[np.arange(-r, r + 1)] * d, indexing="ij"), -1).reshape(-1, d).astype(np.float64)
    # fcc-like offset for 3D to get ~12 neighbours
    order = np.argsort((grid**2).sum(-1), kind="stable")
    pts = grid[order[:n]] * spacing
    x = torch.tensor(pts, dtype=torch.float32)[None].repeat(B, 1, 1)
    x = x + jitter * torch.randn(B, n, d, generator=gen)
    x = x - x.mean(dim=1, keepdim=True)
    return x.reshape(B, n * d)


# ----------------------------------------------------------------------------- schedules
def gen_schedules():
    t = torch.linspace(0, 1, 1001)
    out = {"t": t.numpy()}
    for smin in (0.002, 0.01, 0.05):
        s = noise_schedules.ElucidatingNoiseSchedule(sigma_min=smin, sigma_max=80.0, rho=7)
        h = s.h(t)
        out[f"h_{smin}"] = h.numpy()
        out[f"g_{smin}"] = s.g(t).numpy()
        out[f"dhdt_{smin}"] = s.dh_dt(t).numpy()
        out[f"tinv_{smin}"] = s.t(h).numpy()
    geo = noise_schedules.GeometricNoiseSchedule(0.01, 10.0)
    out["geo_h"] = geo.h(t).numpy()
    out["geo_g"] = geo.g(t).numpy()
    c = annealing_factor_schedules.ConstantAnnealingFactorSchedule(4 / 3)
    l = annealing_factor_schedules.LinearAnnealingFactorSchedule(1.5, 1.0, t_start=0.9, t_end=0.1)
    s = annealing_factor_schedules.SigmoidAnnealingFactorSchedule(1.5, 1.0, t_start=0.9, t_end=0.1, sharpness=10.0)
    tt = torch.linspace(-0.1, 1.1, 241)
    out["tt"] = tt.numpy()
    for nm, sch in (("const", c), ("lin", l), ("sig", s)):
        out[f"gamma_{nm}"] = sch.gamma(tt).numpy()
        out[f"dgamma_{nm}"] = sch.dgamma_dt(tt).numpy()
    save("schedules.npz", **out)


# ----------------------------------------------------------------------------- LJ target
def gen_lj():
    for n in (13, 55):
        gen = torch.Generator().manual_seed(1000 + n)
        D = 3 * n
        cold = lattice_cluster(n, 3, 48, gen)
        warm = lattice_cluster(n, 3, 8, gen, spacing=1.3, jitter=0.25)
        hot = torch.randn(8, D, generator=gen) * 1.0  # adversarial: overlapping particles, |E| ~ 1e6+
        hot = data_utils.remove_mean(hot, n, 3)
        x = torch.cat([cold, warm, hot])
        out = {"x": x.numpy(), "n_cold": 48, "n_warm": 8}
        for T in (1.0, 2.0, 4.0):
            e = LJ(D, n, 3, data_path="", temperature=T)
            lp = e(x.clone())
            lp2, f = e(x.clone(), return_force=True)
            assert torch.equal(lp, lp2)
            out[f"logp_T{T}"] = lp.numpy()
            out[f"force_T{T}"] = f.numpy()
        e = LJ(D, n, 3, data_path="", temperature=1.0, energy_factor=0.5)
        lp, f = e(x.clone(), return_force=True)
        out["logp_ef0.5"] = lp.numpy()
        out["force_ef0.5"] = f.numpy()
        if n == 13:
            # the reference's OWN in-tree LJ13 log-density (sampling/sample_lj13.py:15-30: torch.pdist, no eps, "2 x"
            # unordered pairs + oscillator), executed as shipped -- no bgflow restatement involved.  fp32 and fp64.
            e2 = _load_sample_lj13().energy2
            for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
                xe = x.clone().to(dt).requires_grad_(True)
                lp2 = e2(xe)
                (f2,) = torch.autograd.grad(lp2.sum(), xe)
                out[f"energy2_logp_{tag}"] = lp2.detach().numpy()
                out[f"energy2_force_{tag}"] = f2.numpy()
        save(f"lj{n}_logp_force.npz", **out)


def _load_sample_lj13():
    """Import /root/reference/sampling/sample_lj13.py as shipped (module level only defines functions; the MCMC driver
    sits under ``__main__``).  Its import-time dependency pyro is absent here: placeholder modules, no arithmetic."""
    import importlib.util

    _ref_shims._mod("pyro", sample=_ref_shims._Anything(), factor=_ref_shims._Anything())
    _ref_shims._mod("pyro.distributions", Uniform=_ref_shims._Anything)
    _ref_shims._mod("pyro.infer", MCMC=_ref_shims._Anything, NUTS=_ref_shims._Anything)
    _ref_shims._mod("pyro.infer.mcmc")
    _ref_shims._mod("pyro.infer.mcmc.rwkernel", RandomWalkKernel=object)
    _ref_shims._mod("pyro.ops")
    _ref_shims._mod("pyro.ops.integrator", potential_grad=_ref_shims._Anything())
    spec = importlib.util.spec_from_file_location("ref_sample_lj13", _ref_shims.REF_ROOT + "/sampling/sample_lj13.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def gen_lj_smooth():
    """LennardJonesEnergy(smooth=True): the spline core below r = 0.65 (lennardjones_energy.py:114-119,131-133)."""
    n, D = 13, 39
    gen = torch.Generator().manual_seed(4013)
    cold = lattice_cluster(n, 3, 24, gen)
    squeezed = lattice_cluster(n, 3, 24, gen, spacing=0.62, jitter=0.06)  # most neighbour pairs inside the core
    hot = data_utils.remove_mean(torch.randn(16, D, generator=gen) * 0.7, n, 3)
    x = torch.cat([cold, squeezed, hot])
    out = {"x": x.numpy()}
    for T, ef in ((1.0, 1.0), (2.0, 0.5)):
        e = LJ(D, n, 3, data_path="", temperature=T, energy_factor=ef, smooth=True)
        lp = e(x.clone())
        lp2, f = e(x.clone(), return_force=True)
        assert torch.equal(lp, lp2)
        out[f"logp_T{T}_ef{ef}"] = lp.numpy()
        out[f"force_T{T}_ef{ef}"] = f.numpy()
    pot = e.lennard_jones
    out["spline_c0"] = pot.splines.keywords["c"][:, 0].numpy()
    out["spline_x0"] = pot.splines.keywords["x"][:1].numpy()
    v = x.reshape(-1, n, 3)
    dmin = torch.cdist(v, v).add(torch.eye(n) * 1e3).amin(dim=(1, 2))
    out["n_core_walkers"] = int((dmin < 0.65).sum())
    save("lj13_smooth_logp_force.npz", **out)


# ----------------------------------------------------------------------------- GMM target
def gen_gmm():
    out = {}
    for T in (1.0, 2.0):
        g = gmm_energy.GMM(temperature=T)
        gen = torch.Generator().manual_seed(7)
        x = (torch.rand(256, 2, generator=gen) - 0.5) * 112
        if T == 1.0:
            out["x"] = x.numpy()
            out["means"] = g.gmm.locs.numpy()
            out["scale_trils"] = g.gmm.scale_trils.numpy()
            out["cat_probs"] = g.gmm.cat_probs.numpy()
            xs = x.clone().requires_grad_(True)
            (grad,) = torch.autograd.grad(g(xs).sum(), xs)
            out["grad_T1.0"] = grad.numpy()
        out[f"logp_T{T}"] = g(x).numpy()
    save("gmm40.npz", **out)


# ----------------------------------------------------------------------------- EGNN
def make_egnn(n, d, temp=True):
    torch.manual_seed(12345)
    if temp:
        return egnn_temp_conditioned.EGNN_dynamics(
            n_particles=n, n_dimension=d, hidden_nf=32, n_layers=3, act_fn=torch.nn.SiLU(), recurrent=True,
            tanh=True, attention=True, condition_time=True, condition_temperature=True, agg="sum")
    return egnn.EGNN_dynamics(n_particles=n, n_dimension=d, hidden_nf=32, n_layers=3, act_fn=torch.nn.SiLU(),
                              recurrent=True, tanh=True, attention=True, condition_time=True, agg="sum")


def gen_egnn():
    net13 = make_egnn(13, 3)
    w = sd_np(net13)
    save("egnn_weights_seed12345.npz", **w)
    # a second, "trained-like" weight set: the xavier(gain=1e-3) coord head of the fresh
    # init makes velocities ~1e-4; scale those heads up so every term matters numerically.
    torch.manual_seed(777)
    net_t = make_egnn(13, 3)
    with torch.no_grad():
        for l in range(3):
            getattr(net_t.egnn, f"gcl_{l}").coord_mlp[2].weight.mul_(300.0)
        for p in net_t.parameters():
            p.add_(0.02 * torch.randn_like(p))
    wt = sd_np(net_t)
    save("egnn_weights_trainedlike.npz", **wt)

    for name, n, d in (("lj13", 13, 3), ("dw4", 4, 2), ("lj55", 55, 3)):
        B = 24 if n < 55 else 8
        gen = torch.Generator().manual_seed(4242 + n)
        base = lattice_cluster(n, d, B, gen, spacing=1.1 if d == 3 else 2.5, jitter=0.1)
        hs = torch.tensor([1e-3, 0.0025, 0.1, 1.0, 10.0, 400.0, 6400.0, 3.0])[torch.arange(B) % 8]
        x = base + hs.sqrt()[:, None] * torch.randn(B, n * d, generator=gen)
        betas = torch.tensor([1.0, 1.33, 4.0])[torch.arange(B) % 3]
        out = {"x": x.numpy(), "h": hs.numpy(), "beta": betas.numpy(), "n": n, "d": d}
        for tag, state in (("init", w), ("trained", wt)):
            net = make_egnn(n, d)
            net.load_state_dict({k: torch.tensor(v) for k, v in state.items()})
            sn = score_net.ScoreNet(net)
            with torch.no_grad():
                c_noise = (1 / 8) * torch.log(hs)
                c_in = 1 / (1 + hs) ** 0.5
                F = net(c_noise, c_in[:, None] * x, betas)
                Dth = sn.denoiser(hs, x, betas)
                sc = sn(hs, x, betas)
            out[f"F_{tag}"] = F.numpy()
            out[f"D_{tag}"] = Dth.numpy()
            out[f"score_{tag}"] = sc.numpy()
            # energy-net view of the same backbone (energy_net.py:14-49)
            en = energy_net.EnergyNet(net)
            with torch.no_grad():
                out[f"E_{tag}"] = en.forward_energy(hs, x, betas).numpy()
        # pin the t/beta interleave quirk: recompute the h0 matrix the way the reference does
        t_ = c_noise.unsqueeze(-1)
        b_ = betas.unsqueeze(-1)
        h0 = torch.cat([torch.ones(B, n) * t_, torch.ones(B, n) * b_], dim=-1).reshape(B * n, 2)
        out["h0"] = h0.numpy()
        out["c_noise"] = c_noise.numpy()
        save(f"egnn_{name}_fwd.npz", **out)

    # non-temperature-conditioned egnn.py (in_node_nf=1)
    net1 = make_egnn(13, 3, temp=False)
    gen = torch.Generator().manual_seed(99)
    x = lattice_cluster(13, 3, 8, gen) + 0.3 * torch.randn(8, 39, generator=gen)
    t = torch.linspace(-0.8, 1.0, 8)
    with torch.no_grad():
        y = net1(t, x)
    save("egnn_notemp_lj13_fwd.npz", x=x.numpy(), t=t.numpy(), out=y.numpy(), **{"w." + k: v for k, v in sd_np(net1).items()})


def gen_egnn_ad2cat():
    """EGNN_dynamics_AD2_cat (egnn_dynamics_ad2_cat.py; configs/model/net/egnn_dynamics_ad2_cat.yaml: hidden 64 x 5 layers,
    condition_beta) for 22 atoms: the reference module's output on seeded inputs, its weights and its static node
    features.  mdtraj is only touched for >= 53 particles: a placeholder module satisfies the import."""
    _ref_shims._mod("mdtraj")
    from src.models.components import egnn_dynamics_ad2_cat as ad2

    for tag, kw in (("h64", dict(hidden_nf=64, n_layers=5)), ("h48", dict(hidden_nf=48, n_layers=2, attention=False, tanh=False))):
        torch.manual_seed(2468)
        net = ad2.EGNN_dynamics_AD2_cat(n_particles=22, n_dimensions=3, condition_beta=True, **kw)
        with torch.no_grad():  # the fresh coordinate heads (xavier gain 1e-3) make velocities ~1e-4: scale them up
            for l in range(kw["n_layers"]):
                getattr(net.egnn, f"gcl_{l}").coord_mlp[2].weight.mul_(300.0)
            for p in net.parameters():
                p.add_(0.02 * torch.randn_like(p))
        gen = torch.Generator().manual_seed(97)
        B = 12
        base = lattice_cluster(22, 3, B, gen, spacing=1.1, jitter=0.1)
        hs = torch.tensor([1e-3, 0.1, 1.0, 10.0, 400.0, 3.0])[torch.arange(B) % 6]
        x = base + hs.sqrt()[:, None] * torch.randn(B, 66, generator=gen)
        betas = torch.tensor([1.0, 1.33, 4.0])[torch.arange(B) % 3]
        sn = score_net.ScoreNet(net)
        with torch.no_grad():
            c_noise = (1 / 8) * torch.log(hs)
            c_in = 1 / (1 + hs) ** 0.5
            F = net(c_noise, c_in[:, None] * x, betas)
            Dth = sn.denoiser(hs, x, betas)
            sc = sn(hs, x, betas)
        save(f"egnn_ad2cat_{tag}_fwd.npz", x=x.numpy(), h=hs.numpy(), beta=betas.numpy(), F=F.numpy(), D=Dth.numpy(),
             score=sc.numpy(), h_initial=net.h_initial.numpy().astype(np.float32),
             **{"w." + k: v for k, v in sd_np(net).items()})


def gen_egnn_ad2cat_sizes():
    """EGNN_dynamics_AD2_cat for the other particle counts its get_h_initial knows without a topology file (33 and 42
    atoms: tri- / tetra-alanine; 13 and 55: the LJ systems with a zero feature): the reference module's static node
    features and, for a two-layer hidden-32 net, its backbone output on seeded inputs."""
    _ref_shims._mod("mdtraj")
    from src.models.components import egnn_dynamics_ad2_cat as ad2

    out = {}
    for n in (13, 33, 42, 55):
        torch.manual_seed(1000 + n)
        net = ad2.EGNN_dynamics_AD2_cat(n_particles=n, n_dimensions=3, hidden_nf=32, n_layers=2, condition_beta=True)
        out[f"h_initial_{n}"] = net.h_initial.numpy().astype(np.float32)
        if n not in (33, 42):
            continue
        with torch.no_grad():
            for l in range(2):
                getattr(net.egnn, f"gcl_{l}").coord_mlp[2].weight.mul_(300.0)
            for p in net.parameters():
                p.add_(0.02 * torch.randn_like(p))
        gen = torch.Generator().manual_seed(n)
        B = 6
        x = lattice_cluster(n, 3, B, gen, spacing=1.1, jitter=0.1) + 0.3 * torch.randn(B, n * 3, generator=gen)
        t = torch.rand(B, generator=gen) - 0.5
        betas = torch.tensor([1.0, 1.33, 4.0])[torch.arange(B) % 3]
        with torch.no_grad():
            F = net(t, x, betas)
        out.update({f"x_{n}": x.numpy(), f"t_{n}": t.numpy(), f"beta_{n}": betas.numpy(), f"F_{n}": F.numpy()})
        out.update({f"w{n}." + k: v for k, v in sd_np(net).items()})
    save("egnn_ad2cat_sizes.npz", **out)


def gen_egnn_aldp():
    """``egnn_aldp.EGNN_dynamics`` (egnn_aldp.py:8-197: the other peptide EGNN of the reference -- hidden 64 x 4 layers,
    no attention gate, no tanh bound by default, one-hot atom types with ITS OWN methyl grouping for 22 atoms, t and beta
    as two more node features): the reference module's output on seeded inputs for 22 atoms with the class defaults +
    temperature conditioning, and for a 33-atom net with attention / tanh switched on; weights and static features."""
    from src.models.components import egnn_aldp

    out = {}
    for tag, n, kw in (("n22", 22, dict(condition_temperature=True)),
                       ("n33", 33, dict(condition_temperature=True, attention=True, tanh=True, n_layers=2, hidden_nf=32))):
        torch.manual_seed(1357 + n)
        net = egnn_aldp.EGNN_dynamics(n_particles=n, n_dimension=3, **kw)
        L = kw.get("n_layers", 4)
        with torch.no_grad():
            for l in range(L):
                getattr(net.egnn, f"gcl_{l}").coord_mlp[2].weight.mul_(300.0)
            for p in net.parameters():
                p.add_(0.02 * torch.randn_like(p))
        gen = torch.Generator().manual_seed(n)
        B = 8
        x = lattice_cluster(n, 3, B, gen, spacing=1.1, jitter=0.1) + 0.3 * torch.randn(B, n * 3, generator=gen)
        t = torch.rand(B, generator=gen) - 0.5
        betas = torch.tensor([1.0, 1.33, 4.0])[torch.arange(B) % 3]
        with torch.no_grad():
            F = net(t, x, betas)
        out.update({f"x_{tag}": x.numpy(), f"t_{tag}": t.numpy(), f"beta_{tag}": betas.numpy(), f"F_{tag}": F.numpy(),
                    f"h_initial_{tag}": net.h_initial.numpy().astype(np.float32)})
        out.update({f"w_{tag}." + k: v for k, v in sd_np(net).items()})
    save("egnn_aldp_fwd.npz", **out)


# ----------------------------------------------------------------------------- MLP
def gen_mlp():
    torch.manual_seed(12345)
    net = mlp.MyMLP(hidden_size=128, hidden_layers=3, emb_size=128, out_dim=2, input_dim=2)
    gen = torch.Generator().manual_seed(5)
    B = 64
    hs = torch.exp(torch.rand(B, generator=gen) * (np.log(6400.0) - np.log(1e-4)) + np.log(1e-4))
    x = (torch.rand(B, 2, generator=gen) - 0.5) * 100 + hs.sqrt()[:, None] * torch.randn(B, 2, generator=gen)
    beta = torch.ones(B)
    sn = score_net.ScoreNet(net)
    with torch.no_grad():
        c_noise = (1 / 8) * torch.log(hs)
        c_in = 1 / (1 + hs) ** 0.5
        F = net(c_noise, c_in[:, None] * x, beta)
        sc = sn(hs, x, beta)
    w = {"w." + k: v for k, v in sd_np(net).items()}
    save("mlp_gmm_fwd.npz", x=x.numpy(), h=hs.numpy(), F=F.numpy(), score=sc.numpy(), **w)

    torch.manual_seed(54321)
    # NB the reference sizes the last Linear by emb_size (mlp.py:489-490), so hidden_size must equal emb_size
    net2 = mlp.MyMLPTemperature(hidden_size=64, hidden_layers=2, emb_size=64, out_dim=3, input_dim=3)
    x3 = torch.randn(32, 3, generator=gen) * 3
    t3 = torch.randn(32, generator=gen)
    b3 = torch.rand(32, generator=gen) * 3 + 0.5
    with torch.no_grad():
        y = net2(t3, x3, b3)
    save("mlp_temp_fwd.npz", x=x3.numpy(), t=t3.numpy(), beta=b3.numpy(), out=y.numpy(),
         **{"w." + k: v for k, v in sd_np(net2).items()})


# ----------------------------------------------------------------------------- prior / remove_mean
def gen_prior():
    out = {}
    for n, d in ((13, 3), (4, 2)):
        torch.manual_seed(31 + n)
        raw_state = torch.get_rng_state()
        noise = torch.randn(16, n * d)
        torch.set_rng_state(raw_state)
        scale = 69.28203
        p = base_prior.Prior(scale=scale, n_particles=n, spatial_dim=d)
        s = p.sample(16)
        out[f"noise_{n}"] = noise.numpy()
        out[f"sample_{n}"] = s.numpy()
        out[f"logprob_{n}"] = p.log_prob(s).numpy()
        out["scale"] = scale
        out[f"remove_mean_{n}"] = data_utils.remove_mean(noise, n, d).numpy()
    save("prior.npz", **out)


# ----------------------------------------------------------------------------- resampling
def gen_resample():
    out = {}
    gen = torch.Generator().manual_seed(11)
    cases = {
        "normal": torch.randn(257, generator=gen) * 3,
        "ties": torch.zeros(64),
        "peaked": torch.cat([torch.full((99,), -50.0), torch.tensor([10.0])]),
        "neginf": torch.cat([torch.randn(30, generator=gen), torch.full((2,), -float("inf"))]),
        "huge": torch.randn(1000, generator=gen) * 200,
        "big": torch.randn(5000, generator=gen) * 2,
    }
    real_rand = torch.rand
    for k, logits in cases.items():
        us = []

        def rec_rand(*a, **kw):
            r = real_rand(*a, **kw)
            us.append(r.clone())
            return r

        torch.manual_seed(1234)
        torch.rand = rec_rand
        try:
            ids, _ = utils.sample_cat_sys(logits.shape[0], logits)
        finally:
            torch.rand = real_rand
        out[f"logits_{k}"] = logits.numpy()
        out[f"u_{k}"] = us[0].numpy()
        out[f"ids_{k}"] = np.asarray(ids, dtype=np.int64)
        q = torch.quantile(logits[torch.isfinite(logits)], 0.9)
        out[f"q90_{k}"] = q.numpy()
    save("resample_sys.npz", **out)


# ----------------------------------------------------------------------------- trajectories
class FakeTrainer:
    world_size = 1
    global_rank = 0
    num_nodes = 1


class FakeLM:
    """Single-rank stand-in for the LightningModule the integrator talks to
    (sde_integration.py:227-229,248-251): all_gather adds the leading world dim."""

    trainer = FakeTrainer()

    def all_gather(self, obj):
        if isinstance(obj, dict):
            return {k: self.all_gather(v) for k, v in obj.items()}
        if obj is None:
            return None
        return obj.unsqueeze(0)


class Recorder:
    """Record every randn_like / rand / rand_like draw made by the reference."""

    def __init__(self):
        self.randn, self.rand, self.rand_like = [], [], []
        self._r = (torch.randn_like, torch.rand, torch.rand_like)

    def __enter__(self):
        rn, r, rl = self._r

        def f_randn_like(x, *a, **k):
            v = rn(x, *a, **k)
            self.randn.append(v.detach().clone())
            return v

        def f_rand(*a, **k):
            v = r(*a, **k)
            self.rand.append(v.detach().clone())
            return v

        def f_rand_like(x, *a, **k):
            v = rl(x, *a, **k)
            self.rand_like.append(v.detach().clone())
            return v

        torch.randn_like, torch.rand, torch.rand_like = f_randn_like, f_rand, f_rand_like
        return self

    def __exit__(self, *exc):
        torch.randn_like, torch.rand, torch.rand_like = self._r


def build_lj13_stack(weights, debias, sigma_min=0.05):
    net = make_egnn(13, 3)
    net.load_state_dict({k: torch.tensor(v) for k, v in weights.items()})
    import copy

    sn = score_net.ScoreNet(net)
    en = energy_net.EnergyNet(copy.deepcopy(net))
    sched = noise_schedules.ElucidatingNoiseSchedule(sigma_min=sigma_min, sigma_max=80.0, rho=7)
    sde = sdes.VEReverseSDE(noise_schedule=sched, energy_net=en, score_net=sn,
                            cdf=partial(utils.compute_divergence_exact, sn.forward), pin_energy=False,
                            debias_inference=debias)
    sde.trainer = FakeTrainer()
    return sde, sched


def gen_traj_nodebias():
    wt = dict(np.load(os.path.join(HERE, "egnn_weights_trainedlike.npz")))
    sde, sched = build_lj13_stack(wt, debias=False)
    N, B = 20, 32
    gamma = annealing_factor_schedules.ConstantAnnealingFactorSchedule(4 / 3)
    integ = sde_integration.WeightedSDEIntegrator(
        sde=sde, num_integration_steps=N, start_resampling_step=0, end_resampling_step=N, lightning_module=FakeLM(),
        partial_annealing_factor_schedule=None, resampling_interval=-1, num_negative_time_steps=0, post_mcmc_steps=0,
        batch_size=16, should_mean_free=True)
    e = LJ(39, 13, 3, data_path="", temperature=1.0)
    torch.manual_seed(2024)
    scale = float((sched.h(torch.tensor(1.0)) / gamma.gamma(torch.tensor(1.0))) ** 0.5)
    x1 = base_prior.Prior(scale=scale, n_particles=13, spatial_dim=3).sample(B)
    xs = []
    real_rm = data_utils.remove_mean
    with Recorder() as rec:
        x, logw, uniq, terms, acc = integ.integrate_sde(x1.clone(), e, gamma, inverse_temperature=1.0)
    drift = np.stack([t.drift_X.reshape(B, 39).numpy() for t in terms])
    diffusion = np.stack([t.diffusion.reshape(B, 39).numpy() for t in terms])
    noise = np.stack([torch.cat(rec.randn[2 * k:2 * k + 2]).numpy() for k in range(N)])  # 2 chunks of 16 per step
    save("em_traj_lj13_nodebias.npz", x1=x1.numpy(), x_final=x.detach().numpy(), noise=noise, drift_X=drift,
         diffusion=diffusion, logweights=logw.numpy(), prior_scale=scale, N=N, chunk=16, gamma=4 / 3, beta=1.0,
         sigma_min=0.05)


def pcg_noise(seed, N, B, D):
    """The fixed noise of the long-trajectory fixtures: numpy's PCG64 stream, float32 ziggurat normals.  Stored as
    the SEED only; tests regenerate the identical array."""
    return np.random.Generator(np.random.PCG64(seed)).standard_normal((N, B, D), dtype=np.float32)


def gen_traj_1000():
    """The metric's own trajectory length: LJ13, N = 1000 (experiment/lj13.yaml), B = 16, not debiased, one inference
    chunk, trained-like weights, fixed PCG64 noise.  Records the walkers entering steps 0, 100, ..., 900 (the x the
    reference hands to VEReverseSDE.f, sde_integration.py:326-334), the final x, and drift norms at those steps."""
    wt = dict(np.load(os.path.join(HERE, "egnn_weights_trainedlike.npz")))
    sde, sched = build_lj13_stack(wt, debias=False)
    N, B, seed = 1000, 16, 20261003
    gamma = annealing_factor_schedules.ConstantAnnealingFactorSchedule(4 / 3)
    integ = sde_integration.WeightedSDEIntegrator(
        sde=sde, num_integration_steps=N, start_resampling_step=0, end_resampling_step=N, lightning_module=FakeLM(),
        partial_annealing_factor_schedule=None, resampling_interval=-1, num_negative_time_steps=0, post_mcmc_steps=0,
        batch_size=None, should_mean_free=True)
    e = LJ(39, 13, 3, data_path="", temperature=1.0)
    noise = pcg_noise(seed, N, B, 39)
    scale = float((sched.h(torch.tensor(1.0)) / gamma.gamma(torch.tensor(1.0))) ** 0.5)
    x1 = torch.from_numpy(pcg_noise(seed + 1, 1, B, 39)[0]) * scale
    x1 = data_utils.remove_mean(x1, 13, 3)
    calls = {"k": 0}
    xs = []
    real_f = sde.f

    def rec_f(t, x, *a, **k):
        if calls["k"] % 100 == 0:
            xs.append(x.detach().clone().numpy())
        calls["k"] += 1
        return real_f(t, x, *a, **k)

    sde.f = rec_f
    draws = {"k": 0}
    real_rn = torch.randn_like

    def fixed_randn_like(x, *a, **k):
        v = torch.from_numpy(noise[draws["k"]]).reshape(x.shape)
        draws["k"] += 1
        return v

    torch.randn_like = fixed_randn_like
    try:
        x, logw, uniq, terms, acc = integ.integrate_sde(x1.clone(), e, gamma, inverse_temperature=1.0)
    finally:
        torch.randn_like = real_rn
    assert calls["k"] == N and draws["k"] == N
    drift = np.stack([terms[k].drift_X.reshape(B, 39).numpy() for k in range(0, N, 100)] + [terms[N - 1].drift_X.reshape(B, 39).numpy()])
    save("em_traj_lj13_1000.npz", seed=seed, N=N, B=B, x1=x1.numpy(), x_at=np.stack(xs), at=np.arange(0, N, 100),
         x_final=x.detach().numpy(), drift_at=drift, prior_scale=scale, gamma=4 / 3, beta=1.0, sigma_min=0.05)


def gen_traj_1000_lj55():
    """Config C5's system at the metric's trajectory length: LJ55, EGNN h32x3 (trained-like weights of the LJ13 fixture:
    the parameter shapes do not depend on the particle count), N = 1000, 4 walkers, fixed PCG64 noise."""
    wt = dict(np.load(os.path.join(HERE, "egnn_weights_trainedlike.npz")))
    net = make_egnn(55, 3)
    net.load_state_dict({k: torch.tensor(v) for k, v in wt.items()})
    sn = score_net.ScoreNet(net)
    sched = noise_schedules.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
    sde = sdes.VEReverseSDE(noise_schedule=sched, energy_net=None, score_net=sn, cdf=lambda *a: None, debias_inference=False)
    sde.trainer = FakeTrainer()
    N, B, seed = 1000, 4, 20261055
    gamma = annealing_factor_schedules.ConstantAnnealingFactorSchedule(4 / 3)
    integ = sde_integration.WeightedSDEIntegrator(
        sde=sde, num_integration_steps=N, start_resampling_step=0, end_resampling_step=N, lightning_module=FakeLM(),
        partial_annealing_factor_schedule=None, resampling_interval=-1, num_negative_time_steps=0, post_mcmc_steps=0,
        batch_size=None, should_mean_free=True)
    e = LJ(165, 55, 3, data_path="", temperature=1.0)
    noise = pcg_noise(seed, N, B, 165)
    scale = float((sched.h(torch.tensor(1.0)) / gamma.gamma(torch.tensor(1.0))) ** 0.5)
    x1 = data_utils.remove_mean(torch.from_numpy(pcg_noise(seed + 1, 1, B, 165)[0]) * scale, 55, 3)
    calls, draws, xs = {"k": 0}, {"k": 0}, []
    real_f, real_rn = sde.f, torch.randn_like

    def rec_f(t, x, *a, **k):
        if calls["k"] % 250 == 0:
            xs.append(x.detach().clone().numpy())
        calls["k"] += 1
        return real_f(t, x, *a, **k)

    def fixed_randn_like(x, *a, **k):
        v = torch.from_numpy(noise[draws["k"]]).reshape(x.shape)
        draws["k"] += 1
        return v

    sde.f = rec_f
    torch.randn_like = fixed_randn_like
    try:
        x, logw, uniq, terms, acc = integ.integrate_sde(x1.clone(), e, gamma, inverse_temperature=1.0)
    finally:
        torch.randn_like = real_rn
    assert calls["k"] == N and draws["k"] == N
    save("em_traj_lj55_1000.npz", seed=seed, N=N, B=B, x1=x1.numpy(), x_at=np.stack(xs), at=np.arange(0, N, 250),
         x_final=x.detach().numpy(), prior_scale=scale, gamma=4 / 3, beta=1.0, sigma_min=0.05)


def gen_traj_debias():
    wt = dict(np.load(os.path.join(HERE, "egnn_weights_trainedlike.npz")))
    sde, sched = build_lj13_stack(wt, debias=True)
    N, B = 8, 12
    gamma = annealing_factor_schedules.ConstantAnnealingFactorSchedule(4 / 3)
    integ = sde_integration.WeightedSDEIntegrator(
        sde=sde, num_integration_steps=N, start_resampling_step=1, end_resampling_step=7, lightning_module=FakeLM(),
        partial_annealing_factor_schedule=None, resampling_interval=2, num_negative_time_steps=0, post_mcmc_steps=0,
        batch_size=12, no_grad=True, should_mean_free=True)
    e = LJ(39, 13, 3, data_path="", temperature=1.0)
    torch.manual_seed(77)
    scale = 3.0
    x1 = base_prior.Prior(scale=scale, n_particles=13, spatial_dim=3).sample(B)
    with Recorder() as rec:
        x, logw, uniq, terms, acc = integ.integrate_sde(x1.clone(), e, gamma, inverse_temperature=1.0)
    out = dict(x1=x1.numpy(), x_final=x.detach().numpy(), logweights=logw.detach().numpy(),
               num_unique=np.asarray(uniq), N=N, gamma=4 / 3, beta=1.0, sigma_min=0.05,
               noise=np.stack([r.numpy() for r in rec.randn]),
               u=np.stack([r.numpy() for r in rec.rand]) if rec.rand else np.zeros((0, 1)))
    for nm in ("drift_X", "drift_A", "divergence_score", "cross_term", "dUt_dt"):
        out[nm] = np.stack([getattr(t, nm).reshape(B, -1).squeeze(-1).detach().numpy() for t in terms])
    save("em_traj_lj13_debias.npz", **out)


def gen_traj_debias_end():
    """The LJ13 experiment's variant (experiment/lj13.yaml:27,41): two inference chunks per step (per-chunk quantile
    clamp, sdes.py:230) and ``resample_at_end=True`` (sde_integration.py:158-183)."""
    wt = dict(np.load(os.path.join(HERE, "egnn_weights_trainedlike.npz")))
    sde, sched = build_lj13_stack(wt, debias=True)
    N, B = 8, 12
    gamma = annealing_factor_schedules.ConstantAnnealingFactorSchedule(4 / 3)
    integ = sde_integration.WeightedSDEIntegrator(
        sde=sde, num_integration_steps=N, start_resampling_step=0, end_resampling_step=6, lightning_module=FakeLM(),
        partial_annealing_factor_schedule=None, resampling_interval=3, num_negative_time_steps=0, post_mcmc_steps=0,
        batch_size=6, no_grad=True, should_mean_free=True, resample_at_end=True)
    e = LJ(39, 13, 3, data_path="", temperature=1.0)
    torch.manual_seed(78)
    x1 = base_prior.Prior(scale=3.0, n_particles=13, spatial_dim=3).sample(B)
    with Recorder() as rec:
        x, logw, uniq, terms, acc = integ.integrate_sde(x1.clone(), e, gamma, inverse_temperature=1.0)
    noise = np.stack([torch.cat(rec.randn[2 * k:2 * k + 2]).numpy() for k in range(N)])  # 2 chunks of 6 per step
    save("em_traj_lj13_debias_end.npz", x1=x1.numpy(), x_final=x.detach().numpy(), logweights=logw.detach().numpy(),
         num_unique=np.asarray(uniq), N=N, gamma=4 / 3, beta=1.0, sigma_min=0.05, noise=noise,
         u=np.stack([r.numpy() for r in rec.rand]), chunk=6,
         drift_A=np.stack([t.drift_A.reshape(B).detach().numpy() for t in terms]))


def gen_traj_debias_long(weights="egnn_weights_trainedlike.npz", dt_mala=1e-13, name="em_traj_lj13_debias_long.npz", dt_alt=None):
    """PITA's DEFAULT regime at the LJ13 experiment's settings (configs/experiment/lj13.yaml:24-42 with
    model/energytemp.yaml:64-85) over a real horizon: ``debias_inference=True``, ``resampling_interval=1`` (an event
    after EVERY step of the window), two inference chunks per step (per-chunk 0.9-quantile clamp, sdes.py:230),
    ``resample_at_end=True`` (sde_integration.py:158-183), then the 5 adaptive MALA steps at ``dt_negative_time=1e-13``
    that quirk Q9 leaves switched on.  N = 200, B = 64, window [0, 160).  Every random number the reference draws is
    replaced by a seeded numpy PCG64 stream (the fixture stores the seeds; tests regenerate the arrays):
    ``randn_like`` -> pcg_noise(seed), ``torch.rand`` (one float64 uniform per resampling event, utils.py:112) ->
    PCG64(seed + 4), MALA ``randn_like`` -> pcg_noise(seed + 2), MALA ``rand_like`` -> PCG64(seed + 3).
    ``sample_cat_sys`` is wrapped to record the parent ids of every event."""
    wt = dict(np.load(os.path.join(HERE, weights)))
    sde, sched = build_lj13_stack(wt, debias=True)
    N, B, chunk, seed, end = 200, 64, 32, 20261004, 160
    n_mala = 5
    gamma = annealing_factor_schedules.ConstantAnnealingFactorSchedule(4 / 3)
    integ = sde_integration.WeightedSDEIntegrator(
        sde=sde, num_integration_steps=N, start_resampling_step=0, end_resampling_step=end, lightning_module=FakeLM(),
        partial_annealing_factor_schedule=None, resampling_interval=1, num_negative_time_steps=0,
        post_mcmc_steps=n_mala, adaptive_mcmc=True, dt_negative_time=dt_mala, batch_size=chunk, no_grad=True,
        should_mean_free=True, resample_at_end=True)
    e_raw = LJ(39, 13, 3, data_path="", temperature=1.0)

    class Detached:
        """See gen_post: the reference's MALA needs a detached alias under torch >= 2.10 to run at all."""

        is_molecule, n_particles, n_spatial_dim = True, 13, 3

        plain_calls = []  # walkers of the calls without force: [0] end-of-trajectory reweighting (:163), [1] MALA's first (:418)
        force_calls = []  # mala_proposal (:28-45) asks for the force at the chain's walkers, then at the proposal: [2k] = x of step k

        def __call__(self, x, return_force=False):
            if not return_force and len(self.plain_calls) < 2:
                self.plain_calls.append(x.detach().clone().numpy())
            if return_force:
                self.force_calls.append(x.detach().clone().numpy())
            return e_raw(x.detach(), return_force=return_force)

    Detached.plain_calls, Detached.force_calls = [], []
    e = Detached()
    noise = pcg_noise(seed, N, B, 39)
    mala_noise = pcg_noise(seed + 2, n_mala, B, 39)
    mala_u = np.random.Generator(np.random.PCG64(seed + 3)).random((n_mala, B), dtype=np.float32)
    us = np.random.Generator(np.random.PCG64(seed + 4)).random(end + 1)
    scale = float((sched.h(torch.tensor(1.0)) / gamma.gamma(torch.tensor(1.0))) ** 0.5)
    x1 = data_utils.remove_mean(torch.from_numpy(pcg_noise(seed + 1, 1, B, 39)[0]) * scale, 13, 3)

    cnt = {"f": 0, "rn": 0, "u": 0, "ru": 0}
    xs, ids_all = [], []
    real = (sde.f, torch.randn_like, torch.rand, torch.rand_like, sde_integration.sample_cat_sys)

    def rec_f(t, x, *a, **k):
        if cnt["f"] % (2 * 20) == 0:  # first chunk of steps 0, 20, 40, ...: the walkers ENTERING the step
            xs.append(None)
        if cnt["f"] % (2 * 20) in (0, 1):
            xs[-1] = x.detach().clone().numpy() if xs[-1] is None else np.concatenate([xs[-1], x.detach().numpy()])
        cnt["f"] += 1
        return real[0](t, x, *a, **k)

    def fixed_randn_like(x, *a, **k):
        i = cnt["rn"]
        cnt["rn"] += 1
        if i < 2 * N:
            return torch.from_numpy(noise[i // 2][(i % 2) * chunk:(i % 2 + 1) * chunk].copy()).reshape(x.shape)
        return torch.from_numpy(mala_noise[i - 2 * N].copy()).reshape(x.shape)

    def fixed_rand(*a, **k):
        assert k.get("dtype") == torch.float64 and tuple(k.get("size")) == (1,)
        v = torch.tensor([us[cnt["u"]]], dtype=torch.float64)
        cnt["u"] += 1
        return v

    def fixed_rand_like(x, *a, **k):
        v = torch.from_numpy(mala_u[cnt["ru"]].copy()).reshape(x.shape)
        cnt["ru"] += 1
        return v

    def rec_cat(bs, logits):
        ids, nu = real[4](bs, logits)
        ids_all.append(np.asarray(ids, dtype=np.int64).copy())
        return ids, nu

    sde.f = rec_f
    torch.randn_like, torch.rand, torch.rand_like = fixed_randn_like, fixed_rand, fixed_rand_like
    sde_integration.sample_cat_sys = rec_cat
    try:
        x, logw, uniq, terms, acc = integ.integrate_sde(x1.clone(), e, gamma, inverse_temperature=1.0)
        alt = None
        if dt_alt is not None:  # the same chain (same normals and uniforms) from the same walkers at a second step size
            counts, calls = dict(cnt), list(e.force_calls)
            cnt["rn"], cnt["ru"] = 2 * N, 0
            xa, acc_a = integ.metropolis_hastings_mala_adaptive(torch.from_numpy(e.plain_calls[1].copy()), e, dt_init=dt_alt,
                                                                return_acceptance_rate=True)
            alt = dict(x_final_alt=xa.detach().numpy(), mala_acc_alt=np.asarray(acc_a), dt_mala_alt=dt_alt)
            cnt.update(counts)
            Detached.force_calls = e.force_calls = calls
    finally:
        torch.randn_like, torch.rand, torch.rand_like = real[1:4]
        sde_integration.sample_cat_sys = real[4]
    # the reference's MALA draws one normal tensor in mala_proposal and one uniform tensor per step
    assert cnt == {"f": 2 * N, "rn": 2 * N + n_mala, "u": end + 1, "ru": n_mala}, cnt
    assert len(ids_all) == end + 1 and len(acc) == n_mala
    # the chain's walkers entering every MALA step + the final ones; a walker whose proposal was accepted moved by
    # ~sqrt(dt) (remove_mean of a rejected one moves it by rounding only): the accept mask of every step
    assert len(e.force_calls) == 2 * n_mala and all(c.shape == (B, 39) for c in e.force_calls)  # every log p finite: no reordering (:414-461)
    x_mala = np.stack([e.force_calls[2 * k] for k in range(n_mala)] + [x.detach().numpy()])
    moved = np.abs(x_mala[1:] - x_mala[:-1]).max(-1)
    mala_accept = moved > 1e-3 * np.sqrt(dt_mala)
    assert np.allclose(mala_accept.mean(1), np.asarray(acc), atol=1e-6), (mala_accept.mean(1), acc)
    out = dict(seed=seed, N=N, B=B, chunk=chunk, end=end, n_mala=n_mala, dt_mala=dt_mala, x1=x1.numpy(),
               x_mala=x_mala, mala_accept=mala_accept, logp_post_end=e_raw(torch.from_numpy(e.plain_calls[1])).numpy(),
               x_at=np.stack(xs), at=np.arange(0, N, 20), x_final=x.detach().numpy(), x_pre_end=e.plain_calls[0],
               x_post_end=e.plain_calls[1],
               logweights=logw.detach().numpy(), num_unique=np.asarray(uniq), ids=np.stack(ids_all).astype(np.int16),
               mala_acc=np.asarray(acc), prior_scale=scale, gamma=4 / 3, beta=1.0, sigma_min=0.05)
    for nm in ("drift_A", "divergence_score", "cross_term", "dUt_dt"):
        out[nm] = np.stack([getattr(t, nm).reshape(B).detach().numpy() for t in terms])
    if alt is not None:
        out.update(alt)
    save(name, **out)


def gen_traj_debias_long_init():
    """The same run on the seed-12345 INITIALISATION weights: a backbone whose output is small leaves the walkers near
    the EDM skip connection's Gaussian, so the run does NOT collapse -- log p of the walkers after the end-of-trajectory
    event is finite and of moderate size (-505 .. -755), and the five accept decisions per walker are decided by the
    arithmetic, not by the sign of a rounding difference.  MALA at dt_negative_time = 1e-5, where the decisions are MIXED
    (acceptance 0.41 .. 0.44; the accept mask of every step is stored), and the same chain once more at 4e-4 (the
    step size of post_lj13.npz), where the LJ forces at these walkers make every one of the 320 decisions a rejection."""
    gen_traj_debias_long(weights="egnn_weights_seed12345.npz", dt_mala=1e-5, name="em_traj_lj13_debias_long_init.npz",
                         dt_alt=4e-4)


def gen_debias_variants():
    """Debiased drift terms of VEReverseSDE.f (sdes.py:151-239) in the configurations the experiment files do not
    exercise: pin_energy=True (energy_net.py:43-48: the energy is blended with the clamped target energy, which the
    reference's energy classes return DETACHED, so only log p(x) -- not its force -- enters) and precondition_beta=True on
    both nets (score_net.py:36-38, energy_net.py:40-41), with a Linear annealing schedule (d gamma/dt != 0) and a
    Geometric noise schedule for one case (g^2 = dh/dt holds for it too; checks that dh/dt is not hard-wired to the
    Elucidating form)."""
    import copy

    wt = dict(np.load(os.path.join(HERE, "egnn_weights_trainedlike.npz")))
    e = LJ(39, 13, 3, data_path="", temperature=1.0)
    gen = torch.Generator().manual_seed(4242)
    B = 10
    # near-physical clusters so that log p_target is inside the +-1e3 clamp for most walkers and outside for some
    x = lattice_cluster(13, 3, B, gen, spacing=1.12, jitter=0.05)
    x[7:] = x[7:] * 0.55  # compressed: LJ energy > 1e3 -> clamp active
    x = data_utils.remove_mean(x, 13, 3)
    out = dict(x=x.numpy())
    gam = annealing_factor_schedules.LinearAnnealingFactorSchedule(annealing_factor=1.5, annealing_factor_start=1.0)
    for name, pin, pb, sched in (
            ("pin", True, False, noise_schedules.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)),
            ("pb", False, True, noise_schedules.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)),
            ("pinpb_geo", True, True, noise_schedules.GeometricNoiseSchedule(sigma_min=0.05, sigma_max=20.0))):
        net = make_egnn(13, 3)
        net.load_state_dict({k: torch.tensor(v) for k, v in wt.items()})
        sn = score_net.ScoreNet(net, precondition_beta=pb)
        en = energy_net.EnergyNet(copy.deepcopy(net), precondition_beta=pb)
        sde = sdes.VEReverseSDE(noise_schedule=sched, energy_net=en, score_net=sn,
                                cdf=partial(utils.compute_divergence_exact, sn.forward), pin_energy=pin,
                                debias_inference=True)
        sde.trainer = FakeTrainer()
        for ti, tv in enumerate((0.2, 0.7)):
            terms = sde.f(torch.tensor(tv), x.clone(), 1.25, gam, None, e, resampling_interval=1)
            for nm in ("drift_X", "drift_A", "divergence_score", "cross_term", "dUt_dt"):
                out[f"{name}_t{ti}_{nm}"] = getattr(terms, nm).detach().numpy()
        out[f"{name}_logp"] = e(x.clone()).numpy()
    out["t"] = np.asarray([0.2, 0.7], dtype=F32)
    out["beta"] = np.float32(1.25)
    save("debias_variants_lj13.npz", **out)


def gen_post(n=13, B=16, dt_mala=4e-4):
    """negative-time descent + MALA on the LJ13 / LJ55 target (sde_integration.py:353-470)."""
    e_raw = LJ(3 * n, n, 3, data_path="", temperature=1.0)

    class Detached:
        """torch>=2.10 refuses ``requires_grad = True`` on the non-leaf views the reference's MALA
        passes to LennardJonesEnergy.__call__ (lennardjones_energy.py:216) -- the reference then
        swallows the exception (sde_integration.py:401) and MALA silently does nothing.  Hand the
        energy a detached alias of the same storage so the reference arithmetic runs unchanged."""

        is_molecule, n_particles, n_spatial_dim = True, n, 3

        def __call__(self, x, return_force=False):
            return e_raw(x.detach(), return_force=return_force)

    e = Detached()
    gen = torch.Generator().manual_seed(5150 if n == 13 else 5150 + n)
    x0 = lattice_cluster(n, 3, B, gen, spacing=1.12, jitter=0.1 if n == 13 else 0.04)

    def mk(**kw):
        d = dict(sde=None, num_integration_steps=1, start_resampling_step=0, end_resampling_step=1,
                 lightning_module=FakeLM(), partial_annealing_factor_schedule=None)
        d.update(kw)
        return sde_integration.WeightedSDEIntegrator(**d)

    out = {"x0": x0.numpy()}
    integ = mk(num_negative_time_steps=25, dt_negative_time=1e-4, do_langevin=False)
    out["x_descent"] = integ.negative_time_descent(x0.clone(), e).detach().numpy()
    integ = mk(num_negative_time_steps=10, dt_negative_time=1e-4, do_langevin=True)
    torch.manual_seed(3)
    with Recorder() as rec:
        out["x_langevin"] = integ.negative_time_descent(x0.clone(), e).detach().numpy()
    out["langevin_noise"] = np.stack([r.numpy() for r in rec.randn])
    # plain MALA, 6 steps, dt chosen so acceptance is mixed
    integ = mk(post_mcmc_steps=6, dt_negative_time=dt_mala, adaptive_mcmc=False)
    torch.manual_seed(4)
    with Recorder() as rec:
        xm, accs = integ.metropolis_hastings_mala(x0.clone(), e, return_acceptance_rate=True)
    out["x_mala"] = xm.detach().numpy()
    out["mala_acc"] = np.asarray(accs)
    out["mala_noise"] = np.stack([r.numpy() for r in rec.randn])
    out["mala_u"] = np.stack([r.numpy() for r in rec.rand_like])
    # adaptive MALA
    integ = mk(post_mcmc_steps=6, dt_negative_time=dt_mala, adaptive_mcmc=True)
    torch.manual_seed(5)
    with Recorder() as rec:
        xa, accs = integ.metropolis_hastings_mala_adaptive(x0.clone(), e, dt_init=dt_mala, return_acceptance_rate=True)
    out["x_mala_adaptive"] = xa.detach().numpy()
    out["mala_adaptive_acc"] = np.asarray(accs)
    out["mala_adaptive_noise"] = np.stack([r.numpy() for r in rec.randn])
    out["mala_adaptive_u"] = np.stack([r.numpy() for r in rec.rand_like])
    out["dt_mala"] = dt_mala
    save(f"post_lj{n}.npz", **out)


def gen_post55():
    gen_post(n=55, B=8, dt_mala=1.2e-3)


def gen_traj_gmm():
    """Config C1 plumbing: 40-mode GMM target, MyMLP score net, 100 steps (B reduced to 64 here)."""
    w = {k[2:]: v for k, v in np.load(os.path.join(HERE, "mlp_gmm_fwd.npz")).items() if k.startswith("w.")}
    net = mlp.MyMLP(hidden_size=128, hidden_layers=3, emb_size=128, out_dim=2, input_dim=2)
    net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    sn = score_net.ScoreNet(net)
    sched = noise_schedules.ElucidatingNoiseSchedule(sigma_min=0.01, sigma_max=80.0, rho=7)
    sde = sdes.VEReverseSDE(noise_schedule=sched, energy_net=None, score_net=sn, cdf=lambda *a: None,
                            debias_inference=False)
    N, B = 100, 64
    gamma = annealing_factor_schedules.ConstantAnnealingFactorSchedule(1.0)
    integ = sde_integration.WeightedSDEIntegrator(
        sde=sde, num_integration_steps=N, start_resampling_step=0, end_resampling_step=N, lightning_module=FakeLM(),
        partial_annealing_factor_schedule=None, resampling_interval=-1, num_negative_time_steps=0, post_mcmc_steps=0,
        batch_size=None, should_mean_free=False)
    g = gmm_energy.GMM()
    torch.manual_seed(99)
    x1 = torch.randn(B, 2) * 80.0
    with Recorder() as rec:
        x, logw, uniq, terms, acc = integ.integrate_sde(x1.clone(), g, gamma, inverse_temperature=1.0)
    save("em_traj_gmm_mlp.npz", x1=x1.numpy(), x_final=x.detach().numpy(),
         noise=np.stack([r.numpy() for r in rec.randn]),
         drift_X=np.stack([t.drift_X.reshape(B, 2).numpy() for t in terms]), N=N, sigma_min=0.01)


if __name__ == "__main__":
    which = sys.argv[1:] or ["schedules", "lj", "lj_smooth", "gmm", "egnn", "egnn_ad2cat", "egnn_ad2cat_sizes", "egnn_aldp", "mlp", "prior", "resample", "traj_nodebias", "traj_1000",
                             "traj_debias", "traj_debias_end", "traj_debias_long", "traj_debias_long_init", "debias_variants", "post", "post55", "traj_1000_lj55", "traj_gmm"]
    for w in which:
        globals()["gen_" + w]()
