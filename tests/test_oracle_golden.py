"""Pin the CPU oracle (oracle/pita_oracle.py) to golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import pita_oracle as O

T = torch.tensor


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def test_schedules(golden):
    g = golden("schedules.npz")
    t = T(g["t"])
    for smin in (0.002, 0.01, 0.05):
        s = O.Elucidating(smin, 80.0, 7)
        np.testing.assert_array_equal(s.h(t).numpy(), g[f"h_{smin}"])
        np.testing.assert_array_equal(s.g(t).numpy(), g[f"g_{smin}"])
        np.testing.assert_array_equal(s.dh_dt(t).numpy(), g[f"dhdt_{smin}"])
        np.testing.assert_array_equal(s.t(s.h(t)).numpy(), g[f"tinv_{smin}"])
    geo = O.Geometric(0.01, 10.0)
    np.testing.assert_array_equal(geo.h(t).numpy(), g["geo_h"])
    np.testing.assert_array_equal(geo.g(t).numpy(), g["geo_g"])
    tt = T(g["tt"])
    for nm, sch in (("const", O.GammaConstant(4 / 3)), ("lin", O.GammaLinear(1.5, 1.0, 0.9, 0.1)),
                    ("sig", O.GammaSigmoid(1.5, 1.0, 0.9, 0.1, 10.0))):
        np.testing.assert_allclose(sch.gamma(tt).numpy(), g[f"gamma_{nm}"], rtol=1e-7)
        np.testing.assert_allclose(sch.dgamma_dt(tt).numpy(), g[f"dgamma_{nm}"], rtol=1e-7, atol=1e-30)


@pytest.mark.parametrize("n", [13, 55])
def test_lj(golden, n):
    g = golden(f"lj{n}_logp_force.npz")
    x = T(g["x"])
    nphys = int(g["n_cold"]) + int(g["n_warm"])
    for Tk in (1.0, 2.0, 4.0):
        lp, f = O.lj_logp_force(x, n, 3, temperature=Tk)
        # logp follows the reference op order: equal to rounding even on the adversarial rows
        np.testing.assert_allclose(lp.numpy(), g[f"logp_T{Tk}"], rtol=2e-6)
        # closed-form force vs the reference's autograd force
        assert rel(f[:nphys].numpy(), g[f"force_T{Tk}"][:nphys]) < 2e-6
        assert rel(f[nphys:].numpy(), g[f"force_T{Tk}"][nphys:]) < 1e-5
    lp, f = O.lj_logp_force(x, n, 3, temperature=1.0, energy_factor=0.5)
    np.testing.assert_allclose(lp.numpy(), g["logp_ef0.5"], rtol=2e-6)
    assert rel(f[:nphys].numpy(), g["force_ef0.5"][:nphys]) < 2e-6
    # fp64 oracle agrees with the fp32 reference to fp32 accuracy on physical configs
    lp64, f64 = O.lj_logp_force(x.double(), n, 3)
    assert rel(g["logp_T1.0"][:nphys], lp64[:nphys].numpy()) < 1e-6
    assert rel(g["force_T1.0"][:nphys], f64[:nphys].numpy()) < 1e-5


def test_lj_smooth_core(golden):
    """smooth=True (lennardjones_energy.py:114-119,131-133): spline coefficients and the blended log-density / autograd
    force against the reference's output; the fixture has walkers inside and outside the r < 0.65 core."""
    g = golden("lj13_smooth_logp_force.npz")
    assert int(g["n_core_walkers"]) >= 24
    xs, c = O.lj_smooth_coeffs()
    np.testing.assert_array_equal(c[:, 0].numpy(), g["spline_c0"])
    np.testing.assert_array_equal(xs[:1].numpy(), g["spline_x0"])
    x = T(g["x"])
    for Tk, ef in ((1.0, 1.0), (2.0, 0.5)):
        lp, f = O.lj_smooth_logp_force(x, 13, 3, temperature=Tk, energy_factor=ef)
        np.testing.assert_allclose(lp.numpy(), g[f"logp_T{Tk}_ef{ef}"], rtol=2e-6)
        assert rel(f.numpy(), g[f"force_T{Tk}_ef{ef}"]) < 2e-6
    # the option changes the answer exactly where pairs sit inside the core
    plain = O.lj_logp(x, 13, 3)
    smooth = O.lj_smooth_logp(x, 13, 3)
    assert torch.equal(plain[:24], smooth[:24]) and not torch.allclose(plain[24:48], smooth[24:48])


def test_lj_energy2_second_oracle(golden):
    """The in-tree restatement sampling/sample_lj13.py:energy2 (no distance eps) agrees
    with the bgflow-shimmed reference on physical configurations."""
    g = golden("lj13_logp_force.npz")
    nphys = int(g["n_cold"]) + int(g["n_warm"])
    x = T(g["x"][:nphys]).double()
    e2 = O.lj_energy2(x, 13).numpy()
    # residual = the 1e-6 distance eps (absent in energy2) + fp32 cancellation; SURVEY 8(c) measured 1.5e-4
    np.testing.assert_allclose(e2, g["logp_T1.0"][:nphys], rtol=2e-4, atol=1e-3)


def test_lj_against_reference_held_energy2(golden):
    """A12 pinned by reference-held code alone: the values below were produced by EXECUTING the reference's own
    sampling/sample_lj13.py:energy2 (torch.pdist; autograd for the force) in make_golden.py -- no bgflow restatement
    in the loop.  With the distance eps switched off the oracle's ordered-pair formula must equal it to fp64 rounding;
    with the bgflow eps of 1e-6 (the production setting) within 2e-4, the size of the eps term."""
    g = golden("lj13_logp_force.npz")
    nphys = int(g["n_cold"]) + int(g["n_warm"])
    x = T(g["x"]).double()
    lp0, f0 = O.lj_logp_force(x, 13, 3, dist_eps=0.0)
    np.testing.assert_allclose(lp0.numpy(), g["energy2_logp_f64"], rtol=1e-11)
    assert rel(f0.numpy(), g["energy2_force_f64"]) < 1e-11
    np.testing.assert_allclose(O.lj_energy2(x, 13).numpy(), g["energy2_logp_f64"], rtol=1e-11)
    lp, f = O.lj_logp_force(T(g["x"]), 13, 3)  # fp32, eps = 1e-6: what the product computes
    np.testing.assert_allclose(lp.numpy(), g["energy2_logp_f32"], rtol=2e-4)
    assert rel(f[:nphys].numpy(), g["energy2_force_f32"][:nphys]) < 2e-4
    # and the bgflow-shimmed reference class agrees with the reference-held function to the same bound
    np.testing.assert_allclose(g["logp_T1.0"], g["energy2_logp_f32"], rtol=2e-4)


def test_gmm(golden):
    g = golden("gmm40.npz")
    means, scale = O.gmm_params()
    np.testing.assert_array_equal(means.numpy(), g["means"])
    np.testing.assert_allclose(np.stack([np.diag(s) for s in g["scale_trils"]]), scale.numpy(), rtol=1e-7)
    x = T(g["x"])
    for Tk in (1.0, 2.0):
        np.testing.assert_allclose(O.gmm_logp(x, means, scale, Tk).numpy(), g[f"logp_T{Tk}"], rtol=2e-6, atol=2e-5)
    lp, grad = O.gmm_logp_force(x, means, scale, 1.0)
    np.testing.assert_allclose(grad.numpy(), g["grad_T1.0"], rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize("name", ["lj13", "dw4", "lj55"])
@pytest.mark.parametrize("tag,wfile", [("init", "egnn_weights_seed12345.npz"), ("trained", "egnn_weights_trainedlike.npz")])
def test_egnn(golden, name, tag, wfile):
    g = golden(f"egnn_{name}_fwd.npz")
    w = {k: T(v) for k, v in golden(wfile).items()}
    n, d = int(g["n"]), int(g["d"])
    x, h, beta = T(g["x"]), T(g["h"]), T(g["beta"])
    c_s, c_in, c_out, c_noise = O.edm_coeffs(h)
    np.testing.assert_array_equal(c_noise.numpy(), g["c_noise"])
    F, h0 = O.egnn_forward(w, c_noise, c_in[:, None] * x, beta, n, d, return_h0=True)
    np.testing.assert_array_equal(h0.numpy(), g["h0"])  # quirk Q1
    assert rel(F.numpy(), g[f"F_{tag}"]) < 2e-6
    bb = lambda cn, xs, b: O.egnn_forward(w, cn, xs, b, n, d)
    assert rel(O.denoiser(bb, h, x, beta).numpy(), g[f"D_{tag}"]) < 1e-6
    # score = (D - x)/h amplifies rounding at small h: compare per noise level
    sc = O.score(bb, h, x, beta).numpy()
    for hv in np.unique(g["h"]):
        m = g["h"] == hv
        tol = 5e-5 if hv > 0.05 else 3e-3
        assert rel(sc[m], g[f"score_{tag}"][m]) < tol, hv
    E = O.energy_theta(bb, h, x, beta).numpy()
    np.testing.assert_allclose(E, g[f"E_{tag}"], rtol=2e-4, atol=1e-3)


@pytest.mark.parametrize("tag,L,tanh,att", [("h64", 5, True, True), ("h48", 2, False, False)])
def test_egnn_ad2cat(golden, tag, L, tanh, att):
    """EGNN_dynamics_AD2_cat (the alanine-dipeptide backbone: one-hot atom types + t + beta, hidden 64 x 5 layers; a
    second net with hidden 48, no attention, no tanh): backbone output, denoiser and score of the reference module."""
    g = golden(f"egnn_ad2cat_{tag}_fwd.npz")
    w = {k[2:]: T(v) for k, v in g.items() if k.startswith("w.")}
    np.testing.assert_array_equal(O.egnn_ad2_cat_h_initial(22).numpy(), g["h_initial"])
    x, hs, beta = T(g["x"]), T(g["h"]), T(g["beta"])
    c_s, c_in, c_out, c_noise = O.edm_coeffs(hs)
    F = O.egnn_ad2_cat_forward(w, c_noise, c_in[:, None] * x, beta, 22, 3, n_layers=L, tanh=tanh, attention=att)
    assert rel(F.numpy(), g["F"]) < 2e-6
    bb = lambda cn, xs, b: O.egnn_ad2_cat_forward(w, cn, xs, b, 22, 3, n_layers=L, tanh=tanh, attention=att)
    assert rel(O.denoiser(bb, hs, x, beta).numpy(), g["D"]) < 2e-6
    # the pita_amd module owns parameters with the reference's names and shapes (checkpoints load unchanged)
    from pita_amd.egnn_dynamics_ad2_cat import EGNN_dynamics_AD2_cat

    m = EGNN_dynamics_AD2_cat(22, 3, hidden_nf=w["egnn.embedding.weight"].shape[0], n_layers=L, tanh=tanh, attention=att,
                              condition_beta=True)
    sd = m.state_dict()
    assert list(sd.keys()) == list(w.keys()) and all(sd[k].shape == w[k].shape for k in w)
    assert torch.equal(m.h_initial.float(), T(g["h_initial"]))


def test_egnn_ad2cat_other_sizes(golden):
    """The static node features of the other particle counts EGNN_dynamics_AD2_cat knows (13, 33, 42, 55) as the reference
    module builds them, and its backbone output for 33 and 42 atoms (two-layer hidden-32 net): oracle and the pita_amd
    module's feature tables against the reference."""
    from pita_amd.egnn_dynamics_ad2_cat import EGNN_dynamics_AD2_cat

    g = golden("egnn_ad2cat_sizes.npz")
    for n in (13, 33, 42, 55):
        np.testing.assert_array_equal(O.egnn_ad2_cat_h_initial(n).numpy(), g[f"h_initial_{n}"])
        m = EGNN_dynamics_AD2_cat(n, 3, hidden_nf=32, n_layers=2, condition_beta=True)
        assert torch.equal(m.h_initial.float(), T(g[f"h_initial_{n}"]))
    for n in (33, 42):
        w = {k[len(f"w{n}."):]: T(v) for k, v in g.items() if k.startswith(f"w{n}.")}
        F = O.egnn_ad2_cat_forward(w, T(g[f"t_{n}"]), T(g[f"x_{n}"]), T(g[f"beta_{n}"]), n, 3, n_layers=2)
        assert rel(F.numpy(), g[f"F_{n}"]) < 2e-6, n


def test_egnn_quirk_layout():
    h0 = O.egnn_node_features(T([0.5]), T([2.0]), 13).numpy()
    assert (h0[:6] == [0.5, 0.5]).all() and (h0[6] == [0.5, 2.0]).all() and (h0[7:] == [2.0, 2.0]).all()


def test_egnn_notemp(golden):
    g = golden("egnn_notemp_lj13_fwd.npz")
    w = {k[2:]: T(v) for k, v in g.items() if k.startswith("w.")}
    out = O.egnn_forward(w, T(g["t"]), T(g["x"]), None, 13, 3)
    # fresh-init velocities are ~1e-4 * |x| (xavier gain 1e-3 head): x_final - x cancels ~4 digits
    assert rel(out.numpy(), g["out"]) < 5e-5


def test_mlp(golden):
    g = golden("mlp_gmm_fwd.npz")
    w = {k[2:]: T(v) for k, v in g.items() if k.startswith("w.")}
    x, h = T(g["x"]), T(g["h"])
    c_s, c_in, c_out, c_noise = O.edm_coeffs(h)
    F = O.mlp_forward(w, c_noise, c_in[:, None] * x)
    assert rel(F.numpy(), g["F"]) < 2e-6
    bb = lambda cn, xs, b: O.mlp_forward(w, cn, xs)
    sc = O.score(bb, h, x, 1.0).numpy()
    big = g["h"] > 1e-2
    assert rel(sc[big], g["score"][big]) < 1e-4
    g2 = golden("mlp_temp_fwd.npz")
    w2 = {k[2:]: T(v) for k, v in g2.items() if k.startswith("w.")}
    y = O.mlp_forward(w2, T(g2["t"]), T(g2["x"]), T(g2["beta"]), emb_size=64, hidden_layers=2, temperature_conditioned=True)
    assert rel(y.numpy(), g2["out"]) < 2e-6


def test_prior_and_remove_mean(golden):
    g = golden("prior.npz")
    for n, d in ((13, 3), (4, 2)):
        s = O.prior_from_noise(T(g[f"noise_{n}"]), float(g["scale"]), n, d)
        np.testing.assert_array_equal(s.numpy(), g[f"sample_{n}"])
        np.testing.assert_array_equal(O.remove_mean(T(g[f"noise_{n}"]), n, d).numpy(), g[f"remove_mean_{n}"])


@pytest.mark.parametrize("case", ["normal", "ties", "peaked", "neginf", "huge", "big"])
def test_resample(golden, case):
    g = golden("resample_sys.npz")
    ids = O.sample_cat_sys(T(g[f"logits_{case}"]), float(g[f"u_{case}"][0]))
    np.testing.assert_array_equal(ids, g[f"ids_{case}"])
    lg = T(g[f"logits_{case}"])
    np.testing.assert_array_equal(torch.quantile(lg[torch.isfinite(lg)], 0.9).numpy(), g[f"q90_{case}"])


def test_egnn_aldp(golden):
    """``egnn_aldp.EGNN_dynamics`` (the reference's other peptide EGNN, egnn_aldp.py:8-197): its static node features
    and its output for the class defaults (22 atoms, hidden 64 x 4, no gate, no tanh, temperature conditioned) and for a
    33-atom net with gate and tanh, against the REFERENCE module's own output; and the pita_amd mirror builds the same
    feature table and parameter names."""
    from pita_amd.egnn_aldp import EGNN_dynamics

    g = golden("egnn_aldp_fwd.npz")
    for tag, n, kw in (("n22", 22, dict(n_layers=4, tanh=False, attention=False)),
                       ("n33", 33, dict(n_layers=2, tanh=True, attention=True))):
        w = {k[len(f"w_{tag}."):]: T(v) for k, v in g.items() if k.startswith(f"w_{tag}.")}
        hi = O.egnn_aldp_h_initial(n)
        np.testing.assert_array_equal(hi.numpy(), g[f"h_initial_{tag}"])
        x, t, beta = T(g[f"x_{tag}"]), T(g[f"t_{tag}"]), T(g[f"beta_{tag}"])
        F = O.egnn_ad2_cat_forward(w, t, x, beta, n, 3, h_initial=hi, **kw)
        F64 = O.egnn_ad2_cat_forward({k: v.double() for k, v in w.items()}, t.double(), x.double(), beta.double(), n, 3,
                                     h_initial=hi.double(), **kw)
        err_ref = rel(g[f"F_{tag}"], F64.numpy())
        assert rel(F.numpy(), g[f"F_{tag}"]) < max(2e-5, 6 * err_ref)
        H = w["egnn.embedding.weight"].shape[0]
        net = EGNN_dynamics(n, 3, hidden_nf=H, condition_temperature=True, **kw)
        np.testing.assert_array_equal(net.h_initial.numpy().astype(np.float32), g[f"h_initial_{tag}"])
        assert list(net.state_dict().keys()) == list(w.keys())
        net.load_state_dict(w)  # shapes agree
    with pytest.raises(NotImplementedError):
        EGNN_dynamics(53, 3)  # node features from a topology file


def _lj13_backbone(golden):
    w = {k: T(v) for k, v in golden("egnn_weights_trainedlike.npz").items()}
    return lambda cn, xs, b: O.egnn_forward(w, cn, xs, b, 13, 3)


def test_traj_nodebias(golden):
    g = golden("em_traj_lj13_nodebias.npz")
    bb = _lj13_backbone(golden)
    sched, gam = O.Elucidating(0.05, 80.0, 7), O.GammaConstant(4 / 3)
    assert abs(O.prior_scale(sched, gam, 1.0) - float(g["prior_scale"])) < 1e-4
    N, chunk = int(g["N"]), int(g["chunk"])
    noise = T(g["noise"])
    cfg = O.IntegratorConfig(num_integration_steps=N, end_resampling_step=N, batch_size=chunk)
    drift = lambda t, xc: O.f_not_debiased(bb, sched, gam, t, xc, 1.0)

    def noise_fn(i, shape):  # draw i = step*2 + chunk index
        return noise[i // 2, (i % 2) * chunk:(i % 2 + 1) * chunk]

    out = O.integrate_sde(cfg, T(g["x1"]), drift, sched.g, noise_fn, 13, 3, record=True)
    # per-step drift parity (north_star: "per-step drift"): same inputs each step only up to
    # accumulated rounding, so compare step 0 tightly and the rest loosely
    assert rel(out["drift_X"][0].numpy(), g["drift_X"][0]) < 1e-5
    for k in range(N):
        assert rel(out["drift_X"][k].numpy(), g["drift_X"][k]) < 2e-3, k
    assert rel(out["x"].numpy(), g["x_final"]) < 1e-4
    assert np.all(out["logweights"].numpy() == 0)


def pcg_noise(seed, N, B, D):
    return np.random.Generator(np.random.PCG64(seed)).standard_normal((N, B, D), dtype=np.float32)


@pytest.mark.parametrize("n", [13, 55])
def test_traj_1000_steps(golden, n):
    """The metric's own trajectory length (experiment/lj13.yaml: 1 000 steps): the reference's integrate_sde on fixed
    PCG64 noise, walkers recorded every 100 steps.  The fp32 oracle must track the reference as closely as the
    reference tracks the fp64 oracle (both carry fp32 rounding through the small-h end where score = (D - x)/h
    amplifies it), at every checkpoint."""
    g = golden(f"em_traj_lj{n}_1000.npz")  # LJ55: config C5's system, 4 walkers, checkpoints every 250 steps
    D = 3 * n
    N, B = int(g["N"]), int(g["B"])
    noise = pcg_noise(int(g["seed"]), N, B, D)
    np.testing.assert_array_equal(O.remove_mean(T(pcg_noise(int(g["seed"]) + 1, 1, B, D)[0]) * float(g["prior_scale"]),
                                                n, 3).numpy(), g["x1"])
    at = list(g["at"]) + [N]
    want = list(g["x_at"]) + [g["x_final"]]
    runs = {}
    for dt in (torch.float32, torch.float64):
        w = {k: T(v).to(dt) for k, v in golden("egnn_weights_trainedlike.npz").items()}
        bbd = lambda cn, xs, b: O.egnn_forward(w, cn, xs, b, n, 3)
        sched, gam = O.Elucidating(0.05, 80.0, 7), O.GammaConstant(4 / 3)
        cfg = O.IntegratorConfig(num_integration_steps=N, end_resampling_step=N)
        nz = T(noise).to(dt)
        x1 = T(g["x1"]).to(dt)
        out = O.integrate_sde(cfg, x1, lambda t, xc: O.f_not_debiased(bbd, sched, gam, t, xc, 1.0),
                              sched.g, lambda i, shp: nz[i], n, 3, record=True)
        # traj[k] = walkers after step k = walkers entering step k + 1
        runs[dt] = [x1.numpy()] + [out["traj"][a - 1].numpy() for a in at[1:]]
    for k, (a, ref) in enumerate(zip(at, want)):
        e_ref = rel(ref, runs[torch.float64][k])  # the reference's own fp32 error against fp64 arithmetic
        e_o32 = rel(runs[torch.float32][k], runs[torch.float64][k])
        print(f"[traj1000/lj{n}] step {a:4d}: reference fp32 vs fp64 oracle {e_ref:.2e}, fp32 oracle vs fp64 oracle {e_o32:.2e}")
        assert e_o32 <= 4 * e_ref + 1e-6, (a, e_o32, e_ref)
        assert rel(runs[torch.float32][k], ref) <= 8 * e_ref + 1e-6


def test_traj_debias(golden):
    g = golden("em_traj_lj13_debias.npz")
    bb = _lj13_backbone(golden)
    sched, gam = O.Elucidating(0.05, 80.0, 7), O.GammaConstant(4 / 3)
    N = int(g["N"])
    noise, us = T(g["noise"]), g["u"]
    cfg = O.IntegratorConfig(num_integration_steps=N, start_resampling_step=1, end_resampling_step=7,
                             resampling_interval=2, batch_size=12)
    drift = lambda t, xc: O.f_debiased(bb, bb, sched, gam, t, xc, 1.0)
    resample_steps = [s for s in range(N) if (s + 1) % 2 == 0 and 1 <= s < 7]
    umap = {s: float(us[i][0]) for i, s in enumerate(resample_steps)}
    out = O.integrate_sde(cfg, T(g["x1"]), drift, sched.g, lambda i, shp: noise[i], 13, 3, uniform_fn=lambda s: umap[s],
                          record=True)
    assert out["num_unique"] == list(g["num_unique"])
    assert rel(out["x"].numpy(), g["x_final"]) < 2e-3
    np.testing.assert_allclose(out["logweights"].numpy(), g["logweights"], rtol=5e-3, atol=5e-3)
    # first-step terms (identical inputs)
    t0 = torch.tensor(1.0)
    terms = O.f_debiased(bb, bb, sched, gam, t0, T(g["x1"]), 1.0)
    assert rel(terms.drift_X.numpy(), g["drift_X"][0]) < 1e-4
    np.testing.assert_allclose(terms.drift_A.numpy(), g["drift_A"][0], rtol=2e-3, atol=1e-2)
    np.testing.assert_allclose(terms.divergence_score.numpy(), g["divergence_score"][0], rtol=2e-3, atol=1e-2)


VARIANTS = (("pin", True, False, "elucidating"), ("pb", False, True, "elucidating"), ("pinpb_geo", True, True, "geometric"))


def test_debias_variants(golden):
    """pin_energy / precondition_beta / a Geometric schedule / a Linear annealing schedule in the debiased drift
    (debias_variants_lj13.npz: VEReverseSDE.f of the reference itself)."""
    g = golden("debias_variants_lj13.npz")
    bb = _lj13_backbone(golden)
    x, beta = T(g["x"]), float(g["beta"])
    gam = O.GammaLinear(annealing_factor=1.5, annealing_factor_start=1.0)
    np.testing.assert_allclose(O.lj_logp(x, 13, 3).numpy(), g["pin_logp"], rtol=2e-5)
    for name, pin, pb, sch in VARIANTS:
        sched = O.Elucidating(0.05, 80.0, 7) if sch == "elucidating" else O.Geometric(0.05, 20.0)
        for ti, tv in enumerate(g["t"]):
            terms = O.f_debiased(bb, bb, sched, gam, torch.tensor(float(tv)), x, beta, pin_energy=pin,
                                 target_logp=lambda xx: O.lj_logp(xx, 13, 3), precondition_beta=pb)
            key = f"{name}_t{ti}_"
            assert rel(terms.drift_X.numpy(), g[key + "drift_X"]) < 2e-4, key
            for nm in ("drift_A", "divergence_score", "cross_term", "dUt_dt"):
                ref = g[key + nm]
                np.testing.assert_allclose(getattr(terms, nm).numpy(), ref, rtol=3e-3, atol=3e-3 * np.abs(ref).max(),
                                           err_msg=key + nm)


def test_traj_debias_resample_at_end(golden):
    """The LJ13 experiment's settings: two inference chunks per step (per-chunk clamp) and resample_at_end."""
    g = golden("em_traj_lj13_debias_end.npz")
    bb = _lj13_backbone(golden)
    sched, gam = O.Elucidating(0.05, 80.0, 7), O.GammaConstant(4 / 3)
    N = int(g["N"])
    noise, us = T(g["noise"]), g["u"]
    cfg = O.IntegratorConfig(num_integration_steps=N, start_resampling_step=0, end_resampling_step=6,
                             resampling_interval=3, batch_size=6)
    drift = lambda t, xc: O.f_debiased(bb, bb, sched, gam, t, xc, 1.0)
    umap = {2: float(us[0][0]), 5: float(us[1][0])}
    draws = iter(range(10**6))
    out = O.integrate_sde(cfg, T(g["x1"]), drift, sched.g, lambda i, shp: noise[i // 2][(i % 2) * 6:(i % 2) * 6 + 6], 13, 3,
                          uniform_fn=lambda s: umap[s])
    np.testing.assert_allclose(out["logweights"].numpy(), g["logweights"][:N], rtol=5e-3, atol=5e-3)
    assert out["num_unique"] == list(g["num_unique"][:N])
    t_end = torch.linspace(1.0, 0.0, N + 1)[:-1][6]
    x, a_next, nu = O.resample_at_end(out["x"], out["logweights"][-1], t_end, lambda x: O.lj_logp(x, 13, 3),
                                      lambda tb, x: O.energy_theta(bb, sched.h(tb), x, 1.0), 4 / 3, float(us[2][0]))
    np.testing.assert_allclose(a_next.numpy(), g["logweights"][N], rtol=2e-4)
    assert nu == int(g["num_unique"][N])
    assert rel(x.numpy(), g["x_final"]) < 2e-3


@pytest.mark.parametrize("which", ["trainedlike", "init"])
def test_traj_debias_long_default_regime(golden, which):
    """PITA's default regime at the LJ13 experiment's settings (em_traj_lj13_debias_long.npz: the reference's
    integrate_sde, debiased, an event after EVERY step of [0, 160), two clamp chunks of 32, resample_at_end, 5 adaptive
    MALA steps at dt = 1e-13; N = 200, B = 64).  The oracle restarts from the ten recorded walker sets (the x entering
    steps 0, 20, ..., 180) and runs 4 steps from each on the fixture's noise and uniforms: every weight-drift term of
    every step against the reference's, and the parent ids of every event IDENTICAL to the reference's (the reference's
    ids are fed forward, so a segment never drifts off the recorded trajectory); then the end-of-trajectory reweighting
    from the recorded pre-event walkers and the MALA chain.  (The full 200-step run is the GPU test's; on this CPU it
    would take minutes.)

    ``init``: the same run on the seed-12345 initialisation weights (em_traj_lj13_debias_long_init.npz), which does not
    collapse: log p after the end-of-trajectory event is -505 .. -755, so the MALA accept decisions are arithmetic, and
    the accept MASK of every step and the walkers after every step are compared exactly / tightly, at dt = 1e-5 (mixed
    decisions) and at 4e-4 (all rejected)."""
    from tests._long_fixture import ids_mismatch_is_bin_edge_tie, long_fixture_draws

    init = which == "init"
    g = golden("em_traj_lj13_debias_long_init.npz" if init else "em_traj_lj13_debias_long.npz")
    if init:
        w = {k: T(v) for k, v in golden("egnn_weights_seed12345.npz").items()}
        bb = lambda cn, xs, b: O.egnn_forward(w, cn, xs, b, 13, 3)
    else:
        bb = _lj13_backbone(golden)
    sched, gam = O.Elucidating(0.05, 80.0, 7), O.GammaConstant(4 / 3)
    N, B, chunk, end = (int(g[k]) for k in ("N", "B", "chunk", "end"))
    noise, mala_noise, mala_u, us = long_fixture_draws(g)
    times = torch.linspace(1.0, 0.0, N + 1)[:-1]
    dt = 1.0 / N
    seg = 4
    n_events = n_tie = 0
    for k, s0 in enumerate(g["at"]):
        x = T(g["x_at"][k])
        for s in range(int(s0), int(s0) + seg):
            t = times[s]
            parts = [O.f_debiased(bb, bb, sched, gam, t, x[lo:lo + chunk], 1.0) for lo in range(0, B, chunk)]
            for nm, tol in (("drift_A", 2e-3), ("divergence_score", 2e-3), ("cross_term", 2e-3), ("dUt_dt", 2e-3)):
                got = torch.cat([getattr(p_, nm) for p_ in parts]).numpy()
                np.testing.assert_allclose(got, g[nm][s], rtol=tol, atol=tol * float(np.abs(g[nm][s]).mean()),
                                           err_msg=f"step {s} {nm}")
            dX = torch.cat([p_.drift_X for p_ in parts])
            dA = torch.cat([p_.drift_A for p_ in parts])
            tb = t * torch.ones(B)
            x = x + (dX * dt + (sched.g(tb)[:, None] * T(noise[s])) * np.sqrt(dt))
            if s < end:  # resampling_interval = 1: the log-weights of ONE step decide the event, then reset (:283-297)
                a = dA * dt
                ids = O.sample_cat_sys(a, float(us[s]))
                want = g["ids"][s].astype(np.int64)
                if not np.array_equal(ids, want):
                    assert ids_mismatch_is_bin_edge_tie(a.numpy(), float(us[s]), ids, want), f"event {s}"
                    n_tie += 1
                n_events += 1
                assert len(np.unique(want)) == int(g["num_unique"][s])
                x = x[torch.from_numpy(want)]
            x = O.remove_mean(x, 13, 3)
    assert n_events == 4 * 8 and n_tie <= 1, (n_events, n_tie)
    np.testing.assert_array_equal(g["logweights"][:N], 0.0)  # every in-window step resets a; outside the window a = 0
    # end of trajectory (sde_integration.py:158-183): time of step `end`, a = 0 after the window
    x_pre = T(g["x_pre_end"])
    x, a_next, nu = O.resample_at_end(x_pre, torch.zeros(B), times[end], lambda xx: O.lj_logp(xx, 13, 3),
                                      lambda tb, xx: O.energy_theta(bb, sched.h(tb), xx, 1.0), 4 / 3, float(us[end]))
    np.testing.assert_allclose(a_next.numpy(), g["logweights"][N], rtol=2e-4)
    assert nu == int(g["num_unique"][N])
    np.testing.assert_array_equal(x.numpy(), g["x_post_end"])
    # 5 adaptive MALA steps at dt = 1e-13 (quirk Q9).  log p is ~ -1e10 .. -1e21 on these collapsed walkers, so every
    # accept decision is the sign of a rounding difference: rates are compared loosely, x tightly (moves are ~3e-7).
    lf = lambda xx: O.lj_logp_force(xx, 13, 3)
    x_post = x
    for dt0, tag in ((float(g["dt_mala"]), ""),) + (((float(g["dt_mala_alt"]), "_alt"),) if init else ()):
        x, lp, dtm = x_post, O.lj_logp(x_post, 13, 3), dt0
        if init:
            np.testing.assert_allclose(lp.numpy(), g["logp_post_end"], rtol=1e-6)
        for k in range(int(g["n_mala"])):
            x, lp, acc = O.mala_step(x, lp, lf, dtm, T(mala_noise[k]), torch.log(T(mala_u[k])))
            x = O.remove_mean(x, 13, 3)
            r = acc.float().mean().item()
            if init:  # decided by finite log-densities: the mask itself, and the chain's walkers after the step
                assert r == g["mala_acc" + tag][k]
                if not tag:
                    np.testing.assert_array_equal(acc.numpy().astype(bool), g["mala_accept"][k])
                    assert rel(x.numpy(), g["x_mala"][k + 1]) < 1e-6
            else:
                assert abs(r - g["mala_acc"][k]) < 0.25
            dtm = dtm * 1.1 if r > 0.55 else dtm / 1.1
        assert rel(x.numpy(), g["x_final" + tag]) < (1e-6 if init else 1e-5)


@pytest.mark.parametrize("n", [13, 55])
def test_post(golden, n):
    """negative-time descent, Langevin descent, MALA and adaptive MALA of the reference on the LJ13 and LJ55 targets."""
    g = golden(f"post_lj{n}.npz")
    dtm = float(g["dt_mala"])
    lf = lambda x: O.lj_logp_force(x, n, 3)
    x0 = T(g["x0"])
    xd = O.negative_time_descent(x0, lf, 25, 1e-4, n, 3)
    assert rel(xd.numpy(), g["x_descent"]) < 1e-6
    ln = T(g["langevin_noise"])
    xl = O.negative_time_descent(x0, lf, 10, 1e-4, n, 3, do_langevin=True, noise_fn=lambda k, s: ln[k])
    assert rel(xl.numpy(), g["x_langevin"]) < 1e-6
    # MALA: the reference draws 1 proposal-noise tensor and 1 uniform tensor per step
    x, lp = x0.clone(), O.lj_logp(x0, n, 3)
    accs = []
    for k in range(6):
        x, lp, acc = O.mala_step(x, lp, lf, dtm, T(g["mala_noise"][k]), torch.log(T(g["mala_u"][k])))
        x = O.remove_mean(x, n, 3)
        accs.append(acc.float().mean().item())
    np.testing.assert_allclose(accs, g["mala_acc"], atol=1e-7)
    assert rel(x.numpy(), g["x_mala"]) < 1e-6
    # adaptive
    x, lp, dt = x0.clone(), O.lj_logp(x0, n, 3), dtm
    for k in range(6):
        x, lp, acc = O.mala_step(x, lp, lf, dt, T(g["mala_adaptive_noise"][k]), torch.log(T(g["mala_adaptive_u"][k])))
        x = O.remove_mean(x, n, 3)
        a = acc.float().mean().item()
        dt = dt * 1.1 if a > 0.55 else dt / 1.1
        assert abs(a - g["mala_adaptive_acc"][k]) < 1e-7
    assert rel(x.numpy(), g["x_mala_adaptive"]) < 1e-6


def test_traj_gmm_mlp(golden):
    """Config C1 plumbing: GMM target + MyMLP score net, 100 steps."""
    g = golden("em_traj_gmm_mlp.npz")
    w = {k[2:]: T(v) for k, v in golden("mlp_gmm_fwd.npz").items() if k.startswith("w.")}
    bb = lambda cn, xs, b: O.mlp_forward(w, cn, xs)
    sched, gam = O.Elucidating(0.01, 80.0, 7), O.GammaConstant(1.0)
    N = int(g["N"])
    noise = T(g["noise"])
    cfg = O.IntegratorConfig(num_integration_steps=N, end_resampling_step=N, should_mean_free=False)
    out = O.integrate_sde(cfg, T(g["x1"]), lambda t, xc: O.f_not_debiased(bb, sched, gam, t, xc, 1.0), sched.g,
                          lambda i, shp: noise[i], 1, 2, record=True)
    assert rel(out["drift_X"][0].numpy(), g["drift_X"][0]) < 1e-5
    assert rel(out["x"].numpy(), g["x_final"]) < 5e-3


def test_dw4_force_is_gradient():
    """DW4 is parity-unpinned (not in the reference): check self-consistency force = d logp/dx."""
    torch.manual_seed(0)
    x = (torch.randn(8, 8) * 2).double().requires_grad_(True)
    lp, f = O.dw4_logp_force(x)
    (gr,) = torch.autograd.grad(lp.sum(), x)
    np.testing.assert_allclose(f.detach().numpy(), gr.numpy(), rtol=1e-9, atol=1e-9)


def test_w2():
    a = np.random.default_rng(0).normal(size=1000)
    assert O.w2_1d(a, a) == 0
    assert abs(O.w2_1d(a, a + 2.0) - 2.0) < 1e-12
    assert abs(O.w2_1d(a[:500], np.concatenate([a[:500], a[:500]])) ) < 1e-9


def test_gbsa_obc1_limits():
    """GB-OBC1 restatement (parity unpinned): closed-form limits.  One ion: Born radius = offset radius, energy =
    Born self energy + ACE term.  Two distant ions: Born radii -> offset radii and the pair term -> the screened
    Coulomb correction -k_e (1 - 1/eps) q1 q2 / r."""
    q = torch.tensor([0.7, -0.4], dtype=torch.float64)
    R = torch.tensor([0.15, 0.17], dtype=torch.float64)
    s = torch.tensor([0.8, 0.72], dtype=torch.float64)
    pf = -O.ONE_4PI_EPS0 * (1.0 - 1.0 / 78.5)
    one = O.gbsa_obc1_energy(torch.zeros(1, 1, 3, dtype=torch.float64), q[:1], R[:1], s[:1])
    rho = R - 0.009
    want1 = 0.5 * pf * q[0] ** 2 / rho[0] + 28.3919551 * (R[0] + 0.14) ** 2 * (R[0] / rho[0]) ** 6
    assert abs(one.item() - want1.item()) < 1e-10
    r = torch.zeros(1, 2, 3, dtype=torch.float64)
    r[0, 1, 0] = 50.0
    two = O.gbsa_obc1_energy(r, q, R, s)
    self2 = sum(0.5 * pf * q[i] ** 2 / rho[i] + 28.3919551 * (R[i] + 0.14) ** 2 * (R[i] / rho[i]) ** 6 for i in range(2))
    assert abs(two.item() - (self2 + pf * q[0] * q[1] / 50.0).item()) < 1e-4
    # descreening: a buried neighbour raises the Born radius, i.e. weakens |self energy|
    r[0, 1, 0] = 0.2
    near = O.gbsa_obc1_energy(r, torch.tensor([0.7, 0.0], dtype=torch.float64), R, s, sa_factor=0.0)
    far = 0.5 * pf * 0.49 / rho[0]
    assert far < near.item() < 0.0
