"""CPU tests of host-side logic that needs no GPU."""
from dataclasses import fields

import numpy as np
import torch

import pita_amd
from pita_amd.sdes import SDETerms, TermStats


def test_term_stats_answer_mean_and_std_like_the_tensor():
    gen = torch.Generator().manual_seed(0)
    v = torch.randn(37, 39, generator=gen) * 3 + 0.7
    ts = TermStats(float(v.double().sum()), float((v.double() ** 2).sum()), v.numel())
    assert abs(float(ts.mean()) - float(v.mean())) < 1e-6
    assert abs(float(ts.std()) - float(v.std())) < 1e-5
    assert ts.mean().dim() == 0 and ts.std().dtype == torch.float32
    assert ts.cpu() is ts and ts.numel() == v.numel()
    empty = TermStats(0.0, 0.0, 0)
    assert np.isnan(float(empty.mean())) and np.isnan(float(empty.std()))
    # the reference's consumer stacks the 0-dim results of every step (energytemp_module.py:1138-1143)
    stacked = torch.stack([ts.mean(), empty.mean()])
    assert stacked.shape == (2,)


def test_terms_from_stats_layout():
    st4 = torch.tensor([[1.0, 2.0, 3.0, 4.0], [0.0, 0.0, 0.0, 0.0]], dtype=torch.float64)
    st8 = torch.arange(16, dtype=torch.float64).reshape(2, 8)
    terms = pita_amd.sde_integration._terms_from_stats(st4, st8, 10, 5, [True, False], True)
    assert [f.name for f in fields(SDETerms)] == ["drift_X", "drift_A", "divergence_score", "cross_term", "dUt_dt",
                                                   "diffusion"]  # sdes.py:34-41
    assert abs(float(terms[0].drift_X.mean()) - 0.1) < 1e-7 and terms[0].diffusion.s2 == 4.0
    assert terms[0].drift_A.s == 0.0 and terms[0].divergence_score.s == 2.0 and terms[0].dUt_dt.s2 == 7.0
    assert terms[1].cross_term.n == 0 and np.isnan(float(terms[1].cross_term.mean()))
    nodeb = pita_amd.sde_integration._terms_from_stats(st4, None, 10, 5, [True, True], False)
    assert nodeb[0].divergence_score is None and nodeb[0].drift_A.n == 5


def test_smooth_lj_core_coefficients_match_the_reference(golden):
    """The four spline coefficients LennardJonesEnergy(smooth=True) hands to pita_lj_smooth_logp_force are the reference's
    (lennardjones_energy.py:114-119: scipy CubicSpline through the float32 LJ curve, first interval), bit for bit."""
    import numpy as np

    from pita_amd.lennardjones_energy import smooth_core_coefficients

    g = golden("lj13_smooth_logp_force.npz")
    coef, r0 = smooth_core_coefficients()
    assert coef.dtype == np.float32 and coef.shape == (4,)
    np.testing.assert_array_equal(coef, g["spline_c0"])
    assert np.float32(r0) == g["spline_x0"][0]


def test_alp_energy_facade_and_system_xml_fixture():
    """ALPEnergy mirrors the reference constructor (pita/src/energies/alp_energy.py:41-59: same argument names, order and
    defaults) + ``system_xml``; the committed serialized-System fixture (22-atom synthetic peptide: HarmonicBond/Angle,
    PeriodicTorsion, Nonbonded + exceptions, CustomGBForce OBC1, CMMotionRemover) parses back to the tables it was
    written from.  No GPU: only the host side (the kernel handle is created on first call)."""
    import inspect
    import os

    from pita_amd.alp_energy import ALPEnergy, tables_from_openmm_xml
    from tests._synthetic import openmm_system_xml, synthetic_peptide_gb

    ref = [("data_path", inspect._empty), ("pdb_filename", inspect._empty), ("atom_encoding_filename", "atom_types_ecoding.npy"),
           ("dimensionality", 99), ("n_particles", 33), ("spatial_dim", 3), ("device", "cpu"), ("plot_samples_epoch_period", 5),
           ("plotting_buffer_sample_size", 512), ("data_normalization_factor", 1.0), ("is_molecule", True),
           ("temperature", 1.0), ("should_normalize", True), ("should_remove_mean", False), ("device_index", 0),
           ("debug_train_on_test", False), ("energy_batch_size", 10000)]
    sig = list(inspect.signature(ALPEnergy.__init__).parameters.values())[1:]
    assert [p.name for p in sig[:len(ref)]] == [r[0] for r in ref] and sig[len(ref)].name == "system_xml"
    for p, (name, default) in zip(sig, ref):
        if name in ("data_path", "pdb_filename", "device"):
            continue  # optional here (the sampling path never reads the first two); the device defaults to the GPU
        assert p.default == default, name
    path = os.path.join(os.path.dirname(__file__), "golden", "synthetic_peptide22_system.xml")
    t, _ = synthetic_peptide_gb(22)
    assert open(path).read() == openmm_system_xml(t)  # the fixture is exactly what the committed writer produces
    tt, opts = tables_from_openmm_xml(path)
    assert opts == {"cutoff": 2.0, "rf_dielectric": 78.3, "gb_solute_dielectric": 1.0, "gb_solvent_dielectric": 78.5}
    for k, v in t.items():
        np.testing.assert_allclose(np.asarray(tt[k], dtype=np.float64).reshape(-1), np.asarray(v, dtype=np.float64).reshape(-1),
                                   rtol=1e-15, atol=0)
    e = ALPEnergy(data_path="unused", pdb_filename="A_capped.pdb", dimensionality=66, n_particles=22, temperature=300.0,
                  data_normalization_factor=0.1640, system_xml=path)
    assert (e.n_particles, e.n_spatial_dim, e.is_molecule, e.length_scale, e.energy_batch_size) == (22, 3, True, 0.1640, 10000)
    assert abs(e.kT - 8.314462618e-3 * 300.0) < 1e-12 and e.cutoff == 2.0
    with np.testing.assert_raises(pita_amd._lib.PitaHipError):
        ALPEnergy(data_path="x", pdb_filename="A_capped.pdb", dimensionality=66, n_particles=22)  # no OpenMM here
    with np.testing.assert_raises(ValueError):
        ALPEnergy(data_path="x", pdb_filename="y", dimensionality=99, n_particles=33, system_xml=path)


def test_committed_bench_lines_follow_survey_8d():
    """The committed bench lines of the current and the previous round (profiles/r06_bench_<config>.json, r05_...) can be
    recomputed from their own fields by SURVEY 8(d)'s formulas: roofline.achieved = algorithmic flops per launch / launch
    time, frac = achieved / peak, value = walkers x steps / (steps x ms_per_step), PMC traffic >= algorithmic bytes; the
    rocprofv3 kernel-stats summary of the same command (profiles/<round>_kernel_stats_<config>.csv) agrees with the
    HIP-event launch time; and the round-5 lines carry the end-to-end legs (whole integrate_sde, not debiased over the
    1 000-step grid and the reference's default regime) and say which steps of the grid the timed launches ran."""
    import csv
    import json
    import os

    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    alg = {"lj13": 4196608.0, "dw4": None, "aldp22": None, "lj55": None}  # LJ13: 2 098 304 MAC x 2
    for rnd, cfg in (("r05", "lj13"), ("r05", "dw4"), ("r05", "aldp22"), ("r05", "lj55"), ("r06", "lj13"), ("r06", "dw4"),
                     ("r06", "aldp22"), ("r06", "lj55")):
        path = os.path.join(prof, f"{rnd}_bench_{cfg}.json")
        line = json.loads([ln for ln in open(path) if ln.startswith("{")][-1])
        r = line["roofline"]
        B, c = line["config"]["walkers_per_gpu"], r["steps_per_launch"]
        if alg[cfg] is not None:
            assert abs(r["algorithmic_flop_per_walker_step"] - alg[cfg]) < 1
        want = r["algorithmic_flop_per_walker_step"] * B * c / (r["ms_per_launch"] * 1e-3) / 1e12
        assert abs(r["achieved"] - want) < 1e-6 * want and r["peak"] == 2500.0 and r["unit"] == "TFLOP/s"
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] <= r["frac_executed"] <= 1
        assert abs(line["value"] - line["n_gpus"] * B * line["steps"] / (line["ms_per_step"] * line["steps"] * 1e-3)) < 1e-6 * line["value"]
        assert line["unit"] == "walker-steps/s" and line["higher_is_better"] and line["scaling"] == "weak"
        if r["traffic"] is not None:
            assert r["traffic"] >= r["algorithmic_bytes_per_launch"] == 2 * B * line["config"]["walkers_per_gpu"] // B * 0 + r["algorithmic_bytes_per_launch"]
        cb = line["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
        if rnd >= "r06":  # core count AND CPU model; the reference's own batch sizes beside the metric's
            assert cb["cpu_model"] and cb["logical_cpus"] >= cb["cores"]
            if cfg == "lj13":
                sb = line["small_batch"]["sizes"]
                assert set(sb) == {"512", "2048", "5000", "16384"}
                for Bs, legs in sb.items():
                    m = legs["sampler_mapping"]
                    assert 0 < m["wave_slot_fill"] <= 1 and m["waves"] <= m["resident_wave_slots"]
                    for leg in ("not_debiased", "default_regime"):
                        e = legs[leg]
                        assert e["finite"] and e["walkers"] == int(Bs) and 0 < e["frac_of_full_batch"] <= 1.05
                        assert abs(e["value"] - e["walkers"] * e["steps"] / e["seconds"]) < 1e-6 * e["value"]
                # the review's bar: the not-debiased 2 048-walker trajectory at >= 25 % of the full-batch rate
                assert sb["2048"]["not_debiased"]["frac_of_full_batch"] >= 0.25
        if rnd >= "r05":
            g = line["steps_of_grid"]
            assert g["grid_steps"] >= 1 and 0 <= g["first"] < g["grid_steps"] and 0 <= g["last"] < g["grid_steps"]
            assert "default_regime" in line["e2e"] or cfg != "lj13"  # (the regime's MALA needs a target with forces)
            for leg in ("not_debiased", "default_regime"):
                if leg not in line["e2e"]:
                    continue
                e = line["e2e"][leg]
                assert e["finite"] and e["walkers"] == B and e["unit"] == "walker-steps/s"
                assert abs(e["value"] - e["walkers"] * e["steps"] / e["seconds"]) < 1e-6 * e["value"]
                assert abs(e["ms_per_step"] - 1e3 * e["seconds"] / e["steps"]) < 1e-9 * e["ms_per_step"] + 1e-12
            # the whole integrator is never faster than the bare launches it is made of; for the 13-particle headline it is
            # within 10 % of them (DW4's 0.17 ms steps show the host side of a launch, the others sit in between)
            lo = 0.9 if cfg == "lj13" else 0.5
            assert lo * line["value"] < line["e2e"]["not_debiased"]["value"] <= 1.02 * line["value"], cfg
            if "default_regime" in line["e2e"]:
                assert line["e2e"]["default_regime"]["value"] < line["e2e"]["not_debiased"]["value"]
        # rocprofv3 --kernel-trace --stats of the same command (200 + 100 steps in 100-step launches)
        rows = list(csv.DictReader(open(os.path.join(prof, f"{rnd}_kernel_stats_{cfg}.csv"))))
        samp = [x for x in rows if "egnn_kernel" in x["Name"] and ", 2, true," in x["Name"]]
        assert samp, cfg
        avg_ms = float(max(samp, key=lambda x: float(x["TotalDurationNs"]))["AverageNs"]) * 1e-6
        under = json.loads([ln for ln in open(os.path.join(prof, f"{rnd}_bench_under_rocprof_{cfg}.json")) if ln.startswith("{")][-1])
        ev_ms = under["roofline"]["ms_per_launch"]
        assert abs(avg_ms - ev_ms) < 0.03 * ev_ms, (cfg, avg_ms, ev_ms)


def test_count_distinct_parents_without_host_unique():
    """num_unique_idxs (sde_integration.py:295: len(np.unique(choice))) from the structure of systematic-resampling ids --
    non-decreasing up to the cyclic rotation by the event's uniform -- as one device reduction: equal to np.unique on
    reference-shaped id vectors, including the constant vector and a run that wraps around the end."""
    import numpy as np
    import torch

    from oracle import pita_oracle as O
    from pita_amd.sde_integration import _count_distinct, _host_counts

    gen = torch.Generator().manual_seed(3)
    cases = [torch.zeros(17, dtype=torch.int64), torch.arange(9), torch.tensor([3, 3, 4, 7, 7, 0, 0, 3]),
             torch.tensor([5, 5, 5, 1, 1, 5])]
    for B in (1, 2, 64, 1000):
        for spread in (0.1, 3.0, 30.0):
            logits = torch.randn(B, generator=gen) * spread
            for u0 in (0.0, 0.37, 0.999):
                cases.append(torch.from_numpy(O.sample_cat_sys(logits, u0)))
    counts = [_count_distinct(c) for c in cases]
    assert all(isinstance(c, torch.Tensor) and c.dim() == 0 for c in counts)
    got = _host_counts([7] + counts)  # python ints pass through, device scalars are fetched in one transfer
    assert got[0] == 7 and got[1:] == [len(np.unique(c.numpy())) for c in cases]


def test_step_table_is_cached_by_value_not_by_object():
    """build_step_table keeps the last few tables keyed by the VALUES that determine them (both schedules' class and
    parameters, the time grid, dt, diffusion scale, beta): an equal schedule object gets an equal table without the
    per-step host loop, any changed parameter a fresh one, and writing into a returned table does not reach the cache."""
    import torch

    import pita_amd
    from pita_amd import sde_integration as si

    si._STEP_TABLES.clear()
    times = torch.linspace(1.0, 0.0, 41)[:-1]
    mk = lambda smin=0.05: pita_amd.ElucidatingNoiseSchedule(sigma_min=smin, sigma_max=80.0, rho=7)
    gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
    fresh = si._build_step_table(mk(), gam, times, 1 / 40, 1.0, 1.0)
    a = si.build_step_table(mk(), gam, times, 1 / 40, 1.0, 1.0)
    assert len(si._STEP_TABLES) == 1 and torch.equal(a, fresh)
    a[0, 0] = 123.0  # the caller's copy
    b = si.build_step_table(mk(), pita_amd.ConstantAnnealingFactorSchedule(4 / 3), times.clone(), 1 / 40, 1.0, 1.0)
    assert len(si._STEP_TABLES) == 1 and torch.equal(b, fresh)
    for other in (si.build_step_table(mk(0.01), gam, times, 1 / 40, 1.0, 1.0),
                  si.build_step_table(mk(), pita_amd.ConstantAnnealingFactorSchedule(1.5), times, 1 / 40, 1.0, 1.0),
                  si.build_step_table(mk(), gam, times, 1 / 40, 1.0, 1.3),
                  si.build_step_table(mk(), gam, torch.linspace(0.5, 0.0, 41)[:-1], 0.5 / 40, 1.0, 1.0)):
        assert not torch.equal(other, fresh)
    assert len(si._STEP_TABLES) == 5
    for k in range(12):  # bounded
        si.build_step_table(mk(0.02 + 0.001 * k), gam, times, 1 / 40, 1.0, 1.0)
    assert len(si._STEP_TABLES) <= 8

    class Odd:  # a schedule that carries a tensor is not cached (its value is not a plain key)
        def __init__(self):
            self.w = torch.ones(2)

        def h(self, t):
            return t + 1.0

        def g(self, t):
            return t * 0 + 1.0

    n = len(si._STEP_TABLES)
    si.build_step_table(Odd(), gam, times, 1 / 40, 1.0, 1.0)
    assert len(si._STEP_TABLES) == n
