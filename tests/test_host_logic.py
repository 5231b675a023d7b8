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
