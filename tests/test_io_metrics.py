"""CPU tests of the I/O and metric helpers at the edges of the path."""
import numpy as np
import torch

from oracle import pita_oracle as O


def test_checkpoint_mapping_raw_and_ema():
    import pita_amd
    from pita_amd import io
    from pita_amd.energy_net import EnergyNet

    mk = lambda: pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, tanh=True, attention=True,
                                        condition_temperature=True)
    torch.manual_seed(1)
    src_s, src_e = mk(), mk()
    # a state_dict laid out like energyTempModule's: EMA(ScoreNet(h)), EMA(EnergyNet(h'))
    state = {"score_net.model.model." + k: v.clone() for k, v in src_s.state_dict().items()}
    state.update({"energy_net.model.net." + k: v.clone() for k, v in src_e.state_dict().items()})
    ema_s = [p.detach() * 0.5 for p in src_s.parameters()]
    state.update({f"score_net.shadow_params.{i}": p for i, p in enumerate(ema_s)})
    state["score_net.num_updates"] = torch.tensor(7)
    state["temperatures"] = torch.tensor([4.0, 3.0])
    torch.manual_seed(2)
    sn, en = pita_amd.ScoreNet(mk()), EnergyNet(mk())
    rep = io.load_reference_checkpoint({"state_dict": state}, sn, en, use_ema=True)
    for p, q in zip(sn.model.parameters(), ema_s):
        assert torch.equal(p, q)  # EMA weights win for the score net
    for (k, p), q in zip(en.net.state_dict().items(), src_e.state_dict().values()):
        assert torch.equal(p, q), k  # no shadow params for the energy net here -> raw weights, reported missing
    assert "energy_net.shadow_params.*" in rep["missing"] and rep["ema"] == ["score_net.shadow_params"]
    assert "temperatures" in rep["unused"]
    rep2 = io.load_reference_checkpoint(state, sn, None, use_ema=False)
    assert torch.equal(next(sn.model.parameters()), next(src_s.parameters())) and rep2["ema"] == []


def test_checkpoint_layouts_without_ema_and_empty_checkpoint():
    """ema_decay = 0 checkpoints (energytemp_module.py:107-109: nets not wrapped) and the module-level h_theta copy are
    recognised; a checkpoint without a single backbone tensor raises instead of leaving random weights in place."""
    import pytest

    import pita_amd
    from pita_amd import io
    from pita_amd.energy_net import EnergyNet

    mk = lambda: pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, tanh=True, attention=True,
                                        condition_temperature=True)
    torch.manual_seed(3)
    src_s, src_e = mk(), mk()
    plain = {"score_net.model." + k: v.clone() for k, v in src_s.state_dict().items()}
    plain.update({"energy_net.net." + k: v.clone() for k, v in src_e.state_dict().items()})
    plain.update({"h_theta." + k: v.clone() for k, v in src_s.state_dict().items()})
    torch.manual_seed(4)
    sn, en = pita_amd.ScoreNet(mk()), EnergyNet(mk())
    rep = io.load_reference_checkpoint({"state_dict": plain}, sn, en, use_ema=True)
    assert rep["layout"] == ["score_net.model", "energy_net.net"] and rep["ema"] == []
    for (k, p), q in zip(sn.model.state_dict().items(), src_s.state_dict().values()):
        assert torch.equal(p, q), k
    for (k, p), q in zip(en.net.state_dict().items(), src_e.state_dict().values()):
        assert torch.equal(p, q), k
    only_h = {"h_theta." + k: v.clone() for k, v in src_e.state_dict().items()}
    rep = io.load_reference_checkpoint(only_h, sn, None, use_ema=False)
    assert rep["layout"] == ["h_theta"] and torch.equal(next(sn.model.parameters()), next(src_e.parameters()))
    with pytest.raises(KeyError):
        io.load_reference_checkpoint({"state_dict": {"optimizer.foo": torch.zeros(1)}}, sn, en)


def test_w2_matches_oracle():
    """N4: POT UNPINNED.  The reference computes these distances with POT's ``ot.emd2_1d``
    (distribution_distances.py:16-17), which is not installed here and not in the reference tree: W2 is checked against
    the oracle's own restatement of the 1-D quantile coupling (closed form for equal sample counts), W1 additionally
    against an independent third-party implementation (scipy.stats.wasserstein_distance)."""
    from scipy.stats import wasserstein_distance

    from pita_amd import metrics

    rng = np.random.default_rng(0)
    a, b = rng.normal(size=2000), rng.normal(1.0, 2.0, size=2000)
    d = metrics.energy_distances(torch.tensor(b), torch.tensor(a), prefix="t")
    assert abs(d["t/energy_w2"] - O.w2_1d(a, b)) < 1e-9
    assert abs(metrics._w_1d(torch.tensor(a[:500]), torch.tensor(b), 2) ** 0.5 - O.w2_1d(a[:500], b)) < 1e-9
    assert d["t/num_cropped"] == 0
    assert abs(d["t/energy_w1"] - wasserstein_distance(a, b)) < 1e-9
    assert abs(metrics._w_1d(torch.tensor(a[:500]), torch.tensor(b), 1) - wasserstein_distance(a[:500], b)) < 1e-9


def test_openmm_system_xml_tables_roundtrip(tmp_path):
    """tables_from_openmm_xml on a hand-written file in OpenMM's XmlSerializer layout (CPU: parsing only)."""
    import numpy as np

    from pita_amd.alp_energy import tables_from_openmm_xml

    xml = """<?xml version="1.0" ?>
<System openmmVersion="8.1" type="System" version="1">
 <PeriodicBoxVectors><A x="2" y="0" z="0"/><B x="0" y="2" z="0"/><C x="0" y="0" z="2"/></PeriodicBoxVectors>
 <Particles><Particle mass="12.01"/><Particle mass="1.008"/><Particle mass="14.01"/><Particle mass="16"/></Particles>
 <Constraints/>
 <Forces>
  <Force forceGroup="0" name="HarmonicBondForce" type="HarmonicBondForce" usesPeriodic="0" version="2">
   <Bonds><Bond d=".109" k="284512" p1="0" p2="1"/><Bond d=".1335" k="410031" p1="0" p2="2"/></Bonds></Force>
  <Force forceGroup="0" type="HarmonicAngleForce" usesPeriodic="0" version="2">
   <Angles><Angle a="2.0944" k="418.4" p1="1" p2="0" p3="2"/></Angles></Force>
  <Force forceGroup="0" type="PeriodicTorsionForce" usesPeriodic="0" version="2">
   <Torsions><Torsion k="10.46" p1="1" p2="0" p3="2" p4="3" periodicity="2" phase="3.14159265"/></Torsions></Force>
  <Force alpha="0" cutoff="2" dispersionCorrection="1" forceGroup="0" method="1" rfDielectric="1" type="NonbondedForce" version="4">
   <GlobalParameters/><ParticleOffsets/><ExceptionOffsets/>
   <Particles><Particle eps=".4577" q=".5973" sig=".33997"/><Particle eps=".0657" q=".1123" sig=".26495"/>
    <Particle eps=".7113" q="-.4157" sig=".325"/><Particle eps=".8786" q="-.5679" sig=".29599"/></Particles>
   <Exceptions><Exception eps="0" p1="0" p2="1" q="0" sig="1"/><Exception eps=".1" p1="1" p2="3" q="-.053" sig=".28"/></Exceptions></Force>
  <Force cutoff="2" forceGroup="0" method="1" soluteDielectric="1" solventDielectric="78.5" surfaceAreaEnergy="2.25936" type="GBSAOBCForce" version="2">
   <Particles><Particle q=".5973" r=".17" scale=".72"/><Particle q=".1123" r=".13" scale=".85"/>
    <Particle q="-.4157" r=".155" scale=".79"/><Particle q="-.5679" r=".15" scale=".85"/></Particles></Force>
  <Force forceGroup="0" frequency="1" type="CMMotionRemover" version="1"/>
 </Forces>
</System>"""
    path = tmp_path / "system.xml"
    path.write_text(xml)
    for src in (str(path), xml):
        t, o = tables_from_openmm_xml(src)
        assert t["bond_idx"].tolist() == [[0, 1], [0, 2]] and np.allclose(t["bond_par"][1], [0.1335, 410031])
        assert t["angle_idx"].tolist() == [[1, 0, 2]] and np.allclose(t["angle_par"][0], [2.0944, 418.4])
        assert t["tors_idx"].tolist() == [[1, 0, 2, 3]] and np.allclose(t["tors_par"][0], [2, 3.14159265, 10.46])
        assert np.allclose(t["charge"], [0.5973, 0.1123, -0.4157, -0.5679]) and np.allclose(t["sigma"][2], 0.325)
        assert t["exc_idx"].tolist() == [[0, 1], [1, 3]] and np.allclose(t["exc_par"][1], [-0.053, 0.28, 0.1])
        assert np.allclose(t["gb_radius"], [0.17, 0.13, 0.155, 0.15]) and np.allclose(t["gb_scale"][0], 0.72)
        assert o["cutoff"] == 2.0 and o["rf_dielectric"] == 1.0 and abs(o["gb_surface_area_factor"] - 28.3919551) < 1e-4
