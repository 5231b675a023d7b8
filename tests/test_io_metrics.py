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


def test_w2_matches_oracle():
    from pita_amd import metrics

    rng = np.random.default_rng(0)
    a, b = rng.normal(size=2000), rng.normal(1.0, 2.0, size=2000)
    d = metrics.energy_distances(torch.tensor(b), torch.tensor(a), prefix="t")
    assert abs(d["t/energy_w2"] - O.w2_1d(a, b)) < 1e-9
    assert abs(metrics._w_1d(torch.tensor(a[:500]), torch.tensor(b), 2) ** 0.5 - O.w2_1d(a[:500], b)) < 1e-9
    assert d["t/num_cropped"] == 0
