"""Weights and sample I/O at the edges of the sampling path.

* ``load_reference_checkpoint`` maps a PITA Lightning checkpoint (``energyTempModule.state_dict()``) onto a
  ``(ScoreNet, EnergyNet)`` pair built from pita_amd backbones.  Key layout of the reference
  (pita/src/models/energytemp_module.py:94-111, components/ema.py:6-22): the score net is ``EMA(ScoreNet(h_theta))``,
  so raw weights live under ``score_net.model.model.<backbone key>`` and EMA weights under
  ``score_net.shadow_params.<i>`` (one entry per trainable parameter, in ``parameters()`` order); the energy net is
  ``EMA(EnergyNet(deepcopy(h_theta)))`` -> ``energy_net.model.net.<backbone key>`` / ``energy_net.shadow_params.<i>``.
  With ``ema_decay == 0`` the nets are NOT wrapped (:107-109): raw weights are then under ``score_net.model.<key>`` /
  ``energy_net.net.<key>`` and there are no shadow parameters; ``h_theta.<key>`` (the backbone registered on the
  module itself, :94) always holds the score backbone's raw weights.  All three layouts are recognised.
  ``strict_loading = False`` in the reference (:57), so missing / extra keys are tolerated and reported -- but a
  checkpoint from which NOT ONE backbone tensor can be matched raises instead of silently sampling from random weights.
* ``save_samples`` / ``load_samples`` use the reference's ``torch.save`` of a [B, D] tensor
  (``samples_temperature_*.pt``, energytemp_module.py:1040-1041); ``load_dataset`` reads the
  ``{train,val,test}_split_<name><n>-10000.npy`` files (base_molecule_energy_function.py:48-94).
"""
import os

import numpy as np
import torch


def _apply(backbone, raw_prefixes, shadow_prefix, state, use_ema, report):
    names = [n for n, p in backbone.named_parameters() if p.requires_grad]
    own = backbone.state_dict()
    # the layout (EMA-wrapped, plain, or the module-level h_theta copy) is the prefix under which most keys are found
    raw_prefix = max(raw_prefixes, key=lambda pre: sum((pre + k) in state for k in own))
    new = {}
    for k in own:
        src = raw_prefix + k
        if src in state:
            new[k] = state[src]
        else:
            report["missing"].append(src)
    if not new:
        raise KeyError(f"checkpoint holds no backbone weights under any of {list(raw_prefixes)} "
                       f"(first keys: {list(state)[:5]})")
    report["layout"].append(raw_prefix.rstrip("."))
    if use_ema:
        shadows = [state.get(f"{shadow_prefix}{i}") for i in range(len(names))]
        if all(s is not None for s in shadows):
            for n, s in zip(names, shadows):
                new[n] = s
            report["ema"].append(shadow_prefix.rstrip("."))
        else:
            report["missing"].append(shadow_prefix + "*")
    backbone.load_state_dict({k: torch.as_tensor(v).to(own[k].dtype).reshape(own[k].shape) for k, v in new.items()},
                             strict=False)


def load_reference_checkpoint(ckpt, score_net=None, energy_net=None, use_ema=True):
    """ckpt: path to a Lightning ``.ckpt`` / ``torch.save``d dict, or an already loaded state_dict.
    Returns a report dict {"missing": [...], "ema": [...], "unused": [...]}."""
    if isinstance(ckpt, (str, os.PathLike)):
        ckpt = torch.load(ckpt, map_location="cpu", weights_only=False)
    state = ckpt.get("state_dict", ckpt)
    report = {"missing": [], "ema": [], "unused": [], "layout": []}
    if score_net is not None:
        _apply(score_net.model, ("score_net.model.model.", "score_net.model.", "h_theta."), "score_net.shadow_params.",
               state, use_ema, report)
    if energy_net is not None:
        _apply(energy_net.net, ("energy_net.model.net.", "energy_net.net."), "energy_net.shadow_params.", state, use_ema,
               report)
    used = ("score_net.", "energy_net.", "h_theta.")
    report["unused"] = [k for k in state if not k.startswith(used)]
    return report


def save_samples(samples: torch.Tensor, path: str):
    torch.save(samples.detach().cpu(), path)


def load_samples(path: str, device="cuda") -> torch.Tensor:
    return torch.load(path, map_location="cpu").to(device=device, dtype=torch.float32)


def load_dataset(data_path: str, data_name: str, n_particles: int, temperature: float, split: str = "test", device="cuda"):
    fmt = "{:0.1f}" if "LJ" in data_name else "{:0.2f}"
    f = f"{data_path}{data_name}{n_particles}_temp_{fmt.format(temperature)}/{split}_split_{data_name}{n_particles}-10000.npy"
    return torch.tensor(np.load(f, allow_pickle=True), device=device, dtype=torch.float32)
