"""Import-time shims that let the PITA reference's hot-path modules import in the
build container (CPU only, no lightning / hydra / bgflow / wandb / POT installed).

ONLY used by ``tests/golden/make_golden.py`` (which runs in the build container where
``/root/reference`` is mounted).  Nothing here travels into the product or the GPU
tests: the GPU box has no ``/root/reference``.

The only shim that carries arithmetic is ``bgflow``: the reference's LJ energy
(pita/src/energies/lennardjones_energy.py:9-10,125-127) calls
``bgflow.utils.distance_vectors`` / ``distances_from_vectors`` which are NOT in the
reference tree (environment.yaml:56 pins ``git+https://github.com/atong01/bgflow.git``
without a commit).  The two functions below restate bgflow's published
``bgflow/utils/geometry.py`` behaviour: all ordered difference vectors with the
diagonal removed, and ``sqrt(sum(r^2) + eps)`` with ``eps=1e-6``.
"""
import sys
import types

import torch

REF_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package
    sys.modules[name] = m
    return m


class _Anything:
    """Placeholder class/callable for names only referenced in annotations."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, item):
        return _Anything()


def _bgflow_distance_vectors(x, remove_diagonal=True):
    n = x.shape[1]
    r = x.unsqueeze(2).repeat(1, 1, n, 1)
    r = r - r.permute([0, 2, 1, 3])
    if remove_diagonal:
        r = r[:, torch.eye(n, n) == 0].view(-1, n, n - 1, x.shape[2])
    return r


def _bgflow_distances_from_vectors(r, eps=1e-6):
    return (r.pow(2).sum(dim=-1) + eps).sqrt()


class _BgflowEnergy(torch.nn.Module):
    def __init__(self, dim, **kwargs):
        super().__init__()
        if isinstance(dim, int):
            dim = [dim]
        self._event_shapes = [torch.Size(dim)]

    @property
    def event_shape(self):
        return self._event_shapes[0]

    @property
    def dim(self):
        return self._event_shapes[0][0]


def install():
    if "src" in sys.modules and getattr(sys.modules["src"], "_pita_shimmed", False):
        return
    # --- lightning / pytorch_lightning -------------------------------------------------
    class _RZO:
        rank = 0

        def __call__(self, fn):
            return fn

    rzo = _RZO()
    L = _mod("lightning", LightningModule=torch.nn.Module, LightningDataModule=object,
             Callback=object, Trainer=_Anything, seed_everything=lambda *a, **k: None)
    _mod("lightning.pytorch", LightningModule=torch.nn.Module, Callback=object, Trainer=_Anything)
    _mod("lightning.pytorch.loggers", WandbLogger=_Anything, Logger=_Anything)
    _mod("lightning.pytorch.utilities", rank_zero_only=rzo)
    _mod("lightning_utilities")
    _mod("lightning_utilities.core")
    _mod("lightning_utilities.core.rank_zero", rank_zero_only=rzo, rank_prefixed_message=lambda m, r: m)
    _mod("pytorch_lightning", LightningModule=torch.nn.Module)
    _mod("pytorch_lightning.utilities")
    _mod("pytorch_lightning.loggers", WandbLogger=_Anything)
    _mod("pytorch_lightning.utilities.rank_zero", rank_zero_only=rzo)
    # --- hydra / omegaconf / rootutils / wandb / ot -----------------------------------
    _mod("hydra", main=lambda *a, **k: (lambda f: f))
    _mod("hydra.utils", get_original_cwd=lambda: ".", instantiate=_Anything())
    _mod("hydra.core")
    _mod("hydra.core.hydra_config", HydraConfig=_Anything)
    _mod("omegaconf", DictConfig=dict, OmegaConf=_Anything, open_dict=_Anything)
    _mod("rootutils", setup_root=lambda *a, **k: None)
    _mod("wandb", Image=_Anything, log=lambda *a, **k: None)
    _mod("ot", emd2_1d=_Anything())
    _mod("rich")
    _mod("rich.progress", Progress=_Anything)
    _mod("rich.prompt", Prompt=_Anything)
    _mod("rich.syntax", Syntax=_Anything)
    _mod("rich.tree", Tree=_Anything)
    # --- bgflow (carries real arithmetic, see module docstring) -----------------------
    _mod("bgflow", Energy=_BgflowEnergy, MultiDoubleWellPotential=_Anything,
         OpenMMBridge=_Anything, OpenMMEnergy=_Anything)
    _mod("bgflow.utils", distance_vectors=_bgflow_distance_vectors,
         distances_from_vectors=_bgflow_distances_from_vectors)
    # --- the reference's own packages: bare shells so their __init__ (which drags in
    #     hydra/wandb) is not executed ------------------------------------------------
    src = types.ModuleType("src")
    src.__path__ = [REF_ROOT + "/pita/src"]
    src._pita_shimmed = True
    sys.modules["src"] = src
    for sub in ("utils", "models", "models.components", "energies"):
        m = types.ModuleType("src." + sub)
        m.__path__ = [REF_ROOT + "/pita/src/" + sub.replace(".", "/")]
        sys.modules["src." + sub] = m
    # src.utils.logging_utils drags in omegaconf internals: provide the one symbol used
    _mod("src.utils.logging_utils", fig_to_image=lambda *a, **k: None)
    # fab: gmm_energy.py imports ``fab.fab.target_distributions`` while gmm.py imports
    # ``fab.target_distributions`` -- make both resolve to the same directory.
    fab = types.ModuleType("fab")
    fab.__path__ = [REF_ROOT + "/fab/fab", REF_ROOT + "/fab"]
    sys.modules["fab"] = fab
    fabfab = types.ModuleType("fab.fab")
    fabfab.__path__ = [REF_ROOT + "/fab/fab"]
    sys.modules["fab.fab"] = fabfab
    for pre in ("fab", "fab.fab"):
        td = types.ModuleType(pre + ".target_distributions")
        td.__path__ = [REF_ROOT + "/fab/fab/target_distributions"]
        sys.modules[pre + ".target_distributions"] = td
        _mod(pre + ".utils")
        _mod(pre + ".utils.plotting", plot_contours=_Anything(), plot_marginal_pair=_Anything())
        _mod(pre + ".utils.numerical", MC_estimate_true_expectation=_Anything(),
             effective_sample_size_over_p=_Anything(), importance_weighted_expectation=_Anything(),
             quadratic_function=_Anything(), setup_quadratic_function=_Anything())
        _mod(pre + ".types_", LogProbFunc=object)
        _mod(pre + ".target_distributions.base", TargetDistribution=object)
