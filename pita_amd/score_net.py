"""EDM-preconditioned score / denoiser wrappers with the interface of pita/src/models/components/score_net.py.

Dispatch: a backbone that exposes ``edm`` (the HIP EGNN) evaluates preconditioning and network in ONE kernel launch;
any other backbone module (``forward(t, x, beta)``, e.g. the HIP MLP or a user plug-in) is wrapped by two small HIP
kernels -- ``pita_edm_scale_input`` before it and ``pita_edm_combine`` after it (score_net.py:26-38).
"""
import torch
from torch import nn

from . import _lib


def edm_coefficients(h_t):
    """(c_s, c_in, c_out, c_noise) as device tensors: score_net.py:26-29.  Host-side helper for callers that need the
    coefficients themselves (EnergyNet.forward_energy); the sampler kernels compute them in-kernel."""
    inv = torch.reciprocal(1 + h_t)
    c_in = inv.sqrt()
    return inv, c_in, h_t.sqrt() * c_in, torch.log(h_t) / 8


class _Preconditioned(nn.Module):
    """Shared plumbing: batch-shaped h / beta device tensors and the two wrapper kernels."""

    def __init__(self, model: nn.Module):
        super().__init__()
        self.model = model

    def reinitialize(self, model):
        self.model = model

    @staticmethod
    def _batch(v, B, device):
        if not isinstance(v, torch.Tensor):
            return torch.full((B,), float(v), device=device, dtype=torch.float32)
        return _lib.dev_tensor(v.to(device), "per-walker scalar").reshape(-1).expand(B).contiguous()

    def _wrapped(self, h_t, x_t, beta, want_D, want_score, precondition_beta):
        x = _lib.dev_tensor(x_t, "x_t")
        B, D = x.shape
        h = self._batch(h_t, B, x.device)
        b = self._batch(beta, B, x.device)
        L, st = _lib.lib(), _lib.stream_ptr(x.device)
        xin, cn = torch.empty_like(x), torch.empty(B, device=x.device)
        _lib.check(L.pita_edm_scale_input(h.data_ptr(), x.data_ptr(), xin.data_ptr(), cn.data_ptr(), B, D, st),
                   "pita_edm_scale_input")
        F = _lib.dev_tensor(self.model.forward(cn, xin, b), "backbone output")
        Dn = torch.empty_like(x) if want_D else None
        sc = torch.empty_like(x) if want_score else None
        _lib.check(L.pita_edm_combine(h.data_ptr(), x.data_ptr(), F.data_ptr(), b.data_ptr() if precondition_beta else None,
                                      _lib.ptr(Dn), _lib.ptr(sc), B, D, st), "pita_edm_combine")
        return Dn, sc


class ScoreNet(_Preconditioned):
    def __init__(self, model: nn.Module, precondition_beta=False):
        super().__init__(model)
        self.precondition_beta = precondition_beta

    @property
    def _one_launch(self):
        return hasattr(self.model, "edm") and not self.precondition_beta

    def forward(self, h_t, x_t, beta):
        """score s_theta = (D_theta - x)/h."""
        if self._one_launch:
            return self.model.edm(2, h_t, x_t, beta)
        return self._wrapped(h_t, x_t, beta, False, True, self.precondition_beta)[1]

    def denoiser(self, h_t, x_t, beta, return_score=False):
        if self._one_launch and not return_score:
            return self.model.edm(1, h_t, x_t, beta)
        Dn, sc = self._wrapped(h_t, x_t, beta, True, return_score, self.precondition_beta)
        return (Dn, sc) if return_score else Dn


class FlowNet(_Preconditioned):
    """No preconditioning: the backbone output is used as is (score_net.py:49-67)."""

    def forward(self, h_t, x_t, beta):
        return self.model.forward(h_t, x_t, beta)

    def denoiser(self, h_t, x_t, beta, return_score=False):
        return self.model.forward(h_t, x_t, beta)
