"""Energy view of a score backbone, with the interface of pita/src/models/components/energy_net.py.

* ``forward_energy`` -- E_theta(h, x) (energy_net.py:14-49): one backbone forward on the EDM-scaled input
  (``pita_edm_scale_input``), then the per-walker reduction kernel ``pita_energy_theta``.
* ``forward`` -- grad_x E_theta (autograd in the reference, :51-62): ONE reverse-mode launch of the HIP EGNN
  (``EGNN_dynamics.vjp``): grad E = ((1 + c_s) x - D - J_x D^T x) / h; backbones with a forward-mode derivative only
  (``EGNN_dynamics_AD2_cat.jvp``) assemble J_x D^T x from one launch per direction.
"""
import torch
from torch import nn

from . import _lib
from .score_net import _Preconditioned


class EnergyNet(nn.Module):
    def __init__(self, score_net: nn.Module, precondition_beta=False):
        super().__init__()
        self.net = score_net  # registered as ``net`` (not ``model``): state_dict keys match the reference's
        self.precondition_beta = precondition_beta

    _batch = staticmethod(_Preconditioned._batch)

    def reinitialize(self, score_net: nn.Module):
        self.net = score_net

    def forward_energy(self, ht, xt, beta, pin=False, energy_function=None, t=None):
        x = _lib.dev_tensor(xt, "xt")
        B, D = x.shape
        h = self._batch(ht, B, x.device)
        b = self._batch(beta, B, x.device)
        L, st = _lib.lib(), _lib.stream_ptr(x.device)
        xin, cn = torch.empty_like(x), torch.empty(B, device=x.device)
        _lib.check(L.pita_edm_scale_input(h.data_ptr(), x.data_ptr(), xin.data_ptr(), cn.data_ptr(), B, D, st),
                   "pita_edm_scale_input")
        F = _lib.dev_tensor(self.net(cn, xin, b), "backbone output")
        E = torch.empty(B, device=x.device)
        _lib.check(L.pita_energy_theta(h.data_ptr(), x.data_ptr(), F.data_ptr(),
                                       b.data_ptr() if self.precondition_beta else None, E.data_ptr(), B, D, st),
                   "pita_energy_theta")
        if pin:  # energy_net.py:43-48: blend with the (clamped) target energy near t = 0
            if t is None or energy_function is None:
                raise ValueError("pin=True needs t and energy_function")
            U0 = torch.clamp(-energy_function(x), max=1e3, min=-1e3)
            w = (1 - t) ** 3
            return w * U0 + (1 - w) * E
        return E

    def forward(self, ht, xt, beta, pin=False, t=None, energy_function=None):
        if pin or self.precondition_beta or not (hasattr(self.net, "vjp") or hasattr(self.net, "jvp")):
            raise NotImplementedError("EnergyNet.forward: needs a HIP EGNN backbone, pin=False, precondition_beta=False")
        x = _lib.dev_tensor(xt, "xt")
        h = self._batch(ht, x.shape[0], x.device)
        if hasattr(self.net, "vjp"):
            Dx, jtx = self.net.vjp(h, x, beta)
        else:  # forward-mode backbone (EGNN_dynamics_AD2_cat): (J^T x)_k = <x, J e_k>, one launch per direction
            jtx, Dx = torch.empty_like(x), None
            for k in range(x.shape[1]):
                out, _ = self.net.jvp(h, x, beta, direction=k, want_primal=(k == 0), want_tangent=False, dot_out=jtx,
                                      dot_col=k)
                Dx = out if k == 0 else Dx
        return (((1 + 1 / (1 + h))[:, None] * x - Dx) - jtx) / h[:, None]

    def denoiser(self, h_t, x_t, beta):
        return x_t - h_t[:, None] * self.forward(h_t, x_t, beta)
