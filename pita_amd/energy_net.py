"""Energy view of the score backbone (mirror of pita/src/models/components/energy_net.py).

``forward_energy`` (E_theta built from <F_theta(c_in x), c_in x>, :14-49) needs only backbone
FORWARDS.  ``forward`` (grad_x E_theta, autograd in the reference :51-62) is one reverse-mode launch of the HIP
backbone (``EGNN_dynamics.vjp``).
"""
from typing import Optional

import torch
from torch import nn

from .score_net import edm_coefficients


class EnergyNet(nn.Module):
    def __init__(self, score_net: nn.Module, precondition_beta: Optional[bool] = False):
        super().__init__()
        self.net = score_net
        self.precondition_beta = precondition_beta

    def forward_energy(self, ht, xt, beta, pin=False, energy_function=None, t=None):
        beta = beta * torch.ones(xt.shape[0], device=xt.device)
        c_s, c_in, c_out, c_noise = edm_coefficients(ht)
        xs = c_in[:, None] * xt
        U_theta = torch.sum(self.net(c_noise, xs, beta) * xs, dim=1)
        E = (1 - c_s) / (2 * ht) * torch.linalg.norm(xt, dim=-1) ** 2 - c_out / (c_in * ht) * U_theta
        if self.precondition_beta:
            E = E * beta
        if pin:  # :43-48
            assert t is not None and energy_function is not None
            U0 = torch.clamp(-energy_function(xt), max=1e3, min=-1e3)
            return (1 - t) ** 3 * U0 + (1 - (1 - t) ** 3) * E
        return E

    def forward(self, ht, xt, beta, pin=False, t=None, energy_function=None):
        """grad_x E_theta = ((1 + c_s) x - D - J_x D^T x)/h with D the denoiser of this backbone."""
        if pin or self.precondition_beta or not hasattr(self.net, "vjp"):
            raise NotImplementedError("EnergyNet.forward: needs the HIP EGNN backbone, pin=False, precondition_beta=False")
        Dx, jtx = self.net.vjp(ht, xt, beta)
        c_s = 1 / (1 + ht)
        return ((1 + c_s)[:, None] * xt - Dx - jtx) / ht[:, None]

    def denoiser(self, h_t, x_t, beta):
        return x_t - h_t[:, None] * self.forward(h_t, x_t, beta)

    def reinitialize(self, score_net: nn.Module):
        self.net = score_net
