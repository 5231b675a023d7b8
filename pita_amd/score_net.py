"""EDM-preconditioned score / denoiser wrappers (mirror of pita/src/models/components/score_net.py).

``ScoreNet(model).forward(h_t, x_t, beta)`` = (D_theta - x)/h with
D_theta = c_s x + c_out F(c_noise, c_in x, beta).  When ``model`` is the HIP EGNN the whole
expression runs inside one kernel launch (``model.edm``); any other ``nn.Module`` backbone
(``forward(t, x, beta)``) is composed with device tensor ops -- the plug-in path.
"""
from typing import Optional

import torch
from torch import nn


def edm_coefficients(h_t):
    """c_s, c_in, c_out, c_noise (score_net.py:26-29)."""
    c_s = 1 / (1 + h_t)
    c_in = 1 / (1 + h_t) ** 0.5
    c_out = h_t**0.5 * c_in
    c_noise = (1 / 8) * torch.log(h_t)
    return c_s, c_in, c_out, c_noise


class ScoreNet(nn.Module):
    def __init__(self, model: nn.Module, precondition_beta: Optional[bool] = False):
        super().__init__()
        self.model = model
        self.precondition_beta = precondition_beta

    def _fused(self):
        return hasattr(self.model, "edm") and not self.precondition_beta

    def forward(self, h_t, x_t, beta):
        if self._fused():
            return self.model.edm(2, h_t, x_t, beta)
        return (self.denoiser(h_t, x_t, beta) - x_t) / h_t[:, None]

    def denoiser(self, h_t, x_t, beta, return_score=False):
        if self._fused() and not return_score:
            return self.model.edm(1, h_t, x_t, beta)
        beta = beta * torch.ones(x_t.shape[0], device=x_t.device)
        c_s, c_in, c_out, c_noise = edm_coefficients(h_t)
        D = c_s[:, None] * x_t + c_out[:, None] * self.model.forward(c_noise, c_in[:, None] * x_t, beta)
        score = (D - x_t) / h_t[:, None]
        if self.precondition_beta:  # :36-38
            D = D * beta[:, None] + (1 - beta[:, None]) * x_t
            score = score * beta[:, None]
        return (D, score) if return_score else D

    def reinitialize(self, model):
        self.model = model


class FlowNet(nn.Module):  # score_net.py:49-67: raw backbone pass-through
    def __init__(self, model: nn.Module):
        super().__init__()
        self.model = model

    def forward(self, h_t, x_t, beta):
        return self.model.forward(h_t, x_t, beta)

    def denoiser(self, h_t, x_t, beta, return_score=False):
        return self.model.forward(h_t, x_t, beta)
