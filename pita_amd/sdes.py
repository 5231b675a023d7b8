"""Reverse VE-SDE terms (mirror of pita/src/models/components/sdes.py).

``VEReverseSDE.f`` returns the ``SDETerms`` of one step.
* NOT-debiased regime (``debias_inference=False`` -> ``f_not_debiased``, sdes.py:117-128):
  drift_X = gamma(t) s_theta(h(t), x, beta) g(t)^2, drift_A = 0.  The fused HIP sampler does not call ``f`` per
  step in this regime -- ``WeightedSDEIntegrator`` hands whole trajectories to pita_egnn_sampler_run; ``f`` serves
  the per-step (recording / plug-in) path.
* Debiased Feynman-Kac regime (sdes.py:151-239): drift_X = -gamma grad_x E_theta g^2/2 + gamma s_theta g^2/2 and the
  log-weight drift  gamma^2 <-grad E, b> + gamma div b + gamma dE/dt + gamma'(t) E, clamped at its 0.9 quantile.
  The reference obtains grad_x E (autograd), div s (vmap(jacrev)) and dE/dt (autograd through h(t)); here they are
  assembled (pita_fk_assemble, csrc/fk_kernels.hip) from derivatives of the two denoisers computed by HIP kernels:
  ONE reverse-mode launch on the energy net (pita_egnn_vjp, csrc/egnn_vjp_kernel.hip) that returns J_x D^T x and
  <x, dD/dh> from the same sweep, and the exact trace of the score net's Jacobian (pita_egnn_jacobian_trace,
  csrc/egnn_div_kernel.hip: one launch with the primal that writes the primal cache, tangent-only launches that stream
  it), per step; the clamp is pita_quantile_clamp per inference chunk.
"""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class SDETerms:  # sdes.py:34-92
    drift_X: torch.Tensor
    drift_A: torch.Tensor
    divergence_score: Optional[torch.Tensor] = None
    cross_term: Optional[torch.Tensor] = None
    dUt_dt: Optional[torch.Tensor] = None
    diffusion: Optional[torch.Tensor] = None

    _FIELDS = ("drift_X", "drift_A", "divergence_score", "cross_term", "dUt_dt", "diffusion")

    @staticmethod
    def cpu(data):
        return SDETerms(**{f: (getattr(data, f).cpu() if getattr(data, f) is not None else None)
                           for f in SDETerms._FIELDS})

    @staticmethod
    def concatenate(data_list):
        if not data_list:
            raise ValueError("The data_list is empty.")
        first = data_list[0]
        return SDETerms(**{f: (torch.cat([getattr(d, f) for d in data_list], dim=0)
                               if getattr(first, f) is not None else None) for f in SDETerms._FIELDS})


class TermStats:
    """Moments of one ``SDETerms`` field over the GLOBAL walker batch of one step.  The reference returns the full
    [B, D] / [B] tensors of every step as host copies (sde_integration.py:289) and its callers only ever take
    ``.mean()`` and ``.std()`` of them (energytemp_module.py:1132-1143); the HIP integrator reduces on the device
    (pita_egnn_sampler_run / pita_em_step / pita_moments ``stats_out``) and hands back this stand-in, which answers
    the same two calls with 0-dim CPU tensors.  ``record_terms=True`` on the integrator gives the full tensors."""

    __slots__ = ("s", "s2", "n")

    def __init__(self, s, s2, n):
        self.s, self.s2, self.n = float(s), float(s2), int(n)

    def numel(self):
        return self.n

    def mean(self):
        return torch.tensor(self.s / self.n if self.n > 0 else float("nan"), dtype=torch.float32)

    def std(self):  # unbiased, like torch.Tensor.std()
        if self.n < 2:
            return torch.tensor(float("nan"), dtype=torch.float32)
        var = max(self.s2 - self.s * self.s / self.n, 0.0) / (self.n - 1)
        return torch.tensor(var ** 0.5, dtype=torch.float32)

    def cpu(self):
        return self

    def detach(self):
        return self

    def sum(self):
        return torch.tensor(self.s, dtype=torch.float32)

    def var(self):
        return self.std() ** 2

    def __getattr__(self, name):
        # anything else a caller could ask of the reference's full tensor: say what to do instead of an AttributeError
        # on a float (the reference's own callers take mean / std only, energytemp_module.py:938-945,1132-1143)
        raise AttributeError(f"TermStats carries sum, sum of squares and count of an SDETerms field, not the tensor: "
                             f"'{name}' needs WeightedSDEIntegrator(..., record_terms=True)")

    def __repr__(self):
        return f"TermStats(mean={float(self.mean()):.6g}, std={float(self.std()):.6g}, n={self.n})"


def _per_walker(t, x):
    if t.dim() != 0:
        return t.to(x.device)
    if t.device != x.device:  # a host scalar (what the integrator passes): filled on the device, no synchronisation
        return torch.full((x.shape[0],), float(t), device=x.device, dtype=torch.float32)
    return t * torch.ones(x.shape[0], device=x.device)


class VEReverseSDE:
    def __init__(self, noise_schedule, energy_net=None, score_net=None, cdf=None, pin_energy=False,
                 debias_inference=True):
        self.noise_schedule = noise_schedule
        self.energy_net, self.score_net = energy_net, score_net
        self.pin_energy = pin_energy
        self.debias_inference = debias_inference
        # accepted for interface compatibility and ignored: the exact divergence comes from the backbone's own
        # multi-direction kernel (EGNN_dynamics.jacobian_trace), not from a torch.compile'd vmap(jacrev) closure
        self.compiled_divergence_fn = cdf
        self.trainer = None  # set from outside by the reference (energytemp_module.py:1295)

    def g(self, t):
        return self.noise_schedule.g(t)

    def f_not_debiased(self, t, x, beta, gamma_energy):
        assert self.score_net is not None
        ht = self.noise_schedule.h(t)
        score = self.score_net(ht, x, beta)
        drift_X = gamma_energy * (score * self.g(t).pow(2).unsqueeze(-1))
        return SDETerms(drift_X=drift_X.detach(), drift_A=torch.zeros(x.shape[0], device=x.device))

    def f(self, t, x, beta, gamma_energy_schedule, gamma_score, energy_function, resampling_interval=-1,
          clamp_chunk=None):
        """``clamp_chunk`` (extension): evaluate the whole batch in one set of launches but apply the 0.9-quantile
        clamp of the weight drift per chunk of that many walkers -- what the reference gets by calling ``f`` once per
        inference chunk (sde_integration.py:312-343, sdes.py:230), without its per-chunk launch overhead."""
        if self.debias_inference and isinstance(t, torch.Tensor) and t.dim() == 0 and t.device.type != "cpu":
            # a 0-dim DEVICE step time is the same thing as the integrator's host scalar: one read up front, then the
            # scalar path -- the schedule is evaluated by the same (host) code either way, so both give the same bits
            t = t.detach().cpu()
        gamma_energy = gamma_energy_schedule.gamma(t)  # gamma_score is overwritten by it (sdes.py:142-143)
        if not self.debias_inference:
            t = _per_walker(t, x)
            if isinstance(gamma_energy, torch.Tensor):
                gamma_energy = gamma_energy.to(x.device)
            return self.f_not_debiased(t, x, beta, gamma_energy)
        # gamma and dgamma/dt enter the assembly kernel as scalars, taken from the (host) step time
        dgamma = gamma_energy_schedule.dgamma_dt(t) if t.dim() == 0 else None
        # a host scalar also gives h(t), g(t)^2 and dh/dt as three fills instead of a dozen element-wise launches on [B]
        sched_scalars = None
        if t.dim() == 0 and t.device.type == "cpu":
            t1 = t.detach().reshape(1).to(torch.float32)
            sched_scalars = (float(self.noise_schedule.h(t1)[0]), float(self.g(t1).pow(2)[0]), float(self._dh_dt(t1)[0]))
        t = _per_walker(t, x)
        return self.f_debiased(t, x, beta, gamma_energy, gamma_energy_schedule, clamp_chunk, energy_function, dgamma,
                               sched_scalars)

    # ------------------------------------------------------------------ debiased regime (sdes.py:151-239)
    def _denoiser_jacobian_terms(self, model, ht, x, beta, want_h_direction):
        """From dim (+1) launches of pita_egnn_jvp on one backbone: D, trace(J_x D), J_x D^T x and <x, dD/dh>.
        The per-walker reductions happen inside the kernel; only [B] / [B, D] results touch memory."""
        if not hasattr(model, "jvp"):
            raise NotImplementedError(
                "debias_inference=True needs a backbone with a forward-mode derivative (the HIP EGNN_dynamics.jvp)")
        B, D = x.shape
        trace = torch.zeros(B, device=x.device)
        jtx = torch.empty(B, D, device=x.device)
        Dx = None
        for k in range(D):
            out, _ = model.jvp(ht, x, beta, direction=k, want_primal=(k == 0), want_tangent=False, dot_out=jtx,
                               dot_col=k, diag_acc=trace)
            Dx = out if k == 0 else Dx
        dot_h = None
        if want_h_direction:
            dot_h = torch.empty(B, device=x.device)
            model.jvp(ht, x, beta, direction=-1, vh=torch.ones(B, device=x.device), want_primal=False,
                      want_tangent=False, dot_out=dot_h)
        return Dx, trace, jtx, dot_h

    def _energy_gradient_terms(self, model, ht, x, beta):
        """D, J_x D^T x and <x, dD/dh> from ONE reverse-mode launch (pita_egnn_vjp): all that grad_x E_theta and
        dE_theta/dt need.  Backbones without ``vjp`` fall back to dim + 1 forward-mode launches."""
        if not hasattr(model, "vjp"):
            D_E, _, jtx, dot_h = self._denoiser_jacobian_terms(model, ht, x, beta, True)
            return D_E, jtx, dot_h, None
        if getattr(model, "vjp_h_parts", False):  # + the split of <x, dD/dh> that keeps dE/dh free of cancellation
            return model.vjp(ht, x, beta, want_dot_h=True, want_h_parts=True)
        D_E, jtx, dot_h = model.vjp(ht, x, beta, want_dot_h=True)  # <x, dD/dh> rides on the reverse sweep
        return D_E, jtx, dot_h, None

    def _score_divergence_terms(self, model, ht, x, beta):
        """D and trace(J_x D) of the score net's denoiser from the multi-direction divergence kernel (dim / K launches;
        D is a by-product of the first one); backbones without it use dim single-direction JVP launches."""
        if not hasattr(model, "jacobian_trace"):
            D_S, trace, _, _ = self._denoiser_jacobian_terms(model, ht, x, beta, False)
            return D_S, trace
        trace, D_S = model.jacobian_trace(ht, x, beta, want_denoiser=True)
        return D_S, trace

    def _dh_dt(self, t):
        """dh/dt per walker: the schedule's own ``dh_dt`` when it has one, else autograd through ``h`` like the
        reference (sdes.py:218 differentiates U_t through h(t)); never assumed equal to g(t)^2."""
        sched = self.noise_schedule
        if hasattr(sched, "dh_dt"):
            return sched.dh_dt(t)
        with torch.enable_grad():
            tt = t.detach().clone().requires_grad_(True)
            return torch.autograd.grad(sched.h(tt).sum(), tt)[0]

    def f_debiased(self, t, x, beta, gamma_energy, gamma_energy_schedule, clamp_chunk=None, energy_function=None,
                   dgamma_dt=None, sched_scalars=None):
        assert self.energy_net is not None
        if self.score_net is None:
            # sdes.py:204-216 (Laplacian of E_theta by vmap(hessian)).  Unreachable in the reference as shipped: its
            # constructor dereferences score_net.forward when cdf is None (:111-112) and energyTempModule always passes
            # a score net (energytemp_module.py:124-131); needs second-order derivatives of the backbone.
            raise NotImplementedError("debiased HIP path without a score net (Laplacian of E_theta) is not built")
        from . import _lib

        x = _lib.dev_tensor(x, "x")
        B, D = x.shape
        if sched_scalars is not None:  # uniform step time (the integrator's): the schedule evaluated once on the host
            ht, g2, dhdt = (torch.full((B,), v, device=x.device, dtype=torch.float32) for v in sched_scalars)
        else:
            ht = _lib.dev_tensor(self.noise_schedule.h(t), "h(t)").contiguous()
            g2 = _lib.dev_tensor(self.g(t).pow(2), "g(t)^2").contiguous()
            dhdt = _lib.dev_tensor(self._dh_dt(t), "dh/dt").contiguous()
        pb_e = bool(getattr(self.energy_net, "precondition_beta", False))
        pb_s = bool(getattr(self.score_net, "precondition_beta", False))
        beta_b = None
        if pb_e or pb_s:
            from .score_net import _Preconditioned

            beta_b = _Preconditioned._batch(beta, B, x.device)
        pin_w = pin_dw = 0.0
        logp_t = None
        if self.pin_energy:  # energy_net.py:43-48; t is the integrator's scalar step time repeated per walker
            if energy_function is None:
                raise ValueError("pin_energy=True needs the target energy_function")
            tv = t.reshape(-1)
            if bool((tv != tv[0]).any()):
                raise NotImplementedError("pin_energy with per-walker times")
            one_minus = 1 - tv[0].to(torch.float32).cpu()
            pin_w, pin_dw = float(one_minus**3), float(-3 * one_minus**2)
            logp_t = _lib.dev_tensor(energy_function(x), "energy_function(x)").contiguous()
        gamma = float(gamma_energy.reshape(-1)[0]) if isinstance(gamma_energy, torch.Tensor) else float(gamma_energy)
        dg = gamma_energy_schedule.dgamma_dt(t) if dgamma_dt is None else dgamma_dt
        dgamma = float(dg.reshape(-1)[0]) if isinstance(dg, torch.Tensor) else float(dg)
        D_E, jtx_E, dot_h, h_parts = self._energy_gradient_terms(self.energy_net.net, ht, x, beta)
        D_S, trace_S = self._score_divergence_terms(self.score_net.model, ht, x, beta)
        drift_X = torch.empty_like(x)
        drift_A, div_bt, cross, dUdt, Ut = (torch.empty(B, device=x.device) for _ in range(5))
        _lib.check(_lib.lib().pita_fk_assemble(
            x.data_ptr(), ht.data_ptr(), g2.data_ptr(), dhdt.data_ptr(), D_E.data_ptr(), jtx_E.data_ptr(),
            dot_h.data_ptr(), _lib.ptr(h_parts), D_S.data_ptr(), trace_S.data_ptr(), gamma, dgamma,
            _lib.ptr(beta_b if pb_e else None), _lib.ptr(beta_b if pb_s else None), pin_w, pin_dw, _lib.ptr(logp_t),
            drift_X.data_ptr(), drift_A.data_ptr(), div_bt.data_ptr(), cross.data_ptr(), dUdt.data_ptr(), Ut.data_ptr(),
            B, D, _lib.stream_ptr(x.device)), "pita_fk_assemble")
        # 0.9-quantile clamp of the weight drift over this inference chunk (sdes.py:230), K11
        _lib.check(_lib.lib().pita_quantile_clamp(drift_A.data_ptr(), B, int(clamp_chunk or B), 0.9,
                                                  _lib.stream_ptr(x.device)), "pita_quantile_clamp")
        return SDETerms(drift_X=drift_X, drift_A=drift_A, divergence_score=div_bt, cross_term=cross, dUt_dt=dUdt)

    def diffusion(self, t, x, diffusion_scale):
        t = _per_walker(t, x)
        return diffusion_scale * self.g(t)[:, None] * torch.randn_like(x)
