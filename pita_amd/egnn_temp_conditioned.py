"""E(n)-equivariant GNN backbone executed by the fused HIP kernel.

Mirror of ``EGNN_dynamics`` (pita/src/models/components/egnn_temp_conditioned.py:7-117): same
constructor arguments, same parameter names and creation order (so a seeded construction gives
the reference's weights and its ``state_dict`` / Lightning checkpoints load unchanged), same
``forward(t, xs, beta) -> vel`` contract.  The module only OWNS the parameters; the arithmetic
of EGNN.forward (:172-194) and E_GCL (:197-356) lives in pita_amd/csrc/egnn_kernel.hip.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib


DEFAULT_PRECISION = "f16x2"  # dense-layer arithmetic of the HIP EGNN when the caller does not choose (see EGNN_dynamics)


class _GCLParams(nn.Module):
    """Parameters of one E_GCL layer, created in the reference's order (:232-260)."""

    def __init__(self, hidden_nf, act_fn, attention, tanh, edges_in_d=1):
        super().__init__()
        h = hidden_nf
        self.edge_mlp = nn.Sequential(nn.Linear(2 * h + 1 + edges_in_d, h), act_fn, nn.Linear(h, h), act_fn)
        self.node_mlp = nn.Sequential(nn.Linear(2 * h, h), act_fn, nn.Linear(h, h))
        head = nn.Linear(h, 1, bias=False)
        torch.nn.init.xavier_uniform_(head.weight, gain=0.001)
        coord = [nn.Linear(h, h), act_fn, head]
        if tanh:
            coord.append(nn.Tanh())
        self.coord_mlp = nn.Sequential(*coord)
        if attention:
            self.att_mlp = nn.Sequential(nn.Linear(h, 1), nn.Sigmoid())


class EGNN(nn.Module):
    """Parameter container matching the reference ``EGNN`` module tree (:120-170)."""

    def __init__(self, in_node_nf, in_edge_nf, hidden_nf, act_fn=nn.SiLU(), n_layers=4, recurrent=True,
                 attention=False, norm_diff=True, out_node_nf=None, tanh=False, coords_range=15, agg="sum",
                 has_virtual=False):
        super().__init__()
        if has_virtual or not recurrent or not norm_diff or agg != "sum" or in_edge_nf != 1:
            raise NotImplementedError("HIP EGNN implements recurrent=True, norm_diff=True, agg='sum', no virtual node")
        if not isinstance(act_fn, nn.SiLU):
            raise NotImplementedError("HIP EGNN implements the SiLU activation used by every reference config")
        out_node_nf = in_node_nf if out_node_nf is None else out_node_nf
        self.hidden_nf, self.n_layers = hidden_nf, n_layers
        self.coords_range = float(coords_range)
        self.attention, self.tanh = attention, tanh
        self.embedding = nn.Linear(in_node_nf, hidden_nf)
        self.embedding_out = nn.Linear(hidden_nf, out_node_nf)  # dead in the reference forward (:80,189)
        for i in range(n_layers):
            self.add_module("gcl_%d" % i, _GCLParams(hidden_nf, act_fn, attention, tanh, in_edge_nf))


class EGNN_dynamics(nn.Module):
    def __init__(self, n_particles, n_dimension, hidden_nf=64, act_fn=torch.nn.SiLU(), n_layers=4, recurrent=True,
                 attention=False, condition_time=True, tanh=False, agg="sum", energy=False, add_virtual=False,
                 condition_temperature=False, feature_layout="pita", precision=None):
        super().__init__()
        if energy or add_virtual or not condition_time:
            raise NotImplementedError("HIP EGNN_dynamics implements energy=False, add_virtual=False, condition_time=True")
        self.in_node_nf = 2 if condition_temperature else 1
        self.egnn = EGNN(in_node_nf=self.in_node_nf, in_edge_nf=1, hidden_nf=hidden_nf, act_fn=act_fn,
                         n_layers=n_layers, recurrent=recurrent, attention=attention, tanh=tanh, agg=agg)
        self._n_particles, self._n_dimension = n_particles, n_dimension
        self.condition_time, self.condition_temperature = condition_time, condition_temperature
        # "pita" keeps the reference's t/beta interleave quirk (:68-78); "correct" gives (t, beta) per node
        self.feature_layout = {"pita": 0, "correct": 1}[feature_layout]
        # dense-layer arithmetic, all fp32-accurate: "f32" = f32 MFMA (bit-exact fmaf chains), "bf16x3" = bf16 matrix
        # pipe with an exact three-way operand split, "f16x2" = f16 matrix pipe with a two-way round-to-nearest split
        # and power-of-two operand scaling (|activations| < 4094; see csrc/egnn_common.h)
        precision = precision or os.environ.get("PITA_EGNN_PRECISION", DEFAULT_PRECISION)
        self.precision = {"f32": 0, "bf16x3": 1, "f16x2": 2}[precision]
        self.counter = 0
        self._handle = None
        self._handle_key = None

    # ------------------------------------------------------------------ native handle
    def _config(self):
        e = self.egnn
        return _lib.EgnnConfig(self._n_particles, self._n_dimension, e.hidden_nf, e.n_layers, self.in_node_nf,
                               int(e.attention), int(e.tanh), e.coords_range, self.feature_layout, self.precision)

    def _native(self, device):
        # cheap staleness check on every call: (storage pointer, version counter) of every parameter/buffer
        tensors = self.__dict__.get("_tensor_list")
        if tensors is None:
            tensors = self.__dict__["_tensor_list"] = list(self.parameters()) + list(self.buffers())
        key = (device.index, self.precision) + tuple((p.data_ptr(), p._version) for p in tensors)
        if self._handle is None or key != self._handle_key:
            self._release()
            params = list(self.state_dict().values())
            flat = torch.cat([p.detach().to("cpu", torch.float32).reshape(-1) for p in params]).contiguous().numpy()
            cfg = self._config()
            h = ctypes.c_void_p()
            with torch.cuda.device(device):
                _lib.check(_lib.lib().pita_egnn_create(ctypes.byref(h), ctypes.byref(cfg),
                                                       flat.ctypes.data_as(ctypes.c_void_p), flat.size),
                           "pita_egnn_create")
            self._handle, self._handle_key = h, key
        return self._handle

    def __getstate__(self):
        """The native handle is a per-process device resource: copies / pickles start without one."""
        state = self.__dict__.copy()
        state["_handle"], state["_handle_key"] = None, None
        state.pop("_tensor_list", None)
        return state

    def _release(self):
        if self._handle is not None:
            _lib.lib().pita_egnn_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference interface
    def forward(self, t, xs, beta=None):
        """vel[B, n*d] = backbone(t[B], xs[B, n*d], beta[B]); mean-free (:56-93)."""
        xs = _lib.dev_tensor(xs, "xs")
        B = xs.shape[0]
        t = _lib.dev_tensor(t, "t").reshape(-1).expand(B).contiguous()
        if self.condition_temperature:
            if beta is None:
                raise ValueError("EGNN_dynamics(condition_temperature=True).forward needs beta")
            beta = _lib.dev_tensor(beta, "beta").reshape(-1).expand(B).contiguous()
        else:
            beta = None
        out = torch.empty_like(xs)
        _lib.check(_lib.lib().pita_egnn_forward(self._native(xs.device), t.data_ptr(), xs.data_ptr(), _lib.ptr(beta),
                                                out.data_ptr(), B, _lib.stream_ptr(xs.device)), "pita_egnn_forward")
        self.counter += 1
        return out

    # ------------------------------------------------------------------ fused extensions used by ScoreNet / the sampler
    def edm(self, what, h_t, x_t, beta):
        """what=1: denoiser D_theta, what=2: score (D_theta - x)/h, EDM preconditioning fused (score_net.py:13-43)."""
        x_t = _lib.dev_tensor(x_t, "x_t")
        B = x_t.shape[0]
        h_t = _lib.dev_tensor(h_t, "h_t").reshape(-1).expand(B).contiguous()
        b = None
        if self.condition_temperature:
            b = _as_batch(beta, B, x_t.device)
        out = torch.empty_like(x_t)
        _lib.check(_lib.lib().pita_egnn_edm(self._native(x_t.device), what, h_t.data_ptr(), x_t.data_ptr(), _lib.ptr(b),
                                            out.data_ptr(), B, _lib.stream_ptr(x_t.device)), "pita_egnn_edm")
        return out

    def jvp(self, h_t, x_t, beta, vx=None, direction=-1, vh=None, want_primal=True, want_tangent=True, dot_out=None,
            dot_col=0, diag_acc=None):
        """(D, dD): the denoiser D_theta(h, x) and its forward-mode derivative along ONE tangent direction:
        dD = J_x D . vx + dD/dh . vh.  ``vx``: [B, D] tensor, or None for the unit direction ``direction`` of
        every walker (-1 = zero).  ``vh``: [B] tensor or None.  Optional in-kernel reductions:
        ``dot_out[:, dot_col] = <x, dD>`` and ``diag_acc += dD[:, direction]``.
        (pita_egnn_jvp; fp32-accurate bf16x3 arithmetic.)"""
        x_t = _lib.dev_tensor(x_t, "x_t")
        B = x_t.shape[0]
        h_t = _lib.dev_tensor(h_t, "h_t").reshape(-1).expand(B).contiguous()
        b = _as_batch(beta, B, x_t.device) if self.condition_temperature else None
        if vx is not None:
            vx = _lib.dev_tensor(vx, "vx")
        if vh is not None:
            vh = _lib.dev_tensor(vh, "vh").reshape(-1).expand(B).contiguous()
        out = torch.empty_like(x_t) if want_primal else None
        dout = torch.empty_like(x_t) if want_tangent else None
        stride = 1
        if dot_out is not None:
            assert dot_out.is_cuda and dot_out.dtype == torch.float32 and dot_out.is_contiguous()
            stride = dot_out.shape[1] if dot_out.dim() == 2 else 1
        _lib.check(_lib.lib().pita_egnn_jvp(self._native(x_t.device), h_t.data_ptr(), x_t.data_ptr(), _lib.ptr(b),
                                            _lib.ptr(vx), int(direction), _lib.ptr(vh), _lib.ptr(out), _lib.ptr(dout),
                                            _lib.ptr(dot_out), stride, int(dot_col), _lib.ptr(diag_acc), B,
                                            _lib.stream_ptr(x_t.device)), "pita_egnn_jvp")
        return out, dout

    def jacobian_trace(self, h_t, x_t, beta, want_denoiser=False):
        """trace(J_x D_theta(h, x)) per walker, exactly, over all dim unit directions (pita_egnn_jacobian_trace: the
        primal network is evaluated once, its per-edge factors cached for the tangent-only launches that follow).
        ``want_denoiser``: also return D_theta(h, x), a by-product of the primal -> (trace, D)."""
        x_t = _lib.dev_tensor(x_t, "x_t")
        B, D = x_t.shape
        h_t = _lib.dev_tensor(h_t, "h_t").reshape(-1).expand(B).contiguous()
        b = _as_batch(beta, B, x_t.device) if self.condition_temperature else None
        L = _lib.lib()
        net = self._native(x_t.device)
        trace = torch.empty(B, device=x_t.device)
        den = torch.empty_like(x_t) if want_denoiser else None
        with torch.cuda.device(x_t.device):
            _lib.check(L.pita_egnn_jacobian_trace(net, h_t.data_ptr(), x_t.data_ptr(), _lib.ptr(b), trace.data_ptr(),
                                                  _lib.ptr(den), B, _lib.stream_ptr(x_t.device)),
                       "pita_egnn_jacobian_trace")
        return (trace, den) if want_denoiser else trace

    vjp_h_parts = True  # vjp(..., want_h_parts=True) is available

    def vjp(self, h_t, x_t, beta, cot=None, want_primal=True, want_dot_h=False, want_h_parts=False):
        """(D, J_x D^T cot): the denoiser and its reverse-mode derivative for a per-walker cotangent (default: x_t
        itself, which is what grad_x E_theta needs).  One launch (pita_egnn_vjp) instead of dim JVP launches.
        ``want_dot_h``: also return <cot, dD/dh> [B] from the same sweep -> (D, vjp, dot_h).
        ``want_h_parts`` (with want_dot_h): also [B, 2] = (c_out <cot, F>, <cot, d(c_out F)/dh>), the split of the same
        derivative that pita_fk_assemble turns into E_theta and dE_theta/dh without cancellation at small h
        -> (D, vjp, dot_h, parts)."""
        x_t = _lib.dev_tensor(x_t, "x_t")
        B = x_t.shape[0]
        h_t = _lib.dev_tensor(h_t, "h_t").reshape(-1).expand(B).contiguous()
        b = _as_batch(beta, B, x_t.device) if self.condition_temperature else None
        if cot is not None:
            cot = _lib.dev_tensor(cot, "cot")
        out = torch.empty_like(x_t) if want_primal else None
        vjp = torch.empty_like(x_t)
        dot_h = torch.empty(B, device=x_t.device) if want_dot_h else None
        if want_h_parts and not want_dot_h:
            raise ValueError("want_h_parts needs want_dot_h")
        parts = torch.empty(B, 2, device=x_t.device) if want_h_parts else None
        _lib.check(_lib.lib().pita_egnn_vjp(self._native(x_t.device), h_t.data_ptr(), x_t.data_ptr(), _lib.ptr(b),
                                            _lib.ptr(cot), _lib.ptr(out), vjp.data_ptr(), _lib.ptr(dot_h),
                                            _lib.ptr(parts), B, _lib.stream_ptr(x_t.device)), "pita_egnn_vjp")
        if want_h_parts:
            return out, vjp, dot_h, parts
        return (out, vjp, dot_h) if want_dot_h else (out, vjp)

    def sampler_run(self, x, step_tab, n_steps, noise=None, seed=0, walker_offset=0, step0=0, remove_mean=True,
                    drift_out=None, n_particles=None, n_dim=None, stats_out=None):
        """In-place fused Euler-Maruyama steps (pita_egnn_sampler_run); x: [B, n*d] device tensor.  ``stats_out``:
        optional contiguous float64 device tensor [n_steps, 4] that accumulates the per-step drift / diffusion moments."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        if stats_out is not None:
            assert stats_out.is_cuda and stats_out.dtype == torch.float64 and stats_out.is_contiguous()
            assert stats_out.numel() >= 4 * int(n_steps)
        assert step_tab.is_cuda and step_tab.dtype == torch.float32 and step_tab.is_contiguous()
        _lib.check(_lib.lib().pita_egnn_sampler_run(
            self._native(x.device), x.data_ptr(), x.shape[0], step_tab.data_ptr(), int(n_steps), _lib.ptr(noise),
            int(seed) & 0xFFFFFFFFFFFFFFFF, int(walker_offset), int(step0), int(bool(remove_mean)),
            _lib.ptr(drift_out), _lib.ptr(stats_out), _lib.stream_ptr(x.device)), "pita_egnn_sampler_run")
        return x


    def sampler_work(self, B, device):
        """(16-bit MFMAs, f32 MFMAs) executed per walker-step by the fused sampler at batch B (pita_egnn_sampler_work)."""
        a, b = ctypes.c_double(), ctypes.c_double()
        _lib.check(_lib.lib().pita_egnn_sampler_work(self._native(device), int(B), ctypes.byref(a), ctypes.byref(b)),
                   "pita_egnn_sampler_work")
        return a.value, b.value

    def sampler_mapping(self, B, device):
        """(walkers per wavefront group, wavefronts launched, resident wavefront slots) of the fused sampler at batch B
        (pita_egnn_sampler_mapping)."""
        g, w, s = ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.lib().pita_egnn_sampler_mapping(self._native(device), int(B), ctypes.byref(g), ctypes.byref(w),
                                                        ctypes.byref(s)), "pita_egnn_sampler_mapping")
        return g.value, w.value, s.value


def _as_batch(v, B, device):
    if isinstance(v, torch.Tensor):
        return _lib.dev_tensor(v.to(device), "beta").reshape(-1).expand(B).contiguous()
    return torch.full((B,), float(v), device=device, dtype=torch.float32)
