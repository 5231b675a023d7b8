"""EGNN backbone of the alanine-dipeptide class (hidden 64 x 5 layers, one-hot atom-type node features) on the HIP
kernels of ``pita_egnn_wide_eval``: the matrix-pipe kernel for 22 / 33 / 42 / 13 / 55 particles, the vector-pipe kernel
for every other shape.

Mirror of ``EGNN_dynamics_AD2_cat`` (pita/src/models/components/egnn_dynamics_ad2_cat.py:11-203; the ``_target_`` of
``configs/model/net/egnn_dynamics_ad2_cat.yaml``): same constructor arguments and defaults, same parameter names and
creation order (``egnn.embedding``, ``egnn.embedding_out``, ``egnn.gcl_<l>.{edge_mlp,node_mlp,coord_mlp,att_mlp}``: a
seeded construction gives the reference's weights and its ``state_dict`` loads unchanged), same
``forward(t, xs, beta) -> vel`` contract.  Node features are ``[h_initial (static one-hot rows), t, beta]`` (:157-184);
the arithmetic of ``EGNN.forward`` / ``E_GCL`` (egnn.py:108-346) lives in pita_amd/csrc/egnn_wide_mfma_kernel.hip and
pita_amd/csrc/egnn_wide_kernel.hip (``pita_egnn_wide_eval``).  ``edm`` lets ``ScoreNet`` evaluate the EDM preconditioning in the same launch.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .egnn_temp_conditioned import EGNN, _as_batch


class EGNN_dynamics_AD2_cat(nn.Module):
    def __init__(self, n_particles, n_dimensions, hidden_nf=64, act_fn=torch.nn.SiLU(), n_layers=5, recurrent=True,
                 attention=True, tanh=True, atom_encoding_filename="atom_types_ecoding.npy", data_dir="data/alanine",
                 pdb_filename="", agg="sum", M=128, condition_beta=False, h_initial=None):
        super().__init__()
        self._n_particles, self._n_dimensions = n_particles, n_dimensions
        if h_initial is None:
            h_initial = self.get_h_initial()
        self.h_initial = torch.as_tensor(h_initial)
        self.condition_beta = condition_beta
        h_size = self.h_initial.size(1) + 1 + (1 if condition_beta else 0)  # :44-49
        self.egnn = EGNN(in_node_nf=h_size, in_edge_nf=1, hidden_nf=hidden_nf, act_fn=act_fn, n_layers=n_layers,
                         recurrent=recurrent, attention=attention, tanh=tanh, agg=agg)
        self.counter = 0
        self.M = M
        self._handle = None
        self._handle_key = None

    def get_h_initial(self):
        """The static node features of :66-92 for the particle counts that need no topology file."""
        n = self._n_particles
        groups = {22: [([0, 2, 3], 2), ([19, 20, 21], 20), ([11, 12, 13], 12)],
                  33: [([1, 2, 3], 2), ([9, 10, 11], 10), ([19, 20, 21], 18), ([29, 30, 31], 31)],
                  42: [([1, 2, 3], 2), ([11, 12, 13], 12), ([21, 22, 23], 22), ([31, 32, 33], 32), ([39, 40, 41], 40)]}
        if n in groups:
            atom_types = np.arange(n)
            for idx, v in groups[n]:
                atom_types[idx] = v
            return torch.nn.functional.one_hot(torch.tensor(atom_types))
        if n in (13, 55):
            return torch.zeros(n, 1)
        raise NotImplementedError(
            f"EGNN_dynamics_AD2_cat: the node features of {n} particles come from a topology file (mdtraj, :94-155); "
            "pass h_initial=[n_particles, n_features] instead")

    # ------------------------------------------------------------------ native handle
    def _config(self):
        e = self.egnn
        return _lib.EgnnWideConfig(self._n_particles, self._n_dimensions, e.hidden_nf, e.n_layers,
                                   int(self.h_initial.size(1)), int(bool(self.condition_beta)), int(e.attention),
                                   int(e.tanh), e.coords_range)

    def _native(self, device):
        tensors = self.__dict__.get("_tensor_list")
        if tensors is None:
            tensors = self.__dict__["_tensor_list"] = list(self.parameters()) + list(self.buffers())
        key = (device.index,) + tuple((p.data_ptr(), p._version) for p in tensors)
        if self._handle is None or key != self._handle_key:
            self._release()
            flat = torch.cat([p.detach().to("cpu", torch.float32).reshape(-1) for p in self.state_dict().values()]).contiguous().numpy()
            h0 = np.ascontiguousarray(self.h_initial.detach().to("cpu", torch.float32).numpy())
            cfg = self._config()
            h = ctypes.c_void_p()
            with torch.cuda.device(device):
                _lib.check(_lib.lib().pita_egnn_wide_create(ctypes.byref(h), ctypes.byref(cfg),
                                                            flat.ctypes.data_as(ctypes.c_void_p), flat.size,
                                                            h0.ctypes.data_as(ctypes.c_void_p)), "pita_egnn_wide_create")
            self._handle, self._handle_key = h, key
        return self._handle

    def uses_matrix_pipe(self, device):
        """True when evaluations on ``device`` run on the MFMA kernel (the instantiated particle counts, unless
        PITA_WIDE_NO_MFMA is set)."""
        return bool(_lib.lib().pita_egnn_wide_uses_matrix_pipe(self._native(torch.device(device))))

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_handle"], state["_handle_key"] = None, None
        state.pop("_tensor_list", None)
        return state

    def _release(self):
        if self._handle is not None:
            _lib.lib().pita_egnn_wide_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference interface
    def _eval(self, what, t, xs, beta):
        xs = _lib.dev_tensor(xs, "xs")
        B = xs.shape[0]
        t = _lib.dev_tensor(t, "t").reshape(-1).expand(B).contiguous()
        b = None
        if self.condition_beta:
            if beta is None:
                raise ValueError("EGNN_dynamics_AD2_cat(condition_beta=True) needs beta")
            b = _as_batch(beta, B, xs.device)
        out = torch.empty_like(xs)
        _lib.check(_lib.lib().pita_egnn_wide_eval(self._native(xs.device), int(what), t.data_ptr(), xs.data_ptr(), _lib.ptr(b),
                                                  out.data_ptr(), B, _lib.stream_ptr(xs.device)), "pita_egnn_wide_eval")
        return out

    def forward(self, t, xs, beta=None):
        """vel[B, n*d] = backbone(t[B], xs[B, n*d], beta[B]); mean-free (:157-203)."""
        self.counter += 1
        return self._eval(0, t, xs, beta)

    def edm(self, what, h_t, x_t, beta):
        """what=1: denoiser D_theta, what=2: score (D_theta - x)/h, EDM preconditioning fused (score_net.py:13-43)."""
        return self._eval(what, h_t, x_t, beta)

    def can_fuse(self, n_particles, n_dim):
        return int(n_particles) == self._n_particles and int(n_dim) == self._n_dimensions

    def sampler_run(self, x, step_tab, n_steps, noise=None, seed=0, walker_offset=0, step0=0, remove_mean=True,
                    drift_out=None, n_particles=None, n_dim=None, stats_out=None):
        """In-place fused Euler-Maruyama steps of the not-debiased reverse SDE (pita_egnn_wide_sampler_run): the whole
        stretch between two resampling events in ONE launch, like ``EGNN_dynamics.sampler_run``; x: [B, n*d] device
        tensor, ``step_tab`` from ``sde_integration.build_step_table`` (its beta column conditions the net)."""
        if drift_out is not None:
            raise NotImplementedError("EGNN_dynamics_AD2_cat.sampler_run: drift_out")
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        if stats_out is not None:
            assert stats_out.is_cuda and stats_out.dtype == torch.float64 and stats_out.is_contiguous()
            assert stats_out.numel() >= 4 * int(n_steps)
        assert step_tab.is_cuda and step_tab.dtype == torch.float32 and step_tab.is_contiguous()
        _lib.check(_lib.lib().pita_egnn_wide_sampler_run(
            self._native(x.device), x.data_ptr(), x.shape[0], step_tab.data_ptr(), int(n_steps), _lib.ptr(noise),
            int(seed) & 0xFFFFFFFFFFFFFFFF, int(walker_offset), int(step0), int(bool(remove_mean)), _lib.ptr(stats_out),
            _lib.stream_ptr(x.device)), "pita_egnn_wide_sampler_run")
        return x

    def jvp(self, h_t, x_t, beta, vx=None, direction=-1, vh=None, want_primal=True, want_tangent=True, dot_out=None,
            dot_col=0, diag_acc=None):
        """(D, dD): the EDM denoiser around this backbone and its forward-mode derivative along ONE direction,
        dD = J_x D . vx + dD/dh . vh -- same contract as ``EGNN_dynamics.jvp`` (pita_egnn_wide_jvp, fp32 vector-pipe
        kernel).  With it ``VEReverseSDE(debias_inference=True)`` runs on this backbone: the exact divergence of the
        score (utils.py:30-51), grad_x E_theta (energy_net.py:51-62) and dE_theta/dt (sdes.py:218) are assembled from
        dim + 1 such launches per net (sdes.py's forward-mode path)."""
        x_t = _lib.dev_tensor(x_t, "x_t")
        B = x_t.shape[0]
        h_t = _lib.dev_tensor(h_t, "h_t").reshape(-1).expand(B).contiguous()
        b = None
        if self.condition_beta:
            if beta is None:
                raise ValueError("EGNN_dynamics_AD2_cat(condition_beta=True) needs beta")
            b = _as_batch(beta, B, x_t.device)
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
        _lib.check(_lib.lib().pita_egnn_wide_jvp(self._native(x_t.device), h_t.data_ptr(), x_t.data_ptr(), _lib.ptr(b),
                                                 _lib.ptr(vx), int(direction), _lib.ptr(vh), _lib.ptr(out), _lib.ptr(dout),
                                                 _lib.ptr(dot_out), stride, int(dot_col), _lib.ptr(diag_acc), B,
                                                 _lib.stream_ptr(x_t.device)), "pita_egnn_wide_jvp")
        return out, dout

    def vjp(self, h_t, x_t, beta, cot=None, want_primal=True, want_dot_h=False):
        """(D, J_x D^T cot[, <cot, dD/dh>]): the EDM denoiser around this backbone and its reverse-mode derivative for a
        per-walker cotangent (default: x_t itself) from ONE launch (pita_egnn_wide_vjp) -- same contract as
        ``EGNN_dynamics.vjp``; ``EnergyNet.forward`` and the debiased drift use it instead of dim + 1 ``jvp`` launches."""
        x_t = _lib.dev_tensor(x_t, "x_t")
        B = x_t.shape[0]
        h_t = _lib.dev_tensor(h_t, "h_t").reshape(-1).expand(B).contiguous()
        b = None
        if self.condition_beta:
            if beta is None:
                raise ValueError("EGNN_dynamics_AD2_cat(condition_beta=True) needs beta")
            b = _as_batch(beta, B, x_t.device)
        if cot is not None:
            cot = _lib.dev_tensor(cot, "cot")
        out = torch.empty_like(x_t) if want_primal else None
        vjp = torch.empty_like(x_t)
        dot_h = torch.empty(B, device=x_t.device) if want_dot_h else None
        _lib.check(_lib.lib().pita_egnn_wide_vjp(self._native(x_t.device), h_t.data_ptr(), x_t.data_ptr(), _lib.ptr(b),
                                                 _lib.ptr(cot), _lib.ptr(out), vjp.data_ptr(), _lib.ptr(dot_h), B,
                                                 _lib.stream_ptr(x_t.device)), "pita_egnn_wide_vjp")
        return (out, vjp, dot_h) if want_dot_h else (out, vjp)
