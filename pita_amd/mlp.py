"""MLP score backbones executed by the HIP MFMA kernel.

Mirror of ``MyMLP`` (pita/src/models/components/mlp.py:199-267) and ``MyMLPTemperature``
(:453-524): sinusoidal embedding of every input coordinate (scale 25), of time (and of beta),
Linear -> GELU, ``hidden_layers`` residual GELU blocks, Linear.  Same constructor arguments,
parameter names (``joint_mlp.N[.ff].{weight,bias}``) and creation order as the reference, so
seeded construction and checkpoints carry over.  The arithmetic is pita_amd/csrc/mlp_kernel.hip.
"""
import ctypes

import torch
from torch import nn

from . import _lib


class _Block(nn.Module):  # parameter holder for mlp.py:100-118 (x + GELU(ff(x)))
    def __init__(self, size):
        super().__init__()
        self.ff = nn.Linear(size, size)


class _HipMLP(nn.Module):
    _temperature = False

    def __init__(self, hidden_size=128, hidden_layers=3, emb_size=128, out_dim=2, time_emb="sinusoidal",
                 input_emb="sinusoidal", add_t_emb=False, concat_t_emb=False, input_dim=2, energy_function=None):
        super().__init__()
        if time_emb != "sinusoidal" or input_emb != "sinusoidal" or add_t_emb or concat_t_emb:
            raise NotImplementedError("HIP MLP implements the sinusoidal-embedding configuration of model/net/mlp.yaml")
        self.hidden_size, self.hidden_layers, self.emb_size = hidden_size, hidden_layers, emb_size
        self.input_dim, self.out_dim = input_dim, out_dim
        concat = emb_size * (input_dim + 1 + (1 if self._temperature else 0))
        layers = [nn.Linear(concat, hidden_size)]
        layers += [_Block(hidden_size) for _ in range(hidden_layers)]
        layers.append(nn.Linear(emb_size, out_dim))  # sized by emb_size like the reference (mlp.py:238-239)
        self.joint_mlp = nn.Sequential(*layers)
        self._handle, self._handle_key = None, None

    def _freqs(self):
        half = self.emb_size // 2  # the reference's fp32 torch ops (mlp.py:19-21)
        w = torch.log(torch.Tensor([10000.0])) / (half - 1)
        return torch.exp(-w * torch.arange(half)).contiguous()

    def _native(self, device):
        tensors = self.__dict__.get("_tensor_list")
        if tensors is None:
            tensors = self.__dict__["_tensor_list"] = list(self.parameters()) + list(self.buffers())
        key = (device.index,) + tuple((p.data_ptr(), p._version) for p in tensors)
        if self._handle is None or key != self._handle_key:
            self._release()
            params = list(self.state_dict().values())
            flat = torch.cat([p.detach().to("cpu", torch.float32).reshape(-1) for p in params]).contiguous().numpy()
            fr = self._freqs().numpy()
            cfg = _lib.MlpConfig(self.input_dim, self.out_dim, self.hidden_size, self.hidden_layers, self.emb_size,
                                 int(self._temperature))
            h = ctypes.c_void_p()
            with torch.cuda.device(device):
                _lib.check(_lib.lib().pita_mlp_create(ctypes.byref(h), ctypes.byref(cfg),
                                                      flat.ctypes.data_as(ctypes.c_void_p), flat.size,
                                                      fr.ctypes.data_as(ctypes.c_void_p)), "pita_mlp_create")
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
            _lib.lib().pita_mlp_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _run(self, t, x, beta):
        x = _lib.dev_tensor(x, "x")
        B = x.shape[0]
        t = _lib.dev_tensor(t, "t").reshape(-1).expand(B).contiguous()
        if beta is not None:
            beta = _lib.dev_tensor(beta, "beta").reshape(-1).expand(B).contiguous()
        out = torch.empty(B, self.out_dim, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().pita_mlp_forward(self._native(x.device), t.data_ptr(), x.data_ptr(), _lib.ptr(beta),
                                               out.data_ptr(), B, _lib.stream_ptr(x.device)), "pita_mlp_forward")
        return out


    def sampler_run(self, x, step_tab, n_steps, noise=None, seed=0, walker_offset=0, step0=0, remove_mean=True,
                    drift_out=None, n_particles=None, n_dim=None, stats_out=None):
        """In-place fused Euler-Maruyama steps (pita_mlp_sampler_run); x: [B, D] device tensor.  Geometry defaults to
        one "particle" of dimension D (the GMM convention of the integrator)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        assert step_tab.is_cuda and step_tab.dtype == torch.float32 and step_tab.is_contiguous()
        if drift_out is not None:
            raise NotImplementedError("MLP fused sampler: drift_out is not provided (use the per-step path)")
        if stats_out is not None:
            assert stats_out.is_cuda and stats_out.dtype == torch.float64 and stats_out.is_contiguous()
        D = x.shape[1]
        n, d = (1, D) if n_particles is None else (int(n_particles), int(n_dim))
        _lib.check(_lib.lib().pita_mlp_sampler_run(
            self._native(x.device), x.data_ptr(), x.shape[0], step_tab.data_ptr(), int(n_steps), _lib.ptr(noise),
            int(seed) & 0xFFFFFFFFFFFFFFFF, int(walker_offset), int(step0), int(bool(remove_mean)), n, d,
            _lib.ptr(stats_out), _lib.stream_ptr(x.device)), "pita_mlp_sampler_run")
        return x

    def can_fuse(self, n_particles, n_dim):
        D = n_particles * n_dim
        return self.input_dim == D and self.out_dim == D and D <= 64 and n_dim <= 4


class MyMLP(_HipMLP):
    def forward(self, t, x, x_self_cond=False):
        """The third positional argument swallows beta exactly like the reference (mlp.py:244)."""
        return self._run(t, x, None)


class MyMLPTemperature(_HipMLP):
    _temperature = True

    def forward(self, t, x, beta):
        return self._run(t, x, beta)
