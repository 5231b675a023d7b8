"""ctypes binding of libpita_hip.so (include/pita_hip.h).

The library is the product: there is NO CPU / PyTorch fallback.  If the shared object is
missing or a call fails this module raises; nothing silently degrades.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpita_hip.so")

STEP_STRIDE = 16
ABI_VERSION = 11
ST_CS, ST_CIN, ST_COUT, ST_CNOISE, ST_H, ST_G2, ST_GAMMA, ST_DT, ST_NOISE_SCALE, ST_SQRT_DT, ST_BETA = range(11)


class PitaHipError(RuntimeError):
    pass


class EgnnConfig(ctypes.Structure):
    _fields_ = [("n_particles", c_int), ("n_dim", c_int), ("hidden_nf", c_int), ("n_layers", c_int),
                ("in_node_nf", c_int), ("attention", c_int), ("tanh", c_int), ("coords_range", c_float),
                ("feature_layout", c_int), ("precision", c_int)]


class EgnnWideConfig(ctypes.Structure):
    _fields_ = [("n_particles", c_int), ("n_dim", c_int), ("hidden_nf", c_int), ("n_layers", c_int),
                ("n_static", c_int), ("condition_beta", c_int), ("attention", c_int), ("tanh", c_int),
                ("coords_range", c_float)]


class FfConfig(ctypes.Structure):
    _fields_ = [("n_atoms", c_int),
                ("n_bonds", c_int), ("bond_idx", c_void_p), ("bond_par", c_void_p),
                ("n_angles", c_int), ("angle_idx", c_void_p), ("angle_par", c_void_p),
                ("n_torsions", c_int), ("tors_idx", c_void_p), ("tors_par", c_void_p),
                ("charge", c_void_p), ("sigma", c_void_p), ("epsilon", c_void_p),
                ("n_exceptions", c_int), ("exc_idx", c_void_p), ("exc_par", c_void_p),
                ("use_cutoff", c_int), ("cutoff", c_float), ("rf_dielectric", c_float),
                ("length_scale", c_float), ("kT", c_float),
                ("gb_radius", c_void_p), ("gb_scale", c_void_p), ("gb_solute_dielectric", c_float),
                ("gb_solvent_dielectric", c_float), ("gb_surface_area_factor", c_float)]


class MlpConfig(ctypes.Structure):
    _fields_ = [("input_dim", c_int), ("out_dim", c_int), ("hidden_size", c_int), ("hidden_layers", c_int),
                ("emb_size", c_int), ("temperature_conditioned", c_int)]


_PROTOS = {
    "pita_abi_version": (c_int, []),
    "pita_last_error": (c_int, [c_char_p, c_size_t]),
    "pita_device_count": (c_int, []),
    "pita_lj_logp_force": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_float, c_float,
                                   c_float, c_float, c_float, c_void_p]),
    "pita_lj_smooth_logp_force": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_float, c_float,
                                          c_float, c_float, c_float, c_float, c_void_p, c_void_p]),
    "pita_dw_logp_force": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_float, c_float,
                                   c_float, c_float, c_void_p]),
    "pita_lj_descent": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int] + [c_float] * 6 + [c_int, c_float, c_float,
                                c_float, c_uint64, c_uint64, c_int64, c_int, c_void_p]),
    "pita_dw_descent": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int] + [c_float] * 5 + [c_int, c_float, c_float,
                                c_float, c_uint64, c_uint64, c_int64, c_int, c_void_p]),
    "pita_lj_mala_workspace_bytes": (c_size_t, [c_int]),
    "pita_lj_mala": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int] + [c_float] * 6 +
                     [c_int, c_void_p, c_int, c_int64, c_uint64, c_uint64, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                      c_void_p]),
    "pita_dw_mala": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int] + [c_float] * 5 +
                     [c_int, c_void_p, c_int, c_int64, c_uint64, c_uint64, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                      c_void_p]),
    "pita_gmm_logp_force": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_float,
                                    c_void_p]),
    "pita_ff_create": (c_int, [POINTER(c_void_p), POINTER(FfConfig)]),
    "pita_ff_destroy": (c_int, [c_void_p]),
    "pita_ff_logp_force": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "pita_ff_descent": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_float, c_float, c_uint64,
                                c_uint64, c_int64, c_int, c_void_p]),
    "pita_egnn_create": (c_int, [POINTER(c_void_p), POINTER(EgnnConfig), c_void_p, c_int64]),
    "pita_egnn_destroy": (c_int, [c_void_p]),
    "pita_egnn_num_weights": (c_int64, [POINTER(EgnnConfig)]),
    "pita_egnn_wide_num_weights": (c_int64, [POINTER(EgnnWideConfig)]),
    "pita_egnn_wide_create": (c_int, [POINTER(c_void_p), POINTER(EgnnWideConfig), c_void_p, c_int64, c_void_p]),
    "pita_egnn_wide_destroy": (c_int, [c_void_p]),
    "pita_egnn_wide_uses_matrix_pipe": (c_int, [c_void_p]),
    "pita_egnn_wide_eval": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "pita_egnn_wide_sampler_run": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_uint64, c_uint64,
                                           c_int64, c_int, c_void_p, c_void_p]),
    "pita_egnn_wide_jvp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "pita_egnn_wide_vjp": (c_int, [c_void_p] * 8 + [c_int64, c_void_p]),
    "pita_egnn_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "pita_egnn_edm": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "pita_egnn_jvp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "pita_egnn_vjp": (c_int, [c_void_p] * 9 + [c_int64, c_void_p]),
    "pita_egnn_div_directions": (c_int, [c_void_p]),
    "pita_egnn_div_accumulate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                         c_int64, c_void_p]),
    "pita_egnn_jacobian_trace": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "pita_egnn_div_work": (c_int, [c_void_p, POINTER(c_double), POINTER(c_double)]),
    "pita_fk_assemble": (c_int, [c_void_p] * 10 + [c_float, c_float, c_void_p, c_void_p, c_float, c_float, c_void_p] +
                         [c_void_p] * 6 + [c_int64, c_int, c_void_p]),
    "pita_quantile_clamp": (c_int, [c_void_p, c_int64, c_int64, c_float, c_void_p]),
    "pita_egnn_sampler_run": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_uint64, c_uint64,
                                      c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "pita_egnn_sampler_work": (c_int, [c_void_p, c_int64, POINTER(c_double), POINTER(c_double)]),
    "pita_egnn_sampler_mapping": (c_int, [c_void_p, c_int64, POINTER(c_int), POINTER(c_int64), POINTER(c_int64)]),
    "pita_mlp_create": (c_int, [POINTER(c_void_p), POINTER(MlpConfig), c_void_p, c_int64, c_void_p]),
    "pita_mlp_destroy": (c_int, [c_void_p]),
    "pita_mlp_num_weights": (c_int64, [POINTER(MlpConfig)]),
    "pita_mlp_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "pita_mlp_sampler_run": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_uint64, c_uint64, c_int64,
                                     c_int, c_int, c_int, c_void_p, c_void_p]),
    "pita_em_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_float, c_float, c_uint64,
                             c_uint64, c_int64, c_int, c_void_p, c_void_p]),
    "pita_moments": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "pita_moments4": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "pita_histogram": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "pita_prior_sample": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_uint64, c_uint64, c_int,
                                  c_void_p]),
    "pita_remove_mean": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p]),
    "pita_fill_normal": (c_int, [c_void_p, c_int64, c_int, c_int, c_uint64, c_uint64, c_int64, c_void_p]),
    "pita_edm_scale_input": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "pita_edm_combine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "pita_energy_theta": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "pita_mala_propose": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_uint64,
                                  c_uint64, c_void_p, c_int64, c_void_p]),
    "pita_mala_accept": (c_int, [c_void_p] * 7 + [c_int64, c_int, c_int, c_void_p, c_uint64, c_uint64, c_void_p, c_int64,
                                                  c_int, c_void_p, c_void_p]),
    "pita_mala_adapt": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "pita_resample_workspace_bytes": (c_size_t, [c_int64]),
    "pita_systematic_resample": (c_int, [c_void_p, c_int64, c_double, c_void_p, c_void_p, c_void_p]),
    "pita_gather_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "pita_count_runs": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
}

EXPORTS = tuple(_PROTOS)
_lib = None


def lib():
    """Load (once) and return the shared library; raises PitaHipError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PitaHipError(
                f"{LIB_PATH} not found: build it with `python -m pita_amd.build` (hipcc --offload-arch=gfx950). "
                "pita_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.pita_abi_version() != ABI_VERSION:
            raise PitaHipError("libpita_hip.so ABI version mismatch")
        _lib = L
    return _lib


def last_error() -> str:
    buf = ctypes.create_string_buffer(512)
    lib().pita_last_error(buf, 512)
    return buf.value.decode()


def check(rc: int, what: str = ""):
    if rc != 0:
        raise PitaHipError(f"{what or 'libpita_hip'} failed (code {rc}): {last_error()}")


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def dev_tensor(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    """Validate a device tensor argument: HIP device, expected dtype, contiguous (copy if needed)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise PitaHipError(f"{name}: tensor is on {t.device}; pita_amd runs on the GPU only (no CPU fallback)")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.detach().contiguous()


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()
