"""A/B timing of pita_egnn_sampler_run between two builds of libpita_hip.so (development aid).
usage: python tools/ab_sampler.py pita_amd/libpita_hip.so pita_amd/libpita_hip_old.so"""
import ctypes, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pita_amd
from pita_amd._lib import EgnnConfig
from ctypes import POINTER, c_void_p, c_int, c_int64, c_uint64

w = dict(np.load("tests/golden/egnn_weights_seed12345.npz"))
flat = np.concatenate([np.asarray(v, dtype=np.float32).reshape(-1) for v in w.values()])
sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
N, B, S = 1000, 65536, 100
tab = pita_amd.sde_integration.build_step_table(sched, gam, torch.linspace(1.0, 0.0, N + 1)[:-1], 1.0 / N, 1.0, 1.0).cuda()
x0 = pita_amd.Prior(scale=69.28, n_particles=13, spatial_dim=3, seed=1).sample(B)
libs = []
for path in sys.argv[1:]:
    L = ctypes.CDLL(path)
    L.pita_egnn_create.argtypes = [POINTER(c_void_p), POINTER(EgnnConfig), c_void_p, c_int64]
    L.pita_egnn_sampler_run.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_uint64, c_uint64, c_int64, c_int, c_void_p, c_void_p, c_void_p]
    cfg = EgnnConfig(13, 3, 32, 3, 2, 1, 1, 15.0, 0, 1)
    h = c_void_p()
    assert L.pita_egnn_create(ctypes.byref(h), ctypes.byref(cfg), flat.ctypes.data_as(c_void_p), len(flat)) == 0
    libs.append((path, L, h))
sp = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    for path, L, h in libs:
        x = x0.clone()
        L.pita_egnn_sampler_run(h, x.data_ptr(), B, tab.data_ptr(), S, None, 3, 0, 0, 1, None, None, sp); torch.cuda.synchronize()
        t0 = time.perf_counter()
        L.pita_egnn_sampler_run(h, x.data_ptr(), B, tab.data_ptr(), S, None, 3, 0, 0, 1, None, None, sp); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{path}: {dt*1e3:.2f} ms per {S}-step launch -> {B*S/dt:.3e} walker-steps/s", flush=True)
