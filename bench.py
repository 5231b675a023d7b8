#!/usr/bin/env python3
"""bench.py -- walker-steps/s of the fused HIP annealed-SDE sampler (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config lj13|dw4|aldp22|lj55]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload (BASELINE.json configs[2], the configuration the metric is quoted on): LJ13 (13 particles x 3D), EGNN
score net (hidden 32 x 3 layers, tanh, attention, temperature conditioned; seed-12345 initialisation, i.e. the
reference's own untrained init), Elucidating schedule (sigma_min 0.05, sigma_max 80, rho 7), gamma = 4/3, beta = 1,
diffusion_scale 1, NOT-debiased reverse VE-SDE, resampling off, 65 536 walkers PER GPU (weak scaling), synthetic walkers
from the mean-free prior.  One "step" = one Euler-Maruyama step of all walkers: EDM-preconditioned EGNN forward + drift +
noise + update + remove_mean, all inside the fused kernel `egnn_kernel<...,SAMPLER>`; the K timed steps run as K/c
launches of c = gcd(K, W) steps each (the warm-up uses the same launch size, so every launch of the kernel in this
process is equal and the rocprof average is comparable).  After the timed region separately timed loops give the
roofline of the target's log-density+force kernel and the debiased (Feynman-Kac) regime.

`--config` selects the other GPU configurations of BASELINE.json at their per-GPU shard (dw4: 65 536 walkers; aldp22:
4 096 = 16 384 / 4 GPUs; lj55: 32 768 = 262 144 / 8 GPUs): same measurement, same JSON fields.

`--gpus N` without a torchrun environment starts the N rank processes itself (python -m torch.distributed.run) BEFORE
this process touches the GPU, and relays rank 0's JSON line.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA16_TFLOPS = 2500.0      # MI355X_MICROARCH.md: BF16/FP16 MFMA dense peak (~2.5 PF)
PEAK_F32_VALU_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 vector peak
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak
ACHIEVABLE_HBM_GBS = 6300.0      # MI355X_MICROARCH.md: measured float4 copy (79 % of spec)
MFMA16_FLOP = 2 * 32 * 32 * 16   # one v_mfma_f32_32x32x16_{bf16,f16}
MFMA32_FLOP = 2 * 32 * 32 * 2    # one v_mfma_f32_32x32x2_f32
NOMINAL_CLOCK_HZ = 2.4e9

# name -> particles, dims, walkers per GPU, sigma_min, target kind
CONFIGS = {
    "lj13": dict(n=13, d=3, walkers=65536, sigma_min=0.05, target="lj",
                 label="BASELINE configs[2]: LJ13, 65 536 walkers on 1 GPU"),
    "dw4": dict(n=4, d=2, walkers=65536, sigma_min=0.01, target="dw",
                label="BASELINE configs[1]: DW4, 65 536 walkers on 1 GPU"),
    "aldp22": dict(n=22, d=3, walkers=4096, sigma_min=0.01, target="ff",
                   label="BASELINE configs[3]: 22-atom force field, 16 384 walkers over 4 GPUs = 4 096 per GPU"),
    "lj55": dict(n=55, d=3, walkers=32768, sigma_min=0.05, target="lj",
                 label="BASELINE configs[4]: LJ55, 262 144 walkers over 8 GPUs = 32 768 per GPU"),
}


def egnn_algorithmic_flop(n, H=32, L=3):
    """per layer E(2H+2)H + 2EH^2 + 2EH + n 3H^2 MAC, x L, + 2 n 2 H; flop = 2 MAC
    (LJ13: 2 098 304 MAC = 4.197 MFLOP, counted on the reference's un-split formulation)."""
    E = n * (n - 1)
    mac = L * (E * (2 * H + 2) * H + 2 * E * H * H + 2 * E * H + n * 3 * H * H) + 2 * n * 2 * H
    return 2.0 * mac


def target_bytes(n, d):
    return (2 * n * d + 1) * 4   # read x, write force, write logp


def pmc_summary():
    """Per-walker-step counter summary of the sampler kernel (profiles/pmc_sampler_current.json, produced by
    tools/pmc_summarise.py from separate rocprofv3 --pmc passes over this same command; collected and corrected as
    MI355X_MICROARCH.md prescribes).  PMC collection cannot run inside the timed bench, so the live JSON carries the
    last committed measurement scaled to this run's launch size."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_sampler_current.json")) as f:
            return json.load(f)
    except Exception:
        return None


def self_launch(args):
    """--gpus N without a torchrun environment: start the N ranks as a child job.  Nothing in this process has
    touched the GPU yet (torch is not even imported); the child job's stdout is relayed unchanged."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def build_model(pa, n, d, seed=12345, precision=None):
    import torch

    torch.manual_seed(seed)
    return pa.EGNN_dynamics(n, d, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                            condition_time=True, condition_temperature=True, agg="sum", precision=precision)


def make_target(pa, cfg, dev):
    n, d = cfg["n"], cfg["d"]
    if cfg["target"] == "lj":
        return pa.LennardJonesEnergy(n * d, n, d, device=dev)
    if cfg["target"] == "dw":
        return pa.MultiDoubleWellEnergy(n * d, n, d, device=dev)
    from pita_amd.alp_energy import ForceFieldEnergy
    from tests._synthetic import synthetic_peptide  # synthetic tables: amber14 parameters are not in the reference tree

    tabs, _ = synthetic_peptide(n)
    return ForceFieldEnergy(tabs, n_particles=n, temperature=300.0, data_normalization_factor=0.1640, cutoff=2.0,
                            device=dev)


def cpu_model():
    """Model string of the host CPU (the CPU baseline states core count AND CPU model)."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine()


def cpu_baseline(cfg, n_walkers, n_steps, seed=12345):
    """The oracle (torch-CPU restatement of the reference's sampler, kind = "port") timed on this
    host's cores on a bounded sample of the same workload."""
    import torch

    from oracle import pita_oracle as O

    torch.manual_seed(seed)
    import pita_amd

    n, d = cfg["n"], cfg["d"]
    net = build_model(pita_amd, n, d, seed)
    w = {k: v.detach().clone() for k, v in net.state_dict().items()}
    bb = lambda cn, xs, b: O.egnn_forward(w, cn, xs, b, n, d)
    sched, gam = O.Elucidating(cfg["sigma_min"], 80.0, 7), O.GammaConstant(4 / 3)
    gen = torch.Generator().manual_seed(1)
    x1 = O.prior_from_noise(torch.randn(n_walkers, n * d, generator=gen), O.prior_scale(sched, gam, 1.0), n, d)
    cfg_run = O.IntegratorConfig(num_integration_steps=n_steps, end_resampling_step=n_steps)
    drift = lambda t, xc: O.f_not_debiased(bb, sched, gam, t, xc, 1.0)
    noise_fn = lambda i, shp: torch.randn(shp, generator=gen)
    all_threads = torch.get_num_threads()
    probe_cfg = O.IntegratorConfig(num_integration_steps=8, end_resampling_step=8)
    with torch.no_grad():
        # torch's default (all hardware threads) oversubscribes these small tensors: probe it against 16 threads on a few
        # steps, then time the full sample once with the faster setting
        probe = {}
        for nthr in sorted({all_threads, min(16, all_threads)}, reverse=True):
            torch.set_num_threads(nthr)
            O.integrate_sde(O.IntegratorConfig(num_integration_steps=2, end_resampling_step=2), x1, drift, sched.g,
                            noise_fn, n, d)  # warm-up
            t0 = time.perf_counter()
            O.integrate_sde(probe_cfg, x1, drift, sched.g, noise_fn, n, d)
            probe[nthr] = time.perf_counter() - t0
        nthr = min(probe, key=probe.get)
        torch.set_num_threads(nthr)
        t0 = time.perf_counter()
        O.integrate_sde(cfg_run, x1, drift, sched.g, noise_fn, n, d)
        dt = time.perf_counter() - t0
    torch.set_num_threads(all_threads)
    return {"value": n_walkers * n_steps / dt, "unit": "walker-steps/s", "cores": nthr, "cpu_model": cpu_model(),
            "logical_cpus": os.cpu_count(),
            "kind": "port", "sample": f"oracle integrate_sde, {n}x{d}D EGNN h32x3, {n_walkers} walkers x {n_steps} steps, "
            f"torch-CPU fp32, {dt:.1f} s with {nthr} threads (host has {os.cpu_count()} logical CPUs)"}


def debiased_leg(pita_amd, net, cfg, dev, B, with_cpu):
    """Secondary number: the debiased Feynman-Kac regime (PITA's default; sdes.py:151-239): drift of x and of the
    log-weights through the exact Jacobian trace (pita_egnn_jacobian_trace: one launch with the primal that writes the
    primal cache, tangent-only launches that stream it) + 1 reverse-mode launch (pita_egnn_vjp, which also returns the
    h-derivative term) + assembly + quantile clamp, then the EM update."""
    import copy

    import numpy as np
    import torch

    from pita_amd.energy_net import EnergyNet

    n, d = cfg["n"], cfg["d"]
    sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=cfg["sigma_min"], sigma_max=80.0, rho=7)
    gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
    sde = pita_amd.VEReverseSDE(noise_schedule=sched, score_net=pita_amd.ScoreNet(net),
                                energy_net=EnergyNet(copy.deepcopy(net)), debias_inference=True)
    x = pita_amd.Prior(scale=3.0, n_particles=n, spatial_dim=d, device=dev, seed=7).sample(B)
    t = torch.tensor(0.5)  # host scalar, as the integrator passes the step time
    L = pita_amd._lib.lib()

    def step():
        terms = sde.f(t, x, 1.0, gam, None, None, clamp_chunk=512)
        L.pita_em_step(x.data_ptr(), terms.drift_X.data_ptr(), 0, B, n, d, 1e-3, 0.1, float(np.sqrt(1e-3)), 1, 0, 0, 1, 0,
                       pita_amd._lib.stream_ptr(dev))
        return terms

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    K = max(1, L.pita_egnn_div_directions(net._native(dev)))
    # the divergence launches dominate the step: time them alone and price them against the 16-bit matrix peak
    ht = torch.full((B,), float(sched.h(torch.tensor(0.5))), device=dev)
    bt = torch.ones(B, device=dev)
    net.jacobian_trace(ht, x, bt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    net.jacobian_trace(ht, x, bt)
    e1.record()
    torch.cuda.synchronize()
    tr_ms = e0.elapsed_time(e1)
    import ctypes

    a, b = ctypes.c_double(), ctypes.c_double()
    pita_amd._lib.check(L.pita_egnn_div_work(net._native(dev), ctypes.byref(a), ctypes.byref(b)), "pita_egnn_div_work")
    tf = B * a.value * MFMA16_FLOP / (tr_ms * 1e-3) / 1e12
    pm = pmc_summary() or {}
    pk = pm.get(f"divergence_{cfg['name']}")
    out = {"metric": f"walker-steps/s, debiased (Feynman-Kac) regime, {n}x{d}D", "value": B / dt, "walkers": B,
           "ms_per_step": dt * 1e3,
           "launches_per_step": "exact trace of J_x D (1 launch with the primal + tangent-only launches from the primal "
                                "cache, each followed by its repair pass), 1 reverse-mode launch, assembly, clamp, update",
           "roofline": {"kernel": "egnn_div_fast_kernel + egnn_div_tangent_(shared_)kernel (pita_egnn_jacobian_trace)",
                        "bound": "mfma", "achieved": tf, "peak": PEAK_MFMA16_TFLOPS, "unit": "TFLOP/s",
                        "frac": tf / PEAK_MFMA16_TFLOPS, "ms_per_trace": tr_ms,
                        "executed_mfma_flop_per_walker": a.value * MFMA16_FLOP,
                        "executed_f32_mfma_flop_per_walker": b.value * MFMA32_FLOP,
                        "valu_issue_frac": pk.get("valu_issue_frac") if pk and pk.get("walkers") == B else None,
                        "pmc_kernels": pk.get("kernels") if pk and pk.get("walkers") == B else None,
                        "note": "the tangent-only launches trade matrix flops for streams of the primal cache (~12 GB per "
                                "launch at 65 536 LJ13 walkers, one stream per 16 directions through an LDS ring); they "
                                "are bound by LDS bandwidth and vector issue, not by the matrix pipe: DESIGN.md 4.5"}}
    assert 0.0 < out["roofline"]["frac"] <= 1.0
    out["_x48"] = x[:48].cpu() if with_cpu else None
    # the leg runs right in front of the timed region: its handles (the energy net's copy with its reverse-mode scratch) are
    # destroyed by the caller AFTER that region -- a hipFree is synchronous and can idle the device for milliseconds, which
    # restarts the clock ramp the leg order exists to avoid (one run in six started its timed launches at 5.8 instead of 5.2 ms)
    out["_keepalive"] = sde
    return out


def e2e_legs(pita_amd, net, cfg, dev, B, n_plain, n_default, chunk=512):
    """End-to-end numbers at the metric's own length, through the plug-in class (never inside `value`):
    `not_debiased`   ONE WeightedSDEIntegrator.integrate_sde (sde_integration.py:98-212): B walkers x n_plain steps from
                     Prior.sample, per-step moments on, resampling off, the f16 repair pass as it falls along the real schedule;
    `default_regime` n_default steps with debias_inference=True, resampling_interval=1, inference chunks of 512 with their own
                     0.9-quantile clamp, resample_at_end, 5 adaptive MALA steps at dt = 1e-13 (configs/model/energytemp.yaml:72-84,
                     experiment/lj13.yaml:24-42; quirk Q9)."""
    import copy

    import torch

    from pita_amd.energy_net import EnergyNet

    n, d = cfg["n"], cfg["d"]
    sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=cfg["sigma_min"], sigma_max=80.0, rho=7)
    gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
    energy = make_target(pita_amd, cfg, dev)
    scale = float((sched.h(torch.tensor(1.0)) / gam.gamma(torch.tensor(1.0))) ** 0.5)
    out = {}
    for name, N, debias in (("not_debiased", n_plain, False), ("default_regime", n_default, True)):
        if N <= 0:
            continue
        if debias and cfg["target"] not in ("lj",):
            continue  # the reference's default regime with MALA needs a molecule target with forces (sde_integration.py:397)
        sde = pita_amd.VEReverseSDE(noise_schedule=sched, score_net=pita_amd.ScoreNet(net),
                                    energy_net=EnergyNet(copy.deepcopy(net)) if debias else None, debias_inference=debias)
        kw = dict(resampling_interval=1, batch_size=chunk, resample_at_end=True, post_mcmc_steps=5, adaptive_mcmc=True,
                  dt_negative_time=1e-13) if debias else dict(resampling_interval=-1, post_mcmc_steps=0)
        integ = pita_amd.WeightedSDEIntegrator(sde=sde, num_integration_steps=N, start_resampling_step=0,
                                               end_resampling_step=int(0.9 * N) if debias else N, num_negative_time_steps=0,
                                               **kw)
        x1 = pita_amd.Prior(scale=scale, n_particles=n, spatial_dim=d, device=dev, seed=12345).sample(B)
        if debias:  # a short run first: handles, primal cache, resampling buffers
            warm = pita_amd.WeightedSDEIntegrator(sde=sde, num_integration_steps=2, start_resampling_step=0, end_resampling_step=2,
                                                  num_negative_time_steps=0, **kw)
            warm.integrate_sde(x1.clone(), energy, gam, inverse_temperature=1.0)
        else:
            integ.integrate_sde(x1.clone(), energy, gam, inverse_temperature=1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = integ.integrate_sde(x1, energy, gam, inverse_temperature=1.0)
        host = time.perf_counter() - t0  # the host's share: integrate_sde has returned, the device may still be working
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        xf = res[0]
        out[name] = {"walkers": B, "steps": N, "seconds": dt, "host_seconds": host, "ms_per_step": dt * 1e3 / N,
                     "value": B * N / dt,
                     "unit": "walker-steps/s", "finite": bool(torch.isfinite(xf).all()),
                     "what": ("one WeightedSDEIntegrator.integrate_sde from Prior.sample: per-step moments on, resampling off, "
                              "no MCMC") if not debias else
                             ("one integrate_sde in the reference's default regime: debias_inference, resampling every step "
                              "inside [0, 0.9 N), chunks of 512 with their own quantile clamp, resample_at_end, 5 adaptive "
                              "MALA steps (included in `seconds`)")}
        if debias:
            out[name]["distinct_parents_last_event"] = int(res[2][-1]) if len(res[2]) else None
            out[name]["mala_acceptance"] = [float(a) for a in res[4]]
    return out


def small_batch_legs(pita_amd, net, cfg, dev, sizes=(512, 2048, 5000, 16384), n_plain=1000, n_default=100):
    """The reference's OWN operating points (never inside `value`): PITA generates num_eval_samples = 2 048 walkers
    (configs/experiment/lj13.yaml:32; 5 000 in model/energytemp.yaml:69, num_samples_to_generate_per_epoch 2 000 at :122) in
    inference chunks of 512 (lj13.yaml:27) -- far fewer walkers than one MI355X holds.  Per batch size: the whole
    integrate_sde in both regimes (e2e_legs: 1 000 not-debiased steps; 100 steps of the default regime with chunks of 512),
    the fraction of the full-batch rate, and how the fused sampler's mapping fills the chip at that size."""
    import torch

    out = {"sizes": {}, "what": "whole WeightedSDEIntegrator.integrate_sde per batch size, both regimes; rates relative to the "
                                "same legs at the config's full batch are in `frac_of_full_batch` when `e2e` ran"}
    for B in sizes:
        legs = e2e_legs(pita_amd, net, cfg, dev, B, n_plain, n_default, chunk=min(512, B))
        g, waves, slots = net.sampler_mapping(B, dev)
        legs["sampler_mapping"] = {"walkers_per_wave_group": g, "waves": waves, "resident_wave_slots": slots,
                                   "wave_slot_fill": waves / slots}
        for leg in legs.values():
            leg.pop("what", None)
        out["sizes"][str(B)] = legs
        torch.cuda.synchronize()
    return out


def e2e_multi_rank(pita_amd, net, cfg, dev, B, world, n_steps, gather_fn):
    """Several ranks: ONE WeightedSDEIntegrator.integrate_sde over the GLOBAL batch (world x B walkers from Prior.sample;
    every rank integrates its slice, sde_integration.py:227-233, one final all-gather), timed per rank behind the timed
    region -- the wall seconds of every rank, so that a weak-scaling loss of the whole call can be attributed (slow rank,
    the collective, or neither) without a second run.  Never inside `value`."""
    import torch

    n, d = cfg["n"], cfg["d"]
    sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=cfg["sigma_min"], sigma_max=80.0, rho=7)
    gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
    energy = make_target(pita_amd, cfg, dev)
    scale = float((sched.h(torch.tensor(1.0)) / gam.gamma(torch.tensor(1.0))) ** 0.5)
    sde = pita_amd.VEReverseSDE(noise_schedule=sched, score_net=pita_amd.ScoreNet(net), debias_inference=False)
    integ = pita_amd.WeightedSDEIntegrator(sde=sde, num_integration_steps=n_steps, start_resampling_step=0,
                                           end_resampling_step=n_steps, num_negative_time_steps=0, resampling_interval=-1,
                                           post_mcmc_steps=0)
    x1 = pita_amd.Prior(scale=scale, n_particles=n, spatial_dim=d, device=dev, seed=12345).sample(world * B)
    integ.integrate_sde(x1.clone(), energy, gam, inverse_temperature=1.0)  # warm-up: handles, step table, communicator
    torch.cuda.synchronize()
    torch.distributed.barrier()
    t0 = time.perf_counter()
    res = integ.integrate_sde(x1, energy, gam, inverse_temperature=1.0)
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    walls = gather_fn(torch.tensor([mine], dtype=torch.float64)).reshape(world).tolist()
    slow = max(walls)
    return {"what": "one integrate_sde over the global batch (rank slices + one final all-gather), not debiased, per-step "
                    "moments on; wall seconds of every rank",
            "global_walkers": world * B, "steps": n_steps, "per_rank_wall_s": walls, "seconds": slow,
            "ms_per_step": 1e3 * slow / n_steps, "value": world * B * n_steps / slow, "unit": "walker-steps/s",
            "gathered_rows": int(res[0].shape[0]), "finite": bool(torch.isfinite(res[0]).all())}


def debiased_cpu_baseline(net, cfg, xc):
    """The oracle's debiased drift (autograd + vmap(jacrev)) on a bounded sample, timed on the host cores."""
    import torch

    from oracle import pita_oracle as O

    n, d = cfg["n"], cfg["d"]
    w = {k: v.detach().clone() for k, v in net.state_dict().items()}
    bb = lambda cn, xs, b: O.egnn_forward(w, cn, xs, b, n, d)
    osched, ogam = O.Elucidating(cfg["sigma_min"], 80.0, 7), O.GammaConstant(4 / 3)
    nb = xc.shape[0]
    torch.set_num_threads(min(16, torch.get_num_threads()))
    t0 = time.perf_counter()
    O.f_debiased(bb, bb, osched, ogam, torch.tensor(0.5), xc, 1.0)
    dtc = time.perf_counter() - t0
    return {"value": nb / dtc, "unit": "walker-steps/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(), "kind": "port",
            "sample": f"oracle f_debiased (autograd + vmap(jacrev)), {nb} walkers x 1 step, {dtc:.1f} s"}


def force_roofline(pita_amd, cfg, energy, x, dev, reps):
    """The target's log-density + force kernel, separately timed at this config's batch: HBM roofline for LJ13 / DW4 /
    the 22-atom force field (algorithmic bytes = read x + write force + write logp), fp32-VALU fraction for LJ55
    (AI 31 flop/B is past the ridge)."""
    import torch

    n, d, B = cfg["n"], cfg["d"], x.shape[0]

    def timed(fn, k):
        """us per call over >= k back-to-back calls, after ~40 ms of the same calls: short launches measured from a cold
        start run at the clock the chip is still ramping through (the 2^21-walker LJ13 launch: 183 us cold, 150 us after
        40 ms of work), which says nothing about the kernel."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        per_call_ms = max(e0.elapsed_time(e1) / 3, 1e-3)
        for _ in range(min(4000, int(40.0 / per_call_ms) + 1)):
            fn()
        k = max(k, min(4000, int(20.0 / per_call_ms) + 1))
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / k

    us = timed(lambda: energy(x, return_force=True), reps)
    nbytes = target_bytes(n, d)
    gbs = B * nbytes / (us * 1e-6) / 1e9
    a, b = torch.empty(B * n * d, device=dev), torch.empty(B * n * d, device=dev)
    us_copy = timed(lambda: b.copy_(a), reps)
    out = {"kernel": {"lj": f"pita_lj_logp_force (LJ{n} logp+force)", "dw": "pita_dw_logp_force (DW4 logp+force)",
                      "ff": "pita_ff_logp_force (22-atom force field, synthetic tables, cutoff 2 nm, no GB)"}[cfg["target"]],
           "walkers": B, "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
           "frac_of_measured_achievable_6300": gbs / ACHIEVABLE_HBM_GBS,
           "algorithmic_bytes_per_walker_eval": nbytes, "us_per_launch": us, "walker_evals_per_s": B / (us * 1e-6),
           "launches": reps, "same_bytes_device_copy_us": us_copy,
           "same_bytes_device_copy_GBs": 2 * B * n * d * 4 / (us_copy * 1e-6) / 1e9,
           "note": "stand-alone launches through the plug-in class (output tensors allocated per call)"}
    # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x 2 streaming correction + WRITE_SIZE): the force
    # kernel of this config is the one launched most often under that name in the profiled bench run
    pm = pmc_summary()
    ks = (pm or {}).get(f"force_{cfg['name']}_kernels", {})
    pat = {"lj": "lj13_kernel<2" if n == 13 else "ring_energy", "dw": "ring_energy", "ff": "ff_"}[cfg["target"]]
    ks = {k: v for k, v in ks.items() if pat in k}
    out["traffic"] = None
    if ks and B == cfg["walkers"]:
        v = max(ks.values(), key=lambda e: e["launches"])
        out["traffic"] = v["fetch_bytes_x2_streaming_correction"] + v["write_bytes"]
        out["traffic_source"] = "committed PMC (profiles/pmc_sampler_current.json); not collected in this run"
    if cfg["target"] == "lj" and n == 55:  # VALU-bound: SURVEY 8(d) 1 485 pairs x 28 + 55 x 9 flop per walker-eval
        flop = 1485 * 28 + 55 * 9
        tf = B * flop / (us * 1e-6) / 1e12
        out.update({"bound": "valu", "achieved": tf, "peak": PEAK_F32_VALU_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / PEAK_F32_VALU_TFLOPS, "hbm_GBs_for_information": gbs,
                    "algorithmic_flop_per_walker_eval": flop})
        out.pop("frac_of_measured_achievable_6300")
    if cfg["target"] in ("lj", "dw"):  # the kernel itself: C-ABI calls on preallocated outputs
        L = pita_amd._lib.lib()
        sp = pita_amd._lib.stream_ptr(dev)
        lp, fo = torch.empty(B, device=dev), torch.empty_like(x)
        if cfg["target"] == "lj":
            raw = lambda: L.pita_lj_logp_force(x.data_ptr(), lp.data_ptr(), fo.data_ptr(), B, n, d, 1.0, 1.0, 1e-6, 1.0, 1.0, 1.0, sp)
        else:
            raw = lambda: L.pita_dw_logp_force(x.data_ptr(), lp.data_ptr(), fo.data_ptr(), B, n, d, 1.0, energy.a, energy.b,
                                               energy.c, energy.offset, sp)
        us_raw = timed(raw, reps)
        gbs_raw = B * nbytes / (us_raw * 1e-6) / 1e9
        out.update({"us_per_launch": us_raw, "walker_evals_per_s": B / (us_raw * 1e-6), "hbm_GBs_for_information": gbs_raw,
                    "us_per_launch_through_plugin_class": us, "note": "C-ABI calls on preallocated outputs"})
        if out["bound"] == "hbm":
            out.update({"achieved": gbs_raw, "frac": gbs_raw / PEAK_HBM_GBS,
                        "frac_of_measured_achievable_6300": gbs_raw / ACHIEVABLE_HBM_GBS})
        else:
            tf = B * out["algorithmic_flop_per_walker_eval"] / (us_raw * 1e-6) / 1e12
            out.update({"achieved": tf, "frac": tf / PEAK_F32_VALU_TFLOPS,
                        "frac_of_plain_fma_rate_78.6": tf / (PEAK_F32_VALU_TFLOPS / 2),
                        "note": "C-ABI calls on preallocated outputs.  peak = 157.3 TFLOP/s counts packed FMAs at full "
                                "rate; they issue at half rate on this chip (tools/ubench/isa_rates.hip; a packed pair loop "
                                "measured slower, DESIGN.md 4.2), so the plain-FMA rate is given beside it"})
    if cfg["target"] == "lj" and n == 13:
        L = pita_amd._lib.lib()
        sp = pita_amd._lib.stream_ptr(dev)
        BIG = 1 << 21
        xbig = x.repeat(BIG // B + 1, 1)[:BIG].contiguous()
        lpb, fob = torch.empty(BIG, device=dev), torch.empty_like(xbig)
        us_big = timed(lambda: L.pita_lj_logp_force(xbig.data_ptr(), lpb.data_ptr(), fob.data_ptr(), BIG, 13, 3, 1.0, 1.0,
                                                    1e-6, 1.0, 1.0, 1.0, sp), 20)
        gbs_big = BIG * nbytes / (us_big * 1e-6) / 1e9
        out["large_batch"] = {"walkers": BIG, "us_per_launch": us_big, "achieved": gbs_big, "frac": gbs_big / PEAK_HBM_GBS,
                              "frac_of_measured_achievable_6300": gbs_big / ACHIEVABLE_HBM_GBS}
        # the force kernel where production uses it per step: negative-time descent (sde_integration.py:353-360), all
        # steps of a launch with the walkers resident in LDS (pita_lj_descent).  HBM sees one read + one write of x
        # per LAUNCH: the rate below is walker-evals/s; its "algorithmic-equivalent" GB/s is NOT an HBM-roofline
        # fraction (the bytes do not move) and carries no frac
        S_DESC = 1000
        xd = x.clone()

        def descent(noise_scale):
            pita_amd._lib.check(L.pita_lj_descent(xd.data_ptr(), 0, B, 13, 3, 1.0, 1.0, 1e-6, 1.0, 1.0, 1.0, S_DESC, 1e-7,
                                                  noise_scale, math.sqrt(2e-7), 1, 0, 0, 1, sp), "pita_lj_descent")

        us_desc = timed(lambda: descent(0.0), 2) / S_DESC
        us_ula = timed(lambda: descent(1.0), 2) / S_DESC
        assert torch.isfinite(xd).all(), "descent produced non-finite walkers"
        out["in_descent_loop"] = {"kernel": "lj13_descent_kernel (force + update + centring, walkers LDS-resident)",
                                  "steps_per_launch": S_DESC, "us_per_step": us_desc,
                                  "walker_evals_per_s": B / (us_desc * 1e-6),
                                  "algorithmic_equivalent_GBs_bytes_do_not_move": B * nbytes / (us_desc * 1e-6) / 1e9,
                                  "hbm_bytes_per_launch": 2 * B * 39 * 4, "us_per_step_with_langevin_noise": us_ula}
    return out


def resample_exchange_leg(comm, x, every, B, world, rank, ids_fn, sync, events=3):
    """Optional leg (--resample-every K; NOT part of the metric): what ONE global resampling event costs at this shard
    when a trajectory resamples every K steps (PITA's default regime, sde_integration.py:283-297): all-gather of the B
    log-weights per rank, the identical systematic resampling on every rank (``ids_fn``), and _Comm.exchange_rows -- one
    uneven all_to_all_single that carries each rank the distinct parents it does not hold.  Times are per event on
    this rank; rows_received counts the walkers that crossed ranks."""
    import torch

    gen = torch.Generator().manual_seed(77)
    per_event, rows, sent, splits = [], [], [], None
    for e in range(events + 1):  # the first event is a warm-up (communicator set-up)
        a = (0.5 * torch.randn(B, generator=gen)).to(x.device)  # one step's log-weights: spread ~0.5 (fixture: 0.3-1)
        comm.rows_received = comm.rows_sent = 0
        sync()
        t0 = time.perf_counter()
        ag = comm.all_gather(a)
        ids = ids_fn(ag, 0.37 + 0.01 * e)
        x = comm.exchange_rows(x, ids, B)
        sync()
        if e > 0:
            per_event.append(time.perf_counter() - t0)
            rows.append(int(comm.rows_received))
            sent.append(int(comm.rows_sent))
            splits = getattr(comm, "last_splits", None)
    return {"every": every, "events_timed": events, "ms_per_event_this_rank": [round(1e3 * t, 4) for t in per_event],
            "rows_received_from_other_ranks": rows, "rows_per_rank": B,
            "bytes_received_per_event": [r * x.shape[1] * 4 for r in rows],
            # what this rank actually put on the wire per event: the uneven all_to_all_single's input splits to OTHER ranks
            # (round 6; the reference all-gathers (world - 1) * B rows to every rank every step, sde_integration.py:248-258)
            "rows_sent_to_other_ranks": sent, "bytes_sent_per_event": [r * x.shape[1] * 4 for r in sent],
            "log_weight_allgather_bytes_per_event": 4 * B * (world - 1),
            "reference_allgather_bytes_per_event": (world - 1) * B * x.shape[1] * 4,
            "last_event_rows_by_peer": splits,
            "amortised_ms_per_step": 1e3 * sum(per_event) / len(per_event) / max(every, 1),
            "note": "log-weight all-gather + global systematic resampling + one uneven all_to_all_single "
                    "(_Comm.exchange_rows); not inside the timed region, not part of `value`"}, x


def rank_breakdown(world, gather_fn, wall_s, launches_s, allgather_s):
    """Per-rank attribution of a multi-rank timed region: [wall seconds barrier-to-barrier, seconds inside the
    sampler launches (HIP events), seconds of the final all-gather (HIP events)] of every rank, so a loss against the
    one-GPU number can be pinned on a slow rank, on the collective or on neither (launch gaps / host)."""
    import torch

    mine = torch.tensor([wall_s, launches_s, allgather_s], dtype=torch.float64)
    allr = gather_fn(mine).reshape(world, 3).tolist()
    return {"wall_s": [r[0] for r in allr], "sampler_launches_s": [r[1] for r in allr],
            "final_allgather_ms": [1e3 * r[2] for r in allr],
            "slowest_rank": int(max(range(world), key=lambda r: allr[r][0])),
            "max_over_min_wall": max(r[0] for r in allr) / max(min(r[0] for r in allr), 1e-12)}


def rank_identity(world, rank, dev, rehearsal):
    """What a multi-rank line says about the ranks it ran on, measured rather than asserted: `rccl_ranks_seen` = an
    all_reduce(SUM) of ones over the process group, read back; `devices` = every rank's device name and PCI bus id,
    all-gathered (two ranks on one GPU, or a rank on the wrong device, show here).  `dev` None: gloo dry run."""
    import torch

    ones = torch.ones(1, dtype=torch.float32, device=dev if (dev is not None and not rehearsal) else "cpu")
    torch.distributed.all_reduce(ones)
    if dev is not None:
        pr = torch.cuda.get_device_properties(dev)
        bus = getattr(pr, "pci_bus_id", None)
        mine = {"rank": rank, "device": pr.name, "pci_bus_id": bus if bus is not None else None,
                "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
                "cuda_index": dev.index, "host": socket.gethostname()}
    else:
        mine = {"rank": rank, "device": "cpu (dry run)", "pci": None, "cuda_index": None, "host": socket.gethostname()}
    devs = [None] * world
    torch.distributed.all_gather_object(devs, mine)
    return {"rccl_ranks_seen": int(round(float(ones.item()))), "devices": devs,
            "distinct_devices": len({(d["host"], d["pci"]) for d in devs}) if dev is not None else None}


def init_group(backend, timeout_s, **kw):
    """init_process_group with a deadline: a rank whose peers never arrive exits non-zero instead of hanging the node
    (the launcher then tears the job down; nothing here re-executes a process that has touched the GPU)."""
    import datetime

    import torch

    try:
        torch.distributed.init_process_group(backend, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    except Exception as e:  # rendezvous timeout, refused connection, RCCL bootstrap failure
        print(f"bench.py: rank {os.environ.get('RANK', '?')}: init_process_group({backend}) failed within {timeout_s} s: {e}",
              file=sys.stderr, flush=True)
        os._exit(3)


def dry_run(args, world, rank):
    """The multi-rank protocol of the real run -- process group, barrier, timed region, max-over-ranks, rank-0 JSON,
    final barrier -- with the kernels replaced by nothing, over gloo on the CPU."""
    import torch

    ident = None
    if world > 1:
        init_group("gloo", args.timeout_s)
        assert torch.distributed.get_world_size() == args.gpus
        ident = rank_identity(world, rank, None, False)
        torch.distributed.barrier()
    t0 = time.perf_counter()
    shard = torch.full((4, 3), float(rank))
    t_ag = 0.0
    if world > 1:
        gathered = torch.empty(world * 4, 3)
        ta = time.perf_counter()
        torch.distributed.all_gather_into_tensor(gathered, shard)
        t_ag = time.perf_counter() - ta
        assert [float(gathered[4 * r, 0]) for r in range(world)] == [float(r) for r in range(world)]
        torch.distributed.barrier()
    wall = elapsed = time.perf_counter() - t0
    per_rank = exchange = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

        def gather64(v):
            out = torch.empty(world * v.numel(), dtype=torch.float64)
            torch.distributed.all_gather_into_tensor(out, v)
            return out

        per_rank = rank_breakdown(world, gather64, wall, 0.0, t_ag)
        if args.resample_every > 0:
            from pita_amd.sde_integration import _Comm

            Bl = 64

            def ids_cpu(ag, u0):  # utils.py:111-120 on the host (the dry run has no kernels)
                import numpy as np

                w = torch.clip(torch.softmax(ag, -1), 1e-6, 1.0)
                u = (u0 + torch.arange(ag.shape[0], dtype=torch.float64) / ag.shape[0]) % 1.0
                ids = np.digitize(u.numpy(), torch.cumsum(w, -1).numpy(), right=True)
                return torch.from_numpy(np.minimum(ids, ag.shape[0] - 1))

            xs = torch.arange(Bl * 3, dtype=torch.float32).reshape(Bl, 3) + 1000.0 * rank
            exchange, xs = resample_exchange_leg(_Comm(None), xs, args.resample_every, Bl, world, rank, ids_cpu,
                                                 lambda: None)
            assert xs.shape == (Bl, 3)
    if rank == 0:
        print(json.dumps({"metric": "dry run (no kernels)", "value": None, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "scaling": "strong" if args.strong else "weak",
                          "per_rank": per_rank, "resample_exchange": exchange,
                          "rccl_ranks_seen": ident["rccl_ranks_seen"] if ident else None,
                          "devices": ident["devices"] if ident else None,
                          "config": {"backend": "gloo" if world > 1 else None}}), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="lj13")
    ap.add_argument("--walkers", type=int, default=0, help="walkers per GPU (default: the config's; with --strong: in total)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --walkers is the TOTAL batch, split across ranks")
    ap.add_argument("--chunk", type=int, default=0, help="SDE steps per kernel launch (default gcd(steps, warmup))")
    ap.add_argument("--force-evals", type=int, default=200, help="target force-kernel launches for its roofline")
    ap.add_argument("--force-last", action="store_true", help="run the force-kernel leg after the timed region (A/B aid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-debiased", action="store_true", help="skip the secondary debiased-regime measurement")
    ap.add_argument("--cpu-walkers", type=int, default=512, help="CPU sample: the reference's own inference chunk for LJ13")
    ap.add_argument("--cpu-steps", type=int, default=0, help="CPU sample steps (default sized for ~10-20 s)")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "f16x2"], default=None,
                    help="dense-layer arithmetic of the EGNN kernel (default: the library's)")
    ap.add_argument("--resample-every", type=int, default=0,
                    help="with several ranks: after the timed region, time global resampling events (log-weight all-gather + "
                         "systematic resampling + _Comm.exchange_rows) at this shard, as a trajectory that resamples every K "
                         "steps would pay them; reported beside the metric, never inside it")
    ap.add_argument("--timeout-s", type=float, default=300.0,
                    help="deadline of init_process_group: a rank whose peers never arrive exits with code 3")
    ap.add_argument("--no-e2e", action="store_true", help="skip the whole-integrate_sde legs behind the timed region")
    ap.add_argument("--no-small-batch", action="store_true",
                    help="skip the small-batch legs (the reference's own 512 ... 16 384 walkers) behind the timed region")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / timing protocol only, no kernels and no GPU (gloo): CPU test of the "
                         "multi-rank path; the JSON carries value null")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import numpy as np
    import torch

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry_run:
        return dry_run(args, world, rank)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # PITA_BENCH_ONE_DEVICE=1: rehearsal of the multi-rank control flow on a ONE-GPU box (all ranks share cuda:0, gloo
    # with host staging instead of RCCL, which refuses two ranks on one device); its numbers mean nothing
    rehearsal = world > 1 and os.environ.get("PITA_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ident = None
    if world > 1:
        if rehearsal:
            init_group("gloo", args.timeout_s)
        else:
            init_group("nccl", args.timeout_s, device_id=dev)
        assert torch.distributed.get_world_size() == args.gpus
        ident = rank_identity(world, rank, dev, rehearsal)

    def all_gather(dst, src):
        if rehearsal:
            host = torch.empty(dst.shape, dtype=dst.dtype)
            torch.distributed.all_gather_into_tensor(host, src.cpu())
            dst.copy_(host)
        else:
            torch.distributed.all_gather_into_tensor(dst, src)

    def all_reduce_max(t):
        if rehearsal:
            h = t.cpu()
            torch.distributed.all_reduce(h, op=torch.distributed.ReduceOp.MAX)
            return h
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return t

    import pita_amd

    pita_amd._lib.lib()  # fail loudly if the HIP library is missing
    cfg = dict(CONFIGS[args.config], name=args.config)
    n, d = cfg["n"], cfg["d"]
    D = n * d
    B, K, W = (args.walkers or cfg["walkers"]), args.steps, args.warmup
    if args.strong:
        assert B % world == 0, "--strong: --walkers must be divisible by the number of ranks"
        B //= world
    chunk = args.chunk or (math.gcd(K, W) if W > 0 else K)
    assert K % chunk == 0 and W % chunk == 0

    net = build_model(pita_amd, n, d, precision=args.precision)
    sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=cfg["sigma_min"], sigma_max=80.0, rho=7)
    gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
    NGRID = 1000  # the 1000-step time grid of the reference config; K+W steps walk along it (wrapping)
    times = torch.linspace(1.0, 0.0, NGRID + 1)[:-1]
    tab_h = pita_amd.sde_integration.build_step_table(sched, gam, times, 1.0 / NGRID, 1.0, 1.0)
    idx = torch.arange(W + K) % NGRID
    tab = tab_h[idx].contiguous().to(dev)
    scale = float((sched.h(torch.tensor(1.0)) / gam.gamma(torch.tensor(1.0))) ** 0.5)
    prior = pita_amd.Prior(scale=scale, n_particles=n, spatial_dim=d, device=dev, seed=12345)
    x = prior.sample(B, walker_offset=rank * B)
    seed = 12345

    def run(s0, s1):
        for s in range(s0, s1, chunk):
            net.sampler_run(x, tab[s:s + chunk], chunk, seed=seed, walker_offset=rank * B, step0=s, remove_mean=True)

    # Secondary legs FIRST (rank 0 of a one-GPU run): the target's force-kernel roofline and the debiased regime run
    # before the headline region.  Measured reason (ms_of_each_launch of earlier lines): from a cold start this kernel's
    # launches take 6.4, 5.9, 5.7, 5.6 ... 5.49 ms -- the chip needs ~40 ms of sustained matrix work to reach the clock
    # it then holds, and the driver's --steps 20 --warmup 5 region (27 ms in all) would sit entirely inside that ramp.
    # With the debiased leg (0.2 s of back-to-back MFMA launches) immediately before, the W warm-up steps and the K
    # timed steps run at the sustained clock, which is what a 1 000-step trajectory sees.  The timed region itself is
    # unchanged: W untimed steps, barrier + synchronize, exactly K steps, synchronize + barrier.
    # EVERY rank runs them (rank 0 reports): with several ranks the time is the slowest rank's, and a rank that skipped
    # the legs would enter the timed region cold.
    force_rl = debiased = None
    if args.force_evals > 0 and not args.force_last:
        energy = make_target(pita_amd, cfg, dev)
        xf = x.clone() if cfg["target"] != "ff" else pita_amd.Prior(scale=1.0, n_particles=n, spatial_dim=d, device=dev,
                                                                    seed=3).sample(B)
        force_rl = force_roofline(pita_amd, cfg, energy, xf, dev, args.force_evals)
        del xf
    if not args.no_debiased and not args.force_last:
        # LJ55: at the configuration's own shard (32 768 walkers per GPU: the primal cache is processed in chunks of
        # PITA_DIV_CACHE_GB) unless --walkers asked for something else; rounds 1-4 capped this leg at 4 096
        Bd = B
        debiased = debiased_leg(pita_amd, net, cfg, dev, Bd, with_cpu=world == 1 and not args.no_cpu_baseline and n <= 13)
    ad2cat = None
    if rank == 0 and world == 1 and args.config == "aldp22" and not args.force_last:
        # the reference's alanine-dipeptide EGNN (egnn_dynamics_ad2_cat.yaml: hidden 64 x 5 layers, one-hot atom types) on
        # the matrix-pipe wide kernel, through the integrator's per-step path (fused EDM evaluation + pita_em_step)
        from pita_amd.egnn_dynamics_ad2_cat import EGNN_dynamics_AD2_cat

        torch.manual_seed(12345)
        net64 = EGNN_dynamics_AD2_cat(n, d, hidden_nf=64, n_layers=5, condition_beta=True)
        sn64 = pita_amd.ScoreNet(net64)
        xa = x.clone()
        L_ = pita_amd._lib.lib()

        def step64(k):
            row = tab_h[k]
            ht = torch.full((B,), float(row[pita_amd._lib.ST_H]), device=dev)
            score = sn64(ht, xa, 1.0)
            drift = (float(row[pita_amd._lib.ST_GAMMA]) * (score * float(row[pita_amd._lib.ST_G2]))).contiguous()
            L_.pita_em_step(xa.data_ptr(), drift.data_ptr(), 0, B, n, d, float(row[pita_amd._lib.ST_DT]),
                            float(row[pita_amd._lib.ST_NOISE_SCALE]), float(row[pita_amd._lib.ST_SQRT_DT]), seed, rank * B, k, 1,
                            0, pita_amd._lib.stream_ptr(dev))

        for k in range(2):
            step64(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(2, 10):
            step64(k)
        torch.cuda.synchronize()
        dt64_steps = (time.perf_counter() - t0) / 8
        # the same steps as ONE launch (pita_egnn_wide_sampler_run: walkers on chip between the steps)
        xa = x.clone()
        tab64 = tab_h[2:52].contiguous().to(dev)
        net64.sampler_run(xa, tab64[:10].contiguous(), 10, seed=seed, walker_offset=rank * B, step0=2, remove_mean=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net64.sampler_run(xa, tab64, 50, seed=seed, walker_offset=rank * B, step0=2, remove_mean=True)
        torch.cuda.synchronize()
        dt64 = (time.perf_counter() - t0) / 50
        # the debiased (Feynman-Kac) regime with this backbone as score AND energy net: dim + 1 forward-mode launches for
        # the score net's trace (pita_egnn_wide_jvp: matrix-pipe kernel for 22 atoms), ONE reverse-mode launch on the
        # energy net (pita_egnn_wide_vjp), assembly, clamp
        import copy as _copy

        from pita_amd.energy_net import EnergyNet as _EnergyNet

        sde64 = pita_amd.VEReverseSDE(noise_schedule=sched, score_net=sn64, energy_net=_EnergyNet(_copy.deepcopy(net64)),
                                      debias_inference=True)
        xd = pita_amd.Prior(scale=3.0, n_particles=n, spatial_dim=d, device=dev, seed=7).sample(B)
        td = torch.tensor(0.5)  # host scalar, as the integrator passes the step time (same path as the LJ legs)
        sde64.f(td, xd, 1.0, gam, None, None, clamp_chunk=512)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sde64.f(td, xd, 1.0, gam, None, None, clamp_chunk=512)
        torch.cuda.synchronize()
        dt64_deb = time.perf_counter() - t0
        mac = 5 * (n * (n - 1) * ((2 * 64 + 2) * 64 + 2 * 64 * 64 + 2 * 64) + n * 3 * 64 * 64)
        on_mfma = net64.uses_matrix_pipe(dev)
        ad2cat = {"backbone": "EGNN_dynamics_AD2_cat hidden 64 x 5 layers (pita_egnn_wide_sampler_run: 50 steps per launch, "
                              + ("f16 two-piece MFMA, matrix pipe)" if on_mfma else "fp32 vector pipe)"),
                  "walkers": B, "ms_per_step": dt64 * 1e3, "value": B / dt64, "unit": "walker-steps/s",
                  "ms_per_step_launch_per_step_path": dt64_steps * 1e3,
                  "debiased": {"ms_per_step": dt64_deb * 1e3, "value": B / dt64_deb, "unit": "walker-steps/s",
                               "launches_per_step": f"{n * d + 1} forward-mode launches (pita_egnn_wide_jvp) + 1 reverse-mode launch "
                                                    "(pita_egnn_wide_vjp) + assembly + clamp"},
                  "algorithmic_TFLOPs": 2 * mac * B / dt64 / 1e12,
                  ("frac_of_dense_f16_mfma_peak_2500" if on_mfma else "frac_of_plain_fma_rate_78.6"):
                      2 * mac * B / dt64 / (2500e12 if on_mfma else 78.65e12),
                  "finite": bool(torch.isfinite(xa).all())}
        del xa
    run(0, W)  # warm-up (also builds the native handle)
    gathered = torch.empty(world * B, D, device=dev) if world > 1 else None
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    n_launch = K // chunk
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_launch + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i, s in enumerate(range(W, W + K, chunk)):
        net.sampler_run(x, tab[s:s + chunk], chunk, seed=seed, walker_offset=rank * B, step0=s, remove_mean=True)
        evs[i + 1].record()
    ev_ag = torch.cuda.Event(enable_timing=True)
    if world > 1:  # X1: the only collective of the resampling-free path
        all_gather(gathered, x)
        ev_ag.record()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    wall = elapsed = time.perf_counter() - t0
    per_rank = exchange = None
    launch_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n_launch)]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        elapsed = float(all_reduce_max(t).item())

        def gather64(v):
            out = torch.empty(world * v.numel(), device=dev, dtype=torch.float64)
            all_gather(out, v.to(dev))
            return out.cpu()

        # (the all-gather's event interval starts when the last sampler launch ends on this rank: it includes the wait
        # for the slowest rank to arrive, which is what a rank loses to the collective)
        per_rank = rank_breakdown(world, gather64, wall, sum(launch_ms) * 1e-3, evs[n_launch].elapsed_time(ev_ag) * 1e-3)
    assert torch.isfinite(x).all(), "sampler produced non-finite walkers"
    e2e_ranks = None
    if world > 1 and not args.no_e2e:
        e2e_ranks = e2e_multi_rank(pita_amd, net, cfg, dev, B, world, NGRID, gather64)
    if world > 1 and args.resample_every > 0:
        from pita_amd.sde_integration import _Comm
        from pita_amd.utils import sample_cat_sys

        exchange, _ = resample_exchange_leg(_Comm(None), x.clone(), args.resample_every, B, world, rank,
                                            lambda ag, u0: sample_cat_sys(ag.shape[0], ag, u0)[0], torch.cuda.synchronize)

    if rank == 0 and args.force_evals > 0 and args.force_last:
        energy = make_target(pita_amd, cfg, dev)
        # the sampler's walkers for the pair targets; compact synthetic conformations for the force field
        xf = x if cfg["target"] != "ff" else pita_amd.Prior(scale=1.0, n_particles=n, spatial_dim=d, device=dev,
                                                            seed=3).sample(B)
        force_rl = force_roofline(pita_amd, cfg, energy, xf, dev, args.force_evals)

    if rank == 0:
        avg_ms = float(np.mean(launch_ms))
        m16, m32 = net.sampler_work(B, dev)
        exec16, exec32 = m16 * MFMA16_FLOP, m32 * MFMA32_FLOP
        walker_steps_per_s = B * chunk / (avg_ms * 1e-3)
        achieved = walker_steps_per_s * exec16 / 1e12
        alg = egnn_algorithmic_flop(n)
        # matrix-pipe occupancy at the nominal clock: 32 cycles per 32x32x16 MFMA, 64 per 32x32x2 f32 MFMA, 1024 SIMDs
        pipe_busy = walker_steps_per_s * (m16 * 32 + m32 * 64) / (1024 * NOMINAL_CLOCK_HZ)
        pm = pmc_summary()
        pk = pm.get(f"sampler_{args.config}") if pm else None
        same = pk is not None and pk.get("walkers") == B
        alg_bytes = 2 * B * D * 4  # the walkers in and out once per launch (312 B per LJ13 walker)
        alg_tflops = walker_steps_per_s * alg / 1e12
        # HBM bytes per launch from the committed PMC passes: a fixed part per launch (the walkers in and out, the f16
        # path's backup copy is a separate memcpy and not in the kernel's counters, per-walker-group register spills)
        # + a part per walker-step, fitted from passes at two launch sizes; FETCH_SIZE corrected as the guide prescribes
        traffic = None
        if same and "fixed_bytes_per_launch" in pk:
            traffic = pk["fixed_bytes_per_launch"] + pk["bytes_per_walker_step"] * B * chunk
            assert traffic >= alg_bytes, "PMC traffic below the algorithmic bytes: the counter correction is wrong"
        roof = {"kernel": f"egnn_kernel<{n},{d},...,SAMPLER> (fused EGNN score + EDM + EM step; one launch = {chunk} steps)",
                "bound": "mfma",
                # ALGORITHMIC flops per launch (4.197 MFLOP per LJ13 walker-step, counted on the reference's
                # fp32 formulation) / average launch duration, against the dense 16-bit MFMA peak (the pipe used)
                "achieved": alg_tflops, "peak": PEAK_MFMA16_TFLOPS, "unit": "TFLOP/s",
                "frac": alg_tflops / PEAK_MFMA16_TFLOPS, "frac_algorithmic": alg_tflops / PEAK_MFMA16_TFLOPS,
                "algorithmic_flop_per_walker_step": alg, "algorithmic_flop_per_launch": alg * B * chunk,
                "algorithmic_bytes_per_launch": alg_bytes,
                "traffic": traffic,
                "traffic_source": "committed PMC (profiles/pmc_sampler_current.json), scaled to this launch size; not "
                                  "collected in this run" if traffic is not None else None,
                "traffic_model": {k: pk[k] for k in ("fixed_bytes_per_launch", "bytes_per_walker_step", "fit_from",
                                                     "fetch_size_correction") if k in pk} if same else None,
                "ms_per_launch": avg_ms, "launches": n_launch, "steps_per_launch": chunk,
                "ms_of_each_launch": [round(v, 4) for v in launch_ms[:64]],
                # what the matrix pipe actually executes (fp32-accurate split products included): pipe utilisation
                "achieved_executed": achieved, "frac_executed": achieved / PEAK_MFMA16_TFLOPS,
                "executed_mfma_flop_per_walker_step": exec16,
                "executed_f32_mfma_flop_per_walker_step": exec32,
                "mfma16_per_walker_step": m16, "mfma32_per_walker_step": m32,
                "mfma_pipe_busy_frac_at_2.4GHz": pipe_busy,
                "valu_issue_frac": pk.get("valu_issue_frac") if same else None,
                "pmc_source": ("committed PMC (profiles/pmc_sampler_current.json; not collected in this run): "
                               + str(pk.get("source"))) if same else None,
                "dense_layer_arithmetic": {0: "f32 MFMA", 1: "bf16 MFMA, 3-piece split", 2: "f16 MFMA, 2-piece split"}[net.precision],
                "note": "achieved/frac follow SURVEY 8(d): algorithmic flops of the reference's formulation per launch / "
                        "launch time (HIP events on the launch stream) against the dense 16-bit MFMA peak.  *_executed "
                        "count the 16-bit matrix-pipe flops the kernel really issues (split products).  The kernel is "
                        "VALU-issue-bound (activations + operand splits): valu_issue_frac, DESIGN.md section 4.1.  "
                        "traffic = PMC bytes per launch (fixed part + per walker-step part, FETCH_SIZE x2 as the guide "
                        "prescribes for gfx950): the walkers in and out once per launch + register-spill scratch; no "
                        "walker data is re-read"}
        assert 0.0 < roof["frac"] <= roof["frac_executed"] <= 1.0 and pipe_busy <= 1.0, "roofline fraction must be a fraction"
        out = {
            "metric": "walker-steps/sec (batch x T) LJ13 @ 65k walkers/GPU" if args.config == "lj13"
                      else f"walker-steps/sec (batch x T) {args.config} @ {B} walkers/GPU",
            "value": world * B * K / elapsed,
            "unit": "walker-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed * 1e3 / K,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{cfg['label']}: {n}x{d}D annealed reverse VE-SDE, EGNN h32x3 score net (seed-12345 "
                                   f"init), not-debiased, resampling off, Elucidating({cfg['sigma_min']},80,7), gamma=4/3, "
                                   "beta=1",
                       "walkers_per_gpu": B, "global_walkers": world * B, "steps_per_launch": chunk,
                       "final_allgather_bytes_per_rank": B * D * 4 if world > 1 else 0,
                       "parallelism": f"walker-sharded x{world}, final all_gather only",
                       "backend": ("gloo (one-device rehearsal)" if rehearsal else "nccl (RCCL)") if world > 1 else None},
            "roofline": roof,
            "roofline_force": force_rl,
        }
        # sources the build compiled WITHOUT their optional per-file flags (slower kernels there); [] = none, None = no record
        from pita_amd import build as _build
        out["build_fallback_objects"] = _build.fallback_objects()
        if world > 1:
            out["per_rank"] = per_rank
            out["rccl_ranks_seen"] = ident["rccl_ranks_seen"]
            out["devices"] = ident["devices"]
            out["distinct_devices"] = ident["distinct_devices"]
            if exchange is not None:
                out["resample_exchange"] = exchange
            if e2e_ranks is not None:
                out["e2e"] = {"not_debiased": e2e_ranks}
        if ad2cat is not None:
            out["ad2cat_backbone"] = ad2cat
        # which part of the reference's 1 000-step grid the timed region covered (t_k = 1 - k / 1000)
        g0, g1 = W % NGRID, (W + K - 1) % NGRID
        out["steps_of_grid"] = {"grid_steps": NGRID, "first": g0, "last": g1, "t_first": 1.0 - g0 / NGRID,
                                "t_last": 1.0 - g1 / NGRID, "wraps": (W + K) > NGRID,
                                "note": "the timed launches are bare pita_egnn_sampler_run calls along the grid; `e2e` below is "
                                        "the whole integrate_sde"}
        if world == 1 and not args.no_e2e:
            out["e2e"] = e2e_legs(pita_amd, net, cfg, dev, B, NGRID, 100 if n <= 13 else 20)
        if world == 1 and not args.no_small_batch and args.config == "lj13":
            sb = small_batch_legs(pita_amd, net, cfg, dev)
            full = out.get("e2e") or {}
            for legs in sb["sizes"].values():
                for name in ("not_debiased", "default_regime"):
                    if name in legs and name in full:
                        legs[name]["frac_of_full_batch"] = legs[name]["value"] / full[name]["value"]
            out["small_batch"] = sb
        if world == 1 and not args.no_cpu_baseline:
            # ~10-20 s of CPU work: the cost per walker-step grows with the number of edges
            steps = args.cpu_steps or max(4, int(300 * 156 / (n * (n - 1))))
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_walkers, steps)
        else:
            out["cpu_baseline"] = None
        if not args.no_debiased:
            if debiased is None:
                Bd = B
                debiased = debiased_leg(pita_amd, net, cfg, dev, Bd,
                                        with_cpu=world == 1 and not args.no_cpu_baseline and n <= 13)
            debiased.pop("_keepalive", None)
            xc = debiased.pop("_x48", None)
            if xc is not None:
                debiased["cpu_baseline"] = debiased_cpu_baseline(net, cfg, xc)
            out["debiased"] = debiased
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()  # rank 0 may still be in its force-kernel leg
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
