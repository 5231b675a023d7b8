#!/usr/bin/env python3
"""bench.py -- walker-steps/s of the fused HIP annealed-SDE sampler (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): LJ13
(13 particles x 3D), EGNN score net (hidden 32 x 3 layers, tanh, attention, temperature
conditioned; seed-12345 initialisation, i.e. the reference's own untrained init), Elucidating
schedule (sigma_min 0.05, sigma_max 80, rho 7), gamma = 4/3, beta = 1, diffusion_scale 1,
NOT-debiased reverse VE-SDE, resampling off, 65 536 walkers PER GPU (weak scaling), synthetic
walkers from the mean-free prior.  One "step" = one Euler-Maruyama step of all walkers:
EDM-preconditioned EGNN forward + drift + noise + update + remove_mean, all inside the fused
kernel `egnn_kernel<13,3,7,4>`; the K timed steps run as K/c launches of c = gcd(K, W) steps each
(the warm-up uses the same launch size so every launch of the kernel in this process is equal
and the rocprof average is comparable).  After the timed region a separately timed loop of LJ13
log-density+force evaluations gives the HBM roofline of the pairwise-force kernel.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_WALKER_STEP = 4.197e6   # EGNN forward h32x3, LJ13 (2 x 2 098 304 MAC)
LJ13_BYTES_PER_EVAL = 316        # read x (39 f32) + write force (39) + logp (1)
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak


def pmc_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (collected with this same
    command under rocprofv3 and corrected as MI355X_MICROARCH.md prescribes; see profiles/r01_pmc_traffic.json).  PMC
    collection cannot run inside the timed bench itself, so the live JSON carries the last committed measurement."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return json.load(f)[key]["traffic_bytes"]
    except Exception:
        return None


def build_model(pa, seed=12345):
    torch.manual_seed(seed)
    net = pa.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                           condition_time=True, condition_temperature=True, agg="sum")
    return net


def cpu_baseline(n_walkers, n_steps, seed=12345):
    """The oracle (torch-CPU restatement of the reference's sampler, kind = "port") timed on this
    host's cores on a bounded sample of the same workload."""
    from oracle import pita_oracle as O

    torch.manual_seed(seed)
    import pita_amd

    net = build_model(pita_amd, seed)
    w = {k: v.detach().clone() for k, v in net.state_dict().items()}
    bb = lambda cn, xs, b: O.egnn_forward(w, cn, xs, b, 13, 3)
    sched, gam = O.Elucidating(0.05, 80.0, 7), O.GammaConstant(4 / 3)
    gen = torch.Generator().manual_seed(1)
    x1 = O.prior_from_noise(torch.randn(n_walkers, 39, generator=gen), O.prior_scale(sched, gam, 1.0), 13, 3)
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
                            noise_fn, 13, 3)  # warm-up
            t0 = time.perf_counter()
            O.integrate_sde(probe_cfg, x1, drift, sched.g, noise_fn, 13, 3)
            probe[nthr] = time.perf_counter() - t0
        nthr = min(probe, key=probe.get)
        torch.set_num_threads(nthr)
        t0 = time.perf_counter()
        O.integrate_sde(cfg_run, x1, drift, sched.g, noise_fn, 13, 3)
        best = (time.perf_counter() - t0, nthr)
    torch.set_num_threads(all_threads)
    dt, nthr = best
    return {"value": n_walkers * n_steps / dt, "unit": "walker-steps/s", "cores": nthr,
            "kind": "port", "sample": f"oracle integrate_sde, LJ13 EGNN h32x3, {n_walkers} walkers x {n_steps} steps, "
            f"torch-CPU fp32, {dt:.1f} s with {nthr} threads (host has {os.cpu_count()} logical CPUs)"}


def debiased_leg(pita_amd, net, dev, B, with_cpu):
    """Secondary number: the debiased Feynman-Kac regime (PITA's default; sdes.py:151-239): drift of x and of the
    log-weights through 13 three-direction divergence launches (pita_egnn_div_accumulate) + 1 forward-mode
    (pita_egnn_jvp, h direction) + 1 reverse-mode launch (pita_egnn_vjp) + assembly + quantile clamp, then the EM update."""
    import copy

    from pita_amd.energy_net import EnergyNet

    sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
    gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
    sde = pita_amd.VEReverseSDE(noise_schedule=sched, score_net=pita_amd.ScoreNet(net),
                                energy_net=EnergyNet(copy.deepcopy(net)), debias_inference=True)
    x = pita_amd.Prior(scale=3.0, n_particles=13, spatial_dim=3, device=dev, seed=7).sample(B)
    t = torch.tensor(0.5, device=dev)
    L = pita_amd._lib.lib()

    def step():
        terms = sde.f(t, x, 1.0, gam, None, None, clamp_chunk=512)
        L.pita_em_step(x.data_ptr(), terms.drift_X.data_ptr(), 0, B, 13, 3, 1e-3, 0.1, float(np.sqrt(1e-3)), 1, 0, 0, 1,
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
    out = {"metric": "walker-steps/s, debiased (Feynman-Kac) regime, LJ13", "value": B / dt, "walkers": B,
           "ms_per_step": dt * 1e3, "launches_per_step": 15 + 3}
    if with_cpu:
        from oracle import pita_oracle as O

        w = {k: v.detach().clone() for k, v in net.state_dict().items()}
        bb = lambda cn, xs, b: O.egnn_forward(w, cn, xs, b, 13, 3)
        osched, ogam = O.Elucidating(0.05, 80.0, 7), O.GammaConstant(4 / 3)
        nb = 48
        xc = x[:nb].cpu()
        torch.set_num_threads(min(16, torch.get_num_threads()))
        t0 = time.perf_counter()
        O.f_debiased(bb, bb, osched, ogam, torch.tensor(0.5), xc, 1.0)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": nb / dtc, "unit": "walker-steps/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"oracle f_debiased (autograd + vmap(jacrev)), {nb} walkers x 1 step, {dtc:.1f} s"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--walkers", type=int, default=65536, help="walkers per GPU (with --strong: in total)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --walkers is the TOTAL batch, split across ranks")
    ap.add_argument("--chunk", type=int, default=0, help="SDE steps per kernel launch (default gcd(steps, warmup))")
    ap.add_argument("--force-evals", type=int, default=200, help="LJ13 force-kernel launches for its roofline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-debiased", action="store_true", help="skip the secondary debiased-regime measurement")
    ap.add_argument("--cpu-walkers", type=int, default=512, help="CPU sample: the reference's own inference chunk for LJ13")
    ap.add_argument("--cpu-steps", type=int, default=300)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # PITA_BENCH_ONE_DEVICE=1: rehearsal of the multi-rank control flow on a ONE-GPU box (all ranks share cuda:0, gloo
    # with host staging instead of RCCL, which refuses two ranks on one device); its numbers mean nothing
    rehearsal = world > 1 and os.environ.get("PITA_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)

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
    B, K, W = args.walkers, args.steps, args.warmup
    if args.strong:
        assert B % world == 0, "--strong: --walkers must be divisible by the number of ranks"
        B //= world
    chunk = args.chunk or (math.gcd(K, W) if W > 0 else K)
    assert K % chunk == 0 and W % chunk == 0

    net = build_model(pita_amd)
    sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
    gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
    NGRID = 1000  # the 1000-step time grid of the reference config; K+W steps walk along it (wrapping)
    times = torch.linspace(1.0, 0.0, NGRID + 1)[:-1]
    tab_h = pita_amd.sde_integration.build_step_table(sched, gam, times, 1.0 / NGRID, 1.0, 1.0)
    idx = torch.arange(W + K) % NGRID
    tab = tab_h[idx].contiguous().to(dev)
    scale = float((sched.h(torch.tensor(1.0)) / gam.gamma(torch.tensor(1.0))) ** 0.5)
    prior = pita_amd.Prior(scale=scale, n_particles=13, spatial_dim=3, device=dev, seed=12345)
    x = prior.sample(B, walker_offset=rank * B)
    energy = pita_amd.LennardJonesEnergy(39, 13, 3, device=dev)
    seed = 12345

    def run(s0, s1):
        for s in range(s0, s1, chunk):
            net.sampler_run(x, tab[s:s + chunk], chunk, seed=seed, walker_offset=rank * B, step0=s, remove_mean=True)

    run(0, W)  # warm-up (also builds the native handle)
    gathered = torch.empty(world * B, 39, device=dev) if world > 1 else None
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
    if world > 1:  # X1: the only collective of the resampling-free path
        all_gather(gathered, x)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        elapsed = float(all_reduce_max(t).item())
    launch_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n_launch)]
    assert torch.isfinite(x).all(), "sampler produced non-finite walkers"

    # ---- pairwise-force kernel roofline (HBM-bound), separately timed: at the workload's 65 536 walkers, at a
    #      streaming-size batch (2^21 walkers, 663 MB per launch) and next to a plain device copy of the same bytes
    force_rl = None
    if rank == 0 and args.force_evals > 0:
        L = pita_amd._lib.lib()
        sp = pita_amd._lib.stream_ptr(dev)

        def time_force(xb, reps):
            nb = xb.shape[0]
            lp, fo = torch.empty(nb, device=dev), torch.empty_like(xb)
            for _ in range(3):
                L.pita_lj_logp_force(xb.data_ptr(), lp.data_ptr(), fo.data_ptr(), nb, 13, 3, 1.0, 1.0, 1e-6, 1.0, 1.0, 1.0, sp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                L.pita_lj_logp_force(xb.data_ptr(), lp.data_ptr(), fo.data_ptr(), nb, 13, 3, 1.0, 1.0, 1e-6, 1.0, 1.0, 1.0, sp)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps

        def time_copy(nfloat, reps):
            a, b = torch.empty(nfloat, device=dev), torch.empty(nfloat, device=dev)
            for _ in range(3):
                b.copy_(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps

        us = time_force(x, args.force_evals)
        gbs = B * LJ13_BYTES_PER_EVAL / (us * 1e-6) / 1e9
        BIG = 1 << 21
        xbig = x.repeat(BIG // B + 1, 1)[:BIG].contiguous()
        us_big = time_force(xbig, 20)
        gbs_big = BIG * LJ13_BYTES_PER_EVAL / (us_big * 1e-6) / 1e9
        us_copy = time_copy(B * 39, args.force_evals)  # a device copy moving the same 2 x 10.2 MB
        # the force kernel where production uses it per step: negative-time descent (sde_integration.py:353-360),
        # all steps of a launch with the walkers resident in LDS (pita_lj_descent); algorithmic bytes stay
        # SURVEY 8(d)'s 316 B per walker-eval, actual HBM traffic is one read + one write of x per LAUNCH
        S_DESC = 1000
        xd = x.clone()

        def run_descent(noise_scale):
            pita_amd._lib.check(L.pita_lj_descent(xd.data_ptr(), 0, B, 13, 3, 1.0, 1.0, 1e-6, 1.0, 1.0, 1.0, S_DESC, 1e-7,
                                                  noise_scale, math.sqrt(2e-7), 1, 0, 0, 1, sp), "pita_lj_descent")

        def time_descent(noise_scale):
            run_descent(noise_scale)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run_descent(noise_scale)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / S_DESC

        us_desc, us_ula = time_descent(0.0), time_descent(1.0)
        assert torch.isfinite(xd).all(), "descent produced non-finite walkers"
        gbs_desc = B * LJ13_BYTES_PER_EVAL / (us_desc * 1e-6) / 1e9
        force_rl = {"kernel": "lj13_kernel<2> (LJ13 logp+force, 65 536 walkers)", "bound": "hbm", "achieved": gbs,
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                    "traffic": pmc_traffic("lj13_kernel<2> @65536 walkers") if B == 65536 else None,
                    "us_per_launch": us, "walker_evals_per_s": B / (us * 1e-6), "launches": args.force_evals,
                    "same_bytes_device_copy_us": us_copy,
                    "same_bytes_device_copy_GBs": 2 * B * 39 * 4 / (us_copy * 1e-6) / 1e9,
                    "large_batch": {"kernel": "lj13_kernel<1>", "walkers": BIG, "us_per_launch": us_big,
                                    "achieved": gbs_big, "frac": gbs_big / PEAK_HBM_GBS},
                    "in_descent_loop": {"kernel": "lj13_descent_kernel (force + update + centring, walkers LDS-resident)",
                                        "walkers": B, "steps_per_launch": S_DESC, "us_per_step": us_desc,
                                        "walker_evals_per_s": B / (us_desc * 1e-6), "achieved": gbs_desc,
                                        "frac": gbs_desc / PEAK_HBM_GBS, "unit": "GB/s (algorithmic, 316 B/walker-eval)",
                                        "hbm_bytes_per_launch": 2 * B * 39 * 4,
                                        "us_per_step_with_langevin_noise": us_ula}}
        del xbig

    if rank == 0:
        avg_ms = float(np.mean(launch_ms))
        achieved = B * chunk * FLOP_PER_WALKER_STEP / (avg_ms * 1e-3) / 1e12
        out = {
            "metric": "walker-steps/sec (batch x T) LJ13 @ 65k walkers/GPU",
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
            "config": {"workload": "LJ13 (13x3D) annealed reverse VE-SDE, EGNN h32x3 score net (seed-12345 init), "
                                   "not-debiased, resampling off, Elucidating(0.05,80,7), gamma=4/3, beta=1",
                       "walkers_per_gpu": B, "global_walkers": world * B, "steps_per_launch": chunk,
                       "parallelism": f"walker-sharded x{world}, final all_gather only"},
            "roofline": {"kernel": "egnn_kernel<13,3,7,4,1,true> (fused EGNN score + EM step)", "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": pmc_traffic("egnn_kernel<13,3,7,4,1> @65536 walkers x 100 steps")
                         if (B == 65536 and chunk == 100) else None,
                         "note": "fp32-accurate dense layers run as exact 3-way bf16 splits on the bf16 matrix pipe; "
                                 "algorithmic flops count the reference's un-split first edge layer, so frac can exceed 1; "
                                 "the kernel is VALU-issue-bound (activations + operand splits), see DESIGN.md",
                         "traffic_note": "x is read once and written once per launch (20.4 MB at 65 536 walkers); everything else "
                                         "in `traffic` is register-spill scratch (468 B/lane, outside the edge loop) going "
                                         "to L2 / Infinity Cache at ~0.35 TB/s of the 8 TB/s available",
                         "algorithmic_flop_per_walker_step": FLOP_PER_WALKER_STEP, "ms_per_launch": avg_ms,
                         "launches": n_launch},
            "roofline_force": force_rl,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_walkers, args.cpu_steps)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_debiased:
            out["debiased"] = debiased_leg(pita_amd, net, dev, B, with_cpu=not args.no_cpu_baseline)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()  # rank 0 may still be in its force-kernel leg
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
