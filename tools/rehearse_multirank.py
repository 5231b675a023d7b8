"""Two ranks on ONE GPU (gloo): the sharded integrate_sde must reproduce the single-rank result.
run: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/rehearse_multirank.py
(RCCL refuses two ranks on one device, so this rehearses rank slicing, Philox keying by global walker id, global
resampling and the MALA acceptance all-reduce over gloo.)"""
import copy, os, sys
from types import SimpleNamespace
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pita_amd as pa
from pita_amd.energy_net import EnergyNet

torch.distributed.init_process_group("gloo")
rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
torch.cuda.set_device(0)
w = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "egnn_weights_trainedlike.npz")))
net = pa.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                       condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
sched = pa.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
gam = pa.ConstantAnnealingFactorSchedule(4 / 3)
e = pa.LennardJonesEnergy(39, 13, 3)
single = SimpleNamespace(trainer=SimpleNamespace(world_size=1, global_rank=0))
ok = True
terms_ok = True
stages = {"sde only": dict(num_negative_time_steps=0, post_mcmc_steps=0),
          "+ descent": dict(num_negative_time_steps=5, dt_negative_time=1e-5, post_mcmc_steps=0),
          "+ descent (langevin)": dict(num_negative_time_steps=5, dt_negative_time=1e-5, post_mcmc_steps=0, do_langevin=True),
          "+ MALA": dict(num_negative_time_steps=0, dt_negative_time=1e-7, post_mcmc_steps=3, adaptive_mcmc=False),
          "+ adaptive MALA": dict(num_negative_time_steps=0, dt_negative_time=1e-7, post_mcmc_steps=4, adaptive_mcmc=True),
          "+ resample_at_end": dict(num_negative_time_steps=0, post_mcmc_steps=0, resample_at_end=True, end_resampling_step_override=9),
          "+ descent + adaptive MALA": dict(num_negative_time_steps=5, dt_negative_time=1e-5, post_mcmc_steps=3, adaptive_mcmc=True)}
for debias in (False, True):
    sde = pa.VEReverseSDE(noise_schedule=sched, score_net=pa.ScoreNet(net), energy_net=EnergyNet(copy.deepcopy(net)),
                          debias_inference=debias)
    N, B = (12, 64) if not debias else (4, 32)
    x1 = pa.Prior(scale=3.0, n_particles=13, spatial_dim=3, seed=2).sample(B)
    us = [0.123, 0.456, 0.789, 0.321, 0.654]
    for name, extra in stages.items():
        extra = dict(extra)
        end = extra.pop("end_resampling_step_override", None)
        kw = dict(sde=sde, num_integration_steps=N, start_resampling_step=0,
                  end_resampling_step=N if end is None else min(end, N - 1), resampling_interval=3, seed=11, batch_size=8, **extra)  # chunks of 8 tile both the 16-walker shards and the whole batch
        outs, stats = [], []
        for lm in (None, single):  # None -> torch.distributed world; `single` -> one rank does everything
            integ = pa.WeightedSDEIntegrator(lightning_module=lm, **kw)
            x, logw, uniq, terms, acc = integ.integrate_sde(x1, e, gam, inverse_temperature=1.0, resample_u=us)
            outs.append((x.cpu(), logw.cpu(), uniq, acc))
            stats.append([(float(t.diffusion.mean()), float(t.diffusion.std()), float(t.drift_X.std()), t.diffusion.numel())
                          for t in terms])
        terms_ok = terms_ok and len(stats[0]) == N and np.allclose(stats[0], stats[1], rtol=1e-5, atol=1e-7)
        xa, xb = outs[0][0], outs[1][0]
        bit = torch.equal(xa, xb)
        close = torch.allclose(xa, xb, rtol=1e-4, atol=1e-5)
        finite = int(torch.isfinite(e(xb.cuda())).sum())
        same = (bit or (debias and close)) and outs[0][2] == outs[1][2] and np.allclose(outs[0][3], outs[1][3])
        if rank == 0:
            print(f"debias={debias} {name}: x {'bitwise' if bit else 'allclose' if close else 'DIFFERENT (max %.3g)' % float((xa - xb).abs().max())}, "
                  f"unique {outs[0][2] == outs[1][2]}, rates {outs[0][3]} vs {outs[1][3]}, finite logp {finite}/{B}", flush=True)
        ok = ok and same
if rank == 0:
    print("per-step SDETerms statistics, 2 ranks vs 1 rank: " + ("terms identical" if terms_ok else "TERMS DIFFER"), flush=True)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
sys.exit(0 if (ok and terms_ok) else 1)
