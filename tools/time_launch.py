import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import pita_amd
w = dict(np.load("tests/golden/egnn_weights_trainedlike.npz"))
net = pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                             condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
tab = pita_amd.sde_integration.build_step_table(sched, gam, torch.linspace(1.0, 0.0, 1001)[:-1], 1e-3, 1.0, 1.0).cuda()
x = pita_amd.Prior(scale=69.28, n_particles=13, spatial_dim=3, seed=1).sample(65536)
for chunk in (1, 5, 100):
    n = 100 // chunk
    net.sampler_run(x, tab[:chunk], chunk); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): net.sampler_run(x, tab[i*chunk:(i+1)*chunk], chunk, step0=i*chunk)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"chunk={chunk}: host issue {1e3*(t1-t0)/n:.3f} ms/launch, total {1e3*(t2-t0)/100:.3f} ms/step")
