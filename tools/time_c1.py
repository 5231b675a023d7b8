"""BASELINE config C1 on the GPU: 40-mode GMM, MyMLP 128x3 score net, 100 SDE steps, 1 024 walkers (and a large batch)."""
import sys, time, torch
sys.path.insert(0, ".")
import pita_amd as pa
from pita_amd import mlp
torch.manual_seed(12345)
net = mlp.MyMLP(hidden_size=128, hidden_layers=3, emb_size=128, out_dim=2, input_dim=2)
sched = pa.ElucidatingNoiseSchedule(sigma_min=0.01, sigma_max=80.0, rho=7)
sde = pa.VEReverseSDE(noise_schedule=sched, score_net=pa.ScoreNet(net), debias_inference=False)
for B in (1024, 1 << 20):
    for rec in (False, True):
        integ = pa.WeightedSDEIntegrator(sde=sde, num_integration_steps=100, start_resampling_step=0, end_resampling_step=100,
                                         resampling_interval=-1, num_negative_time_steps=0, post_mcmc_steps=0,
                                         should_mean_free=False, record_terms=rec)
        x1 = torch.randn(B, 2, device="cuda") * 80
        integ.integrate_sde(x1, pa.GMM(), pa.ConstantAnnealingFactorSchedule(1.0)); torch.cuda.synchronize()
        t0 = time.perf_counter()
        integ.integrate_sde(x1, pa.GMM(), pa.ConstantAnnealingFactorSchedule(1.0)); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"C1 B={B} {'per-step' if rec else 'fused'}: {dt*1e3:.2f} ms for 100 steps -> {B*100/dt:.3e} walker-steps/s")
