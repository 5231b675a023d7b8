import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import pita_amd
B = 65536
w = dict(np.load("tests/golden/egnn_weights_trainedlike.npz"))
net = pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                             condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
x = pita_amd.Prior(scale=3.0, n_particles=13, spatial_dim=3).sample(B)
beta = torch.ones(B, device="cuda")
for hv in (25.7, 1.0, 0.01):
    h = torch.full((B,), hv, device="cuda")
    for mode in ("tangent", "reduce"):
        kw = dict(want_primal=False) if mode == "tangent" else dict(want_primal=False, want_tangent=False, dot_out=torch.empty(B, 39, device="cuda"), diag_acc=torch.zeros(B, device="cuda"))
        net.jvp(h, x, beta, direction=0, **kw); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(10): net.jvp(h, x, beta, direction=k, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"h={hv} {mode}: {dt*1e3:.2f} ms per JVP launch")
    t0 = time.perf_counter()
    for k in range(10): net.edm(2, h, x, beta)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"h={hv} forward score: {dt*1e3:.2f} ms")
