"""Multi-rank behaviour on the GPU box (one MI355X): two rank processes share cuda:0 over gloo (RCCL refuses two ranks
on one device), which exercises everything of the N > 1 path except the RCCL transport itself: rank slicing, Philox
keys by global walker index, global resampling, the MALA acceptance all-reduce, the [valid, set-aside] final order and
bench.py's launch / timing protocol."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_two_ranks_reproduce_one_rank_bitwise():
    """tools/rehearse_multirank.py: integrate_sde in 14 combinations (both regimes x {SDE only, descent, Langevin
    descent, MALA, adaptive MALA, resample_at_end, descent + adaptive MALA}); the 2-rank result must equal the
    1-rank result (bitwise in the not-debiased regime), same resampling counts and acceptance rates."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "tools", "rehearse_multirank.py")],
                       capture_output=True, text=True, timeout=900, env=_env())
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("debias=")]
    assert len(lines) == 14 and all("DIFFERENT" not in ln for ln in lines), r.stdout
    assert sum("bitwise" in ln for ln in lines if ln.startswith("debias=False")) == 7, r.stdout
    assert any("terms identical" in ln for ln in r.stdout.splitlines()), r.stdout


def test_bench_two_ranks_on_one_device():
    """`python bench.py --gpus 2` starts its own two ranks (before touching the GPU) and rank 0 reports n_gpus = 2 and
    the whole-job rate; PITA_BENCH_ONE_DEVICE=1 puts both ranks on cuda:0 over gloo."""
    env = _env()
    env["PITA_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--walkers", "1024", "--no-cpu-baseline", "--no-debiased", "--force-evals", "0",
                        "--resample-every", "1"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_walkers"] == 2048 and out["scaling"] == "weak"
    assert out["value"] > 0 and 0 < out["roofline"]["frac"] <= 1
    # the multi-rank line attributes its time: per-rank wall / launch / all-gather seconds, and the optional exchange leg
    pr = out["per_rank"]
    assert len(pr["wall_s"]) == 2 and all(0 < a <= w for a, w in zip(pr["sampler_launches_s"], pr["wall_s"]))
    assert all(ms >= 0 for ms in pr["final_allgather_ms"]) and pr["slowest_rank"] in (0, 1)
    ex = out["resample_exchange"]
    assert ex["rows_per_rank"] == 1024 and len(ex["ms_per_event_this_rank"]) == 3
    assert all(0 < r <= 1024 for r in ex["rows_received_from_other_ranks"])
    assert all(0 < r <= 1024 for r in ex["rows_sent_to_other_ranks"]) and ex["bytes_sent_per_event"][0] == ex["rows_sent_to_other_ranks"][0] * 39 * 4
    # round 6: the whole integrate_sde over the global batch, wall seconds of every rank (attribution of a scaling loss)
    e2 = out["e2e"]["not_debiased"]
    assert e2["global_walkers"] == 2048 and e2["gathered_rows"] == 2048 and e2["finite"] and len(e2["per_rank_wall_s"]) == 2
    assert all(w > 0 for w in e2["per_rank_wall_s"]) and e2["seconds"] == max(e2["per_rank_wall_s"])
