"""World-size-2 gloo tests (CPU) of the multi-GPU host logic: rank/world discovery, the
contiguous walker slicing rule of the reference (sde_integration.py:227-233) and the final
all-gather (X1).  The kernels themselves need a GPU; what is checked here is that shards are cut
and re-assembled correctly and that every rank ends with the same global batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pita_amd.sde_integration import _Comm

        comm = _Comm(None)
        assert (comm.world, comm.rank) == (world, rank)
        Bg, D = 10, 6
        xg = torch.arange(Bg * D, dtype=torch.float32).reshape(Bg, D)
        Bl = Bg // world
        off = rank * Bl
        local = xg[off:off + Bl].clone() * 2.0  # "integrate" the local shard
        out = comm.all_gather(local)
        assert out.shape == (Bg, D)
        assert torch.equal(out, xg * 2.0)
        # a Lightning-style module is honoured when torch.distributed is not the transport
        class LM:
            class trainer:
                world_size, global_rank = world, rank

        c2 = _Comm(LM())
        assert (c2.world, c2.rank) == (world, rank)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _worker_reduce(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace

        from pita_amd.sde_integration import WeightedSDEIntegrator, _Comm, _terms_from_stats

        comm = _Comm(None)
        # per-step moment buffers are summed over ranks before they become TermStats
        st4 = torch.full((3, 4), float(rank + 1), dtype=torch.float64)
        tot = comm.all_reduce_sum(st4)
        assert torch.equal(tot, torch.full((3, 4), 3.0, dtype=torch.float64))
        terms = _terms_from_stats(tot, None, 20, 4, [True] * 3, False)
        assert len(terms) == 3 and abs(float(terms[0].diffusion.mean()) - 3.0 / 20) < 1e-7
        # final gather after MALA: every shard is [valid, set-aside]; the result must be [all valid, all set-aside] (Q7)
        integ = WeightedSDEIntegrator(sde=SimpleNamespace(), num_integration_steps=1, start_resampling_step=0,
                                      end_resampling_step=1, post_mcmc_steps=2)
        Bl = 4
        nvalid = [3, 2][rank]
        integ._last_mala_valid = nvalid
        local = torch.tensor([[10.0 * rank + i] for i in range(Bl)])  # rows 0..nvalid-1 valid, the rest set aside
        out = integ._gather_final(local, comm, Bl)
        assert out.reshape(-1).tolist() == [0.0, 1.0, 2.0, 10.0, 11.0, 3.0, 12.0, 13.0], out.reshape(-1).tolist()
        integ._last_mala_valid = Bl  # nothing set aside anywhere -> plain concatenation
        out = integ._gather_final(local, comm, Bl)
        assert out.reshape(-1).tolist() == [0.0, 1.0, 2.0, 3.0, 10.0, 11.0, 12.0, 13.0]
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_stats_reduction_and_final_gather_order_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_reduce, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_bench_rank_whose_peers_never_arrive_exits_nonzero():
    """--timeout-s: a rank started with WORLD_SIZE=2 whose peer never shows up leaves init_process_group at the deadline
    and exits with code 3 (it does not hang the node, and nothing is re-executed)."""
    import socket
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--timeout-s", "5"],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    assert "init_process_group" in r.stderr and time.time() - t0 < 200


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` (no torchrun environment) must start two rank processes itself and report n_gpus = 2;
    a launcher that started a different number of ranks than --gpus must be refused.  --dry-run: protocol only, gloo."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--dry-run"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 2 and out["config"]["backend"] == "gloo"
    # the multi-rank line explains itself: per-rank wall / launch / all-gather times (gathered from every rank)
    pr = out["per_rank"]
    assert len(pr["wall_s"]) == 2 and len(pr["sampler_launches_s"]) == 2 and len(pr["final_allgather_ms"]) == 2
    assert all(w > 0 for w in pr["wall_s"]) and pr["slowest_rank"] in (0, 1) and pr["max_over_min_wall"] >= 1.0
    assert out["resample_exchange"] is None
    # ... and verifies the ranks it ran on from its own line: an all_reduce of ones read back, every rank's device gathered
    assert out["rccl_ranks_seen"] == 2 and [dv["rank"] for dv in out["devices"]] == [0, 1]
    assert all("device" in dv and "host" in dv for dv in out["devices"])
    # --resample-every: the cost of global resampling events at the shard (log-weight all-gather + _Comm.exchange_rows)
    r3 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                         "--dry-run", "--resample-every", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r3.returncode == 0, r3.stderr[-2000:]
    ex = json.loads([ln for ln in r3.stdout.splitlines() if ln.startswith("{")][0])["resample_exchange"]
    assert ex["every"] == 1 and len(ex["ms_per_event_this_rank"]) == ex["events_timed"] == 3
    assert all(0 <= r <= ex["rows_per_rank"] for r in ex["rows_received_from_other_ranks"])
    assert ex["amortised_ms_per_step"] > 0
    # round 6: what the rank put on the wire, from the uneven all_to_all_single's own split sizes -- with two ranks what
    # rank 0 sends is what rank 1 receives, never more than a shard, and far below the reference's all-gather of every walker
    assert len(ex["rows_sent_to_other_ranks"]) == 3 and all(0 <= r <= ex["rows_per_rank"] for r in ex["rows_sent_to_other_ranks"])
    assert ex["bytes_sent_per_event"] == [r * 3 * 4 for r in ex["rows_sent_to_other_ranks"]]
    assert ex["reference_allgather_bytes_per_event"] == ex["rows_per_rank"] * 3 * 4
    sp = ex["last_event_rows_by_peer"]
    assert len(sp["sent_to"]) == 2 and sum(sp["received_from"]) >= ex["rows_received_from_other_ranks"][-1]
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run"], capture_output=True, text=True,
                        timeout=300, env=env)
    assert r1.returncode == 0 and json.loads(r1.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    bad = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                        text=True, timeout=300, env=bad)
    assert r2.returncode != 0 and "WORLD_SIZE=3" in (r2.stderr + r2.stdout)


def test_single_rank_comm():
    from pita_amd.sde_integration import _Comm

    c = _Comm(None)
    assert (c.world, c.rank) == (1, 0)
    x = torch.randn(4, 3)
    assert c.all_gather(x) is x


def _worker_exchange(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pita_amd.sde_integration import _Comm

        comm = _Comm(None)
        Bl, D = 7, 5
        n = world * Bl
        xg = torch.arange(n * D, dtype=torch.float32).reshape(n, D)
        x = xg[rank * Bl:(rank + 1) * Bl].clone()
        gen = torch.Generator().manual_seed(5)
        patterns = [torch.arange(n),                                        # nobody moves
                    torch.zeros(n, dtype=torch.int64),                      # one parent for everybody
                    torch.full((n,), n - 1, dtype=torch.int64),             # ... living on the last rank
                    torch.sort(torch.randint(0, n, (n,), generator=gen))[0],
                    torch.sort(torch.randint(0, Bl, (n,), generator=gen))[0],        # all parents on rank 0
                    torch.sort(torch.randint(n - 3, n, (n,), generator=gen))[0],
                    torch.roll(torch.arange(n), -3),                        # the rotation by the event's uniform
                    torch.roll(torch.sort(torch.randint(0, n, (n,), generator=gen))[0], 5)]
        moved = []
        for ids in patterns:
            before = getattr(comm, "rows_received", 0)
            got = comm.exchange_rows(x, ids, Bl)
            assert torch.equal(got, xg[ids][rank * Bl:(rank + 1) * Bl]), ids
            moved.append(comm.rows_received - before)
        assert moved[0] == 0                      # identity resampling: no walker crosses a rank boundary
        assert moved[1] == (0 if rank == 0 else 1)  # one distinct parent travels to each other rank, once
        assert all(m <= Bl for m in moved)        # never more than a shard (the all-gather moves (world-1) shards)
        # the resampling uniform is rank 0's on every rank
        u = comm.shared_uniform(0.25 + 0.5 * rank)
        assert float(u) == 0.25
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_resampling_exchange_moves_only_needed_parents(world):
    """_Comm.exchange_rows (the all_to_all_single that replaces the reference's all-gather of every walker at a
    resampling event, sde_integration.py:248-258) returns exactly the local slice of gathered[ids] and moves at most the
    distinct remote parents a rank needs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_exchange, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def _worker_scale(rank, world, port, q, Bl, D):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np

        from oracle import pita_oracle as O
        from pita_amd.sde_integration import _Comm

        torch.set_num_threads(1)
        comm = _Comm(None)
        n = world * Bl
        gen = torch.Generator().manual_seed(1234)  # same stream on every rank: the "global" batch
        xg = torch.randn(n, D, generator=gen)
        x = xg[rank * Bl:(rank + 1) * Bl].clone()
        # the event's uniform is rank 0's; the log-weights are all-gathered (4 B per walker), ids computed identically
        u = float(comm.shared_uniform(0.1 + 0.01 * rank))
        assert u == 0.1
        for spread in (0.0, 0.3, 3.0):  # flat weights (nobody moves), mild, a few heavy parents
            a_local = spread * torch.randn(n, generator=gen)[rank * Bl:(rank + 1) * Bl]
            ag = comm.all_gather(a_local.contiguous())
            assert ag.shape == (n,)
            ids = torch.from_numpy(np.asarray(O.sample_cat_sys(ag, u), dtype=np.int64))
            before = getattr(comm, "rows_received", 0)
            got = comm.exchange_rows(x, ids, Bl)
            assert torch.equal(got, xg[ids][rank * Bl:(rank + 1) * Bl])
            moved = comm.rows_received - before
            assert moved <= Bl
            if spread == 0.0:
                # flat weights: ids are the rotation by floor(u n) (utils.py:111-120: (u + k / n) mod 1 against bins
                # j / n), so a rank receives exactly that many rows from its successor -- and nothing else
                assert abs(moved - min(int(u * n), Bl)) <= 1, (moved, int(u * n))
        # MALA's global acceptance count and the per-step moment buffers: sums over ranks
        cnt = torch.tensor([rank + 1], dtype=torch.int32)
        dist.all_reduce(cnt)
        assert int(cnt) == world * (world + 1) // 2
        st = comm.all_reduce_sum(torch.full((5, 4), 1.0 + rank, dtype=torch.float64))
        assert torch.equal(st, torch.full((5, 4), world * (world + 1) / 2, dtype=torch.float64))
        # X1: the final all-gather returns the global batch in rank order
        out = comm.all_gather(x)
        assert out.shape == (n, D) and torch.equal(out, xg)
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,Bl,D", [(4, 4096, 66), (8, 2048, 165)])
def test_multi_gpu_host_path_at_config_shard_shapes(world, Bl, D):
    """Configs C4 (16 384 walkers x 66 over 4 GPUs = 4 096 per rank) and C5 (262 144 x 165 over 8 GPUs = 32 768 per rank,
    here reduced 16x to 2 048 per rank): world sizes 4 and 8 over gloo run the whole host-side protocol of a resampling
    event (shared uniform, log-weight all-gather, identical ids, uneven all_to_all_single of the distinct parents), the
    acceptance / moment reductions and the final all-gather.  RCCL itself needs the 8-GPU node: unmeasured here."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_scale, args=(r, world, port, q, Bl, D)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res
