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


def test_single_rank_comm():
    from pita_amd.sde_integration import _Comm

    c = _Comm(None)
    assert (c.world, c.rank) == (1, 0)
    x = torch.randn(4, 3)
    assert c.all_gather(x) is x
