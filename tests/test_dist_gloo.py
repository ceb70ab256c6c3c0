"""N > 1 path on CPU: world_size-2 gloo, one process per 'GPU', independent seed shards and the
single collective of the hot path (all_gather of episode returns)."""
import os
import socket

import numpy as np
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
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ao_marl_amd.dist import gather_episode_returns, shard_seeds
    E = 4
    first = shard_seeds(1234, E, rank)
    seeds = first + 16 * np.arange(E)
    ret = torch.tensor(seeds, dtype=torch.float32) * 0.5     # stand-in for episode returns
    allr = gather_episode_returns(ret)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)      # bench.py's MAX-over-ranks timing
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, seeds.tolist(), allr.tolist(), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_and_gather():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_seeds = res[0][1] + res[1][1]
    assert len(set(all_seeds)) == 8                         # disjoint shards
    assert all_seeds == (1234 + 16 * np.arange(8)).tolist()  # one global sequence
    for r in res:
        assert r[2] == [0.5 * s for s in all_seeds]          # every rank sees all returns, in order
        assert r[3] == 2.0
