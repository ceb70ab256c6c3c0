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


def _learner_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ao_marl_amd.agents import AgentLayout
    from ao_marl_amd.sac import BatchedSAC
    lay = AgentLayout(12, [0, 8], 2, include_tip_tilt=True, state_keys=("dm_history_1", "dm_before_linear", "dm_residual"),
                      state_block=10)
    # different initial weights per rank on purpose: sync_learners(init=True) must make them rank 0's
    sac = BatchedSAC(lay, dict(hidden_size_actor=16, hidden_size_critic=16, batch_size=8), seed=100 + rank,
                     device="cpu", memory_size=64, native=False)
    sac.sync_learners(init=True)
    start = torch.cat([t.reshape(-1).clone() for t in (sac._pflat, sac._cflat, sac._ctflat)])
    g = torch.Generator().manual_seed(7 + rank)              # every rank learns from its own transitions
    n = 32
    sac.memory.push(torch.randn(n, lay.state_dim, generator=g), torch.rand(n, lay.action_dim, generator=g) * 2 - 1,
                    -torch.rand(n, lay.n_agents, generator=g), torch.randn(n, lay.state_dim, generator=g), 1.0)
    for _ in range(3):
        sac.update_from_memory(8)
    mine = [t.detach().clone() for t in sac._learner_state()]
    sac.sync_learners()
    after = [t.detach().clone() for t in sac._learner_state()]
    q.put((rank, start.numpy(), [t.numpy() for t in mine], [t.numpy() for t in after],
           sac.alpha.reshape(-1).numpy().copy(), sac.log_alpha.detach().reshape(-1).numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_learners_start_equal_and_are_averaged():
    """ranks must not train diverging learners whose returns are silently averaged.
    train_agent calls BatchedSAC.sync_learners: rank 0's weights at the start, the mean of the ranks'
    parameters / temperature / Adam moments after every episode's updates."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_learner_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, s0, m0, a0, al0, la0), (_, s1, m1, a1, al1, la1) = res
    assert np.array_equal(s0, s1)                           # equal after the initial broadcast
    assert len(m0) == len(m1) == len(a0) >= 4 + 2 * 3       # parameters, temperature, Adam moments
    moved = 0.0
    for x0, x1, y0, y1 in zip(m0, m1, a0, a1):
        assert np.array_equal(y0, y1)                       # one learner afterwards
        assert np.allclose(y0, 0.5 * (x0 + x1), rtol=0, atol=1e-7)
        moved = max(moved, float(np.abs(x0 - x1).max()))
    assert moved > 0                                        # the ranks really learned different things
    assert np.allclose(al0, np.exp(la0)) and np.array_equal(al0, al1)


def test_four_rank_learners_and_shards():
    """World size 4 (the scaling runs go to 8; nothing here is specific to two ranks): one global seed sequence in
    four disjoint shards, every rank sees all sixteen returns in order, the learners start from rank 0's weights and
    end on the mean of the four."""
    world = 4
    ctx = mp.get_context("spawn")
    q, q2 = ctx.Queue(), ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_seeds = sum((r[1] for r in res), [])
    assert all_seeds == (1234 + 16 * np.arange(16)).tolist()
    assert all(r[2] == [0.5 * s for s in all_seeds] and r[3] == 4.0 for r in res)
    port = _free_port()
    ps = [ctx.Process(target=_learner_worker, args=(r, world, port, q2)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q2.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(np.array_equal(res[0][1], r[1]) for r in res[1:])             # rank 0's weights everywhere at the start
    for k in range(len(res[0][2])):
        mean = sum(r[2][k].astype(np.float64) for r in res) / world
        for r in res:
            assert np.array_equal(r[3][k], res[0][3][k])                    # one learner afterwards
            assert np.allclose(r[3][k], mean, rtol=0, atol=2e-7)
    assert max(float(np.abs(res[0][2][k] - res[3][2][k]).max()) for k in range(len(res[0][2]))) > 0
