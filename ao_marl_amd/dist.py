"""Multi-GPU: environments shard embarrassingly (one process per GPU, no data-path collective).
The only exchange is gathering per-environment episode returns (E floats per rank per episode)."""
import torch
import torch.distributed as dist


def shard_seeds(base_seed, envs_per_rank, rank, stride=16):
    """First seed of rank `rank`: ranks own consecutive blocks of the global seed sequence."""
    return base_seed + stride * envs_per_rank * rank


def gather_episode_returns(returns):
    """returns: [E] tensor on this rank -> [world*E] on every rank (all_gather; RCCL on GPUs,
    gloo on CPU).  Latency-bound: <= 1 KB per rank."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return returns.clone()
    out = [torch.empty_like(returns) for _ in range(dist.get_world_size())]
    dist.all_gather(out, returns.contiguous())
    return torch.cat(out)
