"""Ray sharding across the GPUs of one node (not in the reference, which is
single-GPU).  Rays are independent: rank r traces a contiguous range of the ray batch, the
BVH and particles are replicated, and the per-ray results (4 B/ray) are all-gathered.  No
other collective is on the data path.  Pure index arithmetic + one torch.distributed call,
so the same code runs under gloo on CPU tensors (tests) and RCCL on device tensors (bench).
"""
import torch


def shard_size(n_rays, world):
    """Rays per rank: ceil(n / world) rounded up to a whole number of 64-ray packets."""
    per = (n_rays + world - 1) // world
    return (per + 63) // 64 * 64


def shard_bounds(n_rays, world, rank):
    per = shard_size(n_rays, world)
    lo = min(rank * per, n_rays)
    return lo, min(lo + per, n_rays)


def gather_results(my_out, n_rays, world, dist=None):
    """my_out: this rank's padded shard (shard_size elements).  Returns the n_rays results in
    ray order on every rank."""
    if world == 1:
        return my_out[:n_rays]
    if my_out.is_cuda and dist.get_backend() == "gloo":
        # rehearsal without RCCL (several ranks sharing one GPU): gloo gathers host tensors
        host = torch.empty(my_out.numel() * world, dtype=my_out.dtype)
        dist.all_gather_into_tensor(host, my_out.cpu())
        return host.to(my_out.device)[:n_rays]
    full = torch.empty(my_out.numel() * world, dtype=my_out.dtype, device=my_out.device)
    dist.all_gather_into_tensor(full, my_out)
    return full[:n_rays]
