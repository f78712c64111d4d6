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


class GatherPipeline:
    """Double-buffered all-gather for a loop that traces the same shard repeatedly: step k writes
    buffer k mod 2 and starts its gather asynchronously; the gather of step k overlaps the trace
    of step k + 1 (which writes the other buffer).  Under RCCL the collective runs on the
    backend's own stream (`wait()` makes the current stream wait for it); under gloo -- the
    rehearsal backend -- the shard is staged through the host."""

    def __init__(self, per, world, n_rays, dist, device, dtype=torch.float32):
        self.per, self.world, self.n_rays, self.dist = per, world, n_rays, dist
        self.host = world > 1 and dist.get_backend() == "gloo" and torch.device(device).type == "cuda"
        self.outs = [torch.zeros(per, dtype=dtype, device=device) for _ in range(2)]
        full_dev = "cpu" if self.host else device
        self.fulls = [torch.empty(per * world, dtype=dtype, device=full_dev) for _ in range(2)]
        self.works = [None, None]
        self.staged = [None, None]

    def buffer(self, k):
        """The output buffer of step k, safe to overwrite (its previous gather has finished)."""
        b = k & 1
        if self.works[b] is not None:
            self.works[b].wait()
            self.works[b] = None
        return self.outs[b]

    def gather(self, k):
        """Start the gather of step k's buffer; returns the tensor that will hold all results."""
        b = k & 1
        if self.world == 1:
            return self.outs[b][: self.n_rays]
        src = self.outs[b]
        if self.host:
            self.staged[b] = src.cpu()          # (synchronises with the trace: rehearsal only)
            src = self.staged[b]
        self.works[b] = self.dist.all_gather_into_tensor(self.fulls[b], src, async_op=True)
        return self.fulls[b][: self.n_rays]

    def drain(self):
        for b in range(2):
            if self.works[b] is not None:
                self.works[b].wait()
                self.works[b] = None
