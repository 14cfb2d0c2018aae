"""Multi-GPU plumbing of the receive path: independent frames shard across ranks with no data-path
collective (every frame owns its delay line, loop state and timing decision -- the
reference's globals qpsk.c:36-53, costas_loop.c:13-23 become per-frame state).

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests); the only
communication is the barrier around the timed region and a MAX-reduce of the elapsed time.
"""
import os


def shard_range(total_frames, rank, world):
    """Contiguous frame range [lo, hi) of `rank`: GPU g gets frames [g*F/G, (g+1)*F/G)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    lo = (total_frames * rank) // world
    hi = (total_frames * (rank + 1)) // world
    return lo, hi


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def local_device(local_rank, device_count):
    """The GPU of a rank: its own (LOCAL_RANK) on a node with one GPU per rank; on a box with fewer GPUs than
    ranks (a rehearsal: results are the same, timings are not a scaling measurement) ranks share round-robin."""
    if device_count <= 0:
        raise ValueError("no GPU")
    return local_rank % device_count


def init_distributed(backend, device=None):
    """Returns the torch.distributed module (initialised) or None for a single process."""
    rank, _, world = env_rank_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {}
    if device is not None and backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def max_over_ranks(value, dist, device=None):
    """MAX of a python float over all ranks (the slowest rank defines the job's time)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist, device=None):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
