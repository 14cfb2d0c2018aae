"""Multi-GPU plumbing of the receive path: independent frames shard across ranks with no data-path
collective (every frame owns its delay line, loop state and timing decision -- the
reference's globals qpsk.c:36-53, costas_loop.c:13-23 become per-frame state).

One process per GPU.  The ONLY communication of a job is its control plane -- the barrier around the timed
region, a MAX-reduce of the elapsed time and a gather of each rank's device identity -- a few bytes of host
data.  It runs over gloo on CPU tensors WHATEVER the number of GPUs (BASELINE.json north_star: "no RCCL
collective required"), so the code a rank executes on an 8-GPU node is byte for byte the code the two-rank
rehearsal on a one-GPU box and the CPU tests execute; no rank ever creates an RCCL communicator, and xGMI
carries nothing.
"""
import datetime
import os
import socket


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


def init_distributed(backend="gloo", device=None, timeout_s=180):
    """Returns the torch.distributed module (initialised) or None for a single process.  The control plane is
    gloo (see the module docstring); `backend` exists for the tests that say so explicitly.  A rank that never
    arrives (bad device, import error) ends the rendezvous after timeout_s instead of the backend's 30 minutes."""
    rank, _, world = env_rank_world()
    if world <= 1:
        return None
    if backend != "gloo":
        raise ValueError("the control plane runs over gloo; the data path has no collective (got %r)" % (backend,))
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    return dist


def _reduce(value, dist, op_name):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)      # a CPU tensor: gloo
    dist.all_reduce(t, op=getattr(dist.ReduceOp, op_name))
    return float(t.item())


def max_over_ranks(value, dist, device=None):
    """MAX of a python float over all ranks (the slowest rank defines the job's time).  `device` is ignored:
    the reduction is host data over gloo."""
    return _reduce(value, dist, "MAX")


def sum_over_ranks(value, dist, device=None):
    return _reduce(value, dist, "SUM")


def device_identity(torch, device_index):
    """What tells two GPUs apart: host name + PCI address (and the UUID where the runtime exposes it)."""
    rank, local, _ = env_rank_world()
    p = torch.cuda.get_device_properties(device_index)
    pci = None
    if all(hasattr(p, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        pci = "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    uuid = getattr(p, "uuid", None)
    return {"rank": rank, "local_rank": local, "host": socket.gethostname(), "device": int(device_index),
            "name": p.name, "pci_bus_id": pci, "uuid": str(uuid) if uuid is not None else None,
            "visible": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")}


def gather_identities(ident, dist):
    """Every rank's device_identity(), in rank order, on every rank."""
    if dist is None:
        return [ident]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, ident)
    return sorted(out, key=lambda d: d["rank"])


def distinct_devices(idents):
    """Number of different physical GPUs behind a list of identities (ranks that share a GPU count once).  The
    PCI address tells devices apart when the runtime reports it; under per-rank *_VISIBLE_DEVICES masks every
    rank sees "device 0", so the mask joins the key."""
    keys = set()
    for d in idents:
        keys.add((d["host"], d["pci_bus_id"] or d["uuid"] or (d["visible"], d["device"])))
    return len(keys)
