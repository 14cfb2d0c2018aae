"""The N > 1 path on CPU: two processes over gloo shard a batch of independent frames exactly as bench.py
does on GPUs (qpsk_amd/shard.py: contiguous frame ranges, no data-path collective, barrier + MAX-reduce of
the elapsed time).  Each rank demodulates ITS shard (here with the CPU oracle -- this is a test of the
sharding logic, which is all that multi-GPU adds to this path); the gathered result must equal the
single-process result over the whole batch, bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as tdist
from qpsk_amd.shard import env_rank_world, init_distributed, max_over_ranks, sum_over_ranks, shard_range
from oracle.pyoracle import Oracle, TIMING_FIXED
from sigutil import make_frames
rank, local, world = env_rank_world()
dist = init_distributed("gloo")
assert dist is not None and dist.get_world_size() == world == int(os.environ["EXPECT_WORLD"])
FS, RS, L, TOTAL = 19200.0, 2400.0, 1024, 8 * world - 5   # uneven split: 5 + 6 frames on two ranks, 7 + 7 x 8 ... on eight
orc = Oracle()
taps = orc.rrc_make(FS, RS, np.float32(.35))
lo, hi = shard_range(TOTAL, rank, world)
x, _ = make_frames(hi - lo, L, 8, taps, FS, base_seed=77, first_frame=lo)   # frame f is the same on every rank
dist.barrier()
t0 = time.perf_counter()
out = orc.rx_batch(x, FS, RS, timing_mode=TIMING_FIXED, fixed_index=6, threads=1)
dist.barrier()
elapsed = time.perf_counter() - t0
tmax = max_over_ranks(elapsed, dist)
nframes = sum_over_ranks(hi - lo, dist)
assert tmax >= elapsed and nframes == TOTAL
gathered = [None] * world
dist.all_gather_object(gathered, (lo, hi, out["sym"], out["freq"], out["phase"]))
if rank == 0:
    covered = sorted((a, b) for a, b, *_ in gathered)
    assert covered[0][0] == 0 and covered[-1][1] == TOTAL and all(covered[i][1] == covered[i + 1][0] for i in range(world - 1))
    sym = np.concatenate([g[2] for g in sorted(gathered, key=lambda g: g[0])])
    freq = np.concatenate([g[3] for g in sorted(gathered, key=lambda g: g[0])])
    xa, _ = make_frames(TOTAL, L, 8, taps, FS, base_seed=77)
    want = orc.rx_batch(xa, FS, RS, timing_mode=TIMING_FIXED, fixed_index=6, threads=1)
    assert np.array_equal(sym, want["sym"]) and np.array_equal(freq.view(np.uint32), want["freq"].view(np.uint32))
    print("SHARD_OK", tmax > 0)
dist.barrier()
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_ranges_tile_the_batch():
    from qpsk_amd.shard import shard_range
    for total in (1, 7, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            edges = [shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(65536, 3, 8) == (24576, 32768)      # config 4: 8192 frames per GPU
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_over_gloo(tmp_path, oracle, world):
    """world = 8: the control plane of BASELINE config 4 (65536 frames over 8 GPUs) at its real rank count"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), EXPECT_WORLD=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARD_OK True" in outs[0]
