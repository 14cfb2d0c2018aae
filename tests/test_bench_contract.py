"""bench.py prints ONE JSON line with the fields the driver and the judge read; smoke() runs the hot path against the
oracle.  Small shapes: the numbers themselves are bench.py's business, the contract is what is checked here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout          # exactly one line on stdout
    return json.loads(lines[0])


def test_bench_defaults_are_the_contract_defaults():
    """no flags = 1 GPU and a K/W that let the clocks settle (a step is 0.2 ms)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "args.gpus = int(env_world) if env_world else 1" in src      # no flag, no WORLD_SIZE: one GPU
    assert "FRAMES_1GPU = 4096" in src and "FRAMES_PER_GPU_SHARDED = 8192" in src   # configs[1] / configs[3] shapes
    assert '"--steps", type=int, default=200' in src and '"--warmup", type=int, default=20' in src


@pytest.mark.gpu
def test_bench_json_line():
    d = run_bench("--steps", "5", "--warmup", "2", "--frames", "256", "--cpu-frames", "8")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "ranks", "gpu_shared",
              "devices", "library", "control_plane"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["ranks"] == 1 and d["gpu_shared"] is False and len(d["devices"]) == 1 and d["devices"][0]["rank"] == 0
    assert "shard_8192" not in d and "config3" not in d and "hist" not in d and "streams" not in d and "awgn" not in d        # only beside config 2 itself (4096 x 16384), see test_bench_shard_key below
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "Msamples/s" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # value is the whole job over the wall clock of the K steps
    want = d["config"]["frames_per_gpu"] * d["config"]["frame_size"] / (d["ms_per_step"] * 1e-3) / 1e6
    assert abs(d["value"] - want) / want < 1e-6
    rl = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rl, k
    assert rl["bound"] == "hbm" and rl["unit"] == "GB/s" and rl["peak"] == 8000.0
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch over the kernel's average duration
    assert abs(rl["achieved"] - rl["algorithmic_bytes_per_launch"] / (rl["kernel_ms"] * 1e-3) / 1e9) / rl["achieved"] < 1e-9
    assert rl["algorithmic_bytes_per_launch"] == 8 * 256 * d["config"]["frame_size"]
    # provenance: the kernel name comes from the library, the traffic constant names its source (none for this shape),
    # the library file that was actually loaded is identified by path and hash
    assert rl["kernel"] in ("rx_lean_kernel", "rx_fused_pipe_kernel", "rx_pipe2_kernel", "rx_fused_kernel") and "traffic_source" in rl
    assert rl["traffic"] is None and rl["traffic_source"] is None
    import hashlib
    lib = d["library"]
    assert lib["override_QPSK_HIP_LIB"] is False and lib["path"] == os.path.join("qpsk_amd", "libqpsk_hip.so")
    assert lib["sha256"] == hashlib.sha256(open(os.path.join(ROOT, lib["path"]), "rb").read()).hexdigest()
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    # the value is the build's own restatement at -O2 on ONE core; the untouched reference (-O0, oracle/_ref) is a side key
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["value"] == cb["port_1core_msps"]
    assert cb["ref_so"].startswith("present") == ("reference_makefile_flags_msps" in cb)      # the line says whether oracle/_ref travelled
    assert "stimulus" in d and "qpsk_tx_symbols" in d["stimulus"]                  # frames from the library's own transmit chain
    p = d["parity"]
    assert p["symbol_mismatches"] == 0 and p["freq_bit_mismatches"] == 0 and p["phase_bit_mismatches"] == 0
    assert p["hz_frames_checked"] == 256 and p["hz_out_of_range"] == 0       # EVERY frame locked on the +50 Hz carrier


def test_bench_shard_key_is_wired_to_config_2():
    """the 8192-frame per-GPU share rides along in the N = 1 line as `shard_8192` (its own timed region, the same
    code), without touching `value` / `config` / `roofline`"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'res["shard_8192"] = sh' in src and "(F, L) == (FRAMES_1GPU, 16384) and not args.no_shard" in src
    assert src.count("timed_region(") == 6          # the definition and its uses: config 2, the timing modes, histogram mode's two-launch route, the AWGN variant, the shard
    # round 6: burst and sustained, the result gather through the C host layer (item 5)
    assert '"sustained_ms_per_step"' in src and 'sh["sustained_ms_per_step"]' in src and 'res["gather"]' in src and "kernel_ms_event_pair_per_launch\"]" not in src
    # BASELINE configs[2] (FFT timing estimate in front) and the reference's histogram mode ride along the same way
    assert 'res[key] = ent' in src and '("config3", qpsk_amd.TIMING_FFT' in src and '("hist", qpsk_amd.TIMING_HIST' in src


def test_control_plane_is_gloo_only():
    """no rank ever creates an RCCL communicator: the N = 8 run executes the control code of the rehearsals"""
    src = open(os.path.join(ROOT, "bench.py")).read() + open(os.path.join(ROOT, "qpsk_amd", "shard.py")).read()
    assert 'init_process_group("gloo"' in src and '"nccl"' not in src.replace('backend "nccl"', "")
    from qpsk_amd.shard import distinct_devices, init_distributed
    with pytest.raises(ValueError):
        os.environ["WORLD_SIZE"] = "2"
        try:
            init_distributed("nccl")
        finally:
            del os.environ["WORLD_SIZE"]
    a = {"rank": 0, "host": "h", "pci_bus_id": "0000:05:00", "uuid": None, "visible": None, "device": 0}
    b = dict(a, rank=1)
    c = dict(a, rank=2, pci_bus_id="0000:06:00", device=1)
    assert distinct_devices([a]) == 1 and distinct_devices([a, b]) == 1 and distinct_devices([a, b, c]) == 2
    # per-rank visibility masks: every rank sees "device 0" of a different GPU
    m0 = {"rank": 0, "host": "h", "pci_bus_id": None, "uuid": None, "visible": "3", "device": 0}
    m1 = dict(m0, rank=1, visible="4")
    assert distinct_devices([m0, m1]) == 2


@pytest.mark.gpu
def test_smoke_entry_point():
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.smoke()
