"""bench.py contract: one JSON line with the required keys (GPU), and the CLI parses (CPU)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def test_bench_help_runs_without_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_command_is_the_drivers_multi_rank_form():
    b = _bench_module()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5", "--global-paths", "1000000000"]
    cmd = b.launcher_command(argv, 8, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                      # the caller's flags reach every rank unchanged


def test_workload_names_follow_the_numbers_run():
    b = _bench_module()
    assert "BASELINE configs[1]" in b.workload_name("european252", 10_000_000, 10_000_000, 1, False)
    assert "BASELINE configs[1]" not in b.workload_name("european252", 1_000_000, 1_000_000, 1, False)
    s = b.workload_name("european252", 125_000_000, 1_000_000_000, 8, True)
    assert "1B paths" in s and "8 GPU" in s and "configs[4]" in s and "configs[1]" not in s
    assert "configs[2]" in b.workload_name("store", 100_000_000, 100_000_000, 1, False)
    assert "configs[3]" in b.workload_name("nmc", 65_536, 65_536, 1, False)


def test_bare_gpus_2_self_launches_two_ranks_over_gloo():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent spawns 2 ranks (torch.distributed.run, 127.0.0.1),
    they rendezvous, shard the global path range and all-reduce once; rank 0's single JSON line is relayed.
    --rehearse-launch prices nothing, so this runs on a box with no GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                          "--backend", "gloo", "--rehearse-launch", "--global-paths", "1000000001"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["paths_covered"] == 1_000_000_001
    assert d["steps"] == 4 and d["warmup"] == 1 and d["rehearsal"] is True and d["value"] is None


def test_bare_gpus_n_fails_loudly_without_n_devices():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the devices")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "visible GPUs" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_bench_emits_one_json_line_with_roofline_and_cpu_baseline():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--paths", "1000000", "--cpu-sample-paths", "20000", "--no-store-roofline",
                          "--no-accuracy", "--no-sweep", "--no-nmc"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "paths/s" and d["dtype"] == "f64" and d["vs_baseline"] is None and d["value"] > 1e7
    assert "workload" in d["config"] and "model" not in d["config"]
    rl = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rl, k
    assert 0.2 < rl["frac"] <= 1.05
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0
    assert d["within_3se"] in (True, False) and abs(d["price"] - 13.2697) < 0.1
    dev = d["device"]
    assert dev["arch"].startswith("gfx950") and dev["compute_units"] >= 200 and dev["hbm_free_gb"] > 100
    assert rl["valu_slots_per_path_step"] and "stale" not in rl     # the slot count describes the loaded library
    # the fixed yardstick (operation floor of the scheme) and the PMC-derived VALU occupancy ride along
    assert rl["valu_slots_floor"] <= rl["valu_slots_per_path_step"] and 0.2 < rl["frac_vs_floor"] <= rl["frac"]
    assert "valu_busy" in rl and (rl["valu_busy"] is None or "value" in rl["valu_busy"])
    if rl["valu_busy"] and rl["valu_busy"]["value"] is not None:    # digest taken from this very library
        assert 0.5 < rl["valu_busy"]["value"] < 1.1
    assert d["cfg1"]["closed_form"]["evals"] == 1_000_000 and d["cfg1"]["serial_mc_port_f64"]["paths"] == 1_000_000


def _bench_line(argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                         timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("workload,strategy", [("nmc", "wave"), ("nmc", "fused"), ("nmc", "block"), ("store", None)])
def test_two_ranks_sharing_the_card_reproduce_the_one_rank_job(workload, strategy):
    """the store shards by path columns and nested MC by outer path.  `bench.py --gpus 2 --backend gloo` runs
    the real kernels as two ranks on this box's one GPU: rank g works on global paths [g n, (g + 1) n) (path_offset != 0
    on rank 1) and the statistics records are summed over the ranks.  The same global job on ONE rank must give the same
    reduced result — the mean of all point prices (nested MC) or the price (store)."""
    per_rank = 64 if workload == "nmc" else 50_000
    common = ["--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    if strategy:
        common += ["--nmc-strategy", strategy]
    two = _bench_line(["--gpus", "2", "--backend", "gloo", "--paths", str(per_rank), *common])
    one = _bench_line(["--gpus", "1", "--paths", str(2 * per_rank), *common])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["global_paths"] == one["config"]["global_paths"] == 2 * per_rank
    if workload == "nmc":
        assert one["mean_point_price"] > 0
        assert abs(two["mean_point_price"] - one["mean_point_price"]) <= 1e-12 * one["mean_point_price"]
        # the work counters are the whole job's on both lines: same pools (64 is a multiple of 8), same schedule
        assert two["executed_inner_path_steps_per_pass"] == one["executed_inner_path_steps_per_pass"]
        assert two["live_inner_path_steps_per_pass"] == one["live_inner_path_steps_per_pass"]
    else:
        assert abs(two["price"] - one["price"]) <= 1e-12 * one["price"]
        assert abs(two["std_err"] - one["std_err"]) <= 1e-9 * one["std_err"]


@pytest.mark.gpu
def test_store_workload_line_carries_its_same_run_ceiling():
    """`--workload store` (BASELINE configs[2]'s kernel at a small path count): the HBM roofline object names the bytes,
    and beside it the pure-store pass of the same shape measured in the same process."""
    d = _bench_line(["--workload", "store", "--paths", "4000000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    rl = d["roofline"]
    assert rl["bound"] == "hbm" and rl["unit"] == "GB/s" and rl["peak"] == 8000.0 and d["dtype"] == "f32"
    assert rl["algorithmic_bytes_per_launch"] >= 4_000_000 * 252 * 4 + 4_000_000 * 4
    assert 0.05 < rl["frac"] < 1.05
    c = rl["same_run_ceiling"]
    assert c["kernel_ms"] > 0 and c["GB_per_s"] > 500 and 0.3 < rl["frac_of_same_run_store_ceiling"] < 1.3
    assert abs(d["price"] - 13.2697) < 0.2
