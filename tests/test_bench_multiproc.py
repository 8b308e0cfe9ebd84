"""CPU, world_size 2, gloo: the multi-process launch path of bench.py (one process per rank via torch.distributed.run,
barrier + MAX-over-ranks timing, rank 0 prints exactly one JSON line, whole-job aggregate)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_bench_harness_two_ranks_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--selftest-cpu"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 5 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    # rank 1 sleeps 4 ms per step, rank 0 only 2 ms: the MAX over ranks must be what is reported
    assert rec["ms_per_step"] >= 3.9
    assert abs(rec["value"] - 2 * 32 * 1000 * 5 / (rec["ms_per_step"] * 5 / 1e3)) / rec["value"] < 1e-6


def test_bench_harness_single_process():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--selftest-cpu"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["ms_per_step"] >= 1.9


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` outside torchrun (VERDICT r2 missing #2; reference: pl.Trainer(devices=N), src/executor.py:137-139):
    the file starts the 2 ranks itself and relays rank 0's line -- n_gpus must be 2, never a silent single rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--selftest-cpu"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and rec["ms_per_step"] >= 3.9


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """WORLD_SIZE=2 from the launcher but --gpus 1 (or the flag forgotten): non-zero exit, no JSON line."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29579", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--selftest-cpu"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "--gpus 1 but the launcher started WORLD_SIZE=2" in out.stderr


def test_bench_refuses_more_gpus_than_devices():
    """The real (GPU) mode on a box with fewer devices than --gpus: refuses before anything is launched (here: 0 devices)."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has >= 2 devices")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "HIP device(s) visible" in out.stderr


def test_traffic_lookup_tells_the_chain_instances_apart():
    """bench.pick_traffic: the PMC table is keyed by demangled kernel names; the merged launch (conv-in stage inside), the narrower chained launch and the
    stand-alone conv-in chain are instances of ONE template and must not be mistaken for each other."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t = {"cfm_rowchain_kernel<BF16, 256, 2048, 1, true, true, 3, false, false, false, true, false, false, true>": 97.0e6,
         "cfm_rowchain_kernel<BF16, 256, 2048, 1, true, true, 3, false, false, false, true, false, false, false>": 60.0e6,
         "cfm_rowchain_kernel<BF16, 256, 64, 1, false, false, 1, true, false, false, false, false, false, false>": 24.0e6,
         "cfm_rowchain_kernel<BF16, 256, 2048, 0, false, true, 3, false, false, false, false, false, false, false>": 42.0e6,
         "cfm_attn2_kernel<BF16, 1, false, false>": 20.0e6}
    assert bench.pick_traffic(t, "chain_convin_dwfinal_macaron_bf16_d256", 256) == 97.0e6
    assert bench.pick_traffic(t, "chain_dwfinal_macaron_bf16_d256", 256) == 60.0e6
    assert bench.pick_traffic(t, "chain_convin_bf16_d256", 256) == 24.0e6
    assert bench.pick_traffic(t, "chain_macaron_bf16_d256", 256) == 42.0e6
    assert bench.pick_traffic(t, "attn2_rel_bf16", 256) == 20.0e6
    assert bench.pick_traffic(t, "chain_convin_dwfinal_macaron_bf16_d256", 512) is None
