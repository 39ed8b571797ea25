"""bench.py as the driver runs it: one JSON line with the contract's fields; `--gpus 2` started as a plain command
launches its two ranks itself (torch.distributed.run) and every rank renders its sample slice with the HIP library.
The box has one GPU, so the two ranks share it and the sum goes through gloo (RRT_BENCH_REHEARSAL=1): a rehearsal of
the multi-process path, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    out = run_bench(["--config", "2", "--spp", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["unit"] == "Mrays/s" and out["value"] > 0
    assert out["scaling"] == "strong" and out["config"]["spp_total"] == 16 and "16 spp" in out["config"]["workload"]
    roof = out["roofline"]
    assert roof["kernel"] == "k_render" and roof["bound"] in ("issue", "hbm")
    assert roof["frac"] is None or 0 < roof["frac"] <= 1.0  # never a fraction above one
    assert 0 < roof["lane_util_node"] <= 1 and roof["rays_per_launch"] > 0
    assert out["frame_check"].startswith("bit-identical to whole-tile items")  # the bench checked the frame it times


def test_two_ranks_launch_themselves_and_share_the_sample_range():
    one = run_bench(["--config", "1", "--spp", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    two = run_bench(["--gpus", "2", "--config", "1", "--spp", "8", "--steps", "1", "--warmup", "0"], {"RRT_BENCH_REHEARSAL": "1"})
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["spp_total"] == 8 and two["config"]["spp_per_gpu"] == 4
    assert "REHEARSAL" in two["data"]
    # the same frame: the two ranks together trace exactly the rays the single process traces
    assert round(two["rays_per_sample"] * 711 * 400 * 8) == round(one["rays_per_sample"] * 711 * 400 * 8)
    weak = run_bench(["--gpus", "2", "--config", "1", "--spp", "4", "--scaling", "weak", "--steps", "1", "--warmup", "0"], {"RRT_BENCH_REHEARSAL": "1"})
    assert weak["scaling"] == "weak" and weak["config"]["spp_total"] == 8 and weak["config"]["spp_per_gpu"] == 4
    assert round(weak["rays_per_sample"] * 711 * 400 * 8) == round(one["rays_per_sample"] * 711 * 400 * 8)


def test_one_rank_under_the_launcher_runs_the_rccl_calls():
    """The driver's multi-GPU runs use the nccl (= RCCL) backend, which this one-GPU box cannot run with two ranks.  A
    group of one rank still initialises RCCL, reduces the device buffer the kernel wrote and all-reduces the timings: the
    calls of the N > 1 path, with the library and PyTorch in one process."""
    env = dict(os.environ)
    env.update({"RRT_BENCH_FORCE_GROUP": "1", "MASTER_ADDR": "127.0.0.1"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "1", "--spp", "8", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["process_group"] == "nccl" and out["n_gpus"] == 1 and out["value"] > 0
    one = run_bench(["--config", "1", "--spp", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert round(out["rays_per_sample"] * 711 * 400 * 8) == round(one["rays_per_sample"] * 711 * 400 * 8)
