"""bench.py contract on the GPU box: `python bench.py --gpus N` starts N ranks ITSELF (no launcher), the JSON line carries
n_gpus = N, the headline plus the train / render sub-records, each with a roofline.  A 1-GPU box cannot run two RCCL
ranks (RCCL refuses two ranks per device), so the plumbing test puts both ranks on cuda:0 over gloo
(MIPNERF_BENCH_SHARE_GPU / MIPNERF_BENCH_BACKEND: test-only knobs)."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                          env=env, cwd=REPO)


def test_bench_refuses_world_size_mismatch():
    """A launcher environment that disagrees with --gpus is an error, not a silent 1-rank run."""
    out = _run(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=2 but --gpus 1" in (out.stderr + out.stdout)


@pytest.mark.gpu
def test_bench_self_launches_two_ranks():
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rays", "512", "--samples", "32", "--sustain-seconds", "0"],
               {"MIPNERF_BENCH_SHARE_GPU": "1", "MIPNERF_BENCH_BACKEND": "gloo"})
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["config"]["mode"] == "inference" and line["roofline"]["frac"] > 0
    assert line["per_gpu"] * 2 == pytest.approx(line["value"], rel=1e-6)
    for k in ("train", "render"):
        assert line[k]["n_gpus"] == 2 and line[k]["value"] > 0 and line[k]["roofline"]["frac"] > 0, k
    assert line["render"]["scaling"] == "strong" and line["train"]["scaling"] == "weak"
    assert line["cpu_baseline"] is None          # N = 1 only


@pytest.mark.gpu
def test_bench_single_gpu_line():
    out = _run(["--steps", "3", "--warmup", "1", "--rays", "512", "--samples", "32", "--sustain-seconds", "0.2"], {})
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["metric"] == "ray-samples/sec" and line["dtype"] == "bf16"
    assert line["cpu_baseline"]["kind"] in ("reference", "port") and line["cpu_baseline"]["value"] > 0
    assert line["sustained"]["value"] > 0 and "train" in line and "render" in line
