"""`python bench.py --gpus 2` with NO external launcher must start its two ranks itself, split the problem (strong) or
stack it (weak), and print one JSON line with n_gpus = 2.  Rehearsed on the one GPU of the test box through the library's
shared-memory validation transport (RCCL refuses two ranks on one device); on a multi-GPU node the same command without
`--transport shm` runs over RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--nel", "6,6,4", "--kdim", "8",
           "--no-cpu", "--no-units"] + list(extra)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_strong_scaling_uneven_partition():
    """Three ranks on the 6 x 6 x 4 box: recursive coordinate bisection gives uneven sub-boxes (2 x 6 x 4 and two of 4 x 3 x 4), with
    neighbours across faces and an edge, as the bisection of BASELINE.json's 25 x 20 x 20 box over 8 GPUs does (12|13 x 10 x 10)."""
    one = run_bench("--gpus", "1", "--scaling", "strong")
    three = run_bench("--gpus", "3", "--transport", "shm", "--scaling", "strong")
    assert three["n_gpus"] == 3 and three["config"]["global_elements"] == one["config"]["global_elements"] == 144
    assert three["config"]["partition"] == "rcb" and sorted(three["config"]["partition_sizes"]) == [48, 48, 48]
    assert abs(three["config"]["dt"] - one["config"]["dt"]) < 1e-12 * one["config"]["dt"]
    assert abs(three["config"]["pressure_iters_per_time_step"] - one["config"]["pressure_iters_per_time_step"]) <= 0.3 * one["config"]["pressure_iters_per_time_step"] + 2


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_starts_its_own_ranks(scaling):
    one = run_bench("--gpus", "1", "--scaling", scaling)
    two = run_bench("--gpus", "2", "--transport", "shm", "--scaling", scaling)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["scaling"] == two["scaling"] == scaling
    assert two["value"] > 0 and abs(two["value"] * two["ms_per_step"] * 1e-3 - 1.0) < 1e-9      # global matvecs / s, no x N
    c1, c2 = one["config"], two["config"]
    if scaling == "strong":      # the same global problem: same size, same time step, (nearly) the same iteration counts
        assert c2["global_elements"] == c1["global_elements"] == 144 and c2["elements_per_gpu"] == 72
        assert abs(c2["dt"] - c1["dt"]) < 1e-12 * c1["dt"]
        assert abs(c2["pressure_iters_per_time_step"] - c1["pressure_iters_per_time_step"]) <= 0.25 * c1["pressure_iters_per_time_step"] + 2
        assert abs(c2["helmholtz_iters_per_time_step"] - c1["helmholtz_iters_per_time_step"]) <= 1.0
    else:
        assert c2["global_elements"] == 2 * c1["global_elements"] and c2["elements_per_gpu"] == 144
