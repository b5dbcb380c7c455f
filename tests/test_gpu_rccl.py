"""RCCL code paths on one GPU: a single-rank communicator must reproduce the communicator-free results."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run(force):
    env = dict(os.environ)
    env.pop("NLG_FORCE_COMM", None)
    script = os.path.join(ROOT, "tests", "rccl_single_rank.py")
    if force:
        env["NLG_FORCE_COMM"] = "1"
        cmd = [sys.executable, script]
    else:
        cmd = [sys.executable, "-c",
               "import os,runpy;os.environ['NLG_FORCE_COMM']='1';"
               "import neklab_amd.host as h;h.Context.comm_init=lambda *a,**k:None;"
               "runpy.run_path(%r, run_name='__main__')" % script]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0]
    return [float(x) for x in line.split()[1:]]


def test_single_rank_communicator_matches():
    a = run(True)      # RCCL all-reduce / all-gather in the loop
    b = run(False)     # no communicator
    for x, y in zip(a, b):
        assert abs(x - y) <= 1e-14 * max(abs(y), 1.0)
