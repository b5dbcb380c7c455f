"""world_size-2 gloo test of the multi-rank path (CPU)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_gloo_protocol():
    from neklab_amd import build
    build.build_library()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
