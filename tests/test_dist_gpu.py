"""The sharded planner end to end with world > 1: `world` child processes share cuda:0,
exchange the packed (n_r, 1 + A) score/action buffer over gloo (staged through the host), and each must reproduce the
world-1 plan step bit for bit -- HipPlanner(group=...) -> shard -> gather -> replicated re-score -> select.
Children are separate interpreters started with subprocess (no fork of a GPU-initialised parent state is used, nothing is
re-exec'd); a child exits non-zero on any mismatch."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# (the world-2 cases share one pair of child processes: interpreter start-up dominates this file on a cold box)
@pytest.mark.parametrize("world,case", [(2, "c2,c3,c4,odd,attach"), (3, "c2"), (1, "rccl1")])
def test_sharded_planner_equals_single_gpu(world, case):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world), str(port), case],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=600)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} exited {p.returncode}:\n{out[-3000:]}"
        assert out.count("sharded == single") == len(case.split(",")), out[-3000:]
