"""`python bench.py --gpus N` as the driver invokes it (no launcher, WORLD_SIZE unset) must start its N ranks itself, relay
rank 0's JSON line and fail when a rank fails (VERDICT r2 missing 3).  Checked here without a GPU through --dry-run: the
ranks meet over gloo instead of running the planner."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"] + extra, env=env, capture_output=True, text=True,
                          timeout=300)


def test_bench_spawns_its_ranks_and_relays_one_json_line():
    r = _run(["--gpus", "2", "--steps", "7", "--warmup", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 7 and d["warmup"] == 3 and d["metric"].startswith("MPC plan-steps/sec")


def test_bench_fails_when_a_rank_fails():
    r = _run(["--gpus", "2"], {"M3PC_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0


def test_bench_single_process_needs_no_launcher():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_multi_gpu_default_runs_the_collective_legs():
    """N > 1 without flags: behind the headline, the candidate-sharded legs run through the all-gather of the data path (gloo
    here) and the line says how many ranks that collective saw."""
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["rccl_ranks_seen"] == 2 and "c4" in d and "c2_candidates_strong" in d
    assert "c4_pipelined" in d and "c4_full" in d and d["collective_legs_ok"] is True  # 


def test_a_hung_collective_leg_costs_its_entry_not_the_headline():
    """A leg that never returns (a collective that hangs): the watchdog prints the headline line with the leg marked as timed
    out and every rank exits with code 0."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--collective-timeout", "4"], {"M3PC_BENCH_HANG_LEG": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert time.time() - t0 < 120
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["c4"] == {"error": "timeout"} and d["n_gpus"] == 2 and d["steps"] == 5 and d["collective_legs_ok"] is False


def test_a_rank_that_fails_in_a_leg_is_reported_at_once_with_its_message():
    """one rank raises while setting up a leg (before any collective); the other ranks, which would sit in the
    all-gather until the watchdog's limit, see its message in the rendezvous store and leave within seconds; the headline
    line carries the message and "collective_legs_ok": false."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--collective-timeout", "200"], {"M3PC_BENCH_FAIL_LEG_RANK": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert time.time() - t0 < 90
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["collective_legs_ok"] is False and "injected leg failure" in d["c4"]["error"] and "rank 1" in d["c4"]["error"]


def test_no_collective_legs_flag():
    r = _run(["--gpus", "2", "--no-collective-legs"])
    assert r.returncode == 0
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert "c4" not in d
