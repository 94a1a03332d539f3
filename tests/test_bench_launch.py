"""bench.py must be startable exactly as the driver starts it: `python bench.py --gpus N ...` with no launcher
around it for any N, and under torch.distributed.run.  --dry-run stops before the first GPU call, so these run on
a CPU-only machine; what they prove is that N ranks come up, meet, and rank 0 alone prints one JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _json_lines(out: str):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_launches_its_own_ranks(gpus):
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=_clean_env())
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1
    assert lines[0]["n_gpus"] == gpus and lines[0]["ranks"] == list(range(gpus))


def test_bench_under_the_drivers_launcher():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_clean_env())
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks"] == [0, 1]


def test_one_rank_can_be_made_to_use_the_process_group():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run", "--force-dist"], capture_output=True, text=True, timeout=300,
                       env=_clean_env())
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["ranks"] == [0]


def test_a_failing_rank_fails_the_job_without_a_hang():
    # rank 1 dies before the rendezvous: the launcher must end rank 0 and the job must exit non-zero, soon
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--dry-run-fail-rank", "1", "--rendezvous-seconds", "60"],
                       capture_output=True, text=True, timeout=240, env=_clean_env())
    assert p.returncode != 0
    assert "failed" in p.stderr and not _json_lines(p.stdout)


def test_world_size_mismatch_is_refused():
    env = _clean_env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_traffic_figure_is_tied_to_the_kernel_sources():
    sys.path.insert(0, ROOT)
    import bench
    sha = bench.kernels_sha256()
    assert len(sha) == 64
    t, src = bench.committed_traffic("no-such-workload", 1, True)
    assert t is None and "no committed" in src
    recs = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for name, rec in recs.items():
        t, src = bench.committed_traffic(name, rec["frames"], True)
        if rec.get("kernels_sha256") == sha:
            assert t == rec["traffic_bytes"] and "not this run" in src
        else:
            assert t is None and "stale" in src
        assert bench.committed_traffic(name, rec["frames"] + 1, True)[0] is None
        assert bench.committed_traffic(name, rec["frames"], False)[0] is None
