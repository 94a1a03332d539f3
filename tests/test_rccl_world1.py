"""The multi-GPU code path of bench.py on the one GPU a test box has: `--force-dist` makes a single rank initialise
the RCCL process group, so the barrier, the counter all-reduce, the size all-gather and the max-over-ranks timing
all run on a GPU.  The verdict counts must equal those of the same run without a process group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--frames", "262144"] + extra,
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return lines[0]


@pytest.mark.parametrize("workload", ["udp1500", "trace"])
def test_one_rank_over_rccl_equals_no_process_group(workload):
    extra = ["--workload", workload] + (["--total-frames", str(3 * 262144)] if workload == "trace" else [])
    plain = _run(extra)
    rccl = _run(extra + ["--force-dist", "--backend", "nccl"])
    assert rccl["n_gpus"] == 1 and rccl["verdict_counts"] == plain["verdict_counts"]
    assert sum(rccl["verdict_counts"].values()) == (3 * 262144 if workload == "trace" else 262144)
    if workload == "udp1500":
        assert rccl["parity_check"] is True and rccl["config"]["shard_sizes"] == [262144]
