"""`python bench.py --gpus N` starts its own N-rank job (deeplip_amd/launch.py): tested here with CPU ranks over gloo
(--dry-launch: stand-in step, the real exchange / timing / one-line protocol).  What only hardware can add is RCCL itself."""
import json
import os
import subprocess
import sys

import pytest

from deeplip_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, timeout=240):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, BENCH, *argv], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_launch_command_and_argv_parsing():
    cmd = launch.launch_command("/x/bench.py", ["--gpus", "8", "--steps", "5"], 8, port=1234)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "1234" and cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "5"]
    assert launch.argv_gpus(["--steps", "3"]) == 1
    assert launch.argv_gpus(["--gpus", "4"]) == 4 and launch.argv_gpus(["--gpus=2", "--x"]) == 2
    assert launch.maybe_self_launch(BENCH, [], 1) is None          # single GPU: the caller carries on itself
    assert 1024 < launch.free_port() < 65536


def test_inside_a_job_nothing_is_launched(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "1")
    assert launch.in_job() and launch.maybe_self_launch(BENCH, ["--gpus", "2"], 2) is None


def test_bench_gpus2_dry_launch_two_ranks_one_json_line():
    r = _run("--gpus", "2", "--dry-launch", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                               # stdout = rank 0's single JSON line, nothing else
    d = json.loads(lines[0])
    assert d["dry_launch"] is True and d["n_gpus"] == 2 and d["value"] is None and d["steps"] == 3 and d["warmup"] == 1
    assert [x["rank"] for x in d["ranks"]] == [0, 1]
    assert all(x["world_size"] == 2 and x["exchange_ok"] for x in d["ranks"])
    assert len({x["pid"] for x in d["ranks"]}) == 2                # two processes ...
    assert len({x["launched_by"] for x in d["ranks"]}) == 1        # ... started by one launcher
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node 2" in r.stderr


def test_a_failing_rank_fails_the_command():
    r = _run("--gpus", "2", "--dry-launch", "--dry-fail-rank", "1", "--steps", "2", "--warmup", "0")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]     # no result line from a failed job


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="on a GPU box the children would run the real bench")
def test_without_a_gpu_the_children_fail_not_the_launcher():
    """On a CPU box `python bench.py --gpus 2` must get as far as the ranks: the only thing missing is the GPU."""
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "must be launched with" not in r.stderr
    assert "torch.distributed.run" in r.stderr                      # the job was started ...
    assert r.stderr.count("needs a ROCm GPU") >= 1                 # ... and it is a RANK that says so (torchrun may stop the other
                                                                   # rank before it gets to print the same)
