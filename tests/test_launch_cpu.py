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


def test_sigterm_to_the_launcher_ends_the_job():
    """A harness timeout or a preemption sends SIGTERM to the LAUNCHER: the N-rank job it started (own session = own process
    group) must go with it instead of staying behind holding the GPUs; the launcher returns 128 + SIGTERM."""
    import signal
    import time
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    code = ("import sys; sys.path.insert(0, %r); from deeplip_amd import launch; "
            "sys.exit(launch.self_launch(%r, ['90'], 2))" % (ROOT, os.path.join(ROOT, "tests", "launch_sleeper.py")))
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, cwd=ROOT)
    import re
    pids, seen = [], []
    t0 = time.time()
    while len(pids) < 2 and time.time() - t0 < 120:
        line = p.stdout.readline()
        if not line:
            break                                         # the job ended before both ranks reported
        seen.append(line)
        # (anywhere in the line: two ranks write to one pipe, and a line of the launcher's children may arrive glued to another --
        # one run of the suite in ~20 sat here for the sleepers' whole 300 s with one pid read)
        pids += [int(v) for v in re.findall(r"rank-pid (\d+)", line)]
    if len(pids) != 2:
        p.kill()
    assert len(pids) == 2, "the two ranks never started: " + repr(seen[-10:])
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) == 128 + signal.SIGTERM
    for _ in range(50):                                   # the ranks are gone (reaped by their own parent, which is gone too)
        alive = [q for q in pids if os.path.exists(f"/proc/{q}") and open(f"/proc/{q}/stat").read().split()[2] != "Z"]
        if not alive:
            break
        time.sleep(0.2)
    assert not alive, alive


def test_bench_last_leg_watchdog_two_ranks():
    """The scaling run's LAST leg (the DP training epoch every rank takes part in) runs under a watchdog: rehearsed on two CPU ranks --
    the leg completes -> its result is in the line; rank 1 never comes back (rank 0 then waits inside the all-reduce) -> after
    --dp-leg-timeout seconds rank 0's watchdog prints the line without the leg, both ranks end with code 0, the launcher returns 0."""
    r = _run("--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "0", "--dp-leg")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["configs"]["C5_fusion_train_step"] == {"stand_in_allreduce": 3.0, "ranks": 2}
    r = _run("--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "0", "--dp-leg", "--dry-hang-rank", "1", "--dp-leg-timeout", "4")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "did not finish within 4 s" in d["configs"]["C5_fusion_train_step"]["error"]


def test_train_fusion_dry_two_ranks_end_to_end(tmp_path):
    """`python train_fusion.py --mode train --dry --gpus 2`: the trainer launches its own 2-rank job (CPU ranks, gloo) and runs the
    data-parallel protocol of BASELINE config C5 around stand-in arithmetic (train_fusion.py:88-93,241-315): one job name for all
    ranks, replicas broadcast from rank 0 (they are seeded differently on purpose), each rank its own slice of the global batch,
    flat gradient all-reduce, metric reduction, checkpoints written by rank 0 ALONE, and after two epochs of two steps the head
    weights of both ranks bit-identical -- and different from where rank 0 started."""
    import glob
    import torch
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_fusion.py"), "--mode", "train", "--dry", "--gpus", "2", "--set",
                        "train.bs=6", "train.epoch=2", "train.steps_per_epoch=2", "data.n_spk=5", "data.utt_per_spk=4",
                        "train.sgd.init_lr=0.05"],
                       capture_output=True, text=True, timeout=240, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "--nproc-per-node 2" in r.stderr and "DRY (stand-in arithmetic" in r.stdout and "2 rank(s)" in r.stdout
    assert r.stdout.count("Epoch 1 ") == 1 and r.stdout.count("Epoch 2 ") == 1        # rank 0 alone reports, over the GLOBAL batch
    assert "on 2 GPU(s)" in r.stdout
    runs = glob.glob(str(tmp_path / "exp" / "*"))
    assert len(runs) == 1, runs                                                        # one run directory: rank 0's clock
    files = sorted(os.path.basename(f) for f in glob.glob(runs[0] + "/*"))
    assert files == ["dry_rank0.pt", "dry_rank1.pt", "net_1.pth", "net_2.pth"], files
    for e in (1, 2):
        ck = torch.load(os.path.join(runs[0], f"net_{e}.pth"), map_location="cpu")
        assert ck["writer_rank"] == 0 and ck["epoch"] == e
    a, b = (torch.load(os.path.join(runs[0], f"dry_rank{i}.pt")) for i in (0, 1))
    assert all(torch.equal(a[k], b[k]) for k in a) and set(a) == set(b)
    assert all(torch.equal(a[k], ck["state_dict"][k]) for k in a)                       # = rank 0's last checkpoint
    torch.manual_seed(1234)                                                            # rank 0's initial head (train_fusion._init_dry)
    w0 = torch.nn.Linear(1024, 5).weight
    assert not torch.equal(a["fc.weight"], w0.detach())
