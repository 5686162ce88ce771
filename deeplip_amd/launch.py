"""One command starts every GPU: the self-launcher of bench.py and the three train_*.py entry points.

The reference starts all its GPUs from a single ``python train_fusion.py`` (train_fusion.py:88-93, train_audio.py:80-83,
train_video.py:206-207: ``gpus_id`` -> nn.DataParallel threads).  Here data parallelism is one PROCESS per GPU over RCCL, so a
script asked for N > 1 GPUs that finds itself outside a torch.distributed job becomes the launcher of one:

    parent (this process, never touches the GPU)
      └─ child: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>
                 <script> <argv...>          (a fresh subprocess -- never an exec: a process must not replace itself once
                                              anything may have initialised HIP, and the parent stays to relay / reap)
            └─ N ranks, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment

stdout of the job is relayed line by line (rank 0's single JSON line passes through unchanged), stderr is inherited, and the
parent exits with the child's return code: a rank that fails makes torch.distributed.run tear the job down and return non-zero,
and so does the parent.  Nothing here imports torch: the decision is taken before any ``torch.cuda`` call can happen.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
from typing import List, Optional, Sequence


def in_job() -> bool:
    """True inside a torch.distributed.run job (the launcher exported the rendezvous variables)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(script: str, argv: Sequence[str], nproc: int, port: Optional[int] = None) -> List[str]:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *argv]


class _Terminated(BaseException):
    def __init__(self, signum):
        super().__init__(signum)
        self.signum = signum


def _kill_group(p, sig) -> None:
    try:
        os.killpg(p.pid, sig)          # start_new_session: the child's pid is its group id
    except (ProcessLookupError, PermissionError):
        pass


def self_launch(script: str, argv: Sequence[str], nproc: int, relay=None, env=None) -> int:
    """Run ``script argv`` as an ``nproc``-rank job and return its exit code.  ``relay(line)`` receives every stdout line
    of the job (default: print it)."""
    cmd = launch_command(os.path.abspath(script), list(argv), nproc)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or nproc) // nproc)))
    e["DLIP_LAUNCHED_BY"] = str(os.getpid())
    print(f"[launch] {nproc} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    def _stop(signum, _frame):
        raise _Terminated(signum)

    # Handlers FIRST, the job second: the job runs in its own session and no longer receives the terminal's SIGINT / SIGHUP, so a
    # signal that found the launcher between Popen and signal.signal() would kill it by the default action and orphan all N ranks.
    old = {}
    main = threading.current_thread() is threading.main_thread()    # signal.signal / pthread_sigmask of this thread: main thread only
    if main:
        for sig in (signal.SIGTERM, signal.SIGHUP):
            old[sig] = signal.signal(sig, _stop)
    p = None
    try:
        # The job gets its own session (= process group): the launcher can then end exactly the processes it started -- the
        # torch.distributed.run child AND its N ranks -- by group id, also when it is itself told to stop.
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, bufsize=1, env=e, start_new_session=True)
        for line in p.stdout:
            if relay is None:
                sys.stdout.write(line)
                sys.stdout.flush()
            else:
                relay(line)
        rc = p.wait()
    except BaseException as ex:
        # a harness timeout / scheduler preemption (SIGTERM, SIGHUP), Ctrl-C, or an error in relay(): the job must not outlive
        # its launcher holding the GPUs.  The exact process group we started, nothing by pattern.  The teardown itself runs with
        # the stop signals BLOCKED: a second SIGTERM during the grace wait used to raise inside this handler and skip the
        # escalation to SIGKILL.
        prev_mask = None
        if main:     # (the caller's mask comes back afterwards as it was: a harness that keeps SIGINT blocked keeps it blocked)
            prev_mask = signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM, signal.SIGHUP, signal.SIGINT})
        try:
            if p is not None:
                _kill_group(p, signal.SIGTERM)
                try:
                    p.wait(timeout=30)
                except subprocess.TimeoutExpired:
                    _kill_group(p, signal.SIGKILL)
                    p.wait()
        finally:
            if main:
                for sig, h in old.items():          # (restored before unblocking: a pending stop signal then takes its old course)
                    signal.signal(sig, h)
                old = {}
                signal.pthread_sigmask(signal.SIG_SETMASK, prev_mask)
        if isinstance(ex, _Terminated):
            print(f"[launch] signal {ex.signum}: job stopped", file=sys.stderr, flush=True)
            return 128 + ex.signum
        raise
    finally:
        for sig, h in old.items():
            signal.signal(sig, h)
    if rc != 0:
        print(f"[launch] job failed: exit code {rc}", file=sys.stderr, flush=True)
    return rc


def maybe_self_launch(script: str, argv: Sequence[str], gpus: int) -> Optional[int]:
    """If ``gpus`` > 1 and this process is not a rank of a job yet: run the job and return its exit code (the caller
    exits with it).  Otherwise None -- the caller is a rank (or a single-GPU run) and carries on."""
    if gpus <= 1 or in_job():
        return None
    return self_launch(script, argv, gpus)


def argv_gpus(argv: Sequence[str], flag: str = "--gpus", default: int = 1) -> int:
    """``--gpus N`` / ``--gpus=N`` read from a raw argument vector (before argparse, before torch)."""
    n = default
    for i, a in enumerate(argv):
        if a == flag and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith(flag + "="):
            n = int(a.split("=", 1)[1])
    return n
