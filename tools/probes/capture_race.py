"""Why a step plan recorded inside a torch.distributed (RCCL) job sometimes killed the job with "Process group watchdog thread
terminated with exception: HIP error: operation not permitted on an event last recorded in a capturing stream" -- and the check that
deeplip_amd.plan.StepPlan no longer can.  One-rank nccl job per arm (child processes; this driver never touches the GPU):
    python tools/probes/capture_race.py [--long] [iterations]
Every iteration: a few collectives, then IMMEDIATELY a new StepPlan.  `--long`: the recorded function SLEEPS 0.3 s inside the
capture, so that the process group's watchdog (one poll of every un-reaped work per ~100 ms) certainly polls while it is open, and
the warm-up pass in front of it gives the watchdog no time to reap.  Arms: collectives issued on the default stream / with the
stream handed to StepPlan current (what bench.py does: `with torch.cuda.stream(run_stream): exchange(...)`, then
StepPlan(stream=run_stream)); `raw` = the recording forced onto that very stream (StepPlan's behaviour until round 4: aborts at the
first plan -- profiles/r4/capture_race_probe.txt)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(n, long_capture=False, same_stream=False, raw=False):
    import time
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from deeplip_amd import ops, packing
    from deeplip_amd.plan import StepPlan
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    if raw:   # the pre-round-4 behaviour: record on the caller's stream
        import deeplip_amd.plan as _p
        _orig = torch.cuda.Stream
        class _Same:
            def __init__(self): self.s = None
        holder = _Same()
        def _stream(*a, **k):
            return holder.s if holder.s is not None else _orig(*a, **k)
        _p.torch.cuda.Stream = _stream
    g = torch.Generator().manual_seed(0)
    x = ops.split_pack((torch.randn(8, 64, 512, generator=g)).cuda())
    w = torch.randn(512, 1, 512, generator=g) / 22.0
    ws, sc = packing.split_weights(w.double())
    ws, sc = ws.cuda(), sc.cuda()
    b = torch.zeros(512).cuda()

    calls = [0]

    def fn(xx):
        calls[0] += 1                                # StepPlan calls fn twice: the warm-up pass, then the RECORDED pass
        y = xx
        for i in range(4):
            y = ops.conv1d_ntc(y, ws, b, w_scale=sc, x_split=True, out_split=True)
            if long_capture and i == 1 and calls[0] % 2 == 0:
                time.sleep(0.3)                      # only inside the capture: it stays open across three watchdog polls, and the
        return y                                     # warm-up pass gives the watchdog no time to reap the collectives before it

    t = torch.ones(1 << 16, device="cuda")
    run_stream = torch.cuda.Stream()
    if raw:
        holder.s = run_stream
    done = 0
    try:
        for i in range(n):
            if same_stream:
                with torch.cuda.stream(run_stream):
                    for _ in range(3):
                        dist.all_reduce(t)
                    out = torch.empty_like(t)
                    dist.all_gather_into_tensor(out, t)
                    dist.barrier()
                    plan = StepPlan(fn, x, stream=run_stream)
                    plan.run()
                    dist.all_gather_into_tensor(out, t)
            else:
                for _ in range(3):
                    dist.all_reduce(t)
                outs = [torch.empty_like(t)]
                dist.all_gather(outs, t)
                plan = StepPlan(fn, x)          # capture starts right behind the collectives
                plan.run()
            torch.cuda.synchronize()
            del plan
            done += 1
    except Exception as e:  # noqa: BLE001 -- the probe reports whatever ends the loop
        print(f"RESULT survived={done} of {n} error={type(e).__name__}: {str(e)[:300]}", flush=True)
        os._exit(0)
    print(f"RESULT survived={done} of {n} error=none", flush=True)
    os._exit(0)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), long_capture="long" in sys.argv[3:], same_stream="same" in sys.argv[3:], raw="raw" in sys.argv[3:])
    long_form = "--long" in sys.argv
    nums = [a for a in sys.argv[1:] if a.isdigit()]
    n = int(nums[0]) if nums else (12 if long_form else 60)
    arms = [("collectives on the default stream", []), ("collectives on the stream handed to StepPlan", ["same"])]
    if long_form:
        arms = [("collectives on the default stream", ["long"]), ("collectives on the stream handed to StepPlan", ["long", "same"]),
                ("... and the recording forced onto that stream (old behaviour)", ["long", "same", "raw"])]
    for arm, extra in arms:
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n)] + extra, env=env, capture_output=True, text=True, timeout=240)
            out = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
            why = [ln for ln in p.stderr.splitlines() if "terminated with exception" in ln or "HIP error" in ln]
            print(f"{arm}: rc={p.returncode} {out[-1] if out else 'NO RESULT; ' + (why[0][:400] if why else p.stderr[-600:] or p.stdout[-300:])}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"{arm}: timeout", flush=True)
            break
