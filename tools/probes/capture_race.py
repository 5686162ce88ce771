"""Does a step plan recorded right after RCCL collectives fail with hipErrorCapturedEvent, and does plan._quiesce_collectives matter?
The A/B that DESIGN.md section 6 says was never run.  One-rank nccl job per arm (child processes; this driver never touches the GPU):
    python tools/probes/capture_race.py [iterations]        -> one line per arm: iterations survived / first error
Arm "off" = DLIP_PLAN_QUIESCE=0, arm "on" = the default.  Every iteration: a few collectives, then IMMEDIATELY a new StepPlan."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(n):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from deeplip_amd import ops, packing
    from deeplip_amd.plan import StepPlan
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    g = torch.Generator().manual_seed(0)
    x = ops.split_pack((torch.randn(8, 64, 512, generator=g)).cuda())
    w = torch.randn(512, 1, 512, generator=g) / 22.0
    ws, sc = packing.split_weights(w.double())
    ws, sc = ws.cuda(), sc.cuda()
    b = torch.zeros(512).cuda()

    def fn(xx):
        y = xx
        for _ in range(4):
            y = ops.conv1d_ntc(y, ws, b, w_scale=sc, x_split=True, out_split=True)
        return y

    t = torch.ones(1 << 16, device="cuda")
    done = 0
    try:
        for i in range(n):
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
        child(int(sys.argv[2]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    for arm, val in (("off", "0"), ("on", "1"), ("off", "0"), ("on", "1")):
        env = dict(os.environ, DLIP_PLAN_QUIESCE=val, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n)], env=env, capture_output=True, text=True, timeout=240)
            out = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
            print(f"quiesce {arm}: rc={p.returncode} {out[-1] if out else 'NO RESULT: ' + p.stderr[-400:]}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"quiesce {arm}: timeout", flush=True)
            break
