#!/usr/bin/env python3
"""Probe (GPU box): do HIP event-record nodes captured inside a step plan give usable timings after a replay?
Prints elapsed times of events recorded (a) eagerly, (b) by graph replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeplip_amd import ops
from deeplip_amd.plan import StepPlan

x = torch.randn(64, 300, 512, device="cuda")
evs = []

def fn(x):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    y = ops.meanstd_pool(x)
    e1.record()
    z = ops.l2_normalize(y)
    e2.record()
    evs.append((e0, e1, e2))
    return z

try:
    plan = StepPlan(fn, x)
    print("plan launches:", plan.launches, "arena MB:", plan.arena.nbytes() / 1e6)
    ref = fn(x)
    torch.cuda.synchronize()
    print("eager events:", evs[-1][0].elapsed_time(evs[-1][1]), evs[-1][1].elapsed_time(evs[-1][2]))
    for i in range(3):
        out = plan.run()
    torch.cuda.synchronize()
    print("plan == eager:", torch.equal(out, ref))
    e0, e1, e2 = evs[1]   # the events recorded during capture
    try:
        print("graph events:", e0.elapsed_time(e1), e1.elapsed_time(e2))
    except Exception as ex:
        print("graph events FAILED:", repr(ex))
except Exception as ex:
    import traceback; traceback.print_exc()
    print("probe FAILED:", repr(ex))
