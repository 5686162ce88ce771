#!/usr/bin/env python3
"""Board power and shader clock while ONE kernel of the step runs back to back (development probe): is the chip at its power
limit in that loop?  rocm-smi is polled from a thread while the main thread keeps the queue full.
    python3 tools/probes/power_clock.py stem | l3 | l1 | idle"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeplip_amd import ops, packing

what = sys.argv[1] if len(sys.argv) > 1 else "stem"
samples, stop = [], False

def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out.strip().splitlines())
        except Exception as e:   # noqa
            samples.append([repr(e)])
        time.sleep(0.4)

torch.manual_seed(0)
B = 64
if what == "stem":
    x = torch.randn(B, 29, 88, 88, device="cuda")
    img, sc = packing.split_stem_weights(torch.randn(64, 1, 5, 7, 7, dtype=torch.float64) * 0.05)
    img, sc = img.cuda(), sc.cuda(); b = torch.randn(64, device="cuda"); sl = torch.rand(64, device="cuda")
    run = lambda: ops.stem3d_pool(x, img, b, sl, sc)
    fl = 2.0 * B * 29 * 44 * 44 * 64 * 245
elif what in ("l3", "l1", "l2"):
    H, C = {"l3": (6, 256), "l1": (22, 64), "l2": (11, 128)}[what]
    N = B * 29
    x = ops.split_pack(torch.randn(N, H, H, C, device="cuda"))
    wsp, wsc = packing.split_weights(torch.randn(C, 3, 3, C, dtype=torch.float64) * 0.03)
    wsp, wsc = wsp.cuda(), wsc.cuda(); b = torch.randn(C, device="cuda"); sl = torch.rand(C, device="cuda")
    y = torch.empty(N, H, H, C, device="cuda")
    run = lambda: ops.conv_nhwc(x, wsp, b, pad=(1, 1), slope=sl, w_scale=wsc, x_split=True, out_split=True, out=y)
    fl = 2.0 * N * H * H * C * 9 * C
else:
    run, fl = None, 0.0
t = threading.Thread(target=poll); t.start()
t0 = time.time(); n = 0
if run is None:
    time.sleep(4)
else:
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(50):
            run()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    print(f"{what}: {n} launches, {e0.elapsed_time(e1) * 1e3 / n:.1f} us each, {fl * n / e0.elapsed_time(e1) / 1e9:.0f} TFLOP/s")
stop = True; t.join()
hdr = samples[0][0] if samples and samples[0] else ""
print(hdr)
for s in samples:
    for line in s[1:]:
        print(line)
