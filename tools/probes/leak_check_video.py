"""train_video.main() for 12 and for 60 iterations (B = 4 x 9 frames), eager and recorded: torch's peak reserved / allocated device memory and the
host RSS must not depend on the number of iterations.   python tools/probes/leak_check_video.py"""
import os
import resource
import subprocess
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
if len(sys.argv) > 1:          # child: one run
    import tempfile
    sys.path.insert(0, ROOT)
    os.chdir(tempfile.mkdtemp())
    import torch
    import train_video
    extra = [] if sys.argv[2] == "graph" else ["--eager-step"]
    train_video.main(["--steps", sys.argv[1], "--batch-size", "4", "--frames", "9", "--display", "1000"] + extra)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    print(f"MEM {torch.cuda.max_memory_allocated() >> 20} {torch.cuda.max_memory_reserved() >> 20} {(total - free) >> 20} "
          f"{resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10}")
    sys.exit(0)
bad = False
for mode in ("eager", "graph"):
    res = {}
    for steps in (12, 60):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), str(steps), mode], capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("MEM ")]
        res[steps] = [int(v) for v in line[-1].split()[1:]] if line else None
    print(f"{mode}: 12 iterations {res[12]}  60 iterations {res[60]}   (peak allocated, peak reserved, device used, host RSS; MiB)", flush=True)
    if res[12] and res[60]:
        bad |= res[60][1] - res[12][1] > 128 or res[60][2] - res[12][2] > 128 or res[60][3] - res[12][3] > 256
    else:
        bad = True
sys.exit(1 if bad else 0)
