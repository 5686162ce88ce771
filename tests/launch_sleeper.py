"""A rank that does nothing for a long time (tests/test_launch_cpu.py: the launcher is told to stop while its job runs)."""
import os
import sys
import time

print("rank-pid", os.getpid(), flush=True)
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0)
