#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one directory per pass) per kernel.

    python tools/pmc_summary.py OUT.json STEPS dir1 [dir2 ...]

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section), so `hbm_read_bytes` = 2 * FETCH_SIZE * 1024 and
`hbm_write_bytes` = WRITE_SIZE * 1024.  Values are per launch (total / launches)."""
import json
import re
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeplip_amd import pmc      # the parsing lives in the package: bench.py uses it for its own in-run passes


def main():
    out, steps, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    res = pmc.summarise(dirs, steps)
    # stamp: which kernel build these counters belong to (bench.py ignores the file on a mismatch)
    import hashlib, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from deeplip_amd import build
    abi = int(re.search(r"#define DLIP_ABI_VERSION (\d+)", open(os.path.join(root, "include", "deeplip_hip.h")).read()).group(1))
    import socket, time
    # ... and WHEN / WHERE they were collected: tools/collect_profiles.sh runs these passes in the same gpurun call -- the same box,
    # minutes apart -- as the bench line that quotes them, and the line says so (roofline.traffic_source)
    res["_meta"] = {"kernel_sha": build.dominant_kernel_sha(), "kernel_sources": list(build.DOMINANT_KERNEL_SOURCES), "abi": abi, "steps": steps,
                    "collected_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "box": build.box_id(),
                    "library_sha": build.library_sha()[:16]}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
