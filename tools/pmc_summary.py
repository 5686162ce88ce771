#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one directory per pass) per kernel.

    python tools/pmc_summary.py OUT.json STEPS dir1 [dir2 ...]

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section), so `hbm_read_bytes` = 2 * FETCH_SIZE * 1024 and
`hbm_write_bytes` = WRITE_SIZE * 1024.  Values are per launch (total / launches)."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    """Kernel names as bench.py reports them: <BM,BN> plus ',dual' / ',pool' for the LDS-DMA kernel's variants
    (template arguments <BM, BN, WAVES_M, WAVES_N, EPI, NSTAGE, OCC, DUAL, VAR>)."""
    m = re.search(r"conv_igemm_f16x3_dma_kernel<(\d+), (\d+), \d+, \d+, (\d+), \d+, \d+, (true|false)", name)
    if m:
        tag = ",dual" if m.group(4) == "true" else (",pool" if m.group(3) == "2" else "")
        return f"conv_igemm_f16x3_dma_kernel<{m.group(1)},{m.group(2)}{tag}>"
    m = re.search(r"(conv_\w+?_kernel<\d+, \d+)", name)
    if m:
        return m.group(1).replace(" ", "") + ">"
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:40]


def main():
    out, steps, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[(k, f)].add(r["Dispatch_Id"])
    res = {}
    for k, c in agg.items():
        n = max(len(v) for (kk, f), v in launches.items() if kk == k)
        e = {"launches": n, "launches_per_step": n / max(steps, 1)}
        if "FETCH_SIZE" in c:
            e["hbm_read_bytes_per_launch"] = 2 * c["FETCH_SIZE"] * 1024 / n
        if "WRITE_SIZE" in c:
            e["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024 / n
        if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            gui = c["GRBM_GUI_ACTIVE"] / 8
            e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024)
            if "SQ_BUSY_CU_CYCLES" in c:
                e["cu_busy_frac"] = c["SQ_BUSY_CU_CYCLES"] / (gui * 256)
        for name in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"):
            if name in c:
                e[name + "_per_launch"] = c[name] / n
        res[k] = e
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
