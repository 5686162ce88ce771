"""rocprofv3 --pmc counter_collection CSVs -> per-kernel figures (measurement plumbing; no GPU call here).

Used by tools/pmc_summary.py (the committed collections under profiles/) and by bench.py, which runs two short PMC passes of its
own command -- as child processes, before this process touches the GPU -- so that `roofline.traffic` of a bench line is measured in the
very run that prints it.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section), so hbm_read_bytes = 2 * FETCH_SIZE * 1024 and hbm_write_bytes = WRITE_SIZE * 1024, per launch."""
from __future__ import annotations

import collections
import csv
import glob
import re
from typing import Dict, Sequence


def short(name: str) -> str:
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


def summarise(dirs: Sequence[str], steps: int) -> Dict[str, dict]:
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
    return res
