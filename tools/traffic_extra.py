#!/usr/bin/env python3
"""Summarise gpurun_out/traffic/*: HBM bytes per bench step = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE
counts half of the streamed read bytes; WRITE_SIZE is exact; both in KiB) against the algorithmic bytes of the
step.  Every sdsp kernel of a workload is launched once per step, so the per-step traffic is the sum over the
kernels of their mean per launch."""
import csv
import json
import re
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
print("| run | kernels (one launch each per step) | HBM bytes per step (PMC) | algorithmic bytes per step | ratio |\n|---|---|---|---|---|")
for d in sorted((ROOT / "gpurun_out" / "traffic").glob("*")):
    kib = {}
    kernels = set()
    alg = None
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        alg = json.loads((d / c / "bench.json").read_text().strip().splitlines()[-1])["roofline"]["algorithmic_bytes_per_launch"]
        per_kernel = defaultdict(list)
        for f in (d / c).rglob("*counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                m = re.search(r"::(sdsp_[a-z0-9_]+)", row.get("Kernel_Name", ""))
                if m and row["Counter_Name"] == c:
                    per_kernel[m.group(1)].append(float(row["Counter_Value"]))
        kib[c] = sum(sum(v) / len(v) for v in per_kernel.values())
        kernels |= set(per_kernel)
    hbm = (2 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024
    print(f"| {d.name} | {', '.join('`' + k + '`' for k in sorted(kernels))} | {hbm:.4g} | {alg:.4g} | {hbm / alg:.3f} |")
