#!/usr/bin/env python3
"""Summarise gpurun_out/lds/*: LDS bank-conflict cycles / LDS active cycles per kernel."""
import csv
import sys
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
print("| run | kernel | SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE |\n|---|---|---|")
for d in sorted((ROOT / "gpurun_out" / "lds").glob("*")):
    acc = defaultdict(lambda: defaultdict(float))
    for f in d.rglob("*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if "sdsp_" not in name:
                continue
            import re
            m = re.search(r"::(sdsp_[a-z0-9_]+)", name)
            short = m.group(1) if m else name
            acc[short][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in sorted(acc.items()):
        act = v.get("SQ_LDS_IDX_ACTIVE", 0.0)
        conf = v.get("SQ_LDS_BANK_CONFLICT", 0.0)
        print(f"| {d.name} | `{k}` | {conf:.3g} / {act:.3g} = {100 * conf / act if act else float('nan'):.1f} % |")
