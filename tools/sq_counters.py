#!/usr/bin/env python3
"""Mean per launch of each SQ counter per sdsp kernel, from gpurun_out/sq/*."""
import csv
import re
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
for d in sorted((ROOT / "gpurun_out" / "sq").glob("*")):
    acc = defaultdict(lambda: defaultdict(list))
    for f in d.rglob("*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            m = re.search(r"::(sdsp_[a-z0-9_]+)", row.get("Kernel_Name", ""))
            if m:
                acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        print(f"{d.name:8s} {k}: " + ", ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
