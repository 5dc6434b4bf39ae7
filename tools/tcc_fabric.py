#!/usr/bin/env python3
"""Summarise gpurun_out/tcc/* (tools/tcc_fabric.sh): per kernel, the L2 <-> fabric interface's request counts, average request latencies
(LEVEL / REQ, in TCC cycles) and stall cycles relative to the TCC's cycle count (all summed over the 16 x 8 TCC instances)."""
import csv
import re
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
print("| run | kernel | read latency (cycles) | write latency | WRREQ stall / cycle | too-many-WRREQ stall | write DRAM-credit stall | read DRAM-credit stall | tag stall | input-buffer stall | busy / cycle | L2 hit rate |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for d in sorted((ROOT / "gpurun_out" / "tcc").glob("*")):
    acc = defaultdict(lambda: defaultdict(float))
    for f in d.rglob("*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            m = re.search(r"(sdsp_(?!hip\b)[a-z0-9_]+)", name)
            if not m:
                continue
            acc[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in sorted(acc.items()):
        g = lambda n: v.get(n, float("nan"))
        cyc = g("TCC_CYCLE_sum")
        r = lambda a, b: f"{a / b:.3g}" if b and b == b and a == a else "n/a"
        # the sets come from different runs of the same steps: ratios across sets use the cycle count of set 0 (same launches)
        print(f"| {d.name} | `{k}` | {r(g('TCC_EA0_RDREQ_LEVEL_sum'), g('TCC_EA0_RDREQ_sum'))} | {r(g('TCC_EA0_WRREQ_LEVEL_sum'), g('TCC_EA0_WRREQ_sum'))} | "
              f"{r(g('TCC_EA0_WRREQ_STALL_sum'), cyc)} | {r(g('TCC_TOO_MANY_EA_WRREQS_STALL_sum'), cyc)} | {r(g('TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum'), cyc)} | "
              f"{r(g('TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum'), cyc)} | {r(g('TCC_TAG_STALL_sum'), cyc)} | {r(g('TCC_IB_STALL_sum'), cyc)} | {r(g('TCC_BUSY_sum'), cyc)} | "
              f"{r(g('TCC_HIT_sum'), g('TCC_HIT_sum') + g('TCC_MISS_sum'))} |")
