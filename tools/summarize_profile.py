#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile.sh on the GPU box) into tracked files:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (kernel names trimmed)
  profiles/<tag>_summary.md         bench line, per-kernel durations, PMC HBM traffic with the gfx950
                                    FETCH_SIZE correction (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2
                                    of a wide coalesced stream; WRITE_SIZE is exact; both in KiB)
  profiles/traffic.json             HBM bytes per launch per kernel, read by bench.py ("traffic")
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
src = ROOT / "gpurun_out" / f"prof_{tag}"
out = ROOT / "profiles"
out.mkdir(exist_ok=True)


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("sdsp_hip::(anonymous namespace)::", "").replace("sdsp_hip::fft1m::", "").replace("sdsp_hip::fft4096::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:120]


def one(pattern):
    # gpurun MERGES new output into the local folder, so files of earlier runs (other PIDs) linger:
    # always take the newest match
    g = sorted(glob.glob(str(src / pattern)), key=lambda f: Path(f).stat().st_mtime)
    return g[-1] if g else None


lines = [f"# profile {tag}", ""]
bench = json.loads((src / "bench.json").read_text().strip().splitlines()[-1])
lines += ["## bench.py line (un-profiled run, same command)", "", "```json", json.dumps(bench, indent=1), "```", ""]

stats = one("trace/*/*_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
with open(out / f"{tag}_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                    r["MaxNs"], r["StdDev"]])
lines += ["## rocprofv3 --kernel-trace --stats (profiled run)", "",
          "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
for r in rows:
    lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                 f"{float(r['MaxNs'])/1e3:.1f} | {r['Percentage']} |")
lines.append("")

trace = one("trace/*/*_kernel_trace.csv")
if trace:
    t = [r for r in csv.DictReader(open(trace)) if "sdsp" in r["Kernel_Name"]]
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in t]
    lines += ["launch-by-launch durations of the sdsp kernels (us), in order:", "",
              " ".join(f"{d:.0f}" for d in durs), ""]
    # (no resource line here: rocprofv3's VGPR_Count / LDS_Block_Size columns report allocation granules and static LDS only --
    # round 2's summaries said "VGPR 104, LDS 0 B" for a 203-VGPR kernel launched with 9 KiB of dynamic LDS.  The compiler's own
    # figures per kernel are in profiles/r03_kernel_regs.txt (tools/kernel_regs.sh); dynamic LDS sizes are in DESIGN.md section 5.)

traffic = {}
pm = defaultdict(lambda: defaultdict(list))
for name in ("pmc_fetch", "pmc_write"):
    p = one(f"{name}/*/*_counter_collection.csv")
    if not p:
        continue
    for r in csv.DictReader(open(p)):
        if "sdsp" in r["Kernel_Name"]:
            pm[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines += ["## HBM traffic from PMC counters (separate --pmc passes)", "",
          "| kernel | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM bytes/launch = (2*FETCH + WRITE)*1024 |", "|---|---|---|---|"]
for k, c in pm.items():
    f = sum(c["FETCH_SIZE"]) / max(1, len(c["FETCH_SIZE"]))
    wv = sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"]))
    b = (2 * f + wv) * 1024
    traffic[k] = {"hbm_bytes_per_launch": b, "fetch_size_kib_raw": f, "write_size_kib": wv, "profile": tag}
    lines.append(f"| `{k}` | {f:.1f} | {wv:.1f} | {b:.4g} |")
# launches of each kernel per bench step: the bench line's launches_per_step (sdsp_hip_fft_plan_launches: all kernels of the
# path, every workspace slice / launch piece) spread evenly over the path's kernels ("a+b" in roofline.kernel)
rf = bench.get("roofline", {})
path_kernels = [k for k in rf.get("kernel", "").split("+") if k and k != "rows"]
per_kernel = max(1, int(rf.get("launches_per_step", 1)) // max(1, len(path_kernels)))
for k in traffic:
    traffic[k]["launches_per_step"] = per_kernel
alg = rf.get("algorithmic_bytes_per_launch", 0) * rf.get("launches_per_step", 1)  # per STEP
if alg:
    per_step = sum(v["hbm_bytes_per_launch"] * v["launches_per_step"] for v in traffic.values())
    n_inst = max(1, len(traffic) // max(1, len({k.split("<")[0] for k in traffic})))  # fwd/rev instantiations
    lines += ["", f"algorithmic bytes per bench step: {alg}; fabric traffic per step (all kernels, launches per step "
              f"accounted, forward/reverse instantiations averaged): {per_step / n_inst:.4g} -> "
              f"traffic / algorithmic = {per_step / n_inst / alg:.4f}"]
(out / f"{tag}_summary.md").write_text("\n".join(lines) + "\n")

tj = out / "traffic.json"
allt = json.loads(tj.read_text()) if tj.exists() else {}
# key by bare kernel name; forward/reverse instantiations of one template are averaged
merged = {}
for k, v in traffic.items():
    merged.setdefault(k.split("<")[0], []).append(v)
for k, vs in merged.items():
    allt[k] = {"hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] for v in vs) / len(vs),
               # the profiled workload's algorithmic bytes per launch of this kernel: bench.py scales the measured
               # traffic / algorithmic ratio to the launch it reports (same kernel, other batch or sample type)
               "algorithmic_bytes_per_launch": (alg / vs[0]["launches_per_step"]) if alg else None,  # each kernel of a path moves the step's bytes once
               "fetch_size_kib_raw": sum(v["fetch_size_kib_raw"] for v in vs) / len(vs),
               "write_size_kib": sum(v["write_size_kib"] for v in vs) / len(vs),
               "launches_per_step": vs[0]["launches_per_step"], "profile": tag}
    n_cfg = bench.get("config", {}).get("n")
    if n_cfg:  # the same kernel template serves several sizes: also keep the entry under "<kernel>@<n>" (bench.py prefers it)
        allt[f"{k}@{n_cfg}"] = dict(allt[k])
tj.write_text(json.dumps(allt, indent=1) + "\n")
print((out / f"{tag}_summary.md").read_text())
