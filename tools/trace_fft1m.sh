#!/bin/bash
# per-kernel durations of the fft1m passes for a list of plan variants (chunk sizes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for v in "$@"; do
  rm -rf /tmp/tr_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -- python3 $R/tools/sweep_fft1m.py $v > /tmp/tr_$v.log 2>&1
  echo "== $(grep variant /tmp/tr_$v.log)"
  python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("/tmp/tr_$v/*/*_kernel_stats.csv")[0])):
    if "fft1m" in r["Name"]:
        nm = ("cols" if "cols" in r["Name"] else "rows") + ("<rev>" if "<true>" in r["Name"] else "<fwd>")
        print(f"   {nm:12s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
