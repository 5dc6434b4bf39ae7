#!/bin/bash
# Run a list of GPU steps one after the other on the gpurun box; each under its own timeout, each logging to
# gpurun_out/<name>.log.  A step that is killed or times out (rc >= 124) ends the script: no further GPU work is
# started after a kill.  Ordinary failures (a failing test, a lab mismatch) are recorded and the next step runs.
#   tools/gpu_steps.sh "name|seconds|command" ...
mkdir -p gpurun_out
overall=0
for spec in "$@"; do
    name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
    echo "=== step $name (limit ${secs}s): $cmd"
    start=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== step $name rc=$rc after $(( $(date +%s) - start ))s"
    tail -n 6 "gpurun_out/$name.log"
    if [ $rc -ge 124 ]; then
        echo "=== step $name was killed: stopping"
        exit $rc
    fi
    [ $rc -ne 0 ] && overall=1
done
exit $overall
