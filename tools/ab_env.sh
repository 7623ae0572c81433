#!/bin/bash
# A/B of run-time switches with ONE library in ONE GPU call: tools/ab_env.sh "<VAR=val ...>" "<VAR=val ...>" ... -- <bench args>
# runs bench.py once per environment setting and prints value / ms per step / per-kernel device time for each.
settings=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do settings+=("$1"); shift; done
shift
i=0
for st in "${settings[@]}"; do
  i=$((i+1))
  env $st timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --no-plain "$@" > gpurun_out/abenv_$i.log 2>&1 || { echo "[$st]: bench failed, stopping"; tail -3 gpurun_out/abenv_$i.log; exit 1; }
  python3 - "$st" gpurun_out/abenv_$i.log <<'PY'
import sys, json
n, f = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    ro = d.get("roofline", {})
    dm = ro.get("device_ms_per_step", {})
    print("[%s]" % n, d["value"], d["ms_per_step"], "trace_ms", dm.get("wf_trace"), "shade_ms", dm.get("wf_shade"), "launch_ms", ro.get("launch_ms"), flush=True)
except Exception as e:
    print(n, "failed", open(f).read()[-300:], flush=True)
PY
done
