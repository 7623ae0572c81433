#!/bin/bash
# Groups of pool segments passing independently on streams of their own (tuning bits 24-27; bench.py --groups): 1 / 2 / 3 / 1 per config, one GPU call.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "c3 --steps 4 --warmup 1" "c2 --config c2 --steps 4 --warmup 1" "c4 --config c4 --steps 4 --warmup 1" "s1e5 --config s1e5 --steps 4 --warmup 1" "c5 --config c5 --steps 1 --warmup 0"; do
  set -- $cfg; tag=$1; shift
  for g in 1 2 3 1; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --groups $g "$@" > gpurun_out/groups_$tag.json 2> gpurun_out/groups_$tag.err || { tail -3 gpurun_out/groups_$tag.err; exit 1; }
    python3 - $tag $g gpurun_out/groups_$tag.json <<'PY'
import sys, json
tag, g, f = sys.argv[1:4]
d = json.loads(open(f).read().strip().splitlines()[-1]); ms = d["roofline"]["device_ms_per_step"]
print(tag, "groups", g, d["value"], d["ms_per_step"], "trace_ms", ms["wf_trace"], "shade_ms", ms["wf_shade"], "instrumented", (d.get("instrumented_call") or {}).get("value"), flush=True)
PY
  done
done 2>&1 | tee gpurun_out/r3ze_groups.log
