#!/bin/bash
# Pool size (segments of 4096 slots per resident traversal workgroup; bench.py --segs): 8 / 4 / 6 / 8 per config, one GPU call.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "c2 --config c2 --steps 4 --warmup 1" "c3 --steps 4 --warmup 1" "c4 --config c4 --steps 4 --warmup 1" "s1e5 --config s1e5 --steps 4 --warmup 1" "c1 --config c1 --steps 8 --warmup 2"; do
  set -- $cfg; tag=$1; shift
  for g in 8 4 6 8; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --segs $g "$@" > gpurun_out/segs_$tag.json 2> gpurun_out/segs_$tag.err || { tail -3 gpurun_out/segs_$tag.err; exit 1; }
    python3 - $tag $g gpurun_out/segs_$tag.json <<'PY'
import sys, json
tag, g, f = sys.argv[1:4]
d = json.loads(open(f).read().strip().splitlines()[-1]); ms = d["roofline"]["device_ms_per_step"]
print(tag, "segs", g, d["value"], d["ms_per_step"], "one group:", (d.get("instrumented_call") or {}).get("value"), "trace_ms", ms["wf_trace"], "shade_ms", ms["wf_shade"], "launches", d["roofline"]["launches_per_step"], flush=True)
PY
  done
done 2>&1 | tee gpurun_out/r3zh_segs.log
