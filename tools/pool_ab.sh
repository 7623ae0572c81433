#!/bin/bash
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3zi_pytest.log 2>&1 || { tail -40 gpurun_out/r3zi_pytest.log; exit 1; }
tail -2 gpurun_out/r3zi_pytest.log
for cfg in "c1 --config c1 --steps 8 --warmup 2" "c1f1 --config c1 --steps 8 --warmup 2 --frames-per-call 1" "c2s60 --config c2 --steps 4 --warmup 1 --spp 60" "c4s100 --config c4 --steps 4 --warmup 1 --spp 100"; do
  set -- $cfg; tag=$1; shift
  for g in 0 8 0; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --segs $g "$@" > gpurun_out/pool_$tag.json 2> gpurun_out/pool_$tag.err || { tail -3 gpurun_out/pool_$tag.err; exit 1; }
    python3 - $tag $g gpurun_out/pool_$tag.json <<'PY'
import sys, json
tag, g, f = sys.argv[1:4]
d = json.loads(open(f).read().strip().splitlines()[-1]); ms = d["roofline"]["device_ms_per_step"]
print(tag, "segs", g, d["value"], d["ms_per_step"], "pool_slots", d["config"]["pool_slots"], "launches", d["roofline"]["launches_per_step"], flush=True)
PY
  done
done 2>&1 | tee gpurun_out/r3zi_pool.log
