#!/bin/bash
# A/B of library builds on the GPU box: tools/ab.sh <bench args> -- runs bench.py once per raytracer_2022_amd/variants/*.so
# (RT2022_LIB selects the build) and prints value / ms per step for each.
for so in raytracer_2022_amd/variants/*.so; do
  n=$(basename $so .so)
  RT2022_LIB=$PWD/$so timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --no-plain "$@" > gpurun_out/ab_$n.log 2>&1 || { echo "$n: bench failed, stopping"; tail -3 gpurun_out/ab_$n.log; exit 1; }
  python3 - "$n" gpurun_out/ab_$n.log <<'PY'
import sys, json
n, f = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    ms = d["roofline"]["device_ms_per_step"]
    print(n, d["value"], d["ms_per_step"], "trace_ms", ms["wf_trace"], "shade_ms", ms["wf_shade"], flush=True)
except Exception as e:
    print(n, "failed", open(f).read()[-300:], flush=True)
PY
done
