#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/prof_r2.sh <tag> [bench args...]
# 1. rocprofv3 --kernel-trace --stats of bench.py (no PMC in this pass, no CPU leg) -> gpurun_out/prof_<tag>/kernel_stats.txt
# 2. the plain bench line of the same command, with its own in-run PMC passes and the CPU baseline -> gpurun_out/prof_<tag>/bench.json
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# (--groups 1: the traced run keeps the pool in one group, like the instrumented call the bench line's launch durations come from — with the
# library's default of two groups the launches of one group overlap the other's and rocprofv3's averages are those stretched durations)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py "$@" --groups 1 --no-pmc --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err || true
python3 - $OUT <<'PY' > $OUT/kernel_stats.txt
import csv, glob, os, sys, json
root = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(root, 'trace', '**', '*kernel_stats.csv'), recursive=True)):
    rows += list(csv.DictReader(open(f)))
print('# rocprofv3 --kernel-trace --stats -- python3 bench.py <args> --groups 1 --no-pmc --no-cpu-baseline')
for r in rows:
    print('%-110s calls %6s  total %10.3f ms  avg %9.4f ms  %6s %%' % (r['Name'][:110], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e6, r['Percentage']))
try:
    d = json.loads(open(os.path.join(root, 'bench_under_trace.json')).read().strip().splitlines()[-1])
    ro = d['roofline']
    print('# bench.py, same run, HIP events: wf_trace %.3f ms per step over %d launches = %.4f ms per launch; wf_shade %.3f ms per step; value %.1f %s, %.1f ms per step'
          % (ro['device_ms_per_step']['wf_trace'], ro['launches_per_step'], ro['launch_ms'], ro['device_ms_per_step']['wf_shade'], d['value'], d['unit'], d['ms_per_step']))
except Exception as e:
    print('# bench line not parsed:', e)
PY
cat $OUT/kernel_stats.txt
python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || true
tail -c 400 $OUT/bench.err
