#!/bin/bash
# C5 (wwscene): fast-path quorum 12 / 18 / 24 / 32 (bench.py --quorum), then triangle tests per turn 1 / 3 against the shipped 2.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
for q in 18 12 24 32 18; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --config c5 --steps 1 --warmup 0 --quorum $q --groups 1 > gpurun_out/c5q.json 2> gpurun_out/c5q.err || { tail -3 gpurun_out/c5q.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/c5q.json').read().strip().splitlines()[-1]); ms=d['roofline']['device_ms_per_step']
print('c5 quorum $q', d['value'], d['ms_per_step'], 'trace_ms', ms['wf_trace'], 'shade_ms', ms['wf_shade'], flush=True)"
done 2>&1 | tee gpurun_out/r3zj_c5_quorum.log
echo "== triangle reps"; tools/ab.sh --config c5 --steps 1 --warmup 0 2>&1 | tee gpurun_out/r3zj_c5_misc_reps.log
