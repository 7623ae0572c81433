#!/bin/bash
# Round-3 eighth GPU call: the partial-sum ring with add + give-back claims.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ring or strip" > gpurun_out/r3j_pytest.log 2>&1 || { tail -40 gpurun_out/r3j_pytest.log; exit 1; }
tail -2 gpurun_out/r3j_pytest.log
run() { tag=$1; shift
  timeout -k 10 400 python bench.py --no-pmc --no-cpu-baseline --no-plain "$@" > gpurun_out/r3j_$tag.json 2> gpurun_out/r3j_$tag.err || { echo "$tag failed"; tail -5 gpurun_out/r3j_$tag.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/r3j_$tag.json').read().strip().splitlines()[-1]); ms=d['roofline']['device_ms_per_step']
print('$tag', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step trace', ms['wf_trace'], 'shade', ms['wf_shade'], 'passes/step', d['roofline']['launches_per_step'], 'partials GB', round((d['config']['partial_sum_bytes_per_call'] or 0)/1e9, 2))" | tee -a gpurun_out/r3j_ring.log
}
run c3_noring --steps 4 --warmup 1 --partial-ring -1
run c3_ring250_g25 --steps 4 --warmup 1 --partial-ring 256
RT2022_RING_GROUP=10 run c3_ring250_g10 --steps 4 --warmup 1 --partial-ring 256
RT2022_RING_GROUP=50 run c3_ring250_g50 --steps 4 --warmup 1 --partial-ring 256
run c3_noring_b --steps 4 --warmup 1 --partial-ring -1
run c5_auto_f1 --config c5 --steps 2 --warmup 0 --frames-per-call 1
run c5_noring_f1 --config c5 --steps 2 --warmup 0 --frames-per-call 1 --partial-ring -1
