#!/bin/bash
# What the driver runs at round end, rehearsed: the GPU suite, smoke(), the default bench line.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3zs_pytest.log 2>&1 || { tail -40 gpurun_out/r3zs_pytest.log; exit 1; }
tail -2 gpurun_out/r3zs_pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/r3zs_bench_default.json 2> gpurun_out/r3zs_bench_default.err || { tail -5 gpurun_out/r3zs_bench_default.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r3zs_bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['metric'], d['value'], d['unit'], d['ms_per_step'], 'steps', d['steps'], 'calls', d['config']['calls'], 'bound', r['bound'], r['frac'], 'sec8d', r['contract_sec8d']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'instrumented call', d['instrumented_call']['value'])"
timeout -k 10 300 python bench.py --config c1 --steps 8 --warmup 2 > gpurun_out/r3zs_bench_c1.json 2> gpurun_out/r3zs_bench_c1.err || { tail -5 gpurun_out/r3zs_bench_c1.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r3zs_bench_c1.json').read().strip().splitlines()[-1]); print('c1', d['value'], d['ms_per_step'], d['config']['pool_slots'], d['cpu_baseline']['full_frame'])"
timeout -k 10 500 python tools/parity_sweep.py > gpurun_out/r3zs_parity_sweep.log 2>&1 || { tail -5 gpurun_out/r3zs_parity_sweep.log; exit 1; }
tail -1 gpurun_out/r3zs_parity_sweep.log
