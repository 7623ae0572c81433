#!/bin/bash
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3x_pytest.log 2>&1 || { tail -40 gpurun_out/r3x_pytest.log; exit 1; }
tail -2 gpurun_out/r3x_pytest.log
timeout -k 10 500 python tools/parity_sweep.py > gpurun_out/r3x_parity_sweep.log 2>&1 || { tail -5 gpurun_out/r3x_parity_sweep.log; exit 1; }
tail -1 gpurun_out/r3x_parity_sweep.log
for c in s1e4 s1e5 s1e6; do
  timeout -k 10 300 python bench.py --config $c --steps 4 --warmup 1 > gpurun_out/r3x_bench_$c.json 2> gpurun_out/r3x_bench_$c.err || { tail -5 gpurun_out/r3x_bench_$c.err; exit 1; }
done
python3 -c "
import json
for c in ('s1e4','s1e5','s1e6'):
    d=json.loads(open('gpurun_out/r3x_bench_%s.json' % c).read().strip().splitlines()[-1]); r=d['roofline']
    print(c, d['value'], d['unit'], d['ms_per_step'], 'bound', r['bound'], r['frac'], 'sec8d', r['contract_sec8d']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'variant', r.get('trace_variant'))"
