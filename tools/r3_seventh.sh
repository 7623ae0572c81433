#!/bin/bash
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3zc_pytest.log 2>&1 || { tail -40 gpurun_out/r3zc_pytest.log; exit 1; }
tail -2 gpurun_out/r3zc_pytest.log
timeout -k 10 300 python bench.py --config s1e5 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r3zc_bench_s1e5.json 2> gpurun_out/r3zc_bench_s1e5.err || { tail -5 gpurun_out/r3zc_bench_s1e5.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r3zc_bench_s1e5.json').read().strip().splitlines()[-1]); print('s1e5', d['value'], d['roofline']['trace_variant'])"
