#!/bin/bash
# Round 3, second half: what changed since tools/r3_collect.sh — the sphere-only kernel (C1, C2) — collected again on the final build,
# plus the default bench line (headline: its kernels are unchanged) and the parity suite.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3u_pytest.log 2>&1 || { tail -40 gpurun_out/r3u_pytest.log; exit 1; }
tail -2 gpurun_out/r3u_pytest.log
timeout -k 10 300 tools/prof_r2.sh r3u_c2 --config c2 --steps 8 --warmup 1 && echo c2 done
timeout -k 10 300 python bench.py --config c1 --steps 4 --warmup 1 > gpurun_out/r3u_bench_c1.json 2> gpurun_out/r3u_bench_c1.err || { tail -5 gpurun_out/r3u_bench_c1.err; exit 1; }
timeout -k 10 600 python bench.py > gpurun_out/r3u_bench_default.json 2> gpurun_out/r3u_bench_default.err || { tail -5 gpurun_out/r3u_bench_default.err; exit 1; }
python3 -c "
import json
for f in ('gpurun_out/prof_r3u_c2/bench.json', 'gpurun_out/r3u_bench_c1.json', 'gpurun_out/r3u_bench_default.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f, d['value'], d['unit'], d['ms_per_step'], 'bound', r['bound'], r['frac'], 'sec8d', r['contract_sec8d']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'variant', r.get('trace_variant'))"
