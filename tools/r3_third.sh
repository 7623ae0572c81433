#!/bin/bash
# Round-3 third GPU call: parity suite on the new build, A/B of MachineLICM on / off (with the hand-hoisted node loop), A/B of
# frames per call, bench lines of c1 and the synthetic sphere scenes.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r3c_pytest.log 2>&1 || { tail -40 gpurun_out/r3c_pytest.log; exit 1; }
tail -2 gpurun_out/r3c_pytest.log
echo "== A/B headline"; tools/ab.sh --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3c_ab_c3.log
echo "== A/B c2"; tools/ab.sh --config c2 --steps 8 --warmup 1 2>&1 | tee gpurun_out/r3c_ab_c2.log
echo "== A/B c5"; tools/ab.sh --config c5 --steps 1 --warmup 0 2>&1 | tee gpurun_out/r3c_ab_c5.log
echo "== frames per call (default library)"
for f in 1 4 1 4; do
  timeout -k 10 200 python bench.py --steps 8 --warmup 1 --frames-per-call $f --no-pmc --no-cpu-baseline --no-plain > gpurun_out/r3c_fpc$f.json 2> gpurun_out/r3c_fpc$f.err || { tail -5 gpurun_out/r3c_fpc$f.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r3c_fpc$f.json').read().strip().splitlines()[-1]); ms=d['roofline']['device_ms_per_step']
print('frames per call $f:', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step  trace', ms['wf_trace'], 'shade', ms['wf_shade'], 'passes/step', d['roofline']['launches_per_step'])" | tee -a gpurun_out/r3c_fpc.log
done
for c in c1 s1e4 s1e5 s1e6; do
  timeout -k 10 300 python bench.py --config $c --steps 4 --warmup 1 > gpurun_out/r3c_bench_$c.json 2> gpurun_out/r3c_bench_$c.err || { tail -5 gpurun_out/r3c_bench_$c.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/r3c_bench_$c.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$c', d['value'], d['ms_per_step'], 'bound', r['bound'], r['frac'], 'sec8d', r['contract_sec8d']['frac'], 'hbm', (r.get('hbm_counter') or {}).get('frac'), 'valu', (r.get('valu') or {}).get('busy'), 'cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('full_frame'))"
done
