#!/bin/bash
# Round-3 second GPU call: VALU calibration (second form), A/B of the register-diet builds, the new bench line, a pass log.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 tools/valu_calib.sh > gpurun_out/r3b_valu_calib.log 2>&1 || { tail -20 gpurun_out/r3b_valu_calib.log; }
grep -E "^mode|busy|per slot|clock|resident" gpurun_out/valu_calib/summary.txt || true
echo "== A/B headline"; tools/ab.sh --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3b_ab_c3.log
echo "== A/B c5"; tools/ab.sh --config c5 --steps 1 --warmup 0 2>&1 | tee gpurun_out/r3b_ab_c5.log
echo "== A/B c2"; tools/ab.sh --config c2 --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3b_ab_c2.log
echo "== A/B c4"; tools/ab.sh --config c4 --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3b_ab_c4.log
echo "== bench"; timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/r3b_bench_c3.json 2> gpurun_out/r3b_bench_c3.err || { tail -5 gpurun_out/r3b_bench_c3.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r3b_bench_c3.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], d.get('plain_path'))
print({k:r[k] for k in ('bound','achieved','peak','unit','frac','traffic')})
print('valu', r.get('valu')); print('shade', r.get('wf_shade')); print('sec8d frac', r['contract_sec8d']['frac'], r['contract_sec8d']['frac_without_lds_served']); print(r.get('limiter'))
"
echo "== passlog"; RT2022_PASS_LOG=1 timeout -k 10 200 python tools/passlog.py 1000 > gpurun_out/r3b_passlog.txt 2>&1 || tail -3 gpurun_out/r3b_passlog.txt
tail -3 gpurun_out/r3b_passlog.txt
