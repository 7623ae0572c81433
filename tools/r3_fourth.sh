#!/bin/bash
# Round-3 fourth GPU call: parity on the leaner node loop, A/B (previous build at both ends), the what-if builds (sphere / box arm
# arithmetic twice over), six resident workgroups per CU for the sphere-only kernels on the big synthetic scenes.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r3d_pytest.log 2>&1 || { tail -40 gpurun_out/r3d_pytest.log; exit 1; }
tail -2 gpurun_out/r3d_pytest.log
echo "== A/B headline"; tools/ab.sh --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3d_ab_c3.log
echo "== A/B c2 (P, Q, Z only)"; mkdir -p /tmp/keep; mv raytracer_2022_amd/variants/W_sphere2x.so raytracer_2022_amd/variants/X_box2x.so /tmp/keep/
tools/ab.sh --config c2 --steps 8 --warmup 1 2>&1 | tee gpurun_out/r3d_ab_c2.log
echo "== A/B c5"; tools/ab.sh --config c5 --steps 1 --warmup 0 2>&1 | tee gpurun_out/r3d_ab_c5.log
echo "== A/B c4"; tools/ab.sh --config c4 --steps 8 --warmup 1 2>&1 | tee gpurun_out/r3d_ab_c4.log
echo "== lean kernels at 5 / 6 workgroups per CU"
for c in s1e5 s1e6; do for lib in raytracer_2022_amd/librt2022.so raytracer_2022_amd/variants_lean/L6.so raytracer_2022_amd/librt2022.so raytracer_2022_amd/variants_lean/L6.so; do
  RT2022_LIB=$PWD/$lib timeout -k 10 200 python bench.py --config $c --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-plain > gpurun_out/r3d_lean.json 2> gpurun_out/r3d_lean.err || { tail -3 gpurun_out/r3d_lean.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/r3d_lean.json').read().strip().splitlines()[-1]); ms=d['roofline']['device_ms_per_step']
print('$c', '$lib'.split('/')[-1], d['value'], d['ms_per_step'], 'trace', ms['wf_trace'], 'shade', ms['wf_shade'])" | tee -a gpurun_out/r3d_lean.log
done; done
