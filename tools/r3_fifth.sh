#!/bin/bash
# Round-3 fifth GPU call: parity, A/B of the node-loop build (C5 fix), fetch what-ifs (sphere / box record fetched twice).
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r3e_pytest.log 2>&1 || { tail -40 gpurun_out/r3e_pytest.log; exit 1; }
tail -2 gpurun_out/r3e_pytest.log
echo "== A/B headline"; tools/ab.sh --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3e_ab_c3.log
mkdir -p /tmp/keep; mv raytracer_2022_amd/variants/F_sphere_fetch2x.so raytracer_2022_amd/variants/G_box_fetch2x.so /tmp/keep/
echo "== A/B c5"; tools/ab.sh --config c5 --steps 1 --warmup 0 2>&1 | tee gpurun_out/r3e_ab_c5.log
echo "== A/B s1e6"; tools/ab.sh --config s1e6 --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3e_ab_s1e6.log
