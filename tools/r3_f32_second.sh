#!/bin/bash
# Single-precision slab test with the value-relative error bound: parity suite, census of undecided steps, A/B (A / Z = previous build).
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3q_pytest.log 2>&1 || { tail -40 gpurun_out/r3q_pytest.log; exit 1; }
tail -2 gpurun_out/r3q_pytest.log

RT2022_LIB=$PWD/raytracer_2022_amd/variants_lean/C_f32_census.so timeout -k 10 300 python tools/f32_census.py 2>&1 | tee gpurun_out/r3q_f32_census.log
echo "== A/B headline"; tools/ab.sh --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3q_ab_c3.log
echo "== A/B c2"; tools/ab.sh --config c2 --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3q_ab_c2.log
echo "== A/B c4"; tools/ab.sh --config c4 --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3q_ab_c4.log
