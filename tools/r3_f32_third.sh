#!/bin/bash
# Single-precision slab test, sphere-only instance: parity suite, parity sweep of the timed kernels, census with every verdict
# checked against the double-precision test (census build: the test in every whole-table instance), A/B on C1 / C2 (A / Z = built with -DRT2022_F32_SLABS=0).
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3r_pytest.log 2>&1 || { tail -40 gpurun_out/r3r_pytest.log; exit 1; }
tail -2 gpurun_out/r3r_pytest.log
RT2022_LIB=$PWD/raytracer_2022_amd/variants_lean/C_f32_census.so timeout -k 10 300 python tools/f32_census.py 2>&1 | tee gpurun_out/r3r_f32_census.log
timeout -k 10 500 python tools/parity_sweep.py > gpurun_out/r3r_parity_sweep.log 2>&1 || { tail -5 gpurun_out/r3r_parity_sweep.log; exit 1; }
tail -1 gpurun_out/r3r_parity_sweep.log
echo "== A/B c2"; tools/ab.sh --config c2 --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3r_ab_c2.log
echo "== A/B c1"; tools/ab.sh --config c1 --steps 8 --warmup 2 2>&1 | tee gpurun_out/r3r_ab_c1.log
echo "== A/B c4 (unchanged kernel: control)"; tools/ab.sh --config c4 --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3r_ab_c4.log
