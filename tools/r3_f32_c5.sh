#!/bin/bash
# kF32G in the kernels with triangles / movers (RT2022_F32_GLOBAL=2): census on wwscene, A/B on C5 (A / Z = RT2022_F32_GLOBAL=1: sphere scenes only).
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
RT2022_LIB=$PWD/raytracer_2022_amd/variants_lean/C_f32_census.so timeout -k 10 300 python tools/f32_census.py wwscene 480 270 8 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3w_f32_census.log
RT2022_LIB=$PWD/raytracer_2022_amd/variants/B_f32.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5 or golden or assets" > gpurun_out/r3w_pytest.log 2>&1 || { tail -30 gpurun_out/r3w_pytest.log; exit 1; }
tail -2 gpurun_out/r3w_pytest.log
echo "== A/B c5"; tools/ab.sh --config c5 --steps 1 --warmup 0 2>&1 | tee gpurun_out/r3w_ab_c5.log
