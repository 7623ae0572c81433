#!/bin/bash
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
RT2022_LIB=$PWD/raytracer_2022_amd/variants/B_f32mesh_nostash.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5 or golden or assets or hand_built" > gpurun_out/r3zm_pytest.log 2>&1 || { tail -30 gpurun_out/r3zm_pytest.log; exit 1; }
tail -2 gpurun_out/r3zm_pytest.log
echo "== A/B c5"; tools/ab.sh --config c5 --steps 1 --warmup 0 2>&1 | tee gpurun_out/r3zm_ab_c5.log
