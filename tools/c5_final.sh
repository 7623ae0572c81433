#!/bin/bash
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3zn_pytest.log 2>&1 || { tail -40 gpurun_out/r3zn_pytest.log; exit 1; }
tail -2 gpurun_out/r3zn_pytest.log
RT2022_LIB=$PWD/raytracer_2022_amd/variants_lean/C_f32_census.so timeout -k 10 300 python tools/f32_census.py wwscene 480 270 8 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3zn_f32_census.log
timeout -k 10 500 tools/prof_r2.sh r3zn_c5 --config c5 --steps 2 --warmup 1 | grep -v "^void\|^rt2022::\|^__amd\|^(anonymous\|amdgpu.ids" && echo c5 done
