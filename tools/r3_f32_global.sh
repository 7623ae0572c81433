#!/bin/bash
# Single-precision slab test on 32-byte node records from HBM / L2 (kF32G: sphere scenes too large for LDS): parity, census with every
# verdict checked, A/B on the synthetic sphere scenes (A / Z = built with -DRT2022_F32_GLOBAL=0).
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3v_pytest.log 2>&1 || { tail -40 gpurun_out/r3v_pytest.log; exit 1; }
tail -2 gpurun_out/r3v_pytest.log
for k in 50 158 500; do
  RT2022_LIB=$PWD/raytracer_2022_amd/variants_lean/C_f32_census.so timeout -k 10 300 python tools/f32_census.py random_scene 600 400 8 $k 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3v_f32_census.log
done
for c in s1e4 s1e5 s1e6; do
  echo "== A/B $c"; tools/ab.sh --config $c --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3v_ab_$c.log
done
