#!/bin/bash
# The single-precision slab test on the GPU box (how profiles/r3q_ab_f32_slabs.log and r3v_ab_f32_global.log were made): parity suite,
# census with every verdict checked against the double-precision test, A/B on the synthetic sphere scenes. Build first, in the container:
#   make -C raytracer_2022_amd/csrc BUILD=build_A OUT=../variants/A_f64.so EXTRA=-DRT2022_F32_GLOBAL=0 && cp raytracer_2022_amd/variants/A_f64.so raytracer_2022_amd/variants/Z_f64.so
#   cp raytracer_2022_amd/librt2022.so raytracer_2022_amd/variants/B_f32.so
#   make -C raytracer_2022_amd/csrc BUILD=build_C OUT=../variants_lean/C_f32_census.so EXTRA="-DRT2022_F32_CENSUS -DRT2022_F32_SLABS=2"
# (-DRT2022_F32_SLABS=0 / =2 and tools/f32_census.py <scene> for the LDS variant per instance).
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
