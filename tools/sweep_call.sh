#!/bin/bash
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 python tools/sweep.py final_scene 300 all 2>&1 | grep -v amdgpu.ids > gpurun_out/r3zd_sweep_final_scene.log
timeout -k 10 400 python tools/sweep.py random_scene 300 all 2>&1 | grep -v amdgpu.ids > gpurun_out/r3zd_sweep_random_scene.log
echo done
