#!/bin/bash
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 python tools/two_streams.py final_scene 800 800 1000 2 2>&1 | tee gpurun_out/r3t_two_streams.log
timeout -k 10 200 python tools/two_streams.py final_scene 800 800 1000 1 2>&1 | tee -a gpurun_out/r3t_two_streams.log
