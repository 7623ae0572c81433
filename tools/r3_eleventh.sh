#!/bin/bash
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden" > gpurun_out/r3n_pytest.log 2>&1 || { tail -40 gpurun_out/r3n_pytest.log; exit 1; }
tail -2 gpurun_out/r3n_pytest.log
echo "== A/B c5 (T = triangle record fetched twice)"; tools/ab.sh --config c5 --steps 1 --warmup 0 2>&1 | tee gpurun_out/r3n_ab_c5.log
