#!/bin/bash
# Parity suite on the build with RT_FLAG_ASYNC and the hostile-spheres test.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r3s_pytest.log 2>&1 || { tail -60 gpurun_out/r3s_pytest.log; exit 1; }
tail -2 gpurun_out/r3s_pytest.log
