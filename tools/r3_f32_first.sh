#!/bin/bash
# Single-precision slab test, first GPU call: the parity suite on the new build, then A/B against the previous build (A / Z = base at both ends).
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3p_pytest.log 2>&1 || { tail -40 gpurun_out/r3p_pytest.log; exit 1; }
tail -2 gpurun_out/r3p_pytest.log
echo "== A/B headline"; tools/ab.sh --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3p_ab_c3.log
echo "== A/B c2"; tools/ab.sh --config c2 --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3p_ab_c2.log
