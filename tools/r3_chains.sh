#!/bin/bash
# Movers nested directly in one another entered and left in one turn (RT2022_CTX_CHAINS): parity, then A/B on C5 and the headline (A / Z = built with -DRT2022_CTX_CHAINS=0).
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3z_pytest.log 2>&1 || { tail -40 gpurun_out/r3z_pytest.log; exit 1; }
tail -2 gpurun_out/r3z_pytest.log
echo "== A/B c5"; tools/ab.sh --config c5 --steps 1 --warmup 0 2>&1 | tee gpurun_out/r3z_ab_c5.log
echo "== A/B headline"; tools/ab.sh --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3z_ab_c3.log
