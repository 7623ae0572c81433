#!/bin/bash
# Voted node arm: lanes whose next entry is a node again take it in the same turn (RT2022_NODE_REPS = 2 / 4 against 1 at both ends).
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== A/B headline"; tools/ab.sh --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3za_ab_c3.log
echo "== A/B c2"; tools/ab.sh --config c2 --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3za_ab_c2.log
echo "== A/B s1e5"; tools/ab.sh --config s1e5 --steps 4 --warmup 1 2>&1 | tee gpurun_out/r3za_ab_s1e5.log
