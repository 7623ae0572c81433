#!/bin/bash
# The configs whose kernels the mover chains changed (headline, C5), collected again on the final build, and the default bench line.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3zb_pytest.log 2>&1 || { tail -40 gpurun_out/r3zb_pytest.log; exit 1; }
tail -2 gpurun_out/r3zb_pytest.log
timeout -k 10 300 tools/prof_r2.sh r3zb_c3 --steps 8 --warmup 1 && echo c3 done
timeout -k 10 500 tools/prof_r2.sh r3zb_c5 --config c5 --steps 2 --warmup 1 && echo c5 done
timeout -k 10 600 python bench.py > gpurun_out/r3zb_bench_default.json 2> gpurun_out/r3zb_bench_default.err || { tail -5 gpurun_out/r3zb_bench_default.err; exit 1; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
