#!/bin/bash
# Round-3 sixth GPU call: groups sweep, then the final build's numbers — rocprofv3 kernel summaries and bench lines of every config.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== groups"; timeout -k 10 300 python tools/sweep.py final_scene 1000 groups 2>&1 | tee gpurun_out/r3f_groups.log
timeout -k 10 300 tools/prof_r2.sh r3f_c3 --steps 8 --warmup 1 && echo c3 done
timeout -k 10 240 tools/prof_r2.sh r3f_c2 --config c2 --steps 8 --warmup 1 && echo c2 done
timeout -k 10 240 tools/prof_r2.sh r3f_c4 --config c4 --steps 8 --warmup 1 && echo c4 done
timeout -k 10 500 tools/prof_r2.sh r3f_c5 --config c5 --steps 2 --warmup 1 && echo c5 done
for c in c1 s1e4 s1e5 s1e6; do
  timeout -k 10 300 python bench.py --config $c --steps 4 --warmup 1 > gpurun_out/r3f_bench_$c.json 2> gpurun_out/r3f_bench_$c.err || { tail -5 gpurun_out/r3f_bench_$c.err; exit 1; }
done
echo benches done
