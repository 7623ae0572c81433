#!/bin/bash
# Round-3 final collection (second half: final build): parity suite, an N = 2 rehearsal with several steps per call, rocprofv3 kernel summaries and bench lines of every config.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3zg_pytest.log 2>&1 || { tail -40 gpurun_out/r3zg_pytest.log; exit 1; }
tail -2 gpurun_out/r3zg_pytest.log
RT2022_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --spp 16 --no-cpu-baseline > gpurun_out/r3zg_rehearsal_n2.json 2> gpurun_out/r3zg_rehearsal_n2.err || { tail -5 gpurun_out/r3zg_rehearsal_n2.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r3zg_rehearsal_n2.json').read().strip().splitlines()[-1]); print('N=2 rehearsal:', d['n_gpus'], d['steps'], d['config']['calls'], d['value'], d['data'][:40])"
timeout -k 10 300 tools/prof_r2.sh r3zg_c3 --steps 8 --warmup 1 && echo c3 done
timeout -k 10 240 tools/prof_r2.sh r3zg_c2 --config c2 --steps 8 --warmup 1 && echo c2 done
timeout -k 10 240 tools/prof_r2.sh r3zg_c4 --config c4 --steps 8 --warmup 1 && echo c4 done
timeout -k 10 500 tools/prof_r2.sh r3zg_c5 --config c5 --steps 2 --warmup 1 && echo c5 done
for c in c1 s1e4 s1e5 s1e6; do
  timeout -k 10 300 python bench.py --config $c --steps 4 --warmup 1 > gpurun_out/r3zg_bench_$c.json 2> gpurun_out/r3zg_bench_$c.err || { tail -5 gpurun_out/r3zg_bench_$c.err; exit 1; }
done
echo benches done
