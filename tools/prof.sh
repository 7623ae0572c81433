#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]
# Writes rocprofv3 kernel-trace stats and two PMC passes under gpurun_out/prof_<tag>/.
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="$@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err || true
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 bench.py $ARGS --no-cpu-baseline > $OUT/bench_pmc1.json 2> $OUT/pmc1.err || true
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc2 -- python3 bench.py $ARGS --no-cpu-baseline > $OUT/bench_pmc2.json 2> $OUT/pmc2.err || true
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc3 -- python3 bench.py $ARGS --no-cpu-baseline > $OUT/bench_pmc3.json 2> $OUT/pmc3.err || true
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- python3 bench.py $ARGS --no-cpu-baseline > $OUT/bench_pmc4.json 2> $OUT/pmc4.err || true
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $OUT/pmc5 -- python3 bench.py $ARGS --no-cpu-baseline > $OUT/bench_pmc5.json 2> $OUT/pmc5.err || true
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
