#!/bin/bash
# Round-3 opening GPU call: parity suite, counter list, VALU calibration, kernel summaries of all four configs (baseline build).
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1 || { tail -30 gpurun_out/r3a_pytest.log; exit 1; }
tail -2 gpurun_out/r3a_pytest.log
rocprofv3 -L > gpurun_out/r3a_counters.txt 2>&1 || true
echo "counters listed: $(wc -l < gpurun_out/r3a_counters.txt) lines"
timeout -k 10 300 tools/valu_calib.sh > gpurun_out/r3a_valu_calib.log 2>&1 || { tail -20 gpurun_out/r3a_valu_calib.log; }
tail -60 gpurun_out/valu_calib/summary.txt || true
timeout -k 10 240 tools/prof_r2.sh r3a_c3 --steps 5 --warmup 1 && echo c3 done
timeout -k 10 240 tools/prof_r2.sh r3a_c2 --config c2 --steps 5 --warmup 1 && echo c2 done
timeout -k 10 240 tools/prof_r2.sh r3a_c4 --config c4 --steps 5 --warmup 1 && echo c4 done
timeout -k 10 400 tools/prof_r2.sh r3a_c5 --config c5 --steps 2 --warmup 1 && echo c5 done
