#!/bin/bash
# VALU counter calibration on the GPU box: kernels whose vector issue slots are full by construction (rt_debug_valu_probe)
# under the SQ counters bench.py reads for the traversal kernel. Output: gpurun_out/valu_calib/summary.txt.
set -e
OUT=$PWD/gpurun_out/valu_calib
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 tools/valu_calib.py > $OUT/a.log 2>&1 || true
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 tools/valu_calib.py > $OUT/b.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/valu_calib.py > $OUT/t.log 2>&1 || true
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys, re
root = sys.argv[1]
ITERS = 100000
modes = {0: 'v_fma_f64, all lanes, 8 waves/SIMD', 1: '32-bit add/xor, all lanes, 8 waves/SIMD', 2: 'v_fma_f64, 32 of 64 lanes, 8 waves/SIMD',
         3: 'f64 and 32-bit alternating, 8 waves/SIMD', 4: 'v_fma_f64, all lanes, 1 wave/SIMD'}
got, dur = {}, {}
for which in ('a', 'b'):
    for f in glob.glob(os.path.join(root, which, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r'valu_probe_kernel<(\d)>', row.get('Kernel_Name', ''))
            if m:
                d = got.setdefault(int(m.group(1)), {})
                d[row['Counter_Name']] = d.get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
for f in glob.glob(os.path.join(root, 't', '**', '*kernel_trace.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r'valu_probe_kernel<(\d)>', row.get('Kernel_Name', ''))
        if m:
            dur[int(m.group(1))] = (float(row['End_Timestamp']) - float(row['Start_Timestamp'])) * 1e-9
print('# rt_debug_valu_probe under rocprofv3 --pmc; MI355X, ROCm 7.2; %d rounds of 8 vector instructions per lane (+ loop overhead)' % ITERS)
names = sorted({k for d in got.values() for k in d})
for mode, label in modes.items():
    g = got.get(mode, {})
    print('mode %d: %s' % (mode, label))
    print('   kernel time (trace pass) %.3f ms' % (dur.get(mode, float('nan')) * 1e3))
    for k in names:
        print('   %-26s %.6e' % (k, g.get(k, float('nan'))))
    gui = g.get('GRBM_GUI_ACTIVE', 0.0)
    if gui:
        simd_cycles = gui / 8.0 * 1024.0            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
        print('   SIMD-cycles = GRBM_GUI_ACTIVE / 8 x 1024 = %.4e' % simd_cycles)
        for k in ('SQ_ACTIVE_INST_VALU', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_INSTS_VALU', 'SQ_THREAD_CYCLES_VALU', 'SQ_ACTIVE_INST_ANY', 'SQ_BUSY_CU_CYCLES', 'SQ_CYCLES'):
            if k in g:
                print('   %-26s / SIMD-cycles = %.4f' % (k, g[k] / simd_cycles))
        if dur.get(mode):
            print('   clock = GRBM_GUI_ACTIVE / 8 / time(trace pass) = %.3f GHz' % (gui / 8.0 / dur[mode] / 1e9))
PY
