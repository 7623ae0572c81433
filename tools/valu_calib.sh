#!/bin/bash
# VALU counter calibration on the GPU box: kernels whose vector issue slots are full by construction (rt_debug_valu_probe)
# under the SQ counters bench.py reads for the traversal kernel. Output: gpurun_out/valu_calib/summary.txt.
set -e
OUT=$PWD/gpurun_out/valu_calib
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 tools/valu_calib.py > $OUT/a.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 tools/valu_calib.py > $OUT/b.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/valu_calib.py > $OUT/t.log 2>&1 || true
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys, re
root = sys.argv[1]
ITERS = 12500
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
print('# rt_debug_valu_probe under rocprofv3 --pmc; MI355X, ROCm 7.2; %d rounds of 64 vector instructions per lane (+ 3 scalar instructions per round)' % ITERS)
names = sorted({k for d in got.values() for k in d})
for mode, label in modes.items():
    g = got.get(mode, {})
    print('mode %d: %s' % (mode, label))
    print('   kernel time (trace pass) %.3f ms' % (dur.get(mode, float('nan')) * 1e3))
    for k in names:
        print('   %-26s %.6e' % (k, g.get(k, float('nan'))))
    # SQ_BUSY_CYCLES: cycles with a wave present, summed over the 32 shader engines (4 per XCD x 8 XCDs) -> cycles of the kernel;
    # SIMD quad-cycles = 1024 SIMDs x cycles / 4: the issue slots of the vector pipes (an f64 instruction holds one for a whole
    # quad-cycle, two 32-bit ones from different waves can share one: SQ_ACTIVE_INST_VALU2 counts the quad-cycles where that happened)
    cyc = g.get('SQ_BUSY_CYCLES', 0.0) / 32.0
    if cyc:
        slots = 1024.0 * cyc / 4.0
        iv, v2 = g.get('SQ_INSTS_VALU', 0.0), g.get('SQ_ACTIVE_INST_VALU2', 0.0)
        print('   cycles = SQ_BUSY_CYCLES / 32 = %.4e  (GRBM_GUI_ACTIVE / 16 = %.4e); clock = cycles / time(trace pass) = %.3f GHz' % (cyc, g.get('GRBM_GUI_ACTIVE', 0.0) / 16.0, cyc / dur.get(mode, float('nan')) / 1e9))
        print('   SIMD quad-cycles (issue slots) = 1024 x cycles / 4 = %.4e' % slots)
        print('   SQ_INSTS_VALU / slots                           = %.4f   (instructions per slot: <= 1 for f64, <= 2 for 32-bit)' % (iv / slots))
        print('   SQ_ACTIVE_INST_VALU / slots                     = %.4f' % (g.get('SQ_ACTIVE_INST_VALU', 0.0) / slots))
        print('   (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2) / slots  = %.4f   <- VALU busy: share of the slots in which the vector pipe issued' % ((iv - v2) / slots))
        print('   SQ_WAVE_CYCLES / (waves x cycles / 4)           = %.4f   (share of the kernel a wave was resident)' % (g.get('SQ_WAVE_CYCLES', 0.0) / ((1024.0 if mode == 4 else 8192.0) * cyc / 4.0)))
PY
