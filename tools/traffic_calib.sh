#!/bin/bash
# HBM counter calibration on the GPU box: known bytes in the path pool's access patterns vs rocprofv3 FETCH_SIZE / WRITE_SIZE.
set -e
OUT=$PWD/gpurun_out/traffic_calib
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 tools/traffic_calib.py > $OUT/f.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 tools/traffic_calib.py > $OUT/w.log 2>&1 || true
python3 - $OUT <<'PY'
import csv, glob, os, sys, re
root = sys.argv[1]
GIB = 1 << 30; buf = 8 * GIB; n = 48e6
known = {0: ('streaming read, 16 B per lane', buf, 0), 1: ('first 64 B of random 128-B records', 64 * n, 0), 2: ('whole random 128-B records', 128 * n, 0),
         3: ('streaming write, 16 B per lane', 0, buf), 4: ('32 B written at +64 of random records', 0, 32 * n), 5: ('64 + 32 B written per random record', 0, 96 * n),
         6: ('single bytes written at random places', 0, 1 * n)}
got = {}
for which, name in (('f', 'FETCH_SIZE'), ('w', 'WRITE_SIZE')):
    for f in glob.glob(os.path.join(root, which, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r'traffic_probe_kernel<(\d)>', row.get('Kernel_Name', ''))
            if m and row['Counter_Name'] == name:
                got.setdefault(int(m.group(1)), {})[name] = got.setdefault(int(m.group(1)), {}).get(name, 0.0) + float(row['Counter_Value']) * 1024.0
print('# known bytes moved vs rocprofv3 counters (KiB -> bytes), 8 GiB buffer, 48 M random accesses; MI355X, ROCm 7.2')
print('%-44s %14s %14s %8s %14s %14s %8s' % ('pattern', 'read known', 'FETCH_SIZE', 'ratio', 'write known', 'WRITE_SIZE', 'ratio'))
for mode, (label, rd, wr) in known.items():
    g = got.get(mode, {})
    fs, ws = g.get('FETCH_SIZE', float('nan')), g.get('WRITE_SIZE', float('nan'))
    print('%-44s %14.3e %14.3e %8s %14.3e %14.3e %8s' % (label, rd, fs, ('%.3f' % (fs / rd)) if rd else '-', wr, ws, ('%.3f' % (ws / wr)) if wr else '-'))
PY
