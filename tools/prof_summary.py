"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv, glob, os, sys, collections
root = sys.argv[1]
def find(pattern):
    return sorted(glob.glob(os.path.join(root, '**', pattern), recursive=True))
print('== kernel stats ==')
for f in find('*kernel_stats.csv'):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print({k: row[k] for k in row if k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs')})
print('== kernel trace (megakernel dispatches) ==')
for f in find('*kernel_trace.csv'):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if any(k in row.get('Kernel_Name', '') for k in ('pt_megakernel',)):
                dur = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6
                print('dispatch', row.get('Dispatch_Id'), 'ms=%.3f' % dur, 'VGPR', row.get('VGPR_Count'), 'accum', row.get('Accum_VGPR_Count'), 'SGPR', row.get('SGPR_Count'),
                      'LDS', row.get('LDS_Block_Size'), 'scratch', row.get('Scratch_Size'), 'grid', row.get('Grid_Size_X'), 'wg', row.get('Workgroup_Size_X'))
print('== counters (summed over megakernel dispatches) ==')
for kname in ('pt_megakernel', 'wf_trace', 'wf_shade', 'chunk_sum'):
    tot = collections.defaultdict(float); ndisp = collections.defaultdict(set)
    for f in find('*counter_collection.csv'):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kname in row.get('Kernel_Name', '') and ', true' not in row.get('Kernel_Name', '') and '<true' not in row.get('Kernel_Name', ''):
                    tot[row['Counter_Name']] += float(row['Counter_Value']); ndisp[row['Counter_Name']].add(row['Dispatch_Id'])
    if not tot: continue
    print('-- %s (non-counter variant) --' % kname)
    for k in sorted(tot):
        print('%-28s total=%.6g dispatches=%d' % (k, tot[k], len(ndisp[k])))
    g = lambda k: tot.get(k, 0.0)
    if g('SQ_ACTIVE_INST_VALU'):
        # (a fully active streaming kernel — chunk_sum_kernel — reads 1.000 in this formula)
        print('lane utilisation (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64)) = %.3f' % (
            g('SQ_THREAD_CYCLES_VALU') / (g('SQ_ACTIVE_INST_VALU') * 64)))
    if g('FETCH_SIZE') or g('WRITE_SIZE'):
        # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM)
        print('HBM traffic over these dispatches: read %.3f GB (FETCH_SIZE x2), write %.3f GB' % (
            g('FETCH_SIZE') * 2 * 1024 / 1e9, g('WRITE_SIZE') * 1024 / 1e9))
    if g('SQ_WAVE_CYCLES'):
        print('VALU active / wave cycles = %.3f   wait_any / wave cycles = %.3f   wait_inst_any / wave cycles = %.3f' % (
            g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'), g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'), g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES')))
