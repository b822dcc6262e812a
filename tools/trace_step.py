"""Prints the kernels of the LAST train() step of a rocprofv3 --kernel-trace CSV in launch order, with durations and the gaps between them.
usage: python tools/trace_step.py <kernel_trace.csv> [first-kernel-substring]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
first = sys.argv[2] if len(sys.argv) > 2 else "uni_onepass_kernel"
starts = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]
i0 = starts[-1]
step = rows[i0:]
end = next((k for k, r in enumerate(step) if 'export_len' in r['Kernel_Name']), len(step))
step = step[:end]
t0 = int(step[0]['Start_Timestamp']); prev_end = t0; total = 0.0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('colibri::', '').replace('void ', '')[:64]
    print('%9.1f us  +%6.1f gap  %8.1f us  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name))
    total += (e - s) / 1e3; prev_end = e
print('kernels %.1f us, span %.1f us' % (total, (prev_end - t0) / 1e3))
