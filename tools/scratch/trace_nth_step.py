"""kernels of the n-th (from the end, 1 = last) step of a rocprofv3 --kernel-trace CSV: python tools/scratch/trace_nth_step.py <csv> <n> [first-kernel-substring]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]); first = sys.argv[3] if len(sys.argv) > 3 else "uni_onepass_kernel"
starts = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]
print("steps in the trace:", len(starts), "spans (ms):", [round((int(rows[(starts[k + 1] - 1) if k + 1 < len(starts) else -1]['End_Timestamp']) - int(rows[starts[k]]['Start_Timestamp'])) / 1e6, 2) for k in range(len(starts))])
i0 = starts[-n]; i1 = starts[-n + 1] if n > 1 else len(rows)
step = rows[i0:i1]
t0 = int(step[0]['Start_Timestamp']); prev_end = t0; total = 0.0; gaps = 0.0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('colibri::', '').replace('void ', '')[:60]
    print('%9.1f us  +%6.1f gap  %8.1f us  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name))
    total += (e - s) / 1e3; gaps += max(0, (s - prev_end) / 1e3); prev_end = max(prev_end, e)
print('kernels %.1f us, span %.1f us, idle gaps %.1f us' % (total, (prev_end - t0) / 1e3, gaps))
