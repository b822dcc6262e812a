"""device-busy time of the tail of a rocprofv3 --kernel-trace CSV: python tools/busy.py <kernel_trace.csv> <span_ms>  -> over the last span_ms: union of kernel intervals, sum of durations, top kernels"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
span = float(sys.argv[2]) * 1e6
end = max(int(r['End_Timestamp']) for r in rows); t0 = end - span
sel = [(max(int(r['Start_Timestamp']), t0), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows if int(r['End_Timestamp']) > t0]
busy = 0; cur_s = cur_e = None
for s, e, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in sel)
print("window %.2f ms: device busy %.2f ms, sum of kernel durations %.2f ms, %d launches" % (span / 1e6, busy / 1e6, tot / 1e6, len(sel)))
agg = collections.Counter()
for s, e, k in sel: agg[k.split('(')[0].replace('colibri::', '').replace('void ', '')[:50]] += e - s
for k, v in agg.most_common(14): print("  %8.2f ms  %s" % (v / 1e6, k))
