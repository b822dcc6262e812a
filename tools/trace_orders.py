"""print per-order kernel durations of the last train() in a rocprofv3 kernel-trace CSV (tools/profile helpers)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
names = ['bin_emit', 'bin_scatter_kernel<false>', 'bin_hist2', 'bin_scatter_kernel<true>', 'bin_count', 'compact_results', 'bin_resolve', 'count_kernel', 'prune_kernel', 'resolve_kernel', 'clear_table']
seq = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows]
per = collections.defaultdict(list)
for k, us in seq:
    for n in names:
        if n in k:
            per[n].append(us)
for n in names:
    if per[n]:
        last = per[n][-5:]
        print("%-28s %s  sum=%.0f us" % (n, ' '.join('%7.0f' % x for x in last), sum(last)))
