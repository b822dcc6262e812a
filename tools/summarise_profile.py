"""Turn a gpurun_out/prof_<tag>/ directory (tools/profile_bench.sh) into the committed summaries under profiles/:
   profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (verbatim)
   profiles/<tag>_pmc_by_kernel.csv     FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes), KiB as reported
   profiles/pmc_count_kernel.json       HBM bytes per count-kernel launch, corrected as MI355X_MICROARCH.md §HBM prescribes
Usage: python tools/summarise_profile.py gpurun_out/prof_r01a r01 <tokens> <nbytes> <npos>
"""
import collections
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
tokens, nbytes, npos = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench_under_trace.json")):
    shutil.copy(os.path.join(src, "bench_under_trace.json"), os.path.join(out, f"{tag}_bench_under_trace.json"))

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("pmc_fetch", "pmc_write"):
    p = os.path.join(src, name, "bench_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, f"{tag}_pmc_by_kernel.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_sum", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_sum", "WRITE_SIZE_KiB_avg"])
    for k, d in sorted(agg.items(), key=lambda kv: -(sum(kv[1].get("FETCH_SIZE", [0])) + sum(kv[1].get("WRITE_SIZE", [0])))):
        fs, ws = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
        n = max(len(fs), len(ws), 1)
        w.writerow([k, n, round(sum(fs), 1), round(sum(fs) / n, 1), round(sum(ws), 1), round(sum(ws) / n, 1)])

fetch = sum(sum(d.get("FETCH_SIZE", [])) for k, d in agg.items() if "count_kernel" in k)
write = sum(sum(d.get("WRITE_SIZE", [])) for k, d in agg.items() if "count_kernel" in k)
launches = sum(len(d.get("FETCH_SIZE", [])) for k, d in agg.items() if "count_kernel" in k)
# calibration on known byte counts in this very run (MI355X_MICROARCH.md §HBM: FETCH_SIZE reports 1/2 of a wide coalesced
# streaming read on gfx950; other widths must be calibrated): clear_table writes exactly 16 B x cap; prune reads the same.
clear_w = sum(sum(d.get("WRITE_SIZE", [])) for k, d in agg.items() if "clear_table" in k)
prune_f = sum(sum(d.get("FETCH_SIZE", [])) for k, d in agg.items() if "prune_kernel" in k)
# streamed (coalesced) reads of one count launch: corpus bytes + token starts (+ two survivor-id reads for n > 1)
streamed_per_launch = nbytes + 4 * npos * (1 + 2 * 4 / 5)
corrected = (fetch + write) * 1024 / max(1, launches) + streamed_per_launch / 2
json.dump({
    "tokens": tokens, "launches_profiled": launches,
    "fetch_size_kib_per_launch": round(fetch / max(1, launches), 1), "write_size_kib_per_launch": round(write / max(1, launches), 1),
    "hbm_bytes_per_launch_raw": round((fetch + write) * 1024 / max(1, launches)),
    "hbm_bytes_per_launch": round(corrected),
    "correction": "raw = (FETCH_SIZE + WRITE_SIZE) x 1024, separate --pmc passes; + 1/2 of the launch's coalesced 16 B/lane streamed reads "
                  "(gfx950 FETCH_SIZE counts wide streaming reads at half, MI355X_MICROARCH.md §HBM). In-run calibration: clear_table "
                  f"WRITE_SIZE total {clear_w:.3e} KiB for a known 16 B x cap; prune FETCH_SIZE total {prune_f:.3e} KiB reading the same bytes "
                  f"(ratio {prune_f / max(1.0, clear_w):.2f} ~ 0.5 confirms the half-counting of streamed reads).",
    "source": f"profiles/{tag}_pmc_by_kernel.csv",
}, open(os.path.join(out, "pmc_count_kernel.json"), "w"), indent=1)
print(open(os.path.join(out, "pmc_count_kernel.json")).read())
