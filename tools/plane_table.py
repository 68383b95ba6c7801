#!/usr/bin/env python3
"""Where a decoder sub-batch spends its time, plane by plane, from a rocprofv3 kernel trace of bench.py
(tools/collect_profiles.sh leaves one at gpurun_out/prof_<tag>/trace/run_kernel_trace.csv):

    python tools/plane_table.py gpurun_out/prof_r4/trace/run_kernel_trace.csv

Takes the last decompression of the trace (its k_dec_header .. k_lift_xyz_inv), one hardware queue = one sub-batch,
and prints per kernel the summed duration and per plane (a plane starts at k_dec_count) the kernels' durations in
microseconds.  Under the profiler consecutive kernels of a queue follow each other without gaps, so a duration
includes the wait for the kernel before it; the sub-batches' overlap is not what it is without the profiler."""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "sperrhip" not in n:
        continue
    m = re.search(r"sperrhip::(?:\(anonymous namespace\)::)?(\w+)", n)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), m.group(1)))
rows.sort()
heads = [i for i, r in enumerate(rows) if r[3] == "k_dec_header"]
tails = [i for i, r in enumerate(rows) if r[3] == "k_lift_xyz_inv"]
if len(heads) < 2 or not tails:
    sys.exit("no decompression with two sub-batches in this trace")
s, e = heads[-2], tails[-1]
t0 = rows[s][0]
byq = collections.defaultdict(list)
for r in rows[s:e + 1]:
    byq[r[2]].append(r)
print("decompression %.2f ms; queues: %s" % ((rows[e][1] - t0) / 1e6,
      ", ".join("%d: %d kernels, ends at %.2f ms" % (q, len(l), (l[-1][1] - t0) / 1e6) for q, l in byq.items())))
l = byq[rows[s][2]]
agg = collections.defaultdict(lambda: [0, 0])
for r in l:
    agg[r[3]][0] += r[1] - r[0]
    agg[r[3]][1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-20s %8.3f ms %4d launches" % (k, v[0] / 1e6, v[1]))
planes, cur = [], None
for r in l:
    if r[3] == "k_dec_count":
        cur = collections.OrderedDict()
        planes.append(cur)
    if cur is not None:
        cur[r[3]] = cur.get(r[3], 0) + (r[1] - r[0]) / 1e3
names = ["k_dec_count", "k_lip_words", "k_lip_apply", "k_lip_deposit", "k_lis_l0", "k_lis_l1", "k_lis_hi", "k_lis_compact",
         "k_place_scan", "k_place_scatter", "k_leaf_apply", "k_ref_apply2"]
print("plane " + " ".join(n[2:].rjust(13) for n in names) + "     total (us; the last row includes what follows the planes)")
for i, pl in enumerate(planes):
    print("%5d " % i + " ".join("%13.0f" % pl.get(n, 0) for n in names) + " %9.0f" % sum(pl.values()))
