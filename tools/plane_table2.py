#!/usr/bin/env python3
"""Per-plane kernel durations of the LAST decompression in a rocprofv3 kernel trace, for any number of sub-batches
(tools/plane_table.py wants exactly the bench's two):

    python tools/plane_table2.py <run_kernel_trace.csv> [queue-rank]

One hardware queue = one sub-batch; prints the sub-batch whose queue ends last (or the queue-rank-th).  A plane starts
at k_dec_count (or k_plane_small); durations in microseconds; `gap` = time between the end of a kernel and the start of
the next one on the same queue, summed per plane."""
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
if not heads:
    sys.exit("no decompression in this trace")
# the last decompression: the k_dec_header launches that lie within 2 ms of the last one
last = heads[-1]
first = last
for i in reversed(heads):
    if rows[last][0] - rows[i][0] < 2_000_000:
        first = i
enc_after = [i for i, r in enumerate(rows) if i > last and r[3] in ("k_enc_state_init", "k_stride_sums_rows")]
e = enc_after[0] if enc_after else len(rows)
sel = rows[first:e]
t0 = sel[0][0]
byq = collections.defaultdict(list)
for r in sel:
    byq[r[2]].append(r)
qs = sorted(byq.items(), key=lambda kv: -kv[1][-1][1])
print("decompression %.2f ms; queues: %s" % ((max(r[1] for r in sel) - t0) / 1e6,
      ", ".join("%d: %d kernels, %.2f..%.2f ms" % (q, len(l), (l[0][0] - t0) / 1e6, (l[-1][1] - t0) / 1e6) for q, l in qs)))
l = qs[int(sys.argv[2]) if len(sys.argv) > 2 else 0][1]
agg = collections.defaultdict(lambda: [0, 0])
for r in l:
    agg[r[3]][0] += r[1] - r[0]
    agg[r[3]][1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-20s %8.3f ms %4d launches" % (k, v[0] / 1e6, v[1]))
planes, cur, prev_end = [], None, None
for r in l:
    if r[3] in ("k_dec_count", "k_plane_small", "k_small_pre"):
        cur = collections.OrderedDict()
        cur["_start"] = (r[0] - t0) / 1e3
        planes.append(cur)
    if cur is not None:
        cur[r[3]] = cur.get(r[3], 0) + (r[1] - r[0]) / 1e3
        if prev_end is not None:
            cur["gap"] = cur.get("gap", 0) + max(0, r[0] - prev_end) / 1e3
    prev_end = r[1]
names = ["k_plane_small", "k_small_pre", "k_small_post", "k_dec_count", "k_lip_words", "k_lip_scan", "k_lip_apply", "k_lip_deposit", "k_lis_l0", "k_lis_l1", "k_lis_l2",
         "k_lis_hi", "k_lis_compact", "k_place_scan", "k_place_scatter", "k_leaf_apply", "k_ref_apply2", "k_ref_deposit", "gap"]
names = [n for n in names if any(n in pl for pl in planes)]
print("plane  start " + " ".join(n.replace("k_", "")[:9].rjust(9) for n in names) + "     total")
for i, pl in enumerate(planes):
    tot = sum(v for k, v in pl.items() if k != "_start")
    print("%5d %6.0f " % (i, pl["_start"]) + " ".join("%9.0f" % pl.get(n, 0) for n in names) + " %9.0f" % tot)
