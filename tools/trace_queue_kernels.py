"""Per queue of a rocprofv3 kernel trace: summed durations by kernel over the last `tail_ms` ms:
python tools/trace_queue_kernels.py <kernel_trace.csv> [tail_ms] [top]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 8
end = max(int(r["End_Timestamp"]) for r in rows)
t0 = end - int(tail * 1e6)
q = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0:
        continue
    n = r["Kernel_Name"].replace("void ", "").replace("sperrhip::", "").split("(")[0][:34]
    v = q[r["Queue_Id"]][n]
    v[0] += e - s
    v[1] += 1
for k in sorted(q):
    tot = sum(v[0] for v in q[k].values())
    print("queue %s: %.1f ms in %d launches" % (k, tot / 1e6, sum(v[1] for v in q[k].values())))
    for n, v in sorted(q[k].items(), key=lambda kv: -kv[1][0])[:top]:
        print("    %-36s %8.2f ms %5d" % (n, v[0] / 1e6, v[1]))
