"""Per-queue spans of the LAST call's kernels in a rocprofv3 kernel trace:  python tools/trace_queues.py <kernel_trace.csv> [tail_ms]
For every queue: first start, last end, busy time, launches -- within the last `tail_ms` ms of the trace."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
end = max(int(r["End_Timestamp"]) for r in rows)
t0 = end - int(tail * 1e6)
q = collections.defaultdict(lambda: [None, 0, 0, 0, collections.Counter()])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0:
        continue
    v = q[r["Queue_Id"]]
    v[0] = s if v[0] is None else min(v[0], s)
    v[1] = max(v[1], e)
    v[2] += e - s
    v[3] += 1
    v[4][r["Kernel_Name"].split("(")[0][:40]] += e - s
for k, v in sorted(q.items(), key=lambda kv: kv[1][0]):
    top = ", ".join("%s %.1f" % (n, t / 1e6) for n, t in v[4].most_common(3))
    print("queue %s: %.1f .. %.1f ms  busy %.1f ms  %d launches  [%s]" %
          (k, (v[0] - t0) / 1e6, (v[1] - t0) / 1e6, v[2] / 1e6, v[3], top))
