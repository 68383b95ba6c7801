"""Turns gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the committed summaries under
profiles/: rocprofv3 kernel statistics, per-kernel FETCH_SIZE / WRITE_SIZE sums and the corrected
HBM traffic per step that bench.py reports as roofline.traffic.

    python tools/summarize_profiles.py r1c r1 [steps of the profiled run, default 3]
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from sperr_amd.srchash import bench_path_hash, head_commit   # noqa: E402
src_tag, dst_tag = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_" + src_tag)
dst = os.path.join(ROOT, "profiles")


def short(name):
    n = name.replace("void ", "").replace("sperrhip::", "")
    n = n.split("(")[0]
    return n.replace("unsigned int", "uint32_t").replace("unsigned long", "uint64_t")


def pmc(which):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(os.path.join(src, which, "run_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if "sperrhip" not in r["Kernel_Name"]:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


fetch, write = pmc("fetch"), pmc("write")
# steps of the profiled run: tools/collect_profiles.sh runs `--steps 1 --warmup 1` plus the untimed step
# that records the kernel table = 3 (a third argument overrides; k_enc_finalize is launched once per
# part of a compress call, two parts by default)
steps = wsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = {}
for k in sorted(set(fetch) | set(write)):
    f_kb, w_kb = fetch[k][0] / steps, write[k][0] / wsteps
    rows[k] = {
        "launches_per_step": fetch[k][1] / steps,
        "fetch_KB_raw": round(f_kb, 1),
        "write_KB": round(w_kb, 1),
        # MI355X_MICROARCH.md (HBM section): both counters are in KB; on gfx950 FETCH_SIZE counts
        # half of a wide read, WRITE_SIZE is exact
        "hbm_bytes_per_step_corrected": (2 * f_kb + w_kb) * 1024,
        "hbm_bytes_per_step_uncorrected": (f_kb + w_kb) * 1024,
    }
for which, acc in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
    with open(os.path.join(dst, f"{dst_tag}_pmc_{which}_per_kernel_1024cube.csv"), "w") as f:
        f.write(f"kernel,launches_per_step,{which}_KB_per_step\n")
        for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
            f.write(f'"{k}",{n / steps:g},{v / steps:.1f}\n')
with open(os.path.join(dst, f"{dst_tag}_pmc_traffic.json"), "w") as f:
    json.dump({
        "workload": "1024^3 fp32, 64 x 256^3 chunks, bpp 2, one step (compress + decompress)",
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; corrected = "
                "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (HBM section); the x2 "
                "holds for wide coalesced reads and over-counts scattered ones",
        "steps_profiled": steps,
        # the sources the counters belong to (bench.py reports roofline.traffic only while they are unchanged).
        # SRC_HASH of the run that collected the counters, when the collector recorded it; else the tree's now
        "source_sha16": (open(os.path.join(src, "source_sha16.txt")).read().strip()
                         if os.path.exists(os.path.join(src, "source_sha16.txt")) else bench_path_hash()),
        "commit": head_commit(),
        "kernels": rows,
    }, f, indent=1)
shutil.copy(os.path.join(src, "trace", "run_kernel_stats.csv"),
            os.path.join(dst, f"{dst_tag}_rocprofv3_kernel_stats_1024cube.csv"))
shutil.copy(os.path.join(src, "trace", "run_agent_info.csv"),
            os.path.join(dst, f"{dst_tag}_rocprofv3_agent_info.csv"))
shutil.copy(os.path.join(src, "engine_events.csv"),
            os.path.join(dst, f"{dst_tag}_engine_events_1024cube.csv"))
with open(os.path.join(src, "bench.json")) as f:
    line = [l for l in f if l.startswith("{")][-1]
with open(os.path.join(dst, f"{dst_tag}_bench_1024cube.json"), "w") as f:
    json.dump(json.loads(line), f, indent=1)
# per kernel: counter bytes over the SUM of its launch durations (the kernel trace of the same
# workload) = the HBM rate it actually runs at; a bandwidth-shaped kernel far below the 8 TB/s peak
# is latency- or LDS-bound, whatever its share of the traffic
dur = {}
with open(os.path.join(src, "trace", "run_kernel_stats.csv")) as f:
    for r in csv.DictReader(f):
        if "sperrhip" in r["Name"]:
            dur[short(r["Name"])] = (float(r["TotalDurationNs"]) / steps, int(r["Calls"]) / steps)
with open(os.path.join(dst, f"{dst_tag}_kernel_rates.csv"), "w") as f:
    f.write("kernel,launches_per_step,sum_ms_per_step,hbm_GB_per_step,TBps,frac_of_8TBps\n")
    for k, (ns, calls) in sorted(dur.items(), key=lambda kv: -kv[1][0]):
        b = rows.get(k, {}).get("hbm_bytes_per_step_corrected")
        tb = (b / ns * 1e9 / 1e12) if (b and ns > 0) else None
        rows.setdefault(k, {})["sum_ms_per_step"] = round(ns / 1e6, 4)
        if tb is not None:
            rows[k]["TBps"] = round(tb, 3)
        f.write(f'"{k}",{calls:g},{ns / 1e6:.4f},{(b or 0) / 1e9:.3f},{"" if tb is None else f"{tb:.3f}"},'
                f'{"" if tb is None else f"{tb / 8.0:.4f}"}\n')
with open(os.path.join(dst, f"{dst_tag}_pmc_traffic.json")) as f:
    rec = json.load(f)
rec["kernels"] = rows
rec["total_hbm_GB_per_step"] = round(sum(v.get("hbm_bytes_per_step_corrected", 0) for v in rows.values()) / 1e9, 2)
rec["launches_per_step"] = round(sum(c for _, c in dur.values()), 1)
with open(os.path.join(dst, f"{dst_tag}_pmc_traffic.json"), "w") as f:
    json.dump(rec, f, indent=1)
print("total", rec["total_hbm_GB_per_step"], "GB/step,", rec["launches_per_step"], "launches/step")
top = sorted(rows.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_step_corrected", 0))[:8]
for k, v in top:
    print(f"{k:34s} {v['hbm_bytes_per_step_corrected'] / 1e9:8.2f} GB/step  {v.get('sum_ms_per_step', 0):8.3f} ms  {v.get('TBps', 0):6.3f} TB/s")
