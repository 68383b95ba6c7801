#!/bin/bash
# GPU box: kernel trace of the ragged volume's decompression (1000^3 in 256^3 chunks), per queue
set -u
out=gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o run -- python3 tools/ragged_1000.py > $out/run.log 2>&1
echo "rc=$?"; tail -2 $out/run.log
python3 - <<PY
import csv,re,collections
rows=[]
for r in csv.DictReader(open("$out/trace/run_kernel_trace.csv")):
    n=r["Kernel_Name"]
    m=re.search(r"sperrhip::(?:\(anonymous namespace\)::)?(\w+)",n)
    if not m: continue
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),int(r["Queue_Id"]),m.group(1)))
rows.sort()
heads=[i for i,r in enumerate(rows) if r[3]=="k_dec_header"]
# the last decompression: headers within 20 ms of the last
last=heads[-1]; first=last
for i in reversed(heads):
    if rows[last][0]-rows[i][0] < 20_000_000: first=i
enc=[i for i,r in enumerate(rows) if i>last and r[3] in ("k_enc_state_init","k_stride_sums_rows","k_stride_sums")]
e=enc[0] if enc else len(rows)
sel=rows[first:e]; t0=sel[0][0]
print("decompression %.2f ms"%((max(r[1] for r in sel)-t0)/1e6))
byq=collections.defaultdict(list)
for r in sel: byq[r[2]].append(r)
for q,l in sorted(byq.items(), key=lambda kv:-kv[1][-1][1]):
    agg=collections.defaultdict(lambda:[0,0])
    for r in l: agg[r[3]][0]+=r[1]-r[0]; agg[r[3]][1]+=1
    top=sorted(agg.items(), key=lambda kv:-kv[1][0])[:6]
    print("queue %d: %d kernels %.2f..%.2f ms busy %.2f | "%(q,len(l),(l[0][0]-t0)/1e6,(l[-1][1]-t0)/1e6,sum(r[1]-r[0] for r in l)/1e6)+", ".join("%s %.2f (%d)"%(k[2:],v[0]/1e6,v[1]) for k,v in top))
PY
