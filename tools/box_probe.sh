#!/bin/bash
# What a GPU box gives a job: memory (host and cgroup), CPUs, NUMA.  bash tools/box_probe.sh > gpurun_out/box.txt
echo "== meminfo"; grep -E "MemTotal|MemAvailable|MemFree|Unevictable|Mlocked|HugePages_Total" /proc/meminfo
echo "== cgroup v2"; for f in memory.max memory.high memory.current memory.swap.max cpu.max cpuset.cpus.effective cpuset.mems.effective; do
  for d in /sys/fs/cgroup /sys/fs/cgroup$(cut -d: -f3 /proc/self/cgroup | head -1); do [ -r $d/$f ] && echo "$d/$f: $(cat $d/$f)"; done; done
echo "== cgroup v1"; for f in memory/memory.limit_in_bytes memory/memory.usage_in_bytes; do [ -r /sys/fs/cgroup/$f ] && echo "$f: $(cat /sys/fs/cgroup/$f)"; done
echo "== self"; cat /proc/self/cgroup; grep -E "Cpus_allowed_list|Mems_allowed_list" /proc/self/status; nproc; ulimit -l -v -m 2>&1
echo "== numa"; ls /sys/devices/system/node/ 2>/dev/null | tr '\n' ' '; echo; for n in /sys/devices/system/node/node*; do echo "$n cpulist $(cat $n/cpulist) $(grep MemTotal $n/meminfo)"; done
lscpu | grep -E "Model name|Socket|NUMA|^CPU\(s\)"
echo "== gpu pci"; for d in /sys/bus/pci/devices/*; do if [ "$(cat $d/vendor 2>/dev/null)" = "0x1002" ] && [ "$(cat $d/class 2>/dev/null | cut -c1-4)" != "0x06" ]; then echo "$(basename $d) class $(cat $d/class) numa_node $(cat $d/numa_node)"; fi; done
df -h /dev/shm /tmp 2>/dev/null
