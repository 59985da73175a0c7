#!/bin/bash
# Is the sporadic 10-ms-quantised stall of the host loop CPU-bandwidth throttling of the box's cgroup?  nproc, cpu.max, and
# cpu.stat (nr_throttled / throttled_usec) around the same 60-problem host loop with default and with one OpenMP thread.
echo "nproc $(nproc)  online $(cat /sys/devices/system/cpu/online)"
for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us /sys/fs/cgroup/cpu/cpu.stat; do
  [ -e $f ] && { echo "== $f"; cat $f; }
done
cat /proc/self/cgroup
grep -E "^(model name|cpu MHz)" /proc/cpuinfo | sort | uniq -c | head -4
grep -c steal /proc/stat; head -1 /proc/stat
stat() { cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo; head -1 /proc/stat; }
echo "--- default threads"; stat
python tools/diag/dropin_stalls_one.py 60
stat
echo "--- OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1"; 
OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 python tools/diag/dropin_stalls_one.py 60
stat
echo "--- OMP_WAIT_POLICY=passive GOMP_SPINCOUNT=0"
OMP_WAIT_POLICY=passive GOMP_SPINCOUNT=0 python tools/diag/dropin_stalls_one.py 60
stat
