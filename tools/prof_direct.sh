#!/bin/bash
# kernel-trace of the sparse direct solve (tools/direct_timing.py <which>): per-kernel totals of the run
cd /tmp && export TMPDIR=/tmp
W=${1:-only-b}
rm -rf /root/repo/gpurun_out/prof_direct_$W
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_direct_$W -o d -- python3 /root/repo/tools/direct_timing.py $W > /root/repo/gpurun_out/direct_prof_$W.log 2>&1
grep -E "DOF" /root/repo/gpurun_out/direct_prof_$W.log
python3 - <<PY
import sqlite3
c=sqlite3.connect('/root/repo/gpurun_out/prof_direct_$W/d_results.db')
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%mf_%' or s.kernel_name like '%fillBuffer%' group by s.kernel_name order by 3 desc limit 14"
for r in c.execute(q): print("%-40s n=%6d total %9.2f ms avg %8.1f us"%(r[0][:60].replace('_ZN5tlfea12_GLOBAL__N_1','')[:40],r[1],r[2],r[3]))
PY
