#!/bin/bash
# kernel-trace of bench.py at config D: per-kernel totals
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_D
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_D -o d -- python3 /root/repo/bench.py --no-cpu-baseline --config D --steps 6 --warmup 2 --prewarm-s 0 > /root/repo/gpurun_out/prof_D.json 2> /root/repo/gpurun_out/prof_D.err
python3 - <<PY
import sqlite3, json
j=json.loads(open('/root/repo/gpurun_out/prof_D.json').read().strip().splitlines()[-1])
print("ms_per_step", j["ms_per_step"], "its", j["config"]["pcg_outer_iters_per_step"])
c=sqlite3.connect('/root/repo/gpurun_out/prof_D/d_results.db')
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc limit 14"
for r in c.execute(q): print("%-70s n=%6d total %8.2f ms avg %8.1f us"%(r[0][:70],r[1],r[2],r[3]))
PY
