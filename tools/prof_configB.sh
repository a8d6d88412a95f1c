#!/bin/bash
# kernel-trace of bench.py at config B (launch-bound regime): per-kernel counts / averages and the busy fraction of the stream
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_B
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_B -o b -- python3 /root/repo/bench.py --no-cpu-baseline --config B --steps 30 --warmup 3 > /root/repo/gpurun_out/prof_B.json 2> /root/repo/gpurun_out/prof_B.err
python3 - <<PY
import sqlite3, json
j=json.loads(open('/root/repo/gpurun_out/prof_B.json').read().strip().splitlines()[-1])
print("ms_per_step", j["ms_per_step"], "its", j["config"]["pcg_outer_iters_per_step"])
c=sqlite3.connect('/root/repo/gpurun_out/prof_B/b_results.db')
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 2 desc limit 16"
for r in c.execute(q): print("%-64s n=%6d total %8.2f ms avg %6.1f us"%(r[0][:64],r[1],r[2],r[3]))
rows=list(c.execute(f"select start, end from {kd} order by start"))
busy=sum(e-s for s,e in rows); span=rows[-1][1]-rows[0][0]
import statistics
gaps=[rows[i+1][0]-rows[i][1] for i in range(len(rows)-1)]
gaps_small=[g for g in gaps if g<50000]
print("kernels", len(rows), "busy %.1f ms"%(busy/1e6), "span %.1f ms"%(span/1e6), "median gap %.2f us"%(statistics.median(gaps_small)/1e3), "mean small gap %.2f us"%(sum(gaps_small)/len(gaps_small)/1e3))
PY
