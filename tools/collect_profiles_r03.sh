#!/bin/bash
# Round-3 profile collection on the GPU box (run through gpurun): config C (the bench default) kernel trace, PMC traffic
# passes (one counter per pass, never together with trace domains), bench JSON lines for C (default), B and D.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/prof3
rm -rf $O; mkdir -p $O
BENCH="python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --prewarm-s 0"
rocprofv3 --kernel-trace --stats -d $O/kt -o run --output-format csv -- $BENCH > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 1; }
S=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$S" $O/r03_configC_kernel_stats.csv; rm -rf $O/kt
echo "kernel trace done"; head -12 $O/r03_configC_kernel_stats.csv | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_$c -o run --output-format csv -- $BENCH > $O/pmc_$c.log 2>&1 || { tail -20 $O/pmc_$c.log; exit 1; }
  echo "pmc $c done"
done
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/summarize_pmc.py "$F" "$W" $O/r03_configC_pmc_hbm.csv > $O/pmc_summary.txt || exit 1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python3 bench.py > $O/r03_bench_configC.json 2> $O/bench_C.err || { tail -20 $O/bench_C.err; exit 1; }
echo "bench C done"; cut -c1-400 $O/r03_bench_configC.json
python3 bench.py --config B > $O/r03_bench_configB.json 2> $O/bench_B.err || { tail -20 $O/bench_B.err; exit 1; }
echo "bench B done"; cut -c1-300 $O/r03_bench_configB.json
python3 bench.py --config D --max-pcg 400 > $O/r03_bench_configD_ancf3443.json 2> $O/bench_D.err || { tail -20 $O/bench_D.err; exit 1; }
echo "bench D done"; cut -c1-300 $O/r03_bench_configD_ancf3443.json
