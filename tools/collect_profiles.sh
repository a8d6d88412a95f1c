#!/bin/bash
# Round-1 profile collection on the GPU box (run through gpurun): kernel trace + two PMC passes + bench JSONs.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o run --output-format csv -- python3 bench.py --no-cpu-baseline > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 1; }
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o run --output-format csv -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --prewarm-s 0 > $O/pf.log 2>&1 || { tail -20 $O/pf.log; exit 1; }
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o run --output-format csv -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --prewarm-s 0 > $O/pw.log 2>&1 || { tail -20 $O/pw.log; exit 1; }
echo "pmc write done"
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/summarize_pmc.py "$F" "$W" > $O/pmc_summary.txt || exit 1
cp profiles/r01_configB_pmc_hbm.csv $O/
S=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$S" $O/r01_configB_kernel_stats.csv
# SyncedVBD sweep (config C): kernel trace of tools/vbd_bench.py + its JSON lines
rocprofv3 --kernel-trace --stats -d $O/kt_vbd -o run --output-format csv -- python3 tools/vbd_bench.py C > $O/kv.log 2>&1 || { tail -20 $O/kv.log; exit 1; }
SV=$(find $O/kt_vbd -name "*kernel_stats.csv" | head -1); cp "$SV" $O/r01_vbd_configC_kernel_stats.csv
rm -rf $O/kt_vbd
python3 tools/vbd_bench.py B C > $O/r01_vbd_sweep.json 2> $O/vbd.err || { tail -20 $O/vbd.err; exit 1; }
echo "vbd done"
# keep only the small summaries in gpurun_out (the raw traces are tens of MB)
rm -rf $O/kt $O/pmc_fetch $O/pmc_write
python3 bench.py > $O/r01_bench_configB.json 2> $O/bench_B.err || { tail -20 $O/bench_B.err; exit 1; }
python3 bench.py --config C --steps 4 --warmup 1 > $O/r01_bench_configC.json 2> $O/bench_C.err || { tail -20 $O/bench_C.err; exit 1; }
python3 bench.py --config D --steps 2 --warmup 1 --max-pcg 300 --no-cpu-baseline > $O/r01_bench_configD_ancf3443.json 2> $O/bench_D.err || { tail -20 $O/bench_D.err; exit 1; }
head -12 $O/r01_configB_kernel_stats.csv | cut -c1-200
