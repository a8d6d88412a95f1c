#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/microbench/pmc_calib.hip), one counter per pass.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/calib; rm -rf $O; mkdir -p $O
tools/microbench/pmc_calib > $O/plain.txt 2>&1 || { cat $O/plain.txt; exit 1; }
cat $O/plain.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c -d $O/$n -o run --output-format csv -- tools/microbench/pmc_calib > $O/$n.log 2>&1 || { tail -5 $O/$n.log; continue; }
  f=$(find $O/$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY' | tee -a $O/summary.csv
import collections, csv, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k},{c},{len(v)},{sum(v)/len(v):.1f}")
PY
  rm -rf $O/$n
done
