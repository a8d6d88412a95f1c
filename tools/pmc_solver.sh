#!/bin/bash
# PMC passes over the solver's two streaming kernels (fine / coarse polynomial step, fp64 SpMV): L2 hit rate, issue mix.
#   tools/pmc_solver.sh <config> <tag>      (rocprofv3 --pmc only, program directly after --)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
CFG=${1:-C}; TAG=${2:-pmcS}
O=gpurun_out/$TAG
rm -rf $O; mkdir -p $O
KR='cheb32_kernel|spmv_dir_dot'
run() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-include-regex "$KR" -d $O/$name -o run --output-format csv -- python3 tools/prof_elem.py $CFG 3 > $O/$name.log 2>&1 || { tail -20 $O/$name.log; return 1; }
  local f=$(find $O/$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" > $O/$name.csv <<'PY'
import collections, csv, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[(r["Kernel_Name"].split("(")[0][:70] + " grid " + r.get("Grid_Size", "?"), r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel,counter,launches,mean")
for (k, c), v in sorted(agg.items()):
    print(f"{k},{c},{len(v)},{sum(v)/len(v):.1f}")
PY
  rm -rf $O/$name
  cat $O/$name.csv
}
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU || exit 1
