#!/bin/bash
# round-3 solver experiments at config C: mixed-precision outer iteration (TLFEA_SPMV32 = replacement period, 0 = fp64
# SpMV), smoother degree / interval of the p-multigrid cycle.   tools/sweep_solver_r03.sh [tag] [name env... ;]
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-sweepS}
O=gpurun_out/$TAG
mkdir -p $O
run() { # name env...
  name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --config C --steps 6 --warmup 2 > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return; }
  python - <<PY
import json
j=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name", "value %.3e ms %.2f its %s rel %.2e pcg_ms %.2f"%(j["value"],j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"],j["config"]["last_solve_rel_residual"],j["stage_ms_per_step"]["pcg"]), flush=True)
PY
}
run base64 TLFEA_SPMV32=0
run s32_8 TLFEA_SPMV32=8
run s32_16 TLFEA_SPMV32=16
run s32_4 TLFEA_SPMV32=4
run ks1_k4 TLFEA_PMG_KS=1 TLFEA_PMG_KAPPA_S=4
run ks1_k8 TLFEA_PMG_KS=1 TLFEA_PMG_KAPPA_S=8
run ks3_k8 TLFEA_PMG_KS=3 TLFEA_PMG_KAPPA_S=8
run ks3_k16 TLFEA_PMG_KS=3 TLFEA_PMG_KAPPA_S=16
run ks3_k30 TLFEA_PMG_KS=3 TLFEA_PMG_KAPPA_S=30
run ks4_k30 TLFEA_PMG_KS=4 TLFEA_PMG_KAPPA_S=30
