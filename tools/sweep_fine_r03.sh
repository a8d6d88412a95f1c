#!/bin/bash
# fine-level smoother of the three-level cycle at config C (round 3): terms / interval
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-sweepF}
mkdir -p $O
run() { name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --config C --steps 6 --warmup 2 > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; return; }
  python - <<PY
import json
j=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name", "ms %.2f its %s rel %.2e"%(j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"],j["config"]["last_solve_rel_residual"]), flush=True)
PY
}
run default X=1
run fks1_k4 TLFEA_PMG_KS=1 TLFEA_PMG_KAPPA_S=4
run fks1_k8 TLFEA_PMG_KS=1 TLFEA_PMG_KAPPA_S=8
run fks2_k5 TLFEA_PMG_KAPPA_S=5
run fks2_k12 TLFEA_PMG_KAPPA_S=12
run fks2_sm4 TLFEA_PMG_SMOOTHER=4
run l3_kc3_28 TLFEA_PMG_KC3=28 TLFEA_PMG_KAPPA_C3=1200
run graphoff TLFEA_GRAPH=0
