#!/bin/bash
# round-3 sweep of the three-level cycle (TLFEA_PMG_LEVELS=3) at config C: vertex-level smoother terms / interval, level-3 degree
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-sweepP3}
mkdir -p $O
run() { name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --config C --steps 6 --warmup 2 > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; return; }
  python - <<PY
import json
j=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name", "ms %.2f its %s rel %.2e"%(j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"],j["config"]["last_solve_rel_residual"]), flush=True)
PY
}
run two_level TLFEA_PMG_LEVELS=2
run l3_ks2_k16 TLFEA_PMG_LEVELS=3
run l3_ks3_k16 TLFEA_PMG_LEVELS=3 TLFEA_PMG_KS2=3 TLFEA_PMG_KAPPA_S2=16
run l3_ks3_k30 TLFEA_PMG_LEVELS=3 TLFEA_PMG_KS2=3 TLFEA_PMG_KAPPA_S2=30
run l3_ks4_k30 TLFEA_PMG_LEVELS=3 TLFEA_PMG_KS2=4 TLFEA_PMG_KAPPA_S2=30
run l3_ks4_k50 TLFEA_PMG_LEVELS=3 TLFEA_PMG_KS2=4 TLFEA_PMG_KAPPA_S2=50
run l3_ks6_k60 TLFEA_PMG_LEVELS=3 TLFEA_PMG_KS2=6 TLFEA_PMG_KAPPA_S2=60
run l3_ks4_k30_kc3_48 TLFEA_PMG_LEVELS=3 TLFEA_PMG_KS2=4 TLFEA_PMG_KAPPA_S2=30 TLFEA_PMG_KC3=48 TLFEA_PMG_KAPPA_C3=3500
