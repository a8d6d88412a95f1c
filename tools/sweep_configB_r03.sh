#!/bin/bash
# config B (two-level cycle at this size): coarse polynomial degree / interval, smoother interval, forced third level
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-sweepB}
mkdir -p $O
run() { name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --config B --steps 30 --warmup 6 > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; }
  [ -s $O/$name.json ] || return
  python - <<PY
import json
j=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name", "ms %.3f its %s rel %.2e"%(j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"],j["config"]["last_solve_rel_residual"]), flush=True)
PY
}
run default X=1
run kc6 TLFEA_PMG_KC=6
run kc8 TLFEA_PMG_KC=8
run kc10 TLFEA_PMG_KC=10
run kc16 TLFEA_PMG_KC=16
run kc8_kap60 TLFEA_PMG_KC=8 TLFEA_PMG_KAPPA_C=60
run kc8_kap30 TLFEA_PMG_KC=8 TLFEA_PMG_KAPPA_C=30
run kc6_kap30 TLFEA_PMG_KC=6 TLFEA_PMG_KAPPA_C=30
run kaps5 TLFEA_PMG_KAPPA_S=5
run kaps12 TLFEA_PMG_KAPPA_S=12
run ks3 TLFEA_PMG_KS=3 TLFEA_PMG_KAPPA_S=14
run lev3 TLFEA_PMG_LEVELS=3
run default2 X=1
