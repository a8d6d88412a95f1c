#!/bin/bash
# lagged preconditioner (TLFEA_PC_LAG = solves one set-up serves at most): ms per Newton iteration, CG iterations
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-sweepLag}
mkdir -p $O
run() { name=$1; cfg=$2; shift; shift
  env "$@" python bench.py --no-cpu-baseline --config $cfg --steps 12 --warmup 3 > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; }
  if grep -q "Memory access fault" $O/$name.err; then echo "GPU fault in $name: stopping"; exit 9; fi
  [ -s $O/$name.json ] || return
  python - <<PY
import json
j=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name", "ms %.2f its %s rel %.2e"%(j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"],j["config"]["last_solve_rel_residual"]), flush=True)
PY
}
for lag in 1 2 3 6 12; do run C_lag$lag C TLFEA_PC_LAG=$lag; done
for lag in 1 3 12; do run B_lag$lag B TLFEA_PC_LAG=$lag; done
for lag in 1 3 12; do run M2_lag$lag M2 TLFEA_PC_LAG=$lag; done
