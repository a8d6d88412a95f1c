#!/bin/bash
# ANCF node-block (12 x 12) scaling of the polynomial's operator at config D: on/off and polynomial degree / interval
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-sweepB12}
mkdir -p $O
run() { name=$1; shift; args=$1; shift
  env "$@" python bench.py --no-cpu-baseline --config D --steps 6 --warmup 2 $args > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; }
  if grep -q "Memory access fault" $O/$name.err; then echo "GPU fault in $name: stopping"; exit 9; fi
  [ -s $O/$name.json ] || return
  python - <<PY
import json
j=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name", "ms %.2f its %s rel %.2e"%(j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"],j["config"]["last_solve_rel_residual"]), j["stage_ms_per_step"].get("pcg"), flush=True)
PY
}
run off "" TLFEA_ANCF_BLOCK12=0
run on_default "" X=1
run on_d24_k800 "--cheb-deg 24 --cheb-kappa 800" X=1
run on_d24_k400 "--cheb-deg 24 --cheb-kappa 400" X=1
run on_d16_k400 "--cheb-deg 16 --cheb-kappa 400" X=1
run on_d16_k200 "--cheb-deg 16 --cheb-kappa 200" X=1
run on_d12_k200 "--cheb-deg 12 --cheb-kappa 200" X=1
run off_d24_k800 "--cheb-deg 24 --cheb-kappa 800" TLFEA_ANCF_BLOCK12=0
run off_d16_k400 "--cheb-deg 16 --cheb-kappa 400" TLFEA_ANCF_BLOCK12=0
