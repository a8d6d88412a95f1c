#!/bin/bash
# second sweep: finer around the optimum with the 12 x 12 scaling (config D), the ANCF beam (config A), and the plain
# polynomial on T10 (config M2 / C with --precond 1) for the defaults of the 3 x 3 form
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-sweepB12b}
mkdir -p $O
run() { name=$1; shift; cfg=$1; shift; args=$1; shift
  env "$@" python bench.py --no-cpu-baseline --config $cfg --steps 6 --warmup 2 $args > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; }
  if grep -q "Memory access fault" $O/$name.err; then echo "GPU fault in $name: stopping"; exit 9; fi
  [ -s $O/$name.json ] || return
  python - <<PY
import json
j=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name", "ms %.2f its %s rel %.2e"%(j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"],j["config"]["last_solve_rel_residual"]), flush=True)
PY
}
run D_on_d16_k100 D "--cheb-deg 16 --cheb-kappa 100" X=1
run D_on_d16_k150 D "--cheb-deg 16 --cheb-kappa 150" X=1
run D_on_d20_k300 D "--cheb-deg 20 --cheb-kappa 300" X=1
run D_on_d20_k200 D "--cheb-deg 20 --cheb-kappa 200" X=1
run D_on_d12_k100 D "--cheb-deg 12 --cheb-kappa 100" X=1
run D_on_d8_k60 D "--cheb-deg 8 --cheb-kappa 60" X=1
run D_off_d16_k200 D "--cheb-deg 16 --cheb-kappa 200" TLFEA_ANCF_BLOCK12=0
run D_off_d12_k200 D "--cheb-deg 12 --cheb-kappa 200" TLFEA_ANCF_BLOCK12=0
run A_off A "" TLFEA_ANCF_BLOCK12=0
run A_on_default A "" X=1
run A_on_d16_k200 A "--cheb-deg 16 --cheb-kappa 200" X=1
run M2_p1_default M2 "--precond 1" X=1
run M2_p1_d24_k800 M2 "--precond 1 --cheb-deg 24 --cheb-kappa 800" X=1
run M2_p1_d16_k400 M2 "--precond 1 --cheb-deg 16 --cheb-kappa 400" X=1
run M2_p1_d16_k200 M2 "--precond 1 --cheb-deg 16 --cheb-kappa 200" X=1
