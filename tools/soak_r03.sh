#!/bin/bash
# long runs from HEAD (round 3): config C 300 Newton iterations (100 time steps), config D 90, config B 900 -- iteration counts and
# per-iteration times must stay flat, every linear solve must meet its tolerance (bench.py fails otherwise)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/soak3
for spec in "C 300" "D 90" "B 900"; do set -- $spec
  TLFEA_BENCH_VERBOSE=1 python bench.py --no-cpu-baseline --config $1 --steps $2 --warmup 3 > gpurun_out/soak3/$1.json 2> gpurun_out/soak3/$1.err; rc=$?
  python - <<PY
import json,re
j=json.loads(open("gpurun_out/soak3/$1.json").read().strip().splitlines()[-1])
t=open("gpurun_out/soak3/$1.err").read()
its=[int(v) for v in re.search(r"CG iterations: \[(.*?)\]", t).group(1).split(",")]
ms=[float(v) for v in re.search(r"per-iteration ms \(warm-up first\): \[(.*?)\]", t).group(1).split(",")][3:]
q=len(its)//4
print("config $1 rc=$rc: %d Newton iterations, %.3e element-updates/s, %.2f ms; CG iterations min %d max %d, quarters %s; ms min %.2f median %.2f max %.2f; last rel %.1e converged %s"
      % (len(its), j["value"], j["ms_per_step"], min(its), max(its), [round(sum(its[k*q:(k+1)*q])/q,1) for k in range(4)], min(ms), sorted(ms)[len(ms)//2], max(ms), j["config"]["last_solve_rel_residual"], j["config"]["last_solve_converged"]))
PY
done
