#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lp6
for cfg in "12 400" "16 800" "20 800" "24 1600" "32 1600"; do
set -- $cfg
python bench.py --no-cpu-baseline --config C --steps 2 --warmup 1 --cheb-deg $1 --cheb-kappa $2 > gpurun_out/lp6/C_d$1_k$2.json 2> gpurun_out/lp6/C.err || { tail -20 gpurun_out/lp6/C.err; exit 1; }
python bench.py --no-cpu-baseline --cheb-deg $1 --cheb-kappa $2 > gpurun_out/lp6/B_d$1_k$2.json 2> gpurun_out/lp6/B.err || { tail -20 gpurun_out/lp6/B.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/lp6/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"ERR",e); continue
    ra=j["roofline_all"]
    print(f.split("/")[-1], "value %.3e ms %.2f its %s"%(j["value"],j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"]),
          {k:(v["avg_us"],v["frac"]) for k,v in ra.items() if k in("spmv","cheb_step")}, j["stage_ms_per_step"]["pcg"])
PY
