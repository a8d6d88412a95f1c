#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lp4
python -m pytest tests -x -q -m gpu > gpurun_out/lp4/pytest.log 2>&1 || { tail -40 gpurun_out/lp4/pytest.log; exit 1; }
tail -2 gpurun_out/lp4/pytest.log
for g in 0 1; do
for rep in 1 2; do
TLFEA_GRAPH=$g python bench.py --no-cpu-baseline > gpurun_out/lp4/B_g${g}_$rep.json 2> gpurun_out/lp4/B_g$g.err || { tail -20 gpurun_out/lp4/B_g$g.err; exit 1; }
done
done
python bench.py --no-cpu-baseline --cheb-deg 16 > gpurun_out/lp4/B_g1_d16.json 2>/dev/null || exit 1
python bench.py --no-cpu-baseline --config C --steps 2 --warmup 1 --cheb-deg 12 > gpurun_out/lp4/C_d12.json 2> gpurun_out/lp4/C.err || { tail -20 gpurun_out/lp4/C.err; exit 1; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/lp4/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"ERR",e); continue
    ra=j["roofline_all"]
    print(f.split("/")[-1], "value %.3e ms %.2f its %s"%(j["value"],j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"]),
          {k:(v["avg_us"],v["frac"]) for k,v in ra.items() if k in("spmv","cheb_step")}, j["stage_ms_per_step"]["pcg"])
PY
