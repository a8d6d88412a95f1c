#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lp2
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "preconditioner_degrees or against_direct or newton" > gpurun_out/lp2/pytest.log 2>&1 || { tail -30 gpurun_out/lp2/pytest.log; exit 1; }
tail -2 gpurun_out/lp2/pytest.log
for deg in 12 16; do
python bench.py --no-cpu-baseline --cheb-deg $deg > gpurun_out/lp2/B_d$deg.json 2> gpurun_out/lp2/B_d$deg.err || { tail -20 gpurun_out/lp2/B_d$deg.err; exit 1; }
done
TLFEA_LP_LANES=32 python bench.py --no-cpu-baseline > gpurun_out/lp2/B_l32.json 2>/dev/null || exit 1
python bench.py --no-cpu-baseline --config C --steps 2 --warmup 1 --cheb-deg 12 > gpurun_out/lp2/C_d12.json 2> gpurun_out/lp2/C_d12.err || { tail -20 gpurun_out/lp2/C_d12.err; exit 1; }
TLFEA_LP_LANES=32 python bench.py --no-cpu-baseline --config C --steps 2 --warmup 1 --cheb-deg 12 > gpurun_out/lp2/C_d12_l32.json 2>/dev/null || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/lp2/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"ERR",e); continue
    ra=j["roofline_all"]
    print(f.split("/")[-1], "value %.3e ms %.2f its %s"%(j["value"],j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"]),
          {k:(v["avg_us"],v["frac"]) for k,v in ra.items() if k in("spmv","cheb_step")}, j["stage_ms_per_step"])
PY
