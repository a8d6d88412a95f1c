#!/bin/bash
# experiment driver: low-precision Chebyshev preconditioner variants (run on the GPU box via gpurun)
set -o pipefail
mkdir -p gpurun_out/lp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "preconditioner_degrees or against_direct" > gpurun_out/lp/pytest.log 2>&1 || { tail -30 gpurun_out/lp/pytest.log; exit 1; }
tail -2 gpurun_out/lp/pytest.log
for bits in 64 32 16; do
  python bench.py --no-cpu-baseline --cheb-bits $bits > gpurun_out/lp/B_bits$bits.json 2> gpurun_out/lp/B_bits$bits.err || { tail -20 gpurun_out/lp/B_bits$bits.err; exit 1; }
done
for lanes in 8 32; do
  TLFEA_LP_LANES=$lanes python bench.py --no-cpu-baseline --cheb-bits 16 > gpurun_out/lp/B_bits16_l$lanes.json 2> gpurun_out/lp/B_bits16_l$lanes.err || exit 1
done
python bench.py --no-cpu-baseline --cheb-bits 16 --cheb-deg 16 > gpurun_out/lp/B_bits16_d16.json 2>/dev/null || exit 1
python bench.py --no-cpu-baseline --cheb-bits 16 --cheb-deg 24 > gpurun_out/lp/B_bits16_d24.json 2>/dev/null || exit 1
echo "B done"
python bench.py --no-cpu-baseline --config C --steps 2 --warmup 1 --cheb-deg 12 --cheb-bits 16 > gpurun_out/lp/C_d12_b16.json 2> gpurun_out/lp/C_d12_b16.err || { tail -20 gpurun_out/lp/C_d12_b16.err; exit 1; }
echo "C16 done"
python bench.py --no-cpu-baseline --config C --steps 2 --warmup 1 --cheb-deg 12 --cheb-bits 32 > gpurun_out/lp/C_d12_b32.json 2>/dev/null || exit 1
python bench.py --no-cpu-baseline --config C --steps 2 --warmup 1 --cheb-deg 16 --cheb-bits 16 > gpurun_out/lp/C_d16_b16.json 2>/dev/null || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/lp/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"ERR",e); continue
    ra=j["roofline_all"]
    print(f.split("/")[-1], "value %.3e ms %.2f its %s"%(j["value"],j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"]),
          {k:(v["avg_us"],v["frac"]) for k,v in ra.items() if k in("spmv","cheb_step")})
PY
