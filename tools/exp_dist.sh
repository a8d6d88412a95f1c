#!/bin/bash
# multi-rank rehearsal on ONE GPU (gloo host staging): CG iteration counts / collectives of the rank-local preconditioner
set -o pipefail
mkdir -p gpurun_out/dist
export TLFEA_BENCH_BACKEND=gloo OMP_NUM_THREADS=2 MASTER_ADDR=127.0.0.1
for n in 2 4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$n --master-addr 127.0.0.1 --master-port $((29510+n)) bench.py --gpus $n --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/dist/B_n$n.json 2> gpurun_out/dist/B_n$n.err || { tail -20 gpurun_out/dist/B_n$n.err; exit 1; }
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/dist/B_n$n.json") if l.startswith("{")][-1])
print("n=$n value %.3e ms %.2f outer its %s" % (j["value"], j["ms_per_step"], j["config"]["pcg_outer_iters_per_step"]))
PY
done
