mkdir -p gpurun_out/s2
for rr in 1 0 1 0; do
  echo "RR=$rr"; TLFEA_AF_RR=$rr TLFEA_TUNE_QUICK=1 timeout -k 10 240 python3 tools/tune_assemble.py C 2>&1 | grep "assemble_affine" | tail -1
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hessian or assembl" 2>&1 | tail -2
