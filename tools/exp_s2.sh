mkdir -p gpurun_out/s2
TLFEA_TUNE_QUICK=1 timeout -k 10 240 python3 tools/tune_assemble.py C > gpurun_out/s2/tune.log 2>&1; tail -1 gpurun_out/s2/tune.log
TLFEA_AF_TIMING=1 TLFEA_TUNE_QUICK=1 timeout -k 10 240 python3 tools/tune_assemble.py C 2>&1 | grep "assemble_affine timing" | tail -1
