mkdir -p gpurun_out/s2
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py -x -q -m gpu > gpurun_out/s2/parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/s2/parity.log
TLFEA_TUNE_QUICK=1 timeout -k 10 240 python3 tools/tune_assemble.py C > gpurun_out/s2/tune.log 2>&1; tail -1 gpurun_out/s2/tune.log
