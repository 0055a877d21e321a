set -e
mkdir -p gpurun_out/s2
CAT_SIM_LIB=build/var/gf.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_known_answers.py tests/test_gpu_rollout_resident.py tests/test_gpu_parity_full_size.py -x -q > gpurun_out/s2/parity_gf.txt 2>&1 || { tail -40 gpurun_out/s2/parity_gf.txt; exit 1; }
tail -2 gpurun_out/s2/parity_gf.txt
tools/ab_all.sh gpurun_out/s2/ab_gf.log build/var/base.so build/var/gf.so > /dev/null
cat gpurun_out/s2/ab_gf.log
