for d in lcg markov related; do for l in "" _r03; do
  SNACC_HIP_LIB=snacc_amd/libsnacc_hip$l.so timeout -k 10 200 python bench.py --data $d --no-cpu-baseline --no-cli-wall --no-matrix --no-secondary 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d lib$l', 2*j['value'], j['ms_per_step'])"
done; done
SNACC_HIP_LIB=$PWD/snacc_amd/libsnacc_hip_stats.so timeout -k 10 300 python3 tools/gpu_account.py 256 1000000 x 2>/dev/null | grep -A16 cycles_per_exit
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
