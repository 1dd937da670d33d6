set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "rfactor or full_size or hinv or fasterquant or joint or many" > gpurun_out/r02_pytest24.log 2>&1 || { tail -40 gpurun_out/r02_pytest24.log; exit 1; }
tail -3 gpurun_out/r02_pytest24.log
timeout -k 10 300 python tools/solve_probe.py 4096x4096 12288x4096 22016x4096 4096x11008 --actorder
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/r02_bench_rform4.json 2>/dev/null
python -c "
import json; j=json.load(open('gpurun_out/r02_bench_rform4.json')); print(j['value'], j['ms_per_step'], j['phases'])"
