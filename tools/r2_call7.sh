#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x > $O/test7.log 2>&1; echo "gemm tests rc=$?"; tail -3 $O/test7.log
timeout -k 10 300 python tools/gemm_bench.py --variants 2,4 --rounds 12 --shapes qkv,out,fc1,fc2 > $O/gemm_bench_sched.txt 2>&1; echo "bench rc=$?"; grep -v amdgpu $O/gemm_bench_sched.txt
timeout -k 10 300 python tools/gemm_bench.py --variants 2,4 --rounds 12 --shapes qkv,fc1 --epi 8 >> $O/gemm_bench_sched.txt 2>&1; tail -2 $O/gemm_bench_sched.txt
for rep in 1 2; do for sc in 1 0; do
  CH_GEMM_PP_SCHED=$sc timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hamming-scan > $O/sched_${sc}_${rep}.json 2>> $O/sched.err || exit 1
done; done
python - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/r2/sched_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline_pass']['ms_per_step'], {k: v for k, v in d['kernel_ms_per_step'].items() if v > 0.5})
PY
timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2rank_selflaunch.json 2> $O/bench_2rank_selflaunch.err; echo "2-rank self-launch rc=$?"; tail -c 400 $O/bench_2rank_selflaunch.json
