#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_preprocess.py tests/test_encode_gpu.py -m gpu -q -x > $O/test6.log 2>&1; echo "tests rc=$?"; tail -2 $O/test6.log
timeout -k 10 300 python tools/stage_probe.py > $O/stage_probe.txt 2>&1; echo "stage probe rc=$?"; grep -v amdgpu $O/stage_probe.txt
for rep in 1 2; do for sp in 1 0; do
  CH_SERPENTINE=$sp timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hamming-scan > $O/serp_${sp}_${rep}.json 2>> $O/serp.err || exit 1
done; done
python - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/r2/serp_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline_pass']['ms_per_step'], {k: v for k, v in d['kernel_ms_per_step'].items() if v > 0.5})
PY
