#!/bin/bash
# round-2 GPU call 1: parity suite, then the micro-batch stream sweep
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2/test1.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2/test1.log
for s in 1 2 3 4 1 2 3 4; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --streams $s --no-cpu-baseline --no-hamming-scan > gpurun_out/r2/sweep_s${s}_$RANDOM.json 2>> gpurun_out/r2/sweep.err || exit 1
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/r2/sweep_s*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['config']['hip_streams'], d['value'], d['ms_per_step'])
    except Exception as e:
        print(f, 'ERR', e)
PY
