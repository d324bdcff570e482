#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/test_final.log 2>&1; echo "full gpu suite rc=$?"; tail -3 $O/test_final.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_final.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke_final.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r2/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline_pass'], d['roofline'])
print(d['decode_inclusive'])
print({k: v for k, v in d['hamming'].items() if k.startswith('map_eval')})
print(d['cpu_baseline'], d['cpu_baseline_hamming']['value'])
PY
