#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hamming_gpu.py tests/test_surface_gpu.py -m gpu -q -x > $O/test11.log 2>&1; echo "hamming tests (VMEM default) rc=$?"; tail -2 $O/test11.log
CH_HAMMING_VMEM=0 timeout -k 10 600 python -m pytest tests/test_hamming_gpu.py -m gpu -q -x -k "map or nabirds or multi_limit or dataset" > $O/test11b.log 2>&1; echo "hamming tests (scalar) rc=$?"; tail -2 $O/test11b.log
for vm in 1 0; do
  for cfg in "5794 5994 64 200" "24633 23929 64 555" "16384 1000000 128 200"; do set -- $cfg
    CH_HAMMING_VMEM=$vm timeout -k 10 200 python tools/hamming_scan_bench.py --mode map --queries $1 --rows $2 --nbit $3 --classes $4 --reps 3 2>&1 | grep -v amdgpu | sed "s/^/[vmem=$vm] /"
  done
done
