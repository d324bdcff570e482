#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_hamming_gpu.py tests/test_surface_gpu.py tests/test_parity_r2_gpu.py -m gpu -q -s > gpurun_out/r2/test3.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2/test3.log
grep -E "passed|failed" gpurun_out/r2/test3.log | tail -3
timeout -k 10 300 python tools/error_growth.py > gpurun_out/r2/error_growth.txt 2>&1; echo "growth rc=$?"
timeout -k 10 120 python tools/hamming_scan_bench.py --mode both --queries 5794 --rows 5994 --nbit 64 --classes 200 > gpurun_out/r2/ham_cub.txt 2>&1; echo "cub rc=$?"
timeout -k 10 120 python tools/hamming_scan_bench.py --mode both --queries 24633 --rows 23929 --nbit 64 --classes 555 > gpurun_out/r2/ham_nab.txt 2>&1; echo "nab rc=$?"
timeout -k 10 300 python tools/hamming_scan_bench.py --mode both --queries 16384 --rows 1000000 --nbit 128 --classes 200 --reps 3 > gpurun_out/r2/ham_1m.txt 2>&1; echo "1m rc=$?"
cat gpurun_out/r2/ham_cub.txt gpurun_out/r2/ham_nab.txt gpurun_out/r2/ham_1m.txt
