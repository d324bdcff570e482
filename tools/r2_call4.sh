#!/bin/bash
# round-2 GPU call 4: parity re-run, precision probe, new bench.py, Hamming profiles
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_parity_r2_gpu.py -m gpu -q -s -k "map_delta or full_depth or outlier or batch_256" > $O/test4.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed" $O/test4.log | tail -2
timeout -k 10 300 python tools/precision_probe.py > $O/precision_probe.txt 2>&1; echo "probe rc=$?"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench4.json 2> $O/bench4.err; echo "bench rc=$?"
tail -c 600 $O/bench4.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_ham_trace -o ham -- python3 $R/tools/hamming_scan_bench.py --mode both --reps 3 > $O/prof_ham_trace.log 2>&1; echo "ham trace rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -d $O/prof_ham_pmc -o ham -- python3 $R/tools/hamming_scan_bench.py --mode both --reps 1 > $O/prof_ham_pmc.log 2>&1; echo "ham pmc rc=$?"
cd $R
find $O/prof_ham_trace -name "*kernel_stats.csv" | head -2
for f in $(find $O/prof_ham_pmc -name "*counter_collection.csv" | head -1); do python tools/pmc_summary.py $f > $O/ham_pmc_summary.txt; done
cat $O/ham_pmc_summary.txt | head -60
