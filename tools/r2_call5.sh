#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_preprocess.py -m gpu -q -s > $O/test5.log 2>&1; echo "preprocess tests rc=$?"; tail -3 $O/test5.log
timeout -k 10 300 python tools/stage_probe.py > $O/stage_probe.txt 2>&1; echo "stage probe rc=$?"; cat $O/stage_probe.txt | grep -v amdgpu
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ham_trace -o ham -- python3 $R/tools/hamming_scan_bench.py --mode both --reps 3 > $O/prof_ham_trace.log 2>&1; echo "ham trace rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d $O/prof_ham_pmc -o ham -- python3 $R/tools/hamming_scan_bench.py --mode both --reps 1 > $O/prof_ham_pmc.log 2>&1; echo "ham pmc rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ham_nab -o ham -- python3 $R/tools/hamming_scan_bench.py --mode both --queries 24633 --rows 23929 --nbit 64 --classes 555 --reps 3 > $O/prof_ham_nab.log 2>&1; echo "ham nab trace rc=$?"
cd $R
find $O/prof_ham_trace $O/prof_ham_pmc $O/prof_ham_nab -name "*.csv" | head -20
for f in $(find $O/prof_ham_pmc -name "*counter_collection.csv" | head -1); do python tools/pmc_summary.py $f > $O/ham_pmc_summary.txt; done
cat $O/ham_pmc_summary.txt | head -70
