#!/bin/bash
# round-2 profiles: kernel trace of the single-stream bench, PMC traffic passes, MFMA busy; full GPU test suite
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/test8.log 2>&1; echo "full gpu suite rc=$?"; tail -3 $O/test8.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_enc_trace -o enc -- python3 $R/bench.py --streams 1 --no-roofline-pass --no-cpu-baseline --no-hamming-scan --steps 10 --warmup 3 > $O/prof_enc_trace.json 2> $O/prof_enc_trace.err; echo "enc trace rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_enc_fetch -o enc -- python3 $R/bench.py --streams 1 --no-roofline-pass --no-cpu-baseline --no-hamming-scan --steps 3 --warmup 1 > /dev/null 2> $O/prof_enc_fetch.err; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_enc_write -o enc -- python3 $R/bench.py --streams 1 --no-roofline-pass --no-cpu-baseline --no-hamming-scan --steps 3 --warmup 1 > /dev/null 2> $O/prof_enc_write.err; echo "write rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_enc_mfma -o enc -- python3 $R/bench.py --streams 1 --no-roofline-pass --no-cpu-baseline --no-hamming-scan --steps 3 --warmup 1 > /dev/null 2> $O/prof_enc_mfma.err; echo "mfma rc=$?"
cd $R
python tools/make_traffic_json.py $O/prof_enc_fetch/enc_counter_collection.csv $O/prof_enc_write/enc_counter_collection.csv $O/gemm_traffic.json; echo "traffic rc=$?"
python tools/pmc_summary.py $O/prof_enc_mfma/enc_counter_collection.csv > $O/enc_mfma_summary.txt; echo "mfma summary rc=$?"
head -12 $O/prof_enc_trace/enc_kernel_stats.csv | cut -c1-200
