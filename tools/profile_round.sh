#!/bin/bash
# Reproduces the profiles/ evidence of a round on a 1-GPU MI355X box (run from the repo root, e.g. through gpurun):
#   bash tools/profile_round.sh r02
# rocprofv3 runs from /tmp (TMPDIR=/tmp); counters are collected in their own passes (no trace domains next to --pmc).
set -o pipefail
TAG=${1:-rXX}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
BENCH1="python3 $R/bench.py --streams 1 --no-roofline-pass --no-cpu-baseline --no-hamming-scan"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_trace -o enc -- $BENCH1 --steps 10 --warmup 3 > $O/enc_trace.json 2> $O/enc_trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/enc_fetch -o enc -- $BENCH1 --steps 3 --warmup 1 > /dev/null 2> $O/enc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/enc_write -o enc -- $BENCH1 --steps 3 --warmup 1 > /dev/null 2> $O/enc_write.err || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/enc_mfma -o enc -- $BENCH1 --steps 3 --warmup 1 > /dev/null 2> $O/enc_mfma.err || exit 1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/enc_lds -o enc -- $BENCH1 --steps 3 --warmup 1 > /dev/null 2> $O/enc_lds.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ham_trace -o ham -- python3 $R/tools/hamming_scan_bench.py --mode both --reps 3 > $O/ham_1m_timing.txt 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d $O/ham_pmc -o ham -- python3 $R/tools/hamming_scan_bench.py --mode both --reps 1 > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ham_nab -o ham -- python3 $R/tools/hamming_scan_bench.py --mode both --queries 24633 --rows 23929 --nbit 64 --classes 555 --reps 3 > $O/ham_nabirds_timing.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_trace -o tr -- python3 $R/tools/train_bench.py --batches 256 --steps 5 --warmup 2 > $O/train_trace.json 2> $O/train_trace.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_trace32 -o tr -- python3 $R/tools/train_bench.py --batches 32 --steps 20 --warmup 5 > $O/train_trace32.json 2> $O/train_trace32.err || exit 1
cd $R
python tools/train_bench.py > $O/train_bench.txt 2>&1
python tools/make_traffic_json.py $O/enc_fetch/enc_counter_collection.csv $O/enc_write/enc_counter_collection.csv $O/gemm_traffic.json
python tools/pmc_summary.py $O/enc_mfma/enc_counter_collection.csv > $O/enc_mfma_summary.txt
python tools/pmc_summary.py $O/enc_lds/enc_counter_collection.csv > $O/enc_lds_summary.txt
python tools/pmc_summary.py $O/ham_pmc/ham_counter_collection.csv > $O/ham_pmc_summary.txt
python tools/tile_timeline.py > $O/gemm_tile_timeline.txt 2>&1
python tools/precision_probe.py > $O/precision_probe.txt 2>&1
python tools/stage_probe.py > $O/stage_probe.txt 2>&1
python tools/error_growth.py > $O/error_growth.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
echo "summaries under $O: copy the ones to be judged into profiles/ (${TAG}_*)"
