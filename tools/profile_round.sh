#!/bin/bash
# Reproduces the profiles/ evidence of a round on a 1-GPU MI355X box (run from the repo root, e.g. through gpurun):
#   bash tools/profile_round.sh r03 [enc|ham|train|probes|all]
# rocprofv3 runs from /tmp (TMPDIR=/tmp); counters are collected in their own passes (no trace domains next to --pmc).
# ENCODER passes trace `bench.py --encode-only --streams 1`: nothing but the encode + retrieve steps runs in the traced process
# (no training step, no pcie / decode / evaluator blocks), one launch chain, so every GEMM row of a summary is one shape of the
# encoder.  The training step is profiled only through tools/train_bench.py.
set -o pipefail
TAG=${1:-rXX}
WHAT=${2:-all}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
ENC="python3 $R/bench.py --streams 1 --encode-only"
cd /tmp
if [ "$WHAT" = all ] || [ "$WHAT" = enc ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_trace -o enc -- $ENC --steps 10 --warmup 3 > $O/enc_trace.json 2> $O/enc_trace.err || exit 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/enc_fetch -o enc -- $ENC --no-roofline-pass --steps 3 --warmup 1 > /dev/null 2> $O/enc_fetch.err || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/enc_write -o enc -- $ENC --no-roofline-pass --steps 3 --warmup 1 > /dev/null 2> $O/enc_write.err || exit 1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/enc_mfma -o enc -- $ENC --no-roofline-pass --steps 3 --warmup 1 > /dev/null 2> $O/enc_mfma.err || exit 1
  ( cd $R
    python tools/kernel_stats_by_shape.py $O/enc_trace/enc_kernel_trace.csv --bench $O/enc_trace.json > $O/enc_kernel_stats_by_shape.csv || echo "AGREEMENT CHECK FAILED" >> $O/enc_kernel_stats_by_shape.csv
    python tools/make_traffic_json.py $O/enc_fetch/enc_counter_collection.csv $O/enc_write/enc_counter_collection.csv $O/gemm_traffic.json $O/enc_trace.json > $O/gemm_traffic.log 2>&1 || echo "TRAFFIC REFUSED (see gemm_traffic.log)"
    python tools/pmc_summary.py $O/enc_mfma/enc_counter_collection.csv > $O/enc_mfma_summary.txt )
fi
if [ "$WHAT" = all ] || [ "$WHAT" = ham ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ham_trace -o ham -- python3 $R/tools/hamming_scan_bench.py --mode both --reps 3 > $O/ham_1m_timing.txt 2>&1 || exit 1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d $O/ham_pmc -o ham -- python3 $R/tools/hamming_scan_bench.py --mode both --reps 1 > /dev/null 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ham_nab -o ham -- python3 $R/tools/hamming_scan_bench.py --mode both --queries 24633 --rows 23929 --nbit 64 --classes 555 --reps 3 > $O/ham_nabirds_timing.txt 2>&1 || exit 1
  ( cd $R; python tools/pmc_summary.py $O/ham_pmc/ham_counter_collection.csv > $O/ham_pmc_summary.txt )
fi
if [ "$WHAT" = all ] || [ "$WHAT" = train ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_trace -o tr -- python3 $R/tools/train_bench.py --batches 256 --steps 5 --warmup 2 > $O/train_trace.json 2> $O/train_trace.err || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_trace32 -o tr -- python3 $R/tools/train_bench.py --batches 32 --steps 20 --warmup 5 > $O/train_trace32.json 2> $O/train_trace32.err || exit 1
  ( cd $R
    python tools/kernel_stats_by_shape.py $O/train_trace/tr_kernel_trace.csv > $O/train_kernel_stats_by_shape_b256.csv
    python tools/kernel_stats_by_shape.py $O/train_trace32/tr_kernel_trace.csv > $O/train_kernel_stats_by_shape_b32.csv
    python tools/train_bench.py > $O/train_bench.txt 2>&1 )
fi
cd $R
if [ "$WHAT" = all ] || [ "$WHAT" = probes ]; then
  python tools/tile_timeline.py > $O/gemm_tile_timeline.txt 2>&1
  python tools/precision_probe.py > $O/precision_probe.txt 2>&1
  python tools/stage_probe.py > $O/stage_probe.txt 2>&1
  python tools/error_growth.py > $O/error_growth.txt 2>&1
fi
if [ "$WHAT" = all ]; then
  python bench.py > $O/bench.json 2> $O/bench.err
fi
echo "summaries under $O: copy the ones to be judged into profiles/ (${TAG}_*)"
