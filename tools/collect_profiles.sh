#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace summaries and PMC traffic passes for bench.py.
# Raw CSVs land under gpurun_out/prof_$TAG/; tools/summarize_profiles.py turns them into profiles/$TAG_*.
set -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
POOL="python3 bench.py --workload pool --steps 400 --warmup 40 --regions 1 --no-cpu-baseline"
EMBED="python3 bench.py --steps 6 --warmup 2 --regions 1 --no-cpu-baseline --no-power-probe --no-companions"
echo "== kernel trace: pool"   && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pool_trace  -- $POOL  > $OUT/pool_trace.log 2>&1  || tail -5 $OUT/pool_trace.log
echo "== kernel trace: embed"  && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/embed_trace -- $EMBED > $OUT/embed_trace.log 2>&1 || tail -5 $OUT/embed_trace.log
echo "== pmc FETCH_SIZE: pool" && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pool_fetch -- $POOL > $OUT/pool_fetch.log 2>&1 || tail -5 $OUT/pool_fetch.log
echo "== pmc WRITE_SIZE: pool" && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pool_write -- $POOL > $OUT/pool_write.log 2>&1 || tail -5 $OUT/pool_write.log
echo "== pmc FETCH_SIZE: embed" && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/embed_fetch -- $EMBED > $OUT/embed_fetch.log 2>&1 || tail -5 $OUT/embed_fetch.log
echo "== pmc WRITE_SIZE: embed" && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/embed_write -- $EMBED > $OUT/embed_write.log 2>&1 || tail -5 $OUT/embed_write.log
TRAIN="python3 bench.py --workload train --steps 3 --warmup 2 --regions 1 --no-cpu-baseline --no-companions"
echo "== pmc MfmaUtil: embed" && rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/embed_mfma -- $EMBED > $OUT/embed_mfma.log 2>&1 || tail -5 $OUT/embed_mfma.log
echo "== pmc MfmaUtil: train" && rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/train_mfma -- $TRAIN > $OUT/train_mfma.log 2>&1 || tail -5 $OUT/train_mfma.log
echo "== kernel trace: train" && rocprofv3 --kernel-trace --output-format csv -d $OUT/train_trace -- $TRAIN > $OUT/train_trace.log 2>&1 || tail -5 $OUT/train_trace.log
f=$(find $OUT/train_trace -name "*kernel_trace.csv" | head -1) && python3 tools/train_step_mix.py $f $OUT/train_step_mix.md > /dev/null && python3 tools/train_step_seq.py $f $OUT/train_step_sequence.txt > /dev/null
echo "== power test (random vs all-zero data, same launches)" && timeout -k 10 200 python3 tools/_diag/conv_power_test.py > $OUT/conv_power_test.log 2>&1 || tail -3 $OUT/conv_power_test.log
echo "== wgrad forms" && timeout -k 10 200 python3 tools/_diag/wgrad_x8_ab.py > $OUT/wgrad_x8_ab.log 2>&1 || tail -3 $OUT/wgrad_x8_ab.log
echo "== conv tile stamps" && timeout -k 10 200 python3 tools/conv_stamps.py run > $OUT/conv_stamps.log 2>&1 || tail -3 $OUT/conv_stamps.log
find $OUT/train_trace -name "*.csv" -delete
# keep only what the summary needs (the kernel_trace of the embed run is large)
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
grep -h '"metric"' $OUT/*.log | cut -c1-400
