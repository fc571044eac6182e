#!/bin/bash
# Round-3 profiles on one MI355X box (run through gpurun from the repository root):
#   headline: rocprofv3 --kernel-trace --stats of bench.py, then separate --pmc FETCH_SIZE / WRITE_SIZE passes;
#   decode / joiner / CTC: kernel stats of tools/secondary.py legs.
# Raw output under gpurun_out/prof_r03/; tools/prof_summary.py condenses it into profiles/.
export TMPDIR=/tmp
O=gpurun_out/prof_r03
mkdir -p $O
CMD="bench.py --steps 5 --cpu-sample 0 --extra-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $CMD > $O/bench_kt.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $CMD > $O/bench_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $CMD > $O/bench_write.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec -- python3 tools/secondary.py greedy beam hotword > $O/secondary_dec.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/jc -- python3 tools/secondary.py joiner loss_block amp_block ctc > $O/secondary_jc.log 2>&1
echo "exit $?"
tail -2 $O/bench_kt.log | cut -c1-300
