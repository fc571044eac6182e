#!/bin/bash
# Round-2 profile of the headline command: kernel-trace stats, then the two PMC passes (separate runs, as
# MI355X_MICROARCH.md prescribes), then the greedy-search kernel stats.  Run on the GPU box through gpurun.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02b
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 5 --cpu-sample 0 > $O/bench_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/bench_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/greedy -- python3 tools/bench_extra.py greedy --steps 2 > $O/greedy.log 2>&1
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*_agent_info.csv" -delete
du -sh $O
