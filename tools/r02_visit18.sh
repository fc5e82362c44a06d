#!/bin/bash
# visit 18: conv64 microbench + kernel trace of the bench with conv64 on
cd /root/repo
mkdir -p gpurun_out/v18
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 300 python tools/bench_conv64.py > gpurun_out/v18/bench_conv64.log 2>&1
cat gpurun_out/v18/bench_conv64.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof18 -o b -- python /root/repo/bench.py --steps 5 --warmup 2 --streams 1 > /root/repo/gpurun_out/v18/bench_prof.json 2> /root/repo/gpurun_out/v18/bench_prof.err
f=$(find /tmp/prof18 -name "*kernel_stats.csv" | head -1)
cp "$f" /root/repo/gpurun_out/v18/kernel_stats.csv
head -25 /root/repo/gpurun_out/v18/kernel_stats.csv | cut -c1-200
