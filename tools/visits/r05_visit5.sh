#!/bin/bash
# round 5, visit 5: which hipBLASLt kernels does the vendor yardstick land on (names encode macro tile / wave tile / stream-K /
# direct-to-LDS)?  rocprofv3 kernel trace of tools/bench_vendor_gemm.py; a measuring tool only.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/v5; mkdir -p "$OUT"
REPO=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o vg -- python "$REPO/tools/bench_vendor_gemm.py" > "$OUT/vendor.log" 2>&1
cd "$REPO"
cat "$OUT/vendor.log" | grep "TFLOP" | cut -c1-220
for f in $(find "$OUT/prof" -name "*kernel_stats.csv"); do cut -c1-400 "$f" | head -40; cp "$f" "$OUT/vendor_kernel_stats.csv"; done
find "$OUT/prof" -name "*kernel_trace.csv" -delete; find "$OUT/prof" -name "*agent_info.csv" -delete
echo "== done"
