#!/bin/bash
# Round-2 GPU visit 2: per-tile overhead of the GEMM loops, local correlation after the classifier split, stream-split
# determinism matrix, fixed co-run stress, new operator tests.
set -u
OUT=$PWD/gpurun_out/v2
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== new op tests + parity tiny"
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -q -k "local_corr or refiner_input or gp_posterior or tiny_stagewise" 2>&1 | tail -15 > "$OUT/pytest_new.log"; tail -5 "$OUT/pytest_new.log"
echo "== gemm overhead"
timeout 300 python tools/bench_gemm_overhead.py > "$OUT/bench_gemm_overhead.log" 2>&1; cat "$OUT/bench_gemm_overhead.log" | cut -c1-700
echo "== local corr regimes"
timeout 200 python tools/bench_local_corr.py > "$OUT/bench_local_corr.log" 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/v2/bench_local_corr.log"):
    if l.startswith("{"):
        r = json.loads(l)
        print(f"r={r['r']} {r['hw']} {r['dtype']} {r['warp']:10s} tiled {r['tiled']['ms']:.3f} ms {r['tiled']['algorithmic_GBs']:7.0f} GB/s | per-pixel {r['per_pixel']['ms']:.3f} ms {r['per_pixel']['algorithmic_GBs']:7.0f} GB/s")
PY
echo "== stream split determinism matrix"
S="timeout 150 python tools/stress_streams.py --runs 300"
$S > "$OUT/streams_default.log" 2>&1; tail -1 "$OUT/streams_default.log" | cut -c1-600
$S --fuse 0 > "$OUT/streams_unfused.log" 2>&1; tail -1 "$OUT/streams_unfused.log" | cut -c1-600
ROMA_GEMM8P=0 $S --fuse 0 > "$OUT/streams_unfused_classic.log" 2>&1; tail -1 "$OUT/streams_unfused_classic.log" | cut -c1-600
ROMA_LC_MODE=2 $S --fuse 0 > "$OUT/streams_unfused_lc2.log" 2>&1; tail -1 "$OUT/streams_unfused_lc2.log" | cut -c1-600
ROMA_STREAMS_SERIAL=1 $S --fuse 0 > "$OUT/streams_unfused_serial.log" 2>&1; tail -1 "$OUT/streams_unfused_serial.log" | cut -c1-600
$S --fuse 0 --amp f32 --runs 100 > "$OUT/streams_unfused_f32.log" 2>&1; tail -1 "$OUT/streams_unfused_f32.log" | cut -c1-600
echo "== co-run stress (fixed allocation lifetime)"
timeout 300 python tools/corun_stress.py 8 > "$OUT/corun_stress.log" 2>&1; grep -v "   0/" "$OUT/corun_stress.log" | tail -30; grep -c "   0/" "$OUT/corun_stress.log"
echo "== done"
