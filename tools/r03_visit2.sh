#!/bin/bash
# Round-3 visit 2: the f16 library (operators + model gates), nearest mode, tied keypoints, multinomial order; then the whole suite.
set -u
OUT=$PWD/gpurun_out/v2
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== new tests first (reports even if the bounds need adjusting)"
timeout 900 python -m pytest tests/test_gpu_f16.py "tests/test_gpu_parity.py::test_h16_tiny_stagewise_vs_oracle" \
  "tests/test_gpu_parity.py::test_f16_full8_vs_reference_golden" "tests/test_gpu_parity.py::test_bf16_full8_vs_reference_golden" \
  tests/test_gpu_ops.py -k "f16 or h16 or bf16_full8 or nearest or ties or multinomial" -q 2>&1 | tail -60 > "$OUT/pytest_new.log"
tail -40 "$OUT/pytest_new.log"
cp gpurun_out/parity_report.json "$OUT/parity_report_new.json" 2>/dev/null
echo "== whole GPU suite"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -30 > "$OUT/pytest_gpu.log"; tail -12 "$OUT/pytest_gpu.log"
cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null
echo "== f16 bench (same workload, amp_dtype=torch.float16)"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype f16 > "$OUT/bench_f16.json" 2> "$OUT/bench_f16.err"; tail -2 "$OUT/bench_f16.err"; cut -c1-400 "$OUT/bench_f16.json"
echo "== done"
