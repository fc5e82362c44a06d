#!/bin/bash
# Round-3 visit 1: GPU suite after the prune, then A/B of the attention XCD map and of the per-group refiner chains.
set -u
OUT=$PWD/gpurun_out/v1
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== GPU suite"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > "$OUT/pytest_gpu.log"; tail -6 "$OUT/pytest_gpu.log"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity"
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1]))
    k=r.get("kernels",{})
    att=sum(v["ms_per_step"] for n,v in k.items() if n.startswith("attn_"))
    dw=sum(v["ms_per_step"] for n,v in k.items() if n.startswith("dwconv"))
    rb=sum(v["ms_per_step"] for n,v in k.items() if n.startswith("refiner_block"))
    g6=sum(v["ms_per_step"] for n,v in k.items() if n.startswith("gemm6p"))
    print(f"{sys.argv[2]:28s} {r['value']:.2f} pairs/s  {r['ms_per_step']:.2f} ms  attn {att:.2f} dw {dw:.2f} rb {rb:.2f} gemm6p {g6:.2f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run base ROMA_ATTN_XCD=1 ROMA_REFINER_GROUP_MB=0
run attn_plain ROMA_ATTN_XCD=0 ROMA_REFINER_GROUP_MB=0
run grp100 ROMA_ATTN_XCD=1 ROMA_REFINER_GROUP_MB=100
run grp200 ROMA_ATTN_XCD=1 ROMA_REFINER_GROUP_MB=200
run grp400 ROMA_ATTN_XCD=1 ROMA_REFINER_GROUP_MB=400
run base2 ROMA_ATTN_XCD=1 ROMA_REFINER_GROUP_MB=0
echo "== done"
