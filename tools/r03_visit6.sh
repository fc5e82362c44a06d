#!/bin/bash
# Round-3 visit 6: MFMA all-pairs local-correlation tiles (tests, microbench A/B), ring dwconv strip rule, suite, bench.
set -u
OUT=$PWD/gpurun_out/v6
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_f16.py -q -k "local_corr or dwconv" 2>&1 | tail -8
timeout 300 python tools/bench_local_corr.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_local_corr.log" | python -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print(r['warp'][:5], 'r',r['r'],'C',r['C'],r['hw'],r['dtype'], ' '.join(f\"{k}={v['ms']}ms/{v['algorithmic_GBs']}GB/s\" for k,v in r.items() if isinstance(v,dict)), 'diff',round(r['max_abs_diff_between_forms'],5))
"
echo "== GPU suite"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > "$OUT/pytest_gpu.log"; tail -6 "$OUT/pytest_gpu.log"
for m in 0 3; do
  ROMA_LC_MODE=$m timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-other-configs > "$OUT/bench_lc$m.json" 2> "$OUT/bench_lc$m.err"
  python - "$OUT/bench_lc$m.json" $m <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); k=r["kernels"]; kc=r["kernels_coherent"]["kernels"]
dw=[v for n,v in k.items() if n.startswith("dwconv")][0]
print("lc_mode",sys.argv[2],round(r["value"],2),"pairs/s",round(r["ms_per_step"],2),"ms  dwconv",round(dw["ms_per_step"],2))
print("   incoherent:",{n:(round(v["ms_per_step"],3),round(v["GB/s"])) for n,v in k.items() if n.startswith("local_corr")})
print("   coherent:  ",{n:(round(v["ms_per_step"],3),round(v["GB/s"])) for n,v in kc.items() if n.startswith("local_corr")})
PY
done
echo "== done"
