#!/bin/bash
# attention v2 + wave-private C = 24 block + ring dwconv for every launch >= 64 M elements: whole GPU suite + default bench
set -u
OUT=$PWD/gpurun_out/v15
mkdir -p "$OUT"
echo "== full GPU suite"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
echo "== bench (default)"
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"; tail -1 "$OUT/bench_bf16.err" | cut -c1-200; cut -c1-200 "$OUT/bench_bf16.json"
python -c "
import json; r=json.load(open('$OUT/bench_bf16.json')); k=r['kernels']
print({n:(round(x['ms_per_step'],3), round(x.get('TFLOP/s',0))) for n,x in k.items() if n.startswith(('attn','refiner_block','dwconv'))})
p=r['parity']; print(json.dumps(p['coarse_argmax'])); print(json.dumps(p['outputs_with_reference_coarse_match_injected'])[:600])"
echo "== done"
