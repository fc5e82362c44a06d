#!/bin/bash
# round 5, visit 14: slab-major K order of the VGG layers with Cout >= 256 (the nine taps of a 64-channel slab back to back):
# parity, A/B of the whole step and of the conv kernel, and the FETCH_SIZE pass for its L2-miss traffic per launch
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/v14; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py -q -x 2>&1 | tail -3 | tee "$OUT/tests.log"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "conv3x3 or gemm" 2>&1 | tail -2 | tee -a "$OUT/tests.log"
for ko in 1 0 1 0; do
ROMA_CONV_KORDER=$ko timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('korder=$ko pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3),{n:(round(v['ms_per_step'],3),round(v.get('TFLOP/s',0),1)) for n,v in r['kernels'].items() if 'conv3x3,relu' in n})" | tee -a "$OUT/bench.log"
done
for ko in 1 0; do
cd /tmp
ROMA_CONV_KORDER=$ko ROMA_STREAMS=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_k$ko" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-other-configs > "$OUT/pmc_k$ko.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python - <<PY | tee -a "$OUT/fetch.log"
import csv,glob,collections
tot=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob("$OUT/pmc_k$ko/*counter_collection.csv"):
    seen=set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]!="FETCH_SIZE": continue
        name=r["Kernel_Name"].split("(")[0]
        tot[name]+=float(r["Counter_Value"]); 
        if (r["Dispatch_Id"],name) not in seen: seen.add((r["Dispatch_Id"],name)); n[name]+=1
for k,v in sorted(tot.items(),key=lambda kv:-kv[1])[:6]:
    print("korder=$ko %-70s launches %3d FETCH_SIZE raw per launch %.1f MB (KB units x 1024; x2 on gfx950 per the guide)"%(k[:70],n[k],v*1024/n[k]/1e6))
PY
rm -rf "$OUT/pmc_k$ko"
done
echo "== done"
