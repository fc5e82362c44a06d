#!/bin/bash
set -u
export TMPDIR=/tmp
for args in "--others 1" "--others 0" "--others 1 --trace" "--others 0 --seed 7" "--others 1 --amp f16" "--others 1 --amp bf16"; do
  echo "== $args"
  timeout 600 python tools/repro_mixed.py $args 2>&1 | grep -v "amdgpu.ids" | cut -c1-1200
done
echo "== done"
