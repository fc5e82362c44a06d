#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python tools/repro_din.py f16 150 2>&1 | grep -v amdgpu.ids | cut -c1-1200
echo "== done"
