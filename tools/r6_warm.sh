#!/bin/bash
mkdir -p gpurun_out/r6warm
for w in ${WARMS:-16 32 48}; do
  ANNCHOR_STH_WARM=$w timeout 300 python tools/st_ab.py 1000000 h 2>&1 | grep -v "^annchor" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('warm', $w, [r['gemm_s'] for r in d['runs']], [r['fit_s'] for r in d['runs']], d['runs'][0]['tile_pairs'], round(d['recall'],5))"
done
