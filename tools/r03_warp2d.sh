#!/bin/bash
# same-box A/B of the warp kernel's lane -> pixel mapping (16 x 4 tiles per wave against 64 pixels of a row), smooth and rough flows
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "warp" 2>&1 | tail -2
for rep in 1 2; do
for v in 2d 1d; do
  if [ $v = 2d ]; then unset MFN_HIP_SO; else export MFN_HIP_SO=tools/ablate_build/libmfn_warp1d.so; fi
  for fl in smooth rough; do
    python bench.py --flow $fl --steps 1000 --no-e2e --no-epe --no-cpu-baseline --no-side-configs 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $fl', j['value'], j['ms_per_step'], 'warp us', j.get('ops_in_graph_us',{}).get('warp'), 'warp frac', (j.get('roofline_warp') or {}).get('frac'))"
  done
done; done
