#!/bin/bash
# Measurement session (round 5): dc_mma_kernel (kernels/deform_conv_mma.h) -- parity at the bench shapes, then its tilings per level
# against the exact kernel (dc_mma=0), back to back in a graph (tools/corr_ab.py ... deform)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r05_dcm}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16x3 and deform" -s > $O/parity.txt 2>&1
grep -E "max rel err|passed|failed|Error|error" $O/parity.txt | head -40
: > $O/ab.txt
timeout 300 python tools/corr_ab.py "dc_mma=0;;dc_mt=1,dc_pt=4,dc_nw=4;dc_mt=1,dc_pt=12,dc_nw=12;dc_mt=1,dc_pt=1,dc_nw=2" 2 cfg2 5 deform >> $O/ab.txt 2>&1
timeout 300 python tools/corr_ab.py "dc_mma=0;;dc_mt=2,dc_pt=3,dc_nw=12;dc_mt=2,dc_pt=3,dc_nw=3;dc_mt=1,dc_pt=4,dc_nw=4;dc_mt=1,dc_pt=1,dc_nw=4;dc_mt=2,dc_pt=1,dc_nw=4;dc_mt=1,dc_pt=3,dc_nw=6;dc_mt=1,dc_pt=12,dc_nw=12" 3 cfg2 5 deform >> $O/ab.txt 2>&1
timeout 300 python tools/corr_ab.py "dc_mma=0;;dc_mt=1,dc_pt=3,dc_nw=6;dc_mt=3,dc_pt=1,dc_nw=6;dc_mt=1,dc_pt=1,dc_nw=2;dc_mt=1,dc_pt=4,dc_nw=4;dc_mt=1,dc_pt=1,dc_nw=1" 4 cfg2 5 deform >> $O/ab.txt 2>&1
timeout 300 python tools/corr_ab.py "dc_mma=0;;dc_mt=1,dc_pt=1,dc_nw=8;dc_mt=1,dc_pt=1,dc_nw=4;dc_mt=2,dc_pt=1,dc_nw=4;dc_mt=1,dc_pt=3,dc_nw=6;dc_mt=1,dc_pt=4,dc_nw=4;dc_mt=1,dc_pt=1,dc_nw=2" 5 cfg2 5 deform >> $O/ab.txt 2>&1
grep "^deform" $O/ab.txt
timeout 300 python bench.py --no-cpu-baseline --no-epe --no-e2e --no-side-configs --steps 300 > $O/bench.log 2>&1
python - $O/bench.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('pass', d['value'], d['ms_per_step'], d.get('ops_in_graph_us'))
else:
    print(open(sys.argv[1]).read()[-2000:])
PY
