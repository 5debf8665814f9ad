#!/bin/bash
# Measurement session: the deformable convolution's bf16 x 3 form (dc.mma=1) against the exact fp32 kernel, levels 5..2, back to
# back in a graph (tools/corr_ab.py ... deform), after its parity / determinism tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_dcmma}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16x3 or deform" > $O/parity.txt 2>&1
tail -3 $O/parity.txt
: > $O/ab.txt
for lvl in 5 4 3 2; do
timeout 300 python tools/corr_ab.py ";dc_mma=1" $lvl cfg2 5 deform >> $O/ab.txt 2>&1
done
grep "^deform" $O/ab.txt
timeout 300 python bench.py --no-cpu-baseline --no-epe --no-e2e --no-side-configs --tuning dc_mma=1 --steps 300 > $O/bench_mma.log 2>&1
python - <<'PY'
import json,sys
l=[x for x in open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/r04_dcmma/bench_mma.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('dc.mma=1 pass', d['value'], d['ms_per_step'], d.get('ops_in_graph_us'))
PY
