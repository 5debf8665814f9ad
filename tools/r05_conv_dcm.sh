#!/bin/bash
# Measurement: the decoder / context / pyramid convolutions of MaskFlownet-S at the bench batch, conv.dcm = 1 (conv_mfma_kernel's bf16 x 3
# form, round 4's default) against the plan (dc_mma_kernel<.., CONV> where it applies), then the whole network
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "matrix_core_deformable_kernel_error or conv_network_layers or conv_full_batch or conv_every" -s 2>&1 | grep -v "^$" | tail -12
echo "== conv.dcm=1 (conv_mfma_kernel) =="
timeout 300 python tools/conv_time.py 8 conv_dcm=1 2>&1 | grep -v amdgpu.ids | tail -24
echo "== plan =="
timeout 300 python tools/conv_time.py 8 2>&1 | grep -v amdgpu.ids | tail -24
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-epe --no-side-configs 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.items() if k in ('e2e','e2e_fp32','e2e_full','e2e_train')}, d.get('e2e'))"
