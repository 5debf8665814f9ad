#!/bin/bash
# Measurement (round 6, profiles/r06_dc_bwd_input_planes.txt): dc_bwd_input_pix_kernel with wave-private planes (the library) against the region's shared
# planes of rounds 2-5 (tools/ablate_build/libmfn_wp0.so: a build of the sources of commit 3cc5511, or -DMFN_DCP_WP=0 while both forms existed; skipped if absent):
# per-level durations, interleaved, then the phases, then the GPU parity tests of the deformable backward
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for i in 1 2; do
  [ -f tools/ablate_build/libmfn_wp0.so ] && MFN_HIP_SO=tools/ablate_build/libmfn_wp0.so timeout 300 python tools/bwd_levels.py 2>&1 | grep '^L' | sed 's/^/shared  /'
  timeout 300 python tools/bwd_levels.py 2>&1 | grep '^L' | sed 's/^/private /'
done
timeout 600 python tools/phase_bwd_pix.py 2>&1 | grep -v amdgpu.ids | grep "gx=write goffset=write"
timeout 600 python tools/phase_bwd_pix.py detail 2>&1 | grep -v amdgpu.ids | grep "gx=write goffset=write"
[ -n "$NOTEST" ] || timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "deform and (bwd or backward)" 2>&1 | tail -3
