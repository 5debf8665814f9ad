cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/bwd16b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "deform and (bwd or backward)" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python tools/phase_bwd_pix.py > $O/phases.txt 2>&1; grep "^L. gx=write goffset=write" $O/phases.txt
timeout 300 python tools/phase_bwd_pix.py detail > $O/phases_detail.txt 2>&1; grep "^L. gx=write goffset=write" $O/phases_detail.txt
