cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/${1:-r04_corr11}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gram" 2>&1 | tail -2
timeout 300 python tools/corr_ab.py "corr_variant=16;corr_variant=40;corr_variant=42" 2 cfg2 7 2>&1 | grep "corr L" | tee $O/corr_ab_l2.txt
timeout 300 python tools/corr_ab.py "corr_variant=16;corr_variant=40;corr_variant=42" 2 cfg3 7 2>&1 | grep "corr L" | tee $O/corr_ab_l2_cfg3.txt
