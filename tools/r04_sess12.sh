cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/${1:-r04_corr12}; mkdir -p $O
tools/ubench/dot2c_check | tee $O/dot2c_check.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gram" 2>&1 | tail -2
timeout 300 python tools/corr_ab.py "corr_variant=16;corr_variant=40;corr_variant=42" 2 cfg2 7 2>&1 | grep "corr L" | tee $O/corr_ab_l2.txt
echo "dot2c build"; MFN_HIP_SO=tools/ablate_build/libmfn_gram_dot2.so timeout 300 python tools/corr_ab.py "corr_variant=16;corr_variant=40;corr_variant=42" 2 cfg2 7 2>&1 | grep "corr L" | tee $O/corr_ab_l2_dot2.txt
MFN_HIP_SO=tools/ablate_build/libmfn_gram_dot2.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gram" 2>&1 | tail -2
