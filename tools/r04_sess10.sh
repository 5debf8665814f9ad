cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/${1:-r04_corr10}; mkdir -p $O
timeout 200 python tools/corr_ab.py "corr_variant=16;corr_variant=40;corr_variant=42" 2 cfg2 5 2>&1 | grep "corr L" | tee $O/ab.txt
for m in 1 2 8 10 11; do echo "ablate mask $m"; MFN_HIP_SO=tools/ablate_build/libmfn_gram_$m.so timeout 200 python tools/corr_ab.py "corr_variant=40" 2 cfg2 5 2>&1 | grep "corr L"; done | tee $O/gram_ablate.txt
MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so python tools/timeline_gram.py cfg2 40 2>&1 | grep -v amdgpu.ids | tee $O/timeline.txt
