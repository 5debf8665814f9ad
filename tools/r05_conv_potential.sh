cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
export MFN_HIP_SO=tools/ablate_build/libmfn_dcm_noint.so
for c in "" "dc_mt=4,dc_pt=2,dc_nw=4" "dc_mt=4,dc_pt=1,dc_nw=4" "dc_mt=4,dc_pt=4,dc_nw=4" "dc_mt=4,dc_pt=1,dc_nw=2" "dc_mt=2,dc_pt=2,dc_nw=4" "dc_mt=2,dc_pt=4,dc_nw=4" "dc_mt=3,dc_pt=2,dc_nw=4" "dc_mt=3,dc_pt=4,dc_nw=4"; do
  timeout 200 python tools/conv_vs_dcm.py "$c" 2>&1 | grep "Cin" | sed 's/| max rel.*//'
done
