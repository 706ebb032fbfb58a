cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 1 2 8 9; do timeout 120 python tools/ws_lnf_ab.py tools/_abl/libvcx_wlabl$v.so 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05an_ws_lnf_ablate.txt
cat gpurun_out/r05an_ws_lnf_ablate.txt
