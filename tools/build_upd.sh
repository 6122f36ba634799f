# builds the update harness (tools/upd_bench.hip) plain and stamped; extra -D flags as arguments
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function -Wno-unused-result"
hipcc $F "$@" tools/upd_bench.hip -o tools/upd_bench.bin 2>&1 | grep -E "error" &
hipcc $F -DDDRL_STAMPS "$@" tools/upd_bench.hip -o tools/upd_bench_st.bin 2>&1 | grep -E "error" &
wait
