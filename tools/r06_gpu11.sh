set -e
O=gpurun_out/r06; mkdir -p $O /tmp/b
for o in 0 1 2; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSCVX_CHOL_DPP=2 -DSCVX_CHOL_ORDER=$o -Iinclude -Isuccessiveconvexification_amd/csrc -o /tmp/b/chol_o$o tools/micro/chol_dpp_ab.hip; done
for o in 0 1 2; do echo "SCVX_CHOL_ORDER=$o"; timeout -k 10 60 /tmp/b/chol_o$o 2048; timeout -k 10 60 /tmp/b/chol_o$o 64; done > $O/chol_dpp_micro_order.txt 2>&1
cat $O/chol_dpp_micro_order.txt
