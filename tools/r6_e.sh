#!/bin/bash
# round 6: s_setprio experiments on the product kernel, interleaved library A/B in one box (tools/ab_lib.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6e; mkdir -p $O; cd $R
python -m gglasso_amd.build > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
OBJS=$(ls gglasso_amd/lib/*.o | grep -v "\.dev\.o" | grep -v gemm_sym.o)
cp gglasso_amd/lib/libggl_hip.so /tmp/libA.so
for v in 1 2; do
  /opt/rocm/bin/hipcc $FLAGS -DGGL_EXP_SETPRIO=$v -c gglasso_amd/csrc/gemm_sym.hip -o /tmp/gemm_sym_$v.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libB$v.so $OBJS /tmp/gemm_sym_$v.o -L/opt/rocm/lib -lrocsolver -lrocblas -ldl -Wl,-rpath,/opt/rocm/lib || exit 1
done
bash tools/ab_lib.sh r6e_prio1 3 /tmp/libA.so /tmp/libB1.so "" "--workload ggl_K32_p1000 --steps 30" "--workload ggl_K8_p500" > $O/setprio1.txt 2>&1; cat $O/setprio1.txt
bash tools/ab_lib.sh r6e_prio2 3 /tmp/libA.so /tmp/libB2.so "" "--workload ggl_K32_p1000 --steps 30" "--workload ggl_K8_p500" > $O/setprio2.txt 2>&1; cat $O/setprio2.txt
