#!/bin/bash
# build_variant.sh NAME "extra flags": a tools-build variant library ../abl_NAME.so (all translation units recompiled)
cd /root/repo/robust-dynrf_amd/csrc
CX="-O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -disable-promote-alloca-to-lds=1 --offload-arch=gfx950 -I../../include -I. -Wno-unused-result -DRDRF_TOOLS $2"
mkdir -p /tmp/var_$1
for f in rdrf_pack rdrf_fwd rdrf_misc rdrf_bwd rdrf_render rdrf_optim rdrf_loss rdrf_sort; do
  /opt/rocm/bin/hipcc $CX -c $f.hip -o /tmp/var_$1/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_$1/*.o -o ../abl_$1.so && echo built abl_$1.so
