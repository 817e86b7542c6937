#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=1 ..." : tuning build of libmdil_hip.so into gpurun_tmp/libmdil_NAME.so
set -e
name=$1; flags=$2
src=/root/repo/mdil_ss_amd/csrc
tmp=/tmp/variant_$name; mkdir -p $tmp /root/repo/gpurun_tmp
for f in tapconv sconv wconv w4conv c16conv wgrad bn pool outconv loss head adam augment; do
  [ -f $src/$f.hip ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c $src/$f.hip -o $tmp/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $src/errors.cpp -o $tmp/errors.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $src/blocks.cpp -o $tmp/blocks.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $src/prof.cpp -o $tmp/prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $tmp/*.o -o /root/repo/gpurun_tmp/libmdil_$name.so
echo built gpurun_tmp/libmdil_$name.so
