#!/bin/bash
# Build a variant of libgdca.so with extra compiler flags (kernel experiments): tools/build_variant.sh <name> [-DFLAG ...]
# -> gaussdca.jl_amd/libgdca_<name>.so; run anything against it with GDCA_LIB=<that path>.
set -e
name=$1; shift
cd "$(dirname "$0")/../gaussdca.jl_amd/csrc"
obj=_obj_$name
mkdir -p $obj
FLAGS="-O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 -Wall -Wno-unused-function -I../../include -I. $*"
for f in gdca_api k_theta k_hamming k_tally k_elementwise k_inverse k_score k_rank; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $obj/$f.o &
done
g++ -O2 -fPIC -std=c++17 -Wall -pthread -I../../include -I. -c gdca_host.cpp -o $obj/gdca_host.o &
g++ -O2 -fPIC -std=c++17 -Wall -I../../include -I. -c gdca_inflate.cpp -o $obj/gdca_inflate.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgdca_$name.so $obj/*.o -lz -pthread
rm -rf $obj
echo built ../libgdca_$name.so
