#!/bin/bash
# build a measurement variant of the library: tools/build_dbg.sh NAME -DFLAG [-DFLAG...]  -> calibrating_amd/lib/dbg_NAME.so
# (load it with `python bench.py --lib calibrating_amd/lib/dbg_NAME.so`)
NAME=$1; shift
cd "$(dirname "$0")/../calibrating_amd/csrc"
mkdir -p ../lib/dbg_$NAME
for f in api sgbm post remap resize depth tables pointcloud; do
  if [ $f = sgbm ]; then /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function "$@" -c $f.hip -o ../lib/dbg_$NAME/$f.o || exit 1
  else cp ../lib/obj/$f.o ../lib/dbg_$NAME/$f.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/dbg_$NAME.so ../lib/dbg_$NAME/*.o && rm -rf ../lib/dbg_$NAME && echo built dbg_$NAME.so
