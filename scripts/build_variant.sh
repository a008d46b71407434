#!/bin/bash
# usage: bash scripts/build_variant.sh <name> "<extra hipcc flags>": another build of libpcreg_hip.so as pcreg_amd/variants/<name>.so
# (objects under /tmp; select it at run time with PCREG_LIB=pcreg_amd/variants/<name>.so) -- for A/B measurements of compile-time knobs
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2
OBJ=/tmp/pcreg_variant_$NAME; mkdir -p $OBJ $ROOT/pcreg_amd/variants
cd $ROOT/pcreg_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-math-errno -Wall -Wno-unused-function $EXTRA"
for f in api ransac knn_points knn_fast match_features match_sad16 align descriptors sweep io_formats comm; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $OBJ/$f.o &
done
/opt/rocm/bin/hipcc $FLAGS -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form=1 -c knn_mfma16.hip -o $OBJ/knn_mfma16.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/pcreg_amd/variants/$NAME.so $OBJ/*.o -lz -ldl -lrt
echo built pcreg_amd/variants/$NAME.so
