#!/bin/bash
# Builds an experiment variant of liblensflare_hip.so (never shipped) into lens-flare_amd/build_ab/<name>/:
#   bash profiles/build_variant.sh <name> "<extra hipcc flags, e.g. -DLF_STOP_RCP>" [file.hip ...]
# Only the listed translation units (default: csrc/lf_march.hip) are rebuilt with the extra flags; the other
# objects are the shipped build's (make -C lens-flare_amd first).  Run with LF_LIB=<the .so> (profiles/ab_march.sh).
set -e
NAME=$1; EXTRA=$2; shift 2 || true
FILES=${@:-csrc/lf_march.hip}
PKG=$(cd "$(dirname "$0")/../lens-flare_amd" && pwd)
OUT=$PKG/build_ab/$NAME
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -mllvm -structurizecfg-skip-uniform-regions -fno-slp-vectorize -I$PKG/../include -I$PKG/csrc"
OBJS=""
for f in csrc/lf_api.hip csrc/lf_flare_kernels.hip csrc/lf_march.hip csrc/lf_cull.hip csrc/lf_scene.hip csrc/lf_lens_camera.hip csrc/lf_group.hip; do
  b=$(basename $f .hip)
  if [[ " $FILES " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $PKG/$f -o $OUT/$b.o
    OBJS="$OBJS $OUT/$b.o"
  else
    OBJS="$OBJS $PKG/build/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liblensflare_hip.so $OBJS $PKG/build/lf_collada.o -ldl -lpthread
echo "built $OUT/liblensflare_hip.so ($EXTRA)"
