#!/bin/bash
# Builds a variant of libmi355nrphy.so with extra compiler flags on one or more sources, for A/B runs on one box
# (profiles/ab_lib.sh, profiles/ab_variants.sh).  Build container, repository root:
#   bash profiles/make_variant.sh NAME "ofdm_kernels.hip" "-DNRPHY_WIRE_EXP=1"              ->  build/variants/NAME.so
#   bash profiles/make_variant.sh NAME "pdsch_kernels.hip nrphy_host.cpp" "-DNRPHY_CRC_SLICES=3"
set -eu
NAME=$1; SRCS=$2; EXTRA=${3:-}
C=srsran-edgeric-5g_amd/csrc
mkdir -p build/variants
python3 srsran-edgeric-5g_amd/build.py > /dev/null          # the other objects, current
OBJS=$(ls $C/*.o)
for SRC in $SRCS; do
  CONTRACT=-ffp-contract=off
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Iinclude -I$C $CONTRACT $EXTRA -x hip -c $C/$SRC -o build/variants/$NAME.${SRC%.*}.o
  OBJS=$(echo "$OBJS" | tr ' ' '\n' | grep -v "/${SRC%.*}.o$")
  OBJS="$OBJS
build/variants/$NAME.${SRC%.*}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$NAME.so $OBJS
rm -f build/variants/$NAME.*.o
echo build/variants/$NAME.so
