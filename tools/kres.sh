#!/bin/bash
# Resource usage of every kernel of the ISA files of a build directory (default: the product build,
# instance_stixels_amd/csrc/build, where `make` leaves <file>-hip-amdgcn-amd-amdhsa-gfx950.s next to the objects).
DIR=${1:-$(dirname "$0")/../instance_stixels_amd/csrc/build}
for f in "$DIR"/is_k_*-hip-amdgcn-amd-amdhsa-gfx950.s; do
  awk '/^ +\.name: /{n=$2} /^ +\.sgpr_count:/{s=$2} /^ +\.sgpr_spill_count:/{ss=$2} /^ +\.vgpr_count:/{v=$2} /^ +\.group_segment_fixed_size:/{l=$2} /^ +\.private_segment_fixed_size:/{p=$2} /^ +\.vgpr_spill_count:/{printf "vgpr %3d sgpr %3d spill v%d s%d scratch %d lds %d  %s\n", v, s, $2, ss, p, l, n}' "$f" | c++filt | sed 's/(DevParams.*//'
done
