#!/bin/bash
# Resource usage of every kernel in the ISA files that `make -C instance_stixels_amd/csrc asm` leaves in /tmp/is_asm.
for f in /tmp/is_asm/is_k_*-hip-amdgcn-amd-amdhsa-gfx950.s; do
  awk '/^ +\.name: /{n=$2} /^ +\.sgpr_count:/{s=$2} /^ +\.sgpr_spill_count:/{ss=$2} /^ +\.vgpr_count:/{v=$2} /^ +\.group_segment_fixed_size:/{l=$2} /^ +\.private_segment_fixed_size:/{p=$2} /^ +\.vgpr_spill_count:/{printf "%-70s vgpr %3d sgpr %3d spill v%d s%d scratch %d lds %d\n", substr(n,1,70), v, s, $2, ss, p, l}' $f
done
