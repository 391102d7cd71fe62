#!/bin/bash
# usage: tools/kres.sh k_mbf.hip [grep pattern]  - register / LDS / scratch use of every kernel in a source file
cd "$(dirname "$0")/../hmd_ego_pose_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -ffp-contract=fast -c "$1" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" | \
  sed -e 's/.*remark: [^ ]* *//' | paste - - - - - - - | grep -E "${2:-.}" | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | cut -c1-260
