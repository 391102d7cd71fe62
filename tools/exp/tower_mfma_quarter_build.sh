#!/bin/bash
# TIMING-ONLY build (wrong results): the exact-fp32 MFMAs of the head kernels (k_tower.hip) run ONE of every four - the matrix-pipe time a
# split-bf16 (hi + lo) product would leave (3 x 16x16x32_bf16 = 384 cycles against 16 x 16x16x4_f32 = 2048 per n-tile and k = 64) - to price
# that precision change for the fp32 heads before building it.   tools/exp/tower_mfma_quarter_build.sh && tools/exp/lib_ab.sh hmd_ego_pose_amd/libhep.so hmd_ego_pose_amd/libhep_towq.so 3 --precision fp32
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
T="$R/hmd_ego_pose_amd/csrc_towq"; rm -rf "$T"; cp -r "$R/hmd_ego_pose_amd/csrc" "$T"; rm -rf "$T"/build*
python3 - "$T" <<'PY'
import re, sys
p = sys.argv[1] + "/k_tower.hip"
s = open(p).read()
s, n = re.subn(r"for \(int q = 0; q < 4; q\+\+\) c = __builtin_amdgcn_mfma_f32_16x16x4f32", "for (int q = 0; q < 1; q++) c = __builtin_amdgcn_mfma_f32_16x16x4f32", s)
assert n == 1, n
open(p, "w").write(s)
PY
make -C "$T" -j8 OUT="$R/hmd_ego_pose_amd/libhep_towq.so" OBJDIR="$T/build" ROOT="$R" > /dev/null
rm -rf "$T"; ls -la "$R/hmd_ego_pose_amd/libhep_towq.so"
