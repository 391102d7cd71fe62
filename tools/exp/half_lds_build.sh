#!/bin/bash
# TIMING-ONLY build (wrong results): the fused fronts (k_mbf.hip) and the boundary launches (k_xbf.hip) REQUEST half of their dynamic LDS,
# i.e. the footprint they would have if the input tile and the expanded tile were stored as e4m3 instead of bf16 (BASELINE config 5 with the
# producers emitting e4m3 activations).  The kernels still address the full range (out-of-range LDS accesses are dropped by the hardware), so
# the instruction stream, the loads and the barriers are those of today: what changes is how many workgroups the dispatcher co-locates on a
# CU.  An UPPER bound of what e4m3 storage could buy through occupancy - before its conversion instructions cost anything.
#   tools/exp/half_lds_build.sh && tools/exp/lib_ab.sh hmd_ego_pose_amd/libhep.so hmd_ego_pose_amd/libhep_halflds.so 4
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
T="$R/hmd_ego_pose_amd/csrc_halflds"; rm -rf "$T"; cp -r "$R/hmd_ego_pose_amd/csrc" "$T"; rm -rf "$T"/build*
sed -i 's/a\.lds_bytes, s, a)/a.lds_bytes \/ 2, s, a)/g' "$T/k_mbf.hip" "$T/k_xbf.hip"
grep -c 'a.lds_bytes / 2, s, a' "$T/k_mbf.hip" "$T/k_xbf.hip"
make -C "$T" -j8 OUT="$R/hmd_ego_pose_amd/libhep_halflds.so" OBJDIR="$T/build" ROOT="$R" > /dev/null
rm -rf "$T"; ls -la "$R/hmd_ego_pose_amd/libhep_halflds.so"
