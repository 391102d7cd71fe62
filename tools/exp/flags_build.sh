#!/bin/bash
# the default library built with extra compiler flags, for an A/B of code generation options (schedule-only options must come out bit-identical:
# tools/exp/bitcmp.py):   tools/exp/flags_build.sh ilp "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy"  ->  hmd_ego_pose_amd/libhep_ilp.so
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
: "${1:?usage: flags_build.sh <tag> \"<flags>\"}"
make -C "$R/hmd_ego_pose_amd/csrc" -j8 OUT="$R/hmd_ego_pose_amd/libhep_$1.so" OBJDIR="$R/hmd_ego_pose_amd/csrc/build_$1" EXTRA="$2" > /dev/null
rm -rf "$R/hmd_ego_pose_amd/csrc/build_$1"; ls -la "$R/hmd_ego_pose_amd/libhep_$1.so"
