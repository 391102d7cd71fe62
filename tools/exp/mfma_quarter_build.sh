#!/bin/bash
# TIMING-ONLY build: every group of four exact-fp32 MFMAs (v_mfma_f32_16x16x4_f32) runs ONE of them - results are WRONG, loads / LDS
# traffic / barriers are unchanged - to size what the matrix pipe costs an fp32 session (NOTEBOOK.md section 2, round 4: 0.950 ms against
# 1.058 ms one batch, 28.2k against 24.7k frames/s with four in flight).  Works on a scratch copy of csrc/; the product library is untouched.
#   tools/exp/mfma_quarter_build.sh && HEP_LIB=$PWD/hmd_ego_pose_amd/libhep_mfmaq.so python bench.py --precision fp32 --no-cpu-baseline --no-comm --no-fp32
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
T="$R/hmd_ego_pose_amd/csrc_mfmaq"; rm -rf "$T"; cp -r "$R/hmd_ego_pose_amd/csrc" "$T"; rm -rf "$T"/build*
python3 - "$T" <<'PY'
import re, sys
for f in ("k_mbf.hip", "k_pw_impl.h", "k_tower.hip", "k_sep.hip", "k_chain.hip", "k_xbf.hip"):
    p = sys.argv[1] + "/" + f
    s = open(p).read()
    s = re.sub(r"for \(int (q|qq) = 0; \1 < 4; \1\+\+\)(\s*\{?\s*[^;]*?__builtin_amdgcn_mfma_f32_16x16x4f32)", r"for (int \1 = 0; \1 < 1; \1++)\2", s, flags=re.S)
    open(p, "w").write(s)
PY
make -C "$T" -j8 OUT="$R/hmd_ego_pose_amd/libhep_mfmaq.so" OBJDIR="$T/build" ROOT="$R" > /dev/null
rm -rf "$T"; ls -la "$R/hmd_ego_pose_amd/libhep_mfmaq.so"
