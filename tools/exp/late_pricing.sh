#!/bin/bash
# what would an image-resident kernel for blocks 11-15 (and 5-10) buy with four batches in flight?  timing-only library of
# tools/exp/late_pricing_build.sh: the blocks' launches dropped (upper bound), then a G-workgroup stand-in of T us in their place
L="HEP_LIB=$PWD/hmd_ego_pose_amd/libhep_latex.so"
S1="HEP_EXP_SKIP=b11.,b12.,b13.,b14.,b15."
S2="HEP_EXP_SKIP=b5.,b6.,b7.,b8.,b9.,b10.,b11.,b12.,b13.,b14.,b15."
bash tools/exp/bf16_ab.sh "X=0" "$L" "$L $S1" "$L $S1 HEP_EXP_DUMMY=16,100" "$L $S1 HEP_EXP_DUMMY=16,150" "$L $S1 HEP_EXP_DUMMY=16,250" "$L $S1 HEP_EXP_DUMMY=8,200" "$L $S1 HEP_EXP_DUMMY=32,100" \
  "$L $S2" "$L $S2 HEP_EXP_DUMMY=16,300" "$L $S2 HEP_EXP_DUMMY=32,200" "$L $S2 HEP_EXP_DUMMY=64,150" "X=0"
