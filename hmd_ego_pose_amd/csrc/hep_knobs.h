// hep_knobs.h - every environment variable the plan builder looks at, read in ONE place (hep_knobs.cpp) when a session is created.
//
// Default library (libhep.so): the ten knobs below "default build" - each forces a form that the planner selects by itself for
// some other shape or precision (so that the parity tests reach that form at a small size), or is a reporting switch.
// Alternative library (libhep_alt.so, `make -C hmd_ego_pose_amd/csrc alt`, -DHEP_ALT): additionally the A/B knobs whose measurement is a
// recorded loss or tie (NOTEBOOK.md), and the three rejected kernels they select (k_late.hip, k_heads.hip, k_sbf.hip).
#pragma once

namespace hep {

struct Knobs {
  // ---- default build ----
  int lanes = 1;              // HEP_LANES: slices of the batch run as parallel graph branches (1: measured best)
  int chain_stream = -1;      // HEP_CHAIN_STREAM=0|1|2: chain_kernel weight mode (resident / LDS-DMA streamed / pointwise fragments from global memory); -1 = by fit
  int tower_coop = -1;        // HEP_TOWER_COOP=0|1|2|3: cooperative tower kernel off / everywhere / map layers / headers; -1 = by dtype and width
  int mbf_mp = 1;             // HEP_MBF_MP=0|1|2|force(3): multi-pass expand of the fused fronts off / by rounds / bf16 from two rounds / wherever it exists
  int xbf_generic = 0;        // HEP_XBF_GENERIC=1: the generic boundary-kernel instantiation where a shape-specialised one exists
  int stem_mfma = -1;         // HEP_STEM_MFMA=0|1: VALU / MFMA stem; -1 = by width
  double se_maxmb = 4.0;      // HEP_SE_MAXMB: squeeze-excite finished in the project GEMM's prologue below this many MB of re-read expand-FC weights per launch
  int pw_frag = 1;            // HEP_PW_FRAG=0: row-major operands for the split-K project GEMMs (bit-identity test of the fragment order)
  int sep_wlds = 1;           // HEP_SEP_WLDS=0: BiFPN nodes wider than 64 fetch their pointwise weights per fragment instead of staging them in LDS
  int plan_debug = 0;         // HEP_PLAN_DEBUG=1: the front plan of every block on stderr
  // ---- alternative build only (constants in the default build) ----
  int pw_nt2 = 2;             // HEP_PW_NT2: widest split-K tile
  int pw_mt2 = 1;             // HEP_PW_MT2=0: one m-tile per wave in modes 1 / 2
  int pw_wide = 0;            // HEP_PW_WIDE=cap (alt build): split-K project GEMMs with a squeeze-excite input take up to cap n-tiles per workgroup (fewest column chunks)
  int pw_nt3 = 0;             // HEP_PW_NT3=1: one more n-tile per wave where that saves a round of workgroups (fp32 default until round 6: -0.6 % with four batches in flight)
  int pw_w8 = 1;              // HEP_PW_W8=0: four waves on the fp32 split-K GEMMs
  int pw_w8_mink = 512;       // HEP_PW_W8_MINK
  int se_tail = 0;            // HEP_SE_TAIL=1: squeeze-excite finish in the tail of the fused front
  int xbf = 1;                // HEP_XBF=0: no boundary launches
  int xbf_minh = 64;          // HEP_XBF_MINH: smallest input map that takes the boundary kernel
  int xbf_tpw = 0;            // HEP_XBF_TPW: tiles per workgroup (0: by workgroup count)
  int mbf = -1;               // HEP_MBF=all(1)|none(0): fused front everywhere / nowhere; -1 = by map size
  int mbf_maxh = 32;          // HEP_MBF_MAXH: largest input map that takes the fused front
  int mbf_ts8 = 0;            // HEP_MBF_TS=8: 8x8 tiles only
  int mbf_ts16_maxh = 32;     // HEP_MBF_TS16_MAXH
  int mbf_cc = 64;            // HEP_MBF_CC=32|16: widest channel chunk of a fused front; -n: 32 where 64 gives <= n workgroups and 32 still <= 256 (one batch +1.1 %, four in flight -1.4 %)
  int mbf_mp_res = 0;         // HEP_MBF_MP_RES=1: multi-pass with the whole tile held in registers
  int dwlds = -1;             // HEP_DWLDS=0|1: stand-alone depthwise through LDS never / always; -1 = by shape
  int late = 0;               // HEP_LATE=1: blocks 12-15 as one image-resident launch (k_late.hip)
  int late_g = 3;             // HEP_LATE_G: workgroups per image of that launch
  int late_xcd = 0;           // HEP_LATE_XCD=1: a group's members on consecutive ids
  int heads_fused = 0;        // HEP_HEADS_FUSED=1: depth-first head kernel (k_heads.hip)
  int sbf = 0;                // HEP_SBF=1: stem + block 0 depthwise as one launch (k_sbf.hip)
  int tower = 1;              // HEP_TOWER=0: heads on k_sep.hip
  int chain = 2;              // HEP_CHAIN=0|1|2: nodes launch by launch / k_sep.hip chains / LDS-resident chains
  int chain_f32 = 1;          // HEP_CHAIN_F32=0: fp32 chains on k_sep.hip
  int chain_wglobal = 1;      // HEP_CHAIN_WGLOBAL=0: no chain with pointwise weights from global memory
  int pwg = 1;                // HEP_PWG=0: the six lateral convs as six launches
  int sep_ts4_maxhw = 0;      // HEP_SEP_TS4_MAXHW: 4x4 tiles for single BiFPN nodes of levels up to this size
};

Knobs read_knobs();           // reads the environment now (never cached: a session created after the environment changed gets the plan it asked for)

}  // namespace hep
