// hep_internal.h - shared between the host plan builder and the gfx950 kernels.
// Activations are NHWC ([B,H,W,C], C a multiple of 8) in `dtype` (fp32 or bf16);
// biases, depthwise weights, SE weights and fusion weights are always fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// exact unsigned division by a launch-time constant: rcp_u32(d) on the host, udiv_rcp(x, rcp) in the kernel.
// floor(x / d) == mulhi(x, floor(2^32 / d) + 1) whenever x * d < 2^32 (grid indices and tile counts here); d == 1 is
// flagged with rcp == 0.  A run-time integer division costs ~30 VALU instructions per wave.
static inline uint32_t rcp_u32(uint32_t d) { return d <= 1 ? 0u : (uint32_t)(0x100000000ull / d) + 1u; }

#define HEP_MAX_SRC 3
#define SEP_MAX_TILES_N 6    // n-tiles (16 columns) per head-output sepconv segment; wider headers are split into segments
#define SEP_MAX_TILES_MAP 24 // n-tiles of a map-producing segment (BiFPN width <= 384): never split

enum { ACT_NONE = 0, ACT_SWISH = 1, ACT_SIGMOID = 2 };
enum { SRC_SAME = 1, SRC_UP = 2, SRC_DOWN = 3 };   // gather kinds of a BiFPN fusion input

// ---- stem: conv3x3 s2 SAME, Cin=3 (+folded BN, swish); fp32 strided NCHW in -> NHWC out ----
struct StemArgs {
  const float* in; int64_t sn, sc, sh, sw;   // element strides of the caller's tensor
  const float* w;      // [3][3][3][Cout] (ky,kx,ci,co), BN folded
  const float* bias;   // [Cout]
  void* out;           // [B,Ho,Wo,Cout]
  int B, H, W, Ho, Wo, Cout, pad_t, pad_l, bf16;
  int mpw; uint32_t tx_rcp, ho_rcp;          // filled by launch_stem: m-tiles per wave, rcp_u32 of m-tiles per row and of Ho
  int mfma;                                  // plan: 1 = stem_kernel (MFMA form), 0 = stem_valu_kernel (stem_uses_mfma() when the plan is built)
};

#ifdef HEP_ALT      // ---- the alternative library's kernels (make alt: k_sbf.hip, k_late.hip, k_heads.hip; measured losses / ties, NOTEBOOK.md) ----
// ---- stem conv + block 0's depthwise conv as one launch (k_sbf.hip): the stem's output never reaches HBM ----
struct SbfArgs {
  const float* in; int64_t sn, sc, sh, sw;      // the caller's fp32 input through its own element strides
  const float* w_stem; const float* b_stem;     // [3][3][3][C] (ky,kx,ci,co) BN folded; [C]
  const float* wdw; const float* bdw;           // [9][C] BN folded; [C]
  void* stem_out;                               // [B,Hs,Ws,C] (nullable: stored for the stage tests only)
  void* out;                                    // [B,Hs,Ws,C] block 0's depthwise output
  float* hpart; const float* se_wr; int sq, sqp; // [B][tiles][sqp] partial reduce-FC products; reduce weight [sq][C]
  int B, H, W, Hs, Ws, C, pad_t, pad_l, bf16;
  int tiles_x, tiles; uint32_t tiles_rcp, tiles_x_rcp;
  int off_s, off_w, off_red; size_t lds_bytes;
};
void sbf_layout(SbfArgs* a);                     // fills tiles and the LDS layout from the shapes
void launch_sbf(const SbfArgs&, hipStream_t);
#endif

// widest split-K tile (n-tiles per workgroup) the fragment-ordered GEMM is instantiated for: the planner (add_pw), the launcher (launch_nt)
// and hep_kernel_symbol all test against THIS constant - a fragment-ordered tensor handed to a row-major instantiation would be misread
#ifdef HEP_ALT
constexpr int PW_FRAG_MAX_NT = 8;      // (HEP_PW_WIDE)
#else
constexpr int PW_FRAG_MAX_NT = 4;
#endif

// ---- pointwise conv as GEMM: out[M,N] = act((A[M,K] (*se)) . W[N,K]^T + bias) (+res) ----
struct PwArgs {
  const void* A; const void* W; const float* bias;
  // squeeze-excite on the input side (project convs; hpart == nullptr: none): the front kernel left
  // hpart[b][row][sqp] = per-workgroup partial products of the reduce FC with its channel sums; the
  // prologue of this kernel finishes hidden = swish(sum_rows * inv_hw + br) and the expand FC
  // scale[k] = sigmoid(we[k,:] . hidden + be[k]) for the images its rows belong to (k_pw.hip)
  const float* hpart; const float* se_br; const void* se_we /*[K][sqp], session dtype*/; const float* se_be;
  int se_rows, sq, sqp, se_nimg; float inv_hw;
  int se_from_tensor; const float* se_scale;   // the scale [B][K] was finished by se_finish_kernel (large K x sq): only copied to LDS
  const void* res;     // [M,N] residual (nullable)
  void* out;
  int M, K, N, tilesN; // tilesN = ceil(N/16); W holds tilesN*16 rows
  int HW;              // rows per image (for se)
  int act, bf16, MT, NT;
  int fp8; const float* wscale; float a_scale;   // fp8 operands (k_pw_impl.h): W is e4m3 [tilesN*16][K], per-row scales, per-tensor activation scale
  int mode;            // wave arrangement inside a workgroup: 0 along M, 1 along N, 2 split-K (k_pw.hip)
  int nwv;             // waves per workgroup: 4; 8 for the split-K project convs of fp32 sessions (k_pw_impl.h)
  int frag;            // A and W are stored in MFMA fragment order ([tile][k-step][lane][8]: k_pw_impl.h FRAG; the front kernel wrote A that way)
  int chunksN; uint32_t chunksN_rcp;   // filled by launch_pw_prec: workgroups along N and rcp_u32() of that
  unsigned long long* trace_buf;   // profiling builds (-DHEP_PW_TRACE): stamps of this launch go here
};

// ---- several small independent pointwise convs in ONE launch (the BiFPN lateral 1x1 convs) ----
#define PWG_MAX 8
struct PwgSeg { const void* A; const void* W; const float* bias; void* out; int HW, K, N, tilesN, blk_begin; };   // blk_begin: prefix of 16-row strips per image
struct PwgArgs { PwgSeg seg[PWG_MAX]; int nseg, B, bf16, strips_per_image; };

// ---- depthwise kxk conv + folded BN + swish (+ per-block channel sums for SE) ----
struct DwArgs {
  const void* in; const float* w /*[k*k][C]*/; const float* bias; void* out;
  float* hpart;        // [B][blocks_per_image][sqp]: partial reduce-FC products of this block's channel sums (nullable)
  const float* se_wr;  // [sq][C] squeeze-excite reduce weight
  int sq, sqp;
  int B, H, W, C, Ho, Wo, k, s, pad_t, pad_l, act, bf16, TW, blocks_per_image;
  uint32_t cg_magic, sw_magic, c_magic;   // reciprocals for C/8, strips per row and C (filled in by launch_dw)
};

// ---- squeeze-excite finish as its own launch (blocks whose expand-FC matrix is too large for every project workgroup) ----
struct SeFinishArgs {
  const float* hpart; const float* br; const void* we /*[C][sqp] session dtype*/; const float* be; float* scale /*[B][C]*/;
  int B, C, sq, sqp, rows, bf16; float inv_hw;
};
void launch_se_finish(const SeFinishArgs&, hipStream_t);

// ---- fused MBConv front: expand 1x1 (+BN,swish) -> depthwise kxk (+BN,swish) -> SE partial sums ----
struct MbfArgs {
  const void* in;      // [B,H,W,Cin]
  const void* we;      // [ceil16(Cexp)][Cin] expand weight, BN0 folded (unused when !has_expand)
  const float* be;     // [Cexp]
  const float* wdw;    // [k*k][Cexp], BN1 folded
  const float* bdw;    // [Cexp]
  void* out;           // [B,Ho,Wo,Cexp]
  float* hpart;        // [B][tiles*chunks][sqp]: this workgroup's partial reduce-FC products (one row per workgroup)
  const float* se_wr;  // [sq][Cexp] squeeze-excite reduce weight
  int sq, sqp;
  int B, H, W, Cin, Cexp, Ho, Wo, k, s, pad_t, pad_l, has_expand, bf16;
  int CC;              // expanded channels per workgroup (8 * power of two)
  int ts;              // output tile side: 8, or 16 (stride-1 layers on maps >= 16x16; k_mbf.hip)
  int npass, kp;       // multi-pass expand: K input channels in npass slices of kp (npass <= 1: one pass, the whole K staged at once)
  int mp_resident;     // multi-pass: the whole tile is requested at kernel start and held in registers (else slice by slice)
  uint32_t vk_rcp, kpv_rcp;                     // filled by launch_mbf: rcp_u32 of K / 8 and kp / 8
  size_t off_e, off_we, off_w, lds_bytes;
  int fp8; const float* we_scale; float a_scale;   // fp8 sessions: e4m3 expand weights (rows padded to 16 bytes), per-channel / per-tensor scales
  // the squeeze-excite finished in the TAIL of this launch (se_finish.h; alternative plan HEP_SE_TAIL=1, not selected: a launch of its own measured faster): every workgroup writes its
  // partial row through (sc0 sc1), takes a ticket of its image's counter, and the last one to arrive finishes hidden vector and scale
  int se_tail; unsigned* tail_counter /* [B] counters, 32 words (one 128-byte line) apart, zero between launches */; SeFinishArgs tail;
  int out_frag;        // store the output in the project GEMM's fragment order [m-tile = 16 rows of (image, pixel)][k-step][lane][8] (k_pw_impl.h FRAG)
  int trace;           // profiling builds (-DHEP_MBF_TRACE): this launch writes its phase time stamps
  int chunks, tiles_x;                          // filled by launch_mbf: channel chunks per tile, tiles per row
  uint32_t chunks_rcp, tiles_x_rcp, gx_rcp;     // rcp_u32() of chunks, tiles_x, gridDim.x (device: udiv_rcp)
};

// ---- MBConv block BOUNDARY on the big maps as one launch (k_xbf.hip): squeeze-excite finish + project 1x1 (+BN2,
//      +residual) of block i-1  ->  expand 1x1 (+BN0, swish) -> depthwise kxk (+BN1, swish) -> SE partial sums of block i.
//      Neither the block output nor the 6x expanded tensor reaches HBM (the block output is stored only when a later
//      residual / tap / stage test needs it). ----
#define XBF_MAXNT1 3          // n-tiles (16 channels) of the project output: block widths <= 48
struct XbfArgs {
  const void* in;             // [B,H,W,K1] depthwise output of block i-1
  const float* hpart; int se_rows, sq, sqp; float inv_hw;           // its squeeze-excite: partial reduce-FC rows (k_dw.hip / this kernel)
  const float* se_br; const void* se_we /*[K1][sqp] session dtype*/; const float* se_be;
  const void* blob;           // project + expand + depthwise weights and biases of the launch, already in the kernel's LDS layout
  int blob_bytes;
  const void* res; void* mid; // [B,H,W,N1]: residual input of block i-1 (nullable), block i-1 output (nullable: not stored)
  void* out;                  // [B,Ho,Wo,Cexp] depthwise output of block i
  float* hpart_out; const float* se_wr; int sq2, sqp2;               // [B][tiles][sqp2]; reduce-FC weight [sq2][Cexp] of block i
  int B, H, W, K1, N1, NT1, Cexp, NT2, Ho, Wo, k, s, pad_t, pad_l, bf16;
  int toh, tow, tiles_x, tiles, tpw;                                // output tile, tiles per row / per image, tiles per workgroup
  int chunk_tiles, nchunks;                                         // expanded n-tiles per LDS chunk (<= 2 chunks)
  uint32_t tiles_rcp, tiles_x_rcp, sw_rcp;                          // rcp_u32 of workgroups per image, tiles_x, lane sweeps per input tile row
  int trace;                                                         // profiling builds: this launch writes its phase stamps
  int generic;                                                       // plan: 1 = the generic instantiation although a shape-specialised one exists (HEP_XBF_GENERIC, A/B and parity runs)
  int off_w1, off_w2, off_f, off_misc; size_t lds_bytes;            // LDS: [a_s | e_s union][blob: w1 | w2 | floats][scale, hidden, red, csum]
};
size_t xbf_layout(XbfArgs* a);                   // fills the LDS offsets / chunking from the shapes; returns lds_bytes (0: does not fit)
int xbf_supports(int k, int s);                  // tile instantiations
void xbf_tile(int k, int s, int* toh, int* tow);
void launch_xbf(const XbfArgs&, hipStream_t);
int xbf_prepare(void);
int xbf_specialised(const XbfArgs&);          // 1: a shape-specialised instantiation exists (names the device function)

#ifdef HEP_ALT
// ---- image-resident run of late MBConv blocks (k_late.hip): ONE workgroup per image runs several consecutive stride-1 blocks
//      of the 8x8 maps back to back - expand 1x1 -> depthwise k x k -> squeeze-excite -> project 1x1 (+ residual) per block
//      (efficientnet/model.py:69-104) - with block inputs / expanded tiles / squeeze-excite state in LDS and registers and every
//      weight streamed once per image in MFMA fragment order.  bf16 sessions.  Replaces front + se_finish + project launches. ----
#define LATE_MAX_BLOCKS 6
#define LATE_CC 128             // expanded channels per chunk (8 n-tiles)
#define LATE_THREADS 1024
struct LateBlock {
  int Cin, Cexp, N, k, skip;    // 8x8 -> 8x8, stride 1; skip: residual add
  int nchunks;                  // Cexp / LATE_CC
  int sq, sqp;                  // squeeze-excite width, padded to a multiple of 8 (<= 64)
  int ntw, ng;                  // project: n-tiles per wave (2 / 3), n-groups (ng * 2 <= 16 waves: two K halves)
  float inv_hw;
  // byte offsets into the launch's weight blob (all 16-byte aligned)
  uint32_t off_we;              // expand [chunk][n-tile 8][k-step Cin/32][lane 64][8] bf16: one wave instruction = one 1 KB fragment
  uint32_t off_be;              // [Cexp] f32 expand bias (BN0 shift)
  uint32_t off_wdw;             // [chunk][k*k][128] f32 depthwise (BN1 folded)
  uint32_t off_bdw;             // [Cexp] f32
  uint32_t off_w1;              // squeeze-excite reduce FC [chunk][64 rows (>= sq: zeros)][128] f32
  uint32_t off_b1;              // [64] f32
  uint32_t off_w2;              // squeeze-excite expand FC [Cexp][sqp] bf16
  uint32_t off_b2;              // [Cexp] f32
  uint32_t off_wp;              // project [n-group][k-step Cexp/32][n-tile ntw][lane 64][8] bf16 (BN2 folded; rows >= N zero)
  uint32_t off_bp;              // [ng * ntw * 16] f32
  const void* res;              // [B][64][N] residual = the block's input in global memory (skip blocks)
  void* out;                    // [B][64][N] block output (always stored: taps, stage tests, the next launch)
};
struct LateArgs {
  const void* in;               // [B][64][Cin of block 0]
  const unsigned char* blob;
  void* dscratch;               // [B][Cexp_max / 8][64][8] bf16: depthwise outputs on their way from the chunk loop to the project conv (L2)
  int dstride;                  // bytes per image of dscratch
  LateBlock blk[LATE_MAX_BLOCKS];
  int nblk, B;
  int G;                        // workgroups per image (1, or a divisor of every block's chunk count): each runs 1 / G of the chunks, one counter meeting per block
  int cross_xcd;                // HEP_LATE_XCD=1 (tests): a group's workgroups on consecutive ids = different XCDs; default: ids equal mod 8 = one XCD
  unsigned* counters;           // [B][16] (one 64-byte line per image: a meeting counter per block): zero at session start, kept zero by the kernel
  float* hpart;                 // [B][G][64] squeeze-excite reduce-FC sums of the group's workgroups
  int off_e, e_stride, off_wdw, wdw_stride, off_bias, off_csum, off_x, off_scale, off_hid;   // LDS layout (late_layout): two buffers each of expanded tiles / depthwise weights / biases / channel sums
  int lds_bytes;
  unsigned long long* trace;    // profiling builds (-DHEP_LATE_TRACE)
};
int late_layout(LateArgs* a);      // fills the LDS offsets from the blocks' shapes; returns lds_bytes, 0 when the run does not fit
int late_block_supported(int Cin, int Cexp, int N, int k, int stride, int H, int W, int sq);
int late_prepare(void);
void launch_late(const LateArgs&, hipStream_t);

// ---- the five head towers depth-first (k_heads.hip): tower layers + headers of every (net, level) as ONE launch ----
struct HeadItem {               // one workgroup per image: a 16x16 output tile of one (net, level); a table of these opens the blob
  int net, level, hw, y0, x0, nhdr;
  uint32_t off_layers;          // blob offset of the (net, level)'s tower layers: D x { pointwise fragments [n-tile 4][k-step 2][lane 64][8] bf16 (per-level
                                // BatchNorm folded) | bias [64] f32 | depthwise [9][64] f32, swizzled (0,2,1,3 | 4,6,5,7) per channel octet as k_tower.hip }
  uint32_t off_hdr[2];          // per header: { depthwise [9][64] f32 swizzled | bias [ntiles * 16] f32 | fragments [ntiles][2][64][8] bf16 }
  int hdr_ntiles[2], hdr_N[2], hdr_kin[2], hdr_kout[2], hdr_off[2], hdr_act[2], hdr_out[2];
  int pad_[5];                  // (128 bytes)
};
struct HeadsArgs {
  const void* feat[5];          // the last BiFPN cell's maps [B][hw][hw][64] bf16
  float* out[5];                // regression, classification, rotation, translation_raw, hand: [B][num_anchors][K]
  const unsigned char* blob;    // HeadItem table, then the weights
  int level_off[5];
  int nitems, B, D, num_anchors, off_wdw, lds_bytes;
};
int heads_fused_supported(int C, int depth, int bf16);
int heads_lds_bytes(int depth, int* off_wdw);
int heads_prepare(void);
void launch_heads(const HeadsArgs&, hipStream_t);
#endif   // HEP_ALT

// ---- 3x3 s2 max-pool, TF-SAME with ZERO padding (utils_extra.py:72-86) ----
struct PoolArgs { const void* in; void* out; int B, H, W, C, Ho, Wo, pad_t, pad_l, bf16; };

// ---- fused separable conv segment: [fusion gather + swish] -> dw3x3 -> pw(+bias)(+act) ----
struct SepSeg {
  const void* src[HEP_MAX_SRC]; int kind[HEP_MAX_SRC]; float fw[HEP_MAX_SRC];
  int sh[HEP_MAX_SRC], sw[HEP_MAX_SRC];    // source spatial size
  int pool_pad[HEP_MAX_SRC];               // SAME pad-before of the 3/2 max-pool for SRC_DOWN
  int nsrc, pre_act;                       // pre_act: swish after the weighted sum
  int h, w, C;                             // dw input/output spatial size, channels
  const float* wdw;                        // [9][C]
  const void* wpw;                         // [tilesN*16][C] in dtype
  const float* bias;                       // [tilesN*16]
  int N, tilesN, act;
  void* out; int out_f32;                  // fp32 head outputs, otherwise dtype
  int64_t out_bstride, out_off, out_rowstride;   // elements: per image, level offset, per pixel
  int col_kin, col_kout, col_off;          // column nn = n_base + n -> (nn/kin)*kout + nn%kin + off
  int n_base;                              // first output column of this segment (wide headers are split)
  int ts;                                  // tile side: 16 on maps >= 16x16 (bf16), else 8
  int tiles_x, tiles_y, tile_begin;        // tile_begin = prefix of tiles-per-image over the launch's segments
  uint32_t tiles_x_rcp;                    // rcp_u32(tiles_x)
};
struct SepArgs {
  const SepSeg* segs; const int* tile_seg;   // device tables: segments, and tile (blockIdx.x) -> segment
  SepSeg seg0;                               // the only segment of a single-segment launch (kernel argument)
  int nseg; int B; int total_tiles; int bf16; int C;
  int chain;                                 // segments are a dependency chain run by one workgroup per image
  int direct;                                // k_tower.hip (wave-per-patch, no LDS staging): 1 = map layer, 2 = headers
  int coop;                                  // ... in its cooperative form (tower_coop_kernel: one 10x10 halo per workgroup, weights in registers)
  size_t off_atile, off_wdw, off_bias, lds_bytes;   // LDS layout (k_sep.hip: sep_lds_layout)
  size_t off_wpw;                            // 0, or where the node's pointwise weights [C][C + pad] are staged (bf16 nodes wider than 64 that fit)
};

// ---- LDS-resident chain of small-level BiFPN nodes, one workgroup per image (k_chain.hip) ----
#define CH_MAX_NODES 8
#define CH_MAX_EXT 8
struct ChainSrc { int off, kind, sh, sw; float fw; };    // LDS slot (element offset) holding the source map at sh x sw; kind relative to the node's size
struct ChainNode {
  ChainSrc src[HEP_MAX_SRC]; int nsrc, h, w, pool_only, pool_pad;
  int out_off; int widx; void* out;                      // LDS slot and global map of the result [B][h*w][C]; widx: index of its weights in the blob
  uint32_t w_rcp, hs_rcp;                                // floor(2^32 / w) + 1, floor(2^32 / (w + 2)) + 1: pixel index -> (row, column) without a division
};
struct ChainExt {                                        // an input map produced by an earlier launch
  const void* src; void* store;                          // store: the pooled map is also written here (p6_in of cell 0), else NULL
  int sh, sw, kind, h, w, pool_pad, off;                 // kind SRC_DOWN: pooled to h x w while it is loaded; else copied (h x w == sh x sw)
};
struct ChainArgs {
  const ChainNode* nodes; const void* wblob;             // device: node table, the nodes' weights in LDS layout
  ChainExt ext[CH_MAX_EXT];
  int nnodes, nconv, next, B, C, wnode_bytes;           // nconv: nodes with a convolution (the others are plain max-pools)
  int bf16, stream_w;                                    // session dtype; stream_w: 0 all node weights resident in LDS, 1 two nodes' weights in LDS (the next node's streamed by LDS-DMA), 2 pointwise weights straight from global memory
  uint32_t nt_rcp;                                       // rcp_u32(ceil(C / 16)): (m-tile, n-tile) pair -> m-tile (filled by launch_chain)
  size_t off_w, off_halo, off_atile, lds_bytes;
};
void launch_chain(const ChainArgs&, hipStream_t);
int chain_prepare(void);

// ---- decode: boxes + translation from raw heads (loss.py:12-51) ----
struct DecodeArgs {
  const float* regression; const float* translation_raw; const float* camera;
  const float* anchors; const float* t_anchors; float* boxes; float* translation;
  int B, N; float clip_max;
};

// ---- preprocess: uint8 RGB HWC -> (8-bit bilinear resize to nh x nw) -> normalised float32 HWC, zero-padded to S x S ----
struct PreprocArgs { const uint8_t* in; float* out; int B, H, W, S, nh, nw, resize; double inv_scale_x, inv_scale_y; };
// the WebRTC frame path (Program.cs:128-205): YV12 -> BGR of the centre crop; 8-bit bilinear resize, optionally followed by the
// float32 normalisation and the zero padding to S x S
struct Yv12Args { const uint8_t* in; uint8_t* bgr; int B, H, W, crop, ow, oh; };
struct ResizeArgs { const uint8_t* in; void* out; int B, H, W, nh, nw, S, norm; double inv_scale_x, inv_scale_y; };
void launch_yv12_crop(const Yv12Args&, hipStream_t);
void launch_resize_u8(const ResizeArgs&, hipStream_t);

// ---- feature export: NHWC dtype -> NCHW fp32 ----
struct ExportArgs { const void* in; float* out; int B, H, W, C, bf16; };

// ---- detection filter (layers.py:264-400) ----
#define FILTER_MAX_DET 256        // max_detections cap (63 classes x 256 survivors still sort in LDS)
struct FilterArgs {
  const float* boxes; const float* scores; const float* rotation; const float* translation; const float* hand;
  int B, N, max_det; float score_thr, nms_thr;
  int K;               // classes: scores are [B][N][K]; one workgroup per (image, class)
  int any_class;       // class_specific_filter=False (layers.py:359-362): one pass per image over every anchor's best class
  int32_t* part_idx;   // K > 1: [B][K][max_det] anchors kept per class, in NMS order; part_cnt [B][K] (filter_merge_kernel reads them)
  int32_t* part_cnt;
  uint64_t* keys;      // workspace [B * K][Npow2] sort keys
  int npow2;
  float* det_boxes; float* det_scores; int32_t* det_labels; float* det_rotation; float* det_translation;
  float* det_hand; int32_t* det_index; int32_t* det_count;
};

// ---- pose errors: ADD / ADD-S of D pose pairs over P model points (eval/common.py:682-746) ----
struct PoseErrArgs {
  const float* points;                       // [P,3]
  const float* rvec_gt; const float* t_gt;   // [D,3] axis-angle (radians), [D,3]
  const float* rvec_pr; const float* t_pr;
  double* add; double* add_s;                // [D]
  int P, D, max_points;                      // max_points: 1000 in the reference (ADD-S subsampling)
};
void launch_pose_errors(const PoseErrArgs&, hipStream_t);

// ---- training side: anchor-target assignment (generators/utils/anchors.py:69-221) ----
#define AT_MAX_GT 64
struct AnchorTargetArgs {
  const float* anchors; int N;                                     // [N,4]
  const double* gt_boxes; const int32_t* gt_labels;                // [B][kmax][4] x1,y1,x2,y2 ; [B][kmax]
  const float* gt_transform; const float* gt_coords;               // [B][kmax][rt] ; [B][kmax][63] (nullable)
  const int32_t* num_gt; const int32_t* image_hw;                  // [B] ; [B][2] (height, width)
  int B, kmax, num_classes, rt; double negative_overlap, positive_overlap;
  float* labels; float* regression; float* transformation; float* coords;   // [B][N][C+1], [B][N][5], [B][N][rt+1], [B][N][64] (nullable)
};
void launch_anchor_targets(const AnchorTargetArgs&, hipStream_t);

// ---- training side: the five losses of batch_iterate (hmdegopose/loss.py:54-99), forward values ----
#define LOSS_MAX_POINTS 2048
struct LossArgs {
  const float* gt_cls; const float* cls;        // [B][N][K+1] (labels, anchor state) ; [B][N][K] scores
  const float* gt_reg; const float* reg;        // [B][N][5] ; [B][N][4]
  const float* gt_tr; const float* tr;          // [B][N][R+3+3] (rotation, translation, is_symmetric, class, state) ; [B][N][R+3]
  const float* gt_hand; const float* hand;      // [B][N][H+1] ; [B][N][H] (both nullable: hand loss 0)
  const float* points;                          // [classes][P][3]
  int B, N, K, R, H, classes, P;
  float* per_image;                             // [B][5]: classification, regression, rotation, translation, hand
  float* losses;                                // [5]: batch means, regression x 50
};
void launch_losses(const LossArgs&, hipStream_t);

void launch_stem(const StemArgs&, hipStream_t);
int stem_uses_mfma(int cout, int force = -1);      // which of the two stem kernels the plan takes (force: Knobs::stem_mfma, -1 = by width)
void launch_pw(const PwArgs&, hipStream_t);
int pw_se_variant(const PwArgs&);   // 0 none, 1 shallow, 2 deep (template parameter of pw_gemm_kernel)
void launch_pwg(const PwgArgs&, hipStream_t);
void launch_dw(const DwArgs&, hipStream_t);
void launch_pool(const PoolArgs&, hipStream_t);
void launch_mbf(const MbfArgs&, hipStream_t);
size_t mbf_lds_layout(int Cin, int CC, int k, int s, int bf16, int has_expand, int max_inside, int ts, MbfArgs* a, int kp = 0);
int mbf_mp_fits(int CC, int kp, int bf16, int max_inside, int ts);     // the multi-pass plan fits the kernel's compile-time budgets
int mbf_mp_resident(int Cin, int CC, int max_inside, int ts);          // ... with the whole tile held in registers
int mbf_max_inside(int H, int W, int k, int s, int pad_t, int pad_l, int ts);
int mbf_prepare(void);
void launch_sep(const SepArgs&, hipStream_t);
void launch_tower(const SepArgs&, hipStream_t);
#define TOWER_BIAS_MAX 384       // bias floats staged per segment: >= 16 * n-tiles of any segment (24 map tiles, 12 per header chunk)
// n-tiles (16 columns) per header segment of k_tower.hip; wider headers are split into segments.  As many as keep the
// segment's pointwise weights in LDS (TOWER_WLDS_MAX bytes, rows padded by 16 bytes) next to the depthwise weights and the
// operand slots: a workgroup that reads its weights from global memory re-fetches 10 x the bytes of its activations at
// width 160 (measured at phi 3 @ 512 b8: 125 us per tower layer for 141 MB).  bf16: at most 6 since round 6 (was 12, tuned with one
// batch in flight).  At width 64 twelve tiles are 60 KB of LDS = two 4-wave workgroups per CU for the largest streaming launch of the
// step; six are 46 KB = three (the hand header then runs as six segments, each repeating the depthwise conv of its input).  Four
// batches in flight, 18 / 12 / 9 / 6 / 4 / 3 tiles: 56.01k / 55.67k / 55.93k / 56.13k / 56.18k / 55.85k frames/s, one batch within 0.3 %;
// bit-identical (profiles/r06/p_header_segment_width_sweep.txt).
#define TOWER_WLDS_MAX (56 * 1024)
constexpr int tower_hdr_tiles(int C, bool bf16, int cap = 6) {
  if (!bf16) return 12;     // fp32 sessions: 12 tiles, weights from global memory where 12 tiles exceed 48 KB (staging 52 KB of them at
                            // width 64 left one workgroup per CU: 13.7k -> 13.3k frames/s)
  const long es = bf16 ? 2 : 4, wp = C + (bf16 ? 8 : 4), kstep = bf16 ? 32 : 16, ks = (C + kstep - 1) / kstep;
  const long fixed = 9L * C * 4 + 4L * ks * 64 * 16;                    // depthwise weights + operand slots of the 4 waves
  const long room = TOWER_WLDS_MAX < 158L * 1024 - fixed ? TOWER_WLDS_MAX : 158L * 1024 - fixed;
  const long t = room / (16 * wp * es + 64);                            // 16 weight rows + 16 bias floats per n-tile
  return t < 1 ? 1 : (t > cap ? cap : (int)t);
}
// ... of the cooperative form (tower_coop_kernel: a wave holds the weight fragments of its n-tiles in registers): bf16 at width 64 takes
// a whole 567-column hand header (36 tiles, 9 per wave x 2 k-steps x 4 registers) as ONE segment - the depthwise conv of the
// head's last map and its input are then computed / read once, not once per 12-tile chunk
// (the wide bf16 layers keep segments of up to 12 tiles: with 6 phi 3 @ 512 loses 1.8 %, 5618 -> 5519 frames/s)
constexpr int tower_coop_hdr_tiles(int C, bool bf16) { return bf16 && C == 64 ? 36 : tower_hdr_tiles(C, bf16, 12); }
int tower_prepare(void);         // raises the dynamic-LDS limit of the tower kernels (call once per device)
int tower_coop_supported(int C, int bf16);   // widths / dtypes tower_coop_kernel is instantiated for
int tower_supports(int C);       // BiFPN widths k_tower.hip is instantiated for
int tower_map_tiles(int C);      // n-tiles (even) of a map layer: its weight rows are permuted, see k_tower.hip
void launch_decode(const DecodeArgs&, hipStream_t);
void launch_export(const ExportArgs&, hipStream_t);
void launch_preprocess(const PreprocArgs&, hipStream_t);
void launch_filter(const FilterArgs&, hipStream_t);
int filter_prepare(void);     // raises the dynamic-LDS limit of filter_kernel (call once per device)
void launch_amax_bf16(const void* x, int64_t n, unsigned* out /* float bits, zeroed */, hipStream_t);
int dw_blocks_per_image(int Ho, int Wo, int C, int TW);
void sep_lds_layout(int C, int bf16, int ts, int max_cols_f32, int max_cols_map, SepArgs* a, int stage_w = 0);   // stage_w: single nodes / chains of map-to-map nodes may keep their pointwise weights in LDS
int sep_w8(const SepArgs&);   // launch_sep picks the eight-wave single-node instantiation (k_sep.hip)
int sep_prepare(void);   // raises the dynamic-LDS limit of the sepconv kernels (call once per device)
