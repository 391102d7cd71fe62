// hep_host.h - host-side model description, weight pack and execution plan of libhep.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "hep_internal.h"
#include "hep_knobs.h"

namespace hep {

// ---- architecture tables (mirror of hmd_ego_pose_amd/arch.py; reference backbone.py:22-43,
//      efficientnet/utils.py:62-82,138-153,231-257, efficientnet/model.py:144-160) ----
struct MBConv { int cin, cexp, k, stride, se, cout; bool expand, skip; };
struct Arch {
  int phi, stem;
  std::vector<MBConv> blocks;
  int taps[3], tap_channels[3];
  int fpn_w, fpn_cells, head_depth;
  bool attention;
};
bool make_arch(int phi, Arch* out);
void same_pad(int n, int k, int s, int* before, int* after);

// ---- HEPW weight pack (hmd_ego_pose_amd/weights.py) ----
struct PackTensor { std::vector<int64_t> dims; const float* data; size_t count; };
struct Pack {
  std::vector<unsigned char> storage;
  std::unordered_map<std::string, PackTensor> tensors;
  bool parse(const void* blob, size_t n, std::string* err);
  const PackTensor* get(const std::string& name, std::initializer_list<int64_t> dims, std::string* err) const;
};

// ---- plan ----
struct TensorDesc {
  std::string name;
  int H, W, C; bool f32;            // NHWC; f32 forces fp32 storage (head outputs, SE buffers)
  size_t bytes_per_image; size_t offset;
  int first_op = -1, last_op = -1;
  int C_logical = 0;                // channels the layer has when the allocation is padded (0: C); frag: stored in the project GEMM's fragment
  bool frag = false;                // order (k_pw_impl.h FRAG: 16-byte units [m / 16][k-step][lane]) instead of NHWC - hep_debug_tensor undoes both
  bool external = false;            // lives in its own allocation (outputs)
  void* ext_ptr = nullptr;
};

enum OpKind { OP_STEM, OP_PW, OP_DW, OP_POOL, OP_SEP, OP_MBF, OP_PWG, OP_CHAIN, OP_SE, OP_XBF, OP_SBF, OP_LATE, OP_HEADS };
struct Op {
  OpKind kind; std::string name;
  StemArgs stem; PwArgs pw; DwArgs dw; PoolArgs pool; SepArgs sep; MbfArgs mbf; PwgArgs pwg; ChainArgs chain; SeFinishArgs se; XbfArgs xbf;
#ifdef HEP_ALT          // the alternative library's three extra kernels (k_sbf.hip, k_late.hip, k_heads.hip)
  SbfArgs sbf; LateArgs late; HeadsArgs heads;
#endif
  std::vector<SepSeg> segs;         // host copy (device copy uploaded at build)
  std::vector<ChainNode> cnodes;    // host copy of a chain's node table
  std::vector<int> reads, writes;   // tensor ids
  double act_bytes_per_image = 0, flops_per_image = 0, weight_bytes = 0;
};

struct Session {
  Arch arch; int size, max_batch, dtype, device; unsigned flags;
  Knobs knobs;                      // the plan knobs of the environment this session was created in (hep_knobs.h)
  // The batch is cut into `lanes` contiguous slices of lane_batch frames; every lane owns an arena and a
  // patched copy of the plan, and the captured hipGraph runs the lanes as parallel branches: the
  // forward is a chain of ~100 small latency-bound kernels, and independent chains overlap on the chip.
  int lanes = 1, lane_batch = 1;
  std::vector<std::vector<Op>> lane_ops; std::vector<hipStream_t> lane_streams; std::vector<hipEvent_t> lane_events;
  hipEvent_t fork_event = nullptr;
  int levels[5]; int level_off[5]; int num_anchors;
  int class_specific_filter = 1;    // hep_set_class_specific_filter: 0 = one filter pass over every anchor's best class (layers.py:359-362)
  int num_classes = 1;              // columns of the classification output: the classifier header holds 9 * num_classes channels (efficientdet/model.py:393)
  int out_k(int i) const { static const int k[5] = {4, 1, 3, 3, 63}; return i == 1 ? num_classes : k[i]; }   // values per anchor of head output i
  std::vector<TensorDesc> tensors;
  std::vector<Op> ops;
  std::unordered_map<std::string, int> tensor_by_name;
  int feat_ids[5];
  // device memory
  unsigned char* d_weights = nullptr; size_t weights_bytes = 0;
  uint8_t* d_pre[2] = {nullptr, nullptr}; size_t pre_bytes[2] = {0, 0};   // hep_preprocess_i420_device: cropped BGR frames, resized frames
  hipEvent_t pre_event = nullptr; bool pre_pending = false;              // recorded behind the last launch that used d_pre: the next call's stream waits for it
  unsigned char* d_arena = nullptr; size_t arena_bytes = 0;
  float* d_out[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // regression, classification, rotation, translation_raw, hand
  float* d_in = nullptr;            // staging for host-buffer runs
  float* d_feat_nchw[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  float* d_anchors = nullptr; float* d_tanchors = nullptr;
  float* d_boxes = nullptr; float* d_trans = nullptr; float* d_cam = nullptr;
  uint64_t* d_keys = nullptr; int npow2 = 0;
  int32_t* d_part = nullptr;        // per-class survivors of the detection filter (num_classes > 1)
  float* d_det = nullptr; size_t det_floats = 0;   // staging for host-buffer filter
  float* d_stage[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // host-side inputs of hep_decode / hep_filter (never the forward's outputs)
  unsigned* d_sync = nullptr;       // meeting counters of grouped launches (k_late.hip): an allocation of its own, zeroed once - never arena memory
  hipStream_t stream = nullptr;     // handle's own stream (host API, capture, profiling)
  std::map<int, std::vector<hipGraphExec_t>> graphs;   // per batch size: one graph per lane
  std::mutex mu;
  int last_batch = 0;

  ~Session();
  void* tptr(int id, int lane = 0) const { const TensorDesc& t = tensors[id]; return t.external ? t.ext_ptr : (void*)(d_arena + (size_t)lane * arena_bytes + t.offset); }
  int lanes_for(int batch) const { return (batch + lane_batch - 1) / lane_batch; }
  int lane_count(int batch, int lane) const { return std::min(lane_batch, batch - lane * lane_batch); }
  size_t esize() const { return dtype ? 2 : 4; }      // bf16 storage in bf16 AND fp8 sessions
};

int build_session(Session* s, const Pack& pack, std::string* err);
void launch_op(const Session& s, const Op& op, int batch, hipStream_t st, const float* in, const int64_t* strides);   // batch frames starting at `in`
int run_forward(Session* s, const float* in_dev, const int64_t* strides, int batch, hipStream_t st, std::string* err);
int host_anchors(int size, std::vector<float>* anchors, std::vector<float>* tanchors);

}  // namespace hep
