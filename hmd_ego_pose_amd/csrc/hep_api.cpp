// hep_api.cpp - the C ABI of libhep.so (include/hep.h) and the forward executor:
// eager stem launch (it reads the caller's input pointer) + one captured hipGraph for
// everything behind it, replayed on the caller's stream.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <math.h>

#include <chrono>
#include <fstream>
#include <memory>

#include "hep.h"
#include "hep_host.h"

using namespace hep;

struct hep_handle { Session s; };

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPRET(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(HEP_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)


// Nothing may leave an entry point as a C++ exception (include/hep.h: "never throws"; the C# host P/Invokes these symbols, and
// ONNXRuntime - the library this one replaces - reports failures through its API, Program.cs:59-61).  Every extern "C" function
// is a function-try-block ending in one of these handlers: std::bad_alloc / length_error from a hostile weight pack, a vector
// that outgrew memory, anything else -> HEP_ERR_INTERNAL and a message in hep_last_error().
static int hep_caught() noexcept {
  try { throw; }
  catch (const std::bad_alloc&) { try { g_err = "out of host memory (std::bad_alloc)"; } catch (...) {} }
  catch (const std::exception& e) { try { g_err = std::string("internal error: ") + e.what(); } catch (...) {} }
  catch (...) { try { g_err = "internal error: unknown C++ exception"; } catch (...) {} }
  return HEP_ERR_INTERNAL;
}
#define HEP_CATCH_INT catch (...) { return hep_caught(); }
#define HEP_CATCH_VOID catch (...) { hep_caught(); }

namespace hep {

Session::~Session() {
  for (auto& g : graphs) for (hipGraphExec_t ge : g.second) hipGraphExecDestroy(ge);
  for (auto& lo : lane_ops) for (Op& o : lo) {
    if (o.kind == OP_SEP) { hipFree((void*)o.sep.segs); hipFree((void*)o.sep.tile_seg); }
    if (o.kind == OP_CHAIN) hipFree((void*)o.chain.nodes);
  }
  for (hipStream_t st : lane_streams) hipStreamDestroy(st);
  for (hipEvent_t ev : lane_events) hipEventDestroy(ev);
  if (fork_event) hipEventDestroy(fork_event);
  if (pre_event) hipEventDestroy(pre_event);
  hipFree(d_sync); hipFree(d_weights); hipFree(d_arena); hipFree(d_pre[0]); hipFree(d_pre[1]);
  for (int i = 0; i < 5; i++) { hipFree(d_out[i]); hipFree(d_feat_nchw[i]); }
  hipFree(d_in); hipFree(d_anchors); hipFree(d_tanchors); hipFree(d_boxes); hipFree(d_trans); hipFree(d_cam);
  hipFree(d_keys); hipFree(d_det); hipFree(d_part);
  for (int i = 0; i < 5; i++) hipFree(d_stage[i]);
  if (stream) hipStreamDestroy(stream);
}

void launch_op(const Session& s, const Op& op, int batch, hipStream_t st, const float* in, const int64_t* strides) {
  switch (op.kind) {
    case OP_STEM: {
      StemArgs a = op.stem; a.B = batch; a.in = in;
      a.sn = strides[0]; a.sc = strides[1]; a.sh = strides[2]; a.sw = strides[3];
      launch_stem(a, st); break;
    }
    case OP_PW: { PwArgs a = op.pw; a.M = batch * a.HW; launch_pw(a, st); break; }
    case OP_DW: { DwArgs a = op.dw; a.B = batch; launch_dw(a, st); break; }
    case OP_POOL: { PoolArgs a = op.pool; a.B = batch; launch_pool(a, st); break; }
    case OP_PWG: { PwgArgs a = op.pwg; a.B = batch; launch_pwg(a, st); break; }
    case OP_MBF: { MbfArgs a = op.mbf; a.B = batch; launch_mbf(a, st); break; }
    case OP_CHAIN: { ChainArgs a = op.chain; a.B = batch; launch_chain(a, st); break; }
    case OP_SE: { SeFinishArgs a = op.se; a.B = batch; launch_se_finish(a, st); break; }
    case OP_XBF: { XbfArgs a = op.xbf; a.B = batch; launch_xbf(a, st); break; }
#ifdef HEP_ALT
    case OP_LATE: { LateArgs a = op.late; a.B = batch; launch_late(a, st); break; }
    case OP_HEADS: { HeadsArgs a = op.heads; a.B = batch; launch_heads(a, st); break; }
    case OP_SBF: {
      SbfArgs a = op.sbf; a.B = batch; a.in = in;
      a.sn = strides[0]; a.sc = strides[1]; a.sh = strides[2]; a.sw = strides[3];
      launch_sbf(a, st); break;
    }
#endif
    case OP_SEP: {
      SepArgs a = op.sep; a.B = batch;
      if (a.direct) launch_tower(a, st);
      else launch_sep(a, st);
      break;
    }
  }
}

// forward = per lane: ops[0] (stem, eager: its input pointer belongs to the caller) + the rest inside
// ONE hipGraph whose lanes are parallel branches (fork/join captured through events)
int run_forward(Session* s, const float* in_dev, const int64_t* strides, int batch, hipStream_t st, std::string* err) {
  const int64_t S = s->size;
  const int64_t contiguous[4] = {3 * S * S, S * S, S, 1};
  if (!strides) strides = contiguous;
  const int nl = s->lanes_for(batch);
  for (int l = 0; l < nl; l++)
    launch_op(*s, s->lane_ops[l][0], s->lane_count(batch, l), st, in_dev + (int64_t)l * s->lane_batch * strides[0], strides);
  if (s->flags & HEP_FLAG_NO_GRAPH) {
    for (int l = 0; l < nl; l++)
      for (size_t i = 1; i < s->ops.size(); i++) launch_op(*s, s->lane_ops[l][i], s->lane_count(batch, l), st, nullptr, nullptr);
  } else {
    // one captured graph per lane, replayed on the lane's own stream so that the lanes' kernel
    // chains overlap on the device; the caller's stream forks into the lane streams and joins them
    auto it = s->graphs.find(batch);
    if (it == s->graphs.end()) {
      std::vector<hipGraphExec_t> execs;
      for (int l = 0; l < nl; l++) {
        hipGraph_t g; hipGraphExec_t ge;
        hipError_t e = hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) { *err = std::string("hipStreamBeginCapture: ") + hipGetErrorString(e); return HEP_ERR_DEVICE; }
        for (size_t i = 1; i < s->ops.size(); i++) launch_op(*s, s->lane_ops[l][i], s->lane_count(batch, l), s->stream, nullptr, nullptr);
        e = hipStreamEndCapture(s->stream, &g);
        if (e != hipSuccess) { *err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); return HEP_ERR_DEVICE; }
        e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (e != hipSuccess) { *err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e); return HEP_ERR_DEVICE; }
        execs.push_back(ge);
      }
      it = s->graphs.emplace(batch, execs).first;
    }
    hipError_t e = hipSuccess;
    if (nl == 1) e = hipGraphLaunch(it->second[0], st);
    else {
      e = hipEventRecord(s->fork_event, st);
      for (int l = 0; l < nl && e == hipSuccess; l++) {
        hipStream_t ls = s->lane_streams[l];
        e = hipStreamWaitEvent(ls, s->fork_event, 0);
        if (e == hipSuccess) e = hipGraphLaunch(it->second[l], ls);
        if (e == hipSuccess) e = hipEventRecord(s->lane_events[l], ls);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, s->lane_events[l], 0);
      }
    }
    if (e != hipSuccess) { *err = std::string("hipGraphLaunch: ") + hipGetErrorString(e); return HEP_ERR_DEVICE; }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { *err = std::string("kernel launch: ") + hipGetErrorString(e); return HEP_ERR_DEVICE; }
  s->last_batch = batch;
  return 0;
}

}  // namespace hep

// fp8 sessions: one power-of-two scale per quantised GEMM input, from the amax of that tensor on a deterministic
// calibration batch (pseudo-normal frames) run eagerly through the session's own kernels: the scale of a launch is set
// before the launch runs, so every later tensor already sees the quantised arithmetic in front of it.  The e4m3
// conversion saturates at +-448 * scale, and the scale carries 2x headroom over the calibration amax.
static int calibrate_fp8(Session& s, const float* frames_dev = nullptr, int nframes = 0) {
  HIPRET(hipSetDevice(s.device));
  const int nb = frames_dev ? std::min(s.lane_batch, nframes) : std::min(s.lane_batch, 2);
  const size_t in_floats = (size_t)3 * s.size * s.size;
  if (!s.d_in) HIPRET(hipMalloc((void**)&s.d_in, in_floats * s.max_batch * 4));
  if (frames_dev) HIPRET(hipMemcpy(s.d_in, frames_dev, in_floats * nb * 4, hipMemcpyDeviceToDevice));
  else {
    std::vector<float> x(in_floats * nb);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto u01 = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return ((st >> 11) + 0.5) * (1.0 / 9007199254740992.0); };
    for (size_t i = 0; i + 1 < x.size(); i += 2) {          // Box-Muller
      const double r = sqrt(-2.0 * log(u01())), t = 6.283185307179586 * u01();
      x[i] = (float)(r * cos(t)); x[i + 1] = (float)(r * sin(t));
    }
    HIPRET(hipMemcpy(s.d_in, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  }
  unsigned* d_m = nullptr;
  HIPRET(hipMalloc((void**)&d_m, 4));
  const int64_t S = s.size; const int64_t strides[4] = {3 * S * S, S * S, S, 1};
  for (size_t i = 0; i < s.ops.size(); i++) {
    Op& o = s.ops[i];
    const bool q = (o.kind == OP_PW && o.pw.fp8) || (o.kind == OP_MBF && o.mbf.fp8);
    if (q) {
      const TensorDesc& t = s.tensors[o.reads[0]];
      hipMemsetAsync(d_m, 0, 4, s.stream);
      launch_amax_bf16(s.tptr(o.reads[0], 0), (int64_t)nb * t.H * t.W * t.C, d_m, s.stream);
      unsigned bits = 0;
      if (hipMemcpyAsync(&bits, d_m, 4, hipMemcpyDeviceToHost, s.stream) != hipSuccess || hipStreamSynchronize(s.stream) != hipSuccess) {
        hipFree(d_m); return fail(HEP_ERR_DEVICE, "fp8 calibration failed");
      }
      float amax; memcpy(&amax, &bits, 4);
      float sc = 1.f;
      if (amax > 0.f && amax < 3e38f) sc = exp2f(ceilf(log2f(2.f * amax / 448.f)));
      for (auto& lo : s.lane_ops) { if (o.kind == OP_PW) lo[i].pw.a_scale = sc; else lo[i].mbf.a_scale = sc; }
      if (o.kind == OP_PW) o.pw.a_scale = sc; else o.mbf.a_scale = sc;
    }
    launch_op(s, s.lane_ops[0][i], nb, s.stream, s.d_in, strides);
  }
  hipError_t e = hipStreamSynchronize(s.stream);
  hipFree(d_m);
  if (e != hipSuccess || (e = hipGetLastError()) != hipSuccess) return fail(HEP_ERR_DEVICE, std::string("fp8 calibration: ") + hipGetErrorString(e));
  // the scales travel by value in the launch arguments: graphs captured with the old ones are dropped
  for (auto& g : s.graphs) for (hipGraphExec_t ge : g.second) hipGraphExecDestroy(ge);
  s.graphs.clear();
  return 0;
}

extern "C" {

int hep_abi_version(void) { return HEP_ABI_VERSION; }
const char* hep_build_info(void) {
  return "libhep gfx950"
#ifdef HEP_ALT
         " alt"
#endif
#ifdef HEP_WITH_FP8
         " fp8"
#endif
#ifdef HEP_POISON_LDS
         " poison"
#endif
#if defined(HEP_MBF_TRACE) || defined(HEP_TOWER_TRACE) || defined(HEP_PW_TRACE) || defined(HEP_XBF_TRACE)
         " trace"
#endif
      ;
}
const char* hep_last_error(void) { return g_err.c_str(); }

int hep_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
} HEP_CATCH_INT

int hep_create_from_memory(const void* pack, size_t pack_bytes, int phi, int size, int max_batch, int dtype, int device,
                           unsigned flags, hep_handle** out) try {
  if (!out) return fail(HEP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!pack || pack_bytes < 12) return fail(HEP_ERR_PACK, "weight pack: empty");
  if (dtype != HEP_F32 && dtype != HEP_BF16 && dtype != HEP_FP8) return fail(HEP_ERR_INVALID, "dtype must be HEP_F32, HEP_BF16 or HEP_FP8");
#ifndef HEP_WITH_FP8
  if (dtype == HEP_FP8) return fail(HEP_ERR_UNSUPPORTED, "this libhep.so was built without the fp8 path (make -C hmd_ego_pose_amd/csrc FP8=1): it measured slower than bf16");
#endif
  if (max_batch < 1 || max_batch > 4096) return fail(HEP_ERR_INVALID, "max_batch out of range (1..4096)");
  if (size < 128 || size > 2048 || size % 128 != 0)
    return fail(HEP_ERR_UNSUPPORTED, "size must be a multiple of 128 in [128, 2048] (P7 has stride 128)");
  std::unique_ptr<hep_handle> h(new hep_handle);
  Session& s = h->s;
  if (!make_arch(phi, &s.arch)) return fail(HEP_ERR_UNSUPPORTED, "phi must be in 0..7 (phi 8 needs a P8 level)");
  Pack pk; std::string err;                       // host-only: a malformed pack is reported before any device is touched
  if (!pk.parse(pack, pack_bytes, &err)) return fail(HEP_ERR_PACK, err);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(HEP_ERR_DEVICE, "no HIP device visible: libhep has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(HEP_ERR_INVALID, "device index out of range");
  hipDeviceProp_t prop;
  HIPRET(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(HEP_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", libhep is built for gfx950 (MI355X) only");
  s.size = size; s.max_batch = max_batch; s.dtype = dtype; s.device = device; s.flags = flags;
  {   // lanes: slices of the batch that run as parallel graph branches (HEP_LANES overrides)
    s.knobs = read_knobs();
    int lanes = s.knobs.lanes;   // 1: measured on MI355X at bs16, 2/4/8 lanes are 4 % / 75 % / 150 % SLOWER (kernels contend instead of overlapping)
    lanes = std::max(1, std::min(lanes, std::min(max_batch, 16)));
    s.lane_batch = (max_batch + lanes - 1) / lanes;
    s.lanes = (max_batch + s.lane_batch - 1) / s.lane_batch;
  }
  int rc = build_session(&s, pk, &err);
  if (rc != 0) return fail(rc, err);
  if (dtype == HEP_FP8) if (int rc2 = calibrate_fp8(s)) return rc2;
  *out = h.release();
  return 0;
} HEP_CATCH_INT

int hep_create(const char* pack_path, int phi, int size, int max_batch, int dtype, int device, unsigned flags, hep_handle** out) try {
  if (out) *out = nullptr;
  if (!pack_path) return fail(HEP_ERR_INVALID, "pack_path is NULL");
  std::ifstream f(pack_path, std::ios::binary | std::ios::ate);
  if (!f) return fail(HEP_ERR_PACK, std::string("cannot open weight pack '") + pack_path + "'");
  const std::streamsize n = f.tellg();
  f.seekg(0);
  std::vector<char> buf((size_t)n);
  if (!f.read(buf.data(), n)) return fail(HEP_ERR_PACK, "cannot read weight pack");
  return hep_create_from_memory(buf.data(), buf.size(), phi, size, max_batch, dtype, device, flags, out);
} HEP_CATCH_INT

void hep_destroy(hep_handle* h) try {
  if (!h) return;
  hipSetDevice(h->s.device);
  hipDeviceSynchronize();
#ifdef HEP_POISON_LDS
  {   // sanitizer build: the bytes between the end of every activation tensor and the next one still hold the 0xFF the arena was
      // created with - a kernel that stored past the end of its output is named here and the process stops
    const Session& s = h->s;
    std::vector<unsigned char> host(s.arena_bytes * s.lanes);
    if (!host.empty() && hipMemcpy(host.data(), s.d_arena, host.size(), hipMemcpyDeviceToHost) == hipSuccess)
      for (int lane = 0; lane < s.lanes; lane++)
        for (const TensorDesc& t : s.tensors) {
          if (t.external || t.first_op < 0) continue;
          const size_t used = t.bytes_per_image * s.lane_batch, end = ((used + 255) & ~(size_t)255) + 256;
          for (size_t i = used; i < end; i++)
            if (host[(size_t)lane * s.arena_bytes + t.offset + i] != 0xFF) {
              fprintf(stderr, "libhep sanitizer: tensor '%s' (lane %d) was written %zu byte(s) past its end\n", t.name.c_str(), lane, i - used + 1);
              abort();
            }
        }
  }
#endif
  delete h;
} HEP_CATCH_VOID

int hep_num_anchors(const hep_handle* h) try { return h ? h->s.num_anchors : fail(HEP_ERR_INVALID, "handle is NULL"); } HEP_CATCH_INT
int hep_num_classes(const hep_handle* h) try { return h ? h->s.num_classes : fail(HEP_ERR_INVALID, "handle is NULL"); } HEP_CATCH_INT

int hep_output_shape(const hep_handle* h, int index, int batch, int64_t dims[4], int* ndim) try {
  if (!h || !dims || index < 0 || index >= HEP_NUM_OUTPUTS) return fail(HEP_ERR_INVALID, "bad argument");
  const Session& s = h->s;
  if (index < 5) { dims[0] = batch; dims[1] = s.arch.fpn_w; dims[2] = s.levels[index]; dims[3] = s.levels[index]; if (ndim) *ndim = 4; }
  else { dims[0] = batch; dims[1] = s.num_anchors; dims[2] = s.out_k(index - 5); dims[3] = 1; if (ndim) *ndim = 3; }
  return 0;
} HEP_CATCH_INT

int hep_output_device(const hep_handle* h, int index, float** ptr) try {
  if (!h || !ptr || index < 5 || index >= HEP_NUM_OUTPUTS) return fail(HEP_ERR_INVALID, "hep_output_device: index must be one of the five head outputs");
  *ptr = h->s.d_out[index - 5];
  return 0;
} HEP_CATCH_INT

static int check_run(hep_handle* h, const void* input, int batch) {
  if (!h) return fail(HEP_ERR_INVALID, "handle is NULL");
  if (!input) return fail(HEP_ERR_INVALID, "input is NULL");
  if (batch < 1 || batch > h->s.max_batch) return fail(HEP_ERR_UNSUPPORTED, "batch outside 1..max_batch given to hep_create");
  return 0;
}

static int export_feats(Session& s, int batch, float* const feats[5], bool device_dst, hipStream_t st) {
  if (!feats) return 0;
  for (int l = 0; l < 5; l++) {
    if (!feats[l]) continue;
    const TensorDesc& t = s.tensors[s.feat_ids[l]];
    const size_t n = (size_t)batch * t.H * t.W * t.C;
    float* dst = feats[l];
    if (!device_dst) {
      if (!s.d_feat_nchw[l]) HIPRET(hipMalloc((void**)&s.d_feat_nchw[l], (size_t)s.max_batch * t.H * t.W * t.C * 4));
      dst = s.d_feat_nchw[l];
    }
    for (int ln = 0; ln < s.lanes_for(batch); ln++) {
      ExportArgs a; a.in = s.tptr(s.feat_ids[l], ln); a.out = dst + (size_t)ln * s.lane_batch * t.H * t.W * t.C;
      a.B = s.lane_count(batch, ln); a.H = t.H; a.W = t.W; a.C = t.C; a.bf16 = s.dtype;
      launch_export(a, st);
    }
    if (!device_dst) HIPRET(hipMemcpyAsync(feats[l], dst, n * 4, hipMemcpyDeviceToHost, st));
  }
  return 0;
}

int hep_run_device(hep_handle* h, const float* input, const int64_t in_strides[4], int batch, float* const outs[5],
                   float* const feats[5], void* stream) try {
  if (int rc = check_run(h, input, batch)) return rc;
  Session& s = h->s;
  std::lock_guard<std::mutex> lk(s.mu);
  HIPRET(hipSetDevice(s.device));
  hipStream_t st = (hipStream_t)stream;
  std::string err;
  if (int rc = run_forward(&s, input, in_strides, batch, st, &err)) return fail(rc, err);
  if (outs)
    for (int i = 0; i < 5; i++)
      if (outs[i] && outs[i] != s.d_out[i])
        HIPRET(hipMemcpyAsync(outs[i], s.d_out[i], (size_t)batch * s.num_anchors * s.out_k(i) * 4, hipMemcpyDeviceToDevice, st));
  return export_feats(s, batch, feats, true, st);
} HEP_CATCH_INT

int hep_run(hep_handle* h, const float* input_nchw, int batch, float* const feats[5], float* regression, float* classification,
            float* rotation, float* translation_raw, float* hand) try {
  if (int rc = check_run(h, input_nchw, batch)) return rc;
  Session& s = h->s;
  std::lock_guard<std::mutex> lk(s.mu);
  HIPRET(hipSetDevice(s.device));
  const size_t in_floats = (size_t)3 * s.size * s.size;
  if (!s.d_in) HIPRET(hipMalloc((void**)&s.d_in, in_floats * s.max_batch * 4));
  HIPRET(hipMemcpyAsync(s.d_in, input_nchw, in_floats * batch * 4, hipMemcpyHostToDevice, s.stream));
  std::string err;
  if (int rc = run_forward(&s, s.d_in, nullptr, batch, s.stream, &err)) return fail(rc, err);
  float* outs[5] = {regression, classification, rotation, translation_raw, hand};
  for (int i = 0; i < 5; i++)
    if (outs[i]) HIPRET(hipMemcpyAsync(outs[i], s.d_out[i], (size_t)batch * s.num_anchors * s.out_k(i) * 4, hipMemcpyDeviceToHost, s.stream));
  if (int rc = export_feats(s, batch, feats, false, s.stream)) return rc;
  HIPRET(hipStreamSynchronize(s.stream));
  return 0;
} HEP_CATCH_INT

int hep_anchors(int size, float* anchors, float* translation_anchors) try {
  if (size < 128 || size % 128 != 0) return fail(HEP_ERR_UNSUPPORTED, "size must be a positive multiple of 128");
  std::vector<float> a, t;
  const int n = host_anchors(size, anchors ? &a : nullptr, translation_anchors ? &t : nullptr);
  if (anchors) memcpy(anchors, a.data(), a.size() * 4);
  if (translation_anchors) memcpy(translation_anchors, t.data(), t.size() * 4);
  return n;
} HEP_CATCH_INT

int hep_preprocess_u8_device(hep_handle* h, const uint8_t* rgb_hwc, int batch, int height, int width, float* out_hwc, void* stream) try {
  if (!h || !rgb_hwc || !out_hwc || batch < 1 || height < 1 || width < 1) return fail(HEP_ERR_INVALID, "bad argument");
  Session& s = h->s;
  HIPRET(hipSetDevice(s.device));
  PreprocArgs a; a.in = rgb_hwc; a.out = out_hwc; a.B = batch; a.H = height; a.W = width; a.S = s.size;
  // colibri_common.py:631-642: scale = image_size / longer side; that side becomes image_size, the other int(side * scale);
  // cv2.resize(image, (resized_width, resized_height)): with an explicit size the per-axis inverse scale is src / dst
  const int side = std::max(height, width);
  a.resize = side != s.size;
  const double scale = (double)s.size / side;
  a.nh = height > width ? s.size : (int)(height * scale);
  a.nw = height > width ? (int)(width * scale) : s.size;
  if (!a.resize) { a.nh = height; a.nw = width; }
  a.inv_scale_x = (double)width / std::max(a.nw, 1); a.inv_scale_y = (double)height / std::max(a.nh, 1);
  if (a.nh > s.size || a.nw > s.size || a.nh < 1 || a.nw < 1) return fail(HEP_ERR_UNSUPPORTED, "preprocess: resized frame does not fit the network size");
  launch_preprocess(a, (hipStream_t)stream);
  HIPRET(hipGetLastError());
  return 0;
} HEP_CATCH_INT

int hep_preprocess_i420_device(hep_handle* h, const uint8_t* yuv, int batch, int height, int width, int crop, int resized,
                               float* out_hwc, void* stream) try {
  if (!h || !yuv || !out_hwc || batch < 1) return fail(HEP_ERR_INVALID, "bad argument");
  if (height < 2 || width < 2 || (height & 1) || (width & 1)) return fail(HEP_ERR_INVALID, "4:2:0 frames have even sides");
  if (crop < 1 || crop > height || crop > width || resized < 1) return fail(HEP_ERR_INVALID, "crop must fit the frame, resized must be positive");
  Session& s = h->s;
  HIPRET(hipSetDevice(s.device));
  std::lock_guard<std::mutex> lk(s.mu);          // the two scratch buffers belong to the handle
  hipStream_t st = (hipStream_t)stream;
  // The scratch frames are used ASYNCHRONOUSLY on the caller's stream and the mutex only covers the enqueue: a second call on
  // another stream (an in-flight pool) must not overwrite them while the first call's kernels still read them.  An event is
  // recorded behind the last launch of every call and the next call's stream waits for it first (device-side, no host stall).
  if (!s.pre_event) HIPRET(hipEventCreateWithFlags(&s.pre_event, hipEventDisableTiming));
  // sized once for the handle's max_batch frames of this geometry; growing them (a larger geometry later) waits for the
  // last user first - hipFree would stall every stream of the device anyway, so it is kept off the steady state
  const int cap = batch > s.max_batch ? batch : s.max_batch;
  const size_t need[2] = {(size_t)batch * crop * crop * 3, (size_t)batch * resized * resized * 3};
  const size_t want[2] = {(size_t)cap * crop * crop * 3, (size_t)cap * resized * resized * 3};
  for (int i = 0; i < 2; i++)
    if (s.pre_bytes[i] < need[i]) {
      if (s.pre_pending) { HIPRET(hipEventSynchronize(s.pre_event)); s.pre_pending = false; }
      hipFree(s.d_pre[i]); s.d_pre[i] = nullptr; s.pre_bytes[i] = 0;
      HIPRET(hipMalloc((void**)&s.d_pre[i], want[i])); s.pre_bytes[i] = want[i];
    }
  // (the handle-owned event would become a captured event: a later wait outside the capture may fail - this entry point is
  //  not for stream capture)
  {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return fail(HEP_ERR_UNSUPPORTED, "hep_preprocess_i420_device must not be called under stream capture");
  }
  if (s.pre_pending) HIPRET(hipStreamWaitEvent(st, s.pre_event, 0));
  // from the first launch on, EVERY return path leaves the event behind the scratch frames' last user
  struct Guard { Session& s; hipStream_t st; ~Guard() { if (hipEventRecord(s.pre_event, st) == hipSuccess) s.pre_pending = true; } } guard{s, st};
  // Program.cs:161 cvtColor + 383-397 CenterCropAndRescaleMat
  Yv12Args ya; ya.in = yuv; ya.bgr = s.d_pre[0]; ya.B = batch; ya.H = height; ya.W = width; ya.crop = crop;
  ya.ow = (width - crop) / 2; ya.oh = (height - crop) / 2;
  launch_yv12_crop(ya, st);
  ResizeArgs r1; r1.in = s.d_pre[0]; r1.out = s.d_pre[1]; r1.B = batch; r1.H = crop; r1.W = crop; r1.nh = resized; r1.nw = resized; r1.S = resized; r1.norm = 0;
  r1.inv_scale_x = (double)crop / resized; r1.inv_scale_y = r1.inv_scale_x;
  launch_resize_u8(r1, st);
  // Program.cs:399-445 ResizeAndNormalizeMat on the square resized x resized frame: scale = (float)img_size / image_width
  const float scale = (float)s.size / (float)resized;
  ResizeArgs r2; r2.in = s.d_pre[1]; r2.out = out_hwc; r2.B = batch; r2.H = resized; r2.W = resized; r2.S = s.size; r2.norm = 1;
  r2.nw = s.size; r2.nh = (int)((float)resized * scale);
  if (r2.nh < 1 || r2.nh > s.size) return fail(HEP_ERR_UNSUPPORTED, "preprocess: resized frame does not fit the network size");
  r2.inv_scale_x = (double)resized / r2.nw; r2.inv_scale_y = (double)resized / r2.nh;
  launch_resize_u8(r2, st);
  HIPRET(hipGetLastError());
  return 0;
} HEP_CATCH_INT

// ---- decode / filter ----
// The *_locked helpers expect s.mu to be held; the device entry points take it around the launch, the host
// entry points around staging + launch + copy-back + synchronise, so that a concurrent call on the same handle
// (the C# frame callbacks re-enter from WebRTC worker threads) can never see another caller's staged tensors.
// Host inputs are staged in buffers of their own (d_stage), never in the forward's output buffers.
static int decode_locked(Session& s, const float* regression, const float* translation_raw, const float* camera, int batch,
                         float* boxes, float* translation, hipStream_t st) {
  DecodeArgs a;
  a.regression = regression ? regression : s.d_out[0];         // NULL = the handle's own last outputs
  a.translation_raw = translation_raw ? translation_raw : s.d_out[3];
  a.camera = camera; a.anchors = s.d_anchors; a.t_anchors = s.d_tanchors; a.boxes = boxes; a.translation = translation;
  a.B = batch; a.N = s.num_anchors; a.clip_max = (float)(s.size - 1);
  launch_decode(a, st);
  HIPRET(hipGetLastError());
  return 0;
}

int hep_decode_device(hep_handle* h, const float* regression, const float* translation_raw, const float* camera, int batch,
                      float* boxes, float* translation, void* stream) try {
  if (!h || !camera || !boxes || !translation) return fail(HEP_ERR_INVALID, "bad argument");
  Session& s = h->s;
  if (batch < 1 || batch > s.max_batch) return fail(HEP_ERR_UNSUPPORTED, "batch outside 1..max_batch");
  std::lock_guard<std::mutex> lk(s.mu);
  HIPRET(hipSetDevice(s.device));
  return decode_locked(s, regression, translation_raw, camera, batch, boxes, translation, (hipStream_t)stream);
} HEP_CATCH_INT

static int ensure_post(Session& s) {
  const size_t n = (size_t)s.max_batch * s.num_anchors;
  if (!s.d_boxes) HIPRET(hipMalloc((void**)&s.d_boxes, n * 4 * 4));
  if (!s.d_trans) HIPRET(hipMalloc((void**)&s.d_trans, n * 3 * 4));
  if (!s.d_cam) HIPRET(hipMalloc((void**)&s.d_cam, (size_t)s.max_batch * 6 * 4));
  if (!s.d_keys) {
    s.npow2 = 1; while (s.npow2 < s.num_anchors) s.npow2 <<= 1;
    HIPRET(hipMalloc((void**)&s.d_keys, (size_t)s.max_batch * s.num_classes * s.npow2 * 8));
  }
  if (s.num_classes > 1 && !s.d_part) HIPRET(hipMalloc((void**)&s.d_part, (size_t)s.max_batch * s.num_classes * (FILTER_MAX_DET + 1) * 4));
  return 0;
}
// staging buffer i (0 regression .. 4 hand, same widths as the outputs) for host-side inputs
static int ensure_stage(Session& s, int i) {
  if (!s.d_stage[i]) HIPRET(hipMalloc((void**)&s.d_stage[i], (size_t)s.max_batch * s.num_anchors * s.out_k(i) * 4));
  return 0;
}

int hep_decode(hep_handle* h, const float* regression, const float* translation_raw, const float* camera, int batch,
               float* boxes, float* translation) try {
  if (!h || !camera || !boxes || !translation) return fail(HEP_ERR_INVALID, "bad argument");
  Session& s = h->s;
  if (batch < 1 || batch > s.max_batch) return fail(HEP_ERR_UNSUPPORTED, "batch outside 1..max_batch");
  std::lock_guard<std::mutex> lk(s.mu);
  HIPRET(hipSetDevice(s.device));
  if (int rc = ensure_post(s)) return rc;
  const size_t n = (size_t)batch * s.num_anchors;
  const float *dreg = nullptr, *dtrn = nullptr;
  if (regression) {
    if (int rc = ensure_stage(s, 0)) return rc;
    HIPRET(hipMemcpyAsync(s.d_stage[0], regression, n * 16, hipMemcpyHostToDevice, s.stream)); dreg = s.d_stage[0];
  }
  if (translation_raw) {
    if (int rc = ensure_stage(s, 3)) return rc;
    HIPRET(hipMemcpyAsync(s.d_stage[3], translation_raw, n * 12, hipMemcpyHostToDevice, s.stream)); dtrn = s.d_stage[3];
  }
  HIPRET(hipMemcpyAsync(s.d_cam, camera, (size_t)batch * 24, hipMemcpyHostToDevice, s.stream));
  if (int rc = decode_locked(s, dreg, dtrn, s.d_cam, batch, s.d_boxes, s.d_trans, s.stream)) return rc;
  HIPRET(hipMemcpyAsync(boxes, s.d_boxes, n * 16, hipMemcpyDeviceToHost, s.stream));
  HIPRET(hipMemcpyAsync(translation, s.d_trans, n * 12, hipMemcpyDeviceToHost, s.stream));
  HIPRET(hipStreamSynchronize(s.stream));
  return 0;
} HEP_CATCH_INT

static int filter_locked(Session& s, const float* boxes, const float* classification, const float* rotation,
                         const float* translation, const float* hand, int batch, float score_threshold, float nms_threshold,
                         int max_detections, float* det_boxes, float* det_scores, int32_t* det_labels, float* det_rotation,
                         float* det_translation, float* det_hand, int32_t* det_index, int32_t* det_count, hipStream_t st) {
  if (int rc = ensure_post(s)) return rc;
  FilterArgs a;
  a.boxes = boxes; a.scores = classification ? classification : s.d_out[1];
  a.rotation = rotation ? rotation : s.d_out[2]; a.translation = translation ? translation : s.d_trans;
  a.hand = hand ? hand : s.d_out[4];
  a.B = batch; a.N = s.num_anchors; a.max_det = max_detections; a.score_thr = score_threshold; a.nms_thr = nms_threshold;
  a.keys = s.d_keys; a.npow2 = s.npow2;
  a.K = s.num_classes; a.any_class = s.class_specific_filter ? 0 : 1; a.part_idx = s.d_part; a.part_cnt = s.d_part ? s.d_part + (size_t)s.max_batch * s.num_classes * FILTER_MAX_DET : nullptr;
  a.det_boxes = det_boxes; a.det_scores = det_scores; a.det_labels = det_labels; a.det_rotation = det_rotation;
  a.det_translation = det_translation; a.det_hand = det_hand; a.det_index = det_index; a.det_count = det_count;
  launch_filter(a, st);
  HIPRET(hipGetLastError());
  return 0;
}

int hep_set_class_specific_filter(hep_handle* h, int on) try {
  if (!h) return fail(HEP_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lk(h->s.mu);
  h->s.class_specific_filter = on ? 1 : 0;
  return 0;
} HEP_CATCH_INT

int hep_filter_device(hep_handle* h, const float* boxes, const float* classification, const float* rotation,
                      const float* translation, const float* hand, int batch, float score_threshold, float nms_threshold,
                      int max_detections, float* det_boxes, float* det_scores, int32_t* det_labels, float* det_rotation,
                      float* det_translation, float* det_hand, int32_t* det_index, int32_t* det_count, void* stream) try {
  if (!h || !boxes || !det_count) return fail(HEP_ERR_INVALID, "bad argument");
  Session& s = h->s;
  if (batch < 1 || batch > s.max_batch) return fail(HEP_ERR_UNSUPPORTED, "batch outside 1..max_batch");
  if (max_detections < 1 || max_detections > 256) return fail(HEP_ERR_UNSUPPORTED, "max_detections must be in 1..256");
  std::lock_guard<std::mutex> lk(s.mu);
  HIPRET(hipSetDevice(s.device));
  return filter_locked(s, boxes, classification, rotation, translation, hand, batch, score_threshold, nms_threshold, max_detections,
                       det_boxes, det_scores, det_labels, det_rotation, det_translation, det_hand, det_index, det_count, (hipStream_t)stream);
} HEP_CATCH_INT

int hep_filter(hep_handle* h, const float* boxes, const float* classification, const float* rotation, const float* translation,
               const float* hand, int batch, float score_threshold, float nms_threshold, int max_detections, float* det_boxes,
               float* det_scores, int32_t* det_labels, float* det_rotation, float* det_translation, float* det_hand,
               int32_t* det_index, int32_t* det_count) try {
  if (!h || !boxes || !classification || !rotation || !translation || !hand || !det_count) return fail(HEP_ERR_INVALID, "bad argument");
  Session& s = h->s;
  if (batch < 1 || batch > s.max_batch) return fail(HEP_ERR_UNSUPPORTED, "batch outside 1..max_batch");
  if (max_detections < 1 || max_detections > 256) return fail(HEP_ERR_UNSUPPORTED, "max_detections must be in 1..256");
  const size_t n = (size_t)batch * s.num_anchors, M = (size_t)batch * max_detections;
  std::lock_guard<std::mutex> lk(s.mu);
  HIPRET(hipSetDevice(s.device));
  if (int rc = ensure_post(s)) return rc;
  for (int i = 0; i < 5; i++) if (int rc = ensure_stage(s, i)) return rc;
  const size_t need = (size_t)s.max_batch * 256 * (4 + 1 + 1 + 3 + 3 + 63 + 1) + s.max_batch;
  if (!s.d_det) { HIPRET(hipMalloc((void**)&s.d_det, need * 4)); s.det_floats = need; }
  float* d = s.d_det;
  // staged inputs: boxes -> stage 0 (same width as regression), scores 1, rotation 2, translation 3, hand 4
  HIPRET(hipMemcpyAsync(s.d_stage[0], boxes, n * 16, hipMemcpyHostToDevice, s.stream));
  HIPRET(hipMemcpyAsync(s.d_stage[1], classification, n * s.num_classes * 4, hipMemcpyHostToDevice, s.stream));
  HIPRET(hipMemcpyAsync(s.d_stage[2], rotation, n * 12, hipMemcpyHostToDevice, s.stream));
  HIPRET(hipMemcpyAsync(s.d_stage[3], translation, n * 12, hipMemcpyHostToDevice, s.stream));
  HIPRET(hipMemcpyAsync(s.d_stage[4], hand, n * 63 * 4, hipMemcpyHostToDevice, s.stream));
  float* b_ = d; float* sc_ = b_ + M * 4; int32_t* lb_ = (int32_t*)(sc_ + M); float* ro_ = (float*)(lb_ + M);
  float* tr_ = ro_ + M * 3; float* hd_ = tr_ + M * 3; int32_t* ix_ = (int32_t*)(hd_ + M * 63); int32_t* ct_ = ix_ + M;
  if (int rc = filter_locked(s, s.d_stage[0], s.d_stage[1], s.d_stage[2], s.d_stage[3], s.d_stage[4], batch, score_threshold, nms_threshold,
                             max_detections, b_, sc_, lb_, ro_, tr_, hd_, ix_, ct_, s.stream)) return rc;
  if (det_boxes) HIPRET(hipMemcpyAsync(det_boxes, b_, M * 16, hipMemcpyDeviceToHost, s.stream));
  if (det_scores) HIPRET(hipMemcpyAsync(det_scores, sc_, M * 4, hipMemcpyDeviceToHost, s.stream));
  if (det_labels) HIPRET(hipMemcpyAsync(det_labels, lb_, M * 4, hipMemcpyDeviceToHost, s.stream));
  if (det_rotation) HIPRET(hipMemcpyAsync(det_rotation, ro_, M * 12, hipMemcpyDeviceToHost, s.stream));
  if (det_translation) HIPRET(hipMemcpyAsync(det_translation, tr_, M * 12, hipMemcpyDeviceToHost, s.stream));
  if (det_hand) HIPRET(hipMemcpyAsync(det_hand, hd_, M * 63 * 4, hipMemcpyDeviceToHost, s.stream));
  if (det_index) HIPRET(hipMemcpyAsync(det_index, ix_, M * 4, hipMemcpyDeviceToHost, s.stream));
  HIPRET(hipMemcpyAsync(det_count, ct_, (size_t)batch * 4, hipMemcpyDeviceToHost, s.stream));
  HIPRET(hipStreamSynchronize(s.stream));
  return 0;
} HEP_CATCH_INT

// ---- pose errors (evaluator) ----
int hep_pose_errors_device(const float* points, int num_points, const float* rvec_gt, const float* t_gt, const float* rvec_pred,
                           const float* t_pred, int num_pairs, int max_points, double* add, double* add_s, void* stream) try {
  if (!points || !rvec_gt || !t_gt || !rvec_pred || !t_pred || !add || !add_s) return fail(HEP_ERR_INVALID, "bad argument");
  if (num_points < 1 || num_pairs < 0) return fail(HEP_ERR_INVALID, "num_points must be >= 1 and num_pairs >= 0");
  if (max_points < 1 || max_points > 1024) return fail(HEP_ERR_UNSUPPORTED, "max_points must be in 1..1024 (the reference uses 1000)");
  if (num_pairs == 0) return 0;
  PoseErrArgs a; a.points = points; a.rvec_gt = rvec_gt; a.t_gt = t_gt; a.rvec_pr = rvec_pred; a.t_pr = t_pred;
  a.add = add; a.add_s = add_s; a.P = num_points; a.D = num_pairs; a.max_points = max_points;
  launch_pose_errors(a, (hipStream_t)stream);
  HIPRET(hipGetLastError());
  return 0;
} HEP_CATCH_INT

int hep_pose_errors(int device, const float* points, int num_points, const float* rvec_gt, const float* t_gt, const float* rvec_pred,
                    const float* t_pred, int num_pairs, int max_points, double* add, double* add_s) try {
  if (!points || !rvec_gt || !t_gt || !rvec_pred || !t_pred || !add || !add_s) return fail(HEP_ERR_INVALID, "bad argument");
  if (num_points < 1 || num_pairs < 0) return fail(HEP_ERR_INVALID, "num_points must be >= 1 and num_pairs >= 0");
  if (num_pairs == 0) return 0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(HEP_ERR_DEVICE, "no HIP device visible: libhep has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(HEP_ERR_INVALID, "device index out of range");
  HIPRET(hipSetDevice(device));
  const size_t pb = (size_t)num_points * 12, db = (size_t)num_pairs * 12, ob = (size_t)num_pairs * 8;
  unsigned char* buf = nullptr;
  HIPRET(hipMalloc((void**)&buf, pb + 4 * db + 2 * ob + 64));
  struct Free { unsigned char* p; ~Free() { hipFree(p); } } guard{buf};
  float* d_pts = (float*)buf; float* d_rg = (float*)(buf + pb); float* d_tg = d_rg + 3 * num_pairs; float* d_rp = d_tg + 3 * num_pairs; float* d_tp = d_rp + 3 * num_pairs;
  double* d_add = (double*)(buf + ((pb + 4 * db + 7) & ~(size_t)7)); double* d_adds = d_add + num_pairs;
  HIPRET(hipMemcpy(d_pts, points, pb, hipMemcpyHostToDevice));
  HIPRET(hipMemcpy(d_rg, rvec_gt, db, hipMemcpyHostToDevice)); HIPRET(hipMemcpy(d_tg, t_gt, db, hipMemcpyHostToDevice));
  HIPRET(hipMemcpy(d_rp, rvec_pred, db, hipMemcpyHostToDevice)); HIPRET(hipMemcpy(d_tp, t_pred, db, hipMemcpyHostToDevice));
  if (int rc = hep_pose_errors_device(d_pts, num_points, d_rg, d_tg, d_rp, d_tp, num_pairs, max_points, d_add, d_adds, nullptr)) return rc;
  HIPRET(hipMemcpy(add, d_add, ob, hipMemcpyDeviceToHost));
  HIPRET(hipMemcpy(add_s, d_adds, ob, hipMemcpyDeviceToHost));
  return 0;
} HEP_CATCH_INT

// ---- training side ----
int hep_anchor_targets_device(const float* anchors, int num_anchors, const double* gt_boxes, const int32_t* gt_labels,
                              const float* gt_transform, const float* gt_coords, const int32_t* num_gt, const int32_t* image_hw,
                              int batch, int kmax, int num_classes, int num_transform, double negative_overlap, double positive_overlap,
                              float* labels, float* regression, float* transformation, float* coords, void* stream) try {
  if (!anchors || !gt_boxes || !gt_labels || !gt_transform || !num_gt || !image_hw || !labels || !regression || !transformation)
    return fail(HEP_ERR_INVALID, "bad argument");
  if (num_anchors < 1 || batch < 1 || num_classes < 1 || num_transform < 0) return fail(HEP_ERR_INVALID, "bad size");
  if (kmax < 1 || kmax > AT_MAX_GT) return fail(HEP_ERR_UNSUPPORTED, "kmax must be in 1..64 ground-truth boxes per image");
  AnchorTargetArgs a; a.anchors = anchors; a.N = num_anchors; a.gt_boxes = gt_boxes; a.gt_labels = gt_labels; a.gt_transform = gt_transform;
  a.gt_coords = gt_coords; a.num_gt = num_gt; a.image_hw = image_hw; a.B = batch; a.kmax = kmax; a.num_classes = num_classes; a.rt = num_transform;
  a.negative_overlap = negative_overlap; a.positive_overlap = positive_overlap;
  a.labels = labels; a.regression = regression; a.transformation = transformation; a.coords = coords;
  launch_anchor_targets(a, (hipStream_t)stream);
  HIPRET(hipGetLastError());
  return 0;
} HEP_CATCH_INT

int hep_losses_device(const float* gt_classification, const float* classification, const float* gt_regression, const float* regression,
                      const float* gt_transformation, const float* transformation, const float* gt_hand, const float* hand,
                      const float* model_points, int batch, int num_anchors, int num_classes, int num_rotation, int num_hand,
                      int num_model_classes, int num_points, float* per_image, float* losses, void* stream) try {
  if (!gt_classification || !classification || !gt_regression || !regression || !gt_transformation || !transformation || !model_points ||
      !per_image || !losses) return fail(HEP_ERR_INVALID, "bad argument");
  if ((gt_hand == nullptr) != (hand == nullptr)) return fail(HEP_ERR_INVALID, "gt_hand and hand go together");
  if (batch < 1 || num_anchors < 1 || num_classes < 1 || num_rotation != 3 || num_hand < 0 || num_model_classes < 1) return fail(HEP_ERR_INVALID, "bad size");
  if (num_points < 1 || num_points > LOSS_MAX_POINTS) return fail(HEP_ERR_UNSUPPORTED, "num_points must be in 1..2048 model points per class");
  LossArgs a; a.gt_cls = gt_classification; a.cls = classification; a.gt_reg = gt_regression; a.reg = regression; a.gt_tr = gt_transformation;
  a.tr = transformation; a.gt_hand = gt_hand; a.hand = hand; a.points = model_points; a.B = batch; a.N = num_anchors; a.K = num_classes;
  a.R = num_rotation; a.H = num_hand; a.classes = num_model_classes; a.P = num_points; a.per_image = per_image; a.losses = losses;
  launch_losses(a, (hipStream_t)stream);
  HIPRET(hipGetLastError());
  return 0;
} HEP_CATCH_INT

// ---- introspection ----
int hep_debug_tensor_count(const hep_handle* h) try { return h ? (int)h->s.tensors.size() : 0; } HEP_CATCH_INT
int hep_debug_tensor_info(const hep_handle* h, int i, const char** name, int64_t dims[4]) try {
  if (!h || i < 0 || i >= (int)h->s.tensors.size()) return fail(HEP_ERR_INVALID, "bad tensor index");
  const TensorDesc& t = h->s.tensors[i];
  if (name) *name = t.name.c_str();
  if (dims) { dims[0] = h->s.max_batch; dims[1] = t.H; dims[2] = t.W; dims[3] = t.C_logical ? t.C_logical : t.C; }
  return 0;
} HEP_CATCH_INT
int hep_debug_tensor(hep_handle* h, const char* name, int batch, float* out, size_t capacity) try {
  if (!h || !name || !out) return fail(HEP_ERR_INVALID, "bad argument");
  Session& s = h->s;
  auto it = s.tensor_by_name.find(name);
  if (it == s.tensor_by_name.end()) return fail(HEP_ERR_INVALID, std::string("no stage tensor named '") + name + "'");
  if (batch < 1 || batch > s.max_batch) return fail(HEP_ERR_UNSUPPORTED, "batch outside 1..max_batch");
  const TensorDesc& t = s.tensors[it->second];
  const int C = t.C_logical ? t.C_logical : t.C;                     // (the allocation may be padded: NHWC rows are C apart all the same)
  const size_t n = (size_t)batch * t.H * t.W * C;
  if (n > capacity) return fail(HEP_ERR_INVALID, "output buffer too small");
  std::lock_guard<std::mutex> lk(s.mu);
  HIPRET(hipSetDevice(s.device));
  HIPRET(hipDeviceSynchronize());
  const bool f32 = t.f32 || s.dtype == HEP_F32;
  const size_t per = (size_t)t.H * t.W * C, per_alloc = (size_t)t.H * t.W * t.C;
  for (int ln = 0; ln < s.lanes_for(batch); ln++) {
    const int nb = s.lane_count(batch, ln);
    float* dst = out + (size_t)ln * s.lane_batch * per;
    const size_t cnt = (size_t)nb * (t.frag ? per_alloc : per);      // elements to fetch
    std::vector<float> raw(cnt);
    if (f32) { HIPRET(hipMemcpy(raw.data(), s.tptr(it->second, ln), cnt * 4, hipMemcpyDeviceToHost)); }
    else {
      std::vector<uint16_t> tmp(cnt);
      HIPRET(hipMemcpy(tmp.data(), s.tptr(it->second, ln), cnt * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < cnt; i++) { uint32_t u = (uint32_t)tmp[i] << 16; memcpy(&raw[i], &u, 4); }
    }
    if (!t.frag) { memcpy(dst, raw.data(), cnt * 4); continue; }
    // fragment order -> NHWC: row m = (image, pixel) of this lane, channel k lives in the 16-byte unit
    // ((m / 16) * ksteps + k / KSTEP) * 64 + (k % KSTEP) / KLANE * 16 + m % 16 at element k % KLANE (k_mbf.hip ostore)
    const int kstep = f32 ? 16 : 32, klane = f32 ? 4 : 8, kst = (C + kstep - 1) / kstep;
    const size_t rows = (size_t)nb * t.H * t.W;
    for (size_t m = 0; m < rows; m++)
      for (int k = 0; k < C; k++) {
        const size_t unit = ((m >> 4) * kst + k / kstep) * 64 + (size_t)((k % kstep) / klane) * 16 + (m & 15);
        dst[m * C + k] = raw[unit * klane + k % klane];
      }
  }
  return 0;
} HEP_CATCH_INT

int hep_kernel_count(const hep_handle* h, int) try { return h ? (int)h->s.ops.size() : 0; } HEP_CATCH_INT
int hep_kernel_info(const hep_handle* h, int batch, int i, const char** name, double* bytes, double* flops) try {
  if (!h || i < 0 || i >= (int)h->s.ops.size()) return fail(HEP_ERR_INVALID, "bad kernel index");
  const Op& o = h->s.ops[i];
  if (name) *name = o.name.c_str();
  const int per_launch = std::min(batch, h->s.lane_batch);      // frames one launch of this kernel sees
  if (bytes) *bytes = o.act_bytes_per_image * per_launch + o.weight_bytes;
  if (flops) *flops = o.flops_per_image * per_launch;
  return 0;
} HEP_CATCH_INT

int hep_fp8_scale(const hep_handle* h, int i, float* a_scale) try {
  if (!h || !a_scale || i < 0 || i >= (int)h->s.ops.size()) return fail(HEP_ERR_INVALID, "bad kernel index");
  const Op& o = h->s.ops[i];
  *a_scale = (o.kind == OP_PW && o.pw.fp8) ? o.pw.a_scale : ((o.kind == OP_MBF && o.mbf.fp8) ? o.mbf.a_scale : 0.f);
  return 0;
} HEP_CATCH_INT

int hep_calibrate_fp8(hep_handle* h, const float* frames_nchw_device, int batch) try {
  if (!h || !frames_nchw_device || batch < 1) return fail(HEP_ERR_INVALID, "bad argument");
  Session& s = h->s;
  if (s.dtype != HEP_FP8) return fail(HEP_ERR_UNSUPPORTED, "hep_calibrate_fp8 needs an HEP_FP8 session");
  std::lock_guard<std::mutex> lk(s.mu);
  HIPRET(hipDeviceSynchronize());
  return calibrate_fp8(s, frames_nchw_device, batch);
} HEP_CATCH_INT

int hep_kernel_symbol(const hep_handle* h, int i, const char** symbol) try {
  if (!h || !symbol || i < 0 || i >= (int)h->s.ops.size()) return fail(HEP_ERR_INVALID, "bad kernel index");
  static thread_local std::string buf;
  const Op& o = h->s.ops[i];
  const char* t = h->s.dtype ? "true" : "false";
  const int prec = o.kind == OP_PW ? (o.pw.fp8 ? 2 : (h->s.dtype ? 1 : 0)) : 0;
  char tmp[128];
  switch (o.kind) {
    case OP_STEM: if (o.stem.mfma) snprintf(tmp, sizeof tmp, "stem_kernel<%s, %d>", t, (o.stem.Cout + 15) / 16);
                  else snprintf(tmp, sizeof tmp, "stem_valu_kernel<%s>", t);
                  break;
    case OP_PW: { const int sev = pw_se_variant(o.pw);
                  const bool w8 = o.pw.nwv == 8 && prec == 0 && o.pw.mode == 2 && o.pw.act != ACT_SWISH && (sev == 0 || sev == 3) && o.pw.NT <= 2;
                  // (the names rocprofv3 prints: bench.py looks the PMC traffic of a device function up by this string)
                  const bool fr = o.pw.frag && prec != 2 && o.pw.MT == 2 && o.pw.NT <= PW_FRAG_MAX_NT && o.pw.mode == 2 && sev >= 1 && o.pw.act != ACT_SWISH;      // (launch_nt: fragment-ordered operands)
                  snprintf(tmp, sizeof tmp, fr ? "pw_gemm_kernel<%d, %d, %d, %d, %d, %d, %d, true>" : "pw_gemm_kernel<%d, %d, %d, %d, %d, %d, %d>", prec, o.pw.MT, o.pw.NT, o.pw.mode, o.pw.act == ACT_SWISH ? 1 : 0, sev, (w8 || (fr && o.pw.nwv == 8 && prec == 0 && sev == 3 && o.pw.NT <= 2)) ? 8 : 4); break; }
    case OP_DW: snprintf(tmp, sizeof tmp, "dw_kernel<%s, %d, %d, %d>", t, o.dw.k, o.dw.s, o.dw.TW); break;
    case OP_POOL: snprintf(tmp, sizeof tmp, "pool_kernel<%s>", t); break;
    case OP_PWG: snprintf(tmp, sizeof tmp, "pw_group_kernel<%s>", t); break;
    case OP_CHAIN: snprintf(tmp, sizeof tmp, "chain_kernel<%s, %d>", o.chain.bf16 ? "true" : "false", o.chain.stream_w); break;
    case OP_SE: snprintf(tmp, sizeof tmp, "se_finish_kernel<%s>", t); break;
    case OP_MBF: snprintf(tmp, sizeof tmp, "mbf_kernel<%s, %d, %d, %d, %s, %d>", t, o.mbf.k, o.mbf.s, o.mbf.ts, o.mbf.fp8 ? "true" : "false",
                          o.mbf.has_expand && o.mbf.npass > 1 ? (o.mbf.mp_resident ? 2 : 1) : 0);      // (last argument: 1 / 2 = multi-pass expand)
                 break;
#ifdef HEP_ALT
    case OP_SBF: snprintf(tmp, sizeof tmp, "sbf_kernel<%s>", t); break;
    case OP_LATE: snprintf(tmp, sizeof tmp, "late_kernel"); break;
    case OP_HEADS: snprintf(tmp, sizeof tmp, "heads_kernel"); break;
#endif
    case OP_XBF: { const int sp = xbf_specialised(o.xbf);
                   snprintf(tmp, sizeof tmp, "xbf_kernel<%s, %d, %d, %d, %d, %d, %d, %d>", t, o.xbf.k, o.xbf.s, o.xbf.toh, o.xbf.tow, o.xbf.NT1, sp ? o.xbf.K1 : 0, sp ? o.xbf.NT2 : 0); break; }
    default: if (o.sep.direct) snprintf(tmp, sizeof tmp, "%s<%s, %d, %s>", o.sep.coop ? "tower_coop_kernel" : "tower_kernel", t, o.sep.C, o.sep.direct == 2 ? "true" : "false");
             else {
               const int mode = o.sep.chain ? 2 : (o.sep.nseg == 1 ? 0 : 1);
               const bool wl = o.sep.bf16 && o.sep.off_wpw && mode != 1;         // (launch_sep: staged pointwise weights)
               if (mode == 0 && sep_w8(o.sep)) snprintf(tmp, sizeof tmp, "sep_kernel<%s, 0, false, true>", t);
               else snprintf(tmp, sizeof tmp, wl ? "sep_kernel<%s, %d, true, false>" : "sep_kernel<%s, %d, false, false>", t, mode);
             }
             break;
  }
  buf = tmp; *symbol = buf.c_str();
  return 0;
} HEP_CATCH_INT

namespace {
struct EventSet {     // events and streams of the profilers, released on every return path
  std::vector<hipEvent_t> ev; std::vector<hipStream_t> st;
  ~EventSet() { for (hipEvent_t e : ev) hipEventDestroy(e); for (hipStream_t x : st) hipStreamDestroy(x); }
  int add_events(size_t n) { for (size_t i = 0; i < n; i++) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return -1; ev.push_back(e); } return 0; }
  int add_streams(size_t n) { for (size_t i = 0; i < n; i++) { hipStream_t x; if (hipStreamCreateWithFlags(&x, hipStreamNonBlocking) != hipSuccess) return -1; st.push_back(x); } return 0; }
};
}  // namespace

int hep_profile(hep_handle* h, int batch, int iters, float* total_ms_per_iter, float* per_kernel_ms) try {
  if (!h || iters < 1) return fail(HEP_ERR_INVALID, "bad argument");
  Session& s = h->s;
  if (batch < 1 || batch > s.max_batch) return fail(HEP_ERR_UNSUPPORTED, "batch outside 1..max_batch");
  std::lock_guard<std::mutex> lk(s.mu);
  HIPRET(hipSetDevice(s.device));
  const size_t in_floats = (size_t)3 * s.size * s.size;
  if (!s.d_in) { HIPRET(hipMalloc((void**)&s.d_in, in_floats * s.max_batch * 4)); HIPRET(hipMemset(s.d_in, 0, in_floats * s.max_batch * 4)); }
  EventSet es;
  if (es.add_events(2)) return fail(HEP_ERR_DEVICE, "hipEventCreate failed");
  hipEvent_t e0 = es.ev[0], e1 = es.ev[1];
  std::string err;
  const int64_t S = s.size; const int64_t st[4] = {3 * S * S, S * S, S, 1};
  for (int w = 0; w < 3; w++) if (int rc = run_forward(&s, s.d_in, st, batch, s.stream, &err)) return fail(rc, err);
  HIPRET(hipEventRecord(e0, s.stream));
  for (int i = 0; i < iters; i++) if (int rc = run_forward(&s, s.d_in, st, batch, s.stream, &err)) return fail(rc, err);
  HIPRET(hipEventRecord(e1, s.stream));
  HIPRET(hipEventSynchronize(e1));
  float ms = 0; HIPRET(hipEventElapsedTime(&ms, e0, e1));
  if (total_ms_per_iter) *total_ms_per_iter = ms / iters;
  if (per_kernel_ms) {
    // in-sequence timing: the forward is launched eagerly with an event in front of every kernel,
    // so each duration is taken in its real context (cold caches, real predecessor), on the stream
    // the kernel runs on.  It includes the inter-kernel boundary, like a hipGraph replay does.
    const size_t n = s.ops.size();
    if (es.add_events(n + 1)) return fail(HEP_ERR_DEVICE, "hipEventCreate failed");
    hipEvent_t* ev = es.ev.data() + 2;
    std::vector<double> acc(n, 0.0);
    for (int i = 0; i < iters + 1; i++) {
      for (size_t k = 0; k < n; k++) {     // lane 0's launches (lane_batch frames each), one after the other
        HIPRET(hipEventRecord(ev[k], s.stream));
        launch_op(s, s.lane_ops[0][k], s.lane_count(batch, 0), s.stream, s.d_in, st);
      }
      HIPRET(hipEventRecord(ev[n], s.stream));
      HIPRET(hipEventSynchronize(ev[n]));
      if (i == 0) continue;     // first pass warms the instruction caches
      for (size_t k = 0; k < n; k++) { HIPRET(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); acc[k] += ms; }
    }
    for (size_t k = 0; k < n; k++) per_kernel_ms[k] = (float)(acc[k] / iters);
  }
  return 0;
} HEP_CATCH_INT

int hep_profile_concurrent(hep_handle* h, int batch, int iters, int nstreams, float* per_kernel_ms) try {
  if (!h || iters < 1 || nstreams < 1 || nstreams > 16 || !per_kernel_ms) return fail(HEP_ERR_INVALID, "bad argument");
  Session& s = h->s;
  if (batch < 1 || batch > s.max_batch) return fail(HEP_ERR_UNSUPPORTED, "batch outside 1..max_batch");
  std::lock_guard<std::mutex> lk(s.mu);
  HIPRET(hipSetDevice(s.device));
  const size_t in_floats = (size_t)3 * s.size * s.size;
  if (!s.d_in) { HIPRET(hipMalloc((void**)&s.d_in, in_floats * s.max_batch * 4)); HIPRET(hipMemset(s.d_in, 0, in_floats * s.max_batch * 4)); }
  std::string err;
  const int64_t S = s.size; const int64_t st[4] = {3 * S * S, S * S, S, 1};
  if (int rc = run_forward(&s, s.d_in, st, batch, s.stream, &err)) return fail(rc, err);
  HIPRET(hipStreamSynchronize(s.stream));
  EventSet es;
  if (es.add_streams(nstreams)) return fail(HEP_ERR_DEVICE, "hipStreamCreate failed");
  std::vector<hipStream_t>& ss = es.st;
  const size_t n = s.ops.size();
  for (size_t k = 0; k < n; k++) {
    // the same launch repeated on every stream at once (identical inputs, identical outputs): its
    // time per launch with the chip shared is what the launch costs a pipeline of batches in flight
    for (int rep = 0; rep < 2; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      long issued = 0;
      for (int i = 0; i < (rep ? iters : 2); i++)
        for (auto& x : ss) {
          launch_op(s, s.lane_ops[0][k], s.lane_count(batch, 0), x, s.d_in, st);
          issued++;
#ifdef HEP_ALT
          if (s.lane_ops[0][k].kind == OP_LATE && s.lane_ops[0][k].late.G > 1) break;      // (grouped launches of ONE session share its meeting counters: never two at once - that row is a SERIAL cost)
#endif
        }
      for (auto& x : ss) HIPRET(hipStreamSynchronize(x));
      if (rep) per_kernel_ms[k] = (float)(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (double)std::max(1L, issued));      // per launch ISSUED
    }
  }
  return 0;
} HEP_CATCH_INT

}  // extern "C"
