// hep_model.cpp - host side of libhep.so: architecture tables, HEPW pack reader, BatchNorm
// folding + weight layout, and the launch plan (one Op per kernel launch) with a
// liveness-based activation arena.
#include <math.h>
#include <cmath>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <map>

#include "hep.h"
#include "hep_host.h"
#include "hep_knobs.h"

namespace hep {

// =================================================================================================
// architecture (mirror of hmd_ego_pose_amd/arch.py)
// =================================================================================================
static const double kScaling[8][2] = {{1.0, 1.0}, {1.0, 1.1}, {1.1, 1.2}, {1.2, 1.4}, {1.4, 1.8}, {1.6, 2.2}, {1.8, 2.6}, {2.0, 3.1}};
static const int kBackboneOfPhi[9] = {0, 1, 2, 3, 4, 5, 6, 6, 7};
static const int kFpnWidth[9] = {64, 88, 112, 160, 224, 288, 384, 384, 384};
static const int kFpnRepeats[9] = {3, 4, 5, 6, 7, 7, 8, 8, 8};
static const int kHeadDepth[9] = {3, 3, 3, 4, 4, 4, 5, 5, 5};
static const int kStages[7][6] = {{1, 3, 1, 1, 32, 16}, {2, 3, 2, 6, 16, 24}, {2, 5, 2, 6, 24, 40}, {3, 3, 2, 6, 40, 80},
                                  {3, 5, 1, 6, 80, 112}, {4, 5, 2, 6, 112, 192}, {1, 3, 1, 6, 192, 320}};
static const float kBnEps = 1e-3f;      // efficientnet/utils.py:245, efficientdet/model.py:36
static const float kFusionEps = 1e-4f;  // efficientdet/model.py:60

static int round_width(int c, double mult) {
  double c2 = c * mult;
  int r = std::max(8, (int)(c2 + 4) / 8 * 8);
  if (r < 0.9 * c2) r += 8;
  return r;
}

bool make_arch(int phi, Arch* a) {
  if (phi < 0 || phi > 7) return false;
  const double wm = kScaling[kBackboneOfPhi[phi]][0], dm = kScaling[kBackboneOfPhi[phi]][1];
  a->phi = phi; a->stem = round_width(32, wm); a->blocks.clear();
  std::vector<int> tapped;
  for (auto& st : kStages) {
    const int cin = round_width(st[4], wm), cout = round_width(st[5], wm), reps = (int)ceil(dm * st[0]);
    for (int j = 0; j < reps; j++) {
      MBConv b;
      b.cin = j == 0 ? cin : cout; b.cexp = b.cin * st[3]; b.k = st[1]; b.stride = j == 0 ? st[2] : 1;
      b.se = std::max(1, (int)(b.cin * 0.25)); b.cout = cout; b.expand = st[3] != 1; b.skip = j > 0;
      if (b.stride == 2) tapped.push_back((int)a->blocks.size() - 1);
      a->blocks.push_back(b);
    }
  }
  tapped.push_back((int)a->blocks.size() - 1);
  for (int i = 0; i < 3; i++) {
    a->taps[i] = tapped[tapped.size() - 3 + i];
    a->tap_channels[i] = a->blocks[a->taps[i]].cout;
  }
  a->fpn_w = kFpnWidth[phi]; a->fpn_cells = kFpnRepeats[phi]; a->head_depth = kHeadDepth[phi];
  a->attention = phi < 6;
  return true;
}

void same_pad(int n, int k, int s, int* before, int* after) {   // efficientnet/utils_extra.py:33-44
  int extra = ((n + s - 1) / s - 1) * s - n + k;
  if (extra < 0) extra = 0;
  *before = extra / 2; *after = extra - extra / 2;
}

// =================================================================================================
// anchors (reference generators/utils/anchors.py:273-419): float64 math, one cast to float32
// =================================================================================================
int host_anchors(int size, std::vector<float>* anchors, std::vector<float>* tanchors) {
  static const int sizes[5] = {32, 64, 128, 256, 512}, strides[5] = {8, 16, 32, 64, 128};
  // ratios/scales are stored as float32 in the reference and promoted to float64 in the math
  const double ratios[3] = {(double)1.0f, (double)0.5f, (double)2.0f};
  const double scales[3] = {(double)(float)pow(2.0, 0.0), (double)(float)pow(2.0, 1.0 / 3.0), (double)(float)pow(2.0, 2.0 / 3.0)};
  int total = 0;
  if (anchors) anchors->clear();
  if (tanchors) tanchors->clear();
  for (int l = 0; l < 5; l++) {
    const int fm = (size + (1 << (l + 3)) - 1) >> (l + 3);
    double base[9][4];
    for (int i = 0; i < 9; i++) {
      const double sc = scales[i / 3], ra = ratios[i % 3];
      const double wh = sizes[l] * sc;
      const double area = wh * wh;
      const double w = sqrt(area / ra), h = w * ra;
      base[i][0] = 0.0 - w * 0.5; base[i][1] = 0.0 - h * 0.5;
      base[i][2] = w - w * 0.5; base[i][3] = h - h * 0.5;
    }
    for (int y = 0; y < fm; y++)
      for (int x = 0; x < fm; x++) {
        const double cx = (x + 0.5) * strides[l], cy = (y + 0.5) * strides[l];
        for (int i = 0; i < 9; i++) {
          if (anchors) {
            anchors->push_back((float)(base[i][0] + cx)); anchors->push_back((float)(base[i][1] + cy));
            anchors->push_back((float)(base[i][2] + cx)); anchors->push_back((float)(base[i][3] + cy));
          }
          if (tanchors) { tanchors->push_back((float)cx); tanchors->push_back((float)cy); tanchors->push_back((float)strides[l]); }
        }
      }
    total += fm * fm * 9;
  }
  return total;
}

// =================================================================================================
// weight pack
// =================================================================================================
bool Pack::parse(const void* blob, size_t n, std::string* err) {
  storage.assign((const unsigned char*)blob, (const unsigned char*)blob + n);
  const unsigned char* p = storage.data();
  auto fail = [&](const char* m) { *err = std::string("weight pack: ") + m; return false; };
  if (n < 12 || memcmp(p, "HEPW", 4) != 0) return fail("bad magic (expected HEPW)");
  uint32_t ver, count; memcpy(&ver, p + 4, 4); memcpy(&count, p + 8, 4);
  if (ver != 1) return fail("unsupported version");
  size_t q = 12;
  for (uint32_t i = 0; i < count; i++) {
    if (q + 2 > n) return fail("truncated table");
    uint16_t nl; memcpy(&nl, p + q, 2); q += 2;
    if (q + nl + 1 > n) return fail("truncated table");
    std::string name((const char*)p + q, nl); q += nl;
    int nd = p[q]; q += 1;
    if (nd > 8 || q + 4 * nd + 16 > n) return fail("truncated table");
    PackTensor t; size_t cnt = 1; bool big = false;
    for (int d = 0; d < nd; d++) {
      uint32_t v; memcpy(&v, p + q, 4); q += 4; t.dims.push_back(v);
      if (v != 0 && cnt > (n / 4) / v) big = true; else cnt *= v;      // checked product: a tensor cannot hold more floats than the file
    }
    uint64_t off, nb; memcpy(&off, p + q, 8); memcpy(&nb, p + q + 8, 8); q += 16;
    if (big || nb != (uint64_t)cnt * 4 || off > n || nb > n - off || (off & 3) || off < 12) return fail("tensor out of bounds");
    t.data = (const float*)(p + off); t.count = cnt;
    tensors[name] = t;
  }
  return true;
}

const PackTensor* Pack::get(const std::string& name, std::initializer_list<int64_t> dims, std::string* err) const {
  auto it = tensors.find(name);
  if (it == tensors.end()) { *err = "weight pack: missing tensor '" + name + "'"; return nullptr; }
  if (it->second.dims != std::vector<int64_t>(dims)) {
    *err = "weight pack: tensor '" + name + "' has the wrong shape for this phi";
    return nullptr;
  }
  return &it->second;
}

// =================================================================================================
// weight builder: folds BN, lays weights out, converts to the session dtype
// =================================================================================================
static uint16_t f32_to_bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// fp32 -> OCP e4m3fn (bias 7, no infinities, max 448), round to nearest even, saturating
static uint8_t f32_to_e4m3(float f) {
  const uint8_t sign = std::signbit(f) ? 0x80 : 0;
  float a = fabsf(f);
  if (a != a) return sign | 0x7f;
  if (a >= 464.f) return sign | 0x7e;                                  // beyond the last rounding boundary: +-448
  if (a < 0.015625f) {                                                 // below 2^-6: subnormals, step 2^-9
    const int q = (int)nearbyintf(a * 512.f);
    return sign | (uint8_t)q;                                          // q == 8 is the smallest normal, 0x08
  }
  int e; const float m = frexpf(a, &e);                                // a = m * 2^e, m in [0.5, 1)
  int ee = e - 1, mant = (int)nearbyintf((m * 2.f - 1.f) * 8.f);
  if (mant == 8) { mant = 0; ee++; }
  if (ee > 8 || (ee == 8 && mant > 6)) return sign | 0x7e;
  return sign | (uint8_t)(((ee + 7) << 3) | mant);
}

struct WBuilder {
  std::vector<unsigned char> host;
  int dtype;
  size_t alloc(size_t bytes) {
    size_t off = (host.size() + 255) & ~(size_t)255;
    host.resize(off + bytes, 0);
    return off;
  }
  size_t put_f32(const std::vector<float>& v) {
    size_t off = alloc(v.size() * 4);
    memcpy(host.data() + off, v.data(), v.size() * 4);
    return off;
  }
  // e4m3 rows [rows][stride] of a [rows][K] matrix, one scale per row (amax / 448): w ~ e4m3 * scale
  size_t put_fp8(const std::vector<float>& v, int rows, int K, int stride, std::vector<float>* scales) {
    const size_t off = alloc((size_t)rows * stride);
    scales->assign(rows, 1.f);
    for (int n = 0; n < rows; n++) {
      float amax = 0.f;
      for (int k = 0; k < K; k++) amax = std::max(amax, fabsf(v[(size_t)n * K + k]));
      const float sc = amax > 0.f ? amax / 448.f : 1.f;
      (*scales)[n] = sc;
      for (int k = 0; k < K; k++) host[off + (size_t)n * stride + k] = f32_to_e4m3(v[(size_t)n * K + k] / sc);
    }
    return off;
  }
  size_t put_typed(const std::vector<float>& v) {   // dtype elements (fp8 sessions store bf16 everywhere except the quantised weights)
    if (dtype == 0) return put_f32(v);
    size_t off = alloc(v.size() * 2);
    uint16_t* d = (uint16_t*)(host.data() + off);
    for (size_t i = 0; i < v.size(); i++) d[i] = f32_to_bf16(v[i]);
    return off;
  }
};

struct BnFold { std::vector<float> scale, shift; };
static bool fold_bn(const Pack& pk, const std::string& p, int c, BnFold* out, std::string* err) {
  const PackTensor *g = pk.get(p + ".weight", {c}, err), *b = pk.get(p + ".bias", {c}, err),
                   *m = pk.get(p + ".running_mean", {c}, err), *v = pk.get(p + ".running_var", {c}, err);
  if (!g || !b || !m || !v) return false;
  out->scale.resize(c); out->shift.resize(c);
  for (int i = 0; i < c; i++) {
    const float s = g->data[i] / sqrtf(v->data[i] + kBnEps);
    out->scale[i] = s; out->shift[i] = b->data[i] - m->data[i] * s;
  }
  return true;
}

// =================================================================================================
// plan builder
// =================================================================================================
// Because Op structs are stored by value in a vector that grows, pointer patching is done by
// index after the plan is complete: each Op records symbolic references here.
struct Ref { int op; int field; int seg; int idx; size_t woff; int tensor; };
enum { F_STEM_W, F_STEM_B, F_STEM_OUT, F_PW_A, F_PW_W, F_PW_B, F_PW_RES, F_PW_OUT, F_DW_IN, F_DW_W, F_DW_B,
       F_DW_OUT, F_DW_PART, F_DW_WR, F_MBF_IN, F_MBF_WE, F_MBF_BE, F_MBF_WDW, F_MBF_BDW, F_MBF_OUT, F_MBF_PART, F_MBF_WR, F_MBF_WESCALE, F_PW_WSCALE, F_PW_SESCALE, F_SE_HPART, F_SE_SCALE, F_SE_BR, F_SE_WE, F_SE_BE, F_PW_HPART, F_PW_SEBR, F_PW_SEWE, F_PW_SEBE, F_POOL_IN, F_POOL_OUT, F_PWG_A, F_PWG_W, F_PWG_B, F_PWG_OUT,
       F_SEG_SRC, F_SEG_WDW, F_SEG_WPW, F_SEG_BIAS, F_SEG_OUT, F_CH_EXT_SRC, F_CH_EXT_STORE, F_CH_NODE_OUT, F_CH_WBLOB,
       F_SBF_WS, F_SBF_BS, F_SBF_WDW, F_SBF_BDW, F_SBF_STEM, F_SBF_OUT, F_SBF_PART, F_SBF_WR,
       F_XBF_IN, F_XBF_HPART, F_XBF_SEBR, F_XBF_SEWE, F_XBF_SEBE, F_XBF_BLOB, F_XBF_RES, F_XBF_MID, F_XBF_OUT, F_XBF_PART, F_XBF_WR,
       F_MBFT_SCALE, F_MBFT_BR, F_MBFT_WE, F_MBFT_BE, F_LATE_IN, F_LATE_BLOB, F_LATE_DS, F_LATE_RES, F_LATE_OUT, F_LATE_HPART, F_HEADS_FEAT, F_HEADS_BLOB, F_HEADS_OUT };

struct Planner {
  Session* s; const Pack& pk; std::string* err; WBuilder wb; bool ok = true;
  std::vector<Ref> refs;
  int ntails = 0;                     // fused fronts that finish their squeeze-excite in their tail (one counter block each)
  const Knobs kn;                     // plan knobs as the environment had them when the session was created (hep_knobs.h)
  Planner(Session* s_, const Pack& p, std::string* e) : s(s_), pk(p), err(e), kn(s_->knobs) { wb.dtype = s_->dtype; }

  int tensor(const std::string& name, int H, int W, int C, bool f32 = false) {
    TensorDesc t; t.name = name; t.H = H; t.W = W; t.C = C; t.f32 = f32;
    t.bytes_per_image = (size_t)H * W * C * (f32 ? 4 : s->esize());
    s->tensors.push_back(t);
    s->tensor_by_name[name] = (int)s->tensors.size() - 1;
    return (int)s->tensors.size() - 1;
  }
  const PackTensor* get(const std::string& n, std::initializer_list<int64_t> d) {
    const PackTensor* t = pk.get(n, d, err);
    if (!t) ok = false;
    return t;
  }
  void wref(int op, int field, size_t woff, int seg = -1, int idx = 0) { refs.push_back({op, field, seg, idx, woff, -1}); }
  void tref(int op, int field, int tensor, bool write, int seg = -1, int idx = 0) {
    refs.push_back({op, field, seg, idx, 0, tensor});
    Op& o = s->ops[op];
    (write ? o.writes : o.reads).push_back(tensor);
  }
  int new_op(OpKind k, const std::string& name) {
    Op o; memset(&o.stem, 0, sizeof o.stem); memset(&o.pw, 0, sizeof o.pw); memset(&o.dw, 0, sizeof o.dw);
    memset(&o.pool, 0, sizeof o.pool); memset(&o.sep, 0, sizeof o.sep); memset(&o.mbf, 0, sizeof o.mbf); memset(&o.pwg, 0, sizeof o.pwg); memset(&o.chain, 0, sizeof o.chain); memset(&o.se, 0, sizeof o.se); memset(&o.xbf, 0, sizeof o.xbf);
#ifdef HEP_ALT
    memset(&o.sbf, 0, sizeof o.sbf); memset(&o.late, 0, sizeof o.late); memset(&o.heads, 0, sizeof o.heads);
#endif
    o.kind = k; o.name = name;
    s->ops.push_back(o);
    return (int)s->ops.size() - 1;
  }
  double es() const { return (double)s->esize(); }

  // ---- pointwise conv (weights [N][K] as in the state_dict; optional conv bias; optional BN) ----
  // squeeze-excite of a project conv: the front kernel's partial reduce-FC rows + the rest of the two FCs
  struct SeSpec { int hpart_t = -1, rows = 0, sq = 0, sqp = 0; float inv_hw = 0.f; size_t br = 0, we = 0, be = 0; };
  int add_pw(const std::string& name, int in_t, int HW, int K, int N, const std::string& wkey, const std::string& bkey,
             const std::string& bnkey, int act, const SeSpec* se, int res_t, const std::string& out_name, int H, int W, bool quant = false, int out_pre = -1) {
    const PackTensor* w = get(wkey, {N, K, 1, 1});
    const PackTensor* cb = bkey.empty() ? nullptr : get(bkey, {N});
    BnFold bn;
    if (!bnkey.empty() && !fold_bn(pk, bnkey, N, &bn, err)) ok = false;
    if (!ok) return -1;
    const int tilesN = (N + 15) / 16;
    const bool quant_fp8 = quant && s->dtype == 2;
    std::vector<float> wf((size_t)tilesN * 16 * K, 0.f), bf((size_t)tilesN * 16, 0.f);
    for (int n = 0; n < N; n++) {
      const float sc = bnkey.empty() ? 1.f : bn.scale[n], sh = bnkey.empty() ? 0.f : bn.shift[n];
      for (int k = 0; k < K; k++) wf[(size_t)n * K + k] = w->data[(size_t)n * K + k] * sc;
      bf[n] = (cb ? cb->data[n] : 0.f) * sc + sh;
    }
    // tile shape: as many n-tiles per wave as fit (<= 8) so the activation rows are streamed
    // as few times as possible; two m-tiles per wave when the layer has rows to spare
    int pmode, pMT, pNT;
    const int64_t Mmax = (int64_t)HW * s->lane_batch;      // rows one launch sees (one lane of the batch)
    {
      const int64_t strips = (Mmax + 15) / 16;             // 16-row strips
      const int ksteps = (K + (s->dtype ? 32 : 16) - 1) / (s->dtype ? 32 : 16);
      auto clampi = [](int64_t v, int lo, int hi) { return (int)std::max<int64_t>(lo, std::min<int64_t>(hi, v)); };
      if (Mmax >= 16384) {          // big maps: waves along M, >= 256 workgroups anyway
        const int chunks = (tilesN + 7) / 8;
        pmode = 0; pNT = (tilesN + chunks - 1) / chunks;
        pMT = (Mmax >= 65536 && pNT <= 4) ? 2 : 1;
      } else if (ksteps >= 8) {     // small maps, deep K (project): split K over the 4 waves
        // widest split-K tile (alt build: HEP_PW_NT2).  Measured at phi 0 b16 (two m-tiles per wave): 4 / 2 / 1 n-tiles ->
        // one batch 0.6312 / 0.6259 / 0.6360 ms, four in flight 48.2k / 48.3k / 47.0k frames/s: 2
        const int nt2_max = kn.pw_nt2;
        pmode = 2; pMT = 1; pNT = clampi(strips * tilesN / 256, 1, std::min(std::min(8, nt2_max), tilesN));
      } else {                      // small maps, wide N (expand / lateral): waves side by side in N
        pmode = 1; pMT = 1; pNT = clampi(strips * tilesN / (4 * 256), 1, std::min(8, (tilesN + 3) / 4));
      }
      // Two m-tiles per wave in modes 1/2 (every weight fragment feeds two MFMAs, half the weight re-reads
      // from L2): no effect on a single batch in flight, but -2.5 % on the step with 4 batches in flight and
      // at batch 64, where these layers are bound by L2 traffic and not by latency.  (alt build: HEP_PW_MT2=0 disables.)
      {
        if (pmode != 0 && strips >= 8 && kn.pw_mt2 != 0) { pMT = 2; pNT = std::min(pNT, 4); }
        if (pmode == 2) pNT = std::min(pNT, std::max(1, kn.pw_nt2));
        // One more n-tile per wave where that brings the launch from two rounds of workgroups to one (the last project conv of phi 0:
        // 1152 -> 320 on the 8x8 maps, 32 x 10 = 320 workgroups with two n-tiles, 32 x 7 = 224 with three).  Round 4 selected it for fp32
        // sessions (one batch -5 us); re-measured with round 6's kernels it costs throughput: fp32 four in flight 28.84k -> 29.01k frames/s
        // WITHOUT it (four interleaved runs, one batch 17.03k -> 17.06k): the three-tile form runs four waves where the two-tile one
        // runs eight.  Off since; alt build: HEP_PW_NT3=1 selects it.
        if (pmode == 2 && kn.pw_nt3 > 0) {
          const int64_t mblocks = (strips + pMT - 1) / pMT;
          auto wgs = [&](int nt) { return mblocks * ((tilesN + nt - 1) / nt); };
          if (wgs(pNT) > 256 && wgs(pNT) <= 512 && pNT < 8 && wgs(pNT + 1) <= 256) pNT++;
        }
      }
    }
    // (alt build A/B, HEP_PW_WIDE=cap: fewest column chunks - fewer, fatter workgroups, fewer copies of the squeeze-excite prologue.  Measured,
    //  round 6: cap 8 -> 52.4k against 53.3k frames/s, with every squeeze-excite finish in the prologue 51.9k (against 50.4k with the default tiles):
    //  a 6-7 n-tile workgroup takes 11-15 us where the 2 n-tile one takes 6)
    if (pmode == 2 && kn.pw_wide > 0 && se) {
      const int cap = std::min(8, kn.pw_wide), ch = (tilesN + cap - 1) / cap;
      pNT = (tilesN + ch - 1) / ch;
    }
    // Squeeze-excite: finished in this GEMM's prologue (no launch) while the K x sq expand-FC matrix re-read by every
    // workgroup stays below HEP_SE_MAXMB (default 4) MB per launch; beyond that a small launch finishes it once per
    // image (k_dw.hip).  Measured at phi 0, batch 16 (same box, 2 x 300 steps): threshold 1000 / 12 / 8 / 4 / 0 MB ->
    // 40.7k / 41.0k / 41.3k / 41.7k / 41.8k frames/s with four batches in flight and 20.9k / 20.9k / 20.9k / 20.9k /
    // 20.6k with one (55 / 59 / 61 / 65 / 71 launches): the redundant L2 traffic of the prologue costs throughput on the
    // late blocks, the extra launches cost latency on the early ones.  phi 3 @ 512: 3.20k / 3.31k (24 MB) / 3.39k (0).
    int scale_t = -1;
    if (se) {
      const int rows_wg = pmode == 0 ? 64 * pMT : 16 * pMT, per_n = pmode == 1 ? 4 * pNT : pNT;
      const double wgs = (double)((Mmax + rows_wg - 1) / rows_wg) * ((tilesN + per_n - 1) / per_n);
      const double maxmb = kn.se_maxmb;
#ifdef HEP_ALT
      // ... or no launch at all (HEP_SE_TAIL=1, NOT the default): the fused front that produced the partial rows finishes it in its tail
      // (k_mbf.hip: last workgroup of an image to arrive).  Measured, round 5, phi 0 b16: 59 -> 49 launches, bit-identical, four in flight
      // 51.4k against 51.3k frames/s, one batch 0.600 -> 0.605 ms, fp32 one batch 0.980 -> 1.021 ms, phi 3 one batch +2.3 %: the tail
      // (stores acknowledged, ticket, rows and weight rows fetched past L2 by ONE workgroup) costs 6-9 us against the ~5 us of a launch
      // and its boundary - the price list's "5-13 us per seam inside a launch" again.
      int tail_op = -1;
      if (kn.se_tail != 0)
        for (const Ref& r : refs) if (r.field == F_MBF_PART && r.tensor == se->hpart_t) tail_op = r.op;
      if (wgs * K * se->sqp * es() > maxmb * 1e6 && tail_op >= 0 && ntails < 64) {
        scale_t = tensor(name + ".se_scale", 1, 1, K, true);
        MbfArgs& m = s->ops[tail_op].mbf;
        m.se_tail = 1 + ntails++;          // (1 + index of its counter block; the pointers are patched per lane)
        m.tail.C = K; m.tail.sq = se->sq; m.tail.sqp = se->sqp; m.tail.rows = se->rows; m.tail.bf16 = s->dtype; m.tail.inv_hw = se->inv_hw;
        tref(tail_op, F_MBFT_SCALE, scale_t, true);
        wref(tail_op, F_MBFT_BR, se->br); wref(tail_op, F_MBFT_WE, se->we); wref(tail_op, F_MBFT_BE, se->be);
        s->ops[tail_op].name += "+se";      // (the launch list shows which fronts carry the finish)
        s->ops[tail_op].weight_bytes += (double)K * se->sq * es() + ((double)K + se->sq) * 4;
        s->ops[tail_op].flops_per_image += 2.0 * K * se->sq;
      } else
#endif
      if (wgs * K * se->sqp * es() > maxmb * 1e6) {
        scale_t = tensor(name + ".se_scale", 1, 1, K, true);
        const int sop = new_op(OP_SE, name.substr(0, name.find('.')) + ".se");
        Op& so = s->ops[sop];
        so.se.C = K; so.se.sq = se->sq; so.se.sqp = se->sqp; so.se.rows = se->rows; so.se.bf16 = s->dtype; so.se.inv_hw = se->inv_hw;
        tref(sop, F_SE_HPART, se->hpart_t, false); tref(sop, F_SE_SCALE, scale_t, true);
        wref(sop, F_SE_BR, se->br); wref(sop, F_SE_WE, se->we); wref(sop, F_SE_BE, se->be);
        so.act_bytes_per_image = ((double)se->rows * se->sqp + K) * 4; so.weight_bytes = (double)K * se->sq * es() + ((double)K + se->sq) * 4;
        so.flops_per_image = 2.0 * K * se->sq;
      }
    }
    const int out_t = out_pre >= 0 ? out_pre : tensor(out_name, H, W, N);
    const int op = new_op(OP_PW, name);
    Op& o = s->ops[op];
    o.pw.K = K; o.pw.N = N; o.pw.tilesN = tilesN; o.pw.HW = HW; o.pw.act = act; o.pw.bf16 = s->dtype;
    o.pw.mode = pmode; o.pw.MT = pMT; o.pw.NT = pNT;
    // fp32 split-K GEMMs deep enough for eight K slices of two load batches each (HEP_PW_W8=0: four waves as before)
    o.pw.nwv = (s->dtype == 0 && pmode == 2 && pNT <= 2 && K >= kn.pw_w8_mink && kn.pw_w8 != 0) ? 8 : 4;
    tref(op, F_PW_A, in_t, false); tref(op, F_PW_OUT, out_t, true);     // (reads[0] is the GEMM's activation operand: fp8 calibration)
    if (quant && s->dtype == 2) {     // fp8 session: e4m3 weights, one scale per output channel behind the BN fold
      std::vector<float> sc;
      wref(op, F_PW_W, wb.put_fp8(wf, tilesN * 16, K, K, &sc)); wref(op, F_PW_WSCALE, wb.put_f32(sc));
      o.pw.fp8 = 1; o.pw.a_scale = 1.f;
    } else {
      // Fragment-ordered operands (k_pw_impl.h FRAG; HEP_PW_FRAG=0 disables): the split-K project convs whose scale comes from
      // se_finish_kernel and whose activations come from a fused front (k_mbf.hip), K a multiple of the k-step, whole m-tiles per image
      int producer = -1;
      for (const Ref& r : refs) if (r.field == F_MBF_OUT && r.tensor == in_t) producer = r.op;
      const bool frag = s->dtype != 2 && !quant_fp8 && pmode == 2 && pMT == 2 && pNT <= PW_FRAG_MAX_NT && act != ACT_SWISH && se && HW % 16 == 0 && producer >= 0 &&
                        kn.pw_frag != 0;
      if (frag) {
        const int kstep = s->dtype ? 32 : 16, klane = s->dtype ? 8 : 4, kst = (K + kstep - 1) / kstep;
        std::vector<float> wfr((size_t)tilesN * kst * 64 * klane, 0.f);      // K padded to whole k-steps with zeros
        for (int nt = 0; nt < tilesN; nt++) for (int ks = 0; ks < kst; ks++) for (int lane = 0; lane < 64; lane++) for (int e = 0; e < klane; e++) {
          const int k = ks * kstep + klane * (lane >> 4) + e;
          if (k < K) wfr[(((size_t)nt * kst + ks) * 64 + lane) * klane + e] = wf[(size_t)(nt * 16 + (lane & 15)) * K + k];
        }
        wref(op, F_PW_W, wb.put_typed(wfr));
        s->ops[op].pw.frag = 1; s->ops[producer].mbf.out_frag = 1; s->tensors[in_t].frag = true;
      } else wref(op, F_PW_W, wb.put_typed(wf));
    }
    wref(op, F_PW_B, wb.put_f32(bf));
    if (res_t >= 0) tref(op, F_PW_RES, res_t, false);
    o.act_bytes_per_image = ((double)HW * K + (double)HW * N * (res_t >= 0 ? 2 : 1)) * es();
    o.weight_bytes = (double)N * K * (o.pw.fp8 ? 1.0 : es());
    o.flops_per_image = 2.0 * HW * K * N;
    if (se && scale_t >= 0) {
      o.pw.se_from_tensor = 1; o.pw.sq = se->sq; o.pw.sqp = se->sqp;
      tref(op, F_PW_SESCALE, scale_t, false);
      const int rows_wg = o.pw.mode == 0 ? 64 * o.pw.MT : 16 * o.pw.MT;
      o.pw.se_nimg = HW % rows_wg == 0 ? 1 : (rows_wg + HW - 1) / HW + 1;
      o.act_bytes_per_image += (double)K * 4;
    } else if (se) {
      tref(op, F_PW_HPART, se->hpart_t, false);
      wref(op, F_PW_SEBR, se->br); wref(op, F_PW_SEWE, se->we); wref(op, F_PW_SEBE, se->be);
      o.pw.se_rows = se->rows; o.pw.sq = se->sq; o.pw.sqp = se->sqp; o.pw.inv_hw = se->inv_hw;
      // images a workgroup's rows can belong to (its scale table in LDS holds that many): rows per workgroup
      // and HW are multiples of 16 and workgroups start at multiples of their row count
      const int rows_wg = o.pw.mode == 0 ? 64 * o.pw.MT : 16 * o.pw.MT;
      o.pw.se_nimg = HW % rows_wg == 0 ? 1 : (rows_wg + HW - 1) / HW + 1;
      o.act_bytes_per_image += ((double)se->rows * se->sqp + se->sq) * 4;
      o.weight_bytes += (double)K * se->sq * es() + ((double)K + se->sq) * 4;
      o.flops_per_image += 2.0 * K * se->sq;
    }
    return out_t;
  }

  // ---- several independent pointwise convs (conv bias + BN folded, no activation) as ONE launch ----
  struct PwSpec { std::string name, wkey, bkey, bnkey, out_name; int in_t, HW, K, H, W; };
  std::vector<int> add_pw_group(const std::string& name, const std::vector<PwSpec>& specs, int N) {
    std::vector<int> outs;
    const int op = new_op(OP_PWG, name);
    int strips = 0;
    double bytes = 0, wbytes = 0, flops = 0;
    for (size_t i = 0; i < specs.size(); i++) {
      const PwSpec& sp = specs[i];
      const PackTensor* w = get(sp.wkey, {N, sp.K, 1, 1});
      const PackTensor* cb = sp.bkey.empty() ? nullptr : get(sp.bkey, {N});
      BnFold bn;
      if (!sp.bnkey.empty() && !fold_bn(pk, sp.bnkey, N, &bn, err)) ok = false;
      if (!ok) return outs;
      const int tilesN = (N + 15) / 16;
      std::vector<float> wf((size_t)tilesN * 16 * sp.K, 0.f), bf((size_t)tilesN * 16, 0.f);
      for (int n = 0; n < N; n++) {
        const float sc = sp.bnkey.empty() ? 1.f : bn.scale[n], sh = sp.bnkey.empty() ? 0.f : bn.shift[n];
        for (int k = 0; k < sp.K; k++) wf[(size_t)n * sp.K + k] = w->data[(size_t)n * sp.K + k] * sc;
        bf[n] = (cb ? cb->data[n] : 0.f) * sc + sh;
      }
      const int out_t = tensor(sp.out_name, sp.H, sp.W, N);
      outs.push_back(out_t);
      PwgSeg& g = s->ops[op].pwg.seg[i];
      g.HW = sp.HW; g.K = sp.K; g.N = N; g.tilesN = tilesN; g.blk_begin = strips;
      strips += (sp.HW + 15) / 16;
      wref(op, F_PWG_W, wb.put_typed(wf), (int)i); wref(op, F_PWG_B, wb.put_f32(bf), (int)i);
      tref(op, F_PWG_A, sp.in_t, false, (int)i); tref(op, F_PWG_OUT, out_t, true, (int)i);
      bytes += ((double)sp.HW * sp.K + (double)sp.HW * N) * es(); wbytes += (double)N * sp.K * es(); flops += 2.0 * sp.HW * sp.K * N;
    }
    Op& o = s->ops[op];
    o.pwg.nseg = (int)specs.size(); o.pwg.bf16 = s->dtype; o.pwg.strips_per_image = strips;
    o.act_bytes_per_image = bytes; o.weight_bytes = wbytes; o.flops_per_image = flops;
    return outs;
  }

  // ---- MBConv block ----
  // The project conv of a block is DEFERRED until the next block is planned: on the big maps it is fused with that
  // block's expand + depthwise convs into one boundary launch (k_xbf.hip); otherwise flush_project() emits it as a GEMM.
  struct Deferred { bool active = false; int i = 0; MBConv b; int dw_t = -1, inp = -1, out_t = -1, Ho = 0, Wo = 0; SeSpec se; std::string p; } defer;
  void flush_project() {
    if (!defer.active) return;
    defer.active = false;
    char nm[64]; snprintf(nm, sizeof nm, "b%d.project", defer.i);
    add_pw(nm, defer.dw_t, defer.Ho * defer.Wo, defer.b.cexp, defer.b.cout, defer.p + "._project_conv.conv.weight", "", defer.p + "._bn2", ACT_NONE, &defer.se,
           defer.b.skip ? defer.inp : -1, std::string("block") + std::to_string(defer.i), defer.Ho, defer.Wo, true, defer.out_t);
  }
  // boundary launch: [SE + project of the deferred block] + [expand + depthwise + SE partial sums of block i] (k_xbf.hip)
  bool try_xbf(int i, const MBConv& b, int dw_t, const std::vector<float>& wdw, const std::vector<float>& bdw, size_t wr_off, int sqp2,
               int Hin, int Win, int Ho, int Wo, int pt, int pl, bool mid_needed, int* part_t, int* nblk) {
    const int minh = kn.xbf_minh;     // smallest input map that takes the boundary kernel (64; alt build: HEP_XBF_MINH, HEP_XBF=0)
    if (kn.xbf == 0) return false;
    const MBConv& pb = defer.b;
    if (s->dtype == 2 || !b.expand || Hin < minh || !xbf_supports(b.k, b.stride) || pb.cout > 16 * XBF_MAXNT1 || b.cexp % 16 != 0 || pb.cexp % 8 != 0 ||
        (defer.se.sqp != 8 && defer.se.sqp != 16) || pb.cexp > 512) return false;
    XbfArgs xa; memset(&xa, 0, sizeof xa);
    xa.H = Hin; xa.W = Win; xa.K1 = pb.cexp; xa.N1 = pb.cout; xa.NT1 = (pb.cout + 15) / 16; xa.Cexp = b.cexp; xa.NT2 = b.cexp / 16;
    xa.Ho = Ho; xa.Wo = Wo; xa.k = b.k; xa.s = b.stride; xa.pad_t = pt; xa.pad_l = pl; xa.bf16 = s->dtype;
    xa.se_rows = defer.se.rows; xa.sq = defer.se.sq; xa.sqp = defer.se.sqp; xa.inv_hw = defer.se.inv_hw; xa.sq2 = b.se; xa.sqp2 = sqp2;
    xa.generic = kn.xbf_generic != 0;
    if (xbf_layout(&xa) == 0) return false;
    xa.tiles_x = (Wo + xa.tow - 1) / xa.tow; xa.tiles = xa.tiles_x * ((Ho + xa.toh - 1) / xa.toh);
    // weights: project of the deferred block (BN2 folded), expand of this block (BN0 folded), depthwise (BN1 folded, by the caller)
    char pb_[96]; snprintf(pb_, sizeof pb_, "backbone_net.model._blocks.%d", i);
    const std::string p = pb_;
    const PackTensor* w1 = get(defer.p + "._project_conv.conv.weight", {pb.cout, pb.cexp, 1, 1});
    const PackTensor* w2 = get(p + "._expand_conv.conv.weight", {b.cexp, b.cin, 1, 1});
    BnFold bn2, bn0;
    if (!fold_bn(pk, defer.p + "._bn2", pb.cout, &bn2, err) || !fold_bn(pk, p + "._bn0", b.cexp, &bn0, err)) ok = false;
    if (!ok) return true;
    const size_t es_ = s->esize(), pad = s->dtype ? 8 : 4;
    std::vector<unsigned char> blob((size_t)xa.blob_bytes, 0);
    auto put = [&](size_t byte_off, size_t idx, float v) {
      if (s->dtype) { const uint16_t h = f32_to_bf16(v); memcpy(blob.data() + byte_off + idx * 2, &h, 2); }
      else memcpy(blob.data() + byte_off + idx * 4, &v, 4);
    };
    const size_t W1P = xa.K1 + pad, K2P = (size_t)xa.NT1 * 16, W2P = K2P + pad;
    for (int n = 0; n < pb.cout; n++)
      for (int k = 0; k < pb.cexp; k++) put(0, (size_t)n * W1P + k, w1->data[(size_t)n * pb.cexp + k] * bn2.scale[n]);
    const size_t o2 = (size_t)(xa.off_w2 - xa.off_w1);
    for (int n = 0; n < b.cexp; n++)
      for (int k = 0; k < b.cin; k++) put(o2, (size_t)n * W2P + k, w2->data[(size_t)n * b.cin + k] * bn0.scale[n]);
    float* f = reinterpret_cast<float*>(blob.data() + (xa.off_f - xa.off_w1));
    memcpy(f, wdw.data(), wdw.size() * 4); f += wdw.size();                  // [k*k][Cexp]
    memcpy(f, bdw.data(), (size_t)b.cexp * 4); f += b.cexp;                   // depthwise bias
    memcpy(f, bn0.shift.data(), (size_t)b.cexp * 4); f += (size_t)xa.NT2 * 16;   // expand bias
    memcpy(f, bn2.shift.data(), (size_t)pb.cout * 4);                          // project bias (padded with zeros)
    (void)es_;
    const size_t boff = wb.alloc(blob.size());
    memcpy(wb.host.data() + boff, blob.data(), blob.size());
    char nm[64];
    snprintf(nm, sizeof nm, "b%d.se_hpart", i);
    // tiles per workgroup (HEP_XBF_TPW, A/B runs).  Measured at phi 0 b16: 2 tiles per workgroup on the 128x128 map (one
    // round of 512 workgroups instead of two) 31.0 us against 32.7 - the blob / squeeze-excite prologue it amortises is not
    // what a tile costs; 4 tiles per workgroup 35.2 us.  Default 1.
    {
      // default: two tiles per workgroup where one tile per workgroup would need more than the 512 workgroups the GPU holds at
      // once (two per CU) - the second round then runs in the same workgroups, without their blob / squeeze-excite prologue
      const int tpw_env = kn.xbf_tpw;
      const int tpw = tpw_env > 0 ? tpw_env : ((int64_t)xa.tiles * s->lane_batch > 512 ? 2 : 1);
      xa.tpw = std::max(1, std::min(tpw, xa.tiles));
    }
    *nblk = (xa.tiles + xa.tpw - 1) / xa.tpw;
    *part_t = tensor(nm, 1, *nblk, sqp2, true);
    snprintf(nm, sizeof nm, "b%d.project+b%d.front", defer.i, i);
    const int op = new_op(OP_XBF, nm);
    Op& o = s->ops[op];
    o.xbf = xa;
    wref(op, F_XBF_BLOB, boff); wref(op, F_XBF_SEBR, defer.se.br); wref(op, F_XBF_SEWE, defer.se.we); wref(op, F_XBF_SEBE, defer.se.be); wref(op, F_XBF_WR, wr_off);
    tref(op, F_XBF_IN, defer.dw_t, false); tref(op, F_XBF_HPART, defer.se.hpart_t, false);
    if (pb.skip) tref(op, F_XBF_RES, defer.inp, false);
    if (mid_needed) tref(op, F_XBF_MID, defer.out_t, true);
    tref(op, F_XBF_OUT, dw_t, true); tref(op, F_XBF_PART, *part_t, true);
    const double HWin = (double)Hin * Win;
    o.act_bytes_per_image = (HWin * pb.cexp + (double)Ho * Wo * b.cexp + HWin * pb.cout * ((pb.skip ? 1 : 0) + (mid_needed ? 1 : 0))) * es() +
                            ((double)defer.se.rows * defer.se.sqp + (double)*nblk * sqp2) * 4;
    o.weight_bytes = (double)blob.size() + (double)pb.cexp * defer.se.sq * es();
    o.flops_per_image = 2.0 * HWin * pb.cexp * pb.cout + 2.0 * HWin * b.cin * b.cexp + 2.0 * b.k * b.k * Ho * Wo * b.cexp;
    defer.active = false;
    return true;
  }
  std::vector<int> tap_blocks;     // blocks whose outputs feed the BiFPN (their outputs must reach HBM)
  // stem conv fused into block 0's depthwise launch (k_sbf.hip): set by build_session instead of emitting a stem launch
  struct FusedStem { bool on = false; size_t w_off = 0, b_off = 0; int S = 0, pad_t = 0, pad_l = 0, stem_t = -1; } fstem;
  int add_mbconv(int i, const MBConv& b, int x, int* H, int* W) {
    char pb[96]; snprintf(pb, sizeof pb, "backbone_net.model._blocks.%d", i);
    const std::string p = pb;
    char nm[64];
    const int inp = x;
    const int Hin = *H, Win = *W;
    // depthwise geometry
    int pt, pbm, pl, pr; same_pad(Hin, b.k, b.stride, &pt, &pbm); same_pad(Win, b.k, b.stride, &pl, &pr);
    const int Ho = (Hin + b.stride - 1) / b.stride, Wo = (Win + b.stride - 1) / b.stride;
    const PackTensor* wd = get(p + "._depthwise_conv.conv.weight", {b.cexp, 1, b.k, b.k});
    BnFold bn1; if (!fold_bn(pk, p + "._bn1", b.cexp, &bn1, err)) ok = false;
    if (!ok) return -1;
    std::vector<float> wdw((size_t)b.k * b.k * b.cexp);
    for (int c = 0; c < b.cexp; c++)
      for (int t = 0; t < b.k * b.k; t++) wdw[(size_t)t * b.cexp + c] = wd->data[(size_t)c * b.k * b.k + t] * bn1.scale[c];
    snprintf(nm, sizeof nm, "b%d.dw", i);
    const int dw_t = tensor(nm, Ho, Wo, (b.cexp + 31) & ~31);      // (room for the project GEMM's fragment order: K padded to whole k-steps)
    s->tensors[dw_t].C_logical = b.cexp;

    // fused front (expand -> LDS -> depthwise, k_mbf.hip) whenever its LDS tiles fit; the expanded
    // tensor then never reaches HBM.  HEP_NO_MBF=1 forces the two-kernel path (A/B measurements).
    // output tile side of the fused front: 16 on the stride-1 layers of 16x16 / 32x32 maps (the whole 16x16 map per
    // workgroup: no halo re-expansion, a quarter of the workgroups and of their fixed staging / drain latency), else 8.
    // HEP_MBF_TS=8 forces the small tile (A/B measurements, parity test of the alternative plan).
    const int ts16_maxh = kn.mbf_ts16_maxh;     // 32 (alt build: A/B knob)
    int ts = (b.stride == 1 && b.expand && Ho >= 16 && Ho <= ts16_maxh) ? 16 : 8;
    if (kn.mbf_ts8) ts = 8;
    int max_in = mbf_max_inside(Hin, Win, b.k, b.stride, pt, pl, ts);   // rows of the compact input tile in LDS
    if (ts == 16 && mbf_lds_layout(b.cin, 64, b.k, b.stride, s->dtype, b.expand, max_in, ts, nullptr) > 159 * 1024 &&
        mbf_lds_layout(b.cin, 32, b.k, b.stride, s->dtype, b.expand, max_in, ts, nullptr) > 159 * 1024) {
      ts = 8; max_in = mbf_max_inside(Hin, Win, b.k, b.stride, pt, pl, ts);
    }
    int CC = 0, npass = 1, kp = 0;
    // Measured on MI355X at bs16 (profiles/README.md): the fused kernel beats expand+depthwise on input
    // maps up to 32x32 (it expands only the tile pixels inside the image, so on the 8x8 maps the halo costs
    // nothing) and loses on the big early maps (bandwidth-bound, the two-kernel path already streams well;
    // stride-2 halos there cost up to 4.5x recompute).  HEP_MBF=all|none overrides for A/B runs.
    const int maxh = kn.mbf_maxh;     // largest input map that takes the fused front (32)
    const bool want = kn.mbf >= 0 ? kn.mbf == 1 : Hin <= maxh;
    if (want)
      for (int cand : {64, 32, 16})      // (128 would need an item loop in phase C: one (pixel quad, channel quad) item per lane covers 64 channels of an 8x8 / 16x16 tile)
        if ((kn.mbf_cc < 0 || cand <= kn.mbf_cc) && mbf_lds_layout(b.cin, std::min(cand, b.expand ? cand : b.cexp), b.k, b.stride, s->dtype, b.expand, max_in, ts, nullptr) <= 159 * 1024) { CC = cand; break; }
    if (kn.mbf_cc < 0 && CC == 64 && b.expand) {      // (alt build A/B: the narrower chunk where it fills one round of workgroups that the wide one leaves half empty)
      const long tiles = (long)((Ho + ts - 1) / ts) * ((Wo + ts - 1) / ts), wg64 = tiles * ((b.cexp + 63) / 64) * s->lane_batch, wg32 = tiles * ((b.cexp + 31) / 32) * s->lane_batch;
      if (wg64 <= -kn.mbf_cc && wg32 <= 256) CC = 32;
    }
    // The tile / channel chunk / K-slice plan with the fewest ROUNDS of workgroups (HEP_MBF_MP=0 disables, =2 lifts the bf16
    // restriction below, =force takes the multi-pass form wherever it exists).  With the whole K staged at once an fp32 front needs twice the LDS of the bf16 one: the 16x16 tile of blocks 9,
    // 10 fell back to 8x8 (704 workgroups on 256 CUs: three rounds, 37 us against 17 us in bf16), the 8x8 maps kept one workgroup
    // of 146 KB per CU where 288 want to be resident (two rounds).  The multi-pass expand (k_mbf.hip, MP) stages K in slices and
    // gives fp32 the workgroup counts of the bf16 plan.
    {
      const bool force = kn.mbf_mp == 3;                             // parity runs: the multi-pass form wherever it exists, whatever the rounds
      const bool mp_on = kn.mbf_mp != 0;
      // A pass costs ~1 us of phase changes and the wider chunk a longer depthwise phase, so one round saved out of three does not
      // pay (phi 3 blocks 9-12, bf16: 3 -> 2 rounds, 23.6 -> 26.4 us): the multi-pass plan must at least HALVE the rounds.  bf16
      // sessions additionally keep plans of up to two rounds (HEP_MBF_MP=2 lifts that): measured only where the gain is large -
      // phi 3 @ 512 b8 blocks 14-17 (7 -> 2 rounds) 47 -> 38 us, blocks 19-23 26.5 -> 18.9 us, 4.65k -> 4.87k frames/s.
      const long min_old_rounds = (s->dtype == 0 || kn.mbf_mp == 2) ? 2 : 3;
      if (CC && b.expand && mp_on && s->dtype != 2) {
        struct Cand { int ts, CC, npass, kp; long rounds; size_t lds; };
        const int kstep = s->dtype ? 32 : 16, ksteps = (b.cin + kstep - 1) / kstep;
        auto rounds_of = [&](int ts_, int cc_, size_t lds) {
          const long tiles = (long)((Ho + ts_ - 1) / ts_) * ((Wo + ts_ - 1) / ts_), wgs = tiles * ((b.cexp + cc_ - 1) / cc_) * s->lane_batch;
          const long per_cu = std::max<long>(1, std::min<long>((long)(160 * 1024 / lds), ts_ == 16 ? 2 : 4));
          return (wgs + 256 * per_cu - 1) / (256 * per_cu);
        };
        Cand best{ts, CC, 1, 0, rounds_of(ts, CC, mbf_lds_layout(b.cin, CC, b.k, b.stride, s->dtype, 1, max_in, ts, nullptr)), 0};
        const bool ts16_ok = b.stride == 1 && Ho >= 16 && Ho <= ts16_maxh && !kn.mbf_ts8;
        for (int ts_ : {16, 8}) {
          if (ts_ == 16 && !ts16_ok) continue;
          const int mi = mbf_max_inside(Hin, Win, b.k, b.stride, pt, pl, ts_);
          for (int cc_ : {64, 48, 32})                        // (48: three n-tiles - fewer passes where the 64-wide expanded tile leaves no room for a wider K slice)
            for (int kpp = 3; kpp >= 1; kpp--) {              // k-steps per pass
              const int kp_ = kpp * kstep, np_ = (b.cin + kp_ - 1) / kp_;
              if (np_ < 2 || kpp > ksteps || !mbf_mp_fits(cc_, kp_, s->dtype, mi, ts_)) continue;
              const size_t lds = mbf_lds_layout(b.cin, cc_, b.k, b.stride, s->dtype, 1, mi, ts_, nullptr, kp_);
              if (lds > 159 * 1024) continue;
              const Cand c{ts_, cc_, np_, kp_, rounds_of(ts_, cc_, lds), lds};
              // fewer rounds first; then fewer passes (two barriers each); then the wider chunk
              if ((best.npass == 1 && 2 * c.rounds <= best.rounds && best.rounds >= min_old_rounds) || (best.npass > 1 && c.rounds < best.rounds) || (force && best.npass == 1) ||
                  (c.rounds == best.rounds && best.npass > 1 && (c.npass < best.npass || (c.npass == best.npass && c.CC > best.CC)))) best = c;
            }
        }
        if (best.npass > 1) { ts = best.ts; CC = best.CC; npass = best.npass; kp = best.kp; max_in = mbf_max_inside(Hin, Win, b.k, b.stride, pt, pl, ts); }
        if (kn.plan_debug) fprintf(stderr, "libhep plan: block %d front: tile %d, %d channels per workgroup, %d pass(es) of %d input channels, %ld round(s)\n", i, ts, CC, npass, kp ? kp : b.cin, best.rounds);
      }
    }
    // squeeze-excite weights: reduce FC [sq][Cexp] (fp32) for the front kernel; bias, expand FC [Cexp][sqp] (session
    // dtype, rows padded with zeros) and its bias for the project GEMM's prologue
    const PackTensor *wr = get(p + "._se_reduce.conv.weight", {b.se, b.cexp, 1, 1}), *br = get(p + "._se_reduce.conv.bias", {b.se}),
                     *we = get(p + "._se_expand.conv.weight", {b.cexp, b.se, 1, 1}), *be = get(p + "._se_expand.conv.bias", {b.cexp});
    if (!ok) return -1;
    const int sqp = (b.se + 7) & ~7;          // one 16-byte vector of the expand-FC weights = 8 (bf16) / 4 (fp32) hidden units
    if (sqp > 256) { *err = "squeeze-excite width above 256 is not supported"; ok = false; return -1; }
    const size_t wr_off = wb.put_f32(std::vector<float>(wr->data, wr->data + wr->count));
    SeSpec se; se.sq = b.se; se.sqp = sqp; se.inv_hw = 1.0f / (float)(Ho * Wo);
    se.br = wb.put_f32(std::vector<float>(br->data, br->data + br->count));
    {
      std::vector<float> wep((size_t)b.cexp * sqp, 0.f);
      for (int c = 0; c < b.cexp; c++) for (int j = 0; j < b.se; j++) wep[(size_t)c * sqp + j] = we->data[(size_t)c * b.se + j];
      se.we = wb.put_typed(wep);              // session dtype: half the bytes of the project prologue in bf16 sessions
    }
    se.be = wb.put_f32(std::vector<float>(be->data, be->data + be->count));
    int part_t, nblk;
    bool boundary = false;
    if (defer.active) {
      const bool keep = s->flags & 1u;
      const bool mid_needed = keep || b.skip || std::find(tap_blocks.begin(), tap_blocks.end(), defer.i) != tap_blocks.end();
      boundary = try_xbf(i, b, dw_t, wdw, bn1.shift, wr_off, sqp, Hin, Win, Ho, Wo, pt, pl, mid_needed, &part_t, &nblk);
      if (!ok) return -1;
      if (!boundary) flush_project();
      if (!ok) return -1;
    }
    if (boundary) {
    } else if (CC) {
      if (!b.expand) CC = std::min(CC, b.cexp);
      nblk = ((Ho + ts - 1) / ts) * ((Wo + ts - 1) / ts) * ((b.cexp + CC - 1) / CC);      // one row per workgroup
      snprintf(nm, sizeof nm, "b%d.se_hpart", i);
      part_t = tensor(nm, 1, nblk, sqp, true);
      snprintf(nm, sizeof nm, "b%d.front", i);
      const int op = new_op(OP_MBF, nm);
      Op& o = s->ops[op];
      MbfArgs& m = o.mbf; memset(&m, 0, sizeof m);
      m.H = Hin; m.W = Win; m.Cin = b.cin; m.Cexp = b.cexp; m.Ho = Ho; m.Wo = Wo; m.k = b.k; m.s = b.stride;
      m.pad_t = pt; m.pad_l = pl; m.has_expand = b.expand; m.bf16 = s->dtype; m.CC = CC; m.sq = b.se; m.sqp = sqp; m.ts = ts;
      m.npass = npass; m.kp = kp;
      // (the register-resident form - the whole tile requested at kernel start, parked slice by slice - measured SLOWER than fetching
      //  slice by slice: 8x8 maps 24.1 us against 22.4 us per launch, blocks 9 / 10 31.4 against 29.6: a pass is not bound by its
      //  memory round trip but by its three phase changes, ~1 us per pass whatever feeds it; HEP_MBF_MP_RES=1 selects it)
      m.mp_resident = npass > 1 && mbf_mp_resident(b.cin, CC, max_in, ts) && kn.mbf_mp_res != 0;
      mbf_lds_layout(b.cin, CC, b.k, b.stride, s->dtype, b.expand, max_in, ts, &m, kp);
      wref(op, F_MBF_WR, wr_off);
      if (b.expand) {
        const PackTensor* w = get(p + "._expand_conv.conv.weight", {b.cexp, b.cin, 1, 1});
        BnFold bn0; if (!fold_bn(pk, p + "._bn0", b.cexp, &bn0, err)) ok = false;
        if (!ok) return -1;
        const int rows = (b.cexp + 15) / 16 * 16;
        std::vector<float> wf((size_t)rows * b.cin, 0.f);
        for (int n = 0; n < b.cexp; n++) for (int k = 0; k < b.cin; k++) wf[(size_t)n * b.cin + k] = w->data[(size_t)n * b.cin + k] * bn0.scale[n];
        if (s->dtype == 2) {
          std::vector<float> sc;
          wref(op, F_MBF_WE, wb.put_fp8(wf, rows, b.cin, (b.cin + 15) & ~15, &sc)); wref(op, F_MBF_WESCALE, wb.put_f32(sc));
          m.fp8 = 1; m.a_scale = 1.f;
        } else wref(op, F_MBF_WE, wb.put_typed(wf));
        wref(op, F_MBF_BE, wb.put_f32(bn0.shift));
      }
      wref(op, F_MBF_WDW, wb.put_f32(wdw)); wref(op, F_MBF_BDW, wb.put_f32(bn1.shift));
      tref(op, F_MBF_IN, x, false); tref(op, F_MBF_OUT, dw_t, true); tref(op, F_MBF_PART, part_t, true);
      o.act_bytes_per_image = ((double)Hin * Win * b.cin + (double)Ho * Wo * b.cexp) * es();
      o.weight_bytes = (b.expand ? (double)b.cexp * b.cin * (m.fp8 ? 1.0 : es()) : 0.0) + (double)b.k * b.k * b.cexp * 4;
      o.flops_per_image = (b.expand ? 2.0 * Hin * Win * b.cin * b.cexp : 0.0) + 2.0 * b.k * b.k * Ho * Wo * b.cexp;
    } else {
      if (b.expand) {
        snprintf(nm, sizeof nm, "b%d.expand", i);
        x = add_pw(nm, x, Hin * Win, b.cin, b.cexp, p + "._expand_conv.conv.weight", "", p + "._bn0", ACT_SWISH, nullptr, -1,
                   std::string(nm), Hin, Win, true);
        if (!ok) return -1;
      }
      // depthwise on its own: from global memory for the big maps (bandwidth-bound, k_dw.hip), through
      // LDS (the fused kernel without its expand stage) on the 8x8 maps, where the global-memory
      // version is a chain of k*k dependent load latencies.  HEP_DWLDS=0|1 overrides.
      const int dl = kn.dwlds;
      // (measured at bs16: the 5x5 stride-2 layer on the 64x64 map takes 20.6 us through LDS against
      //  27.2 us from global memory - 25 taps per output re-read too much through L1)
      const bool lds_dw = dl >= 0 ? dl != 0 : (Hin <= 8 || (b.k == 5 && Hin <= 64));
      const int ccl = std::min(64, b.cexp);
      const int max_in8 = mbf_max_inside(Hin, Win, b.k, b.stride, pt, pl, 8);
      if (lds_dw && mbf_lds_layout(b.cexp, ccl, b.k, b.stride, s->dtype, 0, max_in8, 8, nullptr) <= 159 * 1024) {
        nblk = ((Ho + 7) / 8) * ((Wo + 7) / 8) * ((b.cexp + ccl - 1) / ccl);
        snprintf(nm, sizeof nm, "b%d.se_hpart", i);
        part_t = tensor(nm, 1, nblk, sqp, true);
        snprintf(nm, sizeof nm, "b%d.dw", i);
        const int op = new_op(OP_MBF, nm);
        Op& o = s->ops[op];
        MbfArgs& m = o.mbf; memset(&m, 0, sizeof m);
        m.H = Hin; m.W = Win; m.Cin = b.cexp; m.Cexp = b.cexp; m.Ho = Ho; m.Wo = Wo; m.k = b.k; m.s = b.stride;
        m.pad_t = pt; m.pad_l = pl; m.has_expand = 0; m.bf16 = s->dtype; m.CC = ccl; m.sq = b.se; m.sqp = sqp; m.ts = 8;
        mbf_lds_layout(b.cexp, ccl, b.k, b.stride, s->dtype, 0, max_in8, 8, &m);
        wref(op, F_MBF_WR, wr_off);
        wref(op, F_MBF_WDW, wb.put_f32(wdw)); wref(op, F_MBF_BDW, wb.put_f32(bn1.shift));
        tref(op, F_MBF_IN, x, false); tref(op, F_MBF_OUT, dw_t, true); tref(op, F_MBF_PART, part_t, true);
        o.act_bytes_per_image = ((double)Hin * Win + (double)Ho * Wo) * b.cexp * es();
        o.weight_bytes = (double)b.k * b.k * b.cexp * 4;
        o.flops_per_image = 2.0 * b.k * b.k * Ho * Wo * b.cexp;
      } else {
#ifdef HEP_ALT
        if (i == 0 && fstem.on) {
          // stem conv + this depthwise conv as ONE launch (k_sbf.hip): the stem's output stays in LDS
          const int op = new_op(OP_SBF, "stem+b0.dw");
          Op& o = s->ops[op];
          SbfArgs& f = o.sbf;
          f.H = fstem.S; f.W = fstem.S; f.Hs = Hin; f.Ws = Win; f.C = b.cexp; f.pad_t = fstem.pad_t; f.pad_l = fstem.pad_l; f.bf16 = s->dtype;
          f.sq = b.se; f.sqp = sqp;
          sbf_layout(&f);
          nblk = f.tiles;
          snprintf(nm, sizeof nm, "b%d.se_hpart", i);
          part_t = tensor(nm, 1, nblk, sqp, true);
          wref(op, F_SBF_WS, fstem.w_off); wref(op, F_SBF_BS, fstem.b_off);
          wref(op, F_SBF_WDW, wb.put_f32(wdw)); wref(op, F_SBF_BDW, wb.put_f32(bn1.shift)); wref(op, F_SBF_WR, wr_off);
          if (s->flags & 1u) tref(op, F_SBF_STEM, fstem.stem_t, true);
          tref(op, F_SBF_OUT, dw_t, true); tref(op, F_SBF_PART, part_t, true);
          o.act_bytes_per_image = 3.0 * fstem.S * fstem.S * 4 + (double)Ho * Wo * b.cexp * es() + (double)nblk * sqp * 4;
          o.weight_bytes = (27.0 + 10.0) * b.cexp * 4;
          o.flops_per_image = 2.0 * 27 * Hin * Win * b.cexp + 2.0 * 9 * Ho * Wo * b.cexp;
        } else
#endif
        {
        // strip width: 4 output pixels per lane on the big maps; the stride-2 layers read 2x the columns per
        // output, so 2 keeps their loads denser (measured 18.1 us against 19.8 us on 128x128 -> 64x64 x 96)
        const int TW = Wo >= 32 ? (b.stride == 2 ? 2 : 4) : (Wo >= 16 ? 2 : 1);
        nblk = dw_blocks_per_image(Ho, Wo, b.cexp, TW);
        snprintf(nm, sizeof nm, "b%d.se_hpart", i);
        part_t = tensor(nm, 1, nblk, sqp, true);
        snprintf(nm, sizeof nm, "b%d.dw", i);
        const int op = new_op(OP_DW, nm);
        Op& o = s->ops[op];
        o.dw.H = Hin; o.dw.W = Win; o.dw.C = b.cexp; o.dw.Ho = Ho; o.dw.Wo = Wo; o.dw.k = b.k; o.dw.s = b.stride;
        o.dw.pad_t = pt; o.dw.pad_l = pl; o.dw.act = ACT_SWISH; o.dw.bf16 = s->dtype; o.dw.TW = TW; o.dw.blocks_per_image = nblk;
        o.dw.sq = b.se; o.dw.sqp = sqp;
        wref(op, F_DW_W, wb.put_f32(wdw)); wref(op, F_DW_B, wb.put_f32(bn1.shift)); wref(op, F_DW_WR, wr_off);
        tref(op, F_DW_IN, x, false); tref(op, F_DW_OUT, dw_t, true); tref(op, F_DW_PART, part_t, true);
        o.act_bytes_per_image = ((double)Hin * Win + (double)Ho * Wo) * b.cexp * es();
        o.weight_bytes = (double)b.k * b.k * b.cexp * 4;
        o.flops_per_image = 2.0 * b.k * b.k * Ho * Wo * b.cexp;
        }
      }
      }
    se.hpart_t = part_t; se.rows = nblk;
    // project + bn2 (+ residual), SE scale applied on the GEMM's input side: deferred (see flush_project / try_xbf)
    defer.active = true; defer.i = i; defer.b = b; defer.dw_t = dw_t; defer.inp = inp; defer.Ho = Ho; defer.Wo = Wo; defer.se = se; defer.p = p;
    defer.out_t = tensor(std::string("block") + std::to_string(i), Ho, Wo, b.cout);
    *H = Ho; *W = Wo;
    return defer.out_t;
  }

#ifdef HEP_ALT
  // ---- image-resident run of late blocks (k_late.hip): blocks i0 .. i1 as ONE launch, one workgroup per image ----
  // how many blocks starting at i0 (input map H x W, bf16 sessions) the late kernel can take: 0 = none
  int late_run(const std::vector<MBConv>& blocks, int i0, int H, int W) const {
    // HEP_LATE: 0 = launch by launch (front / squeeze-excite / project per block), 1 = the image-resident kernel
    if (!kn.late || s->dtype != 1) return 0;
    int n = 0;
    for (size_t i = i0; i < blocks.size() && n < LATE_MAX_BLOCKS; i++, n++) {
      const MBConv& b = blocks[i];
      if (!b.expand || !late_block_supported(b.cin, b.cexp, b.cout, b.k, b.stride, H, W, b.se) || (n > 0 && b.cin != blocks[i - 1].cout)) break;
    }
    if (n < 2) return 0;
    LateArgs probe; memset(&probe, 0, sizeof probe);
    probe.nblk = n;
    for (int j = 0; j < n; j++) { const MBConv& b = blocks[i0 + j]; probe.blk[j].Cin = b.cin; probe.blk[j].Cexp = b.cexp; probe.blk[j].N = b.cout; probe.blk[j].k = b.k; }
    return late_layout(&probe) ? n : 0;
  }
  // returns the output tensors of the blocks
  std::vector<int> add_late(const std::vector<MBConv>& blocks, int i0, int n, int x) {
    std::vector<int> outs;
    flush_project();
    if (!ok) return outs;
    char nm[64]; snprintf(nm, sizeof nm, "b%d-b%d.blocks", i0, i0 + n - 1);
    const int op = new_op(OP_LATE, nm);
    LateArgs la; memset(&la, 0, sizeof la);
    la.nblk = n;
    std::vector<unsigned char> blob;
    auto reserve = [&](size_t bytes) { const size_t off = (blob.size() + 15) & ~(size_t)15; blob.resize(off + bytes, 0); return off; };
    auto put_bf16 = [&](size_t off, size_t idx, float v) { const uint16_t h = f32_to_bf16(v); memcpy(blob.data() + off + idx * 2, &h, 2); };
    auto put_f32v = [&](size_t off, size_t idx, float v) { memcpy(blob.data() + off + idx * 4, &v, 4); };
    int cexp_max = 0;
    double act = 0, flops = 0;
    for (int j = 0; j < n; j++) {
      const MBConv& b = blocks[i0 + j];
      LateBlock& L = la.blk[j];
      L.Cin = b.cin; L.Cexp = b.cexp; L.N = b.cout; L.k = b.k; L.skip = b.skip ? 1 : 0; L.nchunks = b.cexp / LATE_CC;
      L.sq = b.se; L.sqp = (b.se + 7) & ~7; L.inv_hw = 1.0f / 64.0f;
      cexp_max = std::max(cexp_max, b.cexp);
    }
    if (!late_layout(&la)) { *err = "late blocks do not fit LDS"; ok = false; return outs; }      // (fills ntw / ng)
    for (int j = 0; j < n; j++) {
      const MBConv& b = blocks[i0 + j];
      LateBlock& L = la.blk[j];
      char pb[96]; snprintf(pb, sizeof pb, "backbone_net.model._blocks.%d", i0 + j);
      const std::string p = pb;
      const PackTensor *we = get(p + "._expand_conv.conv.weight", {b.cexp, b.cin, 1, 1}), *wd = get(p + "._depthwise_conv.conv.weight", {b.cexp, 1, b.k, b.k}),
                       *wr = get(p + "._se_reduce.conv.weight", {b.se, b.cexp, 1, 1}), *br = get(p + "._se_reduce.conv.bias", {b.se}),
                       *w2 = get(p + "._se_expand.conv.weight", {b.cexp, b.se, 1, 1}), *b2 = get(p + "._se_expand.conv.bias", {b.cexp}),
                       *wp = get(p + "._project_conv.conv.weight", {b.cout, b.cexp, 1, 1});
      BnFold bn0, bn1, bn2;
      if (!fold_bn(pk, p + "._bn0", b.cexp, &bn0, err) || !fold_bn(pk, p + "._bn1", b.cexp, &bn1, err) || !fold_bn(pk, p + "._bn2", b.cout, &bn2, err)) ok = false;
      if (!ok) return outs;
      const int kse = b.cin / 32, ksp = b.cexp / 32, kk = b.k * b.k, NC = L.nchunks, sqp = L.sqp;
      // expand weights in MFMA fragment order: [chunk][n-tile][k-step][lane = (r, g)][8]: W[n = c*128 + t*16 + r][k = ks*32 + 8g + e]
      L.off_we = (uint32_t)reserve((size_t)b.cexp * b.cin * 2);
      for (int c = 0; c < NC; c++) for (int t = 0; t < 8; t++) for (int ks = 0; ks < kse; ks++) for (int lane = 0; lane < 64; lane++) for (int e = 0; e < 8; e++) {
        const int nn = c * LATE_CC + t * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4) + e;
        put_bf16(L.off_we, ((size_t)((c * 8 + t) * kse + ks) * 64 + lane) * 8 + e, we->data[(size_t)nn * b.cin + k] * bn0.scale[nn]);
      }
      L.off_be = (uint32_t)reserve((size_t)b.cexp * 4);
      for (int c = 0; c < b.cexp; c++) put_f32v(L.off_be, c, bn0.shift[c]);
      L.off_wdw = (uint32_t)reserve((size_t)kk * b.cexp * 4);
      for (int c = 0; c < NC; c++) for (int t = 0; t < kk; t++) for (int i = 0; i < LATE_CC; i++) {
        const int ch = c * LATE_CC + i;
        put_f32v(L.off_wdw, ((size_t)c * kk + t) * LATE_CC + i, wd->data[(size_t)ch * kk + t] * bn1.scale[ch]);
      }
      L.off_bdw = (uint32_t)reserve((size_t)b.cexp * 4);
      for (int c = 0; c < b.cexp; c++) put_f32v(L.off_bdw, c, bn1.shift[c]);
      L.off_w1 = (uint32_t)reserve((size_t)NC * sqp * LATE_CC * 4);
      for (int c = 0; c < NC; c++) for (int jj = 0; jj < b.se; jj++) for (int i = 0; i < LATE_CC; i++)
        put_f32v(L.off_w1, ((size_t)c * sqp + jj) * LATE_CC + i, wr->data[(size_t)jj * b.cexp + c * LATE_CC + i]);
      L.off_b1 = (uint32_t)reserve(64 * 4);
      for (int jj = 0; jj < b.se; jj++) put_f32v(L.off_b1, jj, br->data[jj]);
      L.off_w2 = (uint32_t)reserve((size_t)b.cexp * sqp * 2);
      for (int c = 0; c < b.cexp; c++) for (int jj = 0; jj < b.se; jj++) put_bf16(L.off_w2, (size_t)c * sqp + jj, w2->data[(size_t)c * b.se + jj]);
      L.off_b2 = (uint32_t)reserve((size_t)b.cexp * 4);
      for (int c = 0; c < b.cexp; c++) put_f32v(L.off_b2, c, b2->data[c]);
      // project weights: [n-group][k-step][n-tile of the group][lane][8]: W[n = (ng*ntw + jt)*16 + r][k = ks*32 + 8g + e], rows >= N zero
      L.off_wp = (uint32_t)reserve((size_t)L.ng * ksp * L.ntw * 1024);
      for (int ng = 0; ng < L.ng; ng++) for (int ks = 0; ks < ksp; ks++) for (int jt = 0; jt < L.ntw; jt++) for (int lane = 0; lane < 64; lane++) {
        const int nn = (ng * L.ntw + jt) * 16 + (lane & 15);
        if (nn >= b.cout) continue;
        for (int e = 0; e < 8; e++) {
          const int k = ks * 32 + 8 * (lane >> 4) + e;
          put_bf16(L.off_wp, ((size_t)((ng * ksp + ks) * L.ntw + jt) * 64 + lane) * 8 + e, wp->data[(size_t)nn * b.cexp + k] * bn2.scale[nn]);
        }
      }
      L.off_bp = (uint32_t)reserve((size_t)L.ng * L.ntw * 16 * 4);
      for (int c = 0; c < b.cout; c++) put_f32v(L.off_bp, c, bn2.shift[c]);
      const int out_t = tensor(std::string("block") + std::to_string(i0 + j), 8, 8, b.cout);
      outs.push_back(out_t);
      act += 64.0 * b.cout * es() * (b.skip ? 2 : 1);
      flops += 2.0 * 64 * b.cin * b.cexp + 2.0 * kk * 64 * b.cexp + 2.0 * 64 * b.cexp * b.cout + 4.0 * b.cexp * b.se;
    }
    const size_t boff = wb.alloc(blob.size());
    memcpy(wb.host.data() + boff, blob.data(), blob.size());
    // workgroups per image (HEP_LATE_G; default 3 where every block's chunk count divides): each runs 1 / G of the chunk loop -
    // the part that is bound by ONE CU's vector ALU - and they meet once per block (k_late.hip)
    int G = kn.late_g;
    if (G < 1 || G > 8) G = 1;
    for (int j = 0; j < n; j++) if (la.blk[j].nchunks % G != 0) G = 1;
    // the members of a group spin-wait for each other: they must all be resident at once.  A workgroup of this kernel fills a CU (1024
    // threads, near-full LDS), so a launch of lane_batch * G workgroups is only safe while that is a small part of the chip - other
    // streams' workgroups hold CUs as well.  Beyond a quarter of the CUs: one workgroup per image, no meetings.
    {
      int cus = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device) != hipSuccess || cus <= 0) cus = 256;
      if ((long)s->lane_batch * G > cus / 4) G = 1;
    }
    la.G = G;
    la.cross_xcd = kn.late_xcd != 0;      // 1: a group on consecutive workgroup ids = across XCDs (tests: same bits)
    const int ds_t = tensor(std::string(nm) + ".dw", 1, 1, 64 * cexp_max * (G > 1 ? 2 : 1));
    la.dstride = 64 * cexp_max * 2 * (G > 1 ? 2 : 1);
    const int hp_t = tensor(std::string(nm) + ".se_part", 1, 1, 2 * 8 * 64, true);
    s->ops[op].late = la;
    wref(op, F_LATE_BLOB, boff);
    tref(op, F_LATE_IN, x, false); tref(op, F_LATE_DS, ds_t, true); tref(op, F_LATE_HPART, hp_t, true);
    for (int j = 0; j < n; j++) {
      tref(op, F_LATE_OUT, outs[j], true, j);
      if (blocks[i0 + j].skip) tref(op, F_LATE_RES, j == 0 ? x : outs[j - 1], false, j);
    }
    Op& o = s->ops[op];
    o.act_bytes_per_image = 64.0 * blocks[i0].cin * es() + act;      // (the depthwise outputs' round trip through L2 is the kernel's own business)
    o.weight_bytes = (double)blob.size();
    o.flops_per_image = flops;
    return outs;
  }

#endif   // HEP_ALT

  // ---- fused separable conv launch (1..n segments sharing C) ----
  struct SegSpec {
    int src[3]; int kind[3]; float fw[3]; int nsrc; int pre_act;
    int level;                       // pyramid level index 0..4 of the output
    std::string key;                 // state_dict prefix of the SeparableConvBlock
    std::string bn;                  // BN to fold (may be empty)
    int N; int act;
    int out_t;                       // dtype tensor, or -1 for a head output
    int head_out; int col_kin, col_kout, col_off, out_k;   // head output index 0..4 and column mapping
  };
  // bf16 at width 64 (instantiated for the A/B, bit-identical): map layers 16 against 13.3 us, the header launch - the hand head as one
  // 36-tile segment on eight waves - 47.6 against 27.3 us; 49.0k against 49.7k frames/s with both: tower_kernel's four wave-private
  // halos fit at this width and it has no barrier per image
  static int tower_coop_default_bf16_64(int direct) { (void)direct; return 0; }
  int sep_tile_side() const {
    SepArgs probe; memset(&probe, 0, sizeof probe);
    sep_lds_layout(s->arch.fpn_w, s->dtype, 8, 96, s->arch.fpn_w, &probe);
    return probe.lds_bytes > 160 * 1024 ? 4 : 8;
  }
  void add_sep(const std::string& name, const std::vector<SegSpec>& specs, bool chain = false) {
    const int C = s->arch.fpn_w;
    const int op = new_op(OP_SEP, name);
    int tile_begin = 0, ts_max = 4, cols_f32 = 0, cols_map = 0;
    // tile side 8; 4 where the 8x8 tile of this width and dtype does not fit in LDS (fp32, width >= 288);
    // chains (one tile per image) then only cover levels <= 4x4
    const int ts_pick = sep_tile_side();
    double bytes = 0, flops = 0, wbytes = 0;
    // head layers (many independent single-source segments, all maps or all head outputs) run on the
    // wave-per-patch kernel of k_tower.hip; HEP_TOWER=0 keeps them on the tiled kernel of k_sep.hip
    int direct = 0;
    {
      bool simple = specs.size() > 1 && !chain && tower_supports(C) && kn.tower != 0;
      bool maps = true, heads = true;
      for (const SegSpec& sp : specs) {
        simple = simple && sp.nsrc == 1 && sp.kind[0] == SRC_SAME && !sp.pre_act && sp.fw[0] == 1.f;
        maps = maps && sp.out_t >= 0 && sp.N == C; heads = heads && sp.out_t < 0;
      }
      if (simple && (maps || heads)) direct = maps ? 1 : 2;
    }
    // cooperative tower form (tower_coop_kernel) wherever it is instantiated and measured faster: bf16 map layers from width 160, where
    // the wave-private halos of tower_kernel no longer fit (phi 3 @ 512 b8: 5.02k -> 5.15k frames/s; the bf16 header launch there stays
    // on tower_kernel: 113 against 118 us - its segments have 1-10 n-tiles for five waves), fp32 at width 64 (24.7k -> 25.7k).
    // HEP_TOWER_COOP: 0 off, 1 on wherever instantiated, 2 map layers only, 3 headers only
    int coop = 0;
    if (direct && tower_coop_supported(C, s->dtype != 0)) {
      const int v = kn.tower_coop;
      if (v < 0) coop = s->dtype == 0 || (C >= 160 ? direct == 1 : tower_coop_default_bf16_64(direct));
      else coop = v == 1 || (v == 2 && direct == 1) || (v == 3 && direct == 2);
    }
    const int chunk_cols_out = (direct ? (coop ? tower_coop_hdr_tiles(C, s->dtype != 0) : tower_hdr_tiles(C, s->dtype != 0)) : SEP_MAX_TILES_N) * 16;   // head outputs are split into column chunks, maps never
    int tiles_n_max = 0;
    for (size_t i = 0; i < specs.size(); i++) {
      const SegSpec& sp = specs[i];
      const int hw = s->levels[sp.level];
      const PackTensor* wd = get(sp.key + ".depthwise_conv.conv.weight", {C, 1, 3, 3});
      const PackTensor* wp = get(sp.key + ".pointwise_conv.conv.weight", {sp.N, C, 1, 1});
      const PackTensor* bp = get(sp.key + ".pointwise_conv.conv.bias", {sp.N});
      BnFold bn; if (!sp.bn.empty() && !fold_bn(pk, sp.bn, sp.N, &bn, err)) ok = false;
      if (!ok) return;
      std::vector<float> wdw((size_t)9 * C);
      for (int c = 0; c < C; c++) for (int t = 0; t < 9; t++) wdw[(size_t)t * C + c] = wd->data[(size_t)c * 9 + t];
      const size_t wdw_off = wb.put_f32(wdw);
      for (int j = 0; j < sp.nsrc; j++) bytes += (double)s->tensors[sp.src[j]].H * s->tensors[sp.src[j]].W * C * es();
      bytes += (double)hw * hw * sp.N * (sp.out_t >= 0 ? es() : 4.0);
      flops += 2.0 * 9 * hw * hw * C + 2.0 * hw * hw * C * sp.N;
      wbytes += (double)sp.N * C * es() + 9.0 * C * 4;
      const int chunk_cols = sp.out_t >= 0 ? SEP_MAX_TILES_MAP * 16 : chunk_cols_out;
      for (int n0 = 0; n0 < sp.N; n0 += chunk_cols) {        // wide headers: one segment per chunk of columns
        const int Nc = std::min(chunk_cols, sp.N - n0);
        SepSeg sg; memset(&sg, 0, sizeof sg);
        sg.h = hw; sg.w = hw; sg.C = C; sg.nsrc = sp.nsrc; sg.pre_act = sp.pre_act;
        for (int j = 0; j < sp.nsrc; j++) {
          const TensorDesc& t = s->tensors[sp.src[j]];
          sg.kind[j] = sp.kind[j]; sg.fw[j] = sp.fw[j]; sg.sh[j] = t.H; sg.sw[j] = t.W;
          int pb, pa; same_pad(t.H, 3, 2, &pb, &pa); sg.pool_pad[j] = pb;
        }
        const int tilesN = direct == 1 ? tower_map_tiles(C) : (Nc + 15) / 16;
        tiles_n_max = std::max(tiles_n_max, tilesN);
        std::vector<float> wf((size_t)tilesN * 16 * C, 0.f), bf((size_t)tilesN * 16, 0.f);
        for (int n = 0; n < Nc; n++) {
          const float sc = sp.bn.empty() ? 1.f : bn.scale[n0 + n], sh = sp.bn.empty() ? 0.f : bn.shift[n0 + n];
          // k_tower.hip map layers: MFMA row (nt, i) holds channel (i/4)*RUN + 4*nt + i%4 (RUN = 4*tilesN),
          // so that a lane's results over the n-tiles are consecutive channels
          int row = n;
          if (direct == 1) { const int run = 4 * tilesN, gq = n / run, rem = n % run; row = (rem / 4) * 16 + gq * 4 + rem % 4; }
          for (int k = 0; k < C; k++) wf[(size_t)row * C + k] = wp->data[(size_t)(n0 + n) * C + k] * sc;
          bf[n] = bp->data[n0 + n] * sc + sh;
        }
        sg.N = Nc; sg.tilesN = tilesN; sg.act = sp.act; sg.n_base = n0;
        sg.ts = ts_pick;   // 8; 16x16 tiles measured slower (3 dependent gather rounds per lane, 1 workgroup per CU)
        if (!direct && !chain && specs.size() == 1 && hw <= kn.sep_ts4_maxhw) sg.ts = 4;   // latency knob: 4x4 tiles on levels with few 8x8 tiles (one batch in flight +1-3 %, four in flight -1-4 %: DESIGN section 2)
        ts_max = std::max(ts_max, sg.ts);
        if (sp.out_t >= 0) cols_map = std::max(cols_map, Nc); else cols_f32 = std::max(cols_f32, Nc);
        sg.tiles_x = (hw + sg.ts - 1) / sg.ts; sg.tiles_y = sg.tiles_x; sg.tile_begin = tile_begin; sg.tiles_x_rcp = rcp_u32(sg.tiles_x);
        tile_begin += sg.tiles_x * sg.tiles_y;
        if (sp.out_t >= 0) {
          sg.out_f32 = 0; sg.out_bstride = (int64_t)hw * hw * sp.N; sg.out_off = n0; sg.out_rowstride = sp.N;
          sg.col_kin = 1; sg.col_kout = 1; sg.col_off = 0;
        } else {
          sg.out_f32 = 1; sg.out_bstride = (int64_t)s->num_anchors * sp.out_k; sg.out_off = (int64_t)s->level_off[sp.level] * sp.out_k;
          sg.out_rowstride = 9 * sp.out_k; sg.col_kin = sp.col_kin; sg.col_kout = sp.col_kout; sg.col_off = sp.col_off;
        }
        s->ops[op].segs.push_back(sg);
        const int si = (int)s->ops[op].segs.size() - 1;
        wref(op, F_SEG_WDW, wdw_off, si); wref(op, F_SEG_WPW, wb.put_typed(wf), si); wref(op, F_SEG_BIAS, wb.put_f32(bf), si);
        for (int j = 0; j < sp.nsrc; j++) tref(op, F_SEG_SRC, sp.src[j], false, si, j);
        if (sp.out_t >= 0) tref(op, F_SEG_OUT, sp.out_t, true, si);
        else refs.push_back({op, F_SEG_OUT, si, 0, 0, -(sp.head_out + 2)});   // encoded head output
      }
    }
    Op& o = s->ops[op];
    o.sep.nseg = (int)o.segs.size(); o.sep.total_tiles = tile_begin; o.sep.bf16 = s->dtype; o.sep.C = C;
    o.sep.chain = chain && o.segs.size() > 1;
    o.sep.direct = direct;
    o.sep.coop = coop;
    bool all_maps = true;
    for (const SegSpec& sp : specs) all_maps = all_maps && sp.out_t >= 0 && sp.N == C;
    sep_lds_layout(C, s->dtype, ts_max, cols_f32, cols_map, &o.sep, (chain || specs.size() == 1) && all_maps && !direct && kn.sep_wlds != 0);
    if (direct) { o.sep.off_wdw = 0; o.sep.off_bias = (size_t)9 * C * 4; o.sep.lds_bytes = o.sep.off_bias + (size_t)tiles_n_max * 16 * 4; }
    if (o.sep.lds_bytes > 160 * 1024) { *err = "BiFPN width too large for the fused separable-conv tile"; ok = false; return; }
    o.act_bytes_per_image = bytes; o.flops_per_image = flops; o.weight_bytes = wbytes;
  }

#ifdef HEP_ALT
  // ---- the five head towers depth-first (k_heads.hip): ONE launch for the tower layers and headers of every net and level ----
  struct HeaderSpec { int net; std::string key; int kin, kout, off, out, act; };
  bool add_heads_fused(const std::vector<std::string>& nets, const std::vector<HeaderSpec>& hds, const int feat[5], int depth) {
    const int C = s->arch.fpn_w;
    if (!heads_fused_supported(C, depth, s->dtype)) return false;
    const int op = new_op(OP_HEADS, "heads.fused");
    HeadsArgs ha; memset(&ha, 0, sizeof ha);
    ha.D = depth; ha.num_anchors = s->num_anchors;
    for (int l = 0; l < 5; l++) ha.level_off[l] = s->level_off[l];
    ha.lds_bytes = heads_lds_bytes(depth, &ha.off_wdw);
    // items: 16x16 output tiles, sorted by decreasing work - pixels of the tile x (tower layers + the header's n-tiles): the hand net's
    // tiles of the big levels first (the grid is (images, items): item 0 of every image is dispatched before item 1 of any)
    std::vector<HeadItem> items;
    std::vector<double> work;
    for (int l = 0; l < 5; l++)
      for (int n = 0; n < (int)nets.size(); n++) {
        const int hw = s->levels[l];
        int ncols = 0;
        for (const HeaderSpec& h : hds) if (h.net == n) ncols += 9 * h.kin;
        for (int ty = 0; ty < hw; ty += 16) for (int tx = 0; tx < hw; tx += 16) {
          HeadItem it; memset(&it, 0, sizeof it);
          it.net = n; it.level = l; it.hw = hw; it.y0 = ty; it.x0 = tx;
          items.push_back(it);
          const double px = (double)std::min(16, hw - ty) * std::min(16, hw - tx);
          work.push_back(px * (depth * 1.6 * 64 + ncols) + 2000);
        }
      }
    {
      std::vector<int> order(items.size());
      for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
      std::stable_sort(order.begin(), order.end(), [&](int x_, int y_) { return work[x_] > work[y_]; });
      std::vector<HeadItem> sorted;
      for (int i : order) sorted.push_back(items[i]);
      items.swap(sorted);
    }
    std::vector<unsigned char> blob(items.size() * sizeof(HeadItem), 0);
    auto reserve = [&](size_t bytes) { const size_t off = (blob.size() + 15) & ~(size_t)15; blob.resize(off + bytes, 0); return off; };
    auto put_bf16 = [&](size_t off, size_t idx, float v) { const uint16_t h = f32_to_bf16(v); memcpy(blob.data() + off + idx * 2, &h, 2); };
    auto put_f32v = [&](size_t off, size_t idx, float v) { memcpy(blob.data() + off + idx * 4, &v, 4); };
    // depthwise table [9][C] fp32 with k_tower.hip's swizzle: channels (0,2,1,3 | 4,6,5,7) of every octet
    auto put_dw = [&](size_t off, const PackTensor* wd) {
      static const int perm[8] = {0, 2, 1, 3, 4, 6, 5, 7};
      for (int t = 0; t < 9; t++) for (int c = 0; c < C; c++) put_f32v(off, (size_t)t * C + (c & ~7) + (c & 7), wd->data[(size_t)((c & ~7) + perm[c & 7]) * 9 + t]);
    };
    // pointwise fragments [n-tile][k-step][lane = (r, g)][8]: W[n = nt*16 + r][k = ks*32 + 8g + e] (rows >= N zero)
    auto put_frags = [&](size_t off, const PackTensor* wp, int N, int ntiles, const float* scale) {
      for (int nt = 0; nt < ntiles; nt++) for (int ks = 0; ks < C / 32; ks++) for (int lane = 0; lane < 64; lane++) {
        const int n = nt * 16 + (lane & 15);
        if (n >= N) continue;
        for (int e = 0; e < 8; e++) { const int k = ks * 32 + 8 * (lane >> 4) + e; put_bf16(off, ((size_t)(nt * (C / 32) + ks) * 64 + lane) * 8 + e, wp->data[(size_t)n * C + k] * (scale ? scale[n] : 1.f)); }
      }
    };
    double wbytes = 0, flops = 0, obytes = 0;
    std::vector<std::vector<size_t>> layer_off(nets.size(), std::vector<size_t>(5, 0));
    for (size_t n = 0; n < nets.size(); n++)
      for (int l = 0; l < 5; l++) {
        const size_t off = reserve((size_t)depth * (8192 + 256 + 2304));
        layer_off[n][l] = off;
        for (int i = 0; i < depth; i++) {
          const std::string key = nets[n] + ".conv_list." + std::to_string(i);
          const PackTensor* wd = get(key + ".depthwise_conv.conv.weight", {C, 1, 3, 3});
          const PackTensor* wp = get(key + ".pointwise_conv.conv.weight", {C, C, 1, 1});
          const PackTensor* bp = get(key + ".pointwise_conv.conv.bias", {C});
          BnFold bn; if (!fold_bn(pk, nets[n] + ".bn_list." + std::to_string(l) + "." + std::to_string(i), C, &bn, err)) ok = false;
          if (!ok) return true;
          const size_t lo = off + (size_t)i * (8192 + 256 + 2304);
          put_frags(lo, wp, C, 4, bn.scale.data());
          for (int c = 0; c < C; c++) put_f32v(lo + 8192, c, bp->data[c] * bn.scale[c] + bn.shift[c]);
          put_dw(lo + 8192 + 256, wd);
        }
        wbytes += depth * (8192.0 + 256 + 2304);
        flops += depth * (2.0 * 9 * s->levels[l] * s->levels[l] * C + 2.0 * s->levels[l] * s->levels[l] * C * C);
      }
    std::vector<size_t> hdr_off(hds.size(), 0);
    for (size_t h = 0; h < hds.size(); h++) {
      const HeaderSpec& hd = hds[h];
      const int N = 9 * hd.kin, ntiles = (N + 15) / 16;
      const PackTensor* wd = get(hd.key + ".depthwise_conv.conv.weight", {C, 1, 3, 3});
      const PackTensor* wp = get(hd.key + ".pointwise_conv.conv.weight", {N, C, 1, 1});
      const PackTensor* bp = get(hd.key + ".pointwise_conv.conv.bias", {N});
      if (!ok) return true;
      const size_t off = reserve(2304 + (size_t)ntiles * 64 + (size_t)ntiles * 2 * 1024);
      hdr_off[h] = off;
      put_dw(off, wd);
      for (int c = 0; c < N; c++) put_f32v(off + 2304, c, bp->data[c]);
      put_frags(off + 2304 + (size_t)ntiles * 64, wp, N, ntiles, nullptr);
      wbytes += 2304.0 + ntiles * (64.0 + 2048);
      for (int l = 0; l < 5; l++) { flops += 2.0 * 9 * s->levels[l] * s->levels[l] * C + 2.0 * s->levels[l] * s->levels[l] * C * N; obytes += (double)s->levels[l] * s->levels[l] * N * 4; }
    }
    for (HeadItem& it : items) {
      it.off_layers = (uint32_t)layer_off[it.net][it.level];
      for (size_t h = 0; h < hds.size(); h++)
        if (hds[h].net == it.net) {
          const int j = it.nhdr++;
          if (j >= 2) { *err = "more than two headers on one head net"; ok = false; return true; }
          it.off_hdr[j] = (uint32_t)hdr_off[h]; it.hdr_N[j] = 9 * hds[h].kin; it.hdr_ntiles[j] = (9 * hds[h].kin + 15) / 16;
          it.hdr_kin[j] = hds[h].kin; it.hdr_kout[j] = hds[h].kout; it.hdr_off[j] = hds[h].off; it.hdr_act[j] = hds[h].act; it.hdr_out[j] = hds[h].out;
        }
    }
    memcpy(blob.data(), items.data(), items.size() * sizeof(HeadItem));
    ha.nitems = (int)items.size();
    const size_t boff = wb.alloc(blob.size());
    memcpy(wb.host.data() + boff, blob.data(), blob.size());
    s->ops[op].heads = ha;
    wref(op, F_HEADS_BLOB, boff);
    double ibytes = 0;
    for (int l = 0; l < 5; l++) { tref(op, F_HEADS_FEAT, feat[l], false, l); ibytes += (double)s->levels[l] * s->levels[l] * C * es() * nets.size(); }
    for (int k = 0; k < 5; k++) refs.push_back({op, F_HEADS_OUT, k, 0, 0, -(k + 2)});   // encoded head output
    Op& o = s->ops[op];
    o.act_bytes_per_image = ibytes + obytes; o.weight_bytes = wbytes; o.flops_per_image = flops;
    return true;
  }

#endif   // HEP_ALT

  // ---- LDS-resident chain of small-level BiFPN nodes (k_chain.hip): [optional max-pools of cell 0] + nodes ----
  // pools: {source tensor, output tensor, output level} in order (cell 0: p6_in from p6_pre, p7_in from p6_in).
  // Returns false (nothing added) when the chain does not fit the kernel: the caller then emits the k_sep.hip path.
  struct PoolSpec { int src_t, out_t, level; };
  bool add_chain(const std::string& name, const std::vector<PoolSpec>& pools, const std::vector<SegSpec>& specs) {
    const int C = s->arch.fpn_w;
    if (C % 8 != 0 || specs.empty()) return false;
    if (s->dtype == 0 && kn.chain_f32 == 0) return false;      // (alt build A/B knob: fp32 chains back on k_sep.hip)
    // LDS map slots.  Every slot records who writes it (def: -1 = the prologue, else the node index) and the last node that
    // reads it; after the node list is complete the slots are packed by liveness (pack_slots below): the output of node n takes
    // the place of a map of the same size whose last reader is a node <= n.  fp32 sessions need it (maps of 16 KB), bf16
    // sessions just use less LDS.
    struct Slot { int off, h, w; };
    struct SlotLife { int off, elems, def, last; };
    std::map<std::pair<int, int>, Slot> slot;          // (tensor, 0 as stored / 1 pooled to half size) -> LDS slot
    std::vector<SlotLife> life;
    int top = 0;                                        // elements
    auto new_slot = [&](int t, int form, int h, int w, int def) {
      Slot sl{top, h, w}; const int elems = (h * w * C + 7) & ~7;
      life.push_back({top, elems, def, def}); top += elems; slot[{t, form}] = sl; return sl;
    };
    auto touch = [&](int off, int node) { for (SlotLife& l : life) if (l.off == off) l.last = std::max(l.last, node); };
    std::vector<ChainExt> exts; std::vector<int> ext_t, ext_store_t;
    std::vector<ChainNode> nodes; std::vector<int> node_out_t;
    auto pool_pad_of = [&](int n) { int pb, pa; same_pad(n, 3, 2, &pb, &pa); return pb; };
    auto add_ext = [&](int t, bool pooled, int store_t) {
      const TensorDesc& td = s->tensors[t];
      ChainExt x; memset(&x, 0, sizeof x);
      x.sh = td.H; x.sw = td.W; x.kind = pooled ? SRC_DOWN : SRC_SAME;
      x.h = pooled ? (td.H + 1) / 2 : td.H; x.w = pooled ? (td.W + 1) / 2 : td.W; x.pool_pad = pool_pad_of(td.H);
      x.off = new_slot(store_t >= 0 ? store_t : t, store_t >= 0 ? 0 : (pooled ? 1 : 0), x.h, x.w, -1).off;
      exts.push_back(x); ext_t.push_back(t); ext_store_t.push_back(store_t);
    };
    // pools of cell 0
    for (size_t i = 0; i < pools.size(); i++) {
      const PoolSpec& ps = pools[i];
      if (!slot.count({ps.src_t, 0})) {                 // first pool: its input comes from an earlier launch -> pooled while loaded
        add_ext(ps.src_t, true, ps.out_t);
      } else {                                          // pool of a map that lives in a slot
        const Slot in = slot[{ps.src_t, 0}];
        ChainNode nd; memset(&nd, 0, sizeof nd);
        nd.pool_only = 1; nd.nsrc = 1; nd.h = (in.h + 1) / 2; nd.w = (in.w + 1) / 2; nd.pool_pad = pool_pad_of(in.h);
        nd.src[0] = ChainSrc{in.off, SRC_DOWN, in.h, in.w, 1.f};
        touch(in.off, (int)nodes.size());
        nd.out_off = new_slot(ps.out_t, 0, nd.h, nd.w, (int)nodes.size()).off; nd.widx = -1;
        nodes.push_back(nd); node_out_t.push_back(ps.out_t);
      }
    }
    const size_t es2 = s->dtype ? 2 : 4, pad = s->dtype ? 8 : 4;
    const int wnode_bytes = (int)((((size_t)10 * C * 4 + (size_t)C * (C + pad) * es2) + 1023) & ~(size_t)1023);   // whole KB: streamed by LDS-DMA, 1 KB per wave instruction
    std::vector<unsigned char> blob;
    double bytes = 0, flops = 0, wbytes = 0;
    int nconv = 0, hw_max = 0;
    for (const SegSpec& sp : specs) {
      const int hw = s->levels[sp.level];
      if (sp.N != C || sp.out_t < 0 || sp.act != ACT_NONE || !sp.pre_act) return false;
      ChainNode nd; memset(&nd, 0, sizeof nd);
      nd.nsrc = sp.nsrc; nd.h = hw; nd.w = hw; nd.widx = nconv++;
      hw_max = std::max(hw_max, hw);
      const int me = (int)nodes.size();
      for (int j = 0; j < sp.nsrc; j++) {
        const int t = sp.src[j];
        const TensorDesc& td = s->tensors[t];
        bytes += (double)td.H * td.W * C * es();
        if (slot.count({t, 0})) {                       // a map of this chain (or an input already in a slot), at its own size
          const Slot sl = slot[{t, 0}];
          nd.src[j] = ChainSrc{sl.off, sp.kind[j], sl.h, sl.w, sp.fw[j]};
          if (sp.kind[j] == SRC_DOWN) nd.pool_pad = pool_pad_of(sl.h);
        } else if (sp.kind[j] == SRC_DOWN) {            // input from a bigger level: pooled while it is loaded
          if (!slot.count({t, 1})) add_ext(t, true, -1);
          const Slot sl = slot[{t, 1}];
          nd.src[j] = ChainSrc{sl.off, SRC_SAME, sl.h, sl.w, sp.fw[j]};
        } else {
          add_ext(t, false, -1);
          const Slot sl = slot[{t, 0}];
          nd.src[j] = ChainSrc{sl.off, sp.kind[j], sl.h, sl.w, sp.fw[j]};
        }
        touch(nd.src[j].off, me);
      }
      nd.out_off = new_slot(sp.out_t, 0, hw, hw, me).off;
      nodes.push_back(nd); node_out_t.push_back(sp.out_t);
      // weights in the kernel's LDS layout: [9][C] depthwise f32 | [C] bias f32 | [C][C + pad] pointwise rows in the session dtype
      const PackTensor* wd = get(sp.key + ".depthwise_conv.conv.weight", {C, 1, 3, 3});
      const PackTensor* wp = get(sp.key + ".pointwise_conv.conv.weight", {C, C, 1, 1});
      const PackTensor* bp = get(sp.key + ".pointwise_conv.conv.bias", {C});
      BnFold bn; if (!fold_bn(pk, sp.bn, C, &bn, err)) ok = false;
      if (!ok) return true;                             // (the error is reported by the caller through `ok`)
      const size_t base = blob.size();
      blob.resize(base + wnode_bytes, 0);
      float* fdw = reinterpret_cast<float*>(blob.data() + base);
      for (int c = 0; c < C; c++) for (int t9 = 0; t9 < 9; t9++) fdw[(size_t)t9 * C + c] = wd->data[(size_t)c * 9 + t9];
      float* fb = fdw + 9 * C;
      uint16_t* fw16 = reinterpret_cast<uint16_t*>(fb + C);
      float* fw32 = fb + C;
      for (int n = 0; n < C; n++) {
        fb[n] = bp->data[n] * bn.scale[n] + bn.shift[n];
        for (int k = 0; k < C; k++) {
          const float v = wp->data[(size_t)n * C + k] * bn.scale[n];
          if (s->dtype) fw16[(size_t)n * (C + pad) + k] = f32_to_bf16(v); else fw32[(size_t)n * (C + pad) + k] = v;
        }
      }
      bytes += (double)hw * hw * C * es(); flops += 2.0 * hw * hw * C * (9 + C); wbytes += (double)C * C * es() + 10.0 * C * 4;
    }
    if (nodes.size() > CH_MAX_NODES || exts.size() > CH_MAX_EXT) return false;
    // ---- pack the slots by liveness: exact-size reuse (the maps of a chain come in three sizes).  A convolution node reads its
    //      sources in its gather phase, two barriers before its MFMA phase writes the output, so its output may take the slot of
    //      one of its own sources; a pool-only node reads and writes in one phase and may not. ----
    {
      std::vector<int> order(life.size());
      for (size_t i = 0; i < life.size(); i++) order[i] = (int)i;
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return life[x].def < life[y].def; });
      struct Placed { int new_off, elems, last; };
      std::vector<Placed> placed; std::map<int, int> remap; int ntop = 0;
      for (int i : order) {
        const SlotLife& l = life[i];
        int pick = -1;
        // (a pool-only node writes its output in its only phase, with no barrier behind the previous node's copy-out: it never
        //  takes over a slot)
        if (l.def >= 0 && !nodes[l.def].pool_only)
          for (size_t q = 0; q < placed.size(); q++)
            if (placed[q].elems == l.elems && placed[q].last <= l.def) { pick = (int)q; break; }
        if (pick >= 0) { remap[l.off] = placed[pick].new_off; placed[pick].last = std::max(l.last, l.def); }
        else { remap[l.off] = ntop; placed.push_back({ntop, l.elems, std::max(l.last, l.def)}); ntop += l.elems; }
      }
      for (ChainExt& x : exts) x.off = remap[x.off];
      for (ChainNode& nd : nodes) { nd.out_off = remap[nd.out_off]; for (int j = 0; j < nd.nsrc; j++) nd.src[j].off = remap[nd.src[j].off]; }
      top = ntop;
    }
    ChainArgs ca; memset(&ca, 0, sizeof ca);
    ca.nnodes = (int)nodes.size(); ca.nconv = nconv; ca.next = (int)exts.size(); ca.C = C; ca.wnode_bytes = wnode_bytes; ca.bf16 = s->dtype ? 1 : 0;
    const size_t halo_b = (((size_t)(hw_max + 2) * (hw_max + 2) * (C + pad) * es2 + 15) & ~(size_t)15);
    const size_t atile_b = (size_t)((hw_max * hw_max + 15) & ~15) * (C + pad) * es2;
    ca.off_w = ((size_t)top * es2 + 15) & ~(size_t)15;
    // all node weights resident when they fit; else two nodes' weights in LDS, the next node's streamed under the running one (fp32
    // at width 64); else only the depthwise weights + biases in LDS and the pointwise fragments straight from global memory (width 160)
    const size_t wsmall = (size_t)10 * C * 4;
    auto wlds = [&](int mode) { return mode == 2 ? (size_t)nconv * wsmall : (size_t)(mode == 1 ? std::min(2, nconv) : nconv) * wnode_bytes; };
    ca.stream_w = 0;
    while (ca.stream_w < 2 && ca.off_w + wlds(ca.stream_w) + halo_b + atile_b > 158 * 1024) ca.stream_w++;
    if (kn.chain_stream >= 0) ca.stream_w = std::min(2, kn.chain_stream);     // A/B knob, parity tests of the other forms in bf16
    if (ca.stream_w == 2 && (C + (s->dtype ? 31 : 15)) / (s->dtype ? 32 : 16) > 6) return false;         // six k-steps of fragments in registers
    if (ca.stream_w == 2 && kn.chain_wglobal == 0) return false;
    ca.off_halo = ca.off_w + wlds(ca.stream_w);
    ca.off_atile = ca.off_halo + halo_b;
    ca.lds_bytes = ca.off_atile + atile_b;
    if (ca.lds_bytes > 158 * 1024) return false;
    const int op = new_op(OP_CHAIN, name);
    Op& o = s->ops[op];
    for (size_t e = 0; e < exts.size(); e++) {
      ca.ext[e] = exts[e];
      tref(op, F_CH_EXT_SRC, ext_t[e], false, (int)e);
      if (ext_store_t[e] >= 0) tref(op, F_CH_EXT_STORE, ext_store_t[e], true, (int)e);
    }
    for (ChainNode& nd : nodes) { nd.w_rcp = (uint32_t)(0x100000000ull / (uint32_t)nd.w) + 1; nd.hs_rcp = (uint32_t)(0x100000000ull / (uint32_t)(nd.w + 2)) + 1; }
    for (size_t n = 0; n < nodes.size(); n++) tref(op, F_CH_NODE_OUT, node_out_t[n], true, (int)n);
    o.chain = ca; o.cnodes = nodes;
    const size_t boff = wb.alloc(blob.size());
    memcpy(wb.host.data() + boff, blob.data(), blob.size());
    wref(op, F_CH_WBLOB, boff);
    o.act_bytes_per_image = bytes; o.flops_per_image = flops; o.weight_bytes = wbytes;
    return true;
  }
};

static void fusion_weights(const Pack& pk, const std::string& key, int n, bool attention, float* out, bool* ok, std::string* err) {
  if (!attention) { for (int i = 0; i < n; i++) out[i] = 1.f; return; }
  const PackTensor* t = pk.get(key, {n}, err);
  if (!t) { *ok = false; return; }
  float r[3], sum = 0.f;
  for (int i = 0; i < n; i++) { r[i] = std::max(t->data[i], 0.f); sum += r[i]; }
  for (int i = 0; i < n; i++) out[i] = r[i] / (sum + kFusionEps);   // efficientdet/model.py:212-213
}

int build_session(Session* s, const Pack& pack, std::string* err) {
  Planner P(s, pack, err);
  const Arch& A = s->arch;
  const int S = s->size;
  int off = 0;
  for (int l = 0; l < 5; l++) { s->levels[l] = (S + (1 << (l + 3)) - 1) >> (l + 3); s->level_off[l] = off; off += s->levels[l] * s->levels[l] * 9; }
  s->num_anchors = off;
  {   // classes: whatever the classifier's header was trained with (num_anchors * num_classes channels, efficientdet/model.py:393)
    auto it = pack.tensors.find("classifier.header.pointwise_conv.conv.weight");
    if (it == pack.tensors.end() || it->second.dims.size() != 4) { *err = "weight pack: missing tensor 'classifier.header.pointwise_conv.conv.weight'"; return HEP_ERR_PACK; }
    const int64_t n = it->second.dims[0];
    if (n < 9 || n % 9 || n / 9 > 63) { *err = "weight pack: the classifier header must hold 9 * num_classes channels, num_classes in 1..63"; return HEP_ERR_PACK; }
    s->num_classes = (int)(n / 9);
  }

  // ---- stem ----
  int H = (S + 1) / 2, W = (S + 1) / 2;
  int x;
  {
    const PackTensor* w = P.get("backbone_net.model._conv_stem.conv.weight", {A.stem, 3, 3, 3});
    BnFold bn; if (!fold_bn(pack, "backbone_net.model._bn0", A.stem, &bn, err)) P.ok = false;
    if (!P.ok) return HEP_ERR_PACK;
    std::vector<float> wf((size_t)27 * A.stem);
    for (int co = 0; co < A.stem; co++)
      for (int ci = 0; ci < 3; ci++)
        for (int t = 0; t < 9; t++) wf[((size_t)t * 3 + ci) * A.stem + co] = w->data[((size_t)co * 3 + ci) * 9 + t] * bn.scale[co];
    x = P.tensor("stem", H, W, A.stem);
    int pt, pb, pl, pr; same_pad(S, 3, 2, &pt, &pb); same_pad(S, 3, 2, &pl, &pr);
#ifdef HEP_ALT
    // stem + block 0's depthwise conv as one launch (k_sbf.hip), possible when block 0 has no expand conv (every
    // EfficientNet-B0..B7).  NOT the default: measured at phi 0 b16 bf16 the fused launch takes 42.8 us against 14.5 + 16.0 us
    // for the two kernels (workgroup life 14.8 us: 4 us input staging, 6.3 us for the stem phase - instruction-bound, not
    // matrix-pipe-bound: the split-bf16 MFMA form changed nothing - 2 us depthwise, 1.9 us channel sums; 14x14 tiles
    // recompute 1.56x the stem pixels).  HEP_SBF=1 selects it (parity-tested as an alternative plan).
    const MBConv& b0 = A.blocks[0];
    if (P.kn.sbf != 0 && s->dtype != 2 && !b0.expand && b0.k == 3 && b0.stride == 1 && A.stem % 8 == 0 && A.stem <= 64 && b0.cexp == A.stem &&
        b0.se <= 16) {
      P.fstem.on = true; P.fstem.w_off = P.wb.put_f32(wf); P.fstem.b_off = P.wb.put_f32(bn.shift); P.fstem.S = S; P.fstem.pad_t = pt; P.fstem.pad_l = pl; P.fstem.stem_t = x;
    } else
#endif
    {
    const int op = P.new_op(OP_STEM, "stem");
    Op& o = s->ops[op];
    o.stem.H = S; o.stem.W = S; o.stem.Ho = H; o.stem.Wo = W; o.stem.Cout = A.stem; o.stem.pad_t = pt; o.stem.pad_l = pl; o.stem.bf16 = s->dtype; o.stem.mfma = stem_uses_mfma(A.stem, P.kn.stem_mfma);
    P.wref(op, F_STEM_W, P.wb.put_f32(wf)); P.wref(op, F_STEM_B, P.wb.put_f32(bn.shift));
    P.tref(op, F_STEM_OUT, x, true);
    o.act_bytes_per_image = 3.0 * S * S * 4 + (double)H * W * A.stem * P.es();
    o.weight_bytes = 27.0 * A.stem * 4; o.flops_per_image = 2.0 * 27 * H * W * A.stem;
    }
  }
  // ---- backbone ----
  int taps[3] = {-1, -1, -1};
  for (int t = 0; t < 3; t++) P.tap_blocks.push_back(A.taps[t]);
  for (size_t i = 0; i < A.blocks.size(); i++) {
#ifdef HEP_ALT
    if (const int nl = P.late_run(A.blocks, (int)i, H, W)) {        // blocks i .. i + nl - 1 as one image-resident launch (k_late.hip)
      const std::vector<int> outs = P.add_late(A.blocks, (int)i, nl, x);
      if (!P.ok || (int)outs.size() != nl) return HEP_ERR_PACK;
      for (int j = 0; j < nl; j++)
        for (int t = 0; t < 3; t++) if (A.taps[t] == (int)i + j) taps[t] = outs[j];
      x = outs.back();
      i += nl - 1;
      continue;
    }
#endif
    x = P.add_mbconv((int)i, A.blocks[i], x, &H, &W);
    if (!P.ok) return HEP_ERR_PACK;
    for (int t = 0; t < 3; t++) if (A.taps[t] == (int)i) taps[t] = x;
  }
  P.flush_project();
  if (!P.ok) return HEP_ERR_PACK;
  // ---- BiFPN ----
  const int Wf = A.fpn_w;
  int feat[5];
  std::vector<Planner::SegSpec> pending; std::vector<std::string> pending_names;   // small-level BiFPN nodes awaiting a chain launch
  std::vector<Planner::PoolSpec> pending_pools;                                      // cell 0's two max-pools ride in the first chain
  // HEP_CHAIN: 0 = every node its own launch, 1 = chains inside k_sep.hip (mode 2), 2 (default) = LDS-resident chains
  // (k_chain.hip) wherever they fit - BiFPN width 64 in every session dtype (fp32: streamed node weights) - and k_sep.hip chains elsewhere
  const int chain_mode = P.kn.chain;
  for (int r = 0; r < A.fpn_cells; r++) {
    const std::string p = "bifpn." + std::to_string(r);
    const std::string tn = "c" + std::to_string(r) + ".";
    int in[5], in2[5];
    if (r == 0) {
      const int L3 = s->levels[0], L4 = s->levels[1], L5 = s->levels[2];
      // the six lateral 1x1 convs only depend on the backbone taps: one grouped launch (HEP_PWG=0: six launches)
      int p6pre;
      struct Lat { const char* nm; int tap, L; const char* key; const char* out; };
      const Lat lat[6] = {{"p3_down", 0, L3, ".p3_down_channel", "p3_in"}, {"p4_down", 1, L4, ".p4_down_channel", "p4_in"},
                          {"p5_down", 2, L5, ".p5_down_channel", "p5_in"}, {"p4_down2", 1, L4, ".p4_down_channel_2", "p4_in2"},
                          {"p5_down2", 2, L5, ".p5_down_channel_2", "p5_in2"}, {"p5_to_p6", 2, L5, ".p5_to_p6", "p6_pre"}};
      int lat_out[6];
      if (P.kn.pwg != 0) {
        std::vector<Planner::PwSpec> specs;
        for (const Lat& l : lat)
          specs.push_back({tn + l.nm, p + l.key + ".0.conv.weight", p + l.key + ".0.conv.bias", p + l.key + ".1", tn + l.out,
                           taps[l.tap], l.L * l.L, A.tap_channels[l.tap], l.L, l.L});
        const std::vector<int> outs = P.add_pw_group(tn + "laterals", specs, Wf);
        if (!P.ok || outs.size() != 6) return HEP_ERR_PACK;
        for (int i = 0; i < 6; i++) lat_out[i] = outs[i];
      } else {
        for (int i = 0; i < 6; i++)
          lat_out[i] = P.add_pw(tn + lat[i].nm, taps[lat[i].tap], lat[i].L * lat[i].L, A.tap_channels[lat[i].tap], Wf, p + lat[i].key + ".0.conv.weight",
                                p + lat[i].key + ".0.conv.bias", p + lat[i].key + ".1", ACT_NONE, nullptr, -1, tn + lat[i].out, lat[i].L, lat[i].L);
      }
      in[0] = lat_out[0]; in[1] = lat_out[1]; in[2] = lat_out[2]; in2[1] = lat_out[3]; in2[2] = lat_out[4]; p6pre = lat_out[5];
      if (!P.ok) return HEP_ERR_PACK;
      int prev = p6pre;
      for (int l = 3; l < 5; l++) {      // p6_in = pool(p6_pre), p7_in = pool(p6_in): launched with the first chain, or on their own
        const int oh = s->levels[l];
        const int t = P.tensor(tn + (l == 3 ? "p6_in" : "p7_in"), oh, oh, Wf);
        pending_pools.push_back({prev, t, l});
        in[l] = t; prev = t;
      }
      in2[0] = in[0]; in2[3] = in[3]; in2[4] = in[4];
    } else {
      for (int l = 0; l < 5; l++) { in[l] = feat[l]; in2[l] = feat[l]; }
    }
    float w[3];
    // Consecutive nodes on levels <= 8x8 (one tile per image) are launched as ONE chain: a workgroup
    // per image runs them back to back (k_sep.hip mode 2).  Measured at bs16: a 5-node chain takes
    // 29.8 us against 5 x 7 us as separate launches (a node is ~6 us of dependent global round trips
    // inside the kernel either way; the chain saves the launch boundaries).  HEP_CHAIN=0 disables it.
    const bool chain_on = chain_mode != 0;
    auto emit_pools = [&]() {
      for (const Planner::PoolSpec& ps : pending_pools) {
        const TensorDesc& ti = s->tensors[ps.src_t];
        const int ph = ti.H, oh = s->levels[ps.level];
        const int op = P.new_op(OP_POOL, s->tensors[ps.out_t].name.substr(0, s->tensors[ps.out_t].name.size() - 3) + "_pool");
        Op& o = s->ops[op];
        int pb, pa; same_pad(ph, 3, 2, &pb, &pa);
        o.pool.H = ph; o.pool.W = ph; o.pool.C = Wf; o.pool.Ho = oh; o.pool.Wo = oh; o.pool.pad_t = pb; o.pool.pad_l = pb; o.pool.bf16 = s->dtype;
        P.tref(op, F_POOL_IN, ps.src_t, false); P.tref(op, F_POOL_OUT, ps.out_t, true);
        o.act_bytes_per_image = ((double)ph * ph + (double)oh * oh) * Wf * P.es();
      }
      pending_pools.clear();
    };
    auto flush = [&]() {
      if (pending.empty()) { emit_pools(); return; }
      std::string nm = pending_names.front();
      for (size_t i = 1; i < pending_names.size(); i++) nm += "+" + pending_names[i].substr(pending_names[i].find('.') + 1);
      if (chain_mode == 2 && P.add_chain(nm, pending_pools, pending)) pending_pools.clear();
      else { emit_pools(); if (P.ok) P.add_sep(nm, pending, pending.size() > 1); }
      pending.clear(); pending_names.clear();
    };
    auto node = [&](const char* conv, const char* wkey, int nw, int level, std::vector<std::pair<int, int>> srcs, const std::string& oname) {
      fusion_weights(pack, p + "." + wkey, nw, A.attention, w, &P.ok, err);
      Planner::SegSpec sp;
      sp.nsrc = (int)srcs.size(); sp.pre_act = 1; sp.level = level; sp.key = p + "." + conv; sp.bn = p + "." + conv + ".bn";
      sp.N = Wf; sp.act = ACT_NONE; sp.head_out = -1; sp.col_kin = sp.col_kout = 1; sp.col_off = 0; sp.out_k = 0;
      for (int j = 0; j < sp.nsrc; j++) { sp.src[j] = srcs[j].first; sp.kind[j] = srcs[j].second; sp.fw[j] = w[j]; }
      sp.out_t = P.tensor(oname, s->levels[level], s->levels[level], Wf);
      if (chain_on && s->levels[level] <= P.sep_tile_side() && Wf <= SEP_MAX_TILES_MAP * 16) { pending.push_back(sp); pending_names.push_back(tn + conv); }
      else { flush(); P.add_sep(tn + conv, {sp}); }
      return sp.out_t;
    };
    const int p6_up = node("conv6_up", "p6_w1", 2, 3, {{in[3], SRC_SAME}, {in[4], SRC_UP}}, tn + "p6_up"); if (!P.ok) return HEP_ERR_PACK;
    const int p5_up = node("conv5_up", "p5_w1", 2, 2, {{in[2], SRC_SAME}, {p6_up, SRC_UP}}, tn + "p5_up"); if (!P.ok) return HEP_ERR_PACK;
    const int p4_up = node("conv4_up", "p4_w1", 2, 1, {{in[1], SRC_SAME}, {p5_up, SRC_UP}}, tn + "p4_up"); if (!P.ok) return HEP_ERR_PACK;
    const int p3_out = node("conv3_up", "p3_w1", 2, 0, {{in[0], SRC_SAME}, {p4_up, SRC_UP}}, tn + "p3_out"); if (!P.ok) return HEP_ERR_PACK;
    const int p4_out = node("conv4_down", "p4_w2", 3, 1, {{in2[1], SRC_SAME}, {p4_up, SRC_SAME}, {p3_out, SRC_DOWN}}, tn + "p4_out"); if (!P.ok) return HEP_ERR_PACK;
    const int p5_out = node("conv5_down", "p5_w2", 3, 2, {{in2[2], SRC_SAME}, {p5_up, SRC_SAME}, {p4_out, SRC_DOWN}}, tn + "p5_out"); if (!P.ok) return HEP_ERR_PACK;
    const int p6_out = node("conv6_down", "p6_w2", 3, 3, {{in[3], SRC_SAME}, {p6_up, SRC_SAME}, {p5_out, SRC_DOWN}}, tn + "p6_out"); if (!P.ok) return HEP_ERR_PACK;
    const int p7_out = node("conv7_down", "p7_w2", 2, 4, {{in[4], SRC_SAME}, {p6_out, SRC_DOWN}}, tn + "p7_out"); if (!P.ok) return HEP_ERR_PACK;
    feat[0] = p3_out; feat[1] = p4_out; feat[2] = p5_out; feat[3] = p6_out; feat[4] = p7_out;
  }
  if (!pending.empty()) {
    std::string nm = pending_names.front();
    for (size_t i = 1; i < pending_names.size(); i++) nm += "+" + pending_names[i].substr(pending_names[i].find('.') + 1);
    if (!(chain_mode == 2 && P.add_chain(nm, {}, pending))) P.add_sep(nm, pending, pending.size() > 1);
    pending.clear(); pending_names.clear();
  }
  if (!P.ok) return HEP_ERR_PACK;
  for (int l = 0; l < 5; l++) s->feat_ids[l] = feat[l];

  // ---- heads ----
  static const char* nets[5] = {"regressor", "classifier", "rotation_net", "translation_net", "hand_net"};
  struct Hd { int net; const char* key; int kin, kout, off, out, act; };
  const int NC = s->num_classes;     // per anchor the classifier emits one column per class (efficientdet/model.py:406-408)
  const Hd hds[6] = {{0, "regressor.header", 4, 4, 0, 0, ACT_NONE}, {1, "classifier.header", NC, NC, 0, 1, ACT_SIGMOID},
                            {2, "rotation_net.initial_rotation", 3, 3, 0, 2, ACT_NONE},
                            {3, "translation_net.initial_translation_xy", 2, 3, 0, 3, ACT_NONE},
                            {3, "translation_net.initial_translation_z", 1, 3, 2, 3, ACT_NONE},
                            {4, "hand_net.initial_hand_coords", 63, 63, 0, 4, ACT_NONE}};
  // HEP_HEADS_FUSED: 1 = the towers depth-first in ONE launch (k_heads.hip; bf16 at BiFPN width 64), 0 = launch by launch
  bool heads_done = false;
#ifdef HEP_ALT
  if (P.kn.heads_fused != 0) {
    std::vector<std::string> nv(nets, nets + 5);
    std::vector<Planner::HeaderSpec> hv;
    for (const Hd& h : hds) hv.push_back({h.net, h.key, h.kin, h.kout, h.off, h.out, h.act});
    heads_done = P.add_heads_fused(nv, hv, feat, A.head_depth);
    if (!P.ok) return HEP_ERR_PACK;
  }
#endif
  // one launch per tower layer (all five nets x five levels) + one for all headers (k_tower.hip)
  if (!heads_done) {
    int cur[5][5];
    for (int n = 0; n < 5; n++) for (int l = 0; l < 5; l++) cur[n][l] = feat[l];
    for (int i = 0; i < A.head_depth; i++) {
      std::vector<Planner::SegSpec> specs;
      for (int n = 0; n < 5; n++)
        for (int l = 0; l < 5; l++) {
          Planner::SegSpec sp;
          sp.nsrc = 1; sp.src[0] = cur[n][l]; sp.kind[0] = SRC_SAME; sp.fw[0] = 1.f; sp.pre_act = 0; sp.level = l;
          sp.key = std::string(nets[n]) + ".conv_list." + std::to_string(i);
          sp.bn = std::string(nets[n]) + ".bn_list." + std::to_string(l) + "." + std::to_string(i);   // per-level BN, shared conv
          sp.N = Wf; sp.act = ACT_SWISH; sp.head_out = -1; sp.col_kin = sp.col_kout = 1; sp.col_off = 0; sp.out_k = 0;
          sp.out_t = P.tensor(std::string(nets[n]) + ".t" + std::to_string(i) + ".p" + std::to_string(l + 3), s->levels[l], s->levels[l], Wf);
          specs.push_back(sp);
        }
      P.add_sep("heads.tower" + std::to_string(i), specs);
      if (!P.ok) return HEP_ERR_PACK;
      for (int n = 0; n < 5; n++) for (int l = 0; l < 5; l++) cur[n][l] = specs[n * 5 + l].out_t;
    }
    {
      std::vector<Planner::SegSpec> specs;
      for (const Hd& h : hds)
        for (int l = 0; l < 5; l++) {
          Planner::SegSpec sp;
          sp.nsrc = 1; sp.src[0] = cur[h.net][l]; sp.kind[0] = SRC_SAME; sp.fw[0] = 1.f; sp.pre_act = 0; sp.level = l;
          sp.key = h.key; sp.bn = ""; sp.N = 9 * h.kin; sp.act = h.act; sp.out_t = -1; sp.head_out = h.out;
          sp.col_kin = h.kin; sp.col_kout = h.kout; sp.col_off = h.off; sp.out_k = h.kout;
          specs.push_back(sp);
        }
      P.add_sep("heads.headers", specs);
      if (!P.ok) return HEP_ERR_PACK;
    }
  }

  // ---- arena layout: first-fit with liveness-based reuse ----
  const int nops = (int)s->ops.size();
  for (int i = 0; i < nops; i++) {
    for (int t : s->ops[i].writes) { if (s->tensors[t].first_op < 0) s->tensors[t].first_op = i; s->tensors[t].last_op = std::max(s->tensors[t].last_op, i); }
    for (int t : s->ops[i].reads) s->tensors[t].last_op = std::max(s->tensors[t].last_op, i);
  }
  for (int l = 0; l < 5; l++) s->tensors[feat[l]].last_op = nops;   // exported after the last op
  {
    struct Free { size_t off, size; };
    std::vector<Free> freelist; size_t top = 0;
#ifdef HEP_POISON_LDS     // sanitizer build: every tensor keeps a place of its own, followed by a guard band that hep_destroy checks (hep_api.cpp)
    const bool keep = true;
    const size_t guard = 256;
#else
    const bool keep = s->flags & 1u;
    const size_t guard = 0;
#endif
    for (int i = 0; i < nops; i++) {
      for (int t : s->ops[i].writes) {
        TensorDesc& td = s->tensors[t];
        if (td.first_op != i) continue;
        const size_t need = (((td.bytes_per_image * s->lane_batch) + 255) & ~(size_t)255) + guard;
        bool placed = false;
        if (!keep)
          for (size_t f = 0; f < freelist.size(); f++)
            if (freelist[f].size >= need) {
              td.offset = freelist[f].off;
              freelist[f].off += need; freelist[f].size -= need;
              if (freelist[f].size == 0) freelist.erase(freelist.begin() + f);
              placed = true; break;
            }
        if (!placed) { td.offset = top; top += need; }
      }
      if (!keep) {
        std::vector<int> touched = s->ops[i].reads; touched.insert(touched.end(), s->ops[i].writes.begin(), s->ops[i].writes.end());
        std::sort(touched.begin(), touched.end()); touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
        for (int t : touched) {
          TensorDesc& td = s->tensors[t];
          if (td.last_op != i) continue;
          const size_t sz = (((td.bytes_per_image * s->lane_batch) + 255) & ~(size_t)255) + guard;
          freelist.push_back({td.offset, sz});
          // coalesce neighbours
          std::sort(freelist.begin(), freelist.end(), [](const Free& a, const Free& b) { return a.off < b.off; });
          for (size_t f = 0; f + 1 < freelist.size();)
            if (freelist[f].off + freelist[f].size == freelist[f + 1].off) { freelist[f].size += freelist[f + 1].size; freelist.erase(freelist.begin() + f + 1); }
            else f++;
        }
      }
    }
    s->arena_bytes = top;
  }

  // ---- device allocations ----
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { *err = std::string(#x) + ": " + hipGetErrorString(e_); return HEP_ERR_DEVICE; } } while (0)
  HIPCHK(hipSetDevice(s->device));
  if (mbf_prepare() != 0) { *err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for mbf_kernel"; return HEP_ERR_DEVICE; }
  if (xbf_prepare() != 0) { *err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for xbf_kernel"; return HEP_ERR_DEVICE; }
  if (chain_prepare() != 0) { *err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for chain_kernel"; return HEP_ERR_DEVICE; }
  if (tower_prepare() != 0) { *err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for tower_kernel"; return HEP_ERR_DEVICE; }
  if (filter_prepare() != 0) { *err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for filter_kernel"; return HEP_ERR_DEVICE; }
  if (sep_prepare() != 0) { *err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"; return HEP_ERR_DEVICE; }
#ifdef HEP_ALT
  if (heads_prepare() != 0) { *err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for heads_kernel"; return HEP_ERR_DEVICE; }
  if (late_prepare() != 0) { *err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for late_kernel"; return HEP_ERR_DEVICE; }
#endif
  HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  s->weights_bytes = P.wb.host.size();
  HIPCHK(hipMalloc((void**)&s->d_weights, s->weights_bytes));
  HIPCHK(hipMemcpy(s->d_weights, P.wb.host.data(), s->weights_bytes, hipMemcpyHostToDevice));
  s->arena_bytes = (s->arena_bytes + 255) & ~(size_t)255;
  HIPCHK(hipMalloc((void**)&s->d_arena, std::max<size_t>(s->arena_bytes * s->lanes, 256)));
#ifdef HEP_POISON_LDS     // sanitizer build: activation cells nobody has written yet read as NaN (0xFFFF / 0xFFFFFFFF)
  HIPCHK(hipMemset(s->d_arena, 0xFF, std::max<size_t>(s->arena_bytes * s->lanes, 256)));
#endif
  for (int i = 0; i < 5; i++) HIPCHK(hipMalloc((void**)&s->d_out[i], (size_t)s->max_batch * s->num_anchors * s->out_k(i) * 4));
  {
    std::vector<float> a, t; host_anchors(S, &a, &t);
    HIPCHK(hipMalloc((void**)&s->d_anchors, a.size() * 4)); HIPCHK(hipMemcpy(s->d_anchors, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&s->d_tanchors, t.size() * 4)); HIPCHK(hipMemcpy(s->d_tanchors, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  }
  HIPCHK(hipEventCreateWithFlags(&s->fork_event, hipEventDisableTiming));
  // per lane: 16 words per image for the grouped late kernel, then P.ntails blocks of one 128-byte line per image (arrival tickets of the fronts' tails)
  const size_t sync_lane_words = (size_t)s->lane_batch * 16 + (size_t)P.ntails * s->lane_batch * 32;
  HIPCHK(hipMalloc((void**)&s->d_sync, s->lanes * sync_lane_words * 4 + 64));
  HIPCHK(hipMemset(s->d_sync, 0, s->lanes * sync_lane_words * 4 + 64));
  // ---- one patched copy of the plan per lane: own arena slice, own slice of the head outputs ----
  s->lane_ops.assign(s->lanes, s->ops);
  for (int lane = 0; lane < s->lanes; lane++) {
    hipStream_t st; HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); s->lane_streams.push_back(st);
    hipEvent_t ev; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); s->lane_events.push_back(ev);
    std::vector<Op>& ops = s->lane_ops[lane];
    for (const Ref& r : P.refs) {
      Op& o = ops[r.op];
      void* ptr;
      if (r.tensor >= 0) ptr = s->tptr(r.tensor, lane);
      else if (r.tensor <= -2) { const int k = -(r.tensor + 2); ptr = s->d_out[k] + (size_t)lane * s->lane_batch * s->num_anchors * s->out_k(k); }
      else ptr = s->d_weights + r.woff;
      switch (r.field) {
        case F_STEM_W: o.stem.w = (const float*)ptr; break;
        case F_STEM_B: o.stem.bias = (const float*)ptr; break;
        case F_STEM_OUT: o.stem.out = ptr; break;
        case F_PW_A: o.pw.A = ptr; break;
        case F_PW_W: o.pw.W = ptr; break;
        case F_PW_B: o.pw.bias = (const float*)ptr; break;
        case F_PW_HPART: o.pw.hpart = (const float*)ptr; break;
        case F_PW_SEBR: o.pw.se_br = (const float*)ptr; break;
        case F_PW_SEWE: o.pw.se_we = ptr; break;
        case F_PW_SEBE: o.pw.se_be = (const float*)ptr; break;
        case F_PW_RES: o.pw.res = ptr; break;
        case F_PW_OUT: o.pw.out = ptr; break;
        case F_DW_IN: o.dw.in = ptr; break;
        case F_DW_W: o.dw.w = (const float*)ptr; break;
        case F_DW_B: o.dw.bias = (const float*)ptr; break;
        case F_DW_OUT: o.dw.out = ptr; break;
        case F_DW_PART: o.dw.hpart = (float*)ptr; break;
        case F_DW_WR: o.dw.se_wr = (const float*)ptr; break;
        case F_MBF_IN: o.mbf.in = ptr; break;
        case F_MBF_WE: o.mbf.we = ptr; break;
        case F_MBF_BE: o.mbf.be = (const float*)ptr; break;
        case F_MBF_WDW: o.mbf.wdw = (const float*)ptr; break;
        case F_MBF_BDW: o.mbf.bdw = (const float*)ptr; break;
        case F_MBF_OUT: o.mbf.out = ptr; break;
        case F_MBF_PART: o.mbf.hpart = (float*)ptr; o.mbf.tail.hpart = (const float*)ptr; break;
        case F_MBFT_SCALE: o.mbf.tail.scale = (float*)ptr; break;
        case F_MBFT_BR: o.mbf.tail.br = (const float*)ptr; break;
        case F_MBFT_WE: o.mbf.tail.we = ptr; break;
        case F_MBFT_BE: o.mbf.tail.be = (const float*)ptr; break;
        case F_MBF_WR: o.mbf.se_wr = (const float*)ptr; break;
        case F_MBF_WESCALE: o.mbf.we_scale = (const float*)ptr; break;
        case F_PW_WSCALE: o.pw.wscale = (const float*)ptr; break;
        case F_PW_SESCALE: o.pw.se_scale = (const float*)ptr; break;
        case F_SE_HPART: o.se.hpart = (const float*)ptr; break;
        case F_SE_SCALE: o.se.scale = (float*)ptr; break;
        case F_SE_BR: o.se.br = (const float*)ptr; break;
        case F_SE_WE: o.se.we = ptr; break;
        case F_SE_BE: o.se.be = (const float*)ptr; break;
        case F_POOL_IN: o.pool.in = ptr; break;
        case F_POOL_OUT: o.pool.out = ptr; break;
        case F_PWG_A: o.pwg.seg[r.seg].A = ptr; break;
        case F_PWG_W: o.pwg.seg[r.seg].W = ptr; break;
        case F_PWG_B: o.pwg.seg[r.seg].bias = (const float*)ptr; break;
        case F_PWG_OUT: o.pwg.seg[r.seg].out = ptr; break;
        case F_SEG_SRC: o.segs[r.seg].src[r.idx] = ptr; break;
        case F_SEG_WDW: o.segs[r.seg].wdw = (const float*)ptr; break;
        case F_SEG_WPW: o.segs[r.seg].wpw = ptr; break;
        case F_SEG_BIAS: o.segs[r.seg].bias = (const float*)ptr; break;
        case F_SEG_OUT: o.segs[r.seg].out = ptr; break;
        case F_CH_EXT_SRC: o.chain.ext[r.seg].src = ptr; break;
        case F_CH_EXT_STORE: o.chain.ext[r.seg].store = ptr; break;
        case F_CH_NODE_OUT: o.cnodes[r.seg].out = ptr; break;
        case F_CH_WBLOB: o.chain.wblob = ptr; break;
#ifdef HEP_ALT
        case F_SBF_WS: o.sbf.w_stem = (const float*)ptr; break;
        case F_SBF_BS: o.sbf.b_stem = (const float*)ptr; break;
        case F_SBF_WDW: o.sbf.wdw = (const float*)ptr; break;
        case F_SBF_BDW: o.sbf.bdw = (const float*)ptr; break;
        case F_SBF_STEM: o.sbf.stem_out = ptr; break;
        case F_SBF_OUT: o.sbf.out = ptr; break;
        case F_SBF_PART: o.sbf.hpart = (float*)ptr; break;
        case F_SBF_WR: o.sbf.se_wr = (const float*)ptr; break;
#endif
        case F_XBF_IN: o.xbf.in = ptr; break;
        case F_XBF_HPART: o.xbf.hpart = (const float*)ptr; break;
        case F_XBF_SEBR: o.xbf.se_br = (const float*)ptr; break;
        case F_XBF_SEWE: o.xbf.se_we = ptr; break;
        case F_XBF_SEBE: o.xbf.se_be = (const float*)ptr; break;
        case F_XBF_BLOB: o.xbf.blob = ptr; break;
        case F_XBF_RES: o.xbf.res = ptr; break;
        case F_XBF_MID: o.xbf.mid = ptr; break;
        case F_XBF_OUT: o.xbf.out = ptr; break;
        case F_XBF_PART: o.xbf.hpart_out = (float*)ptr; break;
        case F_XBF_WR: o.xbf.se_wr = (const float*)ptr; break;
#ifdef HEP_ALT
        case F_HEADS_FEAT: o.heads.feat[r.seg] = ptr; break;
        case F_HEADS_BLOB: o.heads.blob = (const unsigned char*)ptr; break;
        case F_HEADS_OUT: o.heads.out[r.seg] = (float*)ptr; break;
        case F_LATE_IN: o.late.in = ptr; break;
        case F_LATE_BLOB: o.late.blob = (const unsigned char*)ptr; break;
        case F_LATE_DS: o.late.dscratch = ptr; break;
        case F_LATE_HPART: o.late.hpart = (float*)ptr; break;
        case F_LATE_RES: o.late.blk[r.seg].res = ptr; break;
        case F_LATE_OUT: o.late.blk[r.seg].out = ptr; break;
#endif
        default: break;
      }
    }
    for (Op& o : ops) {
#ifdef HEP_ALT
      if (o.kind == OP_LATE) o.late.counters = s->d_sync + (size_t)lane * sync_lane_words;
#endif
      if (o.kind == OP_MBF && o.mbf.se_tail) o.mbf.tail_counter = s->d_sync + (size_t)lane * sync_lane_words + (size_t)s->lane_batch * 16 + (size_t)(o.mbf.se_tail - 1) * s->lane_batch * 32;
    }
    // segment tables to device
    for (Op& o : ops)
      if (o.kind == OP_SEP) {
        SepSeg* d; HIPCHK(hipMalloc((void**)&d, o.segs.size() * sizeof(SepSeg)));
        HIPCHK(hipMemcpy(d, o.segs.data(), o.segs.size() * sizeof(SepSeg), hipMemcpyHostToDevice));
        o.sep.segs = d;
        std::vector<int> tile_seg(o.sep.total_tiles);
        for (size_t si = 0; si < o.segs.size(); si++)
          for (int t = 0; t < o.segs[si].tiles_x * o.segs[si].tiles_y; t++) tile_seg[o.segs[si].tile_begin + t] = (int)si;
        int* dt; HIPCHK(hipMalloc((void**)&dt, tile_seg.size() * sizeof(int)));
        HIPCHK(hipMemcpy(dt, tile_seg.data(), tile_seg.size() * sizeof(int), hipMemcpyHostToDevice));
        o.sep.tile_seg = dt;
        o.sep.seg0 = o.segs[0];
      } else if (o.kind == OP_CHAIN) {
        // (whole 256-byte chunks: chain_kernel warms the scalar cache with one s_load_dword per 64-byte line of every chunk it touches)
        ChainNode* d; HIPCHK(hipMalloc((void**)&d, (o.cnodes.size() * sizeof(ChainNode) + 255) & ~(size_t)255));
        HIPCHK(hipMemcpy(d, o.cnodes.data(), o.cnodes.size() * sizeof(ChainNode), hipMemcpyHostToDevice));
        o.chain.nodes = d;
      }
  }
#undef HIPCHK
  return 0;
}

}  // namespace hep
