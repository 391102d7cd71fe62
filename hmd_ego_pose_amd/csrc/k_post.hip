// k_post.hip - decode half of the path on gfx950: box / translation decode and the
// detection filter (score threshold -> greedy NMS -> first max_det survivors -> -1 padding).
#include <algorithm>

#include "hep_dev.h"
#include "hep_internal.h"

// ------------------------------------------------------------------------------------------------
// format_bboxes + format_translation (reference hmdegopose/loss.py:12-51 ->
// layers.py:117-249): deltas (ty,tx,th,tw); boxes clipped to [0,S-1];
// Tx=(x/scale-px)*Tz/fx with Tz = raw_z*tz_scale.  Same fp32 operation order as the torch code
// (no fma contraction across the reference's separate ops) so results agree to the last few ulps.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_kernel(DecodeArgs a) {
#pragma clang fp contract(off)
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)a.B * a.N) return;
  const int n = (int)(idx % a.N), b = (int)(idx / a.N);
  const f32x4 an = *reinterpret_cast<const f32x4*>(a.anchors + (int64_t)n * 4);
  const f32x4 d = *reinterpret_cast<const f32x4*>(a.regression + idx * 4);
  const float cxa = (an[0] + an[2]) / 2.f, cya = (an[1] + an[3]) / 2.f;
  const float wa = an[2] - an[0], ha = an[3] - an[1];
  const float ty = d[0], tx = d[1], th = d[2], tw = d[3];
  const float w = expf(tw) * wa, h = expf(th) * ha;
  const float cy = ty * ha + cya, cx = tx * wa + cxa;
  f32x4 o;
  o[0] = fminf(fmaxf(cx - w / 2.f, 0.f), a.clip_max);
  o[1] = fminf(fmaxf(cy - h / 2.f, 0.f), a.clip_max);
  o[2] = fminf(fmaxf(cx + w / 2.f, 0.f), a.clip_max);
  o[3] = fminf(fmaxf(cy + h / 2.f, 0.f), a.clip_max);
  *reinterpret_cast<f32x4*>(a.boxes + idx * 4) = o;

  const float* ta = a.t_anchors + (int64_t)n * 3;
  const float* r = a.translation_raw + idx * 3;
  const float* cam = a.camera + (int64_t)b * 6;
  const float stride = ta[2];
  float x = ta[0] + r[0] * stride, y = ta[1] + r[1] * stride;
  x = x / cam[5] - cam[2];
  y = y / cam[5] - cam[3];
  const float tz = r[2] * cam[4];
  float* t = a.translation + idx * 3;
  t[0] = x * tz / cam[0];
  t[1] = y * tz / cam[1];
  t[2] = tz;
}
void launch_decode(const DecodeArgs& a, hipStream_t s) {
  const int64_t total = (int64_t)a.B * a.N;
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// filter_detections (reference hmdegopose/layers.py:264-400).  class_specific_filter=True (the mode the reference
// constructs, train.py:78-81): one workgroup (1024 lanes) per (image, class); with more than one class
// filter_merge_kernel then takes the top max_detections of all classes' survivors.  class_specific_filter=False
// (layers.py:359-362, FilterArgs::any_class): one workgroup per image over every anchor's best class (max / FIRST argmax over
// the class columns); its NMS order already is the top_k order, so the rows are emitted directly with label = that argmax.
//   1. key[n] = score>thr ? (score_bits << 32 | ~n) : 0      (positive floats order as integers)
//   2. bitonic sort, descending: score desc, equal scores -> lower anchor index first
//   3. greedy NMS over the sorted candidates in chunks of 1024: a candidate dies when its IoU
//      with an already kept box is STRICTLY greater than nms_thr (TensorFlow semantics: plain
//      areas, no +1); stop at max_det kept
//   4. gather the survivors' rows, pad the rest with -1.
// Index work is exact: the same anchors come out as from oracle/decode_ref.py bit for bit.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float iou_box(const f32x4 p, const f32x4 q) {
#pragma clang fp contract(off)
  const float ax0 = fminf(p[0], p[2]), ay0 = fminf(p[1], p[3]), ax1 = fmaxf(p[0], p[2]), ay1 = fmaxf(p[1], p[3]);
  const float bx0 = fminf(q[0], q[2]), by0 = fminf(q[1], q[3]), bx1 = fmaxf(q[0], q[2]), by1 = fmaxf(q[1], q[3]);
  const float aa = (ax1 - ax0) * (ay1 - ay0), ab = (bx1 - bx0) * (by1 - by0);
  if (aa <= 0.f || ab <= 0.f) return 0.f;
  const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.f);
  const float ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.f);
  const float inter = iw * ih;
  return inter / ((aa + ab) - inter);
}

// score and class of an anchor's best class: max over the K columns, the first column on ties (keras.backend.max / argmax)
__device__ __forceinline__ float best_class(const float* row, int K, int* label) {
  float m = row[0]; int l = 0;
  for (int c = 1; c < K; c++) { const float v = row[c]; if (v > m) { m = v; l = c; } }
  *label = l;
  return m;
}

#define FILTER_THREADS 1024
#define FILTER_LDS_KEYS 16384      // candidates sorted in LDS (128 KB); more than that fall back to the global-memory sort

// 64-bit bitonic sort, descending, n a power of two, by the whole workgroup (keys in LDS or global memory)
__device__ __forceinline__ void bitonic_desc(uint64_t* keys, int n, int tid) {
  for (int k = 2; k <= n; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n; i += FILTER_THREADS) {
        const int l = i ^ j;
        if (l > i) {
          const uint64_t x = keys[i], y = keys[l];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) { keys[i] = y; keys[l] = x; }
        }
      }
      __syncthreads();
    }
}

// rows of one image: detection i < nk is anchor anchor_of(i) with class label_of(i); the rest of the max_det rows are -1
template <class FA, class FL>
__device__ __forceinline__ void filter_emit(const FilterArgs& a, int b, int nk, int tid, FA anchor_of, FL label_of) {
  const f32x4* boxes = reinterpret_cast<const f32x4*>(a.boxes) + (int64_t)b * a.N;
  if (tid == 0) a.det_count[b] = nk;
  for (int i = tid; i < a.max_det; i += FILTER_THREADS) {
    const bool ok = i < nk;
    const int n = ok ? anchor_of(i) : 0, label = ok ? label_of(i) : 0;
    const int64_t row = (int64_t)b * a.max_det + i;
    const int64_t src = (int64_t)b * a.N + n;
    if (a.det_index) a.det_index[row] = ok ? n : -1;
    if (a.det_scores) a.det_scores[row] = ok ? a.scores[src * a.K + label] : -1.f;
    if (a.det_labels) a.det_labels[row] = ok ? label : -1;
    if (a.det_boxes) {
      const f32x4 bb = ok ? boxes[n] : (f32x4){-1.f, -1.f, -1.f, -1.f};
      *reinterpret_cast<f32x4*>(a.det_boxes + row * 4) = bb;
    }
    if (a.det_rotation) for (int c = 0; c < 3; c++) a.det_rotation[row * 3 + c] = ok ? a.rotation[src * 3 + c] : -1.f;
    if (a.det_translation) for (int c = 0; c < 3; c++) a.det_translation[row * 3 + c] = ok ? a.translation[src * 3 + c] : -1.f;
  }
  if (a.det_hand)
    for (int i = tid; i < a.max_det * 63; i += FILTER_THREADS) {
      const int d = i / 63, c = i % 63;
      a.det_hand[((int64_t)b * a.max_det + d) * 63 + c] = d < nk ? a.hand[((int64_t)b * a.N + anchor_of(d)) * 63 + c] : -1.f;
    }
}

// Steps: (1) the candidates (score > thr) are COMPACTED into LDS - a trained network passes a handful of the 12 276 anchors,
// and sorting all of them cost 0.9 ms per batch; (2) only the next power of two of their count is sorted (their order before
// the sort does not matter: keys are unique, the sorted order is deterministic); (3) greedy NMS by ONE wave with ballots
// instead of workgroup barriers: 64 sorted candidates at a time, first against the boxes kept so far, then among
// themselves in order; (4) gather.
__global__ __launch_bounds__(FILTER_THREADS) void filter_kernel(FilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lkeys[];          // [cap] candidate keys
  HEP_POISON(lkeys, (size_t)min(a.npow2, FILTER_LDS_KEYS) * 8);             // (launch_filter's size)
  __shared__ f32x4 kept_box[FILTER_MAX_DET];
  __shared__ int kept_idx[FILTER_MAX_DET];
  __shared__ int s_nkept, s_count;
  const int b = blockIdx.x, cls = blockIdx.y, K = a.K, tid = threadIdx.x, lane = tid & 63;
  const bool anyc = a.any_class != 0 && K > 1;                              // (uniform; grid.y is 1 then)
  const float* scores = a.scores + (int64_t)b * a.N * K + cls;              // [N][K]: this class's column
  auto score_of = [&](int n) { int l; return anyc ? best_class(scores + (int64_t)n * K, K, &l) : scores[(int64_t)n * K]; };
  const f32x4* boxes = reinterpret_cast<const f32x4*>(a.boxes) + (int64_t)b * a.N;
  const int cap = min(a.npow2, FILTER_LDS_KEYS);

  if (tid == 0) { s_count = 0; s_nkept = 0; }
  __syncthreads();
  // (1) compaction: one LDS atomic per wave and pass
  for (int n0 = 0; n0 < a.N; n0 += FILTER_THREADS) {
    const int n = n0 + tid;
    const float sc = n < a.N ? score_of(n) : 0.f;
    const bool cand = n < a.N && sc > a.score_thr;
    const unsigned long long m = __ballot(cand);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_count, __popcll(m));
      base = __shfl(base, 0, 64);
      const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
      if (cand && pos < cap) lkeys[pos] = ((uint64_t)__float_as_uint(sc) << 32) | (uint32_t)(~(uint32_t)n);
    }
  }
  __syncthreads();
  const int M = s_count;
  uint64_t* keys = lkeys;
  int np2 = 64;
  if (M <= cap) {
    while (np2 < M) np2 <<= 1;
    for (int i = M + tid; i < np2; i += FILTER_THREADS) lkeys[i] = 0;
    __syncthreads();
    bitonic_desc(lkeys, np2, tid);
  } else {
    // more candidates than LDS holds (large images, untrained scores): all anchors, sorted in global memory
    keys = a.keys + (int64_t)(b * K + cls) * a.npow2; np2 = a.npow2;
    for (int n = tid; n < a.npow2; n += FILTER_THREADS) {
      uint64_t k = 0;
      if (n < a.N) {
        const float sc = score_of(n);
        if (sc > a.score_thr) k = ((uint64_t)__float_as_uint(sc) << 32) | (uint32_t)(~(uint32_t)n);
      }
      keys[n] = k;
    }
    __syncthreads();
    bitonic_desc(keys, np2, tid);
  }

  // (3) greedy NMS, wave 0 only: no workgroup barrier inside.  A candidate dies when its IoU with an already kept box is
  // STRICTLY greater than nms_thr.
  if (tid < 64) {
    int nkept = 0;
    const int ncand = min(M, np2);
    for (int base = 0; base < ncand && nkept < a.max_det; base += 64) {
      const int i = base + lane;
      const uint64_t key = i < ncand ? keys[i] : 0;
      bool alive = key != 0;
      const int n = (int)(~(uint32_t)key);
      f32x4 mine = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (alive) mine = boxes[n];
      for (int j = 0; j < nkept; j++)
        if (alive && iou_box(mine, kept_box[j]) > a.nms_thr) alive = false;
      unsigned long long live = __ballot(alive);
      while (live && nkept < a.max_det) {
        const int w = __ffsll((long long)live) - 1;                   // first surviving candidate of the group: kept
        f32x4 kb;
        kb[0] = __shfl(mine[0], w, 64); kb[1] = __shfl(mine[1], w, 64); kb[2] = __shfl(mine[2], w, 64); kb[3] = __shfl(mine[3], w, 64);
        const int kn = __shfl(n, w, 64);
        if (lane == 0) { kept_box[nkept] = kb; kept_idx[nkept] = kn; }
        nkept++;
        if (lane == w) alive = false;
        else if (alive && lane > w && iou_box(mine, kb) > a.nms_thr) alive = false;
        live = __ballot(alive);
      }
    }
    if (lane == 0) s_nkept = nkept;
  }
  __syncthreads();
  const int nk = s_nkept;
  if (anyc) {        // best class per anchor: NMS order = descending score, ties by anchor = the top_k order of layers.py:365-367
    filter_emit(a, b, nk, tid, [&](int i) { return kept_idx[i]; }, [&](int i) { int l; best_class(scores + (int64_t)kept_idx[i] * K, K, &l); return l; });
    return;
  }
  if (K > 1) {       // one of several classes: the survivors go to filter_merge_kernel (top-k over all classes)
    for (int i = tid; i < nk; i += FILTER_THREADS) a.part_idx[(int64_t)(b * K + cls) * a.max_det + i] = kept_idx[i];
    if (tid == 0) a.part_cnt[b * K + cls] = nk;
    return;
  }
  filter_emit(a, b, nk, tid, [&](int i) { return kept_idx[i]; }, [&](int) { return 0; });
}

// num_classes > 1 (layers.py:347-380): the (anchor, class) pairs every class kept, concatenated class by class, then
// tf.nn.top_k over their scores - descending, ties to the earlier pair - and the first max_detections rows.  One workgroup
// per image sorts at most K * max_det <= 16 128 keys (score bits | ~position in the concatenation) in LDS.
__global__ __launch_bounds__(FILTER_THREADS) void filter_merge_kernel(FilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lkeys[];
  const int b = blockIdx.x, tid = threadIdx.x, K = a.K, total = K * a.max_det;
  int np2 = 64;
  while (np2 < total) np2 <<= 1;
  HEP_POISON(lkeys, (size_t)np2 * 8);
  __shared__ int s_total;
  if (tid == 0) { int t = 0; for (int c = 0; c < K; c++) t += a.part_cnt[b * K + c]; s_total = t; }
  for (int p = tid; p < np2; p += FILTER_THREADS) {
    uint64_t key = 0;
    if (p < total) {
      const int c = p / a.max_det, i = p - c * a.max_det;
      if (i < a.part_cnt[b * K + c]) {
        const int n = a.part_idx[(int64_t)(b * K + c) * a.max_det + i];
        key = ((uint64_t)__float_as_uint(a.scores[((int64_t)b * a.N + n) * K + c]) << 32) | (uint32_t)(~(uint32_t)p);
      }
    }
    lkeys[p] = key;
  }
  __syncthreads();
  bitonic_desc(lkeys, np2, tid);
  const int nk = min(s_total, a.max_det);
  auto pos = [&](int i) { return (int)(~(uint32_t)lkeys[i]); };
  filter_emit(a, b, nk, tid, [&](int i) { const int p = pos(i), c = p / a.max_det; return a.part_idx[(int64_t)(b * K + c) * a.max_det + (p - c * a.max_det)]; },
              [&](int i) { return pos(i) / a.max_det; });
}
int filter_prepare(void) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(filter_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FILTER_LDS_KEYS * 8) != hipSuccess) return -1;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(filter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FILTER_LDS_KEYS * 8) == hipSuccess ? 0 : -1;
}
void launch_filter(const FilterArgs& a, hipStream_t s) {
  const size_t lds = (size_t)std::min(a.npow2, FILTER_LDS_KEYS) * 8;
  const bool anyc = a.any_class != 0 && a.K > 1;
  hipLaunchKernelGGL(filter_kernel, dim3(a.B, anyc ? 1 : a.K), dim3(FILTER_THREADS), lds, s, a);
  if (a.K > 1 && !anyc) {
    int np2 = 64;
    while (np2 < a.K * a.max_det) np2 <<= 1;
    hipLaunchKernelGGL(filter_merge_kernel, dim3(a.B), dim3(FILTER_THREADS), (size_t)np2 * 8, s, a);
  }
}

// ------------------------------------------------------------------------------------------------
// preprocess_image (reference generators/colibri_common.py:622-656): uint8 RGB [B,H,W,3] ->
//   cv2.resize(image, (resized_width, resized_height)) with scale = S / max(H, W), the longer side = S and the
//   shorter one int(side * scale)                                                (skipped when scale == 1)
//   image.astype(float32); image /= 255.; image -= mean; image /= std            (numpy evaluates the two in-place
//   operations with the float64 lists in double and rounds to float32 each time: done exactly so here - the
//   no-resize output is bit-identical to numpy's)
//   zero-pad at the bottom / right to S x S
// The resize is OpenCV's 8-bit INTER_LINEAR as documented in its source (imgproc/resize.cpp, restated from memory:
// cv2 is not in this image, so this convention is PARITY-UNPINNED and stated in DESIGN.md): with an explicit output
// size the per-axis inverse scale is src / dst; source coordinate of output x: fx = (float)((x + 0.5) * (W / nw) - 0.5),
// sx = floor(fx), fx -= sx, clamped at the borders with the weight of the missing tap set to zero; weights as
// shorts round(w * 2048); horizontal pass in int32 (src[sx]*a0 + src[sx+1]*a1), vertical pass
// (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void resize_tap(int o, double inv_scale, int n, int* s0, int* s1, int* w0, int* w1) {
#pragma clang fp contract(off)      // OpenCV (and the oracle) round the product before subtracting 0.5
  float f = (float)(((double)o + 0.5) * inv_scale - 0.5);
  int i = (int)floorf(f);
  f -= (float)i;
  if (i < 0) { f = 0.f; i = 0; }
  if (i >= n - 1) { f = 0.f; i = n - 1; }
  *s0 = i; *s1 = min(i + 1, n - 1);
  *w0 = (int)lrintf((1.f - f) * 2048.f); *w1 = (int)lrintf(f * 2048.f);
}

__global__ __launch_bounds__(256) void preprocess_kernel(PreprocArgs a) {
  const int64_t total = (int64_t)a.B * a.S * a.S * 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % 3);
  const int64_t p = idx / 3;
  const int x = (int)(p % a.S), y = (int)((p / a.S) % a.S), b = (int)(p / ((int64_t)a.S * a.S));
  float v = 0.f;
  if (y < a.nh && x < a.nw) {
    const uint8_t* img = a.in + (int64_t)b * a.H * a.W * 3 + c;
    int u8;
    if (!a.resize) u8 = img[((int64_t)y * a.W + x) * 3];
    else {
      int x0, x1, a0, a1, y0, y1, b0, b1;
      resize_tap(x, a.inv_scale_x, a.W, &x0, &x1, &a0, &a1);
      resize_tap(y, a.inv_scale_y, a.H, &y0, &y1, &b0, &b1);
      const int S0 = img[((int64_t)y0 * a.W + x0) * 3] * a0 + img[((int64_t)y0 * a.W + x1) * 3] * a1;
      const int S1 = img[((int64_t)y1 * a.W + x0) * 3] * a0 + img[((int64_t)y1 * a.W + x1) * 3] * a1;
      u8 = min(255, max(0, (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2));
    }
    const double mean[3] = {0.485, 0.456, 0.406}, sd[3] = {0.229, 0.224, 0.225};
    const float r1 = __fdiv_rn((float)u8, 255.0f);
    const float r2 = (float)((double)r1 - mean[c]);
    v = (float)((double)r2 / sd[c]);
  }
  a.out[idx] = v;
}

// ------------------------------------------------------------------------------------------------
// The frame callback of the reference's streaming app (unity-sandbox/WebRTCNetCoreSandbox/Program.cs:140-205, 381-445):
// cvtColor(YUV2BGR_YV12) -> centre crop -> cv2.resize -> ResizeAndNormalizeMat -> blobFromImage.  OpenCV's ITU-R BT.601
// fixed-point conversion and its 8-bit INTER_LINEAR are restated (oracle/decode_ref.py: yv12_to_bgr, resize_bilinear_u8;
// PARITY UNPINNED - cv2 is not in this image); the app's own quirks are kept: the I420 bytes are read as YV12 (U and V
// exchanged) and the RGB mean / std meet the channels in B, G, R order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void yv12_crop_kernel(Yv12Args a) {
  const int64_t total = (int64_t)a.B * a.crop * a.crop;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % a.crop), y = (int)((idx / a.crop) % a.crop), b = (int)(idx / ((int64_t)a.crop * a.crop));
  const int sx = a.ow + x, sy = a.oh + y, hw = a.H * a.W, q = (a.H / 2) * (a.W / 2);
  const uint8_t* f = a.in + (int64_t)b * (hw + 2 * q);
  const int Y = f[(int64_t)sy * a.W + sx];
  const int V = f[hw + (sy / 2) * (a.W / 2) + sx / 2], U = f[hw + q + (sy / 2) * (a.W / 2) + sx / 2];      // YV12: Y, V, U planes
  const int yy = max(0, Y - 16) * 1220542, uu = U - 128, vv = V - 128, half = 1 << 19;
  const int r = (yy + half + 1673527 * vv) >> 20, g = (yy + half - 852492 * vv - 409993 * uu) >> 20, bl = (yy + half + 2116026 * uu) >> 20;
  uint8_t* o = a.bgr + idx * 3;
  o[0] = (uint8_t)min(255, max(0, bl)); o[1] = (uint8_t)min(255, max(0, g)); o[2] = (uint8_t)min(255, max(0, r));
}
void launch_yv12_crop(const Yv12Args& a, hipStream_t s) {
  const int64_t total = (int64_t)a.B * a.crop * a.crop;
  hipLaunchKernelGGL(yv12_crop_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

// uint8 HWC [B,H,W,3] -> [B,nh,nw,3] (norm == 0), or -> float32 [B,S,S,3]: ((x / 255) - mean) / std in float32 as OpenCV
// computes CV_32F arithmetic (Program.cs:419-427), zero-padded at the bottom / right (norm == 1)
__global__ __launch_bounds__(256) void resize_u8_kernel(ResizeArgs a) {
  const int oh = a.norm ? a.S : a.nh, ow = a.norm ? a.S : a.nw;
  const int64_t total = (int64_t)a.B * oh * ow * 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % 3);
  const int64_t p = idx / 3;
  const int x = (int)(p % ow), y = (int)((p / ow) % oh), b = (int)(p / ((int64_t)ow * oh));
  const bool inside = y < a.nh && x < a.nw;
  int u8 = 0;
  if (inside) {
    const uint8_t* img = a.in + (int64_t)b * a.H * a.W * 3 + c;
    int x0, x1, a0, a1, y0, y1, b0, b1;
    resize_tap(x, a.inv_scale_x, a.W, &x0, &x1, &a0, &a1);
    resize_tap(y, a.inv_scale_y, a.H, &y0, &y1, &b0, &b1);
    const int S0 = img[((int64_t)y0 * a.W + x0) * 3] * a0 + img[((int64_t)y0 * a.W + x1) * 3] * a1;
    const int S1 = img[((int64_t)y1 * a.W + x0) * 3] * a0 + img[((int64_t)y1 * a.W + x1) * 3] * a1;
    u8 = min(255, max(0, (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2));
  }
  if (!a.norm) { reinterpret_cast<uint8_t*>(a.out)[idx] = (uint8_t)u8; return; }
  const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
  float v = 0.f;
  if (inside) v = __fdiv_rn(__fsub_rn(__fdiv_rn((float)u8, 255.0f), mean[c]), sd[c]);
  reinterpret_cast<float*>(a.out)[idx] = v;
}
void launch_resize_u8(const ResizeArgs& a, hipStream_t s) {
  const int64_t total = (int64_t)a.B * (a.norm ? a.S : a.nh) * (a.norm ? a.S : a.nw) * 3;
  hipLaunchKernelGGL(resize_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

void launch_preprocess(const PreprocArgs& a, hipStream_t s) {
  const int64_t total = (int64_t)a.B * a.S * a.S * 3;
  hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// amax of a bf16 tensor (fp8 calibration at hep_create: one power-of-two scale per quantised GEMM input).
// Non-negative floats order like their bit patterns, so the block maxima meet in one integer atomicMax
// (a maximum is order-independent: the result is deterministic).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void amax_kernel(const uint16_t* x, int64_t n, unsigned* out) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(bf16_bits_to_f32(x[i])));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m == m) atomicMax(out, __float_as_uint(m));
}
void launch_amax_bf16(const void* x, int64_t n, unsigned* out, hipStream_t s) {
  const int blocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
  hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(x), n, out);
}
