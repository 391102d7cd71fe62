// k_eval.hip - the pose-error arithmetic of the evaluator on gfx950: ADD and ADD-S of D (ground truth, prediction)
// pose pairs over one object model.
//
// reference: check_6d_pose_add / check_6d_pose_add_s, pytorch-sandbox/eval/common.py:682-746, with the brute-force
// nearest-point search of pytorch-sandbox/generators/utils/calc_min_distances.h:24-35 and the axis-angle ->
// matrix conversion of generators/colibri_common.py:803-813 (cv2.Rodrigues):
//   ADD    mean over ALL model points p of || (R_gt p + t_gt) - (R_pr p + t_pr) ||           (float64, like numpy)
//   ADD-S  both transformed clouds subsampled with step = P / 1000 + 1, cast to float32; for every ground-truth
//          point the minimum over the predicted points of (float) sqrt(d1*d1 + d2*d2 + d3*d3) (float products
//          and sums in that order, no contraction - the C code), then the mean of the minima
// One workgroup per pose pair; the subsampled clouds (<= 1000 points each) live in LDS.  Sums are reduced in a fixed
// order: results are bit-reproducible.  This is HBM/latency-trivial work (a pair is ~1 M distance evaluations) -
// it exists so that the evaluator's metric loop (eval/common.py:866-1121) runs next to the detections it scores.
#include "hep_dev.h"
#include "hep_internal.h"

#define EVAL_THREADS 256
#define EVAL_MAX_SUB 1024

// Rodrigues: R = cos(th) I + (1 - cos(th)) k k^T + sin(th) [k]x, th = |r|, k = r / th; identity below 1e-12
__device__ void rodrigues_f64(const float* rv, double R[9]) {
  const double x = rv[0], y = rv[1], z = rv[2];
  const double th = sqrt(x * x + y * y + z * z);
  if (th < 1e-12) { for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
  const double kx = x / th, ky = y / th, kz = z / th, c = cos(th), s = sin(th), v = 1.0 - c;
  R[0] = c + v * kx * kx;      R[1] = v * kx * ky - s * kz; R[2] = v * kx * kz + s * ky;
  R[3] = v * ky * kx + s * kz; R[4] = c + v * ky * ky;      R[5] = v * ky * kz - s * kx;
  R[6] = v * kz * kx - s * ky; R[7] = v * kz * ky + s * kx; R[8] = c + v * kz * kz;
}

__global__ __launch_bounds__(EVAL_THREADS) void pose_error_kernel(PoseErrArgs a) {
#pragma clang fp contract(off)
  __shared__ double Rg[9], Rp[9], tg[3], tp[3];
  __shared__ float sub_g[EVAL_MAX_SUB][3], sub_p[EVAL_MAX_SUB][3];
  __shared__ double red[EVAL_THREADS];
  const int d = blockIdx.x, t = threadIdx.x;
  if (t == 0) { rodrigues_f64(a.rvec_gt + 3 * d, Rg); for (int i = 0; i < 3; i++) tg[i] = a.t_gt[3 * d + i]; }
  if (t == 64) { rodrigues_f64(a.rvec_pr + 3 * d, Rp); for (int i = 0; i < 3; i++) tp[i] = a.t_pr[3 * d + i]; }
  __syncthreads();
  const int step = a.P / a.max_points + 1, nsub = (a.P + step - 1) / step;
  double acc = 0.0;
  for (int p = t; p < a.P; p += EVAL_THREADS) {
    const double x = a.points[3 * p], y = a.points[3 * p + 1], z = a.points[3 * p + 2];
    // np.dot(points, R.T) + t: row-vector times R^T = R applied to the point, accumulated x, y, z in order
    const double gx = (x * Rg[0] + y * Rg[1]) + z * Rg[2] + tg[0], gy = (x * Rg[3] + y * Rg[4]) + z * Rg[5] + tg[1], gz = (x * Rg[6] + y * Rg[7]) + z * Rg[8] + tg[2];
    const double qx = (x * Rp[0] + y * Rp[1]) + z * Rp[2] + tp[0], qy = (x * Rp[3] + y * Rp[4]) + z * Rp[5] + tp[1], qz = (x * Rp[6] + y * Rp[7]) + z * Rp[8] + tp[2];
    const double dx = gx - qx, dy = gy - qy, dz = gz - qz;
    acc += sqrt((dx * dx + dy * dy) + dz * dz);
    if (p % step == 0) {
      const int i = p / step;
      sub_g[i][0] = (float)gx; sub_g[i][1] = (float)gy; sub_g[i][2] = (float)gz;
      sub_p[i][0] = (float)qx; sub_p[i][1] = (float)qy; sub_p[i][2] = (float)qz;
    }
  }
  red[t] = acc;
  __syncthreads();
  for (int o = EVAL_THREADS / 2; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
  if (t == 0) a.add[d] = red[0] / (double)a.P;
  __syncthreads();
  // ADD-S: float arithmetic exactly as calc_min_distances.h (the minimum of the squared distances first: the
  // final (float) sqrt((double) .) is monotone, so the minimum commutes with it)
  double sacc = 0.0;
  for (int i = t; i < nsub; i += EVAL_THREADS) {
    const float gx = sub_g[i][0], gy = sub_g[i][1], gz = sub_g[i][2];
    float best = 3.4e38f;
    for (int j = 0; j < nsub; j++) {
      const float d1 = gx - sub_p[j][0], d2 = gy - sub_p[j][1], d3 = gz - sub_p[j][2];
      const float s2 = d1 * d1 + d2 * d2 + d3 * d3;
      best = fminf(best, s2);
    }
    sacc += (double)(float)sqrt((double)best);
  }
  red[t] = sacc;
  __syncthreads();
  for (int o = EVAL_THREADS / 2; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
  if (t == 0) a.add_s[d] = red[0] / (double)nsub;
}

void launch_pose_errors(const PoseErrArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(pose_error_kernel, dim3(a.D), dim3(EVAL_THREADS), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// Training-side anchor-target assignment: the reference computes it per batch on the host with
// numpy + a Cython IoU matrix (generators/utils/anchors.py:69-221, compute_overlap.pyx:33-73, bbox_transform
// anchors.py:422-458).  Here one workgroup per image:
//   pass 1  for every ground-truth box k the anchor of greatest IoU (lowest index on ties; anchor 0 for a box that
//           overlaps nothing - numpy's argmax of a zero column): forced positive
//   pass 2  per anchor: IoU with every box in float64 with the "+1" convention, the box of greatest overlap (lowest
//           index on ties), state = 1 if overlap >= positive or forced, -1 if overlap > negative (and not positive) or
//           the anchor's centre lies outside the image, else 0; one-hot label, (ty, tx, th, tw) regression targets
//           (anchor side in float32, box side in float64 like the numpy expressions), transformation / hand targets of
//           the assigned box
// Index decisions are exact (same float64 arithmetic, same tie rules); the regression targets differ from numpy's by
// the last ulps of log().
// ------------------------------------------------------------------------------------------------
#define AT_THREADS 1024
__device__ __forceinline__ double iou_plus1(const float* an, const double* q) {
#pragma clang fp contract(off)
  const double a0 = an[0], a1 = an[1], a2 = an[2], a3 = an[3];
  const double iw = fmin(a2, q[2]) - fmax(a0, q[0]) + 1.0;
  if (!(iw > 0)) return 0.0;
  const double ih = fmin(a3, q[3]) - fmax(a1, q[1]) + 1.0;
  if (!(ih > 0)) return 0.0;
  const double box_area = (q[2] - q[0] + 1.0) * (q[3] - q[1] + 1.0);
  const double ua = (a2 - a0 + 1.0) * (a3 - a1 + 1.0) + box_area - iw * ih;
  return iw * ih / ua;
}

__global__ __launch_bounds__(AT_THREADS) void anchor_targets_kernel(AnchorTargetArgs a) {
#pragma clang fp contract(off)
  __shared__ double red_v[AT_THREADS];
  __shared__ int red_i[AT_THREADS];
  __shared__ int forced[AT_MAX_GT];
  const int b = blockIdx.x, t = threadIdx.x;
  const int K = min(a.num_gt[b], a.kmax);
  const double* gt = a.gt_boxes + (int64_t)b * a.kmax * 4;
  for (int k = 0; k < K; k++) {
    double best = -1.0; int bi = 0x7fffffff;
    for (int i = t; i < a.N; i += AT_THREADS) {
      const double v = iou_plus1(a.anchors + 4 * i, gt + 4 * k);
      if (v > best) { best = v; bi = i; }               // ascending i: the first maximum is kept
    }
    red_v[t] = best; red_i[t] = bi;
    __syncthreads();
    for (int o = AT_THREADS / 2; o > 0; o >>= 1) {
      if (t < o) {
        const double v2 = red_v[t + o]; const int i2 = red_i[t + o];
        if (v2 > red_v[t] || (v2 == red_v[t] && i2 < red_i[t])) { red_v[t] = v2; red_i[t] = i2; }
      }
      __syncthreads();
    }
    if (t == 0) forced[k] = red_i[0];
    __syncthreads();
  }
  const int NC = a.num_classes, RT = a.rt;
  const float img_h = (float)a.image_hw[2 * b], img_w = (float)a.image_hw[2 * b + 1];
  for (int i = t; i < a.N; i += AT_THREADS) {
    const float* an = a.anchors + 4 * i;
    float* lab = a.labels + ((int64_t)b * a.N + i) * (NC + 1);
    float* reg = a.regression + ((int64_t)b * a.N + i) * 5;
    float* tra = a.transformation + ((int64_t)b * a.N + i) * (RT + 1);
    float* crd = a.coords ? a.coords + ((int64_t)b * a.N + i) * 64 : nullptr;
    float state = 0.f;
    int arg = 0;
    for (int c = 0; c < NC; c++) lab[c] = 0.f;
    if (K > 0) {
      double mx = -1.0;
      for (int k = 0; k < K; k++) { const double v = iou_plus1(an, gt + 4 * k); if (v > mx) { mx = v; arg = k; } }
      bool pos = mx >= a.positive_overlap;
      for (int k = 0; k < K; k++) pos = pos || forced[k] == i;
      if (pos) state = 1.f; else if (mx > a.negative_overlap) state = -1.f;
      if (pos) { const int l = a.gt_labels[(int64_t)b * a.kmax + arg]; if (l >= 0 && l < NC) lab[l] = 1.f; }
      // bbox_transform: anchor side float32, box side float64
      float wa = an[2] - an[0], ha = an[3] - an[1];
      const float cxa = an[0] + wa / 2.f, cya = an[1] + ha / 2.f;
      const double* q = gt + 4 * arg;
      double w = q[2] - q[0], h = q[3] - q[1];
      const double cx = q[0] + w / 2.0, cy = q[1] + h / 2.0;
      ha += 1e-7f; wa += 1e-7f; h += 1e-7; w += 1e-7;
      reg[0] = (float)((cy - (double)cya) / (double)ha); reg[1] = (float)((cx - (double)cxa) / (double)wa);
      reg[2] = (float)log(h / (double)ha); reg[3] = (float)log(w / (double)wa);
      for (int j = 0; j < RT; j++) tra[j] = a.gt_transform[((int64_t)b * a.kmax + arg) * RT + j];
      if (crd) for (int j = 0; j < 63; j++) crd[j] = a.gt_coords ? a.gt_coords[((int64_t)b * a.kmax + arg) * 63 + j] : 0.f;
    } else {
      for (int j = 0; j < 4; j++) reg[j] = 0.f;
      for (int j = 0; j < RT; j++) tra[j] = 0.f;
      if (crd) for (int j = 0; j < 63; j++) crd[j] = 0.f;
    }
    const float cx_a = (an[0] + an[2]) / 2.f, cy_a = (an[1] + an[3]) / 2.f;
    if (cx_a >= img_w || cy_a >= img_h) state = -1.f;
    lab[NC] = state; reg[4] = state; tra[RT] = state;
    if (crd) crd[63] = state;
  }
}

void launch_anchor_targets(const AnchorTargetArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(anchor_targets_kernel, dim3(a.B), dim3(AT_THREADS), 0, s, a);
}


// ------------------------------------------------------------------------------------------------------------------
// Training side: forward values of the five losses of `batch_iterate` (reference hmdegopose/loss.py:54-99):
//   classification  focal(alpha 0.25, gamma 1.5) over the non-ignored anchors / max(1, #object anchors)    (102-167)
//   regression      smooth-L1 (sigma 3, `<=` at the knee) over the object anchors / max(1, #)             (170-219)
//   rotation        mean over the object anchors of the mean point distance between the model rotated by the
//                   predicted and by the target axis-angle (x pi): nearest target point when the object is
//                   symmetric; NaN -> 0                                                                    (273-428)
//   translation     torch SmoothL1Loss (beta 1, mean) over the object anchors x 3: NaN without any, as the
//                   reference returns it
//   hand            smooth-L1 (sigma 3) over the object anchors / max(1, #)                                (222-271)
// One workgroup per image (the reference loops over the batch, 54-99); float32 like torch, no contraction; every sum
// is reduced in a fixed order (lane-strided partial sums, then a tree): bit-reproducible.  The object anchors are
// compacted in anchor order with a block scan and their point clouds are rotated one anchor at a time by the
// whole workgroup, the target cloud in LDS.  A second tiny launch averages over the batch (regression x 50).
// ------------------------------------------------------------------------------------------------------------------
#define LOSS_THREADS 1024
#define LOSS_LIST 2048

__device__ __forceinline__ float smooth_l1_sigma3(float d) {
#pragma clang fp contract(off)
  const float s2 = 9.0f;
  d = fabsf(d);
  return d <= 1.0f / s2 ? 0.5f * s2 * (d * d) : d - 0.5f / s2;
}

// block-wide sum in a fixed order; every thread gets the result
__device__ float block_sum(float v, float* red) {
#pragma clang fp contract(off)
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = LOSS_THREADS / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  return red[0];
}

struct AxisAngle { float ax, ay, az, c, s; };
__device__ __forceinline__ AxisAngle axis_angle(const float* r) {
#pragma clang fp contract(off)
  const float pi = 3.14159265358979323846f;
  const float x = r[0] * pi, y = r[1] * pi, z = r[2] * pi;
  const float angle = sqrtf((x * x + y * y) + z * z);
  AxisAngle q; q.ax = x / angle; q.ay = y / angle; q.az = z / angle; q.c = cosf(angle); q.s = sinf(angle);
  return q;
}
// point * cos + cross(axis, point) * sin + axis * dot(axis, point) * (1 - cos)      (loss.py:570-609)
__device__ __forceinline__ void rotate_pt(const AxisAngle& q, const float* p, float o[3]) {
#pragma clang fp contract(off)
  const float dt = (q.ax * p[0] + q.ay * p[1]) + q.az * p[2], omc = 1.0f - q.c;
  const float cx = q.ay * p[2] - q.az * p[1], cy = q.az * p[0] - q.ax * p[2], cz = q.ax * p[1] - q.ay * p[0];
  o[0] = (p[0] * q.c + cx * q.s) + (q.ax * dt) * omc;
  o[1] = (p[1] * q.c + cy * q.s) + (q.ay * dt) * omc;
  o[2] = (p[2] * q.c + cz * q.s) + (q.az * dt) * omc;
}

__global__ __launch_bounds__(LOSS_THREADS) void losses_kernel(LossArgs a) {
#pragma clang fp contract(off)
  __shared__ float red[LOSS_THREADS];
  __shared__ int scan[LOSS_THREADS];
  __shared__ int list[LOSS_LIST];
  __shared__ float tgt[LOSS_MAX_POINTS * 3];
  const int b = blockIdx.x, t = threadIdx.x, N = a.N, K = a.K, R = a.R, H = a.H;
  const float* gc = a.gt_cls + (int64_t)b * N * (K + 1); const float* pc = a.cls + (int64_t)b * N * K;
  const float* gr = a.gt_reg + (int64_t)b * N * 5;       const float* pr = a.reg + (int64_t)b * N * 4;
  const float* gt = a.gt_tr + (int64_t)b * N * (R + 6);  const float* pt = a.tr + (int64_t)b * N * (R + 3);
  const float* gh = a.gt_hand ? a.gt_hand + (int64_t)b * N * (H + 1) : nullptr;
  const float* ph = a.hand ? a.hand + (int64_t)b * N * H : nullptr;

  // ---- pass 1: the per-anchor sums ----
  float s_f = 0.f, s_r = 0.f, s_t = 0.f, s_h = 0.f, n_c = 0.f, n_r = 0.f, n_t = 0.f, n_h = 0.f;
  for (int n = t; n < N; n += LOSS_THREADS) {
    const float st_c = gc[(int64_t)n * (K + 1) + K];
    if (st_c == 1.0f) n_c += 1.f;
    if (st_c != -1.0f)
      for (int k = 0; k < K; k++) {
        const float l = gc[(int64_t)n * (K + 1) + k];
        const float p = fminf(fmaxf(pc[(int64_t)n * K + k], 1e-4f), 1.0f - 1e-4f);
        const float af = l == 1.0f ? 0.25f : 1.0f - 0.25f;
        const float fw = af * powf(l == 1.0f ? 1.0f - p : p, 1.5f);
        const float bce = -(l * logf(p) + (1.0f - l) * logf(1.0f - p));
        if (l != -1.0f) s_f += fw * bce;
      }
    if (gr[(int64_t)n * 5 + 4] == 1.0f) {
      n_r += 1.f;
      for (int k = 0; k < 4; k++) s_r += smooth_l1_sigma3(pr[(int64_t)n * 4 + k] - gr[(int64_t)n * 5 + k]);
    }
    if ((int)rintf(gt[(int64_t)n * (R + 6) + R + 5]) == 1) {
      n_t += 1.f;
      for (int k = 0; k < 3; k++) {
        const float d = fabsf(pt[(int64_t)n * (R + 3) + R + k] - gt[(int64_t)n * (R + 6) + R + k]);
        s_t += d < 1.0f ? 0.5f * d * d : d - 0.5f;
      }
    }
    if (gh && ph && gh[(int64_t)n * (H + 1) + H] == 1.0f) {
      n_h += 1.f;
      for (int k = 0; k < H; k++) s_h += smooth_l1_sigma3(ph[(int64_t)n * H + k] - gh[(int64_t)n * (H + 1) + k]);
    }
  }
  s_f = block_sum(s_f, red); s_r = block_sum(s_r, red); s_t = block_sum(s_t, red); s_h = block_sum(s_h, red);
  n_c = block_sum(n_c, red); n_r = block_sum(n_r, red); n_t = block_sum(n_t, red); n_h = block_sum(n_h, red);

  // ---- pass 2: rotation - the object anchors in ascending order, LOSS_LIST at a time ----
  const int per = (N + LOSS_THREADS - 1) / LOSS_THREADS, n0 = t * per, n1 = min(N, n0 + per);     // a contiguous run per lane
  int mine = 0;
  for (int n = n0; n < n1; n++) mine += (int)rintf(gt[(int64_t)n * (R + 6) + R + 5]) == 1 ? 1 : 0;
  scan[t] = mine;
  __syncthreads();
  for (int o = 1; o < LOSS_THREADS; o <<= 1) {                       // inclusive scan
    const int v = t >= o ? scan[t - o] : 0;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  const int first = scan[t] - mine, total = scan[LOSS_THREADS - 1];
  float rot_sum = 0.f;
  for (int base = 0; base < total; base += LOSS_LIST) {
    __syncthreads();
    int ord = first;
    for (int n = n0; n < n1; n++)
      if ((int)rintf(gt[(int64_t)n * (R + 6) + R + 5]) == 1) { if (ord >= base && ord < base + LOSS_LIST) list[ord - base] = n; ord++; }
    __syncthreads();
    const int cnt = min(LOSS_LIST, total - base);
    for (int i = 0; i < cnt; i++) {
      const int n = list[i];
      const float* g = gt + (int64_t)n * (R + 6);
      const AxisAngle qp = axis_angle(pt + (int64_t)n * (R + 3)), qt = axis_angle(g);
      const bool sym = (int)rintf(g[R + 3]) == 1;
      const int cls = min(max((int)rintf(g[R + 4]), 0), a.classes - 1);
      const float* pts = a.points + (int64_t)cls * a.P * 3;
      __syncthreads();
      for (int j = t; j < a.P; j += LOSS_THREADS) rotate_pt(qt, pts + 3 * j, tgt + 3 * j);
      __syncthreads();
      float dsum = 0.f;
      for (int j = t; j < a.P; j += LOSS_THREADS) {
        float o[3];
        rotate_pt(qp, pts + 3 * j, o);
        float d;
        if (sym) {
          d = INFINITY;
          for (int q = 0; q < a.P; q++) {
            const float dx = o[0] - tgt[3 * q], dy = o[1] - tgt[3 * q + 1], dz = o[2] - tgt[3 * q + 2];
            d = fminf(d, sqrtf((dx * dx + dy * dy) + dz * dz));
          }
        } else {
          const float dx = o[0] - tgt[3 * j], dy = o[1] - tgt[3 * j + 1], dz = o[2] - tgt[3 * j + 2];
          d = sqrtf((dx * dx + dy * dy) + dz * dz);
        }
        dsum += d;
      }
      rot_sum += block_sum(dsum, red) / (float)a.P;
    }
  }
  if (t == 0) {
    float* o = a.per_image + (int64_t)b * 5;
    float rot = rot_sum / n_t;                       // 0 / 0 without object anchors ...
    if (rot != rot) rot = 0.f;                       // ... NaN -> 0 (loss.py:424)
    o[0] = s_f / fmaxf(1.0f, n_c);
    o[1] = s_r / fmaxf(1.0f, n_r);
    o[2] = rot;
    o[3] = s_t / (n_t * 3.0f);                       // NaN without object anchors, like torch's mean of nothing
    o[4] = s_h / fmaxf(1.0f, n_h);
  }
}

__global__ void losses_mean_kernel(LossArgs a) {
#pragma clang fp contract(off)
  const int k = threadIdx.x;
  if (k >= 5) return;
  float s = 0.f;
  for (int b = 0; b < a.B; b++) s += a.per_image[(int64_t)b * 5 + k];
  s /= (float)a.B;
  a.losses[k] = k == 1 ? s * 50.0f : s;
}

void launch_losses(const LossArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(losses_kernel, dim3(a.B), dim3(LOSS_THREADS), 0, s, a);
  hipLaunchKernelGGL(losses_mean_kernel, dim3(1), dim3(64), 0, s, a);
}
