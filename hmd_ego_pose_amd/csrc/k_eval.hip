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
