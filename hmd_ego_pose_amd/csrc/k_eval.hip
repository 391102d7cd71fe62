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
