/* hep.h - C ABI of libhep.so, the MI355X-native (gfx950) EfficientPose / HMD-EgoPose
 * inference path: EfficientNet-B{phi} MBConv backbone -> BiFPN -> box / class / rotation /
 * translation / hand heads -> anchor, box and translation decode -> detection filter.
 *
 * Drop-in boundary.  Every entry point names the reference interface it replaces
 * (paths relative to the reference repository root):
 *
 *   hep_create / hep_destroy   <- `new InferenceSession(ModelFilePath, sessionOptions)` / Dispose
 *                                 unity-sandbox/OpenCVDNNSandboxNetCore/Program.cs:39-61,
 *                                 unity-sandbox/WebRTCNetCoreSandbox/Program.cs:57-78;
 *                                 Python: HMDEgoPose(...)+load_state_dict, pytorch-sandbox/evaluate.py:84-119
 *   hep_run                    <- `Session.Run({"input": float[1,3,S,S]})` -> 10 outputs in the order fixed by
 *                                 pytorch-sandbox/hmdegopose/misc_utils.py:77-83 (feat1..5, regression,
 *                                 classification, rotation, translation_raw, hand); Program.cs:95-122 / :211-229
 *   hep_run_device             <- HMDEgoPose.forward, pytorch-sandbox/backbone.py:104-125 (device tensors,
 *                                 asynchronous on the caller's HIP stream; what the torch custom op calls)
 *   hep_anchors                <- anchors_for_shape, pytorch-sandbox/generators/utils/anchors.py:273-318 and the
 *                                 files anchors_256.txt / translation_anchors_256.txt the C# hosts load
 *                                 (Program.cs:26-28)
 *   hep_decode / _device       <- format_bboxes + format_translation, pytorch-sandbox/hmdegopose/loss.py:12-51
 *                                 (C# twins Program.cs:225-470)
 *   hep_filter / _device       <- FilterDetections / filter_detections, pytorch-sandbox/hmdegopose/layers.py:264-482
 *                                 (C# twin Program.cs:472-627)
 *   hep_anchor_targets_device  <- anchor_targets_bbox, pytorch-sandbox/generators/utils/anchors.py:69-221 (training side)
 *   hep_losses_device          <- batch_iterate, pytorch-sandbox/hmdegopose/loss.py:54-428 (training side, forward values)
 *   hep_pose_errors / _device  <- check_6d_pose_add / check_6d_pose_add_s, pytorch-sandbox/eval/common.py:682-746 with
 *                                 c_min_distances, pytorch-sandbox/generators/utils/calc_min_distances.h:24-35 (the
 *                                 metric arithmetic of evaluate.py's loop, eval/common.py:866-1121)
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a
 * negative hep_status, never throws (every entry point catches what its C++ body could raise -> HEP_ERR_INTERNAL) and never aborts; hep_last_error() gives a thread-local
 * message.  A handle serialises its own calls with an internal mutex (the C# frame callback
 * re-enters Run from WebRTC worker threads, Program.cs:128); several handles may coexist.
 * There is NO CPU fallback: without a usable gfx950 device hep_create fails.
 */
#ifndef HEP_H_
#define HEP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HEP_ABI_VERSION 1

typedef struct hep_handle hep_handle;

typedef enum hep_status {
  HEP_OK = 0,
  HEP_ERR_INVALID = -1,      /* bad argument                                  */
  HEP_ERR_PACK = -2,         /* weight pack missing / malformed / wrong shape  */
  HEP_ERR_DEVICE = -3,       /* HIP error or no gfx950 device                  */
  HEP_ERR_UNSUPPORTED = -4,  /* phi / size / batch outside the built range     */
  HEP_ERR_INTERNAL = -5      /* a C++ exception (out of host memory, ...) was caught at the ABI: nothing ever propagates into the caller */
} hep_status;

typedef enum hep_dtype {
  HEP_F32 = 0,   /* fp32 storage, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): the parity mode        */
  HEP_BF16 = 1,  /* bf16 activations+weights, fp32 accumulate (v_mfma_f32_16x16x32_bf16): the bench */
  HEP_FP8 = 2    /* bf16 session whose backbone pointwise convs (expand / project, 62 % of the MACs) run with OCP e4m3
                    operands (v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulate): weights stored as e4m3 with one scale per
                    output channel behind the BN fold, activations converted on the fly with one power-of-two scale per
                    tensor calibrated at hep_create; depthwise, BiFPN and heads stay bf16 (BASELINE config 5).
                    OPT-IN BUILD (make -C hmd_ego_pose_amd/csrc fp8 -> libhep_fp8.so): on MI355X it measured slower than
                    HEP_BF16 at every batch size and ten times less accurate, so the default libhep.so answers
                    HEP_ERR_UNSUPPORTED to it (and says so in hep_last_error)                                       */
} hep_dtype;

/* hep_create flags */
#define HEP_FLAG_KEEP_INTERMEDIATES 1u   /* no activation-buffer reuse: hep_debug_tensor works for every stage */
#define HEP_FLAG_NO_GRAPH 2u             /* launch kernels eagerly instead of replaying a captured hipGraph    */

/* Output indices (the ONNX export order, misc_utils.py:77-83). */
enum { HEP_OUT_FEAT1 = 0, HEP_OUT_FEAT5 = 4, HEP_OUT_REGRESSION = 5, HEP_OUT_CLASSIFICATION = 6,
       HEP_OUT_ROTATION = 7, HEP_OUT_TRANSLATION_RAW = 8, HEP_OUT_HAND = 9, HEP_NUM_OUTPUTS = 10 };

int hep_abi_version(void);
/* Which build of the library this is: "libhep gfx950" followed by the optional parts it was compiled with - " alt" (the rejected plan
 * alternatives and their knobs: make alt), " fp8" (make fp8), " poison" (sanitizer build), " trace" (phase stamps).  Static storage. */
const char* hep_build_info(void);
const char* hep_last_error(void);

/* Number of HIP devices visible; does not initialise the runtime beyond counting. */
int hep_device_count(void);

/* Build a session: reads a HEPW weight pack (the reference state_dict by name, fp32;
 * see hmd_ego_pose_amd/weights.py), folds BatchNorm, lays weights out for the kernels
 * and allocates the activation arena for batches up to max_batch on HIP device `device`. */
int hep_create(const char* pack_path, int phi, int size, int max_batch, int dtype, int device,
               unsigned flags, hep_handle** out);
int hep_create_from_memory(const void* pack, size_t pack_bytes, int phi, int size, int max_batch,
                           int dtype, int device, unsigned flags, hep_handle** out);
void hep_destroy(hep_handle* h);

/* Geometry. */
int hep_num_anchors(const hep_handle* h);                       /* N = 9 * sum(level cells)          */
/* Classes of the classifier, read from the weights: its header holds 9 * num_classes channels (efficientdet/model.py:393;
 * backbone.py:14 `num_classes`, 1 at every reference call site).  1..63. */
int hep_num_classes(const hep_handle* h);
int hep_output_shape(const hep_handle* h, int index, int batch, int64_t dims[4], int* ndim);
/* The handle's OWN device buffer of head output `index` (HEP_OUT_REGRESSION .. HEP_OUT_HAND), fp32 [max_batch, N, K]:
 * where hep_run_device leaves its results when outs == NULL.  Valid until hep_destroy; rewritten by the next forward on
 * this handle (stream-ordered).  Lets a serving loop with several handles in flight hand out results without the
 * device-to-device copies that caller-provided outs[] cost (75 MB per batch of 16 at phi 0). */
int hep_output_device(const hep_handle* h, int index, float** ptr);

/* Session.Run replacement: host buffers in, host buffers out, synchronous.
 * input: fp32 NCHW [batch,3,S,S], already normalised.  feats may be NULL (or hold NULLs):
 * feature maps are then not exported.  Outputs are caller-allocated, fp32:
 * feats[l] NCHW [batch,W,s_l,s_l]; regression [batch,N,4]; classification [batch,N,num_classes] (post-sigmoid);
 * rotation [batch,N,3]; translation_raw [batch,N,3]; hand [batch,N,63]. */
int hep_run(hep_handle* h, const float* input_nchw, int batch, float* const feats[5],
            float* regression, float* classification, float* rotation, float* translation_raw, float* hand);

/* HMDEgoPose.forward on device memory, asynchronous on `stream` (a hipStream_t; NULL = default).
 * in_strides: element strides of (n,c,h,w) - an NHWC-memory view (eval/common.py:397) is accepted
 * as is; NULL = contiguous NCHW.  outs[5] = regression, classification, rotation, translation_raw,
 * hand (device, fp32; NULL = leave them in the handle's own buffers, read back through
 * hep_decode_device / hep_filter_device with NULL inputs).  feats may be NULL.  Every batch size replays
 * its own hipGraph.
 * A handle owns ONE activation arena: work enqueued through it is stream-ordered, so use a handle on one
 * stream at a time (or let the previous forward finish before switching streams).  For several batches in
 * flight create several handles from the same weight pack - see INTEGRATION.md section 4. */
int hep_run_device(hep_handle* h, const float* input, const int64_t in_strides[4], int batch,
                   float* const outs[5], float* const feats[5], void* stream);

/* anchors_for_shape((size,size)): host, float64 arithmetic, one cast to float32.
 * anchors [N,4] x1,y1,x2,y2; translation_anchors [N,3] cx,cy,stride.  Either may be NULL.
 * Returns N (or a negative status). */
int hep_anchors(int size, float* anchors, float* translation_anchors);

/* Box + translation decode.  camera [batch,6] = fx,fy,px,py,tz_scale,image_scale.
 * boxes [batch,N,4] = xmin,ymin,xmax,ymax clipped to [0,S-1]; translation [batch,N,3] = Tx,Ty,Tz. */
int hep_decode(hep_handle* h, const float* regression, const float* translation_raw, const float* camera,
               int batch, float* boxes, float* translation);
int hep_decode_device(hep_handle* h, const float* regression, const float* translation_raw, const float* camera,
                      int batch, float* boxes, float* translation, void* stream);

/* Detection filter per image: score > score_threshold -> greedy NMS (IoU strictly greater than
 * nms_threshold suppresses; candidates by descending score, ties by lower anchor index) -> first
 * max_detections survivors -> rows padded with -1.  classification is [batch,N,num_classes]; with more than one class
 * the filter is class-specific (layers.py:347-358, the reference's default and the only mode it constructs,
 * train.py:78-81): every class is thresholded and suppressed on its own, the (anchor, class) pairs of all classes are
 * concatenated class by class and the max_detections best scores are kept (ties: the earlier pair); det_labels holds
 * the class.  Unlike the reference (which returns only the
 * last batch item, layers.py:466-482) every image gets its rows.
 * det_boxes [batch,M,4], det_scores [batch,M], det_labels [batch,M] (int32), det_rotation [batch,M,3],
 * det_translation [batch,M,3], det_hand [batch,M,63], det_index [batch,M] (int32 anchor index),
 * det_count [batch] (int32).  Any output except det_count may be NULL. */
/* class_specific_filter of FilterDetections (layers.py:403-433, filter_detections :347-362).  on != 0 (the default, and the only mode
 * the reference constructs): as described above.  on == 0: ONE pass per image over every anchor's best class - score = max over the
 * class columns, det_labels = the first argmax - thresholded, suppressed and cut to max_detections like a single class.  With one
 * class both modes are the same pass.  Applies to the following hep_filter / hep_filter_device calls on this handle. */
int hep_set_class_specific_filter(hep_handle* h, int on);
int hep_filter(hep_handle* h, const float* boxes, const float* classification, const float* rotation,
               const float* translation, const float* hand, int batch, float score_threshold,
               float nms_threshold, int max_detections, float* det_boxes, float* det_scores,
               int32_t* det_labels, float* det_rotation, float* det_translation, float* det_hand,
               int32_t* det_index, int32_t* det_count);
int hep_filter_device(hep_handle* h, const float* boxes, const float* classification, const float* rotation,
                      const float* translation, const float* hand, int batch, float score_threshold,
                      float nms_threshold, int max_detections, float* det_boxes, float* det_scores,
                      int32_t* det_labels, float* det_rotation, float* det_translation, float* det_hand,
                      int32_t* det_index, int32_t* det_count, void* stream);

/* ADD and ADD-S of num_pairs (ground truth, prediction) poses over one object model: points [num_points,3]; rotations
 * as axis-angle vectors in radians (what _get_detections hands on: network output * pi), translations in the model's
 * unit.  add[i] = mean over ALL points of ||(R_gt p + t_gt) - (R_pr p + t_pr)|| (float64); add_s[i] = mean over the
 * ground-truth cloud subsampled with step = num_points / max_points + 1 (max_points = 1000 in the reference) of the
 * float32 distance to the nearest point of the equally subsampled predicted cloud.  A pose counts as correct when the
 * value is <= 0.1 * diameter (the caller's comparison).  The host variant copies through a temporary device buffer. */
int hep_pose_errors(int device, const float* points, int num_points, const float* rvec_gt, const float* t_gt,
                    const float* rvec_pred, const float* t_pred, int num_pairs, int max_points, double* add, double* add_s);
int hep_pose_errors_device(const float* points, int num_points, const float* rvec_gt, const float* t_gt,
                           const float* rvec_pred, const float* t_pred, int num_pairs, int max_points, double* add,
                           double* add_s, void* stream);

/* Training side: anchor_targets_bbox, pytorch-sandbox/generators/utils/anchors.py:69-221 with the IoU
 * matrix of generators/utils/compute_overlap.pyx:33-73 and bbox_transform anchors.py:422-458, on device memory: per
 * image the ground-truth boxes [batch][kmax][4] (float64 x1,y1,x2,y2; the first num_gt[b] are valid), their labels,
 * transformation targets [batch][kmax][num_transform] and (optionally) hand coordinates [batch][kmax][63]; image_hw
 * [batch][2] = (height, width) of the unpadded image.  Outputs as the reference builds them (float32): labels
 * [batch][N][num_classes+1], regression [batch][N][5] = (ty,tx,th,tw,state), transformation [batch][N][num_transform+1],
 * coords [batch][N][64] (may be NULL); the last column is the anchor state: -1 ignore, 0 background, 1 object.
 * They feed hep_losses_device below. */
int hep_anchor_targets_device(const float* anchors, int num_anchors, const double* gt_boxes, const int32_t* gt_labels,
                              const float* gt_transform, const float* gt_coords, const int32_t* num_gt, const int32_t* image_hw,
                              int batch, int kmax, int num_classes, int num_transform, double negative_overlap, double positive_overlap,
                              float* labels, float* regression, float* transformation, float* coords, void* stream);

/* batch_iterate, pytorch-sandbox/hmdegopose/loss.py:54-99 - forward values of the five training losses on device
 * memory (float32, as torch computes them): focal classification loss (102-167), smooth-L1 box regression x 50
 * (170-219), rotation = mean (nearest, for symmetric objects) model-point distance between the predicted and the target
 * axis-angle rotation (273-428), translation = torch SmoothL1Loss on the object anchors (NaN when an image has none - the
 * reference's behaviour), smooth-L1 hand loss (222-271; gt_hand / hand may both be NULL).  Layouts as the generator and
 * the network produce them: gt_classification [batch][N][num_classes+1], classification [batch][N][num_classes]
 * (post-sigmoid), gt_regression [batch][N][5], regression [batch][N][4], gt_transformation [batch][N][num_rotation+3+3] =
 * (rotation, translation, is_symmetric, class index, anchor state), transformation [batch][N][num_rotation+3] =
 * cat(rotation head, decoded translation) (train.py:49), gt_hand [batch][N][num_hand+1], hand [batch][N][num_hand],
 * model_points [num_model_classes][num_points][3] (num_points <= 2048).  Outputs: per_image [batch][5] and losses [5] =
 * (classification, regression, rotation, translation, hand), the batch means.  Backward stays with the caller's
 * autograd (the reference's optimiser loop, train.py:88-342, is out of scope). */
int hep_losses_device(const float* gt_classification, const float* classification, const float* gt_regression, const float* regression,
                      const float* gt_transformation, const float* transformation, const float* gt_hand, const float* hand,
                      const float* model_points, int batch, int num_anchors, int num_classes, int num_rotation, int num_hand,
                      int num_model_classes, int num_points, float* per_image, float* losses, void* stream);

/* preprocess_image (reference generators/colibri_common.py:622-656): device uint8 RGB [batch, height, width, 3] ->
 * device float32 [batch, size, size, 3]: resize by scale = size / max(height, width) (8-bit bilinear, OpenCV
 * INTER_LINEAR fixed-point convention - restated, parity unpinned: cv2 is absent here; skipped when scale == 1, every
 * 256x256 syn_colibri frame), then ((x / 255) - mean) / std with numpy's float64 intermediate steps (the no-resize
 * output is bit-identical to the numpy code), zero-padded at the bottom / right.  Hand the result to hep_run_device as
 * the NCHW view of NHWC memory (strides {size*size*3, 1, size*3, 3}), exactly what eval/common.py:397 does; the camera
 * vector's image_scale entry is size / max(height, width). */
int hep_preprocess_u8_device(hep_handle* h, const uint8_t* rgb_hwc, int batch, int height, int width,
                             float* out_hwc, void* stream);

/* The frame callback of the reference's streaming app (unity-sandbox/WebRTCNetCoreSandbox/Program.cs:140-205, 381-445) on
 * the device: 4:2:0 planar frames [batch][height * width * 3 / 2] bytes as WebRTC delivers them -> cvtColor(YUV2BGR_YV12)
 * (the app reads its I420 bytes as YV12: U and V exchanged, kept) -> centre crop `crop` x `crop` -> resize to `resized` x
 * `resized` -> ResizeAndNormalizeMat to the session's size (float32 / 255, - mean, / std applied to B, G, R in that order as
 * the C# Scalars meet a BGR Mat, zero-padded) -> out_hwc float32 [batch, size, size, 3] in B, G, R channel order, the
 * NHWC memory of blobFromImage(swapRB = false)'s NCHW result (hand it to hep_run_device with strides
 * {size*size*3, 1, size*3, 3}).  The app uses crop 256, resized 512.  OpenCV's BT.601 fixed-point conversion and 8-bit
 * INTER_LINEAR are restated from its source: parity unpinned (cv2 is absent from the build image).
 * Not for stream capture (the handle orders its scratch frames across streams with an event of its own): HEP_ERR_UNSUPPORTED
 * when `stream` is capturing. */
int hep_preprocess_i420_device(hep_handle* h, const uint8_t* yuv, int batch, int height, int width, int crop, int resized,
                               float* out_hwc, void* stream);

/* Introspection used by tests, bench.py and DESIGN.md tables. */
int hep_debug_tensor_count(const hep_handle* h);
int hep_debug_tensor_info(const hep_handle* h, int i, const char** name, int64_t dims[4] /* B,H,W,C */);
/* Copy stage tensor `name` of the last run (as fp32, NHWC) into host memory. */
int hep_debug_tensor(hep_handle* h, const char* name, int batch, float* out, size_t capacity_floats);
int hep_kernel_count(const hep_handle* h, int batch);      /* launches in one forward                      */
/* Per-launch description of the forward plan: name, algorithmic bytes and flops for `batch`. */
int hep_kernel_info(const hep_handle* h, int batch, int i, const char** name, double* bytes, double* flops);
/* fp8 sessions: the calibrated per-tensor activation scale of launch i (0 when the launch has no e4m3 operands). */
int hep_fp8_scale(const hep_handle* h, int i, float* a_scale);
/* HEP_FP8 sessions fix one power-of-two scale per quantised GEMM input at hep_create, from the amax of that tensor on two
 * frames of pseudo-normal noise with 2x headroom; the e4m3 conversion SATURATES silently at +-448 * scale.  Real frames
 * (normalised to about [-2.1, 2.6], structured, zero-padded) through trained weights can exceed that range in the deeper
 * layers: recalibrate on representative frames before serving.  frames: contiguous fp32 [batch,3,S,S] on the session's
 * device (the first min(batch, frames per lane) frames are used: all of max_batch with the default single lane, max_batch / HEP_LANES otherwise).  Not to be called while a run is in flight on this handle.
 * Replaces the fixed scales of ONNXRuntime's static quantisation tables, which the reference does not use (fp32 ORT). */
int hep_calibrate_fp8(hep_handle* h, const float* frames_nchw_device, int batch);
/* Device function (as rocprofv3 --kernel-trace names it, e.g. "sep_kernel<true>") behind launch i. */
int hep_kernel_symbol(const hep_handle* h, int i, const char** symbol);
/* Time `iters` replays of the forward at `batch` with HIP events on the handle's own stream; when
 * per_kernel_ms is non-NULL (length hep_kernel_count) also run the forward eagerly with a HIP event
 * in front of every launch and return each launch's average in-sequence duration. */
int hep_profile(hep_handle* h, int batch, int iters, float* total_ms_per_iter, float* per_kernel_ms);
/* Throughput cost of every launch: launch i is issued `iters` times on each of `nstreams` HIP streams
 * at once and per_kernel_ms[i] = wall time / (iters * nstreams).  A launch that fills the chip costs
 * its full duration, one that leaves CUs idle costs less than it takes alone (bench.py keeps several
 * batches in flight, so this - not the stand-alone duration - is what a launch costs the pipeline). */
int hep_profile_concurrent(hep_handle* h, int batch, int iters, int nstreams, float* per_kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* HEP_H_ */
