/* libvse_hip.so — C ABI of the MI355X-native subtitle-OCR hot path (DB text detection + CTC recognition).
 *
 * The reference (eritpchy/video-subtitle-extractor v2.2.0) has no FFI for this path: its operator
 * interface is a set of Python callables that end in third-party paddleocr / Paddle Inference.  Each entry
 * point below names the reference call site whose work it replaces (file:line under /root/reference).
 * The Python binding a maintainer adds is in INTEGRATION.md (ctypes; it is what vse_amd/engine.py does).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative VSE_E_*
 * code and never throws; vse_last_error() returns a thread-local message.  All device work is enqueued on
 * the hipStream_t passed in (as void*) and is asynchronous with respect to the host unless stated.
 * The caller owns every activation / input / output buffer (e.g. torch allocator); the library owns only
 * the uploaded weights.  One vse_ctx per (process, device); calls on one ctx are not thread-safe.
 */
#ifndef VSE_HIP_H
#define VSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSE_OK 0
#define VSE_E_INVAL (-1)
#define VSE_E_HIP (-2)
#define VSE_E_UNSUPPORTED (-3)
#define VSE_E_NOMEM (-4)

typedef struct vse_ctx vse_ctx;
typedef struct vse_plan vse_plan;

/* A strided NHWC tensor view inside one of the arenas (see ir.py VIEW_DT; 40 bytes, packed). */
#pragma pack(push, 1)
typedef struct vse_view {
    int64_t off;    /* byte offset of element (0,0,0,0) in its arena */
    int32_t arena;  /* 0 workspace, 1 weights, 2+k external pointer k (0 = input, 1.. = outputs) */
    int32_t n, h, w;
    int32_t c;      /* physical channel span */
    int32_t ld;     /* elements between consecutive pixels */
    int32_t esize;  /* 2 = fp16, 4 = fp32/int32 */
    int32_t pad;
} vse_view;

/* One fused kernel launch of the engine program (see ir.py OP_DT). */
typedef struct vse_op {
    int32_t kind;
    int32_t flags;
    int32_t p[22];
    float f[8];
    vse_view in0, in1, in2, out, out2;
    int64_t w_off, b_off, aux_off;
} vse_op;
#pragma pack(pop)

/* ---- context ------------------------------------------------------------------------------------ */
/* Replaces: paddle predictor creation inside paddleocr TextDetector/TextRecognizer.__init__, reached from
 * backend/tools/subtitle_detect.py:22 and backend/tools/ocr.py:91 (PaddleOCR(...)). */
int vse_init(int device_id, vse_ctx** ctx);
void vse_destroy(vse_ctx* ctx);
const char* vse_last_error(void);
size_t vse_sizeof_op(void);
size_t vse_sizeof_view(void);
int vse_abi_version(void);
/* 1 when the library was compiled with -DVSE_DEV_BUILD (experimental kernels + their environment switches), 0 for the product build.
 * The host side consults it before honouring VSE_DEV_BUILD=1 in the environment (vse_amd.engine.load_library): compiler-side experiment
 * switches must never route to kernels a product library refuses.  No reference counterpart (build hygiene, not an operator). */
int vse_is_dev_build(void);

/* ---- network programs ----------------------------------------------------------------------------- */
/* Upload (or replace) the packed weight blob of one model; returns a weights handle id >= 0.
 * Replaces: loading inference.pdiparams (backend/tools/paddle_model_config.py:100-106 + Paddle loader). */
int vse_weights_upload(vse_ctx* ctx, const void* host_blob, size_t nbytes);
int vse_weights_free(vse_ctx* ctx, int weights_id);

/* Create an executable plan from `n_ops` vse_op records compiled for one static input shape. */
int vse_plan_create(vse_ctx* ctx, int weights_id, const vse_op* ops, int n_ops, size_t ws_bytes, vse_plan** plan);
void vse_plan_destroy(vse_plan* plan);

/* Run the network: `ws` is a device workspace of >= ws_bytes, ext[0] the fp16 NHWC(8) input, ext[1..] the
 * output buffers.  The caller zero-fills `ws` ONCE before its first use with a plan (channel-padding lanes that no kernel
 * writes are read against zero weights and must hold finite values) and gives every run that may be in flight at the same
 * time its own workspace; a plan keeps no state between runs.  Replaces the Paddle predictor.run() inside paddleocr predict_det.py / predict_rec.py,
 * i.e. the device work behind backend/tools/subtitle_detect.py:25 and backend/tools/ocr.py:27. */
int vse_plan_run(vse_plan* plan, void* ws, void* const* ext, int n_ext, void* stream);

/* Ragged recogniser batches.  The reference recognises the crops of ONE frame in chunks of rec_batch_num (6,
 * backend/config.py:58 -> backend/tools/ocr.py:99), every chunk zero-padded to its own widest crop; what a crop's logits are
 * depends on that padded width (conv borders, SVTR attention span), not on its neighbours.  A plan compiled for ragged batches
 * takes a batch whose tensor is `Wmax` wide and a device table d_widths[level][n] (int32; level 0 = the padded width sample
 * n would have had in its reference chunk, further levels = that width behind each stride / pooling step, computed by the
 * compiler's Program.width_table): every kernel treats x >= width as outside the image, so sample n receives bit for bit the
 * values a batch of exactly its width yields — crops of many frames share one launch sequence.  vse_plan_run refuses such a plan. */
int vse_plan_run_ragged(vse_plan* plan, void* ws, void* const* ext, int n_ext, const int32_t* d_widths, void* stream);
/* Number of width levels a ragged plan expects in d_widths (0 for an ordinary plan). */
int vse_plan_width_levels(vse_plan* plan);

/* A detector plan compiled with the pre-processing fused into its stem conv (compiler fuse_preprocess: F_U8SRC) takes the uint8
 * BGR frames themselves as ext[0]; their geometry (what vse_det_preprocess gets as arguments) is set here before vse_plan_run /
 * vse_plan_profile, and by vse_det_forward.  Host-side state of the plan: set + run from one thread.  Replaces paddleocr
 * DetResizeForTest + NormalizeImage + ToCHWImage (App. C.1) behind backend/tools/subtitle_detect.py:25, without the pass. */
int vse_plan_set_source(vse_plan* plan, int src_h, int src_w, int64_t pitch, int64_t frame_stride);
/* 1 when the plan takes uint8 frames (above), 0 when it takes the fp16 input of vse_det_preprocess. */
int vse_plan_takes_frames(vse_plan* plan);

/* ---- model-level calls ------------------------------------------------------------------------------------
 * One call per network invocation, composed of the entry points of this header over a compiled plan.
 * vse_det_forward = vse_det_preprocess + vse_plan_run: uint8 BGR frames -> DB probability maps d_prob fp32 [n, dst_h, dst_w]
 * (what paddleocr TextDetector computes before its post-processing, behind backend/tools/subtitle_detect.py:25).  d_in_f16 is
 * caller-owned scratch [n, dst_h, dst_w, 8] fp16 (may be NULL for a plan that takes the frames, vse_plan_takes_frames); raw_input
 * != 0 for a plan compiled with the normalisation in its stem.
 * vse_rec_forward = vse_plan_run(_ragged) + vse_ctc_collapse(_ragged): recogniser input fp16 [b, h, w, 8] (vse_rec_preprocess)
 * -> arg-max / max-probability pairs d_idx_maxp [b, t, 2] and the CTC-collapsed class ids, lengths and mean confidences (what
 * paddleocr TextRecognizer computes behind backend/tools/ocr.py:27).  d_widths / out_level: the ragged plan's width table and
 * the level of its output sequence (NULL / 0 for an ordinary plan). */
int vse_det_forward(vse_ctx* ctx, vse_plan* det_plan, void* ws, const void* d_bgr, int n, int src_h, int src_w, int64_t pitch,
                    int64_t frame_stride, int dst_h, int dst_w, int raw_input, void* d_in_f16, float* d_prob, void* stream);
int vse_rec_forward(vse_ctx* ctx, vse_plan* rec_plan, void* ws, const void* d_rec_in_f16, const int32_t* d_widths, int out_level,
                    void* d_idx_maxp, int b, int t, int32_t* d_out_idx, int32_t* d_out_len, float* d_out_conf, void* stream);

/* vse_rec_forward captured as ONE HIP graph against fixed buffers (input, width table, workspace, outputs stay at these addresses;
 * the caller refills input and width table before every vse_graph_launch on the same stream).  `stream` must not be the default
 * stream.  Replaces ~80 kernel launches per recogniser invocation by one graph launch (backend/tools/ocr.py:27 -> TextRecognizer). */
typedef struct vse_graph vse_graph;
int vse_rec_graph_create(vse_ctx* ctx, vse_plan* rec_plan, void* ws, const void* d_rec_in_f16, const int32_t* d_widths, int out_level,
                         void* d_idx_maxp, int b, int t, int32_t* d_out_idx, int32_t* d_out_len, float* d_out_conf, void* stream,
                         vse_graph** graph);
int vse_graph_launch(vse_graph* graph, void* stream);
void vse_graph_destroy(vse_graph* graph);

/* Per-op timing of one run with HIP events on `stream` (synchronises); ms[n_ops] filled.  d_widths as for
 * vse_plan_run_ragged (NULL for an ordinary plan). */
int vse_plan_profile(vse_plan* plan, void* ws, void* const* ext, int n_ext, const int32_t* d_widths, void* stream, float* ms);

/* Which kernel instantiation op `i` dispatches to, as the name rocprofv3 reports ("conv_c3_kernel<4, 2>",
 * "conv_gemm_kernel<256, 256, 4, 4, 64, 2, 0>", "dwconv_kernel" ...): lets bench.py attribute the time vse_plan_profile
 * measures to the kernels of the committed rocprof summaries.  The string is thread-local and valid until the next call. */
const char* vse_plan_op_kernel_name(vse_plan* plan, int i);
/* The same as an integer key (kernel family x template parameters; 0 for non-conv ops) for per-layer A/B tools
 * (tools/bench_conv.py); the encoding is private to csrc/vse_runtime.hip — use vse_plan_op_kernel_name for anything shown. */
int vse_plan_op_variant(vse_plan* plan, int i);

/* ---- det pre-processing ----------------------------------------------------------------------------- */
/* uint8 BGR frames [n, src_h, src_w, 3] (row pitch `pitch` bytes, frame stride `frame_stride` bytes) ->
 * bilinear resize to [dst_h, dst_w] with OpenCV's fixed-point INTER_LINEAR arithmetic -> (x/255-mean)/std ->
 * fp16 NHWC with 8 physical channels (3 real).  Replaces paddleocr DetResizeForTest + NormalizeImage +
 * ToCHWImage behind backend/tools/subtitle_detect.py:25.
 * mean3 == std3 == NULL: RAW mode — channels 0..2 hold the resized uint8 values themselves (exact in fp16) and channel 3
 * the constant 1; a detector plan compiled with the normalisation folded into its stem conv (compiler input_norm) takes
 * this input and computes on exactly the reference's normalised values instead of their fp16 roundings. */
int vse_det_preprocess(vse_ctx* ctx, const void* d_bgr, int n, int src_h, int src_w, int64_t pitch,
                       int64_t frame_stride, void* d_out_f16, int dst_h, int dst_w, const float* mean3,
                       const float* std3, void* stream);

/* ---- DB post-processing (device part) ----------------------------------------------------------------- */
/* prob map fp32 [n,h,w] -> connected components (8-connectivity) of (prob > thresh) with, per component,
 * pixel count, bounding box and per-row x-extents (the rows' extreme pixels are a superset of the convex
 * hull vertices).  Host finishing (hull, min-area rect, score, unclip: csrc/db_geometry.h) happens inside
 * vse_db_postprocess() below.
 * Replaces cv2.findContours / minAreaRect / fillPoly+mean inside paddleocr DBPostProcess (App. C.2). */
typedef struct vse_db_params {
    double box_thresh;     /* 0.6: compared with the box score as Python floats (doubles) in the reference */
    double unclip_ratio;   /* 1.5: distance = area * unclip_ratio / perimeter is double arithmetic in the reference */
    float thresh;          /* 0.3: compared with the float32 map in float32 */
    int max_candidates;    /* 1000 */
    int min_size;          /* 3 */
} vse_db_params;

typedef struct vse_box {
    float pts[4][2];   /* tl, tr, br, bl in source-frame pixels */
    float score;
    int frame;
} vse_box;

/* Workspace size (bytes) for n maps of h x w. */
size_t vse_db_workspace_bytes(int n, int h, int w);
/* Full DB post-process for a batch of maps: device CCL + device scoring + host geometry.
 * boxes[max_boxes] is host memory; *n_boxes receives the count.  Synchronises `stream`.
 * src_h/src_w: original frame size the boxes are scaled to. */
int vse_db_postprocess(vse_ctx* ctx, const float* d_prob, int n, int h, int w, int src_h, int src_w,
                       const vse_db_params* prm, void* d_ws, size_t ws_bytes, vse_box* boxes, int max_boxes,
                       int* n_boxes, void* stream);

/* ---- rec pre-processing ------------------------------------------------------------------------------- */
/* One perspective crop per box from the ORIGINAL uint8 BGR frames (bicubic, replicate border, 90-degree
 * rotation when h/w >= 1.5), then bilinear resize to height `rec_h`, (x/255-0.5)/0.5, zero right-pad to
 * `rec_w`, written as fp16 NHWC(8) rows of a [n_boxes, rec_h, rec_w, 8] batch.
 * Replaces paddleocr get_rotate_crop_image + resize_norm_img (App. C.4-C.5) behind backend/tools/ocr.py:27. */
typedef struct vse_crop {
    float quad[4][2];   /* source quad (tl,tr,br,bl) in frame pixels */
    int frame;          /* which frame of the batch */
    int crop_w, crop_h; /* integer size of the rectified crop before the 48-high resize */
    int resized_w;      /* width after resize to rec_h (<= rec_w) */
    int rotate;         /* 1: rotate 90 (np.rot90) before resize */
} vse_crop;

/* `crops` is a HOST array (the box list comes from the host-side DB geometry); the library solves the four
 * point homographies in double and uploads them.  d_scratch holds the rectified uint8 crops. */
int vse_rec_preprocess(vse_ctx* ctx, const void* d_bgr, int n_frames, int src_h, int src_w, int64_t pitch,
                       int64_t frame_stride, const vse_crop* crops, int n_crops, void* d_out_f16, int rec_h,
                       int rec_w, void* d_scratch, size_t scratch_bytes, void* stream);
size_t vse_rec_preprocess_scratch_bytes(int n_crops, int max_crop_w, int max_crop_h);

/* ---- CTC greedy collapse --------------------------------------------------------------------------------- */
/* idx_maxp: int32/fp32 pairs [b, t, 2] from the softmax head.  Keeps t where idx[t] != idx[t-1] and idx != 0
 * (wavefront ballot scan), writes kept class ids compacted per row, their count and the mean kept prob.
 * Replaces paddleocr CTCLabelDecode (App. C.6) behind backend/tools/ocr.py:27. */
int vse_ctc_collapse(vse_ctx* ctx, const void* d_idx_maxp, int b, int t, int32_t* d_out_idx, int32_t* d_out_len,
                     float* d_out_conf, void* stream);
/* The same over a ragged batch: d_tlen[b] (device int32) = sequence length of each row (the last level's row of the width
 * table handed to vse_plan_run_ragged); time steps at or behind it are not decoded.  d_tlen == NULL = vse_ctc_collapse. */
int vse_ctc_collapse_ragged(vse_ctx* ctx, const void* d_idx_maxp, int b, int t, const int32_t* d_tlen, int32_t* d_out_idx,
                            int32_t* d_out_len, float* d_out_conf, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VSE_HIP_H */
