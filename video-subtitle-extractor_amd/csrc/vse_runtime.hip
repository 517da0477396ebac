// C-ABI runtime: context, weight arenas, plan creation and the op dispatch loop (see include/vse_hip.h).
#include <stdlib.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "conv_common.h"

static thread_local std::string g_err;
static void set_err(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}
void vse_set_error(const char* msg) { g_err = msg ? msg : ""; }
#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return VSE_E_HIP;                                                          \
        }                                                                              \
    } while (0)

struct vse_ctx {
    int device;
    std::vector<void*> weights;       // device blobs
    std::vector<size_t> weight_bytes;
    void* zero_page;                  // 4 KiB of zeros (gather target for nothing yet; keeps pointers valid)
};

struct vse_plan {
    vse_ctx* ctx;
    int weights_id;
    std::vector<vse_op> ops;
    size_t ws_bytes;
    int max_ext;
    int n_levels;                     // ragged plans: width levels referenced by the ops (0 = not a ragged plan)
    int batch;                        // images per run (n of the first op's input)
    bool u8_source;                   // an op pre-processes the uint8 frames itself (F_U8SRC): ext[0] = frames, geometry below
    int src_h, src_w;
    long src_pitch, src_fstride;
};

extern "C" {

const char* vse_last_error(void) { return g_err.c_str(); }
size_t vse_sizeof_op(void) { return sizeof(vse_op); }
size_t vse_sizeof_view(void) { return sizeof(vse_view); }
int vse_abi_version(void) { return 2; }
int vse_is_dev_build(void) {
#ifdef VSE_DEV_BUILD
    return 1;
#else
    return 0;
#endif
}

int vse_init(int device_id, vse_ctx** out) {
    if (!out) return VSE_E_INVAL;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device_id < 0 || device_id >= count) {
        set_err("device %d not present (%d visible)", device_id, count);
        return VSE_E_INVAL;
    }
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        set_err("libvse_hip is built for gfx950 (MI355X) only; device %d is %s", device_id, prop.gcnArchName);
        return VSE_E_UNSUPPORTED;
    }
    vse_ctx* c = new vse_ctx();
    c->device = device_id;
    c->zero_page = nullptr;
    if (hipMalloc(&c->zero_page, 4096) != hipSuccess || hipMemset(c->zero_page, 0, 4096) != hipSuccess) {
        set_err("vse_init: cannot allocate the zero page on device %d: %s", device_id, hipGetErrorString(hipGetLastError()));
        if (c->zero_page) (void)hipFree(c->zero_page);
        delete c;
        return VSE_E_HIP;
    }
    *out = c;
    return VSE_OK;
}

void vse_destroy(vse_ctx* c) {
    if (!c) return;
    for (void* w : c->weights)
        if (w) (void)hipFree(w);
    if (c->zero_page) (void)hipFree(c->zero_page);
    delete c;
}

int vse_weights_upload(vse_ctx* c, const void* host_blob, size_t nbytes) {
    if (!c || !host_blob || !nbytes) return VSE_E_INVAL;
    HIP_TRY(hipSetDevice(c->device));
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, nbytes + 256));
    HIP_TRY(hipMemcpy(d, host_blob, nbytes, hipMemcpyHostToDevice));
    c->weights.push_back(d);
    c->weight_bytes.push_back(nbytes);
    return (int)c->weights.size() - 1;
}

int vse_weights_free(vse_ctx* c, int id) {
    if (!c || id < 0 || id >= (int)c->weights.size()) return VSE_E_INVAL;
    if (c->weights[id]) (void)hipFree(c->weights[id]);
    c->weights[id] = nullptr;
    return VSE_OK;
}

int vse_plan_create(vse_ctx* c, int weights_id, const vse_op* ops, int n_ops, size_t ws_bytes, vse_plan** out) {
    if (!c || !ops || n_ops <= 0 || !out) return VSE_E_INVAL;
    if (weights_id < 0 || weights_id >= (int)c->weights.size() || !c->weights[weights_id]) {
        set_err("bad weights id %d", weights_id);
        return VSE_E_INVAL;
    }
    vse_plan* p = new vse_plan();
    p->ctx = c;
    p->weights_id = weights_id;
    p->ops.assign(ops, ops + n_ops);
    p->ws_bytes = ws_bytes;
    p->max_ext = -1;
    p->n_levels = 0;
    p->batch = ops[0].in0.n;
    p->u8_source = false;
    p->src_h = p->src_w = 0;
    p->src_pitch = p->src_fstride = 0;
    const size_t wbytes = c->weight_bytes[weights_id];
    for (int i = 0; i < n_ops; ++i) {
        const vse_op& o = ops[i];
        if (o.kind < OP_CONV || o.kind > OP_CHAIN) {
            set_err("op %d: unknown kind %d", i, o.kind);
            delete p;
            return VSE_E_INVAL;
        }
        if (o.p[P_WLIN] < 0 || o.p[P_WLOUT] < 0) {
            set_err("op %d: negative width level", i);
            delete p;
            return VSE_E_INVAL;
        }
        p->n_levels = std::max(p->n_levels, std::max(o.p[P_WLIN], o.p[P_WLOUT]));
        if (o.kind == OP_CONV && (o.flags & F_U8SRC)) {
            if (!(o.flags & F_STEM) || o.in0.arena != 2) {
                set_err("op %d: F_U8SRC is for a stem conv that reads the plan input", i);
                delete p;
                return VSE_E_INVAL;
            }
            p->u8_source = true;
        }
        const vse_view* vs[5] = {&o.in0, &o.in1, &o.in2, &o.out, &o.out2};
        for (const vse_view* v : vs) {
            if (v->n == 0) continue;
            if (v->arena >= 2) p->max_ext = std::max(p->max_ext, v->arena - 2);
            if (v->arena == 0) {
                const size_t end = (size_t)v->off + ((size_t)v->n * v->h * v->w - 1) * v->ld * v->esize + (size_t)v->c * v->esize;
                if (end > ws_bytes) {
                    set_err("op %d: view exceeds workspace (%zu > %zu)", i, end, ws_bytes);
                    delete p;
                    return VSE_E_INVAL;
                }
            }
        }
        if ((size_t)o.w_off > wbytes || (size_t)o.b_off > wbytes) {
            set_err("op %d: weight offset out of range", i);
            delete p;
            return VSE_E_INVAL;
        }
        if (o.kind == OP_CHAIN) {
            // the chain kernels read their descriptor from the weight blob: check, once, that the record and the blob describe the
            // same chain (header: magic, stages, buffers, LDS image bytes / offset / total; compiler.py emit_chain) and that the
            // descriptor + LDS image lie inside the blob — a stale or foreign blob must fail here, not inside a kernel
            int hdr[16];
            const size_t words = 16 + (size_t)std::max(o.p[4], 0) * 16 + (size_t)std::max(o.p[3], 0) * 28;
            if ((size_t)o.w_off + words * 4 > wbytes || (o.w_off & 3) ||
                hipMemcpy(hdr, reinterpret_cast<const char*>(c->weights[weights_id]) + o.w_off, sizeof hdr, hipMemcpyDeviceToHost) != hipSuccess) {
                set_err("op %d: chain descriptor out of the weight blob", i);
                delete p;
                return VSE_E_INVAL;
            }
            if (hdr[0] != 0x43484e31 || hdr[1] != o.p[3] || hdr[2] != o.p[4] || hdr[5] != o.p[2] || hdr[3] < 0 || hdr[4] < 0 ||
                (size_t)o.w_off + (size_t)hdr[4] + (size_t)hdr[3] > wbytes) {
                set_err("op %d: chain descriptor does not match its record (magic %#x, stages %d / %d, buffers %d / %d, LDS %d / %d)", i, hdr[0],
                        hdr[1], o.p[3], hdr[2], o.p[4], hdr[5], o.p[2]);
                delete p;
                return VSE_E_INVAL;
            }
        }
    }
    *out = p;
    return VSE_OK;
}

void vse_plan_destroy(vse_plan* p) { delete p; }

static inline TView resolve(const vse_view& v, char* ws, char* wts, void* const* ext) {
    TView t;
    t.ptr = nullptr;
    t.n = v.n; t.h = v.h; t.w = v.w; t.c = v.c; t.ld = v.ld; t.esize = v.esize;
    if (v.n == 0) return t;
    char* base = v.arena == 0 ? ws : (v.arena == 1 ? wts : reinterpret_cast<char*>(ext[v.arena - 2]));
    t.ptr = base + v.off;
    return t;
}

// geometry of the uint8 source frames of one run (F_U8SRC plans)
struct SrcGeom { int h, w; long pitch, fstride; };

static int run_op(vse_plan* p, int i, char* ws, void* const* ext, const int32_t* wtab, hipStream_t st, const SrcGeom& src) {
    const vse_op& o = p->ops[i];
    // ragged plans: widths[level][n]
    const int* wl_in = (wtab && o.p[P_WLIN]) ? wtab + (size_t)(o.p[P_WLIN] - 1) * p->batch : nullptr;
    const int* wl_out = (wtab && o.p[P_WLOUT]) ? wtab + (size_t)(o.p[P_WLOUT] - 1) * p->batch : nullptr;
    char* wts = reinterpret_cast<char*>(p->ctx->weights[p->weights_id]);
    const TView in0 = resolve(o.in0, ws, wts, ext), in1 = resolve(o.in1, ws, wts, ext),
                in2 = resolve(o.in2, ws, wts, ext), out = resolve(o.out, ws, wts, ext),
                out2 = resolve(o.out2, ws, wts, ext);
    int rc;
    if (o.kind == OP_CONV) {
        ConvArgs a;
        a.in = in0; a.res = in1; a.out = out;
        a.w = reinterpret_cast<const half_t*>(wts + o.w_off);
        a.bias = reinterpret_cast<const float*>(wts + o.b_off);
        a.zero = reinterpret_cast<const half_t*>(p->ctx->zero_page);
        a.kh = o.p[P_KH]; a.kw = o.p[P_KW]; a.sh = o.p[P_SH]; a.sw = o.p[P_SW]; a.ph = o.p[P_PH]; a.pw = o.p[P_PW];
        a.act = o.p[P_ACT]; a.act2 = o.p[P_ACT2]; a.Np = o.p[P_COUT]; a.Kp = o.p[P_KTOT];
        a.inshift = o.p[P_INSHIFT]; a.resshift = o.p[P_RESSHIFT]; a.cinp = o.p[P_CINP]; a.flags = o.flags;
        a.act_a = o.f[FS_ACT_A]; a.act_b = o.f[FS_ACT_B]; a.post_a = o.f[FS_POST_A]; a.post_b = o.f[FS_POST_B];
        a.dotw = reinterpret_cast<const float*>(wts + o.aux_off);
        a.dotb = o.f[FS_PRE_B]; a.dotact = o.p[P_DOTACT]; a.dot_out = out2;
        a.in2 = in2; a.in2shift = o.p[P_IN2SHIFT];
        if (o.flags & F_IMGW) a.w = reinterpret_cast<const half_t*>(in2.ptr);      // per-image weights in the workspace
        a.wl_out = wl_out;
        a.lo_off = o.p[P_LO_OUT];
        a.res_lo_off = o.p[P_LO_RES];
        a.in_lo_off = o.p[P_LO_IN];
        a.u8src = nullptr; a.u8_h = a.u8_w = 0; a.u8_pitch = a.u8_fstride = 0;
        if (o.flags & F_U8SRC) {
            a.u8src = reinterpret_cast<const uint8_t*>(ext[0]);
            a.u8_h = src.h; a.u8_w = src.w; a.u8_pitch = src.pitch; a.u8_fstride = src.fstride;
        }
        rc = launch_conv(a, st);
    } else if (o.kind == OP_CHAIN) {
        rc = launch_chain(o, in0, out, out2, in2, wts, st);
    } else {
        rc = launch_simple_op(o, in0, in1, in2, out, out2, wts, wl_in, wl_out, st);
    }
    if (rc != VSE_OK) set_err("op %d (kind %d) failed to launch: rc=%d (%s)", i, o.kind, rc, hipGetErrorString(hipGetLastError()));
    return rc;
}

static int check_run(vse_plan* p, void* ws, void* const* ext, int n_ext, const int32_t* d_widths, const char* who) {
    if (!p || !ext || n_ext <= p->max_ext) {
        set_err("%s: need %d external pointers", who, p ? p->max_ext + 1 : 0);
        return VSE_E_INVAL;
    }
    if (!ws && p->ws_bytes) return VSE_E_INVAL;
    if (p->n_levels && !d_widths) {
        set_err("%s: the plan was compiled for ragged batches (%d width levels): run it with a width table", who, p->n_levels);
        return VSE_E_INVAL;
    }
    if (!p->n_levels && d_widths) {
        set_err("%s: width table given to a plan that was not compiled for ragged batches", who);
        return VSE_E_INVAL;
    }
    if (p->u8_source && (p->src_h <= 0 || p->src_w <= 0)) {
        set_err("%s: the plan pre-processes the uint8 frames in its stem: call vse_plan_set_source (or vse_det_forward) first", who);
        return VSE_E_INVAL;
    }
    return VSE_OK;
}

int vse_plan_set_source(vse_plan* p, int src_h, int src_w, int64_t pitch, int64_t frame_stride) {
    if (!p || src_h <= 0 || src_w <= 0 || pitch < (int64_t)src_w * 3 || frame_stride < 0) return VSE_E_INVAL;
    if (!p->u8_source) {
        set_err("vse_plan_set_source: the plan takes a pre-processed fp16 input, not uint8 frames");
        return VSE_E_INVAL;
    }
    p->src_h = src_h; p->src_w = src_w; p->src_pitch = pitch; p->src_fstride = frame_stride;
    return VSE_OK;
}
int vse_plan_takes_frames(vse_plan* p) { return p ? (p->u8_source ? 1 : 0) : VSE_E_INVAL; }

int vse_plan_run(vse_plan* p, void* ws, void* const* ext, int n_ext, void* stream) {
    return vse_plan_run_ragged(p, ws, ext, n_ext, nullptr, stream);
}

// the frame geometry travels with the RUN (a by-value copy taken here), never read from the plan while ops are being launched:
// the same plan may serve callers with different source sizes (vse_det_forward hands over its own arguments)
static int run_all(vse_plan* p, void* ws, void* const* ext, const int32_t* d_widths, void* stream, const SrcGeom src) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int i = 0; i < (int)p->ops.size(); ++i) {
        const int rc = run_op(p, i, reinterpret_cast<char*>(ws), ext, d_widths, st, src);
        if (rc != VSE_OK) return rc;
    }
    return VSE_OK;
}

int vse_plan_run_ragged(vse_plan* p, void* ws, void* const* ext, int n_ext, const int32_t* d_widths, void* stream) {
    int rc = check_run(p, ws, ext, n_ext, d_widths, "vse_plan_run");
    if (rc != VSE_OK) return rc;
    return run_all(p, ws, ext, d_widths, stream, SrcGeom{p->src_h, p->src_w, p->src_pitch, p->src_fstride});
}

int vse_plan_width_levels(vse_plan* p) { return p ? p->n_levels : VSE_E_INVAL; }

// ---- model-level calls: one call per network invocation over a compiled plan ---------------------------------
int vse_det_forward(vse_ctx* c, vse_plan* det_plan, void* ws, const void* d_bgr, int n, int src_h, int src_w, int64_t pitch,
                    int64_t frame_stride, int dst_h, int dst_w, int raw_input, void* d_in_f16, float* d_prob, void* stream) {
    if (!c || !det_plan || !d_bgr || !d_prob) return VSE_E_INVAL;
    if (det_plan->batch != n || det_plan->ops[0].in0.h != dst_h || det_plan->ops[0].in0.w != dst_w) {
        set_err("vse_det_forward: the plan was compiled for %d x %d x %d, called with %d x %d x %d", det_plan->batch,
                det_plan->ops[0].in0.h, det_plan->ops[0].in0.w, n, dst_h, dst_w);
        return VSE_E_INVAL;
    }
    if (det_plan->u8_source) {
        // the plan's stem resizes the frames itself: no pre-processing pass, no fp16 input tensor; the geometry of THIS call's
        // frames goes to the kernels directly (nothing is stored on the shared plan)
        if (src_h <= 0 || src_w <= 0 || pitch < (int64_t)src_w * 3 || frame_stride < 0 || (!ws && det_plan->ws_bytes)) return VSE_E_INVAL;
        if (det_plan->max_ext > 1 || det_plan->n_levels) {
            set_err("vse_det_forward: not a one-map detector plan");
            return VSE_E_INVAL;
        }
        void* ext[2] = {const_cast<void*>(d_bgr), d_prob};
        return run_all(det_plan, ws, ext, nullptr, stream, SrcGeom{src_h, src_w, (long)pitch, (long)frame_stride});
    }
    if (!d_in_f16) return VSE_E_INVAL;
    static const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};      // paddleocr NormalizeImage (DB detectors)
    int rc = vse_det_preprocess(c, d_bgr, n, src_h, src_w, pitch, frame_stride, d_in_f16, dst_h, dst_w, raw_input ? nullptr : mean,
                                raw_input ? nullptr : sd, stream);
    if (rc != VSE_OK) return rc;
    void* ext[2] = {d_in_f16, d_prob};
    return vse_plan_run(det_plan, ws, ext, 2, stream);
}

int vse_rec_forward(vse_ctx* c, vse_plan* rec_plan, void* ws, const void* d_rec_in_f16, const int32_t* d_widths, int out_level,
                    void* d_idx_maxp, int b, int t, int32_t* d_out_idx, int32_t* d_out_len, float* d_out_conf, void* stream) {
    if (!c || !rec_plan || !d_rec_in_f16 || !d_idx_maxp) return VSE_E_INVAL;
    if (rec_plan->batch != b || rec_plan->max_ext != 1) {
        set_err("vse_rec_forward: the plan takes %d crops and %d external buffers (compile it with want_probs=False)", rec_plan->batch,
                rec_plan->max_ext + 1);
        return VSE_E_INVAL;
    }
    if (d_widths && (out_level < 0 || out_level >= rec_plan->n_levels)) return VSE_E_INVAL;
    void* ext[2] = {const_cast<void*>(d_rec_in_f16), d_idx_maxp};
    int rc = vse_plan_run_ragged(rec_plan, ws, ext, 2, d_widths, stream);
    if (rc != VSE_OK) return rc;
    return vse_ctc_collapse_ragged(c, d_idx_maxp, b, t, d_widths ? d_widths + (size_t)out_level * b : nullptr, d_out_idx, d_out_len,
                                   d_out_conf, stream);
}

// ---- a recogniser invocation as ONE HIP graph ---------------------------------------------------------------------------------
// ~80 launches of a recogniser plan + the CTC collapse, captured once against FIXED buffers (the caller keeps input, width table,
// workspace and outputs at the same addresses and refills them) and replayed with a single hipGraphLaunch.
struct vse_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

int vse_rec_graph_create(vse_ctx* c, vse_plan* rec_plan, void* ws, const void* d_rec_in_f16, const int32_t* d_widths, int out_level,
                         void* d_idx_maxp, int b, int t, int32_t* d_out_idx, int32_t* d_out_len, float* d_out_conf, void* stream,
                         vse_graph** out) {
    if (!out) return VSE_E_INVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!st) {
        set_err("vse_rec_graph_create: capture needs a non-default stream");
        return VSE_E_INVAL;
    }
    HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = vse_rec_forward(c, rec_plan, ws, d_rec_in_f16, d_widths, out_level, d_idx_maxp, b, t, d_out_idx, d_out_len, d_out_conf, stream);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != VSE_OK) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess || !g) {
        set_err("hipStreamEndCapture failed: %s", hipGetErrorString(e));
        return VSE_E_HIP;
    }
    vse_graph* vg = new vse_graph{g, nullptr};
    const hipError_t e2 = hipGraphInstantiate(&vg->exec, g, nullptr, nullptr, 0);
    if (e2 != hipSuccess) {
        set_err("hipGraphInstantiate failed: %s", hipGetErrorString(e2));
        (void)hipGraphDestroy(g);
        delete vg;
        return VSE_E_HIP;
    }
    *out = vg;
    return VSE_OK;
}
int vse_graph_launch(vse_graph* g, void* stream) {
    if (!g || !g->exec) return VSE_E_INVAL;
    HIP_TRY(hipGraphLaunch(g->exec, reinterpret_cast<hipStream_t>(stream)));
    return VSE_OK;
}
void vse_graph_destroy(vse_graph* g) {
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

int vse_plan_op_variant(vse_plan* p, int i) {
    if (!p || i < 0 || i >= (int)p->ops.size()) return VSE_E_INVAL;
    const vse_op& o = p->ops[i];
    if (o.kind != OP_CONV) return 0;
    if (o.flags & F_UP2HEAD) return 400000;   // conv_head_up2_kernel
    if (o.flags & F_STEM) return 500000;      // conv_stem_kernel
    if (o.flags & F_DWPRE) {
        const int rs = conv_dwpw_rows_stride(o.p[P_KH], o.p[P_PH], o.p[P_SH], o.p[P_CINP], o.p[P_LO_IN]);
        if (rs) return 860000 + 10 * ((o.p[P_CINP] + 15) / 16) + rs;                       // conv_dwpw_rows_kernel<KS, LO, S>
        return 850000 + 10 * ((o.p[P_CINP] + 15) / 16) + o.p[P_KH];                        // conv_dwpw_kernel<KS, K, LO>
    }
    if (o.flags & F_PW) return ((o.flags & F_TAIL2) ? 810000 : 800000) + (o.p[P_CINP] + 15) / 16;   // conv_pw_kernel<KS> / conv_pw_tail_kernel<KS>
    if ((o.flags & F_COL) && o.p[P_KH] == 3 && o.p[P_KW] == 3) {   // conv_c3_kernel<RW, 8 / RW>
        int rw;
        conv_c3_plan(o.out.h, o.out.w, &rw);
        return 700000 + rw + ((o.p[P_COUT] <= 32 && !(o.flags & F_HLSUM)) ? 50000 : 0);      // + 50000: the 32-cout form conv_c3n32_kernel (F_HLSUM: the 64-row form, hi | lo)
    }
    if (o.flags & F_COL) return 600000 + 100 * o.p[P_KH] + conv_col_bn(o.p[P_COUT]);   // conv_col_kernel<KH, BN>
    if (o.flags & F_PATCH) {   // conv_patch_kernel<TH, BN, BIGP> -> 100000*BIGP + 1000*TH + BN
        // conv_patch_kernel<TH, BN, MODE> -> 100000*MODE + 1000*TH + BN
        int th, bn, mode;
        conv_patch_plan(o.p[P_KH], o.p[P_KW], (o.flags & F_DOT1) ? o.out2.h : o.out.h, o.p[P_COUT], o.flags, &th, &bn, &mode);
        return 100000 * mode + 1000 * th + bn;
    }
    // conv_gemm_kernel configuration c, MASK m -> 200000 + 10*c + m; conv_mfma_kernel<.., UPS> -> 10000*UPS + BN
    static const bool use_gemm = [] { const char* e = vse_dev_getenv("VSE_CONV_GEMM"); return !(e && e[0] == '0'); }();
    {
        long m = (long)o.out.n * o.out.h * o.out.w;
        if (conv_smallk_shape_ok(o.p[P_KH], o.p[P_KW], o.p[P_SH], o.p[P_SW], o.p[P_PH], o.p[P_PW], o.p[P_INSHIFT], o.in0.h == o.out.h && o.in0.w == o.out.w, o.flags,
                                 o.p[P_CINP], m, o.p[P_COUT]))
            return 900000 + ((o.flags & F_WK32) ? 32 : 64) + ((o.flags & F_HILO) ? 1000 : 0);   // conv_smallm_kernel<KT> (small 1x1 problems)
    }
    const int mode = use_gemm ? conv_gemm_mode(o.p[P_KH], o.p[P_KW], o.p[P_SH], o.p[P_SW], o.p[P_PH], o.p[P_PW], o.p[P_CINP],
                                               o.p[P_KTOT], o.p[P_INSHIFT], o.flags) : 0;
    if (mode) {
        long m = (long)o.out.n * o.out.h * o.out.w;
        if (o.flags & F_PIXSHUF) m /= 4;
        if (conv_smallm_shape_ok(mode, m, o.p[P_SH], o.p[P_SW], o.in0.h == o.out.h && o.in0.w == o.out.w, o.flags, o.p[P_CINP]))
            return 900000 + ((o.flags & F_WK32) ? 32 : 64) + ((o.flags & F_HILO) ? 1000 : 0);   // conv_smallm_kernel<KT> (+ 1000: conv_smallm_hl_kernel)
        return 200000 + 10 * conv_gemm_config(o.p[P_COUT], o.p[P_CINP], m) + (mode == 1 ? 1 : 0);
    }
    return (o.p[P_INSHIFT] ? 10000 : 0) + conv_tile_bn(o.p[P_COUT]);
}

// The kernel instantiation op `i` dispatches to, spelled the way rocprofv3 reports it (thread-local storage).
const char* vse_plan_op_kernel_name(vse_plan* p, int i) {
    static thread_local char buf[96];
    buf[0] = 0;
    if (!p || i < 0 || i >= (int)p->ops.size()) return buf;
    const vse_op& o = p->ops[i];
    static const char* simple[] = {"", "", "dwconv_kernel", "pool_kernel", "gap_kernel", "scale_kernel", "binary_kernel", "resize_kernel",
                                   "unary_kernel", "layernorm_kernel", "attn_kernel", "softmax_kernel", "lstm_kernel", "wscale_kernel", "chain_kernel"};
    if (o.kind != OP_CONV) {
        snprintf(buf, sizeof buf, "%s", (o.kind >= 2 && o.kind <= OP_CHAIN) ? simple[o.kind] : "?");
        return buf;
    }
    const int code = vse_plan_op_variant(p, i);
    if (code >= 901000) snprintf(buf, sizeof buf, "conv_smallm_hl_kernel<%d>", code - 901000);
    else if (code >= 900000) snprintf(buf, sizeof buf, "conv_smallm_kernel<%d>", code - 900000);
    else if (code >= 860000) snprintf(buf, sizeof buf, "conv_dwpw_rows_kernel<%d, %s, %d>", (code - 860000) / 10, o.p[P_LO_IN] ? "true" : "false", code % 10);
    else if (code >= 850000) snprintf(buf, sizeof buf, "conv_dwpw_kernel<%d, %d, %s>", (code - 850000) / 10, code % 10, o.p[P_LO_IN] ? "true" : "false");
    else if (code >= 810000) snprintf(buf, sizeof buf, "conv_pw_tail_kernel<%d>", code - 810000);
    else if (code >= 800000) snprintf(buf, sizeof buf, "conv_pw_kernel<%d>", code - 800000);
    else if (code >= 750000) snprintf(buf, sizeof buf, "conv_c3n32_kernel<%d, %d>", code - 750000, 8 / (code - 750000));
    else if (code >= 700000) snprintf(buf, sizeof buf, "conv_c3_kernel<%d, %d>", code - 700000, 8 / (code - 700000));
    else if (code >= 600000) snprintf(buf, sizeof buf, "conv_col_kernel<%d, %d>", (code - 600000) / 100, code % 100);
    else if (code >= 500000) snprintf(buf, sizeof buf, "conv_stem_kernel");
    else if (code >= 400000) {
        // (the launcher's switch, conv_head.hip: the persistent resident-weight form unless VSE_HEAD_RESIDENT=0)
        static const bool resident = [] { const char* e = getenv("VSE_HEAD_RESIDENT"); return !(e && e[0] == '0'); }();
        snprintf(buf, sizeof buf, resident ? "conv_head_up2r_kernel" : "conv_head_up2_kernel");
    }
    else if (o.flags & F_PATCH) snprintf(buf, sizeof buf, "conv_patch_kernel<%d, %d, %d>", (code / 1000) % 100, code % 1000, code / 100000);
    else if (code >= 200000) {
        static const char* cfg[] = {"128, 128, 2, 2, 32, 3", "256, 64, 4, 1, 32, 3", "256, 32, 4, 1, 32, 3", "", "", "", "256, 128, 4, 2, 32, 3",
                                    "", "", "", "", "", "", "", "", "", "256, 256, 4, 4, 32, 3", "256, 192, 8, 2, 32, 3",
                                    "256, 256, 4, 4, 64, 2", "256, 192, 8, 2, 64, 2"};
        const int c = (code - 200000) / 10;
        snprintf(buf, sizeof buf, "conv_gemm_kernel<%s, %d>", (c >= 0 && c < 20) ? cfg[c] : "?", code % 10);
    } else {
        const int bn = code % 1000;
        snprintf(buf, sizeof buf, "conv_mfma_kernel<%s, %s>", bn == 128 ? "128, 128, 2, 2" : (bn == 64 ? "256, 64, 4, 1" : "256, 32, 4, 1"),
                 code >= 10000 ? "true" : "false");
    }
    return buf;
}

int vse_plan_profile(vse_plan* p, void* ws, void* const* ext, int n_ext, const int32_t* d_widths, void* stream, float* ms) {
    if (!ms) return VSE_E_INVAL;
    int rc0 = check_run(p, ws, ext, n_ext, d_widths, "vse_plan_profile");
    if (rc0 != VSE_OK) return rc0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int n = (int)p->ops.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventRecord(ev[0], st));
    for (int i = 0; i < n; ++i) {
        int rc = run_op(p, i, reinterpret_cast<char*>(ws), ext, d_widths, st, SrcGeom{p->src_h, p->src_w, p->src_pitch, p->src_fstride});
        if (rc != VSE_OK) return rc;
        HIP_TRY(hipEventRecord(ev[i + 1], st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < n; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
    for (auto& e : ev) (void)hipEventDestroy(e);
    return VSE_OK;
}

}  // extern "C"
