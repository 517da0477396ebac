"""Engine IR shared by the Python graph compiler and the C-ABI runtime (csrc/vse_runtime.hip).

One `vse_op` record = one HIP kernel launch.  The record layout here MUST match `struct vse_op` in
include/vse_hip.h (checked by tests/test_abi.py through vse_sizeof_op()).
"""
import numpy as np

KT = 64           # K tile of the packed conv weights: [Kp/KT][Np][KT] fp16, Kp a multiple of KT

# op kinds ---------------------------------------------------------------------------------------
OP_CONV = 1       # implicit-GEMM MFMA conv (also linear / 2x2 s2 transposed conv via pixel-shuffle store)
OP_DWCONV = 2     # depthwise kxk conv
OP_POOL = 3       # max / avg pooling window
OP_GAP = 4        # global average pool -> [N,1,1,C]
OP_SCALE = 5      # out = x * s[n,c]   (SE gate);  flag: + x (residual SE)
OP_BINARY = 6     # out = act(x (+|*) up(y))
OP_RESIZE = 7     # out = nearest-upsample(x)  (copy when shift == 0); writes into concat slices
OP_UNARY = 8      # out = act(x*a+b)*a2+b2
OP_LAYERNORM = 9  # over channels
OP_ATTN = 10      # fused multi-head self-attention over T (SVTR mixer)
OP_SOFTMAX = 11   # class softmax -> fp32 probs (optional) + argmax + max prob
OP_LSTM = 12      # one direction of one LSTM layer (recurrent part; input projection is an OP_CONV)
OP_WSCALE = 13    # per-image conv weights: out[n][e] = fp16(W[e] * gate[n][k(e)]) over the tiled 1x1 weight blob at w_off
                  # ([Kp/kt][Np][kt], kt = p[2]); in0 = SE gate [N,1,1,C], out = [N,1,1,Kp*Np] in the workspace; p[0] = Kp, p[1] = Np

ACT_NONE, ACT_RELU, ACT_HSWISH, ACT_SWISH, ACT_SIGMOID, ACT_HSIGMOID = 0, 1, 2, 3, 4, 5

# arenas -----------------------------------------------------------------------------------------
ARENA_WS = 0       # activation workspace (caller-owned)
ARENA_W = 1        # weights (library-owned, uploaded once)
ARENA_EXT0 = 2     # external pointers passed to vse_plan_run(): 2 = input, 3.. = outputs

VIEW_DT = np.dtype([
    ("off", "<i8"),      # byte offset of element (n=0,h=0,w=0,c=0) inside the arena
    ("arena", "<i4"),
    ("n", "<i4"), ("h", "<i4"), ("w", "<i4"),
    ("c", "<i4"),        # physical channel span of the view (multiple of 8 unless ld == 1)
    ("ld", "<i4"),       # elements between consecutive pixels
    ("esize", "<i4"),    # 2 = f16, 4 = f32/i32
    ("pad", "<i4"),
], align=False)

OP_DT = np.dtype([
    ("kind", "<i4"),
    ("flags", "<i4"),
    ("p", "<i4", (22,)),
    ("f", "<f4", (8,)),
    ("in0", VIEW_DT), ("in1", VIEW_DT), ("in2", VIEW_DT), ("out", VIEW_DT), ("out2", VIEW_DT),
    ("w_off", "<i8"), ("b_off", "<i8"), ("aux_off", "<i8"),
], align=False)

# p[] slots for OP_CONV
P_KH, P_KW, P_SH, P_SW, P_PH, P_PW = 0, 1, 2, 3, 4, 5
P_ACT, P_ACT2 = 6, 7
P_COUT = 8          # GEMM N (physical out channels incl. padding; x4 for pixel-shuffle)
P_KTOT = 9          # padded K (multiple of KT)
P_INSHIFT = 10      # nearest-upsample shift applied when gathering the input
P_RESSHIFT = 11     # nearest-upsample shift applied when reading the residual
P_CINP = 12         # physical input channels (multiple of 8)
P_LO_IN = 17        # F_DWPRE: != 0: the input (in0) is an fp16 hi + lo pair
P_LO_RES = 16       # != 0: the residual (in1) is an fp16 hi + lo pair: both halves are added
P_LO_OUT = 15       # != 0: the output is an fp16 hi + lo PAIR: fp16(v - fp16(v)) is stored P_LO_OUT channels behind the hi value
# flags for OP_CONV
F_RES = 1           # has residual (in1)
F_PIXSHUF = 2       # 2x2 stride-2 transposed conv: N = 4*Coutp ordered (dy,dx,co)
F_OUT_F32 = 4       # store fp32
F_PATCH = 8         # LDS-resident-patch conv kernel; weights packed [chunk][tap][Np][32]
F_DOT1 = 16         # epilogue ends in a fused 1x1 conv to ONE channel: out2 = act(sum_c y[c]*dotw[c] + f[FS_PRE_B]); y is not stored
P_DOTACT = 13       # activation of the fused 1-channel projection
F_SRC2 = 32         # in2 is a second input source: channels [in0.c, in0.c+in2.c) of a virtual concat, own shift p[P_IN2SHIFT]
P_IN2SHIFT = 14
F_HILO = 1024       # weights stored as fp16 hi + fp16 lo (w = hi + lo to ~22 bits): the kernel walks K twice over the same
                    # activations, the lo tiles follow the hi tiles in the blob; OP_DWCONV: the filter table is fp32 [taps][C] either way
                    # (fp16 hi + fp16 lo summed by the compiler: the depthwise kernels multiply-add fp16 activations with fp32 weights)
F_STEM = 512        # 3x3 conv over an image-like input (<= 4 real channels): conv_stem_kernel, weights packed [Np][10 taps][8]
F_GATE = 256        # OP_DWCONV: in1 = SE gate [N,1,1,C]; the input is multiplied by it (rounded to fp16) on load — the
                    # separate OP_SCALE pass of an SE block whose only consumer is this depthwise conv disappears
F_WK32 = 128        # weights tiled [Kp/32][Np][32] (one wave DMA = 1 KiB contiguous) for conv_gemm_kernel; else [Kp/64][Np][64]
F_PW = 4096         # pointwise conv over <= 64 input channels / <= 64 couts (conv_pw.hip): weights plain [Np][cinp] fp16
F_IMGW = 8192       # OP_CONV (1x1 on conv_gemm_kernel): weights differ per image and come from in2 ([N,1,1,Kp*Np], written by
                    # OP_WSCALE): an SE gate folded into its 1x1 consumer; M tiles do not straddle images
F_U8SRC = 65536     # OP_CONV | F_STEM reading the plan input: the input is the uint8 BGR frames and the kernel resizes them itself (cv2
                    # fixed-point bilinear, the bytes vse_det_preprocess writes in raw mode); frame geometry via vse_plan_set_source
F_OGATE = 131072    # OP_CONV: in2 = gate [N,1,1,Cout] fp16; out = act(conv) * (1 + gate[n, c]) (+ residual): an SE block with shortcut
                    # behind a 1x1 conv folded into it (Compiler._rewrite_se_laterals)
F_DWPRE = 262144    # OP_CONV | F_PW | F_HILO: a depthwise k x k conv (+ bias + activation) is applied to the input ON THE FLY in front of the
                    # 1x1 conv (csrc/conv_dwpw.hip): p[P_KH..P_PW] = the depthwise geometry, aux_off = fp32 blob [k, s, pad, act, act_a, act_b,
                    # post_a, post_b, table [k*k + 1][Kp] (last row = bias)], p[P_LO_IN] = pair offset of the INPUT
F_ONECH = 32768     # OP_CONV with F_PIXSHUF | F_OUT_F32 and ONE real cout: the output is the 1-channel fp32 map itself (ld = 1); every
                    # lane's 8-channel run is one pixel-shuffle quad whose first channel is stored
F_HLSUM = 524288    # OP_CONV | F_COL, 3x3, <= 32 couts of a hi + lo net (the mobile detectors' 96 -> 24 neck convs): ONE pass over K with the lo
                    # weights as 32 more weight rows of a 64-row stage ([3 dy][hi 32 | lo 32][16]); conv_c3_kernel adds its two 32-cout
                    # accumulator tiles in the epilogue (instead of walking the patch chunks twice over a half-empty 32-cout tile)
F_TAIL2 = 1048576   # OP_CONV | F_PW | F_PIXSHUF (2x2 s2 transposed conv c0 -> c1): a SECOND 2x2 s2 transposed conv c1 -> 1 (+ activation) is applied to
                    # the fp16 values stored as `out`, in the same launch (conv_pw_tail_kernel): out2 = the dense fp16 map [n, 4h, 4w] (ld 1),
                    # aux_off = stage B's MFMA A fragments [4 c1p / 16][2][16][8] fp16, f[FS_PRE_B] = its bias, p[P_DOTACT] = its activation
F_COL = 2048        # column-per-step LDS-patch kernel (conv_col.hip): weights packed [cinp/16][kw][kh][Np][16] + 3 zero stages
F_UP2HEAD = 64      # F_SRC2 | F_DOT1 3x3 conv over [1-channel full-res map, x2-upsampled 64-channel map] evaluated on the LOW-RES
                    # grid: weights packed [chunk][parity][2x2 tap][Np][32] + [Np][32] for the 1-channel source (conv_head.hip)
# f[] slots (all kinds that carry an activation)
FS_ACT_A, FS_ACT_B = 0, 1      # hard_sigmoid slope/offset
FS_POST_A, FS_POST_B = 2, 3    # scalar affine after the activation
FS_PRE_A, FS_PRE_B = 4, 5      # scalar affine before the activation (OP_UNARY)
FS_EPS = 6
FS_SCALE = 7

# OP_POOL p[]: P_KH..P_PW as conv, then
P_POOL_MAX, P_POOL_CEIL, P_POOL_EXCL = 6, 7, 8
# OP_BINARY p[]
P_BIN_MUL, P_BIN_SHIFT, P_BIN_ACT = 0, 1, 2
# OP_SCALE flags: F_RES -> out = x + x*s
# OP_ATTN p[]
P_HEADS, P_HDIM = 0, 1
# OP_SOFTMAX p[]
P_NCLS = 0
# OP_LSTM p[]
P_HID, P_REVERSE = 0, 1          # P_REVERSE: 0 forward, 1 reverse, 2 = BOTH directions in one launch (in0 = forward gates, in1 = reverse gates)
F_LSTM_MFMA = 16384              # OP_LSTM: W_hh^T packed in MFMA fragment order [dir][wave 8][k-slice 16][gate 4][lane 64][8] (H = 256, csrc/lstm.hip)


# ragged recogniser batches (every op kind): 1 + width level of in0 / of the output, 0 = the tensor has no per-sample width.
# A plan compiled with ragged=True runs with a device table widths[level][n] (vse_plan_run_ragged): every producer writes
# zeros at x >= widths[out level][n], so each sample computes exactly what a batch of its own width would have given it.
P_WLIN, P_WLOUT = 20, 21


def empty_view():
    return np.zeros((), dtype=VIEW_DT)


# ---- OP_CHAIN: a chain of 1x1 convs (PW) and depthwise convs (DW) with LDS-resident intermediates (csrc/chain.hip) -------------------
# in0 = the chain's input (NHWC fp16), out / out2 / in2 = the (up to three) tensors it stores; w_off = the chain blob:
#   int32 header[CH_HDR] | buffers[nbufs][CH_BUF] | stages[nstages][CH_STAGE] | (16-byte aligned) LDS image
# The LDS image is copied to LDS offset 0 by every block: per PW stage the weight fragments of v_mfma_f32_32x32x16_f16 in lane order
# ([pass hi, lo][cout tile][K slice][lane 64][8 halves]; lane l holds cout tile row conv_wrow(l & 31), k = 16 ks + 8 (l >> 5) + j)
# followed by its fp32 bias [32 nct]; per DW stage one fp32 record per channel [k*k weights, bias, zeros up to a multiple of 4].
OP_CHAIN = 14
CH_MAGIC = 0x43484E31
CH_HDR, CH_BUF, CH_STAGE = 16, 16, 28
# header words
CHH_MAGIC, CHH_NSTAGES, CHH_NBUFS, CHH_LDSW_BYTES, CHH_LDSIMG_OFF, CHH_LDS_TOTAL, CHH_TH, CHH_TW, CHH_TILES_H, CHH_TILES_W = range(10)
# buffer words: kind 0 = channel-minor fp16 hi (+ lo) [pixel][Cp] with a pixel stride of 2 Cp + 16 bytes, 1 = planar fp32 [C][stride],
# 2 = the region of the chain's last output (tile geometry only: nothing is kept in LDS)
(CHB_KIND, CHB_OFF_HI, CHB_OFF_LO, CHB_C, CHB_CP, CHB_STRIDE, CHB_TH, CHB_AH, CHB_EH, CHB_TW, CHB_AW, CHB_EW, CHB_P, CHB_HIMG,
 CHB_WIMG) = range(15)
# stage words (floats stored by bit pattern)
(CHS_TYPE, CHS_IN, CHS_OUT, CHS_RES, CHS_CIN, CHS_COUT, CHS_NKS, CHS_NCT, CHS_K, CHS_S, CHS_PAD, CHS_ACT, CHS_ACT_A, CHS_ACT_B,
 CHS_POST_A, CHS_POST_B, CHS_WLDS, CHS_BLDS, CHS_DWW, CHS_DWB, CHS_GOUT, CHS_MASK, CHS_HASLO, CHS_ACT2, CHS_SHUF) = range(25)
CH_PW, CH_DW = 0, 1
# op.p[] of an OP_CHAIN record
P_CH_TILES_H, P_CH_TILES_W, P_CH_LDS, P_CH_NSTAGES, P_CH_NBUFS = 0, 1, 2, 3, 4
P_CH_PW2, P_CH_IMG = 5, 6     # P_CH_PW2 = 1: a 1x1 -> 1x1 chain with the pixel-shuffle map store (the DB head's tail): the launcher may run the
                               # REGISTER form chain_pw2_kernel on the same blob; P_CH_IMG = bytes of the blob's LDS image
P_CH_LO_IN, P_CH_LO_OUT0, P_CH_LO_OUT1, P_CH_LO_OUT2 = 10, 11, 12, 13     # channel offset of the lo half of a hi + lo tensor (0 = plain fp16)


def dev_switch(name, default=None):
    """An EXPERIMENT switch of the compiler (kernel-family routing, thresholds, ablations): read from the environment only in a development
    build (VSE_DEV_BUILD=1, the build that also carries the experimental kernels); the product always takes the default.  The product's
    own switches are listed in INTEGRATION.md."""
    import os
    return os.environ.get(name, default) if os.environ.get("VSE_DEV_BUILD", "0") == "1" else default
