/* cpr_hip.h -- C ABI of libcprhip.so: the MI355X (gfx950) kernels of the CPR / P2PNet point-localization hot path.
 *
 * Conventions:
 *   - every function returns 0, CPR_ERR_ARG (-1001, shape/pointer check failed), CPR_ERR_UNSUPPORTED (-1002) or
 *     -(hipError_t); nothing is thrown; the Python host raises RuntimeError on non-zero
 *   - all pointers are DEVICE pointers unless marked [host]; the caller allocates every output and workspace
 *     (no hidden allocation, no ownership transfer); the library keeps NO mutable state: every call is a pure function of
 *     its arguments and may be issued from any host thread on any stream (what a launch did -- e.g. which template instance
 *     the tile heuristic picked -- comes back through [host] out-parameters, never through a "last call" global).
 *     The only exception is the measurement build (-DCPR_BENCH_HOOKS, libcprhip_bench.so, see the end of this file)
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); no host synchronisation inside
 *   - activations are NHWC fp32; index outputs are int64 (torch long) where the reference returns long
 * Each entry cites the reference interface it replaces (T/ = TOV_mmdetection/).
 */
#ifndef CPR_HIP_H
#define CPR_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define CPR_ERR_ARG (-1001)
#define CPR_ERR_UNSUPPORTED (-1002)

int cpr_version(void);

/* ---- conv stack ------------------------------------------------------------------------------------------------
 * Replaces ATen/cuDNN conv2d + eval BatchNorm + ReLU + residual add as used by
 *   ResNet.forward / Bottleneck.forward  T/mmdet/models/backbones/resnet.py:262-302,630-645
 *   FPN.forward                          T/mmdet/models/necks/fpn.py:166-194
 *   CPRHead.forward_single               T/mmdet/models/point/dense_heads/cpr_head.py:1033-1043
 *   P2PHead.forward_single               T/mmdet/models/point/dense_heads/p2p_head.py:113-123
 * in  (N,H,W,Cin)  wgt [Cout][KH][KW][Cin] rows padded to Kpad  out (N,OH,OW,Cout)
 * epilogue  y = acc*scale[c] + bias[c] (+ residual[m][c]) (ReLU);   scale/bias/residual may be NULL
 * in_a/in_b (N,Cin) optional: input is read as relu?(x*a+b) (fused GroupNorm-apply of the producer; needs H*W%128==0)
 * gn_part optional [N*OH*OW/128][Cout][2]: per-tile per-channel (sum, sumsq) of the output (needs OH*OW%128==0)
 * Cin must be a multiple of 32, or 4 (3-channel stem padded with a zero channel).
 * flags: CPR_CONV_* bits below.  variant_out [host, may be NULL]: bm*1e6 + bn*1e3 + mode*100 + xf*10 + pipe of the template
 * instance that was launched (for profilers; with CPR_CONV_COLSUM the caller needs bm to size the partials). */
#define CPR_CONV_RELU 1     /* ReLU in the epilogue */
#define CPR_CONV_OUT_BF16 2 /* write bf16 (the 3-channel stem runs on the fp32 kernel and hands bf16 to the bf16 layers) */
#define CPR_CONV_RES_MASK 4 /* residual is a ReLU mask source: out = residual > 0 ? v : 0 (backward of a fused ReLU) */
#define CPR_CONV_COLSUM 8   /* gn_part [ceil(M/bm)][Cout][2] holds per-tile per-channel sums for a whole-tensor column sum */
int cpr_conv2d_fwd(const float* in, const float* wgt, float* out, const float* scale, const float* bias,
                   const float* residual, const float* in_a, const float* in_b, float* gn_part, int N, int H, int W,
                   int Cin, int Cout, int KH, int KW, int stride, int pad, int Kpad, int flags, int in_relu,
                   int* variant_out, void* stream);

/* First block of a ResNet stage (Bottleneck.forward, T/mmdet/models/backbones/resnet.py:262-302 with the projection shortcut
 * built by ResLayer, T/mmdet/models/utils/res_layer.py / resnet.py:564-610): out = relu?((conv(in, wgt) * scale + bias)
 * + (conv1x1(in2, wgt2; stride2, unpadded) * scale2 + bias2)) in ONE launch -- the shortcut map is never stored.  Both
 * convolutions produce the same (N, OH, OW, Cout) grid; in2 is (N, H2, W2, Cin2), wgt2 [Cout][Kpad2 = Cin2]; Cin, Cin2
 * multiples of 32.  Bit-identical to cpr_conv2d_fwd(in2, wgt2) followed by cpr_conv2d_fwd(in, wgt, residual = that). */
int cpr_conv2d_dual_fwd(const float* in, const float* wgt, const float* in2, const float* wgt2, float* out,
                        const float* scale, const float* bias, const float* scale2, const float* bias2, int N, int H, int W,
                        int Cin, int Cout, int KH, int KW, int stride, int pad, int Kpad, int H2, int W2, int Cin2,
                        int stride2, int Kpad2, int flags, int* variant_out, void* stream);

/* The HBM-bound 1x1 convs of the bottleneck (conv3 + bn3 + shortcut add + ReLU and the like, T/mmdet/models/backbones/resnet.py:
 * 262-302) as a stream: one persistent workgroup per CU, weight panel resident in LDS, pixel tiles through LDS-DMA
 * (csrc/conv1x1_stream.hip).  in [M][Cin], wgt [Cout][Cin], out / residual [M][Cout]; Cin in {64, 128}: M a multiple of
 * 8192 / Cin, Cout a multiple of 16384 / Cin; Cin = 256: M a multiple of 128, Cout of 64 (a power of two up to 2048); flags = CPR_CONV_RELU | CPR_CONV_RES_MASK; CPR_ERR_UNSUPPORTED otherwise.
 * cpr_conv2d_fwd takes this path by itself for such shapes when the launch has >= 1024 tiles; the results are bit-identical to
 * its tiled kernel (same accumulation order, same epilogue), this entry exists for tests and tools. */
int cpr_conv1x1_stream_fwd(const float* in, const float* wgt, float* out, const float* scale, const float* bias,
                           const float* residual, long long M, int Cin, int Cout, int flags, void* stream);

/* Weight gradient of a stride-1 conv (k = 1, or k = 3 with padding 1) on the bf16 matrix cores -- the mixed-precision training
 * step (reference analogue: torch autograd under mmcv's Fp16OptimizerHook, T/mmdet/apis/train.py:116-119).  dy (N,H,W,Cout) and
 * x (N,H,W,Cin) fp32 or bf16 (dy_bf16, x_bf16; fp32 maps are rounded on the way in), grad [Cout][Cin][k][k] fp32 (accumulate: +=), Cin % 256 == 0, Cout % 64 == 0.  Both maps
 * are rewritten channel-major over a zero-bordered pixel axis (bf16), every tap is an NT GEMM on the LDS-DMA kernel split over
 * the pixels, the partials are summed in fp32 (csrc/conv_wgrad_bf16.hip).  ws: cpr_conv_wgrad_bf16_workspace(...) x 256 bytes
 * (the query returns units of 256 bytes; negative = unsupported shape). */
int cpr_conv_wgrad_bf16_workspace(int N, int H, int W, int Cin, int Cout, int k);
int cpr_conv_wgrad_bf16(const void* dy, int dy_bf16, const void* x, int x_bf16, float* grad, void* ws, int N, int H, int W, int Cin,
                        int Cout, int k, int accumulate, void* stream);
/* Which kernel cpr_conv_wgrad_bf16 takes when both maps are bf16: 0 = channel-major rewrites of both maps + NT GEMM (rounds 3-6 default),
 * 1 = the pixel-major kernel of csrc/conv_wgrad_bf16_tn.hip (round 6: reads the NHWC maps as they are through ds_read_b64_tr_b16, no
 * rewrites).  Initial value from CPR_WGRAD_TN; on < 0 only queries.  Returns the previous value. */
int cpr_wgrad_bf16_set_tn(int on);
/* cpr_conv_wgrad_bf16 with a stride (1 or 2; dy is (N,OH,OW,Cout), OH = (H + 2 (k/2) - k) / stride + 1) -- the strided 3x3 / projection
 * layers of a stage's first block.  Stride 2, Cin % 256 != 0 (Cin % 64 == 0 suffices) and 1x1 layers below 256 couts are the pixel-major
 * kernel's alone: both maps bf16, else CPR_ERR_UNSUPPORTED.  Workspace: cpr_conv_wgrad_bf16_workspace_s (units of 256 bytes). */
int cpr_conv_wgrad_bf16_workspace_s(int N, int H, int W, int Cin, int Cout, int k, int stride);
int cpr_conv_wgrad_bf16_s(const void* dy, int dy_bf16, const void* x, int x_bf16, float* grad, void* ws, int N, int H, int W, int Cin,
                          int Cout, int k, int stride, int accumulate, void* stream);

/* 3x3 / stride 1 / pad 1 convolution as fused Winograd F(2x2,3x3) on the fp32 matrix cores (2.25x fewer multiplies than
 * cpr_conv2d_fwd; same call sites: the CPR head towers cpr_head.py:1033-1043, the FPN output conv fpn.py:190-194, the 3x3 of
 * every stride-1 bottleneck resnet.py:630-645).  The transforms only add/subtract, the result differs from the direct fp32
 * sum by ~1e-6 of the map's max per layer (tests/test_gpu_wino.py states the bound).
 *   cpr_wino_pack_weights: wgt = the cpr_conv2d_fwd layout [Cout][3][3][Cin] (row stride Kpad) -> u, 16*Cin*Cout floats
 *                          (G g G^T per (cout, cin), stored as the kernel's LDS chunk images); Cin % 8 == 0, Cout % 64 == 0
 *                          (the conv itself needs Cin % 16 == 0, Cin >= 32).
 *   cpr_conv3x3_wino_fwd:  in (N,H,W,Cin) -> out (N,H,W,Cout) = conv * scale[c] + bias[c] (ReLU when flags & CPR_CONV_RELU);
 *                          in_a/in_b (N,Cin) optional (Cin <= 512): the input is read as relu?(x*a+b), in_relu selects the ReLU
 *                          (GroupNorm apply of the producing layer fused into the load, as in cpr_conv2d_fwd);
 *                          layout: CPR_WINO_IN_B8 / CPR_WINO_OUT_B8 -- `in` / `out` are channel-blocked [N][C/8][H][W][8]
 *                          instead of NHWC (whole cache lines per 8-channel K chunk: the fast form between chained layers);
 *                          gn_part [N * ceil(H/16) * ceil(W/16)][Cout][2] (may be NULL): per-channel (sum, sumsq) of every 16x16
 *                          output region (an image's regions are contiguous: cpr_gn_finalize with P = regions per image).
 *                          Tensors < 2 GiB (CPR_ERR_UNSUPPORTED).
 *   cpr_gn_apply_b8:       channel-blocked x -> NHWC y = relu?(x*a+b) (a, b NULL: layout conversion only). */
#define CPR_WINO_IN_B8 1
#define CPR_WINO_OUT_B8 2
int cpr_wino_pack_weights(const float* wgt, float* u, int Cin, int Cout, int Kpad, void* stream);
int cpr_conv3x3_wino_fwd(const float* in, const float* u, float* out, const float* scale, const float* bias,
                         const float* in_a, const float* in_b, float* gn_part, int N, int H, int W, int Cin, int Cout,
                         int flags, int in_relu, int layout, void* stream);
/* The two-workgroups-per-CU form of the same convolution (csrc/conv_wino32.hip, round 4): 4-wave workgroups on 8 x 16 pixel
 * regions with 4-channel K chunks, < 80 KB of LDS each, so two are resident per CU and one's epilogue / prologue hides behind the
 * other's MFMAs.  Same arguments as cpr_conv3x3_wino_fwd except: `u` comes from cpr_wino32_pack_weights (16 * Cin * Cout floats,
 * chunks of 4 input channels; Cin % 8 == 0, Cin >= 16, Cout % 64 == 0), gn_part holds cpr_conv3x3_wino32_slots(H, W) slots per
 * image (one per 8 x 16 region), a fused input affine needs Cin <= 256.  Agrees with cpr_conv3x3_wino_fwd to fp32 rounding
 * (different accumulation order), not bit for bit. */
int cpr_wino32_pack_weights(const float* wgt, float* u, int Cin, int Cout, int Kpad, void* stream);
int cpr_conv3x3_wino32_slots(int H, int W);
int cpr_conv3x3_wino32_fwd(const float* in, const float* u, float* out, const float* scale, const float* bias,
                           const float* in_a, const float* in_b, float* gn_part, int N, int H, int W, int Cin, int Cout,
                           int flags, int in_relu, int layout, void* stream);
int cpr_gn_apply_b8(const float* x, const float* a, const float* b, float* y, int N, int H, int W, int C, int relu,
                    void* stream);

/* bf16 compute mode (BASELINE.json configs[4]): bf16 activations / weights / residual, fp32 accumulate
 * (v_mfma_f32_32x32x16_bf16), K chunks of 64 (Cin % 64 == 0, Kpad == KH*KW*Cin), output bf16 or fp32 (out_fp32).
 * No fused producer-GroupNorm input; GroupNorm statistics come from the fp32 accumulators.
 * variant_out [host, may be NULL]: bm*1000 + bn of the launched instance.
 * wgt_frag [may be NULL; Cout % 256 == 0 only]: a second image of the same weights in MFMA-fragment order,
 * [cout / 64][k / 16][j = 0, 1][lane = l + 32 h][8] = wgt[64 g + 2 l + j][16 ks + 8 h .. + 8]; with it the 256 x 256 tile's launches
 * take the instance that loads the weight operand straight into registers (conv_bf16_dma_kernel<4, 2, 4, true>). */
int cpr_conv2d_fwd_bf16(const void* in, const void* wgt, const void* wgt_frag, void* out, const float* scale, const float* bias,
                        const void* residual, float* gn_part, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                        int stride, int pad, int Kpad, int relu, int out_fp32, int* variant_out, void* stream);
/* Mask mode of the call above (relu == 2; the mixed-precision backward, training.py: a data gradient is a forward convolution of the
 * gradient map with the rotated weights): `residual` is the bf16 map a forward ReLU produced, out = v where that map is positive, 0
 * elsewhere (nothing is added), and gn_part receives per-channel SUMS of the masked result, [slots][Cout][2] floats with the sum in
 * element 0 (cpr_part_colsum adds the slots up).  Only the LDS-DMA instances know the mode: cpr_conv2d_bf16_mask_slots returns the
 * number of slots a launch of this shape writes, or 0 when the shape would not run in mask mode (cpr_conv2d_fwd_bf16 then returns
 * CPR_ERR_UNSUPPORTED for relu == 2).  Reference: torch autograd's ReLU backward + conv2d input gradient + the bias-gradient row sum,
 * three kernels behind mmcv's Fp16OptimizerHook (T/mmdet/apis/train.py:116-119). */
int cpr_conv2d_bf16_mask_slots(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int out_fp32);
/* The block-boundary data gradient of the same backward in one launch: g = mask > 0 ? conv(in, wgt) + add32 : 0 (add32: the shortcut
 * gradient, fp32, the output's shape), written twice -- out32 (fp32: the shortcut chain stays fp32) and out16 (its bf16 rounding, what
 * the next weight / data gradients read) -- with the column sums of g in part (slots: cpr_conv2d_bf16_mask_slots with out_fp32 = 2;
 * 0 = no instance for this shape, the call then returns CPR_ERR_UNSUPPORTED).  Replaces conv + fp32 store + one streaming pass
 * (cpr_relu_bwd_colsum with add): 20 bytes per element -> 12.  wgt_frag is required (the instances are the pair-epilogue ones). */
int cpr_conv2d_dgrad_bf16_fused(const void* in, const void* wgt, const void* wgt_frag, float* out32, void* out16, const float* add32,
                                const void* mask, float* part, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                int pad, int Kpad, int* variant_out, void* stream);
/* bf16 compute mode stem: conv 7x7 / stride 2 / pad 3, 3 -> 64 channels + folded BatchNorm + ReLU (resnet.py:630-636) on the
 * bf16 matrix cores.  in: layout 0 = (N,H,W,4) fp32 (4th channel ignored), layout 1 = (N,3,H,W) fp32 planes (the NCHW network input); wgt (64, 224) bf16 with k = (kh * 8 + kw) * 4 + c (kw = 7 and
 * c = 3 zero), out (N, (H-1)/2+1, (W-1)/2+1, 64) bf16.  CPR_ERR_UNSUPPORTED when the output reaches 2 GiB. */
int cpr_stem7x7s2_bf16(const float* in, const void* wgt, const float* scale, const float* bias, void* out, int N, int H, int W,
                       int relu, int layout, void* stream);
/* The same stem with the max-pool 3x3 / stride 2 / pad 1 (resnet.py:637) fused: out (N, PH, PW, 64) bf16, PH = (OH-1)/2+1.
 * Bit-identical to cpr_stem7x7s2_bf16 (relu = 1) followed by cpr_maxpool3x3s2_bf16. */
int cpr_stem7x7s2_pool_bf16(const float* in, const void* wgt, const float* scale, const float* bias, void* out, int N, int H,
                            int W, int layout, void* stream);
int cpr_maxpool3x3s2_bf16(const void* in, void* out, int N, int H, int W, int C, void* stream);
int cpr_gn_stats_bf16(const void* x, float* part, int N, int HW, int C, int P, void* stream);
int cpr_gn_apply_bf16(const void* x, const float* a, const float* b, const void* up, void* y, int N, int H, int W,
                      int C, int UH, int UW, int relu, void* stream);

/* network input (N,C<=4,H,W) NCHW -> (N,H,W,4) NHWC, missing channels zero */
int cpr_nchw_to_nhwc4(const float* in, float* out, int N, int C, int H, int W, void* stream);
/* (N,H,W,C) -> dense (N,C,H,W) (export in the reference's layout) */
int cpr_nhwc_to_nchw(const float* in, float* out, int N, int C, int H, int W, void* stream);
/* nn.MaxPool2d(3, 2, 1) of the ResNet stem (resnet.py:610,637); NHWC */
/* The ResNet stem in one kernel, exact fp32 (resnet.py:630-637): conv 7x7 / stride 2 / pad 3, 3 -> 64 + folded BatchNorm + ReLU
 * + max-pool 3x3 / stride 2 / pad 1.  in: layout 0 = (N,H,W,4) fp32 (4th channel ignored), layout 1 = (N,3,H,W) fp32 planes; wgt (64, 154) fp32 = [cout][kh][kw * 3 + c] with
 * slot 21 of every kernel row zero, out (N, PH, PW, 64) fp32, OH = (H-1)/2+1, PH = (OH-1)/2+1. */
int cpr_stem7x7s2_pool_f32(const float* in, const float* wgt, const float* scale, const float* bias, float* out, int N, int H,
                           int W, int layout, void* stream);
int cpr_maxpool3x3s2(const float* in, float* out, int N, int H, int W, int C, void* stream);

/* GroupNorm of mmcv ConvModule (fpn.py:124-144, cpr_head.py:990-991) as three streaming steps:
 *   stats    part [N][P][C][2] per-slot per-channel (sum, sumsq)
 *   finalize per (image, channel) affine a = rstd*gamma, b = beta - mean*a   (y = x*a + b); mean/rstd optional (N,G)
 *   apply    y = x*a + b (ReLU) (+ nearest-upsampled `up` (N,UH,UW,C): the FPN top-down add, fpn.py:179-188) */
int cpr_gn_stats(const float* x, float* part, int N, int HW, int C, int P, void* stream);
int cpr_gn_finalize(const float* part, const float* gamma, const float* beta, float* a_out, float* b_out,
                    float* mean_out, float* rstd_out, int N, int P, int C, int G, int HW, float eps, void* stream);
int cpr_gn_apply(const float* x, const float* a, const float* b, const float* up, float* y, int N, int H, int W,
                 int C, int UH, int UW, int relu, void* stream);

/* ---- CPR point kernels -----------------------------------------------------------------------------------------
 * pseudo_bbox_to_center (cpr_head.py:1293-1301): boxes (n,4) -> centers (n,2) */
int cpr_box_centers(const float* boxes, float* centers, int n, void* stream);

/* Probability types of CPRHead.get_cls_prob (cpr_head.py:1080-1099), argument `prob_type` below:
 *   0 sigmoid   1 softmax over the class dim   2 normed_sigmoid (sigmoid, then L_p normalisation over the class dim,
 *   p = norm_p)   3 (cpr_mil_loss only) the cls channels already hold probabilities. */
#define CPR_PROB_SIGMOID 0
#define CPR_PROB_SOFTMAX 1
#define CPR_PROB_NORMED_SIGMOID 2
#define CPR_PROB_IDENTITY 3

/* CPRHead.get_pts_outs for EVERY pixel (cls_out / ins_out = nn.Linear(Cin -> C), cpr_head.py:1007-1014,1045-1078) as one
 * HBM-streaming pass: x (N,HW,Cin) fp32 NHWC, w (J,Cin), bias (J) -> out (N,HW,J); in_a/in_b (N,Cin) optional: the input
 * is read as relu?(x*a+b) (GroupNorm-apply of the last tower layer fused into the load).  Small J only: returns
 * CPR_ERR_UNSUPPORTED when J > 8 or Cin is not 64, 128, 192 or 256 (the caller then uses cpr_conv2d_fwd). */
int cpr_logit_project(const float* x, const float* w, const float* bias, const float* in_a, const float* in_b,
                      float* out, int N, int HW, int Cin, int J, int in_relu, void* stream);

/* OutCirclePtFeatGenerator.generate + neg branch of CPRHead.loss0 (cpr_head.py:254-290,1219-1228):
 * logit (N,H,W,J) (class logits in channels [0,C)); annotated points in CSR form (centers (P,2), labels (P),
 * gt_start (N+1)) -- with num_refine > 1 every refine point is a row, carrying its gt's label;
 * pad_hw (N,2) int32.  mask (N*H*W, C) uint8 = the reference's neg `valid`; partial[] (double, N*ceil(H*W*C/256))
 * = block partial sums of gfocal(prob(logit), 0, valid).  d2_thr = smallest fp32 whose torch-CPU sqrt is
 * >= stride*radius (the reference thresholds cdist, i.e. a sqrt).  n_partial [host, may be NULL] receives the count.
 * mask_classes = C, or 1 with C = 2 (normal_cfg.out_bg_cls, cpr_head.py:953: the one-class validity broadcasts over
 * [class, background] outputs). */
int cpr_neg_mask_loss(const float* logit, int J, const float* centers, const int* labels, const int* gt_start,
                      const int* pad_hw, unsigned char* mask, double* partial, int N, int H, int W, int C,
                      float stride, float d2_thr, float eps, int class_wise, int prob_type, float norm_p,
                      int mask_classes, int* n_partial, void* stream);

/* CirclePtFeatGenerator.generate (cpr_head.py:453-497,172-199): bag points (rings + the point itself last), validity
 * and bilinear samples of a J-channel NHWC map, one bag per annotated / refine point.  offsets (K-1,2) from the host.
 * pts (G,K,2) valid (G,K) out (G,K,J).  align_corners: the generators' align_corners=True form (cpr_head.py:81-84: grid
 * 2x/(w-1)-1, zeros padding instead of the border clip); pad_value (J) or NULL: what a dropped tap contributes per unit of
 * weight -- the projection's bias when the map holds logits (the reference samples zero FEATURES, then applies the Linear). */
int cpr_bag_sample(const float* map, int J, const float* centers, const int* gt_img, const int* pad_hw,
                   const float* offsets, float* pts, unsigned char* valid, float* out, int G, int K, int H, int W,
                   float stride, int align_corners, const float* pad_value, void* stream);

/* GridCirclesPtFeatGenerator.generate (cpr_head.py:296-352,405-438): per gt (R refine points each, points (G*R,2),
 * gt_img (G)) every grid point within radius_px of any of its refine points, row-major, zero-padded to Kmax, then the R
 * refine points in reversed order.  pts (G,Kmax+R,2), valid (G,Kmax+R) u8, out (G,Kmax+R,J) (grid entries = map value of
 * the cell, refine points = bilinear samples, padding slots = pad_value (J) or 0 when NULL: the reference pads with zero
 * FEATURES, i.e. the projection's bias on a logit map), count (G) int32 = grid points found (> Kmax: the reference raises),
 * ws_cell (G,Kmax+R) int32 workspace. */
int cpr_grid_bag(const float* map, int J, const float* points, const int* gt_img, int R, int Kmax, float radius_px,
                 const float* pad_value, float* pts, unsigned char* valid, int* ws_cell, int* count, float* out, int G,
                 int H, int W, float stride, int align_corners, void* stream);

/* MILLoss.forward / AllPosLoss.forward + gt loss + neg normalisation (multi_instance_learning_loss.py:153-243,
 * cpr_head.py:1159-1228).  logits (entries,J): cls in [0,C), ins in [ins_off, ins_off + C*(1+binary_ins)).
 * Bag b = entries [b*bag_stride + bag_off, + bag_len): covers refine_bag_policy independent_with_gt_bag (stride K),
 * merge_to_gt_bag (stride R*K, len R*K) and only_refine_bag (off si*K) without re-packing.  Annotated-point ("gt") loss
 * entries of bag b: b*bag_stride + ctr_off + j*ctr_stride (j < ctr_count), used when b % ctr_mod == 0.
 * labels / gt_weight are per bag.  allpos = AllPosLoss instead of MILLoss.  bag_ws (num_bags,5) workspace.
 * out5 = {gt_loss, pos_loss, bag_acc, neg_loss, num_sample} (device scalars, no host sync); neg_from_gt: average the
 * negative loss over the gt count instead of num_sample (with_mil_loss = False). */
int cpr_mil_loss(const float* logits, int J, int ins_off, const unsigned char* valid, const int* labels,
                 const float* gt_weight, float* bag_ws, const double* neg_partial, int n_partial, int num_bags,
                 int bag_stride, int bag_off, int bag_len, int ctr_off, int ctr_stride, int ctr_count, int ctr_mod,
                 int C, float eps, int prob_type, float norm_p, int binary_ins, int allpos, float w_mil, float w_gt,
                 float w_neg, int neg_from_gt, float* out5, void* stream);

/* PointRefiner.refine_single (cpr_head.py:711-850).  A gt owns Kt = Rv*Kv bag entries (Rv sub-bags of Kv; entry Kv-1 is
 * the annotated point); centers holds ctr_stride points per gt of which the first Rv take part in the nearest filter.
 * refine_pts (G,2), scores (G), not_refine (G) u8, chosen (G,Kt) u8.  score_max: return_score_type 'max'
 * (cpr_head.py:840-842) instead of the mean of the kept probabilities. */
int cpr_refine(const float* logits, int J, const float* pts, const unsigned char* valid, const float* centers, int Rv,
               int ctr_stride, const int* labels, const int* gt_img, const int* gt_start, const int* img_hw,
               const unsigned char* not_refine_in, float* refine_pts, float* scores, unsigned char* not_refine,
               unsigned char* chosen, int G, int Kt, int Kv, int C, int prob_type, float norm_p, float gt_alpha,
               float merge_th, float refine_th, int use_nearest, int use_classify, int score_max, void* stream);

/* ---- assigners -------------------------------------------------------------------------------------------------
 * PointAssigner.assign (T/mmdet/core/bbox/assigners/point_assigner.py:23-133): points (n,3)=(x,y,stride),
 * gt_bboxes (k,4) -> gt_inds (n) int64 (0 bg, j+1 = gt j).  ws_best (n) float, ws_lvl (n) int32 workspaces. */
int cpr_point_assign(const float* points, const float* gt_bboxes, int n, int k, float scale, int pos_num,
                     long long* gt_inds, float* ws_best, int* ws_lvl, void* stream);

/* FocalLossCost + DisCostV2 (T/mmdet/core/bbox/match_costs/match_cost.py:84-100,197-214):
 * costT (G,M) = transpose of the reference's (M,G) cost; pred (M,pred_stride>=2), logits (M,C), gt (G,2);
 * p_norm = DisCostV2's p: 1 (|dx|+|dy|) or 2 (torch.cdist's Euclidean forms) */
int cpr_hungarian_cost(const float* pred, int pred_stride, const float* logits, int C, const float* gt,
                       const int* labels, float* costT, int M, int G, float w_cls, float alpha, float gamma,
                       float eps, float w_dis, float fx, float fy, int p_norm, void* stream);
/* The general cost matrix of HungarianAssignerV2 / HungarianAssigner (hungarian_assigner.py:15-145,166-229): any list of the costs of
 * T/mmdet/core/bbox/match_costs/match_cost.py, summed as ``sum(cls_costs) + sum(reg_costs)`` sums them.  terms [host]: 6 floats per
 * term (type, weight, a, b, c, d), the ncls classification terms first -- cls 0 FocalLossCost (alpha, gamma, eps), 1 ClassificationCost
 * (-softmax[label]), 2 ClassificationCostV2(use_sigmoid) (-sigmoid[label]), 3 ZeroCost; reg 0 DisCostV2 (p, fx, fy) on points (pdim 2),
 * 1 BBoxL1Cost, 2 IoUCost 'iou', 3 IoUCost 'giou' on xyxy boxes (pdim 4).  costT (G, M) as cpr_hungarian_cost. */
int cpr_match_cost(const float* pred, int pdim, const float* logits, int C, const float* gt, const int* labels, float* costT,
                   int M, int G, const float* terms, int ncls, int nreg, void* stream);

/* The linear_sum_assignment loop of HungarianAssignerV2.assign (hungarian_assigner.py:229-268; replaces scipy and
 * the device->host->device round trip), including scipy's tie-breaking order.  A batch of problems, one workgroup
 * each: problem b has costT at cost_off[b] (G_b x M_b, M_b >= G_b), column arrays at col_off[b], row arrays at
 * row_off[b].  gt_inds (sum M) int64: 0 background, j+1 = gt j.  status[b] != 0: infeasible.
 * Workspaces: per column ws_v/ws_spc (double), ws_path/ws_row4col/ws_cols/ws_remaining/ws_pos (int32), ws_sc/ws_active
 * (uint8); per row ws_u (double), ws_col4row (int32), ws_sr (uint8).  max_cols / max_rows: [host] upper bounds of M_b / G_b
 * over the batch (0 = unknown): when max_cols <= 32768 and topk * max_rows < 65534 the search state of a row lives in
 * registers / LDS (one 4-byte cost read per column per scan instead of ~33 bytes of state; same results bit for bit). */
int cpr_lsa_topk(const float* costT, const int* m_of, const int* g_of, const long long* cost_off,
                 const long long* col_off, const long long* row_off, int num_problems, int topk, long long* gt_inds,
                 double* ws_v, double* ws_spc, int* ws_path, int* ws_row4col, unsigned char* ws_sc,
                 unsigned char* ws_active, int* ws_cols, int* ws_remaining, int* ws_pos, double* ws_u,
                 int* ws_col4row, unsigned char* ws_sr, int* status, int max_cols, int max_rows, void* stream);

/* ---- P2P inference ---------------------------------------------------------------------------------------------
 * per-level top-k of P2PHead._get_bboxes_single (p2p_head.py:367-373): scores (n) -> k largest, sorted descending
 * (ties: lower index first).  k <= 4096.  out_vals (k) float, out_idx (k) int64. */
int cpr_topk_desc(const float* scores, int n, int k, float* out_vals, long long* out_idx, void* stream);

/* Candidate list of multiclass_nms (T/mmdet/core/post_processing/bbox_nms.py:28-60): every (proposal, class) pair with
 * scores[p][c] > score_thr (scores (n,C+1), last column = background), row-major order; boxes (n, box_stride) with
 * box_stride = 4 (shared by the classes) or 4*C; factors (n) optional score_factors.  Outputs sized n*C:
 * cand_boxes (.,4), cand_scores, cand_labels (int32), cand_inds (int64, p*C + c), count (1) int32. */
int cpr_nms_candidates(const float* boxes, int box_stride, const float* scores, const float* factors, int n, int C,
                       float score_thr, float* cand_boxes, float* cand_scores, int* cand_labels, long long* cand_inds,
                       int* count, void* stream);
/* mmcv.ops.nms.batched_nms as called by multiclass_nms (T/mmdet/core/post_processing/bbox_nms.py:85; mmcv-full
 * 1.3.x, third-party): class-offset trick, sort by score (descending, ties by index), greedy IoU > thr suppression.
 * boxes (n,4), scores (n), labels (n) int32.  keep_idx (n) int64: indices of kept boxes in descending score order; num_keep (1)
 * int32.  ws_order (n) int32, ws_boxes (n,4) float, ws_mask cpr_nms_workspace(n) uint64 words.  No candidate ceiling below
 * 262 144 (round 6; the reference has none: bbox_nms.py:7-94): above 8192 candidates the pair mask is computed and scanned a
 * band of 8192 rows at a time, the removed bits and the keep count carried from band to band. */
int cpr_nms_workspace(int n);
int cpr_nms(const float* boxes, const float* scores, const int* labels, int n, float iou_thr, long long* keep_idx,
            int* num_keep, int* ws_order, float* ws_boxes, unsigned long long* ws_mask, void* stream);
/* The same three stages for ALL images of a batch with no host round trip in between (P2PHead.get_bboxes loops
 * _get_bboxes_single over the images, p2p_head.py:330-343; each image's torch.topk / nonzero / nms sync the host there):
 *   cpr_topk_desc_batched        scores (B, n) -> vals (B, k), idx (B, k) (segment-relative), one workgroup per image
 *   cpr_nms_candidates_batched   boxes (B, n, box_stride), scores (B, n, C+1), factors (B, n) or NULL -> slabs of capacity
 *                                n * C per image: cand_boxes (B, n*C, 4), cand_scores / labels / inds (B, n*C), count (B) [device]
 *   cpr_nms_batched              reads the per-image candidate counts n_dev (B) FROM THE DEVICE; keep_idx (B, cap) slab-relative,
 *                                num_keep (B) [device]; ws_order (B, cap) int32, ws_boxes (B, cap, 4), ws_mask B * ws_stride
 *                                64-bit words, ws_stride >= max(cap * ceil(cap / 64), next pow2 >= cap); cap <= 16384
 * The caller reads (count, num_keep) once per batch. */
int cpr_topk_desc_batched(const float* scores, int B, int n, int k, float* out_vals, long long* out_idx, void* stream);
int cpr_nms_candidates_batched(const float* boxes, int box_stride, const float* scores, const float* factors, int B, int n,
                               int C, float score_thr, float* cand_boxes, float* cand_scores, int* cand_labels,
                               long long* cand_inds, int* count, void* stream);
int cpr_nms_batched(const float* boxes, const float* scores, const int* labels, const int* n_dev, int B, int cap,
                    float iou_thr, long long* keep_idx, int* num_keep, int* ws_order, float* ws_boxes,
                    unsigned long long* ws_mask, long long ws_stride, void* stream);

/* P2PHead.get_pred_points (p2p_head.py:125-170): reg (N,H,W,2k) -> pred (N,H*W*k,3) = anchor + point_anchor*stride +
 * reg*gamma*stride, third column = stride; anchor (same shape) optional.  point_anchor (k,2). */
int cpr_p2p_decode(const float* reg, const float* point_anchor, float* pred, float* anchor, int N, int H, int W, int k,
                   float stride, float gamma, void* stream);
/* max_c sigmoid(logits[m][c]) -> out (M): the score P2PHead._get_bboxes_single ranks with (p2p_head.py:362-369) */
/* out (N,H,W,J) = bias[j] + sum over the 3x3 taps of R[n][y+kh-1][x+kw-1][(kh*3+kw)*J + j] (taps outside the map add 0):
 * the second half of a 3x3 / pad 1 convolution with J output channels computed as a 1x1 projection to 9 J tap responses
 * (P2PHead's cls_out / reg_out, p2p_head.py:84-123: 1-2 output channels).  bias may be NULL. */
int cpr_tap_sum3x3(const float* R, const float* bias, float* out, int N, int H, int W, int J, void* stream);
int cpr_rowmax_sigmoid(const float* logits, float* out, long long M, int C, void* stream);
/* elementwise sigmoid with the bits of torch's CPU kernel (p2p_head.py:362: the scores that order top-k and NMS) */
int cpr_sigmoid(const float* x, float* y, long long n, void* stream);
/* P2PHead.loss_single + sample_result_to_target (p2p_head.py:220-248,308-328) straight from gt_inds (B,M) int64.
 * cls_mode 0: sigmoid focal loss (T/mmdet/models/losses/focal_loss.py:11-56), averaged by the batch's positives; 1: CrossEntropyLoss
 * (use_sigmoid=True) = BCE with logits on the one-hot labels (cross_entropy_loss.py:58-99), averaged by ALL proposals of the batch
 * (p2p_head.py:222-224); 2: softmax cross entropy over C = num_classes + 1 logits, background last.  reg_mode 0: SmoothL1(beta)
 * (smooth_l1_loss.py:11-28), 1: MSE (mse_loss.py), 2: L1; averaged by the positives.  (1, 1) are the reference head's own defaults
 * (p2p_head.py:38-46), (0, 0) the shipped config.  ws_partial (B*ceil(M/256)*3) double; out (B,2) = per image {loss_cls, loss_pts}. */
int cpr_p2p_loss(const float* logits, const float* pred, const long long* gt_inds, const float* gt_pts,
                 const int* gt_labels, const int* gt_start, double* ws_partial, float* out, int B, int M, int C,
                 float alpha, float gamma, float beta, float pos_w, float neg_w, float reg_norm, float w_cls,
                 float w_reg, int cls_mode, int reg_mode, void* stream);

/* ---- training step: backward + optimizer ------------------------------------------------
 * The reference gets these from torch autograd over mmcv ConvModule / nn.GroupNorm / F.grid_sample and
 * torch.optim.SGD driven by mmcv's OptimizerHook (T/configs2/COCO/CPRNet/CPR_R50_FPN_1x_coco_coarse.py optimizer /
 * optimizer_config); tests/test_gpu_backward.py checks every function against torch autograd on the oracle.
 *
 * weight gradient of conv2d_fwd: dy (N,OH,OW,Cout) x (N,H,W,Cin) NHWC fp32 -> grad_w [Cout][Cin][KH][KW] (the
 * parameter's own layout).  Optional fused input transform x' = max(x*in_a[n,c]+in_b[n,c], in_relu?0:-inf) (the
 * GroupNorm(+ReLU) the forward applied on load).  ws: cpr_conv2d_wgrad_workspace(...) floats.  Cin%4==0, Cout%4==0. */
/* benchmark hook: main-loop ablations of the plain weight-gradient kernel (1 no loads, 2 no LDS stores, 4 no barrier,
 * 8 no fragment reads, 3 = 1|2, 15 = all); results are wrong by construction, 0 restores the product kernel */
int cpr_conv2d_wgrad_workspace(int N, int OH, int OW, int Cin, int Cout, int KH, int KW);
int cpr_conv2d_wgrad(const float* dy, const float* x, const float* in_a, const float* in_b, float* grad_w, float* ws,
                     int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int in_relu,
                     int accumulate, void* stream);
/* The same weight gradient for 3x3 / stride 1 / pad 1 layers as fused Winograd F(2x2,3x3) (the adjoint of
 * cpr_conv3x3_wino_fwd with respect to the weights: 2.25x fewer multiplies; csrc/conv_wino_wgrad.hip).  Cin % 64 == 0,
 * Cout % 64 == 0; ws: cpr_conv3x3_wino_wgrad_workspace(...) floats (split-K partials, reduced inside the call). */
int cpr_conv3x3_wino_wgrad_workspace(int N, int H, int W, int Cin, int Cout);
int cpr_conv3x3_wino_wgrad(const float* dy, const float* x, const float* in_a, const float* in_b, float* grad_w, float* ws,
                           int N, int H, int W, int Cin, int Cout, int in_relu, int accumulate, void* stream);
/* nn.GroupNorm (+ReLU) backward from the raw conv output x, the forward affine a,b (N,C), mean/rstd (N,G):
 * dx (N,HW,C), dgamma/dbeta (C).  ws_part N*P*C*2 floats, ws_k 2*N*G + 2*N*C floats. */
int cpr_gn_bwd(const float* x, const float* dz, const float* a, const float* b, const float* mean, const float* rstd,
               const float* gamma, float* dx, float* dgamma, float* dbeta, float* ws_part, float* ws_k, int N, int HW,
               int C, int G, int P, int relu, int accumulate, void* stream);
/* The same backward reading the bf16 map the mixed-precision forward recorded (widened in registers: the values are those of
 * cpr_gn_bwd on the widened map) and writing dx (fp32) and / or dx_bf16 (its round-to-nearest-even narrowing, what the bf16 weight /
 * data gradient kernels read) -- at least one of the two. */
int cpr_gn_bwd_bf16(const void* x_bf16, const float* dz, const float* a, const float* b, const float* mean, const float* rstd,
                    const float* gamma, float* dx, void* dx_bf16, float* dgamma, float* dbeta, float* ws_part, float* ws_k, int N,
                    int HW, int C, int G, int P, int relu, int accumulate, void* stream);
/* ... with the upstream gradient dz in bf16 as well (round 6: the bf16 data gradient of the layer above writes it in bf16) */
int cpr_gn_bwd_bf16_dz16(const void* x_bf16, const void* dz_bf16, const float* a, const float* b, const float* mean, const float* rstd,
                         const float* gamma, float* dx, void* dx_bf16, float* dgamma, float* dbeta, float* ws_part, float* ws_k,
                         int N, int HW, int C, int G, int P, int relu, int accumulate, void* stream);
/* FPN top-down path backward (fpn.py:176-185): dcoarse (N,UH,UW,C) (+)= sum over the nearest-upsample children of dfine */
int cpr_upsample_add_bwd(const float* dfine, float* dcoarse, int N, int H, int W, int UH, int UW, int C, int accumulate,
                         void* stream);
/* backward of the fused conv epilogue y = relu?(conv*scale + shift (+ identity)) of an eval-mode BatchNorm
 * (resnet.py Bottleneck.forward): g = dy*(y>0) (y NULL: g = dy) is the shortcut gradient and the un-scaled conv-output
 * gradient; colsum (C) (+)= per-channel sums of g (= dshift; also the Linear/conv bias gradient).  (M,C) row-major,
 * C%4==0.  g_out may be NULL (sums only).  ws_part: cpr_relu_bwd_colsum_ws(M, C) floats. */
int cpr_relu_bwd_colsum_ws(long long M, int C);
int cpr_relu_bwd_colsum(const float* dy, const float* add, const void* y, int y_bf16, float* g_out, void* g16_out, float* colsum,
                        float* ws_part, long long M, int C, int accumulate, void* stream);   /* y_bf16: y is the bf16 map the mixed-precision
                        forward recorded; g16_out (optional): the bf16 rounding of g, written by the same pass; add (optional, fp32, same
                        shape): g = (dy + add) masked -- the shortcut gradient joining a block's data gradient (round 6) */
/* column sums (C) of a conv output from the epilogue partials cpr_conv2d_fwd wrote into gn_part [tiles][C][2];
 * ws: 64*C floats */
int cpr_part_colsum(const float* part, float* out, float* ws, int tiles, int C, void* stream);
/* parameter side of the folded BN: Gw = cpr_conv2d_wgrad(g, x) [Cout][K] -> in place dW = scale[c]*Gw[c];
 * dgamma = inv_sigma*(<W[c],Gw[c]> - mean*colsum_g), dbeta = colsum_g (either may be NULL). */
int cpr_bn_fold_bwd(float* Gw, const float* weight, const float* scale, const float* mean, const float* inv_sigma,
                    const float* colsum_g, float* dgamma, float* dbeta, int Cout, int K, void* stream);
/* the same with the column sums still in a conv epilogue's partials [tiles][Cout][2] (element 0 = sum; mask mode / fused data gradients of
 * cpr_conv2d_fwd_bf16, CPR_CONV_COLSUM of cpr_conv2d_fwd): the kernel adds each channel's column up itself (fixed order) */
int cpr_bn_fold_bwd_part(float* Gw, const float* weight, const float* scale, const float* mean, const float* inv_sigma, const float* part,
                         int tiles, float* dgamma, float* dbeta, int Cout, int K, void* stream);
/* y = alpha*x + beta*y on flat fp32 buffers */
int cpr_axpby(float* y, const float* x, float alpha, float beta, long long n, void* stream);
/* phase decomposition of a stride-s data gradient: dst[n, s*i+py, s*j+px, :] += src[n, i+sh, j+sw, :] for all targets
 * inside (H, W); src (N,Hs,Ws,C) = the stride-1 sub-convolution over dy that serves output parity (py, px) */
int cpr_phase_scatter_add(const float* src, float* dst, int N, int Hs, int Ws, int C, int H, int W, int py, int px,
                          int sh, int sw, int s, void* stream);
/* out (N,H,W,C) = dy (N,OH,OW,C) with s-1 zeros inserted between pixels (data gradient of a stride-s conv as a
 * stride-1 conv over the dilated gradient) */
int cpr_zero_insert(const float* dy, float* out, int N, int OH, int OW, int C, int H, int W, int s, void* stream);
/* d(gt_loss + pos_loss + neg_loss)/d(logit map) of CPRHead.loss (cpr_head.py:1101-1229): negative-grid term, MIL bag
 * and gt-centre terms taken back through the bilinear taps -- deterministically: every bag's taps are gathered into a
 * win x win cell window (win >= 2 * ceil(max |offset| / stride) + 3; win_ws (G, win, win, J) fp32 and win_org (G, 2) int32
 * are caller workspaces), the windows are added per image in gt order (gt_img must ascend).  Inputs are the forward's own
 * buffers (neg mask, out5, bag logits, valid, bag_ws).  dbag_ws (G,K,J) workspace; dmap (N,H,W,Jd), Jd >= J, fully written.
 * upstream: NULL (the three losses enter the total with weight 1) or [device] 5 floats, the gradient of the total wrt
 * (gt_loss, pos_loss, bag_acc, neg_loss, num_sample) as torch autograd hands it to the head's loss Function; entries 2 and 4
 * carry no gradient and are ignored. */
int cpr_loss_bwd(const float* lmap, const unsigned char* neg_mask, const float* out5, const float* bag_logits,
                 const unsigned char* valid, const int* labels, const float* gt_weight, const float* bag_ws,
                 const float* centers, const int* gt_img, const float* offsets, float* dbag_ws, float* dmap, float* win_ws,
                 int* win_org, int win, int N, int H, int W, int J, int Jd, int ins_off, int G, int K, int C, float stride,
                 float eps, float w_mil, float w_gt, float w_neg, const float* upstream, void* stream);
/* win == 0 in cpr_loss_bwd: the bag logits were NOT sampled from the logit map (CPRHead num_cls_fcs > 0 samples the features and
 * runs them through the FC stack, cpr_head.py:1055-1059) -- dmap then holds the negative-grid term alone and dbag_ws (G,K,J) is
 * the gradient wrt the bag logits.  cpr_bag_gather_bwd is the gather stage on its own: dsample (G,K,J), the gradient wrt the
 * bilinear samples of a (N,H,W,.) map, is ADDED onto dmap (N,H,W,Jd) through bag_sample's taps (same windows, same order). */
int cpr_bag_gather_bwd(const float* dsample, int J, const float* centers, const int* gt_img, const float* offsets,
                       float* win_ws, int* win_org, int win, float* dmap, int N, int H, int W, int Jd, int G, int K,
                       float stride, void* stream);
/* The gather stage for bags whose taps are not ring offsets around a centre (round 5): GridCirclesPtFeatGenerator bags
 * (cpr_head.py:296-352,405-438) and align_corners=True sampling (:73-93,126).  Entries are the forward's own outputs: pts (G,Kt,2)
 * and code (G,Kt) -- cpr_grid_bag's ws_cell: cell index y*W+x | -1 padding slot | <= -2 bilinear sample at pts; NULL = every entry
 * is a bilinear sample (cpr_bag_sample).  dsample (G,Kt,J) is ADDED onto dmap (N,H,W,Jd) per image in gt order (gt_img ascends),
 * deterministically.  wout_ws (G*Kt) and dbias (J): both or neither; dbias <- sum over entries of (weight of the entry's dropped
 * taps) * dsample, the share of the projection's bias when the sampled map held logits (pad_value of cpr_bag_sample / cpr_grid_bag). */
int cpr_bag_points_gather_bwd(const float* dsample, int J, const float* pts, const int* code, const int* gt_img, float* dmap,
                              float* wout_ws, float* dbias, int N, int H, int W, int Jd, int G, int Kt, float stride,
                              int align_corners, void* stream);
/* The same loss gradients for every CPRHead loss option the forward kernels take (round 5): prob_type 0 sigmoid / 1 softmax /
 * 2 normed_sigmoid(norm_p) class probabilities (cpr_head.py:1080-1099), MILLoss(binary_ins) / AllPosLoss
 * (multi_instance_learning_loss.py:153-243), the bag / annotated-point geometry of cpr_mil_loss (refine_bag_policy, gt_loss_type:
 * cpr_head.py:1159-1211), C = classifier outputs (num_classes + 1 with out_bg_cls), neg_from_gt (with_mil_loss=False: the negative
 * term is averaged over the annotated-point positives).  dmap (N,H,W,Jd) <- the negative-grid term alone; dbag
 * (num_bags * bag_stride, J) <- the gradient wrt every bag entry's logits, NOT gathered (cpr_bag_gather_bwd /
 * cpr_bag_points_gather_bwd add it onto dmap).  neg_mask NULL: no negative-grid term (loss_cfg with_neg=False), dmap <- zeros. */
int cpr_loss_bwd_general(const float* lmap, const unsigned char* neg_mask, const float* out5, const float* bag_logits,
                         const unsigned char* valid, const int* labels, const float* gt_weight, const float* bag_ws, float* dbag,
                         float* dmap, int N, int H, int W, int J, int Jd, int ins_off, int num_bags, int bag_stride, int bag_off,
                         int bag_len, int ctr_off, int ctr_stride, int ctr_count, int ctr_mod, int C, float eps, int prob_type,
                         float norm_p, int binary_ins, int allpos, float w_mil, float w_gt, float w_neg, int neg_from_gt,
                         const float* upstream, void* stream);
/* OIHW fp32 master weights -> the conv kernels' layout [rows][KH][KW][cols'] (row stride Kpad, zero padded).
 * transpose 0: forward pack (rows = O).  transpose 1: data-gradient pack (rows = I, taps flipped, optional per-O scale =
 * the folded BatchNorm scale of the forward conv).  colsp = padded column count (4 for <= 4 channels). */
int cpr_pack_weights(const float* w, const float* scale, float* out, int O, int I, int KH, int KW, int colsp, int Kpad,
                     int transpose, void* stream);
/* The same packs for the bf16 conv kernels: out [rows][KH][KW][cols] bf16 (cols even, no padding), each element the
 * round-to-nearest-even of the fp32 value cpr_pack_weights would write; frag (NULL or rows % 64 == 0, KH*KW*cols % 16 == 0): the
 * same values in the fragment order cpr_conv2d_fwd_bf16's wgt_frag documents.  One launch per layer. */
int cpr_pack_weights_bf16(const float* w, const float* scale, void* out, void* frag, int O, int I, int KH, int KW, int transpose,
                          void* stream);
/* One idle wave for `ticks` of the 100 MHz wall clock (<= 1 s) on `stream`: a probe for whether two streams share a hardware queue
 * (the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES queues; two of these launches on one queue run back to back). */
int cpr_spin(long long ticks, void* stream);
/* eval-mode BatchNorm (resnet.py norm_eval) -> conv-epilogue affine: scale = gamma/sqrt(var+eps), shift = beta - mean*scale,
 * inv_sigma (optional) = 1/sqrt(var+eps) */
int cpr_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                float* shift, float* inv_sigma, int C, void* stream);
/* Multi-tensor forms of cpr_bn_fold and cpr_pack_weights_bf16 (round 6; the per-step refresh of the mixed-precision / fp32 training
 * step, layers._PackCache.refresh_all): one launch over a DEVICE table of 64-byte jobs --
 *   fold job: { const float *gamma, *beta, *mean, *var; float *scale, *shift, *inv (each may be NULL); int C; float eps; }
 *   pack job: { const float* w; const float* scale; void* out; void* frag; int O, I, KH, KW, transpose, block0, nblocks, pad; }
 * (pack jobs in ascending block0; job j owns workgroups [block0, block0 + nblocks) of the total_blocks launched).  Results are
 * bit for bit the single-tensor entries'. */
int cpr_bn_fold_multi(const void* jobs_dev, int n, int max_c, void* stream);
/* ... and of cpr_pack_weights (fp32 packs): job { const float* w; const float* scale; float* out; int nblocks, pad;
 * int O, I, KH, KW, colsp, Kpad, transpose, block0; } */
int cpr_pack_weights_multi(const void* jobs_dev, int n, int total_blocks, void* stream);
int cpr_pack_weights_bf16_multi(const void* jobs_dev, int n, int total_blocks, void* stream);
/* gradient of sum_b (loss_cls[b] + loss_pts[b]) of cpr_p2p_loss wrt the class logits (B*M, C) and the regression output
 * (p2p_head.py:220-248; sigmoid focal loss with its un-detached focal weight, SmoothL1 through
 * pred = anchor + (point_anchor + reg*gamma_p)*stride).  npos: device scalar, positives in the batch.  dcls (B*M, Cp),
 * dreg (B*M, Rp): channel-padded for the conv gradient kernels, padding columns written as zero.  upstream: NULL (every loss
 * term enters the total with weight 1) or [device] (B, 2) gradients of the total wrt (loss_cls[b], loss_pts[b]) -- what
 * torch autograd hands the head's loss Function (replaces loss.backward() through p2p_head.py:220-248). */
int cpr_p2p_loss_bwd(const float* logits, const float* pred, const long long* gt_inds, const float* gt_pts,
                     const int* gt_labels, const int* gt_start, const float* npos, float* dcls, float* dreg, int B, int M,
                     int C, int Cp, int Rp, float alpha, float gamma, float beta, float pos_w, float neg_w, float reg_norm,
                     float w_cls, float w_reg, float gamma_p, const float* upstream, int cls_mode, int reg_mode, void* stream);
/* sum of squares of a flat gradient buffer into out[0] (double; accumulate across buffers); ws_partial 1024 doubles */
int cpr_grad_sumsq(const float* g, long long n, double* ws_partial, double* out, int accumulate, void* stream);
/* torch.optim.SGD step (momentum, weight decay) with clip_grad_norm_'s coefficient taken from norm2 on the device:
 * g = min(1, max_norm/(sqrt(norm2)*grad_scale+1e-6)) * grad_scale * grad + wd*p; buf = first ? g : mu*buf+g; p -= lr*buf */
int cpr_sgd_step(float* p, const float* grad, float* buf, const double* norm2, long long n, float lr, float mu, float wd,
                 float max_norm, float grad_scale, int first, void* stream);

/* ---- data side feeding the path ----------------------------------------------------------
 * RandomFlip(horizontal) -> Normalize -> Pad(0 after normalisation) -> channels-last float of the mmdet pipeline
 * (T/mmdet/datasets/pipelines/transforms.py:431-480,560-680, formating.py:180-214) in one pass: img (N,H,W,3) uint8
 * decoded images on the device -> out (N,Hp,Wp,4) fp32, the stem's input layout (4th channel and padding zero).
 * mean3 / stdinv3: HOST pointers to 3 floats (stdinv = float32(1/float64(std)) as mmcv.imnormalize_); flip (N) int32
 * device flags or NULL. */
int cpr_preprocess_u8(const unsigned char* img, const int* flip, const float* mean3, const float* stdinv3, int to_rgb,
                      float* out, int N, int H, int W, int Hp, int Wp, void* stream);
/* Box side of Resize(scale 1) -> RandomFlip, in place: clip every box to its image when `clip` (Resize._resize_bboxes,
 * bbox_clip_border=True, transforms.py:241-249), then mirror it when flip[img] (RandomFlip.bbox_flip, :397-415).
 * boxes (n,4) xyxy, img_of (n) image index, flip (N), img_hw (N,2) int32 = img_shape[:2] */
int cpr_clip_flip_boxes(float* boxes, const int* img_of, const int* flip, const int* img_hw, int n, int clip,
                        void* stream);

/* ---- measurement build only (-DCPR_BENCH_HOOKS; python -m pointtinybenchmark_amd.build --bench-hooks) ------------------
 * Process-global, not thread-safe switches used by the scripts under tools/ to A/B the conv kernels.  NOT part of the
 * product library. */
#ifdef CPR_BENCH_HOOKS
int cpr_conv_force_tile(int bm, int bn); /* force the conv output tile (0 = heuristic, 64, 128) */
int cpr_conv_set_pipeline(int mode);     /* K-loop schedule: 1 = interleaved (product), 0 = phase-separated */
int cpr_conv_set_stream(int on);         /* 0 = the streamed 1x1 kernel (conv1x1_stream.hip) is never chosen by cpr_conv2d_fwd (A/B) */
int cpr_conv_set_ablation(int mode);     /* loop ablations: results are WRONG when non-zero */
int cpr_wgrad_set_ablation(int mode);    /* same for the weight-gradient kernel */
int cpr_conv_set_extra_lds(int bytes);   /* occupancy probe: dynamic LDS added to every direct-conv launch */
int cpr_wino_set_variant(int sched, int ablate); /* Winograd: sched 1 = the other placement of the patch transform (see conv_wino.hip); loop ablations */
int cpr_bf16_set_dma(int on);            /* bf16 mode: 0 = every layer on the register-staged kernels (A/B of conv_bf16_dma.hip) */
int cpr_bf16_set_wfrag(int on);          /* bf16 mode: 0 = ignore wgt_frag (A/B of the weights-direct-to-registers instance) */
int cpr_wino_set_staging(int var, int tpx);      /* Winograd: staging variant (kernel template VAR) and cout tiles per XCD; -1 = the product's choice */
int cpr_wino32_set_debug(int ablate, int wg_per_cu, int stagger_pct); /* conv_wino32.hip: loop ablations, workgroups per CU (1 / 2), stagger of the second wave of workgroups */
int cpr_lsa_phase_clocks(long long* host_out, int reset);            /* assign.hip: shader clocks workgroup 0 of lsa_topk_reg_kernel spent per phase (8 values) */
#endif

#ifdef __cplusplus
}
#endif
#endif /* CPR_HIP_H */
