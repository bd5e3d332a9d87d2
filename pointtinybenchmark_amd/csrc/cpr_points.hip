// CPR-specific point kernels: everything after the head towers operates on the tiny projected logit map
// L[n][y][x][J] (J = cls channels ++ ins channels), produced by ONE 1x1 conv launch over the head feature
// map (Linear(256->C) commutes with bilinear sampling because num_cls_fcs == 0 in every shipped config:
// Linear(bilinear(F)) == bilinear(conv1x1(F))).  The same kernels also run on the raw
// 256-channel feature map (J = 256) for the general num_cls_fcs > 0 path.
//
// Bit-exactness notes (integer / bool outputs are held to the reference's CPU results):
//  * The reference measures point<->gt distances with torch.cdist, which for > 25 rows is the matmul form
//    [-2x, -2y, |p|^2, 1] . [cx, cy, 1, |c|^2] evaluated by MKL sgemm as a k-ordered fp32 FMA chain, followed
//    by sqrt.  d2_chain() below is that chain, operation for operation (explicit *_rn intrinsics so hipcc's
//    default fp-contract cannot re-associate it), so squared distances are bit-identical to the reference's;
//    thresholds are compared in d2 space against the host-derived smallest float whose torch-sqrt passes.
//  * Ring offsets come from the host (torch CPU cos/sin) -- device cosf/sinf round differently.
#include "common.h"

#define MAX_GT_LDS 1024

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------------------------
// CPRHead.get_cls_prob (cpr_head.py:1080-1099): class probabilities of one point from its C class logits.
//   0 sigmoid          sigmoid(l_c)                 (torch-CPU sigmoid bit for bit: the refine thresholds compare it)
//   1 softmax          softmax over the class dim
//   2 normed_sigmoid   sigmoid then F.normalize(p=norm_p) over the class dim (denominator clamped at 1e-12)
#define CPR_PROB_SIGMOID 0
#define CPR_PROB_SOFTMAX 1
#define CPR_PROB_NORMED_SIGMOID 2
__device__ __forceinline__ float cls_prob(const float* __restrict__ l, int C, int c, int ptype, float norm_p) {
    if (ptype == CPR_PROB_SIGMOID) return sigmoid_torch_cpu(l[c]);
    if (ptype == CPR_PROB_SOFTMAX) {
        float m = -INFINITY;
        for (int j = 0; j < C; ++j) m = fmaxf(m, l[j]);
        float se = 0.f;
        for (int j = 0; j < C; ++j) se = __fadd_rn(se, sleef_expf_u10(__fsub_rn(l[j], m)));
        return __fdiv_rn(sleef_expf_u10(__fsub_rn(l[c], m)), se);
    }
    float nrm = 0.f;
    for (int j = 0; j < C; ++j) {
        const float sj = sigmoid_torch_cpu(l[j]);
        nrm += (norm_p == 1.f) ? sj : (norm_p == 2.f ? sj * sj : powf(sj, norm_p));
    }
    if (norm_p == 2.f) nrm = sqrtf(nrm);
    else if (norm_p != 1.f) nrm = powf(nrm, 1.f / norm_p);
    return __fdiv_rn(sigmoid_torch_cpu(l[c]), fmaxf(nrm, 1e-12f));
}


// ------------------------------------------------------------------------------------------------
// gt centres from pseudo boxes: (x1+x2)/2, (y1+y2)/2   (cpr_head.py:1293-1301)
__global__ void box_centers_kernel(const float* __restrict__ boxes, float* __restrict__ ctr, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        ctr[i * 2 + 0] = __fadd_rn(boxes[i * 4 + 0], boxes[i * 4 + 2]) * 0.5f;
        ctr[i * 2 + 1] = __fadd_rn(boxes[i * 4 + 1], boxes[i * 4 + 3]) * 0.5f;
    }
}
extern "C" int cpr_box_centers(const float* boxes, float* centers, int n, hipStream_t stream) {
    CPR_CHECK_ARG(n >= 0);
    if (n == 0) return CPR_OK;
    CPR_CHECK_ARG(boxes && centers);
    hipLaunchKernelGGL(box_centers_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, boxes, centers, n);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// Negative (whole-grid) branch: OutCirclePtFeatGenerator mask + sigmoid + gfocal(q=0) partial sums
// (cpr_head.py:254-290, 1219-1228; multi_instance_learning_loss.py:148-151).
// One thread per (pixel, class); the image's gts sit in LDS.  mask layout (N*H*W, C) uint8.
__global__ void neg_mask_loss_kernel(const float* __restrict__ logit, int J, const float* __restrict__ ctr,
                                     const int* __restrict__ labels, const int* __restrict__ gt_start,
                                     const int* __restrict__ pad_hw, unsigned char* __restrict__ mask,
                                     double* __restrict__ partial, int H, int W, int C, float stride,
                                     float d2_thr, float eps, int class_wise, int ptype, float norm_p, int Cm) {
    __shared__ float sx[MAX_GT_LDS], sy[MAX_GT_LDS], sn[MAX_GT_LDS];
    __shared__ int sl[MAX_GT_LDS];
    __shared__ double red[4];
    const int n = blockIdx.y;
    const int g0 = gt_start[n], g1 = gt_start[n + 1];
    const int HW = H * W;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // pixel*C + class
    const bool live = idx < HW * C;
    const int pix = live ? idx / C : 0, cls = live ? idx - pix * C : 0;
    const int y = pix / W, x = pix - y * W;
    const float px = (float)x * stride + stride * 0.5f, py = (float)y * stride + stride * 0.5f;
    const float pn = sq_norm(px, py);
    float dmin = INFINITY;
    for (int base = g0; base < g1; base += MAX_GT_LDS) {
        const int cnt = min(MAX_GT_LDS, g1 - base);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const float cx = ctr[(base + i) * 2], cy = ctr[(base + i) * 2 + 1];
            sx[i] = cx; sy[i] = cy; sn[i] = sq_norm(cx, cy); sl[i] = labels[base + i];
        }
        __syncthreads();
        for (int i = 0; i < cnt; ++i) {
            // Cm < C (normal_cfg.out_bg_cls, cpr_head.py:953): the generator's (.., Cm) validity broadcasts over the C outputs
            if (!class_wise || sl[i] == (cls < Cm ? cls : Cm - 1)) {
                float d2 = d2_chain(px, py, pn, sx[i], sy[i], sn[i]);
                d2 = fmaxf(d2, 0.f);  // clamp_min_(0) before the sqrt
                dmin = fminf(dmin, d2);
            }
        }
    }
    const float ph = (float)pad_hw[n * 2], pw = (float)pad_hw[n * 2 + 1];
    const bool valid = live && (0.f <= px) && (px < pw) && (0.f <= py) && (py < ph) && (dmin >= d2_thr);
    double term = 0.0;
    if (live) {
        mask[((size_t)n * HW + pix) * C + cls] = valid ? 1 : 0;
        if (valid) {
            const float* lp = logit + ((size_t)n * HW + pix) * J;
            const float p = (ptype == CPR_PROB_SIGMOID) ? sigmoidf_(lp[cls]) : cls_prob(lp, C, cls, ptype, norm_p);
            term = (double)(-(p * p) * logf(1.f - p + eps));
        }
    }
    term = wave_sum_d(term);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = term;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
        partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
    }
}

extern "C" int cpr_neg_mask_loss(const float* logit, int J, const float* centers, const int* labels,
                                 const int* gt_start, const int* pad_hw, unsigned char* mask, double* partial,
                                 int N, int H, int W, int C, float stride, float d2_thr, float eps, int class_wise,
                                 int prob_type, float norm_p, int mask_classes, int* n_partial, hipStream_t stream) {
    CPR_CHECK_ARG(logit && gt_start && pad_hw && mask && partial && N > 0 && H > 0 && W > 0 && C > 0 && J >= C);
    CPR_CHECK_ARG(mask_classes == C || (mask_classes == 1 && C == 2));   // the reference's broadcast only exists for one class
    CPR_CHECK_ARG(prob_type >= 0 && prob_type <= 2 && norm_p > 0.f);
    const int blocks = cdiv(H * W * C, 256);
    if (n_partial) *n_partial = blocks * N;
    hipLaunchKernelGGL(neg_mask_loss_kernel, dim3(blocks, N), dim3(256), 0, stream, logit, J, centers, labels,
                       gt_start, pad_hw, mask, partial, H, W, C, stride, d2_thr, eps, class_wise, prob_type, norm_p,
                       mask_classes);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// Bilinear sample of 4 channels of a J-channel NHWC map at image point (px, py) with the exact coordinate round trip of
// cpr_head.py:73-93,182-199 (pt/stride -> normalise -> grid_sample un-normalise (align_corners=False) -> border clip,
// 4-tap sum in nw,ne,sw,se order with ATen's vectorised weight formulas).
// align != 0: the generators' align_corners=True form (cpr_head.py:81-84): grid = 2 x / (w - 1) - 1, grid_sample(align_corners=
// True, padding_mode='zeros') -- no clip, a tap outside the map contributes nothing.  When the map holds LOGITS (the projected
// map, see project.hip) a dropped tap must still contribute the projection's bias (the reference samples zero FEATURES and applies
// the Linear afterwards): pad[c] * (weight of the dropped taps) is added when pad is given.
__device__ __forceinline__ void sample_point4(const float* __restrict__ base, int H, int W, int J, float px, float py,
                                              float stride, int ch, int nch, float* __restrict__ acc, int align = 0,
                                              const float* __restrict__ pad = nullptr) {
    const float fw = (float)W, fh = (float)H;
    if (align) {
        const float gx = __fsub_rn(__fdiv_rn(2.f * __fdiv_rn(px, stride), fw - 1.f), 1.f);
        const float gy = __fsub_rn(__fdiv_rn(2.f * __fdiv_rn(py, stride), fh - 1.f), 1.f);
        const float ix = __fmul_rn(__fadd_rn(gx, 1.f), (fw - 1.f) * 0.5f), iy = __fmul_rn(__fadd_rn(gy, 1.f), (fh - 1.f) * 0.5f);
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float ww = __fsub_rn(ix, x0f), we = __fsub_rn(1.f, ww), wn_ = __fsub_rn(iy, y0f), ws = __fsub_rn(1.f, wn_);
        const float wt[4] = {__fmul_rn(ws, we), __fmul_rn(ws, ww), __fmul_rn(wn_, we), __fmul_rn(wn_, ww)};   // nw ne sw se
        // (float compares: a point far outside the map must not overflow an int)
        const bool xin0 = (x0f >= 0.f) && (x0f <= fw - 1.f), xin1 = (x0f + 1.f >= 0.f) && (x0f + 1.f <= fw - 1.f);
        const bool yin0 = (y0f >= 0.f) && (y0f <= fh - 1.f), yin1 = (y0f + 1.f >= 0.f) && (y0f + 1.f <= fh - 1.f);
        const bool in[4] = {xin0 && yin0, xin1 && yin0, xin0 && yin1, xin1 && yin1};
        const int x0 = xin0 ? (int)x0f : 0, x1 = xin1 ? (int)x0f + 1 : 0, y0 = yin0 ? (int)y0f : 0, y1 = yin1 ? (int)y0f + 1 : 0;
        const float* tp[4] = {base + ((size_t)y0 * W + x0) * J + ch, base + ((size_t)y0 * W + x1) * J + ch,
                              base + ((size_t)y1 * W + x0) * J + ch, base + ((size_t)y1 * W + x1) * J + ch};
        float wout = 0.f;
        for (int t = 0; t < 4; ++t) wout += in[t] ? 0.f : wt[t];
        for (int c = 0; c < nch; ++c) {
            float v = 0.f;
            for (int t = 0; t < 4; ++t)
                if (in[t]) v = __fadd_rn(v, __fmul_rn(tp[t][c], wt[t]));
            acc[c] = pad ? __fadd_rn(v, __fmul_rn(pad[ch + c], wout)) : v;
        }
        return;
    }
    // pt/stride -> (2x+1)/w - 1 -> ((g+1)*w - 1)/2 -> clip [0, w-1]
    float gx = __fsub_rn(__fdiv_rn(__fadd_rn(2.f * __fdiv_rn(px, stride), 1.f), fw), 1.f);
    float gy = __fsub_rn(__fdiv_rn(__fadd_rn(2.f * __fdiv_rn(py, stride), 1.f), fh), 1.f);
    float ix = __fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), fw), 1.f) * 0.5f;
    float iy = __fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), fh), 1.f) * 0.5f;
    ix = fminf(fw - 1.f, fmaxf(ix, 0.f));
    iy = fminf(fh - 1.f, fmaxf(iy, 0.f));
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    // weights as ATen's vectorised CPU grid sampler forms them: w = x - floor(x), e = 1 - w, n = y - floor(y), s = 1 - n
    const float ww = __fsub_rn(ix, x0f), we = __fsub_rn(1.f, ww), wn_ = __fsub_rn(iy, y0f), ws = __fsub_rn(1.f, wn_);
    const float wnw = __fmul_rn(ws, we), wne = __fmul_rn(ws, ww), wsw = __fmul_rn(wn_, we), wse = __fmul_rn(wn_, ww);
    const bool x1ok = x1 < W, y1ok = y1 < H;
    const float* pnw = base + ((size_t)y0 * W + x0) * J + ch;
    const float* pne = base + ((size_t)y0 * W + (x1ok ? x1 : x0)) * J + ch;
    const float* psw = base + ((size_t)(y1ok ? y1 : y0) * W + x0) * J + ch;
    const float* pse = base + ((size_t)(y1ok ? y1 : y0) * W + (x1ok ? x1 : x0)) * J + ch;
    for (int c = 0; c < nch; ++c) {
        float v = __fmul_rn(pnw[c], wnw);
        if (x1ok) v = __fadd_rn(v, __fmul_rn(pne[c], wne));
        if (y1ok) v = __fadd_rn(v, __fmul_rn(psw[c], wsw));
        if (x1ok && y1ok) v = __fadd_rn(v, __fmul_rn(pse[c], wse));
        acc[c] = v;
    }
}

// Positive bags of CirclePtFeatGenerator (cpr_head.py:441-497): ring points + the point itself LAST, validity against the
// padded image, bilinear samples.  One thread per (point, bag entry, 4-channel group); a "point" is one (gt, refine) pair.
__global__ void bag_sample_kernel(const float* __restrict__ map, int J, const float* __restrict__ ctr,
                                  const int* __restrict__ gt_img, const int* __restrict__ pad_hw,
                                  const float* __restrict__ offs, float* __restrict__ pts,
                                  unsigned char* __restrict__ valid, float* __restrict__ out, int G, int K, int H,
                                  int W, float stride, int align, const float* __restrict__ pad) {
    const int J4 = (J + 3) >> 2;
    const long long total = (long long)G * K * J4;
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j4 = (int)(i % J4);
    const long long gk = i / J4;
    const int k = (int)(gk % K), g = (int)(gk / K);
    const int n = gt_img[g];
    const float cx = ctr[g * 2], cy = ctr[g * 2 + 1];
    // ring offsets then the centre LAST (cpr_head.py:492-496)
    const float px = (k < K - 1) ? __fadd_rn(offs[k * 2], cx) : cx;
    const float py = (k < K - 1) ? __fadd_rn(offs[k * 2 + 1], cy) : cy;
    if (j4 == 0) {
        pts[gk * 2] = px;
        pts[gk * 2 + 1] = py;
        const float ph = (float)pad_hw[n * 2], pw = (float)pad_hw[n * 2 + 1];
        valid[gk] = ((0.f <= px) && (px < pw) && (0.f <= py) && (py < ph)) ? 1 : 0;
    }
    const int ch = j4 * 4;
    const int nch = min(4, J - ch);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    sample_point4(map + (size_t)n * H * W * J, H, W, J, px, py, stride, ch, nch, acc, align, pad);
    float* dst = out + gk * J + ch;
    for (int c = 0; c < nch; ++c) dst[c] = acc[c];
}

extern "C" int cpr_bag_sample(const float* map, int J, const float* centers, const int* gt_img, const int* pad_hw,
                              const float* offsets, float* pts, unsigned char* valid, float* out, int G, int K,
                              int H, int W, float stride, int align_corners, const float* pad_value, hipStream_t stream) {
    CPR_CHECK_ARG(G >= 0 && K > 0 && J > 0 && H > 0 && W > 0 && stride > 0);
    CPR_CHECK_ARG(!align_corners || (H > 1 && W > 1));
    if (G == 0) return CPR_OK;
    CPR_CHECK_ARG(map && centers && gt_img && pad_hw && pts && valid && out && (K == 1 || offsets));
    const long long total = (long long)G * K * ((J + 3) / 4);
    hipLaunchKernelGGL(bag_sample_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, stream, map, J, centers,
                       gt_img, pad_hw, offsets, pts, valid, out, G, K, H, W, stride, align_corners, pad_value);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// Positive bags of GridCirclesPtFeatGenerator (cpr_head.py:296-352,405-438): the bag of a gt is every grid (anchor)
// point within radius*stride of ANY of its R refine points, in row-major grid order, zero-padded to Kmax entries, followed
// by the R refine points themselves in REVERSED order (so the annotated point is last).  Grid entries take the map value
// of their cell (no interpolation), the refine points are bilinear samples, padding slots take pad_value[J] (zero
// features in the reference = the bias of the projection when the map holds logits) or zeros when it is NULL.  Distances are torch.norm(p=2, dim=-1) of the
// fp32 difference = IEEE sqrt(fma(dy, dy, dx*dx)) (verified bit for bit against torch on the build host).
// Kernel 1 (one wave per gt) selects: pts (G,Kt,2), valid (G,Kt), cell (G,Kt) = y*W+x | -1 padding | -2-r refine point r;
// count[g] = number of grid points found (the reference raises when it exceeds Kmax; the host checks).
__global__ void grid_select_kernel(const float* __restrict__ ctr, const int* __restrict__ gt_img, int R, int Kmax,
                                   float* __restrict__ pts, unsigned char* __restrict__ valid, int* __restrict__ cell,
                                   int* __restrict__ count, int G, int H, int W, float stride, float thr) {
    const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= G) return;
    const int Kt = Kmax + R;
    float lox = INFINITY, hix = -INFINITY, loy = INFINITY, hiy = -INFINITY;
    for (int r = 0; r < R; ++r) {
        const float cx = ctr[((size_t)g * R + r) * 2], cy = ctr[((size_t)g * R + r) * 2 + 1];
        lox = fminf(lox, cx); hix = fmaxf(hix, cx); loy = fminf(loy, cy); hiy = fmaxf(hiy, cy);
    }
    // conservative cell range (one extra cell each side); the exact test below decides
    const int x0 = max(0, (int)floorf((lox - thr) / stride) - 1), x1 = min(W - 1, (int)ceilf((hix + thr) / stride) + 1);
    const int y0 = max(0, (int)floorf((loy - thr) / stride) - 1), y1 = min(H - 1, (int)ceilf((hiy + thr) / stride) + 1);
    int n = 0;
    for (int y = y0; y <= y1; ++y) {
        const float py = (float)y * stride + stride * 0.5f;
        for (int xb = x0; xb <= x1; xb += 64) {
            const int x = xb + lane;
            const float px = (float)x * stride + stride * 0.5f;
            bool in = false;
            if (x <= x1) {
                for (int r = 0; r < R; ++r) {
                    const float dx = __fsub_rn(px, ctr[((size_t)g * R + r) * 2]);
                    const float dy = __fsub_rn(py, ctr[((size_t)g * R + r) * 2 + 1]);
                    in = in || (__fsqrt_rn(__fmaf_rn(dy, dy, __fmul_rn(dx, dx))) <= thr);
                }
            }
            const unsigned long long bal = __ballot(in);
            const int slot = n + __popcll(bal & ((1ull << lane) - 1ull));
            if (in && slot < Kmax) {
                const size_t e = (size_t)g * Kt + slot;
                pts[e * 2] = px; pts[e * 2 + 1] = py; valid[e] = 1; cell[e] = y * W + x;
            }
            n += __popcll(bal);
        }
    }
    for (int k = min(n, Kmax) + lane; k < Kmax; k += 64) {   // padding: zero point, invalid
        const size_t e = (size_t)g * Kt + k;
        pts[e * 2] = 0.f; pts[e * 2 + 1] = 0.f; valid[e] = 0; cell[e] = -1;
    }
    for (int j = lane; j < R; j += 64) {                       // the refine points, reversed (:343-349)
        const int r = R - 1 - j;
        const size_t e = (size_t)g * Kt + Kmax + j;
        pts[e * 2] = ctr[((size_t)g * R + r) * 2]; pts[e * 2 + 1] = ctr[((size_t)g * R + r) * 2 + 1];
        valid[e] = 1; cell[e] = -2 - r;
    }
    if (lane == 0) count[g] = n;
}
// Kernel 2: one thread per (gt, entry, 4-channel group) gathers / samples the map.
__global__ void grid_gather_kernel(const float* __restrict__ map, int J, const float* __restrict__ pts,
                                   const int* __restrict__ cell, const int* __restrict__ gt_img,
                                   const float* __restrict__ pad_value, float* __restrict__ out, long long total,
                                   int Kt, int H, int W, float stride, int align) {
    const int J4 = (J + 3) >> 2;
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j4 = (int)(i % J4);
    const long long e = i / J4;
    const int g = (int)(e / Kt);
    const int ch = j4 * 4, nch = min(4, J - ch);
    const float* base = map + (size_t)gt_img[g] * H * W * J;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int c = cell[e];
    if (c >= 0) {
        for (int q = 0; q < nch; ++q) acc[q] = base[(size_t)c * J + ch + q];
    } else if (c <= -2) {
        sample_point4(base, H, W, J, pts[e * 2], pts[e * 2 + 1], stride, ch, nch, acc, align, pad_value);
    } else if (pad_value) {   // padding slots hold zero FEATURES in the reference: on a projected map that is the bias
        for (int q = 0; q < nch; ++q) acc[q] = pad_value[ch + q];
    }
    for (int q = 0; q < nch; ++q) out[e * J + ch + q] = acc[q];
}

extern "C" int cpr_grid_bag(const float* map, int J, const float* points, const int* gt_img, int R, int Kmax,
                            float radius_px, const float* pad_value, float* pts, unsigned char* valid, int* ws_cell,
                            int* count, float* out, int G, int H, int W, float stride, int align_corners,
                            hipStream_t stream) {
    CPR_CHECK_ARG(G >= 0 && R > 0 && Kmax > 0 && J > 0 && H > 0 && W > 0 && stride > 0 && radius_px >= 0);
    CPR_CHECK_ARG(!align_corners || (H > 1 && W > 1));
    if (G == 0) return CPR_OK;
    CPR_CHECK_ARG(map && points && gt_img && pts && valid && ws_cell && count && out);
    hipLaunchKernelGGL(grid_select_kernel, dim3(cdiv(G, 4)), dim3(256), 0, stream, points, gt_img, R, Kmax, pts, valid,
                       ws_cell, count, G, H, W, stride, radius_px);
    const long long total = (long long)G * (Kmax + R) * ((J + 3) / 4);
    hipLaunchKernelGGL(grid_gather_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, stream, map, J, pts,
                       ws_cell, gt_img, pad_value, out, total, Kmax + R, H, W, stride, align_corners);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// MIL bag loss + gt (annotated-point) loss, one wave per bag (multi_instance_learning_loss.py:153-203 MILLoss,
// :207-243 AllPosLoss; cpr_head.py:1159-1217).  logits[entry][J]: cls = [0,C), ins = [ins_off, ins_off + C*(1+binary)).
// Bag geometry (in entries) covers the three refine_bag_policy values and the grid generators without re-packing:
//   bag b = entries [b*bag_stride + bag_off, +bag_len)
//     independent_with_gt_bag : one bag per (gt, refine) sub-bag     stride K,   off 0,    len K
//     merge_to_gt_bag         : one bag per gt                       stride R*K, off 0,    len R*K
//     only_refine_bag         : refine points 1.. of each gt         stride R*K, off si*K, len (R-si)*K
//   annotated-point ("gt") loss entries of bag b: b*bag_stride + ctr_off + j*ctr_stride, j < ctr_count, taken only when
//   b % ctr_mod == 0 (gt_loss_type 'gt' with independent bags: refine 0 only).
// Per-bag outputs (reduced deterministically by loss_finalize): bag[b] = {mil_loss, gt_loss, has_valid | #valid entries
// (AllPosLoss), #gt-valid, correct | #correct entries (AllPosLoss)}.
#define CPR_PROB_IDENTITY 3   // the cls channels already hold probabilities (MILLoss.forward's reference signature)
__device__ __forceinline__ float bag_prob(const float* __restrict__ l, int C, int c, int ptype, float norm_p) {
    if (ptype == CPR_PROB_SIGMOID) return sigmoidf_(l[c]);
    if (ptype == CPR_PROB_IDENTITY) return l[c];
    return cls_prob(l, C, c, ptype, norm_p);
}
__device__ __forceinline__ float gfocal_term(float p, float q, float eps) {
    const float l1 = (p - q) * (p - q);
    const float l2 = q * logf(p + eps) + (1.f - q) * logf(1.f - p + eps);
    return -(l1 * l2);
}
__global__ void mil_bag_kernel(const float* __restrict__ logits, int J, int ins_off,
                               const unsigned char* __restrict__ valid, const int* __restrict__ labels,
                               const float* __restrict__ gt_weight, float* __restrict__ bag, int G, int bag_stride,
                               int bag_off, int K, int ctr_off, int ctr_stride, int ctr_count, int ctr_mod, int C,
                               float eps, int ptype, float norm_p, int binary_ins, int allpos) {
    const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= G) return;
    const size_t full = (size_t)g * bag_stride;
    const float* L = logits + (full + bag_off) * J;
    const unsigned char* V = valid + full + bag_off;
    const int label = labels[g];
    const float wg = gt_weight ? gt_weight[g] : 1.f;
    // annotated-point loss (cpr_head.py:1159-1184): gfocal on the class probabilities of the gt entries
    float gloss = 0.f, gcount = 0.f;
    if (ctr_count > 0 && (g % ctr_mod) == 0) {
        for (int j = 0; j < ctr_count; ++j) {
            const size_t e = full + ctr_off + (size_t)j * ctr_stride;
            const float gtv = valid[e] ? wg : 0.f;                      // gt_weights_rep = gt_valid * gt_weights
            for (int c = 0; c < C; ++c)
                gloss += gfocal_term(bag_prob(logits + e * J, C, c, ptype, norm_p), (c == label) ? 1.f : 0.f, eps) * gtv;
            gcount += gtv > 0.f ? 1.f : 0.f;
        }
    }
    if (allpos) {   // AllPosLoss: every bag entry is a positive sample of the gt's class
        float loss = 0.f, ns = 0.f, nc = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float w = V[k] ? wg : 0.f;
            float bm = -INFINITY;
            int bc = 0;
            for (int c = 0; c < C; ++c) {
                const float p = bag_prob(L + (size_t)k * J, C, c, ptype, norm_p);
                if (p > bm) { bm = p; bc = c; }
                loss += gfocal_term(p, (c == label) ? 1.f : 0.f, eps) * w;
            }
            ns += w > 0.f ? 1.f : 0.f;
            nc += (bc == label) ? 1.f : 0.f;
        }
        loss = wave_sum(loss); ns = wave_sum(ns); nc = wave_sum(nc);
        if (lane == 0) {
            float* o = bag + (size_t)g * 5;
            o[0] = loss; o[1] = gloss; o[2] = ns; o[3] = gcount; o[4] = nc;
        }
        return;
    }
    float nvalid = 0.f;
    for (int k = lane; k < K; k += 64) nvalid += V[k] ? 1.f : 0.f;
    nvalid = wave_sum(nvalid);
    const float lw = (nvalid * wg > 0.f) ? 1.f : 0.f;  // label_weights = (valid.sum(dim=1) > 0)
    float loss = 0.f, best = -INFINITY;
    int best_c = 0;
    const int nj = binary_ins ? 2 : 1;
    for (int c = 0; c < C; ++c) {
        for (int j = 0; j < nj; ++j) {
            const int ic = ins_off + c * nj + j;   // bag_ins_outs.reshape(B, N, C, -1)
            float m = -INFINITY;
            for (int k = lane; k < K; k += 64) m = fmaxf(m, L[(size_t)k * J + ic]);
            m = wave_max(m);
            float se = 0.f;
            for (int k = lane; k < K; k += 64) se += expf(L[(size_t)k * J + ic] - m);
            se = wave_sum(se);
            float sv = 0.f, sp = 0.f;  // sum of valid-masked softmax, and of prob_cls * it
            for (int k = lane; k < K; k += 64) {
                const float pi = expf(L[(size_t)k * J + ic] - m) / se * (V[k] ? wg : 0.f);
                sv += pi;   // pi >= 0 so the L1 norm is the plain sum
                sp += bag_prob(L + (size_t)k * J, C, c, ptype, norm_p) * pi;
            }
            sv = wave_sum(sv);
            sp = wave_sum(sp);
            const float p = sp / fmaxf(sv, 1e-12f);  // F.normalize(p=1, eps=1e-12) then the weighted sum
            if (j == 0 && p > best) { best = p; best_c = c; }
            // binary_ins: the second instance branch is a "negative bag" -- its labels are all zero (:179-184)
            loss += gfocal_term(p, (j == 0 && c == label) ? 1.f : 0.f, eps) * lw;
        }
    }
    if (lane == 0) {
        float* o = bag + (size_t)g * 5;
        o[0] = loss; o[1] = gloss; o[2] = lw; o[3] = gcount; o[4] = (best_c == label) ? 1.f : 0.f;
    }
}

// The same bag loss with the classes of a bag spread over the waves of a workgroup (round 4; C >= 8, e.g. the 80-class COCO
// form of BASELINE.json configs[2]): mil_bag_kernel walks the classes one after the other, three strided passes over the bag per
// class -- 391 us for 256 bags x 80 classes (2.4 % of the configs[2] step) on 256 waves.  Here workgroup = one bag, wave w takes
// classes w, w + NW, ...; every per-class quantity is computed by the SAME code (lanes over the bag's points, the same wave
// reductions), parked in LDS, and lane 0 of wave 0 then adds the terms and scans for the best class in ascending class order --
// the order mil_bag_kernel uses: bit-identical outputs.
constexpr int MIL_NW = 8, MIL_MAXT = 512;
__global__ __launch_bounds__(64 * MIL_NW) void mil_bag_cls_kernel(
    const float* __restrict__ logits, int J, int ins_off, const unsigned char* __restrict__ valid, const int* __restrict__ labels,
    const float* __restrict__ gt_weight, float* __restrict__ bag, int G, int bag_stride, int bag_off, int K, int ctr_off,
    int ctr_stride, int ctr_count, int ctr_mod, int C, float eps, int ptype, float norm_p, int binary_ins) {
    __shared__ float s_p[MIL_MAXT];          // p of (class, branch)
    const int g = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t full = (size_t)g * bag_stride;
    const float* L = logits + (full + bag_off) * J;
    const unsigned char* V = valid + full + bag_off;
    const int label = labels[g];
    const float wg = gt_weight ? gt_weight[g] : 1.f;
    float nvalid = 0.f;
    for (int k = lane; k < K; k += 64) nvalid += V[k] ? 1.f : 0.f;
    nvalid = wave_sum(nvalid);
    const float lw = (nvalid * wg > 0.f) ? 1.f : 0.f;
    const int nj = binary_ins ? 2 : 1;
    for (int c = wave; c < C; c += MIL_NW) {
        for (int j = 0; j < nj; ++j) {
            const int ic = ins_off + c * nj + j;
            float m = -INFINITY;
            for (int k = lane; k < K; k += 64) m = fmaxf(m, L[(size_t)k * J + ic]);
            m = wave_max(m);
            float se = 0.f;
            for (int k = lane; k < K; k += 64) se += expf(L[(size_t)k * J + ic] - m);
            se = wave_sum(se);
            float sv = 0.f, sp = 0.f;
            for (int k = lane; k < K; k += 64) {
                const float pi = expf(L[(size_t)k * J + ic] - m) / se * (V[k] ? wg : 0.f);
                sv += pi;
                sp += bag_prob(L + (size_t)k * J, C, c, ptype, norm_p) * pi;
            }
            sv = wave_sum(sv);
            sp = wave_sum(sp);
            if (lane == 0) s_p[c * nj + j] = sp / fmaxf(sv, 1e-12f);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float gloss = 0.f, gcount = 0.f;
        if (ctr_count > 0 && (g % ctr_mod) == 0) {
            for (int j = 0; j < ctr_count; ++j) {
                const size_t e = full + ctr_off + (size_t)j * ctr_stride;
                const float gtv = valid[e] ? wg : 0.f;
                for (int c = 0; c < C; ++c)
                    gloss += gfocal_term(bag_prob(logits + e * J, C, c, ptype, norm_p), (c == label) ? 1.f : 0.f, eps) * gtv;
                gcount += gtv > 0.f ? 1.f : 0.f;
            }
        }
        float loss = 0.f, best = -INFINITY;
        int best_c = 0;
        for (int c = 0; c < C; ++c)
            for (int j = 0; j < nj; ++j) {
                const float pv = s_p[c * nj + j];
                if (j == 0 && pv > best) { best = pv; best_c = c; }
                loss += gfocal_term(pv, (j == 0 && c == label) ? 1.f : 0.f, eps) * lw;
            }
        float* o = bag + (size_t)g * 5;
        o[0] = loss; o[1] = gloss; o[2] = lw; o[3] = gcount; o[4] = (best_c == label) ? 1.f : 0.f;
    }
}

// out[0..4] = {gt_loss, pos_loss, bag_acc, neg_loss, num_sample}  (cpr_head.py:1180-1184,1216-1228).  The negative loss
// is averaged over the LAST num_pos the reference computed: the MIL num_sample, or the gt count when with_mil_loss is off.
__global__ void loss_finalize_kernel(const float* __restrict__ bag, int G, const double* __restrict__ neg_partial,
                                     int n_partial, float w_mil, float w_gt, float w_neg, float acc_den,
                                     int neg_from_gt, float* __restrict__ out) {
    __shared__ double sh[6][4];
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (int g = threadIdx.x; g < G; g += blockDim.x)
        for (int k = 0; k < 5; ++k) a[k] += (double)bag[(size_t)g * 5 + k];
    for (int i = threadIdx.x; i < n_partial; i += blockDim.x) a[5] += neg_partial[i];
    for (int k = 0; k < 6; ++k) {
        const double s = wave_sum_d(a[k]);
        if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[6];
        for (int k = 0; k < 6; ++k) t[k] = sh[k][0] + sh[k][1] + sh[k][2] + sh[k][3];
        const double num_sample = t[2] > 1.0 ? t[2] : 1.0;
        const double num_pos_gt = t[3] > 1.0 ? t[3] : 1.0;
        out[0] = (float)(w_gt * (t[1] / num_pos_gt));
        out[1] = (float)(t[0] / num_sample * w_mil);
        out[2] = (float)(acc_den > 0.f ? t[4] * 100.0 / acc_den : 0.0);
        out[3] = (float)(w_neg * (t[5] / (neg_from_gt ? num_pos_gt : num_sample)));
        out[4] = (float)num_sample;
    }
}

extern "C" int cpr_mil_loss(const float* logits, int J, int ins_off, const unsigned char* valid, const int* labels,
                            const float* gt_weight, float* bag_ws, const double* neg_partial, int n_partial,
                            int num_bags, int bag_stride, int bag_off, int bag_len, int ctr_off, int ctr_stride,
                            int ctr_count, int ctr_mod, int C, float eps, int prob_type, float norm_p, int binary_ins,
                            int allpos, float w_mil, float w_gt, float w_neg, int neg_from_gt, float* out5,
                            hipStream_t stream) {
    const int G = num_bags, K = bag_len;
    CPR_CHECK_ARG(G > 0 && K > 0 && C > 0 && bag_off >= 0 && bag_stride >= bag_off + K && ctr_count >= 0 && ctr_mod >= 1);
    CPR_CHECK_ARG(J >= ins_off + C * (binary_ins ? 2 : 1) && logits && valid && labels && bag_ws && out5);
    CPR_CHECK_ARG(ctr_count == 0 || (ctr_off >= 0 && ctr_off + (ctr_count - 1) * ctr_stride < bag_stride));
    CPR_CHECK_ARG(prob_type >= 0 && prob_type <= 3 && norm_p > 0.f && (n_partial == 0 || neg_partial));
    const int terms = C * (binary_ins ? 2 : 1);
    if (!allpos && terms >= MIL_NW && terms <= MIL_MAXT)       // many classes: one workgroup per bag, classes over its waves
        hipLaunchKernelGGL(mil_bag_cls_kernel, dim3(G), dim3(64 * MIL_NW), 0, stream, logits, J, ins_off, valid, labels, gt_weight,
                           bag_ws, G, bag_stride, bag_off, K, ctr_off, ctr_stride, ctr_count, ctr_mod, C, eps, prob_type, norm_p,
                           binary_ins);
    else
        hipLaunchKernelGGL(mil_bag_kernel, dim3(cdiv(G, 4)), dim3(256), 0, stream, logits, J, ins_off, valid, labels,
                           gt_weight, bag_ws, G, bag_stride, bag_off, K, ctr_off, ctr_stride, ctr_count, ctr_mod, C, eps,
                           prob_type, norm_p, binary_ins, allpos);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, bag_ws, G, neg_partial, n_partial, w_mil,
                       w_gt, w_neg, allpos ? (float)G * (float)K : (float)G, neg_from_gt, out5);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// PointRefiner.refine_single (cpr_head.py:780-850), one wave per gt.  A gt owns Kt = Rv*Kv bag entries: Rv sub-bags
// (one per refine point for the circle generator, Rv = 1 for the grid generators) of Kv entries each; entry Kv-1 is the
// annotated (refine-0) point itself.  Filters: nearest point among the class-mates' Rv points each (class-wise cdist +
// first arg-min, :711-743 -- only when the class has more than one gt), classify filter (arg-max class, :745-756),
// probability thresholds (:823), inside-image (:773-778); then the probability-weighted merge (:830-833), mean score
// (:835) and the fall-back to the annotated point.
// Distances are the cdist values themselves: the sgemm chain of d2_chain(), clamp_min(0), IEEE sqrt; equal distances
// keep the first candidate in (gt, refine) order like torch.min(dim).  (torch's CPU sqrt is MKL VML and differs from
// IEEE in the last bit for ~0.65 % of arguments, host dependent, so a tie between two candidates whose squared
// distances are 1-2 ulp apart can still fall differently.)
__global__ void refine_kernel(const float* __restrict__ logits, int J, const float* __restrict__ pts,
                              const unsigned char* __restrict__ valid, const float* __restrict__ ctr, int Rv,
                              int ctr_stride, const int* __restrict__ labels, const int* __restrict__ gt_img,
                              const int* __restrict__ gt_start, const int* __restrict__ img_hw,
                              const unsigned char* __restrict__ not_refine_in, float* __restrict__ refine_pts,
                              float* __restrict__ scores, unsigned char* __restrict__ not_refine,
                              unsigned char* __restrict__ chosen, int G, int Kt, int Kv, int C, int ptype,
                              float norm_p, float gt_alpha, float merge_th, float refine_th, int use_nearest,
                              int use_classify, int score_max) {
    const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= G) return;
    const int n = gt_img[g], label = labels[g];
    const int g0 = gt_start[n], g1 = gt_start[n + 1];
    const float* L = logits + (size_t)g * Kt * J;
    const float gt_prob = cls_prob(L + (size_t)(Kv - 1) * J, C, label, ptype, norm_p);
    const float gate = __fmul_rn(gt_prob, gt_alpha);
    const float ih = (float)img_hw[n * 2], iw = (float)img_hw[n * 2 + 1];
    int same = 0;
    for (int o = g0; o < g1; ++o) same += (labels[o] == label) ? 1 : 0;
    float sw = 0.f, sx = 0.f, sy = 0.f, cnt = 0.f, pmax = 0.f;
    // pass 1: decide membership, accumulate sum of probabilities
    for (int k = lane; k < Kt; k += 64) {
        const float px = pts[((size_t)g * Kt + k) * 2], py = pts[((size_t)g * Kt + k) * 2 + 1];
        bool ok = valid[(size_t)g * Kt + k] != 0;
        if (use_nearest && same > 1) {
            const float pn = sq_norm(px, py);
            float best = INFINITY;
            int bo = -1, br = -1;
            for (int o = g0; o < g1; ++o) {  // ascending (gt, refine) order + strict '<' = first minimum, as torch.min
                if (labels[o] != label) continue;
                for (int r = 0; r < Rv; ++r) {
                    const float cx = ctr[((size_t)o * ctr_stride + r) * 2], cy = ctr[((size_t)o * ctr_stride + r) * 2 + 1];
                    const float d = __fsqrt_rn(fmaxf(d2_chain(px, py, pn, cx, cy, sq_norm(cx, cy)), 0.f));
                    if (d < best) { best = d; bo = o; br = r; }
                }
            }
            ok = ok && (bo == g) && (br == k / Kv);
        }
        const float* lk = L + (size_t)k * J;
        const float p = cls_prob(lk, C, label, ptype, norm_p);
        if (use_classify) {   // bag_cls_prob.max(dim=-1): first maximum of the PROBABILITIES
            float bm = -INFINITY;
            int bc = 0;
            for (int c = 0; c < C; ++c) {
                const float v = (c == label) ? p : cls_prob(lk, C, c, ptype, norm_p);
                if (v > bm) { bm = v; bc = c; }
            }
            ok = ok && (bc == label);
        }
        ok = ok && (p > merge_th) && (p > gate);
        ok = ok && (px < iw) && (px >= 0.f) && (py < ih) && (py >= 0.f);
        const float pm = ok ? p : 0.f;
        chosen[(size_t)g * Kt + k] = (pm > 0.f) ? 1 : 0;
        sw += pm;
        pmax = fmaxf(pmax, pm);
        cnt += (pm > 0.f) ? 1.f : 0.f;
    }
    sw = wave_sum(sw);
    pmax = wave_max(pmax);
    cnt = wave_sum(cnt);
    const float denom = sw + 1e-8f;
    for (int k = lane; k < Kt; k += 64) {
        if (chosen[(size_t)g * Kt + k]) {
            const float p = cls_prob(L + (size_t)k * J, C, label, ptype, norm_p);
            const float w = p / denom;
            sx += pts[((size_t)g * Kt + k) * 2] * w;
            sy += pts[((size_t)g * Kt + k) * 2 + 1] * w;
        }
    }
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    if (lane == 0) {
        const float score = sw / (cnt + 1e-8f);
        bool nr = score < refine_th;
        if (not_refine_in) nr = nr || (not_refine_in[g] != 0);
        refine_pts[g * 2] = nr ? ctr[(size_t)g * ctr_stride * 2] : sx;
        refine_pts[g * 2 + 1] = nr ? ctr[(size_t)g * ctr_stride * 2 + 1] : sy;
        // return_score_type 'max' (cpr_head.py:840-842): the largest kept probability, refine_th / 2 when nothing was kept;
        // the not_refine decision above always uses the mean
        scores[g] = score_max ? (pmax == 0.f ? refine_th * 0.5f : pmax) : score;
        not_refine[g] = nr ? 1 : 0;
    }
}

extern "C" int cpr_refine(const float* logits, int J, const float* pts, const unsigned char* valid,
                          const float* centers, int Rv, int ctr_stride, const int* labels, const int* gt_img,
                          const int* gt_start, const int* img_hw, const unsigned char* not_refine_in,
                          float* refine_pts, float* scores, unsigned char* not_refine, unsigned char* chosen, int G,
                          int Kt, int Kv, int C, int prob_type, float norm_p, float gt_alpha, float merge_th,
                          float refine_th, int use_nearest, int use_classify, int score_max, hipStream_t stream) {
    CPR_CHECK_ARG(G > 0 && Kt > 0 && Kv > 0 && Rv > 0 && Kt == Rv * Kv && ctr_stride >= Rv && C > 0 && J >= C);
    CPR_CHECK_ARG(prob_type >= 0 && prob_type <= 2 && norm_p > 0.f);
    CPR_CHECK_ARG(logits && pts && valid && centers && labels && gt_img && gt_start && img_hw && refine_pts &&
                  scores && not_refine && chosen);
    hipLaunchKernelGGL(refine_kernel, dim3(cdiv(G, 4)), dim3(256), 0, stream, logits, J, pts, valid, centers, Rv,
                       ctr_stride, labels, gt_img, gt_start, img_hw, not_refine_in, refine_pts, scores, not_refine,
                       chosen, G, Kt, Kv, C, prob_type, norm_p, gt_alpha, merge_th, refine_th, use_nearest, use_classify,
                       score_max);
    CPR_LAUNCH_STATUS();
}
