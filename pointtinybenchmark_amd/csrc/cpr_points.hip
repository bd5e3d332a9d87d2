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

__device__ __forceinline__ float sq_norm(float x, float y) {
    // x.pow(2).sum(-1): each square rounded, then one add (cpr_head.py:277 via torch.cdist)
    return __fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y));
}
__device__ __forceinline__ float d2_chain(float px, float py, float pn, float cx, float cy, float cn) {
    float acc = __fmul_rn(-2.f * px, cx);
    acc = __fmaf_rn(-2.f * py, cy, acc);
    acc = __fadd_rn(pn, acc);
    acc = __fadd_rn(acc, cn);
    return acc;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

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
                                     float d2_thr, float eps, int class_wise) {
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
            if (!class_wise || sl[i] == cls) {
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
            const float p = sigmoidf_(logit[((size_t)n * HW + pix) * J + cls]);
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
                                 int* n_partial, hipStream_t stream) {
    CPR_CHECK_ARG(logit && gt_start && pad_hw && mask && partial && N > 0 && H > 0 && W > 0 && C > 0 && J >= C);
    const int blocks = cdiv(H * W * C, 256);
    if (n_partial) *n_partial = blocks * N;
    hipLaunchKernelGGL(neg_mask_loss_kernel, dim3(blocks, N), dim3(256), 0, stream, logit, J, centers, labels,
                       gt_start, pad_hw, mask, partial, H, W, C, stride, d2_thr, eps, class_wise);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// Positive bags: CirclePtFeatGenerator points + validity + bilinear sampling of a J-channel NHWC map with
// the exact coordinate round trip of cpr_head.py:73-93,182-199 (normalise, grid_sample un-normalise,
// border clip, 4-tap sum in nw,ne,sw,se order).  One thread per (gt, bag point, 4-channel group).
__global__ void bag_sample_kernel(const float* __restrict__ map, int J, const float* __restrict__ ctr,
                                  const int* __restrict__ gt_img, const int* __restrict__ pad_hw,
                                  const float* __restrict__ offs, float* __restrict__ pts,
                                  unsigned char* __restrict__ valid, float* __restrict__ out, int G, int K, int H,
                                  int W, float stride) {
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
    // pt/stride -> (2x+1)/w - 1 -> ((g+1)*w - 1)/2 -> clip [0, w-1]
    const float fw = (float)W, fh = (float)H;
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
    const float* base = map + (size_t)n * H * W * J;
    const int ch = j4 * 4;
    const int nch = min(4, J - ch);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
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
    float* dst = out + gk * J + ch;
    for (int c = 0; c < nch; ++c) dst[c] = acc[c];
}

extern "C" int cpr_bag_sample(const float* map, int J, const float* centers, const int* gt_img, const int* pad_hw,
                              const float* offsets, float* pts, unsigned char* valid, float* out, int G, int K,
                              int H, int W, float stride, hipStream_t stream) {
    CPR_CHECK_ARG(G >= 0 && K > 0 && J > 0 && H > 0 && W > 0 && stride > 0);
    if (G == 0) return CPR_OK;
    CPR_CHECK_ARG(map && centers && gt_img && pad_hw && pts && valid && out && (K == 1 || offsets));
    const long long total = (long long)G * K * ((J + 3) / 4);
    hipLaunchKernelGGL(bag_sample_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, stream, map, J, centers,
                       gt_img, pad_hw, offsets, pts, valid, out, G, K, H, W, stride);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// MIL bag loss + gt (centre) loss, one wave per bag (multi_instance_learning_loss.py:153-203,
// cpr_head.py:1159-1184).  logits[g][k][J]: cls = [0,C), ins = [ins_off, ins_off + C).
// Per-bag outputs (reduced deterministically by loss_finalize): bag[g] = {mil_loss, gt_loss, has_valid,
// gt_valid, correct}.
__global__ void mil_bag_kernel(const float* __restrict__ logits, int J, int ins_off,
                               const unsigned char* __restrict__ valid, const int* __restrict__ labels,
                               const float* __restrict__ gt_weight, float* __restrict__ bag, int G, int K, int C,
                               float eps) {
    const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= G) return;
    const float* L = logits + (size_t)g * K * J;
    const unsigned char* V = valid + (size_t)g * K;
    const int label = labels[g];
    const float wg = gt_weight ? gt_weight[g] : 1.f;
    float nvalid = 0.f;
    for (int k = lane; k < K; k += 64) nvalid += V[k] ? 1.f : 0.f;
    nvalid = wave_sum(nvalid);
    const float lw = (nvalid * wg > 0.f) ? 1.f : 0.f;  // label_weights = (valid.sum(dim=1) > 0)
    const float gtv = V[K - 1] ? wg : 0.f;             // gt_weights_rep = gt_valid * gt_weights
    float loss = 0.f, gloss = 0.f, best = -INFINITY;
    int best_c = 0;
    for (int c = 0; c < C; ++c) {
        float m = -INFINITY;
        for (int k = lane; k < K; k += 64) m = fmaxf(m, L[(size_t)k * J + ins_off + c]);
        m = wave_max(m);
        float se = 0.f;
        for (int k = lane; k < K; k += 64) se += expf(L[(size_t)k * J + ins_off + c] - m);
        se = wave_sum(se);
        float sv = 0.f, sp = 0.f;  // sum of valid-masked softmax, and of prob_cls * it
        for (int k = lane; k < K; k += 64) {
            const float pi = expf(L[(size_t)k * J + ins_off + c] - m) / se * (V[k] ? wg : 0.f);
            sv += pi;   // pi >= 0 so the L1 norm is the plain sum
            sp += sigmoidf_(L[(size_t)k * J + c]) * pi;
        }
        sv = wave_sum(sv);
        sp = wave_sum(sp);
        const float p = sp / fmaxf(sv, 1e-12f);  // F.normalize(p=1, eps=1e-12) then the weighted sum
        if (p > best) { best = p; best_c = c; }
        const float q = (c == label) ? 1.f : 0.f;
        const float l1 = (p - q) * (p - q);
        const float l2 = q * logf(p + eps) + (1.f - q) * logf(1.f - p + eps);
        loss += -(l1 * l2 * lw);
        const float gp = sigmoidf_(L[(size_t)(K - 1) * J + c]);
        const float g1 = (gp - q) * (gp - q);
        const float g2 = q * logf(gp + eps) + (1.f - q) * logf(1.f - gp + eps);
        gloss += -(g1 * g2 * gtv);
    }
    if (lane == 0) {
        float* o = bag + (size_t)g * 5;
        o[0] = loss; o[1] = gloss; o[2] = lw; o[3] = gtv > 0.f ? 1.f : 0.f; o[4] = (best_c == label) ? 1.f : 0.f;
    }
}

// out[0..4] = {gt_loss, pos_loss, bag_acc, neg_loss, num_pos}  (cpr_head.py:1180-1184,1216-1228)
__global__ void loss_finalize_kernel(const float* __restrict__ bag, int G, const double* __restrict__ neg_partial,
                                     int n_partial, float w_mil, float w_gt, float w_neg, float* __restrict__ out) {
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
        out[2] = (float)(G > 0 ? t[4] * 100.0 / G : 0.0);
        out[3] = (float)(w_neg * (t[5] / num_sample));
        out[4] = (float)num_sample;
    }
}

extern "C" int cpr_mil_loss(const float* logits, int J, int ins_off, const unsigned char* valid, const int* labels,
                            const float* gt_weight, float* bag_ws, const double* neg_partial, int n_partial, int G,
                            int K, int C, float eps, float w_mil, float w_gt, float w_neg, float* out5,
                            hipStream_t stream) {
    CPR_CHECK_ARG(G > 0 && K > 0 && C > 0 && J >= ins_off + C && logits && valid && labels && bag_ws && out5);
    CPR_CHECK_ARG(n_partial == 0 || neg_partial);
    hipLaunchKernelGGL(mil_bag_kernel, dim3(cdiv(G, 4)), dim3(256), 0, stream, logits, J, ins_off, valid, labels,
                       gt_weight, bag_ws, G, K, C, eps);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, bag_ws, G, neg_partial, n_partial, w_mil,
                       w_gt, w_neg, out5);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// PointRefiner.refine_single (cpr_head.py:780-850), one wave per gt: nearest-gt filter (class-wise cdist +
// argmin, :711-743), classify filter (argmax class, :745-756), probability thresholds (:823), inside-image
// (:773-778), probability-weighted merge (:830-833), mean score (:835), fallback to the annotated point.
__global__ void refine_kernel(const float* __restrict__ logits, int J, const float* __restrict__ pts,
                              const unsigned char* __restrict__ valid, const float* __restrict__ ctr,
                              const int* __restrict__ labels, const int* __restrict__ gt_img,
                              const int* __restrict__ gt_start, const int* __restrict__ img_hw,
                              const unsigned char* __restrict__ not_refine_in, float* __restrict__ refine_pts,
                              float* __restrict__ scores, unsigned char* __restrict__ not_refine,
                              unsigned char* __restrict__ chosen, int G, int K, int C, float gt_alpha,
                              float merge_th, float refine_th, int use_nearest, int use_classify) {
    const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= G) return;
    const int n = gt_img[g], label = labels[g];
    const int g0 = gt_start[n], g1 = gt_start[n + 1];
    const float* L = logits + (size_t)g * K * J;
    const float gt_prob = sigmoidf_(L[(size_t)(K - 1) * J + label]);
    const float ih = (float)img_hw[n * 2], iw = (float)img_hw[n * 2 + 1];
    int same = 0;
    for (int o = g0; o < g1; ++o) same += (labels[o] == label) ? 1 : 0;
    float sw = 0.f, sx = 0.f, sy = 0.f, cnt = 0.f;
    // pass 1: decide membership, accumulate sum of probabilities
    for (int k = lane; k < K; k += 64) {
        const float px = pts[((size_t)g * K + k) * 2], py = pts[((size_t)g * K + k) * 2 + 1];
        bool ok = valid[(size_t)g * K + k] != 0;
        if (use_nearest && same > 1) {
            const float pn = sq_norm(px, py);
            float best = INFINITY;
            int bi = -1;
            for (int o = g0; o < g1; ++o) {  // ascending index + strict '<' = first minimum, as torch.min
                if (labels[o] != label) continue;
                const float cx = ctr[o * 2], cy = ctr[o * 2 + 1];
                const float d2 = fmaxf(d2_chain(px, py, pn, cx, cy, sq_norm(cx, cy)), 0.f);
                if (d2 < best) { best = d2; bi = o; }
            }
            ok = ok && (bi == g);
        }
        const float p = sigmoidf_(L[(size_t)k * J + label]);
        if (use_classify) {
            float bm = -INFINITY;
            int bc = 0;
            for (int c = 0; c < C; ++c) {  // sigmoid is monotone: argmax over logits == argmax over probs
                const float v = L[(size_t)k * J + c];
                if (v > bm) { bm = v; bc = c; }
            }
            ok = ok && (bc == label);
        }
        ok = ok && (p > merge_th) && (p > gt_prob * gt_alpha);
        ok = ok && (px < iw) && (px >= 0.f) && (py < ih) && (py >= 0.f);
        const float pm = ok ? p : 0.f;
        chosen[(size_t)g * K + k] = (pm > 0.f) ? 1 : 0;
        sw += pm;
        cnt += (pm > 0.f) ? 1.f : 0.f;
    }
    sw = wave_sum(sw);
    cnt = wave_sum(cnt);
    const float denom = sw + 1e-8f;
    for (int k = lane; k < K; k += 64) {
        if (chosen[(size_t)g * K + k]) {
            const float p = sigmoidf_(L[(size_t)k * J + label]);
            const float w = p / denom;
            sx += pts[((size_t)g * K + k) * 2] * w;
            sy += pts[((size_t)g * K + k) * 2 + 1] * w;
        }
    }
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    if (lane == 0) {
        const float score = sw / (cnt + 1e-8f);
        bool nr = score < refine_th;
        if (not_refine_in) nr = nr || (not_refine_in[g] != 0);
        refine_pts[g * 2] = nr ? ctr[g * 2] : sx;
        refine_pts[g * 2 + 1] = nr ? ctr[g * 2 + 1] : sy;
        scores[g] = score;
        not_refine[g] = nr ? 1 : 0;
    }
}

extern "C" int cpr_refine(const float* logits, int J, const float* pts, const unsigned char* valid,
                          const float* centers, const int* labels, const int* gt_img, const int* gt_start,
                          const int* img_hw, const unsigned char* not_refine_in, float* refine_pts, float* scores,
                          unsigned char* not_refine, unsigned char* chosen, int G, int K, int C, float gt_alpha,
                          float merge_th, float refine_th, int use_nearest, int use_classify, hipStream_t stream) {
    CPR_CHECK_ARG(G > 0 && K > 0 && C > 0 && J >= C);
    CPR_CHECK_ARG(logits && pts && valid && centers && labels && gt_img && gt_start && img_hw && refine_pts &&
                  scores && not_refine && chosen);
    hipLaunchKernelGGL(refine_kernel, dim3(cdiv(G, 4)), dim3(256), 0, stream, logits, J, pts, valid, centers, labels,
                       gt_img, gt_start, img_hw, not_refine_in, refine_pts, scores, not_refine, chosen, G, K, C,
                       gt_alpha, merge_th, refine_th, use_nearest, use_classify);
    CPR_LAUNCH_STATUS();
}
