// P2P inference post-processing (T/mmdet/models/point/dense_heads/p2p_head.py:345-405):
//   cpr_topk_desc  per-level top-k of the max class score (p2p_head.py:367-373, torch.topk)
//   cpr_nms        mmcv.ops.nms.batched_nms as used by multiclass_nms (bbox_nms.py:85; third-party mmcv-full 1.3.x):
//                  class-offset boxes, score sort, 64x64 bitmask IoU tiles (one wavefront = one 64-bit mask row),
//                  serial keep scan by a single wave.
#include <limits.h>
#include "common.h"

__device__ __forceinline__ unsigned f2key(float f) {  // monotone: larger float -> larger key
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// in-place bitonic sort, DESCENDING, of P (power of two) 64-bit words by one workgroup
__device__ void block_bitonic_desc(unsigned long long* a, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long x = a[i], y = a[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) { a[i] = y; a[l] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// radix-select the k-th largest key (4 x 8-bit passes), gather, bitonic-sort the k survivors in LDS.
__global__ void topk_desc_kernel(const float* __restrict__ scores, int n, int k, float* __restrict__ out_vals,
                                 long long* __restrict__ out_idx) {
    // one workgroup per segment (image): blockIdx.x selects scores[b * n ..], outputs [b * k ..]; indices are segment-relative
    scores += (size_t)blockIdx.x * n;
    out_vals += (size_t)blockIdx.x * k;
    out_idx += (size_t)blockIdx.x * k;
    __shared__ unsigned long long sel[4096];
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_remaining, s_count, s_base, s_wcnt[16];
    const int tid = threadIdx.x;
    if (tid == 0) { s_prefix = 0; s_remaining = (unsigned)k; s_count = 0; s_base = 0; }
    unsigned mask = 0;
    __syncthreads();
    for (int pass = 3; pass >= 0; --pass) {
        const int shift = pass * 8;
        for (int b = tid; b < 256; b += blockDim.x) hist[b] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        for (int i = tid; i < n; i += blockDim.x) {
            const unsigned key = f2key(scores[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned cum = 0, rem = s_remaining;
            for (int b = 255; b >= 0; --b) {
                if (cum + hist[b] >= rem) { s_prefix = prefix | ((unsigned)b << shift); s_remaining = rem - cum; break; }
                cum += hist[b];
            }
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    const unsigned T = s_prefix, take_eq = s_remaining;
    // keys > T: any order (sorted below); keys == T: the `take_eq` lowest indices (ordered scan)
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + tid;
        const unsigned key = i < n ? f2key(scores[i]) : 0u;
        const bool gt = i < n && key > T, eq = i < n && key == T;
        if (gt) sel[atomicAdd(&s_count, 1u)] = ((unsigned long long)key << 32) | (unsigned)(~(unsigned)i);
        const unsigned long long bal = __ballot(eq);
        const int lane = tid & 63, w = tid >> 6;
        if (lane == 0) s_wcnt[w] = __popcll(bal);
        __syncthreads();
        unsigned off = s_base;
        for (int q = 0; q < w; ++q) off += s_wcnt[q];
        const unsigned rank = off + __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();
        if (tid == 0) {
            unsigned tot = 0;
            for (int q = 0; q < (int)(blockDim.x >> 6); ++q) tot += s_wcnt[q];
            s_base += tot;
        }
        if (eq && rank < take_eq) sel[(k - take_eq) + rank] = ((unsigned long long)key << 32) | (unsigned)(~(unsigned)i);
        __syncthreads();
    }
    int P = 1;
    while (P < k) P <<= 1;
    for (int i = k + tid; i < P; i += blockDim.x) sel[i] = 0ull;
    __syncthreads();
    block_bitonic_desc(sel, P);
    for (int i = tid; i < k; i += blockDim.x) {
        out_vals[i] = key2f((unsigned)(sel[i] >> 32));
        out_idx[i] = (long long)(~(unsigned)(sel[i] & 0xffffffffull));
    }
}

extern "C" int cpr_topk_desc(const float* scores, int n, int k, float* out_vals, long long* out_idx,
                             hipStream_t stream) {
    CPR_CHECK_ARG(n >= 0 && k >= 0 && k <= n && k <= 4096);
    if (k == 0) return CPR_OK;
    CPR_CHECK_ARG(scores && out_vals && out_idx);
    hipLaunchKernelGGL(topk_desc_kernel, dim3(1), dim3(1024), 0, stream, scores, n, k, out_vals, out_idx);
    CPR_LAUNCH_STATUS();
}
// B equal-length segments in one launch (the images of a batch): scores (B, n) -> out_vals (B, k), out_idx (B, k) segment-relative
extern "C" int cpr_topk_desc_batched(const float* scores, int B, int n, int k, float* out_vals, long long* out_idx,
                                     hipStream_t stream) {
    CPR_CHECK_ARG(B >= 0 && n >= 0 && k >= 0 && k <= n && k <= 4096);
    if (k == 0 || B == 0) return CPR_OK;
    CPR_CHECK_ARG(scores && out_vals && out_idx);
    hipLaunchKernelGGL(topk_desc_kernel, dim3(B), dim3(1024), 0, stream, scores, n, k, out_vals, out_idx);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// Candidate list of multiclass_nms (T/mmdet/core/post_processing/bbox_nms.py:28-60): every (proposal, class) pair whose
// score exceeds score_thr, in row-major (proposal, class) order -- boxes shared by all classes (box_stride 4) or one box
// per class (box_stride 4*C); the stored score is score * score_factor[proposal] (the filter uses the raw score, :44-49).
// One workgroup, ordered compaction with wave ballots (the candidate count is a few thousand).
__global__ void nms_candidates_kernel(const float* __restrict__ boxes, int box_stride, const float* __restrict__ scores,
                                      const float* __restrict__ factors, int n, int C, float thr,
                                      float* __restrict__ cboxes, float* __restrict__ cscores,
                                      int* __restrict__ clabels, long long* __restrict__ cinds,
                                      int* __restrict__ count) {
    {   // one workgroup per image: inputs (B, n, .), outputs (B, n * C, .) -- every image owns a slab of the capacity n * C
        const size_t b = blockIdx.x, cap = (size_t)n * C;
        boxes += b * n * box_stride;
        scores += b * n * (C + 1);
        if (factors) factors += b * n;
        cboxes += b * cap * 4;
        cscores += b * cap;
        clabels += b * cap;
        cinds += b * cap;
        count += b;
    }
    __shared__ int s_wcnt[16], s_base;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, w = tid >> 6;
    const long long total = (long long)n * C;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (long long base = 0; base < total; base += nt) {
        const long long j = base + tid;
        const int row = j < total ? (int)(j / C) : 0, cls = j < total ? (int)(j - (long long)row * C) : 0;
        const float sc = j < total ? scores[(size_t)row * (C + 1) + cls] : 0.f;
        const bool a = j < total && sc > thr;
        const unsigned long long bal = __ballot(a);
        if (lane == 0) s_wcnt[w] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int q = 0; q < w; ++q) off += s_wcnt[q];
        if (a) {
            const int o = off + __popcll(bal & ((1ull << lane) - 1ull));
            const float* bp = boxes + (size_t)row * box_stride + (box_stride > 4 ? cls * 4 : 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) cboxes[(size_t)o * 4 + c] = bp[c];
            cscores[o] = factors ? __fmul_rn(sc, factors[row]) : sc;
            clabels[o] = cls;
            cinds[o] = j;
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int q = 0; q < (nt >> 6); ++q) tot += s_wcnt[q];
            s_base += tot;
        }
        __syncthreads();
    }
    if (tid == 0) *count = s_base;
}
extern "C" int cpr_nms_candidates(const float* boxes, int box_stride, const float* scores, const float* factors, int n,
                                  int C, float score_thr, float* cand_boxes, float* cand_scores, int* cand_labels,
                                  long long* cand_inds, int* count, hipStream_t stream) {
    CPR_CHECK_ARG(n >= 0 && C > 0 && (box_stride == 4 || box_stride == 4 * C) && count);
    if (n == 0) {
        hipError_t e = hipMemsetAsync(count, 0, sizeof(int), stream);
        return e == hipSuccess ? CPR_OK : -(int)e;
    }
    CPR_CHECK_ARG(boxes && scores && cand_boxes && cand_scores && cand_labels && cand_inds);
    hipLaunchKernelGGL(nms_candidates_kernel, dim3(1), dim3(1024), 0, stream, boxes, box_stride, scores, factors, n, C,
                       score_thr, cand_boxes, cand_scores, cand_labels, cand_inds, count);
    CPR_LAUNCH_STATUS();
}

// all images of a batch in one launch, NO host read: cand_* are (B, n * C, .) slabs, count (B) stays on the device
extern "C" int cpr_nms_candidates_batched(const float* boxes, int box_stride, const float* scores, const float* factors,
                                          int B, int n, int C, float score_thr, float* cand_boxes, float* cand_scores,
                                          int* cand_labels, long long* cand_inds, int* count, hipStream_t stream) {
    CPR_CHECK_ARG(B >= 0 && n >= 0 && C > 0 && (box_stride == 4 || box_stride == 4 * C));
    if (B == 0) return CPR_OK;
    CPR_CHECK_ARG(count);
    if (n == 0) {
        hipError_t e = hipMemsetAsync(count, 0, sizeof(int) * B, stream);
        return e == hipSuccess ? CPR_OK : -(int)e;
    }
    CPR_CHECK_ARG(boxes && scores && cand_boxes && cand_scores && cand_labels && cand_inds);
    hipLaunchKernelGGL(nms_candidates_kernel, dim3(B), dim3(1024), 0, stream, boxes, box_stride, scores, factors, n, C,
                       score_thr, cand_boxes, cand_scores, cand_labels, cand_inds, count);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// The three NMS kernels serve one problem (cpr_nms: n on the host) or a batch (cpr_nms_batched: blockIdx selects the image,
// n_b is READ FROM THE DEVICE -- the candidate count the kernel above left there -- so no host round trip separates the
// stages; `cap` is the per-image slab size of every array, nblk_cap = ceil(cap / 64) the mask row stride).
struct NmsBatch {
    const int* n_dev;      // per-image candidate counts on the device (nullptr: single problem, n below)
    int n, cap, nblk_cap;
    size_t ws_stride;      // 64-bit words of mask / sort workspace per image
};
__device__ __forceinline__ int nms_n(const NmsBatch& nb, int b) { return nb.n_dev ? nb.n_dev[b] : nb.n; }

__global__ void nms_prepare_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                   const int* __restrict__ labels, NmsBatch nb, int* __restrict__ order,
                                   float* __restrict__ sboxes, unsigned long long* __restrict__ ws) {
    __shared__ float red[16];
    __shared__ float s_max;
    const int tid = threadIdx.x;
    const int n = nms_n(nb, blockIdx.x);
    {
        const size_t b = blockIdx.x;
        boxes += b * nb.cap * 4; scores += b * nb.cap; labels += b * nb.cap; order += b * nb.cap; sboxes += b * nb.cap * 4;
        ws += b * nb.ws_stride;
    }
    if (n == 0) return;
    float m = -INFINITY;
    for (int i = tid; i < n * 4; i += blockDim.x) m = fmaxf(m, boxes[i]);
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        float r = red[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, red[w]);
        s_max = r;
    }
    int P = 1;
    while (P < n) P <<= 1;
    for (int i = tid; i < P; i += blockDim.x)
        ws[i] = i < n ? (((unsigned long long)f2key(scores[i]) << 32) | (unsigned)(~(unsigned)i)) : 0ull;
    __syncthreads();
    block_bitonic_desc(ws, P);
    const float off1 = __fadd_rn(s_max, 1.f);  // max_coordinate + 1
    for (int i = tid; i < n; i += blockDim.x) {
        const int src = (int)(~(unsigned)(ws[i] & 0xffffffffull));
        order[i] = src;
        const float o = __fmul_rn((float)labels[src], off1);
#pragma unroll
        for (int c = 0; c < 4; ++c) sboxes[i * 4 + c] = __fadd_rn(boxes[src * 4 + c], o);
    }
}

// rb0: first 64-row block of this launch (the streamed form of cpr_nms computes the mask a band of rows at a time; the mask rows
// are stored band-relative).  0 for the whole-matrix launches.
__global__ void nms_mask_kernel(const float* __restrict__ b, NmsBatch nb, float thr, unsigned long long* __restrict__ mask, int rb0) {
    const int rb = rb0 + blockIdx.y, cb = blockIdx.x;
    const int n = nms_n(nb, blockIdx.z);
    const int nblk = (n + 63) / 64;       // mask row stride of THIS image (the scan kernel derives the same value)
    if (cb < rb || cb >= nblk) return;    // only j > i matters; tiles past this image's candidates do not exist
    b += (size_t)blockIdx.z * nb.cap * 4;
    mask += (size_t)blockIdx.z * nb.ws_stride;
    __shared__ float cbx[64 * 4];
    const int lane = threadIdx.x;
    const int cj = cb * 64 + lane;
    if (cj < n) {
#pragma unroll
        for (int c = 0; c < 4; ++c) cbx[lane * 4 + c] = b[cj * 4 + c];
    }
    __syncthreads();
    const int i = rb * 64 + lane;
    if (i >= n) return;
    const float x1 = b[i * 4], y1 = b[i * 4 + 1], x2 = b[i * 4 + 2], y2 = b[i * 4 + 3];
    const float ai = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
    unsigned long long bits = 0;
    const int cnt = min(64, n - cb * 64);
    for (int t = 0; t < cnt; ++t) {
        const int j = cb * 64 + t;
        if (j <= i) continue;
        const float u1 = cbx[t * 4], v1 = cbx[t * 4 + 1], u2 = cbx[t * 4 + 2], v2 = cbx[t * 4 + 3];
        const float aj = __fmul_rn(__fsub_rn(u2, u1), __fsub_rn(v2, v1));
        const float w = fmaxf(0.f, __fsub_rn(fminf(x2, u2), fmaxf(x1, u1)));
        const float h = fmaxf(0.f, __fsub_rn(fminf(y2, v2), fmaxf(y1, v1)));
        const float inter = __fmul_rn(w, h);
        const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(ai, aj), inter));
        if (ovr > thr) bits |= 1ull << t;
    }
    mask[(size_t)(i - rb0 * 64) * nblk + cb] = bits;
}

// NMS_SCAN_WORDS 64-bit words of "removed" bits live in LDS: 4096 words = 262 144 candidates per problem.
constexpr int NMS_SCAN_WORDS = 4096;
// Rows [i0, i1) of the greedy scan; the mask rows are band-relative (row i of the band at (i - i0) * nblk).  The whole-matrix
// launches pass (0, n, nullptr); the streamed form carries the removed bits and the keep count from band to band in `carry`
// (nblk words + 1 word for the count).
__global__ void nms_scan_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ order, NmsBatch nb,
                                long long* __restrict__ keep_idx, int* __restrict__ num_keep, int i0, int i1,
                                unsigned long long* __restrict__ carry) {
    __shared__ unsigned long long removed[NMS_SCAN_WORDS];
    const int lane = threadIdx.x;
    const int n = nms_n(nb, blockIdx.x), nblk = (n + 63) / 64;
    {
        const size_t b = blockIdx.x;
        mask += b * nb.ws_stride; order += b * nb.cap; keep_idx += b * nb.cap; num_keep += b;
    }
    if (i1 > n) i1 = n;
    const bool first = carry == nullptr || i0 == 0;
    for (int c = lane; c < nblk; c += 64) removed[c] = first ? 0ull : carry[c];
    __syncthreads();
    int cnt = first ? 0 : (int)carry[nblk];
    for (int i = i0; i < i1; ++i) {
        const unsigned long long w = removed[i >> 6];
        if (!((w >> (i & 63)) & 1ull)) {
            if (lane == 0) keep_idx[cnt] = order[i];
            ++cnt;
            const int c0 = i >> 6;  // columns before c0 hold only j <= i: nothing left to suppress there
            for (int c = c0 + lane; c < nblk; c += 64) removed[c] |= mask[(size_t)(i - i0) * nblk + c];
        }
        __syncthreads();
    }
    if (carry) {
        for (int c = lane; c < nblk; c += 64) carry[c] = removed[c];
        if (lane == 0) carry[nblk] = (unsigned long long)cnt;
    }
    if (lane == 0) *num_keep = cnt;
}

// rows of the pair mask one band of the streamed form holds (64-bit words: band * ceil(n / 64))
constexpr int NMS_BAND_ROWS = 8192;
extern "C" int cpr_nms_workspace(int n) {      // 64-bit words of ws_mask cpr_nms needs for n candidates (-> include/cpr_hip.h)
    if (n <= 0) return 1;
    if (n > 64 * NMS_SCAN_WORDS) return CPR_ERR_ARG;
    const long long nblk = (n + 63) / 64;
    long long P = 1;
    while (P < n) P <<= 1;
    const long long band = n < NMS_BAND_ROWS ? n : NMS_BAND_ROWS;
    const long long m = band * nblk + nblk + 1;
    return (int)(m > P ? m : P);
}

// n is unlimited up to 64 * NMS_SCAN_WORDS: the pair mask is computed and scanned a band of NMS_BAND_ROWS rows at a time (the
// reference has no ceiling either: T/mmdet/core/post_processing/bbox_nms.py:7-94; round 5 refused more than 16 384 candidates).
// ws_mask: cpr_nms_workspace(n) words.
extern "C" int cpr_nms(const float* boxes, const float* scores, const int* labels, int n, float iou_thr,
                       long long* keep_idx, int* num_keep, int* ws_order, float* ws_boxes,
                       unsigned long long* ws_mask, hipStream_t stream) {
    CPR_CHECK_ARG(n >= 0 && n <= 64 * NMS_SCAN_WORDS && num_keep);
    if (n == 0) {
        hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int), stream);
        return e == hipSuccess ? CPR_OK : -(int)e;
    }
    CPR_CHECK_ARG(boxes && scores && labels && keep_idx && ws_order && ws_boxes && ws_mask);
    const int nblk = cdiv(n, 64);
    NmsBatch nb;
    nb.n_dev = nullptr; nb.n = n; nb.cap = n; nb.nblk_cap = nblk; nb.ws_stride = 0;
    hipLaunchKernelGGL(nms_prepare_kernel, dim3(1), dim3(1024), 0, stream, boxes, scores, labels, nb, ws_order, ws_boxes,
                       ws_mask);
    if (n <= NMS_BAND_ROWS) {
        hipLaunchKernelGGL(nms_mask_kernel, dim3(nblk, nblk, 1), dim3(64), 0, stream, ws_boxes, nb, iou_thr, ws_mask, 0);
        hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, stream, ws_mask, ws_order, nb, keep_idx, num_keep, 0, n, nullptr);
        CPR_LAUNCH_STATUS();
    }
    unsigned long long* carry = ws_mask + (size_t)NMS_BAND_ROWS * nblk;
    for (int i0 = 0; i0 < n; i0 += NMS_BAND_ROWS) {
        const int i1 = i0 + NMS_BAND_ROWS < n ? i0 + NMS_BAND_ROWS : n;
        hipLaunchKernelGGL(nms_mask_kernel, dim3(nblk, cdiv(i1 - i0, 64), 1), dim3(64), 0, stream, ws_boxes, nb, iou_thr, ws_mask, i0 / 64);
        hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, stream, ws_mask, ws_order, nb, keep_idx, num_keep, i0, i1, carry);
    }
    CPR_LAUNCH_STATUS();
}

// B problems in one launch each, candidate counts read from the device (cpr_nms_candidates_batched's `count`): boxes (B, cap, 4),
// scores / labels (B, cap) -> keep_idx (B, cap) (slab-relative, descending score), num_keep (B).  Workspaces: ws_order (B, cap)
// int32, ws_boxes (B, cap, 4), ws_mask B * ws_stride 64-bit words with ws_stride >= max(cap * ceil(cap / 64), pow2 >= cap).
// cap <= 16384.  No host synchronisation: the caller reads (count, num_keep) once per batch.
extern "C" int cpr_nms_batched(const float* boxes, const float* scores, const int* labels, const int* n_dev, int B, int cap,
                               float iou_thr, long long* keep_idx, int* num_keep, int* ws_order, float* ws_boxes,
                               unsigned long long* ws_mask, long long ws_stride, hipStream_t stream) {
    CPR_CHECK_ARG(B >= 0 && cap >= 0 && cap <= 16384);
    if (B == 0) return CPR_OK;
    CPR_CHECK_ARG(num_keep && n_dev);
    if (cap == 0) {
        hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int) * B, stream);
        return e == hipSuccess ? CPR_OK : -(int)e;
    }
    CPR_CHECK_ARG(boxes && scores && labels && keep_idx && ws_order && ws_boxes && ws_mask);
    const int nblk = cdiv(cap, 64);
    int P = 1;
    while (P < cap) P <<= 1;
    CPR_CHECK_ARG(ws_stride >= (long long)cap * nblk && ws_stride >= P);
    NmsBatch nb;
    nb.n_dev = n_dev; nb.n = 0; nb.cap = cap; nb.nblk_cap = nblk; nb.ws_stride = (size_t)ws_stride;
    hipLaunchKernelGGL(nms_prepare_kernel, dim3(B), dim3(1024), 0, stream, boxes, scores, labels, nb, ws_order, ws_boxes,
                       ws_mask);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nblk, nblk, B), dim3(64), 0, stream, ws_boxes, nb, iou_thr, ws_mask, 0);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, stream, ws_mask, ws_order, nb, keep_idx, num_keep, 0, cap, nullptr);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// P2PHead.get_pred_points (p2p_head.py:125-170): pred = anchor + point_anchor*stride + reg*gamma*stride.
// reg (N,H,W,2k) NHWC -> pred (N, H*W*k, 3) = (x, y, stride); anchors are PointGenerator.grid_points
// (x*stride, y*stride) with NO half-stride offset (T/mmdet/core/anchor/point_generator.py:17-25).
__global__ void p2p_decode_kernel(const float* __restrict__ reg, const float* __restrict__ point_anchor,
                                  float* __restrict__ pred, float* __restrict__ anchor, int N, int H, int W, int k,
                                  float stride, float gamma) {
    const long long total = (long long)N * H * W * k;
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int a = (int)(i % k);
    const long long pix = i / k;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const float ax = __fadd_rn((float)x * stride, __fmul_rn(point_anchor[a * 2], stride));
    const float ay = __fadd_rn((float)y * stride, __fmul_rn(point_anchor[a * 2 + 1], stride));
    const float rx = reg[i * 2], ry = reg[i * 2 + 1];
    pred[i * 3 + 0] = __fadd_rn(ax, __fmul_rn(__fmul_rn(rx, gamma), stride));
    pred[i * 3 + 1] = __fadd_rn(ay, __fmul_rn(__fmul_rn(ry, gamma), stride));
    pred[i * 3 + 2] = stride;
    if (anchor) { anchor[i * 3] = ax; anchor[i * 3 + 1] = ay; anchor[i * 3 + 2] = stride; }
}

extern "C" int cpr_p2p_decode(const float* reg, const float* point_anchor, float* pred, float* anchor, int N, int H,
                              int W, int k, float stride, float gamma, hipStream_t stream) {
    CPR_CHECK_ARG(reg && point_anchor && pred && N > 0 && H > 0 && W > 0 && k > 0);
    const long long total = (long long)N * H * W * k;
    hipLaunchKernelGGL(p2p_decode_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, stream, reg, point_anchor,
                       pred, anchor, N, H, W, k, stride, gamma);
    CPR_LAUNCH_STATUS();
}

// 3x3 / pad 1 convolution with a HANDFUL of output channels as a 1x1 projection + a tap sum (round 4).  P2PHead's output convs
// (cls_out: 256 -> C k, reg_out: 256 -> 2 k; T/mmdet/models/point/dense_heads/p2p_head.py:84-105,113-123) have 1-2 output
// channels: on the 64-cout tiles of the matrix-core conv 97 % of the multiplies meet zero padding (1.0 ms per launch at B = 16,
// 6.5 % of the P2PNet step for two launches).  conv(x, w)[y][x][j] = sum_taps <x[y+dy][x+dx], w[j][:, tap]> -- so ONE 1x1 GEMM
// with 9 J output rows (row tap * J + j = w[j][:, kh][kw]) gives every pixel's response to every tap, R (N,H,W,9J), and this
// kernel adds the nine shifted responses (a tap that falls outside the map contributes 0 = the conv's zero padding) + bias.
__global__ void tap_sum3x3_kernel(const float* __restrict__ R, const float* __restrict__ bias, float* __restrict__ out,
                                  int N, int H, int W, int J) {
    const long long total = (long long)N * H * W * J;
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % J);
    long long r = i / J;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    float s = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int yy = y + kh - 1;
        if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int xx = x + kw - 1;
            if ((unsigned)xx >= (unsigned)W) continue;
            s += R[(((size_t)n * H + yy) * W + xx) * (9 * J) + (kh * 3 + kw) * J + j];
        }
    }
    out[i] = s + (bias ? bias[j] : 0.f);
}
extern "C" int cpr_tap_sum3x3(const float* R, const float* bias, float* out, int N, int H, int W, int J, hipStream_t stream) {
    CPR_CHECK_ARG(R && out && N > 0 && H > 0 && W > 0 && J > 0);
    const long long total = (long long)N * H * W * J;
    hipLaunchKernelGGL(tap_sum3x3_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, stream, R, bias, out, N, H, W, J);
    CPR_LAUNCH_STATUS();
}

// max over classes of sigmoid(logit): the per-proposal score fed to topk (p2p_head.py:362-369)
__global__ void rowmax_sigmoid_kernel(const float* __restrict__ logits, float* __restrict__ out, long long M, int C) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= M) return;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, sigmoid_torch_cpu(logits[i * C + c]));   // torch's CPU sigmoid bit for bit: decides top-k / NMS order
    out[i] = m;
}
// cls_score.sigmoid() of P2PHead._get_bboxes_single (p2p_head.py:362), elementwise, with torch's CPU bits (the scores
// order the candidates of top-k and NMS, so a last-bit difference can swap two detections)
__global__ void sigmoid_exact_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n) y[i] = sigmoid_torch_cpu(x[i]);
}
extern "C" int cpr_sigmoid(const float* x, float* y, long long n, hipStream_t stream) {
    CPR_CHECK_ARG(n >= 0);
    if (n == 0) return CPR_OK;
    CPR_CHECK_ARG(x && y);
    hipLaunchKernelGGL(sigmoid_exact_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, stream, x, y, n);
    CPR_LAUNCH_STATUS();
}

extern "C" int cpr_rowmax_sigmoid(const float* logits, float* out, long long M, int C, hipStream_t stream) {
    CPR_CHECK_ARG(M >= 0 && C > 0);
    if (M == 0) return CPR_OK;
    CPR_CHECK_ARG(logits && out);
    hipLaunchKernelGGL(rowmax_sigmoid_kernel, dim3((unsigned)cdivll(M, 256)), dim3(256), 0, stream, logits, out, M, C);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// P2PHead.loss_single (p2p_head.py:220-248) fused with sample_result_to_target (:308-328): sigmoid focal loss
// (py_sigmoid_focal_loss, T/mmdet/models/losses/focal_loss.py:11-56) + SmoothL1 (smooth_l1_loss.py:11-28) straight
// from the assignment.  One image per blockIdx.y; per-block partial sums in double, reduced by p2p_loss_finalize.
// cls_mode: 0 sigmoid focal (avg factor = positives), 1 CrossEntropyLoss(use_sigmoid=True) = binary_cross_entropy_with_logits on the
// one-hot labels (cross_entropy_loss.py:58-99; avg factor = ALL proposals of the batch, p2p_head.py:222-224), 2 softmax cross
// entropy over C = num_classes + 1 logits, background = the last (cross_entropy_loss.py:9-40).  reg_mode: 0 SmoothL1(beta), 1 MSE
// (mse_loss.py), 2 L1.  The reference's own P2PHead defaults are modes (1, 1) (p2p_head.py:38-46).
__global__ void p2p_loss_kernel(const float* __restrict__ logits, const float* __restrict__ pred,
                                const long long* __restrict__ gt_inds, const float* __restrict__ gt_pts,
                                const int* __restrict__ gt_labels, const int* __restrict__ gt_start,
                                double* __restrict__ partial, int M, int C, float alpha, float gamma, float beta,
                                float pos_w, float neg_w, float reg_norm, int cls_mode, int reg_mode) {
    __shared__ double red[3][4];
    const int b = blockIdx.y;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    double lc = 0, lp = 0, np_ = 0;
    if (m < M) {
        const size_t r = (size_t)b * M + m;
        const long long gi = gt_inds[r];
        const bool pos = gi > 0;
        const int g = pos ? gt_start[b] + (int)gi - 1 : 0;
        const int nfg = cls_mode == 2 ? C - 1 : C;                   // foreground classes; the background label
        const int label = pos ? gt_labels[g] : nfg;
        // gi < 0: a cell outside the padded image (valid_flags false): the reference un-maps it with label weight 0
        // (p2p_head.py:300-305, unmap fill 0), i.e. it contributes to neither loss -- and with LABEL 0 (softmax mode: class 0)
        const float w = (gi < 0) ? 0.f : pos ? pos_w : (neg_w <= 0.f ? 1.f : neg_w);
        if (cls_mode == 2) {
            float mx = -INFINITY;
            for (int c = 0; c < C; ++c) mx = fmaxf(mx, logits[r * C + c]);
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += expf(logits[r * C + c] - mx);
            const int lab = gi < 0 ? 0 : label;
            lc += (double)((logf(se) + mx - logits[r * C + lab]) * w);
        } else {
            for (int c = 0; c < C; ++c) {
                const float x = logits[r * C + c];
                const float t = (c == label) ? 1.f : 0.f;
                // binary_cross_entropy_with_logits: max(x,0) - x*t + log(1 + exp(-|x|))
                const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
                if (cls_mode == 1) { lc += (double)(bce * w); continue; }
                const float p = 1.f / (1.f + expf(-x));
                const float pt = (1.f - p) * t + p * (1.f - t);
                const float fw = (alpha * t + (1.f - alpha) * (1.f - t)) * ((gamma == 2.f) ? pt * pt : powf(pt, gamma));
                lc += (double)(bce * fw * w);
            }
        }
        if (pos) {
            np_ = 1.0;
            const float s = pred[r * 3 + 2];
            for (int d = 0; d < 2; ++d) {
                const float a = pred[r * 3 + d] / s / reg_norm, t = gt_pts[g * 2 + d] / s / reg_norm;
                const float e = a - t, diff = fabsf(e);
                lp += (double)(reg_mode == 1 ? e * e : reg_mode == 2 ? diff : (diff < beta) ? 0.5f * diff * diff / beta : diff - 0.5f * beta);
            }
        }
    }
    lc = wave_sum_d(lc); lp = wave_sum_d(lp); np_ = wave_sum_d(np_);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lc; red[1][threadIdx.x >> 6] = lp; red[2][threadIdx.x >> 6] = np_; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = partial + ((size_t)b * gridDim.x + blockIdx.x) * 3;
        for (int k = 0; k < 3; ++k) o[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
    }
}

// out (B,2): per image {loss_cls, loss_pts}; avg factor = total positives over the batch (p2p_head.py:199-200)
// (cls_total > 0: the classification loss is averaged over ALL proposals of the batch -- the CrossEntropyLoss modes)
__global__ void p2p_loss_finalize_kernel(const double* __restrict__ partial, int B, int nblk, float w_cls, float w_reg,
                                         float* __restrict__ out, double cls_total) {
    // round 4: every sum is a fixed-shape tree (strided per-thread partials -> wave shuffle -> 16 wave totals in order) instead
    // of one thread walking B * nblk doubles (171 us at B = 16); deterministic, double accumulation as before
    __shared__ double red[16];
    __shared__ double s_np;
    const int tid = threadIdx.x, nt = blockDim.x;
    double t = 0;
    for (int i = tid; i < B * nblk; i += nt) t += partial[(size_t)i * 3 + 2];
    t = wave_sum_d(t);
    if ((tid & 63) == 0) red[tid >> 6] = t;
    __syncthreads();
    if (tid == 0) {
        double r = 0;
        for (int w = 0; w < (nt >> 6); ++w) r += red[w];
        s_np = r;
    }
    __syncthreads();
    const double np_ = s_np;
    // one wave per image (16 waves): lanes stride over the image's blocks
    const int lane = tid & 63;
    for (int b = tid >> 6; b < B; b += (nt >> 6)) {
        double lc = 0, lp = 0;
        for (int i = lane; i < nblk; i += 64) { lc += partial[((size_t)b * nblk + i) * 3]; lp += partial[((size_t)b * nblk + i) * 3 + 1]; }
        lc = wave_sum_d(lc);
        lp = wave_sum_d(lp);
        if (lane == 0) {
            out[b * 2] = (float)(lc / (cls_total > 0 ? cls_total : np_) * w_cls);
            out[b * 2 + 1] = (float)(lp / np_ * w_reg);
        }
    }
}

extern "C" int cpr_p2p_loss(const float* logits, const float* pred, const long long* gt_inds, const float* gt_pts,
                            const int* gt_labels, const int* gt_start, double* ws_partial, float* out, int B, int M,
                            int C, float alpha, float gamma, float beta, float pos_w, float neg_w, float reg_norm,
                            float w_cls, float w_reg, int cls_mode, int reg_mode, hipStream_t stream) {
    CPR_CHECK_ARG(B > 0 && B <= 1024 && M > 0 && C > 0 && (beta > 0 || reg_mode != 0));
    CPR_CHECK_ARG(cls_mode >= 0 && cls_mode <= 2 && reg_mode >= 0 && reg_mode <= 2 && (cls_mode != 2 || C >= 2));
    CPR_CHECK_ARG(logits && pred && gt_inds && gt_pts && gt_labels && gt_start && ws_partial && out);
    const int nblk = cdiv(M, 256);
    hipLaunchKernelGGL(p2p_loss_kernel, dim3(nblk, B), dim3(256), 0, stream, logits, pred, gt_inds, gt_pts, gt_labels,
                       gt_start, ws_partial, M, C, alpha, gamma, beta, pos_w, neg_w, reg_norm, cls_mode, reg_mode);
    hipLaunchKernelGGL(p2p_loss_finalize_kernel, dim3(1), dim3(1024), 0, stream, ws_partial, B, nblk, w_cls, w_reg,
                       out, cls_mode == 0 ? 0.0 : (double)B * (double)M);
    CPR_LAUNCH_STATUS();
}
