// Point <-> gt assignment kernels (integer outputs: bit-exact bar against the reference's CPU path).
//   cpr_point_assign     PointAssigner.assign          T/mmdet/core/bbox/assigners/point_assigner.py:23-133
//   cpr_hungarian_cost   FocalLossCost + DisCostV2     T/mmdet/core/bbox/match_costs/match_cost.py:84-100,197-214
//   cpr_lsa_topk         HungarianAssignerV2's LSA loop T/mmdet/core/bbox/assigners/hungarian_assigner.py:229-268
//                        (replaces scipy.optimize.linear_sum_assignment + the GPU->CPU->GPU round trip)
#include <limits.h>
#include "common.h"

// ------------------------------------------------------------------------------------------------
// block-wide argmin over (value, index): smaller value first, then smaller index.  1024 threads max.
struct MinIdx {
    float v;
    int i;
};
__device__ __forceinline__ MinIdx min_idx(MinIdx a, MinIdx b) {
    return (b.v < a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ MinIdx block_argmin(MinIdx x, MinIdx* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MinIdx y;
        y.v = __shfl_xor(x.v, o, 64);
        y.i = __shfl_xor(x.i, o, 64);
        x = min_idx(x, y);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    MinIdx r = sh[0];
    for (int w = 1; w < (int)((blockDim.x + 63) >> 6); ++w) r = min_idx(r, sh[w]);
    return r;
}

// One workgroup walks the gts in order (the reference's sequential "closer gt wins" overwrite semantics).
__global__ void point_assign_kernel(const float* __restrict__ points, const float* __restrict__ gtb, int n, int k,
                                    float scale, int pos_num, long long* __restrict__ gt_inds,
                                    float* __restrict__ best, int* __restrict__ plvl) {
    __shared__ MinIdx sh[16];
    __shared__ int s_lmin, s_lmax;
    __shared__ int chosen[16];
    __shared__ float chosen_d[16];
    const int tid = threadIdx.x;
    int lmin = INT_MAX, lmax = INT_MIN;
    for (int i = tid; i < n; i += blockDim.x) {
        const int l = (int)log2f(points[i * 3 + 2]);  // torch.log2(stride).int(): truncation
        plvl[i] = l;
        gt_inds[i] = 0;
        best[i] = INFINITY;
        lmin = min(lmin, l);
        lmax = max(lmax, l);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lmin = min(lmin, __shfl_xor(lmin, o, 64));
        lmax = max(lmax, __shfl_xor(lmax, o, 64));
    }
    if (tid == 0) { s_lmin = INT_MAX; s_lmax = INT_MIN; }
    __syncthreads();
    if ((tid & 63) == 0) { atomicMin(&s_lmin, lmin); atomicMax(&s_lmax, lmax); }
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        const float x1 = gtb[j * 4], y1 = gtb[j * 4 + 1], x2 = gtb[j * 4 + 2], y2 = gtb[j * 4 + 3];
        const float gx = __fadd_rn(x1, x2) * 0.5f, gy = __fadd_rn(y1, y2) * 0.5f;
        const float gw = fmaxf(__fsub_rn(x2, x1), 1e-6f), gh = fmaxf(__fsub_rn(y2, y1), 1e-6f);
        int glvl = (int)(__fadd_rn(log2f(__fdiv_rn(gw, scale)), log2f(__fdiv_rn(gh, scale))) * 0.5f);
        glvl = min(max(glvl, s_lmin), s_lmax);
        for (int r = 0; r < pos_num; ++r) {
            MinIdx m;
            m.v = INFINITY;
            m.i = INT_MAX;
            for (int i = tid; i < n; i += blockDim.x) {
                if (plvl[i] != glvl) continue;
                bool taken = false;
                for (int q = 0; q < r; ++q) taken |= (chosen[q] == i);
                if (taken) continue;
                const float dx = __fdiv_rn(__fsub_rn(points[i * 3], gx), gw);
                const float dy = __fdiv_rn(__fsub_rn(points[i * 3 + 1], gy), gh);
                const float d = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
                MinIdx c;
                c.v = d;
                c.i = i;
                m = min_idx(m, c);
            }
            m = block_argmin(m, sh);
            if (tid == 0) { chosen[r] = m.i; chosen_d[r] = m.v; }
            __syncthreads();
        }
        if (tid == 0) {
            for (int r = 0; r < pos_num; ++r) {
                const int p = chosen[r];
                if (p == INT_MAX) continue;
                if (chosen_d[r] < best[p]) { gt_inds[p] = j + 1; best[p] = chosen_d[r]; }
            }
        }
        __syncthreads();
    }
}

extern "C" int cpr_point_assign(const float* points, const float* gt_bboxes, int n, int k, float scale, int pos_num,
                                long long* gt_inds, float* ws_best, int* ws_lvl, hipStream_t stream) {
    CPR_CHECK_ARG(n >= 0 && k >= 0 && pos_num > 0 && pos_num <= 16 && scale > 0);
    if (n == 0) return CPR_OK;
    CPR_CHECK_ARG(points && gt_inds && ws_best && ws_lvl && (k == 0 || gt_bboxes));
    hipLaunchKernelGGL(point_assign_kernel, dim3(1), dim3(1024), 0, stream, points, gt_bboxes, n, k, scale, pos_num,
                       gt_inds, ws_best, ws_lvl);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// cost^T[g][m] = w_cls * (pos - neg)[m, label_g] + w_dis * L1(pred_m / f, gt_g / f), fp32, evaluation order
// of the reference expressions kept.  Exactly tied optimal assignments are common with an L1 cost, so the cost BITS
// decide which one scipy returns; the transcendental functions therefore mirror what torch's CPU kernels compute:
//  * sigmoid = 1 / (1 + Sleef_expf_u10(-x))   (ATen's vectorised sigmoid; sleef_expf_u10() below reproduces the
//    Sleef routine operation for operation: 0 mismatches against torch.sigmoid on 2^20 random inputs)
//  * log goes through MKL VML (HA mode, ~correctly rounded: 0.1 % of values differ from the correctly rounded result);
//    log_cr() evaluates in float64 and rounds once, i.e. the correctly rounded value.
__device__ __forceinline__ float log_cr(float x) { return (float)log((double)x); }
__global__ void hungarian_cost_kernel(const float* __restrict__ pred, int pred_stride,
                                      const float* __restrict__ logits, int C, const float* __restrict__ gt,
                                      const int* __restrict__ labels, float* __restrict__ costT, int M, int G,
                                      float w_cls, float alpha, float gamma, float eps, float w_dis, float fx,
                                      float fy, int p_norm) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = blockIdx.y;
    if (m >= M) return;
    const int l = labels[g];
    const float x = logits[(size_t)m * C + l];
    const float p = sigmoid_torch_cpu(x);   // torch.sigmoid on CPU, bit for bit
    const float pg = (gamma == 2.f) ? __fmul_rn(p, p) : powf(p, gamma);
    const float q1 = __fsub_rn(1.f, p);
    const float qg = (gamma == 2.f) ? __fmul_rn(q1, q1) : powf(q1, gamma);
    const float neg = __fmul_rn(__fmul_rn(-log_cr(__fadd_rn(q1, eps)), 1.f - alpha), pg);
    const float pos = __fmul_rn(__fmul_rn(-log_cr(__fadd_rn(p, eps)), alpha), qg);
    const float cls = __fmul_rn(__fsub_rn(pos, neg), w_cls);
    const float px = __fdiv_rn(pred[(size_t)m * pred_stride], fx), py = __fdiv_rn(pred[(size_t)m * pred_stride + 1], fy);
    const float gx = __fdiv_rn(gt[g * 2], fx), gy = __fdiv_rn(gt[g * 2 + 1], fy);
    float dist;
    if (p_norm == 1) {              // torch.cdist(p=1): |dx| + |dy|
        dist = __fadd_rn(fabsf(__fsub_rn(px, gx)), fabsf(__fsub_rn(py, gy)));
    } else if (M > 25 || G > 25) {  // torch.cdist(p=2), matmul form (use_mm_for_euclid_dist_if_necessary), IEEE sqrt
        dist = __fsqrt_rn(fmaxf(d2_chain(px, py, sq_norm(px, py), gx, gy, sq_norm(gx, gy)), 0.f));
    } else {                        // small problems: the direct kernel sqrt(dx^2 + dy^2)
        const float dx = __fsub_rn(px, gx), dy = __fsub_rn(py, gy);
        dist = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    }
    const float dis = __fmul_rn(dist, w_dis);
    costT[(size_t)g * M + m] = __fadd_rn(cls, dis);
}

extern "C" int cpr_hungarian_cost(const float* pred, int pred_stride, const float* logits, int C, const float* gt,
                                  const int* labels, float* costT, int M, int G, float w_cls, float alpha,
                                  float gamma, float eps, float w_dis, float fx, float fy, int p_norm,
                                  hipStream_t stream) {
    CPR_CHECK_ARG(M >= 0 && G >= 0 && C > 0 && pred_stride >= 2 && (p_norm == 1 || p_norm == 2));
    if (M == 0 || G == 0) return CPR_OK;
    CPR_CHECK_ARG(pred && logits && gt && labels && costT);
    hipLaunchKernelGGL(hungarian_cost_kernel, dim3(cdiv(M, 256), G), dim3(256), 0, stream, pred, pred_stride, logits,
                       C, gt, labels, costT, M, G, w_cls, alpha, gamma, eps, w_dis, fx, fy, p_norm);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// The general cost matrix of HungarianAssignerV2 / HungarianAssigner (round 6): any LIST of the reference's classification and
// regression costs (T/mmdet/core/bbox/match_costs/match_cost.py), summed the way ``sum(cls_costs) + sum(reg_costs)`` sums them
// (hungarian_assigner.py:225-229: left to right from 0).  The shipped P2P pair keeps the fused kernel above.  Terms:
//   cls  0 FocalLossCost (weight, alpha, gamma, eps)   1 ClassificationCost / ClassificationCostV2(use_sigmoid=False): -softmax[label]
//        2 ClassificationCostV2(use_sigmoid=True): -sigmoid[label]   3 ZeroCost
//   reg  0 DisCostV2 (weight, p, fx, fy) on (x, y) points   1 BBoxL1Cost on xyxy boxes (L1 over the four coordinates)
//        2 IoUCost 'iou' / 3 IoUCost 'giou' on xyxy boxes (bbox_overlaps, eps 1e-6)
// sigmoid, the focal logs and both distances carry the CPU bits as in the fused kernel; the softmax is max-subtracted with the
// Sleef exp and a left-to-right sum (ATen's vectorised row reduction adds in lane order: the last bit can differ -- the indices
// are pinned on reference fixtures, the cost to 2 ulp).
struct CostTerm { int type; float w, a, b, c, d; };
struct CostTerms { int ncls, nreg; CostTerm cls[4], reg[4]; };
__global__ void match_cost_kernel(const float* __restrict__ pred, int pdim, const float* __restrict__ logits, int C,
                                  const float* __restrict__ gt, const int* __restrict__ labels, float* __restrict__ costT,
                                  int M, int G, CostTerms t) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = blockIdx.y;
    if (m >= M) return;
    const int l = labels[g];
    float cls = 0.f;
    for (int k = 0; k < t.ncls; ++k) {
        const CostTerm& ct = t.cls[k];
        float v = 0.f;
        const float x = logits[(size_t)m * C + l];
        if (ct.type == 0) {
            const float alpha = ct.a, gamma = ct.b, eps = ct.c;
            const float p = sigmoid_torch_cpu(x);
            const float pg = (gamma == 2.f) ? __fmul_rn(p, p) : powf(p, gamma);
            const float q1 = __fsub_rn(1.f, p);
            const float qg = (gamma == 2.f) ? __fmul_rn(q1, q1) : powf(q1, gamma);
            const float neg = __fmul_rn(__fmul_rn(-log_cr(__fadd_rn(q1, eps)), 1.f - alpha), pg);
            const float pos = __fmul_rn(__fmul_rn(-log_cr(__fadd_rn(p, eps)), alpha), qg);
            v = __fmul_rn(__fsub_rn(pos, neg), ct.w);
        } else if (ct.type == 1) {
            float mx = -INFINITY;
            for (int c = 0; c < C; ++c) mx = fmaxf(mx, logits[(size_t)m * C + c]);
            float se = 0.f;
            for (int c = 0; c < C; ++c) se = __fadd_rn(se, sleef_expf_u10(__fsub_rn(logits[(size_t)m * C + c], mx)));
            v = __fmul_rn(-__fdiv_rn(sleef_expf_u10(__fsub_rn(x, mx)), se), ct.w);
        } else if (ct.type == 2) {
            v = __fmul_rn(-sigmoid_torch_cpu(x), ct.w);
        }
        cls = __fadd_rn(cls, v);
    }
    float reg = 0.f;
    for (int k = 0; k < t.nreg; ++k) {
        const CostTerm& rt = t.reg[k];
        float v = 0.f;
        const float* pb = pred + (size_t)m * pdim;
        const float* gb = gt + (size_t)g * pdim;
        if (rt.type == 0) {
            const float fx = rt.b, fy = rt.c;
            const float px = __fdiv_rn(pb[0], fx), py = __fdiv_rn(pb[1], fy), gx = __fdiv_rn(gb[0], fx), gy = __fdiv_rn(gb[1], fy);
            float dist;
            if (rt.a == 1.f) dist = __fadd_rn(fabsf(__fsub_rn(px, gx)), fabsf(__fsub_rn(py, gy)));
            else if (M > 25 || G > 25) dist = __fsqrt_rn(fmaxf(d2_chain(px, py, sq_norm(px, py), gx, gy, sq_norm(gx, gy)), 0.f));
            else {
                const float dx = __fsub_rn(px, gx), dy = __fsub_rn(py, gy);
                dist = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
            }
            v = __fmul_rn(dist, rt.w);
        } else if (rt.type == 1) {
            float d = 0.f;
            for (int c = 0; c < 4; ++c) d = __fadd_rn(d, fabsf(__fsub_rn(pb[c], gb[c])));
            v = __fmul_rn(d, rt.w);
        } else {
            // bbox_overlaps(mode, is_aligned=False, eps=1e-6): T/mmdet/core/bbox/iou_calculators/iou2d_calculator.py
            const float a1 = __fmul_rn(__fsub_rn(pb[2], pb[0]), __fsub_rn(pb[3], pb[1]));
            const float a2 = __fmul_rn(__fsub_rn(gb[2], gb[0]), __fsub_rn(gb[3], gb[1]));
            const float w = fmaxf(__fsub_rn(fminf(pb[2], gb[2]), fmaxf(pb[0], gb[0])), 0.f);
            const float h = fmaxf(__fsub_rn(fminf(pb[3], gb[3]), fmaxf(pb[1], gb[1])), 0.f);
            const float ov = __fmul_rn(w, h);
            const float uni = fmaxf(__fsub_rn(__fadd_rn(a1, a2), ov), 1e-6f);
            float iou = __fdiv_rn(ov, uni);
            if (rt.type == 3) {
                const float ew = fmaxf(__fsub_rn(fmaxf(pb[2], gb[2]), fminf(pb[0], gb[0])), 0.f);
                const float eh = fmaxf(__fsub_rn(fmaxf(pb[3], gb[3]), fminf(pb[1], gb[1])), 0.f);
                const float ea = fmaxf(__fmul_rn(ew, eh), 1e-6f);
                iou = __fsub_rn(iou, __fdiv_rn(__fsub_rn(ea, uni), ea));
            }
            v = __fmul_rn(-iou, rt.w);
        }
        reg = __fadd_rn(reg, v);
    }
    costT[(size_t)g * M + m] = __fadd_rn(cls, reg);
}

// terms: 6 floats per term (type, weight, a, b, c, d), the ncls classification terms first (host memory)
extern "C" int cpr_match_cost(const float* pred, int pdim, const float* logits, int C, const float* gt, const int* labels,
                              float* costT, int M, int G, const float* terms, int ncls, int nreg, hipStream_t stream) {
    CPR_CHECK_ARG(M >= 0 && G >= 0 && C > 0 && (pdim == 2 || pdim == 4) && ncls >= 0 && ncls <= 4 && nreg >= 0 && nreg <= 4 && terms);
    if (M == 0 || G == 0) return CPR_OK;
    CPR_CHECK_ARG(pred && logits && gt && labels && costT);
    CostTerms t;
    t.ncls = ncls; t.nreg = nreg;
    for (int k = 0; k < ncls + nreg; ++k) {
        CostTerm& x = k < ncls ? t.cls[k] : t.reg[k - ncls];
        const float* f = terms + 6 * k;
        x.type = (int)f[0]; x.w = f[1]; x.a = f[2]; x.b = f[3]; x.c = f[4]; x.d = f[5];
        CPR_CHECK_ARG(x.type >= 0 && x.type <= 3);
        if (k >= ncls) CPR_CHECK_ARG(x.type == 0 ? pdim == 2 : pdim == 4);
    }
    hipLaunchKernelGGL(match_cost_kernel, dim3(cdiv(M, 256), G), dim3(256), 0, stream, pred, pdim, logits, C, gt, labels, costT, M, G, t);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// Rectangular linear sum assignment, shortest augmenting path with float64 duals: an operation-for-operation
// parallelisation of scipy's linear_sum_assignment (rectangular_lsap.cpp, Crouse 2016), INCLUDING its tie-breaking.
// The L1 distance term makes exactly tied optima common (swapping two matches often leaves the total unchanged), so
// matching scipy's indices needs its scan order: columns are visited in the order of its `remaining` array
// (initialised nc-1..0, swap-with-last on removal); among equal shortest-path values the LAST unassigned column in scan
// order wins, otherwise the FIRST assigned one.  `remaining` / `pos` reproduce that order; the parallel arg-min carries
// (value, unassigned, scan position).  One workgroup per problem.  Rows are the G gts, columns the M proposals (M >= G).
// `topk` rounds: matched proposals are retired and the problem is re-solved on the order-preserving compaction of the
// rest, exactly as ``cost[cost_assign == 0]`` does (hungarian_assigner.py:248-268).
struct ArgD {
    double v;
    int unassigned;
    int it;  // position in scipy's `remaining` array
    int c;   // compact column id
};
__device__ __forceinline__ ArgD argd_min(ArgD a, ArgD b) {
    if (b.v < a.v) return b;
    if (b.v > a.v) return a;
    if (b.unassigned != a.unassigned) return b.unassigned > a.unassigned ? b : a;
    if (a.unassigned) return b.it > a.it ? b : a;  // last unassigned in scan order
    return b.it < a.it ? b : a;                    // first assigned in scan order
}

constexpr int LSA_MAX_VIS = 1024;   // columns one row search can scan = rows on the alternating path + 1 <= G + 1

__global__ void lsa_topk_kernel(const float* __restrict__ costT_all, const int* __restrict__ m_of,
                                const int* __restrict__ g_of, const long long* __restrict__ cost_off,
                                const long long* __restrict__ col_off, const long long* __restrict__ row_off, int topk,
                                long long* __restrict__ gt_inds_all, double* __restrict__ v_all,
                                double* __restrict__ spc_all, int* __restrict__ path_all, int* __restrict__ row4col_all,
                                unsigned char* __restrict__ sc_all, unsigned char* __restrict__ active_all,
                                int* __restrict__ cols_all, int* __restrict__ remaining_all, int* __restrict__ pos_all,
                                double* __restrict__ u_all, int* __restrict__ col4row_all,
                                unsigned char* __restrict__ sr_all, int* __restrict__ status) {
    __shared__ ArgD sh[16];
    __shared__ int s_i, s_sink, s_fail, s_nact, s_nrem, s_base, s_wcnt[16];
    __shared__ int s_nvis, s_vis[LSA_MAX_VIS], s_vidx[LSA_MAX_VIS], s_vlast[LSA_MAX_VIS];
    __shared__ double s_minval;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int M = m_of[b], G = g_of[b];
    const float* costT = costT_all + cost_off[b];
    long long* gt_inds = gt_inds_all + col_off[b];
    double* v = v_all + col_off[b];
    double* spc = spc_all + col_off[b];
    int* path = path_all + col_off[b];
    int* row4col = row4col_all + col_off[b];
    unsigned char* SC = sc_all + col_off[b];
    unsigned char* active = active_all + col_off[b];
    int* cols = cols_all + col_off[b];
    int* remaining = remaining_all + col_off[b];
    int* pos = pos_all + col_off[b];
    double* u = u_all + row_off[b];
    int* col4row = col4row_all + row_off[b];
    unsigned char* SR = sr_all + row_off[b];

    for (int j = tid; j < M; j += nt) { gt_inds[j] = 0; active[j] = 1; }
    if (tid == 0) s_fail = 0;
    __syncthreads();
    if (G == 0 || M == 0) return;
    if (G + 1 > LSA_MAX_VIS) {   // the visited-column log of a row search would not fit: refuse (status 2), never truncate
        if (tid == 0 && status) status[b] = 2;
        return;
    }

    for (int round = 0; round < topk; ++round) {
        // ordered compaction of the still-active proposals: cols[c] = original index of the c-th active one
        if (tid == 0) s_base = 0;
        __syncthreads();
        for (int base = 0; base < M; base += nt) {
            const int j = base + tid;
            const bool a = j < M && active[j];
            const unsigned long long bal = __ballot(a);
            const int lane = tid & 63, w = tid >> 6;
            if (lane == 0) s_wcnt[w] = __popcll(bal);
            __syncthreads();
            int off = s_base;
            for (int q = 0; q < w; ++q) off += s_wcnt[q];
            if (a) cols[off + __popcll(bal & ((1ull << lane) - 1ull))] = j;
            __syncthreads();
            if (tid == 0) {
                int tot = 0;
                for (int q = 0; q < (nt >> 6); ++q) tot += s_wcnt[q];
                s_base += tot;
            }
            __syncthreads();
        }
        if (tid == 0) s_nact = s_base;
        __syncthreads();
        const int Mc = s_nact;
        if (Mc / G == 0) break;  // cost_new.shape[0] // num_gts != 0
        // per-round state of the column scan order (scipy's `remaining` array and its inverse) and the visited flags: set
        // up once; every row search below undoes its own few edits instead of re-initialising 25 600 entries per gt
        for (int c = tid; c < Mc; c += nt) {
            v[c] = 0.0; row4col[c] = -1;
            SC[c] = 0;
            remaining[c] = Mc - 1 - c;  // remaining[it] = nc - it - 1
            pos[c] = Mc - 1 - c;
        }
        for (int i = tid; i < G; i += nt) { u[i] = 0.0; col4row[i] = -1; }
        __syncthreads();
        for (int cur = 0; cur < G; ++cur) {
            for (int i = tid; i < G; i += nt) SR[i] = 0;
            if (tid == 0) { s_i = cur; s_sink = -1; s_minval = 0.0; s_nrem = Mc; s_nvis = 0; }
            __syncthreads();
            bool first = true;   // first scan of this row: shortestPathCosts = inf, path = -1 for every column (not stored)
            while (true) {
                const int i = s_i;
                const double minval = s_minval, ui = u[i];
                const float* crow = costT + (size_t)i * M;
                ArgD best;
                best.v = INFINITY; best.unassigned = 0; best.it = INT_MAX; best.c = -1;
                for (int c = tid; c < Mc; c += nt) {
                    if (SC[c]) continue;
                    const double r = ((minval + (double)crow[cols[c]]) - ui) - v[c];
                    double s = first ? (double)INFINITY : spc[c];
                    if (r < s) { path[c] = i; spc[c] = r; s = r; }
                    else if (first) { path[c] = -1; spc[c] = s; }
                    ArgD x;
                    x.v = s; x.unassigned = (row4col[c] == -1) ? 1 : 0; x.it = pos[c]; x.c = c;
                    best = (best.c < 0) ? x : argd_min(best, x);
                }
                first = false;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    ArgD y;
                    y.v = __shfl_xor(best.v, o, 64);
                    y.unassigned = __shfl_xor(best.unassigned, o, 64);
                    y.it = __shfl_xor(best.it, o, 64);
                    y.c = __shfl_xor(best.c, o, 64);
                    if (y.c >= 0) best = (best.c < 0) ? y : argd_min(best, y);
                }
                __syncthreads();
                if ((tid & 63) == 0) sh[tid >> 6] = best;
                __syncthreads();
                if (tid == 0) {
                    ArgD r = sh[0];
                    for (int w = 1; w < (nt + 63) / 64; ++w)
                        if (sh[w].c >= 0) r = (r.c < 0) ? sh[w] : argd_min(r, sh[w]);
                    SR[i] = 1;
                    if (r.c < 0 || r.v == INFINITY) {
                        s_fail = 1; s_sink = -2;
                    } else {
                        s_minval = r.v;
                        SC[r.c] = 1;
                        const int idx = pos[r.c], last = remaining[s_nrem - 1];  // remaining[index] = remaining[--n]
                        remaining[idx] = last;
                        pos[last] = idx;
                        s_nrem -= 1;
                        s_vis[s_nvis] = r.c; s_vidx[s_nvis] = idx; s_vlast[s_nvis] = last;   // a search visits <= G columns
                        s_nvis += 1;
                        if (row4col[r.c] == -1) s_sink = r.c; else s_i = row4col[r.c];
                    }
                }
                __syncthreads();
                if (s_sink != -1) break;
            }
            if (s_fail) break;
            const double minval = s_minval;
            const int nvis = s_nvis;
            for (int i = tid; i < G; i += nt) {
                if (i == cur) u[i] += minval;
                else if (SR[i]) u[i] += minval - spc[col4row[i]];
            }
            __syncthreads();
            if (tid < nvis) {            // v[j] -= minVal - shortestPathCosts[j] for the scanned columns only
                const int c = s_vis[tid];
                v[c] -= minval - spc[c];
                SC[c] = 0;
            }
            __syncthreads();
            if (tid == 0) {
                int j = s_sink;          // augment along the path
                while (true) {
                    const int i = path[j];
                    row4col[j] = i;
                    const int t = col4row[i];
                    col4row[i] = j;
                    j = t;
                    if (i == cur) break;
                }
                for (int k = nvis - 1; k >= 0; --k) {   // put the scan order back (reverse of the swaps above)
                    remaining[s_vidx[k]] = s_vis[k];
                    pos[s_vlast[k]] = Mc - k - 1;
                    pos[s_vis[k]] = s_vidx[k];
                }
            }
            __syncthreads();
        }
        if (s_fail) break;
        for (int i = tid; i < G; i += nt) {
            const int j = cols[col4row[i]];
            gt_inds[j] = i + 1;
            active[j] = 0;
        }
        __syncthreads();
    }
    if (tid == 0 && status) status[b] = s_fail;
}


// ------------------------------------------------------------------------------------------------
// Round 4: the same algorithm with NO per-column search state in memory.  The kernel above streams ~33 bytes per column per
// scan through one CU (v, shortest-path cost, path, row4col, visited flag, scan position, compaction index, cost: 0.85 MB per
// scan, 161 scans per image) and waits for every one of them: 5.5 ms per launch at 16 images x 32 gts x 25 600 proposals, 15 %
// of the P2PNet step (profiles/round3_p2p_kernel_stats.csv).  Two observations remove the state:
//   * the shortest-path cost of column c after the n-th scan of a search is min over the scanned rows i_m of
//     ((minval_m + cost[i_m][c]) - u[i_m]) - v[c], and path[c] is the FIRST row attaining it (the update is `r < s`, strict).
//     Both are functions of the n (row, minval) pairs of the search -- kept in LDS -- so they are RECOMPUTED instead of stored:
//     a scan reads n cost entries per column.  n is 1 for 99 % of the searches (a gt's best proposal is free), so a scan is one
//     read of the cost row, 13 loads in flight per thread, and nothing else; the terms and their evaluation order are the
//     stored form's, so every comparison sees the same doubles.
//   * thread t owns columns t + 1024 k (k < KMAX) for a whole round and keeps their flags in registers, one bit per column:
//     visited in this search / assigned (row4col != -1) / "dual v is non-zero" (v is read only then: v - 0.0 is exact; v != 0
//     only for the few hundred columns a search has ever visited).  The scan position pos[c] = Mc - 1 - c differs only for the
//     <= G low columns the swap-with-last of scipy's `remaining` array moved during THIS search: LDS shadows of the first /
//     last LSA_MAX_VIS entries of pos / remaining.  u and col4row live in LDS; thread 0's bookkeeping between two scans touches
//     LDS only (the winner's row4col rides in the arg-min record).
// Arithmetic, comparison order and tie-breaking are the kernel above's, operation for operation: the two are interchangeable
// bit for bit (tests/test_gpu_assigners.py runs both against each other and against scipy).
#ifdef CPR_BENCH_HOOKS
// measurement build only (libcprhip_bench.so): shader-clock time thread 0 of workgroup 0 spends in the phases of a row search
__device__ long long lsa_phase_clk[8];
#define LSA_T(k)                                                     \
    do {                                                             \
        if (tid == 0 && b == 0) {                                    \
            const long long t_ = clock64();                          \
            lsa_phase_clk[k] += t_ - t_last;                         \
            t_last = t_;                                             \
        }                                                            \
    } while (0)
extern "C" int cpr_lsa_phase_clocks(long long* host_out, int reset) {
    if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lsa_phase_clk), sizeof(long long) * 8) != hipSuccess) return -1;
    if (reset) {
        long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lsa_phase_clk), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define LSA_T(k)
#endif

struct ArgR {
    double v;
    int r4c;   // row4col of the column (-1 = unassigned)
    int it;    // position in scipy's `remaining` array
    int c;     // compact column id
};
__device__ __forceinline__ ArgR argr_min(ArgR a, ArgR b) {
    if (b.v < a.v) return b;
    if (b.v > a.v) return a;
    const int ua = a.r4c == -1, ub = b.r4c == -1;
    if (ub != ua) return ub > ua ? b : a;
    if (ua) return b.it > a.it ? b : a;            // last unassigned in scan order
    return b.it < a.it ? b : a;                    // first assigned in scan order
}

template <int KMAX>
__global__ void __launch_bounds__(1024) lsa_topk_reg_kernel(
        const float* __restrict__ costT_all, const int* __restrict__ m_of, const int* __restrict__ g_of,
        const long long* __restrict__ cost_off, const long long* __restrict__ col_off, const long long* __restrict__ row_off,
        int topk, long long* __restrict__ gt_inds_all, double* __restrict__ v_all, int* __restrict__ row4col_all,
        unsigned char* __restrict__ active_all, int* __restrict__ cols_all, int* __restrict__ status) {
    constexpr int NT = 1024, S = LSA_MAX_VIS;
    constexpr int CH = KMAX >= 20 ? (KMAX + 2) / 3 : (KMAX > 8 ? KMAX / 2 : KMAX);   // columns whose loads are in flight together (9 of 25)
    static_assert(S == NT, "slot 0 of a thread must hold exactly the columns of the LDS shadows");
    __shared__ ArgR sh[16];
    __shared__ int s_sink, s_fail, s_nact, s_base, s_wcnt[16], s_nvis, s_win, s_nrow;
    __shared__ int s_vis[S], s_vq[S], s_vlast[S];     // visited columns of a search; the shadow entries each removal edited
    __shared__ int s_pos[S], s_rem[S];                // pos[c] for c < S; remaining[Mc - 1 - q] for q < S
    __shared__ int s_c4r[S];                          // col4row
    __shared__ int s_row[S];                          // the rows scanned in this search, in scan order (s_row[0] = cur)
    __shared__ double s_u[S];                         // row duals
    __shared__ double s_rowmin[S];                    // minval when row s_row[m] was scanned (0 for the first)
    __shared__ double s_vspc[S];                      // shortest-path cost of the k-th visited column (= minval at its selection)
    const int b = blockIdx.x, tid = threadIdx.x;
    const int M = m_of[b], G = g_of[b];
    const float* costT = costT_all + cost_off[b];
    long long* gt_inds = gt_inds_all + col_off[b];
    double* v = v_all + col_off[b];
    int* row4col = row4col_all + col_off[b];
    unsigned char* active = active_all + col_off[b];
    int* cols = cols_all + col_off[b];

    for (int j = tid; j < M; j += NT) { gt_inds[j] = 0; active[j] = 1; }
    if (tid == 0) s_fail = 0;
    __syncthreads();
    if (G == 0 || M == 0) return;
    if (G + 1 > S || M > KMAX * NT) {      // never truncate: refuse (the launcher checks too)
        if (tid == 0 && status) status[b] = 2;
        return;
    }
#ifdef CPR_BENCH_HOOKS
    long long t_last = clock64();
#endif
    // the cost matrix through a buffer descriptor: one 32-bit lane offset per column + the row's scalar offset (64-bit per-lane
    // addresses would double the registers the columns cost); G * M * 4 < 2^31 is checked by the launcher
    const __amdgpu_buffer_rsrc_t rs_cost = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(costT), 0, G * M * 4, 0x00020000);
    unsigned colreg[KMAX];                  // BYTE offset of the thread's k-th column in a cost row (4 * cols[c]); 0 past Mc (never used there)

    for (int round = 0; round < topk; ++round) {
        LSA_T(7);
        // ordered compaction of the still-active proposals: cols[c] = original index of the c-th active one
        if (tid == 0) s_base = 0;
        __syncthreads();
        for (int base = 0; base < M; base += NT) {
            const int j = base + tid;
            const bool a = j < M && active[j];
            const unsigned long long bal = __ballot(a);
            const int lane = tid & 63, w = tid >> 6;
            if (lane == 0) s_wcnt[w] = __popcll(bal);
            __syncthreads();
            int off = s_base;
            for (int q = 0; q < w; ++q) off += s_wcnt[q];
            if (a) cols[off + __popcll(bal & ((1ull << lane) - 1ull))] = j;
            __syncthreads();
            if (tid == 0) {
                int tot = 0;
                for (int q = 0; q < (NT >> 6); ++q) tot += s_wcnt[q];
                s_base += tot;
            }
            __syncthreads();
        }
        if (tid == 0) s_nact = s_base;
        __syncthreads();
        const int Mc = s_nact;
        if (Mc / G == 0) break;  // cost_new.shape[0] // num_gts != 0
        int tr = tid;                       // (opaque per round: keeps 3 x KMAX column addresses out of registers, see `tq` below)
        asm volatile("" : "+v"(tr));
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = tr + k * NT;
            colreg[k] = c < Mc ? (unsigned)cols[c] * 4u : 0u;
            if (c < Mc) { v[c] = 0.0; row4col[c] = -1; }
        }
        for (int i = tid; i < G; i += NT) { s_u[i] = 0.0; s_c4r[i] = -1; }
        for (int q = tid; q < S; q += NT) { s_pos[q] = Mc - 1 - q; s_rem[q] = q; }
        unsigned asgbits = 0, vbits = 0;
        __syncthreads();
        LSA_T(0);                                   // round set-up (compaction, state init)
        for (int cur = 0; cur < G; ++cur) {
            if (tid == 0) { s_row[0] = cur; s_rowmin[0] = 0.0; s_nrow = 1; s_sink = -1; s_nvis = 0; }
            unsigned scbits = 0;
            __syncthreads();
            LSA_T(1);                               // search set-up
            while (true) {
                const int nrow = s_nrow;            // rows scanned so far, the current one included (its scan is this one)
                ArgR best;
                best.v = INFINITY; best.r4c = 0; best.it = INT_MAX; best.c = -1;
                // opaque copy of the thread id, re-made per scan: the column addresses into v / row4col (2 arrays x KMAX 64-bit
                // pointers) are then not loop-invariant, so the compiler computes them where they are used instead of hoisting
                // ~100 registers of addresses out of the search loop (and spilling them)
                int tq = tid;
                asm volatile("" : "+v"(tq));
                if (nrow == 1) {
                    // the common case (99 % of the searches end after their first scan): one row -- every cost load of the thread
                    // is issued before the first is consumed (KMAX registers, one memory latency per scan), candidates are
                    // folded straight into the arg-min, no per-column minimum to carry
                    const int i = __builtin_amdgcn_readfirstlane(s_row[0]);
                    const double minval = s_rowmin[0], ui = s_u[i];
                    const int row_off_b = i * M * 4;
                    float cst[KMAX];
#pragma unroll
                    for (int k = 0; k < KMAX; ++k)
                        cst[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_cost, (int)colreg[k], row_off_b, 0));
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        const int c = tq + k * NT;
                        if (c < Mc && !((scbits >> k) & 1u)) {
                            const double vv = ((vbits >> k) & 1u) ? v[c] : 0.0;
                            double sp1 = (double)INFINITY;
                            const double r = ((minval + (double)cst[k]) - ui) - vv;
                            if (r < sp1) sp1 = r;
                            ArgR x;
                            x.v = sp1; x.r4c = ((asgbits >> k) & 1u) ? row4col[c] : -1;
                            x.it = (k == 0) ? s_pos[c] : Mc - 1 - c; x.c = c;
                            best = (best.c < 0) ? x : argr_min(best, x);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int kb = 0; kb < KMAX; kb += CH) {
                        double sp[CH];
#pragma unroll
                        for (int j = 0; j < CH; ++j) sp[j] = (double)INFINITY;
                        // shortest-path cost of the chunk's columns = min over the scanned rows, rows in scan order, strict `<`
                        for (int m = 0; m < nrow; ++m) {
                            const int i = __builtin_amdgcn_readfirstlane(s_row[m]);      // uniform: the row offset is the load's scalar
                            const double minval = s_rowmin[m], ui = s_u[i];               // offset, the column its 32-bit lane offset
                            const int row_off_b = i * M * 4;
                            float cst[CH];
#pragma unroll
                            for (int j = 0; j < CH; ++j) {            // CH independent loads in flight (clamped: no branch around a load)
                                const int k = kb + j;
                                cst[j] = (k < KMAX) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_cost, (int)colreg[k < KMAX ? k : 0], row_off_b, 0)) : 0.f;
                            }
#pragma unroll
                            for (int j = 0; j < CH; ++j) {
                                const int k = kb + j;
                                if (k < KMAX) {
                                    const int c = tq + k * NT;
                                    const double vv = ((vbits >> k) & 1u) ? v[c] : 0.0;
                                    const double r = ((minval + (double)cst[j]) - ui) - vv;
                                    if (r < sp[j]) sp[j] = r;
                                }
                            }
                        }
#pragma unroll
                        for (int j = 0; j < CH; ++j) {
                            const int k = kb + j;
                            if (k < KMAX) {
                                const int c = tq + k * NT;
                                if (c < Mc && !((scbits >> k) & 1u)) {
                                    ArgR x;
                                    x.v = sp[j]; x.r4c = ((asgbits >> k) & 1u) ? row4col[c] : -1;
                                    x.it = (k == 0) ? s_pos[c] : Mc - 1 - c; x.c = c;      // (S == NT: slot 0 = the columns below S)
                                    best = (best.c < 0) ? x : argr_min(best, x);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);     // fold column by column: CH candidate records at once would spill
                        }
                    }
                }
                LSA_T(2);                           // scan (thread 0's own columns)
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    ArgR y;
                    y.v = __shfl_xor(best.v, o, 64);
                    y.r4c = __shfl_xor(best.r4c, o, 64);
                    y.it = __shfl_xor(best.it, o, 64);
                    y.c = __shfl_xor(best.c, o, 64);
                    if (y.c >= 0) best = (best.c < 0) ? y : argr_min(best, y);
                }
                __syncthreads();
                if ((tid & 63) == 0) sh[tid >> 6] = best;
                __syncthreads();
                LSA_T(3);                           // wave reduce + waiting for the slowest wave's scan
                ArgR r;                               // the 16 wave minima: a 4-step shuffle tree in wave 0 (argr_min is a total order
                r.v = INFINITY; r.r4c = 0; r.it = INT_MAX; r.c = -1;     // -- scan positions are unique -- so any tree gives the same record)
                if (tid < 64) {
                    if (tid < NT / 64) r = sh[tid];
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) {
                        ArgR y;
                        y.v = __shfl_xor(r.v, o, 64);
                        y.r4c = __shfl_xor(r.r4c, o, 64);
                        y.it = __shfl_xor(r.it, o, 64);
                        y.c = __shfl_xor(r.c, o, 64);
                        if (y.c >= 0) r = (r.c < 0) ? y : argr_min(r, y);
                    }
                }
                if (tid == 0) {
                    if (r.c < 0 || r.v == INFINITY) {
                        s_fail = 1; s_sink = -2; s_win = -1;
                    } else {
                        s_win = r.c;                                   // SC[r.c] = 1: the owner sets its bit below
                        const int k = s_nvis;                          // removals so far: remaining[--n] sits at shadow slot k
                        const int idx = r.c < S ? s_pos[r.c] : Mc - 1 - r.c, last = s_rem[k];
                        const int q = Mc - 1 - idx;                    // remaining[idx] = last (positions outside the shadow are
                        if (q < S) s_rem[q] = last;                    //   never read again in this search and are restored after it)
                        s_pos[last] = idx;                             // last < S always: the tail holds low columns only
                        s_vis[k] = r.c; s_vq[k] = q < S ? q : -1; s_vlast[k] = last; s_vspc[k] = r.v;
                        s_nvis = k + 1;
                        if (r.r4c == -1) s_sink = r.c;
                        else { s_row[nrow] = r.r4c; s_rowmin[nrow] = r.v; s_nrow = nrow + 1; }     // next scan: row4col[j], minVal
                    }
                }
                __syncthreads();
                LSA_T(4);                           // thread 0's bookkeeping
                const int win = s_win;
                if (win >= 0 && (win & (NT - 1)) == tid) scbits |= 1u << (win >> 10);
                if (s_sink != -1) break;
            }
            if (s_fail) break;
            const int nvis = s_nvis, nrow = s_nrow;
            const double minval = s_vspc[nvis - 1];                   // minVal of the search = the sink's shortest-path cost
            if (tid == 0) {
                // augment along the path -- BEFORE the duals move: path[j] is re-derived from the duals the search ran on.
                // Column j was selected as the jk-th winner, after scans 0..jk: path[j] = the first of those rows attaining
                // min_m ((minval_m + cost[i_m][j]) - u[i_m]) - v[j]  (the stored form updates on a strict `<` only)
                int j = s_sink;
                while (true) {
                    int jk = 0;
                    for (int k = 0; k < nvis; ++k) if (s_vis[k] == j) { jk = k; break; }      // every path column was visited
                    const double vj = v[j];
                    const size_t oj = (size_t)cols[j];
                    double sbest = INFINITY;
                    int i = cur;
                    for (int m = 0; m <= jk; ++m) {
                        const int im = s_row[m];
                        const double r = ((s_rowmin[m] + (double)costT[(size_t)im * M + oj]) - s_u[im]) - vj;
                        if (r < sbest) { sbest = r; i = im; }
                    }
                    row4col[j] = i;
                    const int t = s_c4r[i];
                    s_c4r[i] = j;
                    j = t;
                    if (i == cur) break;
                }
            }
            __syncthreads();
            // u[cur] += minVal; u[i] += minVal - shortestPathCosts[col4row[i]] for the other scanned rows: row s_row[m] (m >= 1)
            // was reached through the (m-1)-th visited column, whose cost is s_vspc[m - 1]
            if (tid < nrow) {
                const int i = s_row[tid];
                s_u[i] += (tid == 0) ? minval : minval - s_vspc[tid - 1];
            }
            if (tid < nvis) {            // v[j] -= minVal - shortestPathCosts[j] for the scanned columns only
                const int c = s_vis[tid];
                v[c] -= minval - s_vspc[tid];
            }
            for (int k = 0; k < nvis; ++k) {     // owners refresh their flags: a visited column now has v != 0 and may be assigned
                const int c = s_vis[k];
                if ((c & (NT - 1)) == tid) {
                    vbits |= 1u << (c >> 10);
                    if (row4col[c] != -1) asgbits |= 1u << (c >> 10);
                }
            }
            if (tid < nvis) {            // put the scan order shadows back
                if (s_vq[tid] >= 0) s_rem[s_vq[tid]] = s_vq[tid];
            }
            __syncthreads();
            if (tid < nvis) s_pos[s_vlast[tid]] = Mc - 1 - s_vlast[tid];
            __syncthreads();
            LSA_T(5);                               // dual updates, augmentation, flag refresh, shadow restore
        }
        if (s_fail) break;
        for (int i = tid; i < G; i += NT) {
            const int j = cols[s_c4r[i]];
            gt_inds[j] = i + 1;
            active[j] = 0;
        }
        __syncthreads();
    }
    if (tid == 0 && status) status[b] = s_fail;
}

extern "C" int cpr_lsa_topk(const float* costT, const int* m_of, const int* g_of, const long long* cost_off,
                            const long long* col_off, const long long* row_off, int num_problems, int topk,
                            long long* gt_inds, double* ws_v, double* ws_spc, int* ws_path, int* ws_row4col,
                            unsigned char* ws_sc, unsigned char* ws_active, int* ws_cols, int* ws_remaining,
                            int* ws_pos, double* ws_u, int* ws_col4row, unsigned char* ws_sr, int* status, int max_cols,
                            int max_rows, hipStream_t stream) {
    CPR_CHECK_ARG(num_problems >= 0 && topk >= 1);
    if (num_problems == 0) return CPR_OK;
    CPR_CHECK_ARG(costT && m_of && g_of && cost_off && col_off && row_off && gt_inds && ws_v && ws_spc && ws_path &&
                  ws_row4col && ws_sc && ws_active && ws_cols && ws_remaining && ws_pos && ws_u && ws_col4row && ws_sr);
    // max_cols / max_rows: [host] upper bounds of the problems' sizes (the per-problem tables live on the device); the stateless
    // kernel serves max_cols <= 32768, anything else (or max_cols <= 0 = unknown) the memory-resident one.  Bit-identical results.
    const bool reg_ok = max_cols > 0 && max_cols <= 32 * 1024 && max_rows > 0 && max_rows + 1 <= LSA_MAX_VIS &&
                        (long long)max_cols * max_rows * 4 < (1ll << 31);
#define LSA_REG(K)                                                                                                           \
    hipLaunchKernelGGL((lsa_topk_reg_kernel<K>), dim3(num_problems), dim3(1024), 0, stream, costT, m_of, g_of, cost_off,     \
                       col_off, row_off, topk, gt_inds, ws_v, ws_row4col, ws_active, ws_cols, status)
    if (reg_ok && max_cols <= 8 * 1024) LSA_REG(8);
    else if (reg_ok && max_cols <= 16 * 1024) LSA_REG(16);
    else if (reg_ok && max_cols <= 25 * 1024) LSA_REG(25);
    else if (reg_ok) LSA_REG(32);
    else
        hipLaunchKernelGGL(lsa_topk_kernel, dim3(num_problems), dim3(1024), 0, stream, costT, m_of, g_of, cost_off,
                           col_off, row_off, topk, gt_inds, ws_v, ws_spc, ws_path, ws_row4col, ws_sc, ws_active, ws_cols,
                           ws_remaining, ws_pos, ws_u, ws_col4row, ws_sr, status);
#undef LSA_REG
    CPR_LAUNCH_STATUS();
}

extern "C" int cpr_version(void) { return 1; }
