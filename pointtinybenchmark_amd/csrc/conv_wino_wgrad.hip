// Weight gradient of a 3x3 / stride 1 / pad 1 convolution as fused Winograd F(2x2,3x3) on the fp32 matrix cores.
//
// The reference gets this from torch autograd (cuDNN's backward-filter, which uses Winograd for exactly these shapes) behind
// mmcv ConvModule in the CPR head towers and the FPN output conv (T/mmdet/models/point/dense_heads/cpr_head.py:1033-1043,
// T/mmdet/models/necks/fpn.py:190-194); csrc/conv_wgrad.hip is the direct form (0.84 of the fp32 MFMA peak), this file does
// 2.25x fewer multiplies -- the adjoint of csrc/conv_wino.hip with respect to the weights:
//   dU_f[ci][co] = sum over Winograd tiles t of  V_f[t][ci] * Z_f[t][co],   V = B^T d B (4x4 input patch),  Z = A dY A^T (2x2 dy tile)
//   dg = G^T dU G   (3x3 taps from the 4x4 frequencies)
// i.e. 16 GEMMs with M = Cin, N = Cout and K = tiles (N * H/2 * W/2: hundreds of thousands) -> split-K over workgroups into
// a partial buffer, then one small kernel reduces the slices and applies G.
//
// One workgroup = 512 threads = 8 waves (one per CU), a (64 ci) x (64 co) block of all 16 frequencies = the forward kernel's
// accumulator layout (wave (i, h): frequency row i, co half h: 4 x [64 ci x 32 co] = 128 registers) and its MFMA loop,
// fragment layout and swizzle verbatim, with "tile" -> ci rows and "cout" -> co rows; K chunk = 8 tiles = one strip of
// 2 x 16 output pixels (patch 4 x 18 pixels of x, 2 x 16 pixels of dy), a workgroup walks the strips of its K slice: a
// contiguous run of the strips in (image, tile row, strip) order -- one pipeline across image boundaries, where only the fused
// GroupNorm affine of x (4 channels per thread) is re-fetched.  ceil(CUs / blocks) slices: one workgroup per CU and a
// partial buffer of 16 x [16][Cin][Cout] instead of one slice per image (round 2: 64 slices, 268 MB per head-layer launch).
// Staging per chunk, all waves alike: G global -> registers (x patch 72 pixels x 64 channels = 3 x 16 B per thread, dy 32 x 64 =
// 1 x 16 B), R registers -> raw LDS (affine + ReLU of x here, once per element; pixel stride 68 floats so that the
// transform's reads are conflict free), T raw -> V / Z (thread = (tile of the strip, channel): x 16 reads + 32 adds + 16 writes,
// dy 4 reads + 12 adds + 16 writes).  The raw buffers are single (LDS is full: 128 KB of V/Z + 28 KB raw), hence two barriers per
// chunk: T (reads raw, writes V/Z of chunk c+1) | barrier | R (writes raw of chunk c+2), G (requests c+3) | barrier.
#include <type_traits>
#include "common.h"

struct WinoWgradParams {
    const float* dy;      // (N,H,W,Cout) NHWC
    const float* x;       // (N,H,W,Cin) NHWC
    const float* in_a;    // [N][Cin] or null: x is read as relu?(x * a + b)
    const float* in_b;
    float* part;          // [slices][16][Cin][Cout]
    int N, H, W, Cin, Cout, in_relu;
    int TY, SX;           // tile rows (ceil(H/2)), strips per tile row (ceil(W/16))
    int tilesCi, tilesCo;
    int slices, spp, total;   // K split: slices of spp strips out of N * TY * SX
};

constexpr int WWBUF = 16 * 64 * 8;    // floats in one V or Z chunk image (32 KB)
constexpr int WPS = 68;               // raw pixel stride (floats): 64 channels + 4, 8 tiles x 2 pixels apart -> 8 distinct bank groups
constexpr int WRAWX = 72 * WPS;       // 4 x 18 patch pixels
constexpr int WRAWD = 32 * WPS;       // 2 x 16 dy pixels

template <bool XF>
__global__ __launch_bounds__(512, 1) void conv_wino_wgrad_kernel(WinoWgradParams p) {
    __shared__ __attribute__((aligned(16))) float smem[4 * WWBUF + WRAWX + WRAWD + 128];   // V0 V1 Z0 Z1 rawx rawd [a | b]
    float* Vs = smem;
    float* Zs = smem + 2 * WWBUF;
    float* Rx = smem + 4 * WWBUF;
    float* Rd = Rx + WRAWX;
    float* ABs = Rd + WRAWD;

    // block -> (slice, ci tile, co tile): the tilesCi * tilesCo blocks of one slice are consecutive (they re-read the same
    // strips of x and dy: one L2), XCD-aware as in the forward kernel
    const int nblk = p.tilesCi * p.tilesCo;
    const int T = p.slices * nblk;
    const int per = (T + 7) >> 3;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile >= T) return;
    const int slice = tile / nblk, blk = tile - slice * nblk;
    const int tci = blk / p.tilesCo, tco = blk - tci * p.tilesCo;
    const int g0 = slice * p.spp;                          // first strip of this slice
    const int nk = min(p.spp, p.total - g0);               // chunks (strips) of this slice, >= 1
    const int spi = p.TY * p.SX;                           // strips per image
    const int ci0 = tci * 64, co0 = tco * 64;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 3, nh = wave >> 2;   // MFMA role: frequency row, co half

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (int)((size_t)p.N * p.H * p.W * p.Cout * 4), 0x00020000);

    // ---- G / R roles.  x unit u = tid + 512 k (k < 3, u < 1152): patch pixel u >> 4 = (row r, column c) of 4 x 18, channels
    // ci0 + 4 (u & 15) .. +3.  dy unit = tid: pixel tid >> 4 = (row a, column c) of 2 x 16, channels co0 + 4 (tid & 15) .. +3.
    int xr[3], xc[3], xrel[3], xwr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int u = tid + 512 * k, px = u >> 4;
        xr[k] = px / 18;
        xc[k] = px - xr[k] * 18;
        xrel[k] = ((xr[k] * p.W + xc[k]) * p.Cin + (u & 15) * 4) * 4;   // bytes from the patch origin (2 ty - 1, 16 sx - 1)
        xwr[k] = px * WPS + (u & 15) * 4;
    }
    const bool unit2 = tid + 1024 < 1152;          // third x unit exists (waves 0, 1)
    const bool unit2_wave = wave < 2;              // scalar
    const int da = tid >> 8, dc = (tid >> 4) & 15;
    const int drel = ((da * p.W + dc) * p.Cout + (tid & 15) * 4) * 4;
    const int dwr = (tid >> 4) * WPS + (tid & 15) * 4;
    const float relu_floor = (XF && p.in_relu) ? 0.f : -INFINITY;
    // XF: the affine of the image whose chunk sits in the staging registers -- the same 4 channels for all of a thread's
    // units; re-fetched (two 16-byte loads, L2) when the slice crosses into the next image
    f32x4 xa4 = {1.f, 1.f, 1.f, 1.f}, xb4 = {0.f, 0.f, 0.f, 0.f};
    int img_tab = -1, img_regs = 0;   // scalars: image of (xa4, xb4) / of the chunk in the staging registers
    (void)ABs;

    f32x4 sx[3], sd;            // staging registers of the chunk in flight
    unsigned okbits = 0;        // bit k: x unit k of that chunk is inside the image (XF: padding stays 0 after the affine)
    auto g_x = [&](int k_chunk, int k) {      // G: request x unit k of chunk k_chunk (clamped: the pipeline runs two chunks past the end)
        if (k == 2 && !unit2_wave) return;
        const int g = g0 + (k_chunk < nk ? k_chunk : nk - 1);
        const int n = g / spi, rs = g - n * spi;
        const int tyr = rs / p.SX, sxi = rs - tyr * p.SX;
        if (k == 0) img_regs = n;
        const int iy0 = 2 * tyr - 1, ix0 = 16 * sxi - 1;
        const int xbase = (((n * p.H + iy0) * p.W + ix0) * p.Cin + ci0) * 4;        // may be "negative": only used when in range
        const bool ok = (k < 2 || unit2) & ((unsigned)(iy0 + xr[k]) < (unsigned)p.H) & ((unsigned)(ix0 + xc[k]) < (unsigned)p.W);
        if (k == 0) okbits = 0;
        okbits |= (ok ? 1u : 0u) << k;
        sx[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? xbase + xrel[k] : (int)0x80000000, 0, 0));
    };
    auto g_d = [&](int k_chunk) {             // G: request the dy unit
        const int g = g0 + (k_chunk < nk ? k_chunk : nk - 1);
        const int n = g / spi, rs = g - n * spi;
        const int tyr = rs / p.SX, sxi = rs - tyr * p.SX;
        const bool okd = (2 * tyr + da < p.H) & (16 * sxi + dc < p.W);
        const int dbase = (((n * p.H + 2 * tyr) * p.W + 16 * sxi) * p.Cout + co0) * 4;
        sd = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, okd ? dbase + drel : (int)0x80000000, 0, 0));
    };
    auto g_all = [&](int k_chunk) { g_x(k_chunk, 0); g_x(k_chunk, 1); g_x(k_chunk, 2); g_d(k_chunk); };
    auto r_x = [&](int k) {                  // R: x unit k -> raw (affine + ReLU of the producer's GroupNorm; padding stays 0)
        if (k == 2 && !unit2_wave) return;
        if (XF && k == 0 && img_regs != img_tab) {   // wave-uniform, once per image of the slice
            img_tab = img_regs;
            xa4 = *reinterpret_cast<const f32x4*>(p.in_a + (size_t)img_tab * p.Cin + ci0 + (tid & 15) * 4);
            xb4 = *reinterpret_cast<const f32x4*>(p.in_b + (size_t)img_tab * p.Cin + ci0 + (tid & 15) * 4);
        }
        f32x4 v = sx[k];
        if (XF) {
            v = v * xa4 + xb4;
            v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor); v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
            if (!((okbits >> k) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (k < 2 || unit2) *reinterpret_cast<f32x4*>(Rx + xwr[k]) = v;
    };
    auto r_d = [&]() { *reinterpret_cast<f32x4*>(Rd + dwr) = sd; };

    // ---- T role: thread = (tile t of the strip, channel), wave = channel block of 8.  ds_read_b32 serves lanes 0-31 and 32-63
    // in one cycle each over 32 banks: tiles are 2 pixels = 136 floats = 8 banks apart, so a 32-lane half holds FOUR tiles
    // (banks 0, 8, 16, 24) x EIGHT channels (8 consecutive banks each) -- all 32 banks once.  (Round 2 had eight tiles x four
    // channels per half: tiles t and t + 4 on the same banks, every transform read 2-way conflicted -- 262 M conflict cycles
    // per head-layer launch, profiles/round2_pmc_wino_wgrad_kernel.json.)
    const int tt = (tid & 3) | (((tid >> 5) & 1) << 2), chl = wave * 8 + ((tid >> 2) & 7);
    const float* tx_rd = Rx + (2 * tt) * WPS + chl;        // + (r * 18 + s) * WPS
    const float* td_rd = Rd + (2 * tt) * WPS + chl;        // + (a * 16 + b) * WPS
    const int vz_wr = chl * 8 + 4 * ((tt >> 2) ^ ((chl >> 3) & 1)) + (tt & 3);   // this thread's float in a V / Z row (row = channel, k = tile)
    float d[16], e[4];
    auto tx_read = [&](int r) {
#pragma unroll
        for (int s = 0; s < 4; ++s) d[r * 4 + s] = tx_rd[(r * 18 + s) * WPS];
    };
    auto tx_row = [&](int buf, int i) {      // row i of B^T d, then the column pass: frequencies (i, 0..3)
        float t[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float d0 = d[s], d1 = d[4 + s], d2 = d[8 + s], d3 = d[12 + s];
            t[s] = i == 0 ? d0 - d2 : i == 1 ? d1 + d2 : i == 2 ? d2 - d1 : d1 - d3;
        }
        float* dst = Vs + buf * WWBUF + (i * 4) * 512 + vz_wr;
        dst[0 * 512] = t[0] - t[2];
        dst[1 * 512] = t[1] + t[2];
        dst[2 * 512] = t[2] - t[1];
        dst[3 * 512] = t[1] - t[3];
    };
    auto td_read = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) e[a * 2 + b] = td_rd[(a * 16 + b) * WPS];
    };
    auto td_rows = [&](int buf, int half2) {  // Z = A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]: rows 2 half2, 2 half2 + 1
        // column pass first: c[a][j] = sum_b dY[a][b] A[j][b]
        float c[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            c[a][0] = e[a * 2];
            c[a][1] = e[a * 2] + e[a * 2 + 1];
            c[a][2] = e[a * 2] - e[a * 2 + 1];
            c[a][3] = -e[a * 2 + 1];
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = half2 * 2 + ii;
            float* dst = Zs + buf * WWBUF + (i * 4) * 512 + vz_wr;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                dst[j * 512] = i == 0 ? c[0][j] : i == 1 ? c[0][j] + c[1][j] : i == 2 ? c[0][j] - c[1][j] : -c[1][j];
        }
    };

    f32x16 acc[4][2];   // [frequency column j][ci block]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][tb][r] = 0.f;

    // ---- MFMA operand fragments: A[m = lane & 31][k = lane >> 5] = V row (ci), B = Z row (co)
    const int l31 = lane & 31, half = lane >> 5;
    const float* a_lds = Vs + (wi * 4 * 64 + l31) * 8 + 4 * (half ^ ((l31 >> 3) & 1));
    const float* b_lds = Zs + (wi * 4 * 64 + nh * 32 + l31) * 8 + 4 * (half ^ ((l31 >> 3) & 1));
    f32x4 fa0[2], fb0, fa1[2], fb1;
#define WFRAG(FA, FB, buf, j, z)                                                                                    \
    do {                                                                                                            \
        if ((z) < 2) FA[(z) < 2 ? (z) : 0] = *reinterpret_cast<const f32x4*>(a_lds + (buf) * WWBUF + (j) * 512 + ((z) < 2 ? (z) : 0) * 256); \
        else FB = *reinterpret_cast<const f32x4*>(b_lds + (buf) * WWBUF + (j) * 512);                               \
    } while (0)
#define WMFMA(FA, FB, j, q) \
    acc[j][(q) & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[(q) & 1][(q) >> 1], FB[(q) >> 1], acc[j][(q) & 1], 0, 0, 0)

    // prologue: V/Z[0] = chunk 0, raw = chunk 1, registers = chunk 2
    g_all(0);
    r_x(0); r_x(1); r_x(2); r_d();
    g_all(1);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) tx_read(r);
    td_read();
#pragma unroll
    for (int i = 0; i < 4; ++i) tx_row(0, i);
    td_rows(0, 0); td_rows(0, 1);
    __syncthreads();                 // every T read of raw is done
    r_x(0); r_x(1); r_x(2); r_d();
    g_all(2);
    __syncthreads();
#pragma unroll
    for (int z = 0; z < 3; ++z) WFRAG(fa0, fb0, 0, 0, z);

    // One chunk c (V/Z buffers buf = c & 1); every staging instruction sits behind one of the wave's own MFMAs:
    //   j = 0: MFMAs | prefetch the j = 1 fragments | T reads of chunk c+1 (raw, complete since barrier B)
    //   j = 1: MFMAs | prefetch j = 2 | T rows of chunk c+1 -> V/Z[buf^1]
    //   barrier A (raw is free; V/Z[buf^1] visible)
    //   j = 2: MFMAs | prefetch j = 3 | R chunk c+2: registers -> raw
    //   j = 3: MFMAs | prefetch j = 0 of chunk c+1 | G chunk c+3
    //   barrier B (raw visible; every fragment read of V/Z[buf] is done)
    for (int c = 0; c < nk; ++c) {
        const int buf = c & 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            WMFMA(fa0, fb0, 0, q);
            if (q < 3) WFRAG(fa1, fb1, buf, 1, q);
            if (q >= 3 && q < 7) tx_read(q - 3);
            if (q == 7) td_read();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            WMFMA(fa1, fb1, 1, q);
            if (q < 3) WFRAG(fa0, fb0, buf, 2, q);
            if (q < 4) tx_row(buf ^ 1, q);
            if (q == 4 || q == 5) td_rows(buf ^ 1, q - 4);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0); global loads stay in flight
        __builtin_amdgcn_s_barrier();         // A
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            WMFMA(fa0, fb0, 2, q);
            if (q < 3) WFRAG(fa1, fb1, buf, 3, q);
            if (q == 3) r_x(0);
            if (q == 4) r_x(1);
            if (q == 5) r_x(2);
            if (q == 6) r_d();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            WMFMA(fa1, fb1, 3, q);
            if (q < 3) WFRAG(fa0, fb0, buf ^ 1, 0, q);
            if (q == 1) g_x(c + 3, 0);          // one request every other MFMA: a burst stalls the wave AT the loads
            if (q == 3) g_x(c + 3, 1);
            if (q == 5) g_x(c + 3, 2);
            if (q == 7) g_d(c + 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();         // B
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
#undef WFRAG
#undef WMFMA

    // ---- partials: D layout of a 32x32 block: col = lane & 31 (co), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (ci)
    const int co = co0 + nh * 32 + l31;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float* dst = p.part + ((size_t)(slice * 16 + wi * 4 + j) * p.Cin + ci0) * p.Cout + co;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                dst[(size_t)ci * p.Cout] = acc[j][tb][r];
            }
    }
}

// slices reduced in fp32, then dg = G^T dU G, written (or accumulated) into grad_w [Cout][Cin][3][3]
__global__ void wino_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ grad, int slices, int Cin,
                                         int Cout, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int co = idx % Cout, ci = idx / Cout;      // consecutive threads -> consecutive co: coalesced partial reads
    float u[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) u[f] = 0.f;
    for (int s = 0; s < slices; ++s)
#pragma unroll
        for (int f = 0; f < 16; ++f) u[f] += part[((size_t)(s * 16 + f) * Cin + ci) * Cout + co];
    // t = G^T dU (3 x 4), G^T = [[1, .5, .5, 0], [0, .5, -.5, 0], [0, .5, .5, 1]]
    float t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = u[j] + 0.5f * (u[4 + j] + u[8 + j]);
        t[1][j] = 0.5f * (u[4 + j] - u[8 + j]);
        t[2][j] = 0.5f * (u[4 + j] + u[8 + j]) + u[12 + j];
    }
    float* g = grad + ((size_t)co * Cin + ci) * 9;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float g0 = t[k][0] + 0.5f * (t[k][1] + t[k][2]);
        const float g1 = 0.5f * (t[k][1] - t[k][2]);
        const float g2 = 0.5f * (t[k][1] + t[k][2]) + t[k][3];
        if (accumulate) { g[k * 3] += g0; g[k * 3 + 1] += g1; g[k * 3 + 2] += g2; }
        else { g[k * 3] = g0; g[k * 3 + 1] = g1; g[k * 3 + 2] = g2; }
    }
}

// K split: slices of consecutive strips, (CUs of an MI355X) / blocks of them: one resident workgroup per CU and slice
static void wino_wgrad_split(int N, int H, int W, int Cin, int Cout, int* slices, int* spp) {
    const long long total = (long long)N * ((H + 1) / 2) * ((W + 15) / 16);
    const int blocks = (Cin / 64) * (Cout / 64);
    long long want = (256 + blocks - 1) / blocks;
    if (want > total) want = total;
    const long long per = (total + want - 1) / want;
    *spp = (int)per;
    *slices = (int)((total + per - 1) / per);
}

// C-ABI ------------------------------------------------------------------------------------------
extern "C" int cpr_conv3x3_wino_wgrad_workspace(int N, int H, int W, int Cin, int Cout) {
    CPR_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 64 == 0);
    int slices, spp;
    wino_wgrad_split(N, H, W, Cin, Cout, &slices, &spp);
    const long long n = (long long)slices * 16 * Cin * Cout;
    return n < (1ll << 31) ? (int)n : CPR_ERR_UNSUPPORTED;
}
static int wino_wgrad_launch(const float* dy, const float* x, const float* in_a, const float* in_b, float* grad_w,
                             float* ws, int N, int H, int W, int Cin, int Cout, int in_relu, int accumulate,
                             hipStream_t stream) {
    CPR_CHECK_ARG(dy && x && grad_w && ws && N > 0 && H > 0 && W > 0);
    CPR_CHECK_ARG(Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 64 == 0);
    if (in_a) CPR_CHECK_ARG(in_b != nullptr);
    if ((long long)N * H * W * Cin * 4 >= (1ll << 31) || (long long)N * H * W * Cout * 4 >= (1ll << 31)) return CPR_ERR_UNSUPPORTED;
    WinoWgradParams p;
    p.dy = dy; p.x = x; p.in_a = in_a; p.in_b = in_b; p.part = ws;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.in_relu = in_relu;
    p.TY = (H + 1) / 2; p.SX = (W + 15) / 16;
    wino_wgrad_split(N, H, W, Cin, Cout, &p.slices, &p.spp);
    p.total = N * p.TY * p.SX;
    p.tilesCi = Cin / 64; p.tilesCo = Cout / 64;
    const long long nfl = (long long)p.slices * 16 * Cin * Cout;
    if (nfl >= (1ll << 31)) return CPR_ERR_UNSUPPORTED;
    const long long T = (long long)p.slices * p.tilesCi * p.tilesCo;
    const int grid = (int)((T + 7) / 8 * 8);
    if (in_a) hipLaunchKernelGGL((conv_wino_wgrad_kernel<true>), dim3(grid), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((conv_wino_wgrad_kernel<false>), dim3(grid), dim3(512), 0, stream, p);
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(cdiv(Cin * Cout, 256)), dim3(256), 0, stream, ws, grad_w, p.slices,
                       Cin, Cout, accumulate);
    CPR_LAUNCH_STATUS();
}

// >= 2 GiB maps: balanced chunks of whole images; chunks after the first accumulate into grad_w.  The workspace of the whole
// batch serves every chunk (a sub-batch never needs more slices than the whole batch) -- checked per launch.
extern "C" int cpr_conv3x3_wino_wgrad(const float* dy, const float* x, const float* in_a, const float* in_b, float* grad_w,
                                      float* ws, int N, int H, int W, int Cin, int Cout, int in_relu, int accumulate,
                                      hipStream_t stream) {
    CPR_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 64 == 0);
    const int per = cpr_images_per_launch(N, (long long)H * W * cpr_max2(Cin, Cout) * 4);
    if (per <= 0) return CPR_ERR_UNSUPPORTED;
    int slices_all, spp;
    wino_wgrad_split(N, H, W, Cin, Cout, &slices_all, &spp);
    for (int n0 = 0; n0 < N; n0 += per) {
        const int n = N - n0 < per ? N - n0 : per;
        int slices;
        wino_wgrad_split(n, H, W, Cin, Cout, &slices, &spp);
        if (slices > slices_all) return CPR_ERR_UNSUPPORTED;   // workspace was sized for the whole batch
        const int rc = wino_wgrad_launch(dy + (size_t)n0 * H * W * Cout, x + (size_t)n0 * H * W * Cin,
                                         in_a ? in_a + (size_t)n0 * Cin : nullptr, in_b ? in_b + (size_t)n0 * Cin : nullptr,
                                         grad_w, ws, n, H, W, Cin, Cout, in_relu, n0 == 0 ? accumulate : 1, stream);
        if (rc != CPR_OK) return rc;
    }
    return CPR_OK;
}
