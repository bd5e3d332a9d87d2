// 3x3 / stride 1 / pad 1 convolution as fused Winograd F(2x2, 3x3), fp32 on the matrix cores -- the TWO-WORKGROUPS-PER-CU form
// (round 4).  Same mathematics and call sites as conv_wino.hip (T/mmdet/models/point/dense_heads/cpr_head.py:1033-1043,
// T/mmdet/models/necks/fpn.py:190-194, the stride-1 3x3 of every bottleneck T/mmdet/models/backbones/resnet.py:630-645).
//
// Why a second form: conv_wino.hip runs ONE 8-wave workgroup per CU (156 KB of LDS, 64 tiles x 64 couts x 16 frequencies of
// accumulators) and sits at 0.785 of the fp32 MFMA peak with its K loop at 0.95 -- 13 % of every tile passes with the matrix
// pipe idle (output transform through LDS, the next tile's prologue, wave skew) and nothing else is resident to
// use it.  Here a workgroup is HALF of that -- 4 waves, a region of 8 x 16 output pixels = 4 x 8 Winograd tiles (GEMM M = 32),
// 64 output channels, the same 128 accumulator registers per wave -- and needs < 80 KB of LDS, so TWO are resident per CU,
// half a tile out of step (the launcher staggers the second wave of workgroups): while one is in its epilogue / prologue the
// other one's MFMAs own the pipe.  Half-size tiles also quantise better on small launches (B = 2: 1600 tiles over 512 slots).
//
//   wave w: frequency row i = w (f = 4w .. 4w+3), all 32 tiles x 64 couts: acc[j][nb] = 8 x f32x16
//   K loop over chunks of FOUR input channels (16-byte rows: one ds_read_b64 per operand block = 2 MFMA k-steps; lanes < 32
//   take floats 0-1, lanes >= 32 floats 2-3 of a row, both operands alike, so MFMA q multiplies channels {q, q + 2}):
//     V[2][16][32 tiles][4]  transformed input (8 KB each)       U[3][16][64 couts][4]  transformed weights (16 KB each, ring)
//     raw[4][10 rows][18 px][4]  the chunk's input patch (ring of four 3 KB slots)
//   ALL global -> LDS traffic of the loop is LDS-DMA (buffer_load_dwordx4 ... lds, inline asm, counted by hand): the weight
//   image of chunk c+2 and the patch of chunk c+4 are requested at the top of chunk c; `s_waitcnt vmcnt(own requests of this
//   chunk)` + ONE barrier per chunk make everything older visible.  Per chunk c (buffers b = c & 1), every wave alike:
//     MFMAs of chunk c (16)  |  T: raw[(c+1) & 3] -> B^T d B -> V[b ^ 1] (thread = (tile, channel, half of the frequency rows):
//     12 LDS reads, 16 adds, 8 LDS writes)  |  A (fused-affine instances): the producer's GroupNorm apply (+ReLU) IN PLACE on
//     raw[(c+2) & 3], one 16-byte unit per thread, padding pixels stay 0
//   Epilogue as in conv_wino.hip: R[i][b] = sum_j M[i][j] A[j][b] in registers, exchanged through LDS (64 KB, aliasing the loop
//   buffers), Y[a][b] = sum_i A[i][a] R[i][b], scale / bias / ReLU, 16-byte stores, optional GroupNorm (sum, sumsq) per region
//   (one slot per 8 x 16 region: cpr_conv3x3_wino32_slots tells the caller how many).
// The accumulation order differs from conv_wino.hip's (4- vs 8-channel chunks, other region sums), so the two kernels agree to
// fp32 rounding, not bit for bit; which one serves a layer is a function of the layer's shape alone, never of the batch size
// (an image of a big batch equals its single-image run bit for bit either way).
#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct Wino32Params {
    const float* in;      // NHWC, or channel-blocked [N][Cin/8][H][W][8] (layout bit CPR_WINO_IN_B8)
    const float* u;       // packed transformed weights, see wino32_pack_kernel
    float* out;           // NHWC, or channel-blocked [N][Cout/8][H][W][8]
    const float* scale;   // [Cout] or null
    const float* bias;    // [Cout] or null
    float* gn_part;       // [regions][Cout][2] per-region per-channel (sum, sumsq) of the output, or null
    const float* in_a;    // [N][Cin] or null: the input is read as relu?(x * a + b)
    const float* in_b;
    int in_relu, in_b8, out_b8;
    int N, H, W, Cin, Cout, relu, RY, RX, regions, tilesN, nch;   // nch = Cin / 4
    int stagger;          // s_sleep(127) iterations the second wave of workgroups waits before its first tile
};

constexpr int W32_V = 16 * 32 * 4;      // floats in one V buffer (8 KB)
constexpr int W32_U = 16 * 64 * 4;      // floats in one U buffer (16 KB)
constexpr int W32_RAW = 192 * 4;        // floats in one raw slot: 180 units of 16 bytes, padded to 3 wave requests (192 units)
constexpr int W32_XFMAX = 256;          // fused-affine launches keep the image's (a, b) table in LDS (Cin <= 256)
constexpr int W32_LDS = 2 * W32_V + 3 * W32_U + 4 * W32_RAW + 2 * W32_XFMAX;   // 19968 floats = 79872 bytes: two per CU

// weights [Cout][3][3][Cin] (row stride Kpad, the direct kernel's pack) -> U image per (cout tile of 64, chunk of 4 cin):
//   [tn][chunk][f = 4i+j][cout 64][4]
__global__ void wino32_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout, int Kpad) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int ci = idx % Cin, co = idx / Cin;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = w[(size_t)co * Kpad + (a * 3 + b) * Cin + ci];
    float t[4][3];   // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int tn = co >> 6, cl = co & 63, chunk = ci >> 2, k = ci & 3, nch = Cin >> 2;
    float* dst = u + ((size_t)(tn * nch + chunk) * 16) * 256 + cl * 4 + k;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dst[(i * 4 + 0) * 256] = t[i][0];
        dst[(i * 4 + 1) * 256] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
        dst[(i * 4 + 2) * 256] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
        dst[(i * 4 + 3) * 256] = t[i][2];
    }
}

// ABL (measurement build only, results are then WRONG): bit0 = no transform / affine pass, bit1 = no LDS-DMA requests and no waits for
// them, bit2 = no chunk barrier, bit3 = no fragment reads.  ABL = 0 in every product launch.
template <bool XF, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_wino32_kernel(Wino32Params p) {
    __shared__ __attribute__((aligned(16))) float smem[W32_LDS];
    float* Vs = smem;                                   // V[2]
    float* Us = smem + 2 * W32_V;                       // U[3]
    float* Rw = smem + 2 * W32_V + 3 * W32_U;           // raw[4]
    float* ABs = Rw + 4 * W32_RAW;                      // XF: a[256] | b[256] of the current tile's image

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar
    const int l31 = lane & 31, half = lane >> 5;

    // ---- LDS-DMA descriptors (raw SGPR words) and this thread's request geometry
    const size_t in_addr = (size_t)p.in, u_addr = (size_t)p.u;
    const i32x4 rs_in = {(int)(unsigned)in_addr, (int)(unsigned)(in_addr >> 32) & 0xffff,
                         (int)((size_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000};
    const i32x4 rs_u = {(int)(unsigned)u_addr, (int)(unsigned)(u_addr >> 32) & 0xffff,
                        (int)((size_t)p.tilesN * p.nch * W32_U * 4), 0x00020000};
    const int lds_u0 = (int)(unsigned)(size_t)Us, lds_raw0 = (int)(unsigned)(size_t)Rw;
    const int pixbytes = p.in_b8 ? 32 : p.Cin * 4;
    // patch unit of this thread (waves 0-2): u = tid, pixel (py, px) of the 10 x 18 patch, the chunk's 4 channels = 16 bytes
    const int upy = tid / 18, upx = tid - upy * 18;
    const bool unit_live = tid < 180;
    // T role: tile row ty = wave; lane = channel (2 bits) | tile column (3 bits) | half (1 bit): half 0 -> frequency rows 0, 1
    // (patch rows 0-2 of the tile), half 1 -> rows 3, 2 (patch rows 1-3)
    const int tch = lane & 3, ttx = (lane >> 2) & 7;
    const int t_rd = (2 * wave + half) * 72 + (2 * ttx) * 4 + tch;                       // + r * 72 + s * 4
    const int t_wrA = ((half ? 12 : 0) * 32 + wave * 8 + ttx) * 4 + tch;                 // frequency row 0 / 3: + j * 128
    const int t_wrB = ((half ? 8 : 4) * 32 + wave * 8 + ttx) * 4 + tch;                  // frequency row 1 / 2
    // MFMA fragments: A[m = lane & 31][k = lane >> 5] = V row (tile m), B = U row (cout)
    const int a_rd = (wave * 4 * 32 + l31) * 4 + 2 * half;                               // + j * 128
    const int b_rd = (wave * 4 * 64 + l31) * 4 + 2 * half;                               // + j * 256 + nb * 128

    const int T = p.regions * p.tilesN;
    const int per = (T + 7) >> 3;
    const float relu_floor = (XF && p.in_relu) ? 0.f : -INFINITY;

    // the second resident workgroup of every CU starts half a tile late (see the header)
    if ((int)blockIdx.x >= (int)(gridDim.x >> 1)) {
        for (int s = 0; s < p.stagger; ++s) __builtin_amdgcn_s_sleep(127);
    }

    int cur_n = -1;
    for (int vb = blockIdx.x;; vb += gridDim.x) {
        const int tile = (vb & 7) * per + (vb >> 3);
        if (!(vb < per * 8 && tile < T)) break;          // scalar: the end of this workgroup's run
        const int rg = tile / p.tilesN, tn = tile - rg * p.tilesN;
        const int n = rg / (p.RY * p.RX);
        const int rrem = rg - n * p.RY * p.RX;
        const int ry = rrem / p.RX, rx = rrem - ry * p.RX;
        const int oy0 = ry * 8, ox0 = rx * 16, n0 = tn * 64;
        // this thread's patch pixel: byte offset of its 4-channel unit for chunk 0, or out of range (the request returns 0)
        const int iy = oy0 - 1 + upy, ix = ox0 - 1 + upx;
        const bool uok = unit_live & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        const int img_base = p.in_b8 ? (n * (p.Cin >> 3)) * p.H * p.W * 32 : n * p.H * p.W * p.Cin * 4;
        const int voff_a = uok ? img_base + (iy * p.W + ix) * pixbytes : (int)0x80000000;

        // request helpers (hand-counted vmcnt: per chunk every wave issues 4 weight requests, waves 0-2 one patch request more)
        auto dma_u = [&](int chunk, int slot) {
            if (ABL & 2) return;
            const int ck = chunk < p.nch ? chunk : p.nch - 1;          // past the end: clamp (never used, keeps the counts uniform)
            const int soff = (tn * p.nch + ck) * (W32_U * 4);
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                const int piece = (z * 4 + wave) * 1024;
                const int m0v = lds_u0 + slot * (W32_U * 4) + piece;
                asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                             :: "s"(m0v), "v"(piece + lane * 16), "s"(rs_u), "s"(soff) : "memory");
            }
        };
        auto dma_a = [&](int chunk, int slot) {
            if ((ABL & 2) || wave >= 3) return;                                      // scalar branch: 180 units = waves 0-2
            const int ck = chunk < p.nch ? chunk : p.nch - 1;
            // chunk ck = channels 4 ck .. 4 ck + 3: NHWC: 16 bytes further per chunk; blocked: block ck >> 1, half ck & 1
            const int soff = p.in_b8 ? (ck >> 1) * (p.H * p.W * 32) + (ck & 1) * 16 : ck * 16;
            const int m0v = lds_raw0 + slot * (W32_RAW * 4) + wave * 1024;
            asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                         :: "s"(m0v), "v"(voff_a), "s"(rs_in), "s"(soff) : "memory");
        };
        auto wait_older_than_this_chunk = [&]() {                       // everything requested before this chunk's requests has landed
            if (ABL & 2) return;
            if (wave < 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        };
        auto affine_pass = [&](int chunk, int slot) {                   // A: GroupNorm apply (+ReLU) of the producer, in place
            if (!XF || (ABL & 1)) return;
            if (unit_live) {
                const int ck = chunk < p.nch ? chunk : p.nch - 1;
                float* q = Rw + slot * W32_RAW + tid * 4;
                f32x4 v = *reinterpret_cast<const f32x4*>(q);
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ABs + ck * 4);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(ABs + W32_XFMAX + ck * 4);
                v = v * a4 + b4;
                v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor); v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
                if (!uok) v = f32x4{0.f, 0.f, 0.f, 0.f};                // padding must be 0 AFTER the affine
                *reinterpret_cast<f32x4*>(q) = v;
            }
        };
        float e[3][4] = {};                                             // T: three patch rows x four columns of (tile, channel)
        auto t_read = [&](int slot, int r) {
            if (ABL & 1) return;
#pragma unroll
            for (int s = 0; s < 4; ++s) e[r][s] = Rw[slot * W32_RAW + t_rd + r * 72 + s * 4];
        };
        auto t_write = [&](int buf) {
            if (ABL & 1) return;
            // rows of B^T d: half 0: (d0 - d2, d1 + d2) = (e0 - e2, e1 + e2); half 1: (d1 - d3, d2 - d1) = (e0 - e2, e1 - e0)
            float* dA = Vs + buf * W32_V + t_wrA;
            float* dB = Vs + buf * W32_V + t_wrB;
            float ta[4], tb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                ta[s] = e[0][s] - e[2][s];
                tb[s] = e[1][s] + (half ? -e[0][s] : e[2][s]);
            }
            dA[0 * 128] = ta[0] - ta[2]; dA[1 * 128] = ta[1] + ta[2]; dA[2 * 128] = ta[2] - ta[1]; dA[3 * 128] = ta[1] - ta[3];
            dB[0 * 128] = tb[0] - tb[2]; dB[1 * 128] = tb[1] + tb[2]; dB[2 * 128] = tb[2] - tb[1]; dB[3 * 128] = tb[1] - tb[3];
        };

        // ---- prologue: (a, b) table of the image, chunks 0-3 of the patch, weight images 0 and 1; then A(0), A(1), T(0)
        if (XF && n != cur_n) {
            for (int c = tid; c < p.Cin; c += 256) { ABs[c] = p.in_a[n * p.Cin + c]; ABs[W32_XFMAX + c] = p.in_b[n * p.Cin + c]; }
            cur_n = n;
        }
        dma_u(0, 0); dma_u(1, 1);
        dma_a(0, 0); dma_a(1, 1); dma_a(2, 2); dma_a(3, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        affine_pass(0, 0); affine_pass(1, 1);
        if (XF) __syncthreads();
#pragma unroll
        for (int r = 0; r < 3; ++r) t_read(0, r);
        t_write(0);
        __syncthreads();

        f32x16 acc[4][2];   // [frequency column j][cout block nb]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][nb][r] = 0.f;

        for (int c = 0; c < p.nch; ++c) {
            const int b = c & 1;
            const int uslot = c % 3;
            // requests of this chunk: weight image c+2 -> U[(c+2) % 3] (last read in chunk c-1), patch c+4 -> raw[c & 3] (last read
            // by T(c) in chunk c-1)
            dma_u(c + 2, (c + 2) % 3);
            dma_a(c + 4, c & 3);
            const float* va = Vs + b * W32_V + a_rd;
            const float* ub = Us + uslot * W32_U + b_rd;
            f32x2 fa[2], fb[2][2];
            fa[0] = *reinterpret_cast<const f32x2*>(va);
            fb[0][0] = *reinterpret_cast<const f32x2*>(ub);
            fb[0][1] = *reinterpret_cast<const f32x2*>(ub + 128);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cu = j & 1, nx = cu ^ 1;
                // MFMA 0 | prefetch the next frequency's fragments
                acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cu].x, fb[cu][0].x, acc[j][0], 0, 0, 0);
                if (j < 3 && !(ABL & 8)) {
                    fa[nx] = *reinterpret_cast<const f32x2*>(va + (j + 1) * 128);
                    fb[nx][0] = *reinterpret_cast<const f32x2*>(ub + (j + 1) * 256);
                    fb[nx][1] = *reinterpret_cast<const f32x2*>(ub + (j + 1) * 256 + 128);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cu].x, fb[cu][1].x, acc[j][1], 0, 0, 0);
                // staging behind the MFMAs: T reads (j = 0, 1: rows 0-2 of chunk c+1's patch), A pass (j = 2), T writes (j = 3)
                if (j == 0) { t_read((c + 1) & 3, 0); }
                if (j == 1) { t_read((c + 1) & 3, 1); }
                if (j == 2) { t_read((c + 1) & 3, 2); }
                __builtin_amdgcn_sched_barrier(0);
                acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cu].y, fb[cu][0].y, acc[j][0], 0, 0, 0);
                if (j == 1) affine_pass(c + 2, (c + 2) & 3);
                __builtin_amdgcn_sched_barrier(0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cu].y, fb[cu][1].y, acc[j][1], 0, 0, 0);
                if (j == 3) t_write(b ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("" ::: "memory");
            wait_older_than_this_chunk();
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS traffic is done; the new requests stay in flight
            if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        // the clamped requests past the last chunk are still in flight: they must land before the epilogue reuses the LDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- epilogue.  D layout of a 32x32 block: col = lane & 31 (cout), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (tile).
        float* Rs = smem;                                               // R[4 i][2 b][32 tiles][64 couts] = 64 KB
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int trow = (r & 3) + 8 * (r >> 2) + 4 * half;
                const float m0 = acc[0][nb][r], m1 = acc[1][nb][r], m2 = acc[2][nb][r], m3 = acc[3][nb][r];
                Rs[((wave * 2 + 0) * 32 + trow) * 64 + nb * 32 + l31] = m0 + m1 + m2;
                Rs[((wave * 2 + 1) * 32 + trow) * 64 + nb * 32 + l31] = m1 - m2 - m3;
            }
        __syncthreads();
        const int c4 = tid & 15;   // this thread's 4 output channels
        const int co = n0 + c4 * 4;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + co);
        if (p.bias) bi = *reinterpret_cast<const f32x4*>(p.bias + co);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
            p.out, 0, (int)((size_t)p.N * p.H * p.W * p.Cout * 4), 0x00020000);
        f32x4 gs = {0.f, 0.f, 0.f, 0.f}, gq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int it = (tid >> 4) + 16 * k;          // 64 (tile, b) items
            const int t = it & 31, bq = it >> 5;
            const int oy = oy0 + 2 * (t >> 3), ox = ox0 + 2 * (t & 7) + bq;      // GEMM row t = 8 * tile row + tile column
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(Rs + ((0 * 2 + bq) * 32 + t) * 64 + c4 * 4);
            const f32x4 r1 = *reinterpret_cast<const f32x4*>(Rs + ((1 * 2 + bq) * 32 + t) * 64 + c4 * 4);
            const f32x4 r2 = *reinterpret_cast<const f32x4*>(Rs + ((2 * 2 + bq) * 32 + t) * 64 + c4 * 4);
            const f32x4 r3 = *reinterpret_cast<const f32x4*>(Rs + ((3 * 2 + bq) * 32 + t) * 64 + c4 * 4);
            f32x4 y[2] = {r0 + r1 + r2, r1 - r2 - r3};
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                f32x4 v = y[a] * sc + bi;
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                const bool ok = (oy + a < p.H) & (ox < p.W);   // partial regions at the right / bottom edge
                const unsigned off = !ok ? 0x80000000u
                    : p.out_b8 ? (unsigned)((((n * (p.Cout >> 3) + (co >> 3)) * p.H + oy + a) * p.W + ox) * 8 + (co & 7)) * 4u
                               : (unsigned)(((n * p.H + oy + a) * p.W + ox) * p.Cout + co) * 4u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rs_out, (int)off, 0, 0);
                if (p.gn_part && ok) { gs += v; gq += v * v; }
            }
        }
        if (p.gn_part) {
            // per-channel sums of the region's 128 pixels: the 4 lanes of a wave that share c4, then the 4 waves through LDS
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gs[q] += __shfl_xor(gs[q], 16, 64); gs[q] += __shfl_xor(gs[q], 32, 64);
                gq[q] += __shfl_xor(gq[q], 16, 64); gq[q] += __shfl_xor(gq[q], 32, 64);
            }
            __syncthreads();   // all R reads are done
            if (lane < 16) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    smem[(wave * 64 + c4 * 4 + q) * 2 + 0] = gs[q];
                    smem[(wave * 64 + c4 * 4 + q) * 2 + 1] = gq[q];
                }
            }
            __syncthreads();
            if (tid < 64) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) { s += smem[(w4 * 64 + tid) * 2]; q += smem[(w4 * 64 + tid) * 2 + 1]; }
                float* dst = p.gn_part + ((size_t)rg * p.Cout + n0 + tid) * 2;   // one slot per region (regions of an image are contiguous)
                dst[0] = s; dst[1] = q;
            }
        }
        __syncthreads();   // the next tile's prologue overwrites the LDS this epilogue read
    }
}

// C-ABI ------------------------------------------------------------------------------------------
#ifdef CPR_BENCH_HOOKS   // measurement build only (libcprhip_bench.so): loop ablations, workgroups per CU, stagger -- process-global
static int w32_ablate = 0, w32_wg_per_cu = 2, w32_stagger_pct = 100;
extern "C" int cpr_wino32_set_debug(int ablate, int wg_per_cu, int stagger_pct) {
    CPR_CHECK_ARG(ablate >= 0 && ablate <= 15 && wg_per_cu >= 1 && wg_per_cu <= 2 && stagger_pct >= 0 && stagger_pct <= 400);
    w32_ablate = ablate; w32_wg_per_cu = wg_per_cu; w32_stagger_pct = stagger_pct;
    return CPR_OK;
}
#else
constexpr int w32_ablate = 0, w32_wg_per_cu = 2, w32_stagger_pct = 100;
#endif
extern "C" int cpr_wino32_pack_weights(const float* wgt, float* u, int Cin, int Cout, int Kpad, hipStream_t stream) {
    CPR_CHECK_ARG(wgt && u && Cin > 0 && Cout > 0 && Cin % 4 == 0 && Cout % 64 == 0 && Kpad >= 9 * Cin);
    hipLaunchKernelGGL(wino32_pack_kernel, dim3(cdiv(Cin * Cout, 256)), dim3(256), 0, stream, wgt, u, Cin, Cout, Kpad);
    CPR_LAUNCH_STATUS();
}

// GroupNorm-statistics slots (8 x 16 pixel regions) the kernel writes per image
extern "C" int cpr_conv3x3_wino32_slots(int H, int W) { return H > 0 && W > 0 ? ((H + 7) / 8) * ((W + 15) / 16) : CPR_ERR_ARG; }

static int wino32_fwd_launch(const float* in, const float* u, float* out, const float* scale, const float* bias,
                             const float* in_a, const float* in_b, float* gn_part, int N, int H, int W, int Cin,
                             int Cout, int flags, int in_relu, int layout, hipStream_t stream) {
    CPR_CHECK_ARG(in && u && out && N > 0 && H > 0 && W > 0);
    CPR_CHECK_ARG(Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 64 == 0 && Cin >= 16);
    CPR_CHECK_ARG((flags & ~CPR_CONV_RELU) == 0 && (layout & ~3) == 0);
    if (in_a) CPR_CHECK_ARG(in_b && Cin <= W32_XFMAX);
    if ((long long)N * H * W * Cin * 4 >= (1ll << 31) || (long long)N * H * W * Cout * 4 >= (1ll << 31)) return CPR_ERR_UNSUPPORTED;
    Wino32Params p;
    p.in = in; p.u = u; p.out = out; p.scale = scale; p.bias = bias; p.gn_part = gn_part;
    p.in_a = in_a; p.in_b = in_b; p.in_relu = in_relu;
    p.in_b8 = (layout & CPR_WINO_IN_B8) ? 1 : 0; p.out_b8 = (layout & CPR_WINO_OUT_B8) ? 1 : 0;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = flags & CPR_CONV_RELU;
    p.RY = (H + 7) / 8; p.RX = (W + 15) / 16;
    p.regions = N * p.RY * p.RX; p.tilesN = Cout / 64; p.nch = Cin / 4;
    const long long T = (long long)p.regions * p.tilesN;
    if (T >= (1ll << 30)) return CPR_ERR_UNSUPPORTED;
    static int cu_of_device[64] = {};       // immutable hardware property, looked up once per device
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (cu_of_device[dev] == 0) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cu_of_device[dev] = v;
        }
        if (cu_of_device[dev] > 0) ncu = cu_of_device[dev];
    }
    ncu = (ncu + 7) / 8 * 8;
    int grid = (int)((T + 7) / 8 * 8);
    if (grid > w32_wg_per_cu * ncu) grid = w32_wg_per_cu * ncu;    // persistent: two workgroups per CU
    // stagger: half a tile's K loop at two workgroups per CU = nch chunks x 16 MFMAs x ~65 cycles; s_sleep(127) ~ 8128 cycles
    p.stagger = grid > ncu ? (int)(((long long)p.nch * 16 * 65 * w32_stagger_pct / 100 + 8127) / 8128) : 0;
#ifdef CPR_BENCH_HOOKS
#define W32L(A_)                                                                                                      \
    do {                                                                                                              \
        if (in_a) hipLaunchKernelGGL((conv_wino32_kernel<true, A_>), dim3(grid), dim3(256), 0, stream, p);           \
        else hipLaunchKernelGGL((conv_wino32_kernel<false, A_>), dim3(grid), dim3(256), 0, stream, p);               \
    } while (0)
    switch (w32_ablate) {
        case 0: W32L(0); break;
        case 1: W32L(1); break;
        case 2: W32L(2); break;
        case 3: W32L(3); break;
        case 4: W32L(4); break;
        case 7: W32L(7); break;
        case 8: W32L(8); break;
        case 15: W32L(15); break;
        default: return CPR_ERR_UNSUPPORTED;
    }
#undef W32L
#else
    if (in_a) hipLaunchKernelGGL((conv_wino32_kernel<true>), dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv_wino32_kernel<false>), dim3(grid), dim3(256), 0, stream, p);
#endif
    CPR_LAUNCH_STATUS();
}

// Same contract as cpr_conv3x3_wino_fwd (conv_wino.hip) with 8 x 16 pixel regions: gn_part holds cpr_conv3x3_wino32_slots(H, W)
// slots per image; `u` comes from cpr_wino32_pack_weights.  A batch whose maps reach 2 GiB runs as balanced chunks of whole images.
extern "C" int cpr_conv3x3_wino32_fwd(const float* in, const float* u, float* out, const float* scale, const float* bias,
                                      const float* in_a, const float* in_b, float* gn_part, int N, int H, int W, int Cin,
                                      int Cout, int flags, int in_relu, int layout, hipStream_t stream) {
    CPR_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
    const int per = cpr_images_per_launch(N, (long long)H * W * cpr_max2(Cin, Cout) * 4);
    if (per <= 0) return CPR_ERR_UNSUPPORTED;
    const size_t slots = (size_t)((H + 7) / 8) * ((W + 15) / 16);
    for (int n0 = 0; n0 < N; n0 += per) {
        const int n = N - n0 < per ? N - n0 : per;
        const int rc = wino32_fwd_launch(in + (size_t)n0 * H * W * Cin, u, out + (size_t)n0 * H * W * Cout, scale, bias,
                                         in_a ? in_a + (size_t)n0 * Cin : nullptr, in_b ? in_b + (size_t)n0 * Cin : nullptr,
                                         gn_part ? gn_part + (size_t)n0 * slots * Cout * 2 : nullptr, n, H, W, Cin, Cout, flags,
                                         in_relu, layout, stream);
        if (rc != CPR_OK) return rc;
    }
    return CPR_OK;
}
