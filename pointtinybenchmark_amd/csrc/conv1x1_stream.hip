// Streaming 1x1 convolution for the HBM-bound layers of the backbone (fp32, v_mfma_f32_32x32x2_f32).
//
// The bottleneck's 1x1 convs with 64 / 128 input channels (T/mmdet/models/backbones/resnet.py:262-302: conv3 + bn3 + shortcut
// add + ReLU at 160x160 / 80x80) move 14-30 flop per byte: the roof is the HBM stream (a torch elementwise kernel with the
// same read : write mix reaches 6.0 TB/s on this part, tools/diag/hbm_stream_mix.py), and the tiled kernel of conv_mfma.hip
// reaches 4.0-4.2 TB/s on them -- its workgroups load, multiply and store one after the other, each phase behind the
// latency of the previous one, and every register-returning load costs the issuing wave ~150 cycles of MFMA cadence
// (tools/diag/mfma_shadow).  Here ONE persistent workgroup per CU keeps its weight panel in LDS for its whole life and
// streams pixel tiles through it:
//   LDS (128 KB): weights [BN couts][K] once | two pixel tiles [BM][K], both filled by LDS-DMA (buffer_load ... lds: no
//   staging registers, no ds_write, nothing in front of the MFMAs); rows are K floats, the 16-byte unit u of row r sits at
//   u ^ (r & 15) -- applied on the SOURCE side of the DMA, whose LDS image is lane-linear -- so the 16 lanes of a
//   ds_read_b128 service group hit 16 different slots.
//   tile t: request tile t+1 | request the residual rows of tile t (they fly during the MFMAs; by LDS-DMA too where the
//           tile's residual fits the 32 KB of LDS left, i.e. K = 128 and K = 256, else 16 dword loads per block) | MFMAs over the whole K |
//           wait + ONE barrier | affine + residual + ReLU, stores.  In flight per CU during the MFMAs: the next pixel tile,
//           this tile's residual and the previous tile's stores (~290 KB).
// K = Cin in {64, 128}: BM x BN = 128 x 256 / 64 x 128 (weight panel and each pixel tile are 64 KB / 32 KB for both); K = 256: a
// second kernel below streams the tile in four k chunks.
// The accumulation order per output element is the tiled kernel's (8-wide k groups, lanes < 32 take k0..k0+3, lanes >= 32
// k0+4..k0+7, four MFMAs per group) and so is the epilogue arithmetic: results are BIT-IDENTICAL to conv_mfma_kernel, which
// keeps serving the shapes this kernel does not take (tests/test_gpu_kernels.py compares the two).
#include "common.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct StreamParams {
    const float* in;        // [M][K]
    const float* wgt;       // [Cout][K]
    float* out;             // [M][Cout]
    const float* scale;     // [Cout] or null
    const float* bias;      // [Cout] or null
    const float* residual;  // [M][Cout] or null
    int M, Cout, relu, res_mask, tilesM, tilesN;
};

template <int K>
__global__ __launch_bounds__(512, 1) void conv1x1_stream_kernel(StreamParams p) {
    constexpr int BM = 8192 / K, BN = 16384 / K;   // 128 x 256 (K = 64), 64 x 128 (K = 128)
    constexpr int UPR = K / 4;                     // 16-byte units per row
    constexpr int WM = BM / 2, WN = BN / 4;        // 8 waves = 2 (pixels) x 4 (couts)
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int KK = K / 8;                      // 8-wide k groups
    constexpr bool RLDS = BM * BN * 4 <= 32768;    // the tile's residual rows fit the 32 KB of LDS left: they come by LDS-DMA too
    __shared__ __attribute__((aligned(16))) float smem[(BN + 2 * BM) * K + (RLDS ? BM * BN : 0)];
    float* Bs = smem;
    float* As = smem + BN * K;
    float* Rs = smem + (BN + 2 * BM) * K;          // RLDS: [BM pixels][BN couts], the tile's rows of the residual map as they lie in HBM

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, half = lane >> 5;

    // Tile schedule: the NT workgroups that read one pixel tile (one per cout panel) sit on ONE XCD (block b runs on XCD b % 8)
    // and walk the pixel tiles in step, so a tile crosses HBM once and is served to the other panels from that XCD's L2.
    const int NT = p.tilesN, G = gridDim.x;        // host: G % (8 NT) == 0
    const int b = blockIdx.x;
    const int tn = (b >> 3) % NT;
    const int s0 = (b & 7) + 8 * ((b >> 3) / NT), sstep = G / NT;
    const int n0 = tn * BN;

    // ---- LDS-DMA: piece q of a [rows][K] block = 512 units of 16 bytes, unit P = 512 q + tid -> row P / UPR, slot P % UPR;
    // the slot holds the row's unit (slot ^ (row & 15)).  512 / UPR rows per piece is a multiple of 16, so the lane's source
    // offset inside a piece is the same for every piece: one VGPR, the piece and the tile go into the scalar offset.
    const int prow = tid / UPR, pslot = tid % UPR;
    const int voff = (prow * K + ((pslot ^ (prow & 15)) << 2)) * 4;
    const size_t in_addr = (size_t)p.in, w_addr = (size_t)p.wgt;
    const i32x4 rs_in = {(int)(unsigned)in_addr, (int)(unsigned)(in_addr >> 32) & 0xffff, (int)((size_t)p.M * K * 4), 0x00020000};
    const i32x4 rs_w = {(int)(unsigned)w_addr, (int)(unsigned)(w_addr >> 32) & 0xffff, (int)((size_t)p.Cout * K * 4), 0x00020000};
    const int lds_b = (int)(unsigned)(size_t)Bs, lds_a = (int)(unsigned)(size_t)As;
    auto dma = [&](const i32x4& rs, int lds_base, int row0, int q) {   // piece q of the block that starts at row row0
        const int m0v = __builtin_amdgcn_readfirstlane(lds_base + q * 8192 + wave * 1024);
        const int soff = __builtin_amdgcn_readfirstlane(row0 * K * 4 + q * 8192);
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                     :: "s"(m0v), "v"(voff), "s"(rs), "s"(soff) : "memory");
    };
    auto dma_tile = [&](int mt, int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dma(rs_in, lds_a + buf * (BM * K * 4), mt * BM, q);
    };
    // RLDS: piece q of the residual tile = 512 units of 16 bytes; a row of the tile is BN / 4 units, unit P = 512 q + tid -> row
    // P / (BN / 4), unit P % (BN / 4); rows are Cout floats apart in HBM
    const size_t r_addr = (size_t)(p.residual ? p.residual : p.out);
    const i32x4 rs_r = {(int)(unsigned)r_addr, (int)(unsigned)(r_addr >> 32) & 0xffff, (int)((size_t)p.M * p.Cout * 4), 0x00020000};
    constexpr int RUPR = BN / 4, RPIECES = RLDS ? BM * BN * 4 / 8192 : 0;
    const int rvoff = ((tid / RUPR) * p.Cout + (tid % RUPR) * 4) * 4;
    const int lds_r = (int)(unsigned)(size_t)Rs;
    auto dma_res = [&](int mt) {
#pragma unroll
        for (int q = 0; q < RPIECES; ++q) {
            const int m0v = __builtin_amdgcn_readfirstlane(lds_r + q * 8192 + wave * 1024);
            const int soff = __builtin_amdgcn_readfirstlane(((mt * BM + q * (512 / RUPR)) * p.Cout + n0) * 4);
            asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                         :: "s"(m0v), "v"(rvoff), "s"(rs_r), "s"(soff) : "memory");
        }
    };

    // ---- fragments: A[m = l31][k], B[cout = l31][k]; lane reads the 4 floats k0 + 4 half .. +3 of an 8-wide group
    const int hx = half ^ (l31 & 15);              // unit (2 kk + half) of row r (r & 15 == l31 & 15) sits in slot (2 kk) ^ hx
    const float* a_row = As + (wm * WM + l31) * K;
    const float* b_row = Bs + (wn * WN + l31) * K;
    auto frags = [&](int buf, int kk, f32x4 (&fa)[MI], f32x4 (&fb)[NI]) {
        const int slot4 = ((2 * kk) ^ hx) << 2;
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a_row + buf * (BM * K) + i * 32 * K + slot4);
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b_row + j * 32 * K + slot4);
    };

    // ---- epilogue constants (the cout panel never changes)
    const unsigned row_bytes = (unsigned)p.Cout * 4u;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)p.M * p.Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.residual ? p.residual : p.out), 0, (int)((size_t)p.M * p.Cout * 4), 0x00020000);
    float sc[NI], bi[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int c = n0 + wn * WN + j * 32 + l31;
        sc[j] = p.scale ? p.scale[c] : 1.f;
        bi[j] = p.bias ? p.bias[c] : 0.f;
    }

    // ---- prologue: the weight panel and the first pixel tile
    if (s0 < p.tilesM) {
#pragma unroll
        for (int q = 0; q < 8; ++q) dma(rs_w, lds_b, n0, q);
        dma_tile(s0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int buf = 0;
    for (int mt = s0; mt < p.tilesM; mt += sstep, buf ^= 1) {
        const int m0 = mt * BM;
        if (mt + sstep < p.tilesM) dma_tile(mt + sstep, buf ^ 1);   // every wave is past its reads of that buffer (barrier below)
        // residual rows of this tile: requested now, used after the MFMAs.  D layout of a 32 x 32 block: column = l31 (cout),
        // row = (r & 3) + 8 (r >> 2) + 4 half (pixel)
        float res[MI][NI][16];
        unsigned off0[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int c = n0 + wn * WN + j * 32 + l31;
                const int rbase = m0 + wm * WM + i * 32 + 4 * half;
                off0[i][j] = (unsigned)(rbase * p.Cout + c) * 4u;
                if (RLDS) {
                } else if (p.residual) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        res[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rs_res, (int)off0[i][j], (int)((unsigned)((r & 3) + 8 * (r >> 2)) * row_bytes), 0));   // row in the scalar offset
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) res[i][j][r] = 0.f;
                }
            }
        if (RLDS && p.residual) dma_res(mt);     // (the residual buffer is free: barrier behind the previous tile's reads of it)
        __builtin_amdgcn_sched_barrier(0);

        f32x16 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        f32x4 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
        frags(buf, 0, fa0, fb0);
#pragma unroll
        for (int kk = 0; kk < KK; kk += 2) {
            frags(buf, kk + 1, fa1, fb1);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[i][t], fb0[j][t], acc[i][j], 0, 0, 0);
            if (kk + 2 < KK) frags(buf, kk + 2, fa0, fb0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[i][t], fb1[j][t], acc[i][j], 0, 0, 0);
        }
        // the next tile and this tile's residual have landed (requests complete in order; the previous tile's stores were
        // issued a whole tile ago); after the barrier every wave is past its fragment reads of this buffer
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (RLDS) {      // residual values of this lane's outputs from LDS; a second barrier frees the buffer for the next tile's request
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        res[i][j][r] = p.residual ? Rs[(wm * WM + i * 32 + 4 * half + (r & 3) + 8 * (r >> 2)) * BN + wn * WN + j * 32 + l31]
                                                  : 0.f;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }

#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] * sc[j] + bi[j];
                    if (p.res_mask) v = res[i][j][r] > 0.f ? v : 0.f;
                    else v += res[i][j][r];
                    if (p.relu) v = fmaxf(v, 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, (int)off0[i][j],
                                                          (int)((unsigned)((r & 3) + 8 * (r >> 2)) * row_bytes), 0);
                }
    }
}

// K = 256 (conv1 of the 256-channel stages, conv3 of layer3: 160x160 256->64 / ->128, 40x40 256->1024 + shortcut): the weight panel
// is [64 couts][256] (64 KB, stored as four 64-wide k chunks), a pixel tile [128][256] streams through as four chunks of [128][64]
// (32 KB each, double-buffered) -- one barrier per chunk, 32 MFMAs per wave between barriers, 8 waves = 4 (pixels) x 2 (couts), one
// 32 x 32 block each.  The chunk stream runs across tiles (the first chunk of tile t+1 is requested behind the last MFMAs of tile t);
// the residual rows of a tile are requested behind its first chunk.  Same accumulation order as the tiled kernel (k ascending).
__global__ __launch_bounds__(512, 1) void conv1x1_stream_k256_kernel(StreamParams p) {
    constexpr int K = 256, BM = 128, BN = 64, KC = 64, NCH = K / KC;
    __shared__ __attribute__((aligned(16))) float smem[BN * K + 2 * BM * KC + BM * BN];
    float* Bs = smem;                 // [chunk][64][64]
    float* As = smem + BN * K;        // [2][128][64]
    float* Rs = As + 2 * BM * KC;     // [128 pixels][64 couts]: the tile's rows of the residual map, by LDS-DMA like the operands
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int l31 = lane & 31, half = lane >> 5;
    const int NT = p.tilesN, G = gridDim.x;
    const int b = blockIdx.x;
    const int tn = (b >> 3) % NT;
    const int s0 = (b & 7) + 8 * ((b >> 3) / NT), sstep = G / NT;
    const int n0 = tn * BN;

    // LDS-DMA: a [rows][64] chunk block = 16 units of 16 bytes per row; piece q = units 512 q + tid -> row 32 q + tid / 16, slot
    // tid % 16 holding the row's unit slot ^ (row & 15); global rows are K floats apart, the chunk is a 256-byte column offset
    const int prow = tid >> 4, pslot = tid & 15;
    const int voff = prow * (K * 4) + ((pslot ^ (prow & 15)) << 4);
    const size_t in_addr = (size_t)p.in, w_addr = (size_t)p.wgt;
    const i32x4 rs_in = {(int)(unsigned)in_addr, (int)(unsigned)(in_addr >> 32) & 0xffff, (int)((size_t)p.M * K * 4), 0x00020000};
    const i32x4 rs_w = {(int)(unsigned)w_addr, (int)(unsigned)(w_addr >> 32) & 0xffff, (int)((size_t)p.Cout * K * 4), 0x00020000};
    const int lds_b = (int)(unsigned)(size_t)Bs, lds_a = (int)(unsigned)(size_t)As;
    auto dma = [&](const i32x4& rs, int lds_dst, int row0, int kc, int q) {
        const int m0v = __builtin_amdgcn_readfirstlane(lds_dst + q * 8192 + wave * 1024);
        const int soff = __builtin_amdgcn_readfirstlane((row0 + 32 * q) * (K * 4) + kc * (KC * 4));
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                     :: "s"(m0v), "v"(voff), "s"(rs), "s"(soff) : "memory");
    };
    auto dma_chunk = [&](int mt, int kc, int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dma(rs_in, lds_a + buf * (BM * KC * 4), mt * BM, kc, q);
    };
    // residual tile: 16 units of 16 bytes per row (64 couts), piece q = rows 32 q + tid / 16; rows are Cout floats apart in HBM
    const size_t r_addr = (size_t)(p.residual ? p.residual : p.out);
    const i32x4 rs_r = {(int)(unsigned)r_addr, (int)(unsigned)(r_addr >> 32) & 0xffff, (int)((size_t)p.M * p.Cout * 4), 0x00020000};
    const int rvoff = ((tid >> 4) * p.Cout + (tid & 15) * 4) * 4;
    const int lds_r = (int)(unsigned)(size_t)Rs;
    auto dma_res = [&](int mt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m0v = __builtin_amdgcn_readfirstlane(lds_r + q * 8192 + wave * 1024);
            const int soff = __builtin_amdgcn_readfirstlane(((mt * BM + 32 * q) * p.Cout + n0) * 4);
            asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                         :: "s"(m0v), "v"(rvoff), "s"(rs_r), "s"(soff) : "memory");
        }
    };
    const int hx = half ^ (l31 & 15);
    const float* a_row = As + (wm * 32 + l31) * KC;
    const float* b_row = Bs + (wn * 32 + l31) * KC;

    const unsigned row_bytes = (unsigned)p.Cout * 4u;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)p.M * p.Cout * 4), 0x00020000);
    const int cch = n0 + wn * 32 + l31;
    const float sc = p.scale ? p.scale[cch] : 1.f, bi = p.bias ? p.bias[cch] : 0.f;

    if (s0 < p.tilesM) {
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc)
#pragma unroll
            for (int q = 0; q < 2; ++q) dma(rs_w, lds_b + kc * (BN * KC * 4), n0, kc, q);
        dma_chunk(s0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int buf = 0;
    for (int mt = s0; mt < p.tilesM; mt += sstep) {
        const int m0 = mt * BM;
        const unsigned off0 = (unsigned)((m0 + wm * 32 + 4 * half) * p.Cout + cch) * 4u;
        float res[16];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc) {
            // the next chunk of the stream -> the other buffer (every wave is past its reads of it: barrier at the end of the last chunk)
            if (kc + 1 < NCH) dma_chunk(mt, kc + 1, buf ^ 1);
            else if (mt + sstep < p.tilesM) dma_chunk(mt + sstep, 0, buf ^ 1);
            if (kc == 0 && p.residual) dma_res(mt);      // four more requests, younger than the chunk's: they may stay in flight below
            f32x4 fa0, fb0, fa1, fb1;
            const float* ap = a_row + buf * (BM * KC);
            const float* bp = b_row + kc * (BN * KC);
            fa0 = *reinterpret_cast<const f32x4*>(ap + ((0 ^ hx) << 2));
            fb0 = *reinterpret_cast<const f32x4*>(bp + ((0 ^ hx) << 2));
#pragma unroll
            for (int kk = 0; kk < 8; kk += 2) {
                fa1 = *reinterpret_cast<const f32x4*>(ap + (((2 * (kk + 1)) ^ hx) << 2));
                fb1 = *reinterpret_cast<const f32x4*>(bp + (((2 * (kk + 1)) ^ hx) << 2));
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[t], fb0[t], acc, 0, 0, 0);
                if (kk + 2 < 8) {
                    fa0 = *reinterpret_cast<const f32x4*>(ap + (((2 * (kk + 2)) ^ hx) << 2));
                    fb0 = *reinterpret_cast<const f32x4*>(bp + (((2 * (kk + 2)) ^ hx) << 2));
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[t], fb1[t], acc, 0, 0, 0);
            }
            // the next chunk has landed (behind the first chunk of a tile its 4 residual requests are younger and may stay in flight)
            if (kc == 0 && p.residual) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            buf ^= 1;
        }
        // residual values of this lane's outputs (landed: the waits of chunks 1..3 drained the queue); the barrier frees the buffer
        // for the next tile's request
#pragma unroll
        for (int r = 0; r < 16; ++r)
            res[r] = p.residual ? Rs[(wm * 32 + 4 * half + (r & 3) + 8 * (r >> 2)) * BN + wn * 32 + l31] : 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r] * sc + bi;
            if (p.res_mask) v = res[r] > 0.f ? v : 0.f;
            else v += res[r];
            if (p.relu) v = fmaxf(v, 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, (int)off0,
                                                  (int)((unsigned)((r & 3) + 8 * (r >> 2)) * row_bytes), 0);
        }
    }
}

// Takes the launch when it is one of the streamed shapes; CPR_ERR_UNSUPPORTED = not ours (the caller runs the tiled kernel).
int conv1x1_stream_launch(const float* in, const float* wgt, float* out, const float* scale, const float* bias,
                          const float* residual, long long M, int Cin, int Cout, int relu, int res_mask, int min_tiles,
                          hipStream_t stream) {
    if (!(Cin == 64 || Cin == 128 || Cin == 256)) return CPR_ERR_UNSUPPORTED;
    const int bm = Cin == 256 ? 128 : 8192 / Cin, bn = Cin == 256 ? 64 : 16384 / Cin;
    if (M <= 0 || M % bm != 0 || Cout % bn != 0) return CPR_ERR_UNSUPPORTED;
    if (M * Cin * 4 >= (1ll << 31) || M * Cout * 4 >= (1ll << 31)) return CPR_ERR_UNSUPPORTED;
    StreamParams p;
    p.in = in; p.wgt = wgt; p.out = out; p.scale = scale; p.bias = bias; p.residual = residual;
    p.M = (int)M; p.Cout = Cout; p.relu = relu; p.res_mask = res_mask;
    p.tilesM = (int)(M / bm); p.tilesN = Cout / bn;
    if (p.tilesN > 32 || 32 % p.tilesN != 0) return CPR_ERR_UNSUPPORTED;        // panels of a pixel tile share an XCD: 8 NT | grid
    if ((long long)p.tilesM * p.tilesN < min_tiles) return CPR_ERR_UNSUPPORTED;  // too few tiles to fill 256 persistent workgroups
    if (res_mask && !residual) return CPR_ERR_ARG;
    int grid = 256;                                                              // one workgroup per CU, a multiple of 8 NT
    const long long need = ((long long)p.tilesM + 7) / 8 * 8 * p.tilesN;
    if (need < grid) grid = (int)need;
    if (Cin == 64) hipLaunchKernelGGL(conv1x1_stream_kernel<64>, dim3(grid), dim3(512), 0, stream, p);
    else if (Cin == 128) hipLaunchKernelGGL(conv1x1_stream_kernel<128>, dim3(grid), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL(conv1x1_stream_k256_kernel, dim3(grid), dim3(512), 0, stream, p);
    CPR_LAUNCH_STATUS();
}

// Explicit entry (tests, tools): the streamed kernel on any shape it can take, whatever the tile count.
extern "C" int cpr_conv1x1_stream_fwd(const float* in, const float* wgt, float* out, const float* scale, const float* bias,
                                      const float* residual, long long M, int Cin, int Cout, int flags, hipStream_t stream) {
    CPR_CHECK_ARG(in && wgt && out && M > 0 && Cin > 0 && Cout > 0);
    CPR_CHECK_ARG((flags & ~(CPR_CONV_RELU | CPR_CONV_RES_MASK)) == 0);
    return conv1x1_stream_launch(in, wgt, out, scale, bias, residual, M, Cin, Cout, flags & CPR_CONV_RELU,
                                 (flags & CPR_CONV_RES_MASK) ? 1 : 0, 1, stream);
}
