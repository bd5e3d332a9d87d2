// Data side feeding the path: the per-image tail of the mmdet train/test pipeline on the device.
//   RandomFlip(horizontal) -> Normalize(mean, std, to_rgb) -> Pad(size_divisor) -> DefaultFormatBundle
//   (T/mmdet/datasets/pipelines/transforms.py:431-480 (flip), :560-600 (Normalize -> mmcv.imnormalize),
//    :603-680 (Pad, pad_val 0 AFTER normalisation), formating.py:180-214)
// fused into one pass from the decoded uint8 HWC image straight to the stem's input layout (N, Hp, Wp, 4) fp32 (4th channel
// zero), so the float NCHW image of the reference is never materialised.  One thread per output pixel: 3 B in, 16 B out.
// Arithmetic as mmcv.imnormalize_: float32(img); optional BGR->RGB; (x - mean_f32) * stdinv_f32, stdinv = 1/float64(std)
// rounded to fp32 (OpenCV arithmetic on a CV_32F array converts the scalar to float); no FMA contraction.
#include "common.h"

__global__ void preprocess_u8_kernel(const unsigned char* __restrict__ img, const int* __restrict__ flip,
                                     float m0, float m1, float m2, float s0, float s1, float s2, int to_rgb,
                                     float* __restrict__ out, int N, int H, int W, int Hp, int Wp) {
    const long long total = (long long)N * Hp * Wp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wp);
        const long long r = i / Wp;
        const int y = (int)(r % Hp);
        const int n = (int)(r / Hp);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y < H && x < W) {
            const int sx = (flip && flip[n]) ? W - 1 - x : x;
            const unsigned char* p = img + (((size_t)n * H + y) * W + sx) * 3;
            const float c0 = (float)p[0], c1 = (float)p[1], c2 = (float)p[2];
            const float r0 = to_rgb ? c2 : c0, r2 = to_rgb ? c0 : c2;
            v[0] = __fmul_rn(__fsub_rn(r0, m0), s0);
            v[1] = __fmul_rn(__fsub_rn(c1, m1), s1);
            v[2] = __fmul_rn(__fsub_rn(r2, m2), s2);
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

extern "C" int cpr_preprocess_u8(const unsigned char* img, const int* flip, const float* mean3, const float* stdinv3,
                                 int to_rgb, float* out, int N, int H, int W, int Hp, int Wp, hipStream_t stream) {
    // img (N,H,W,3) uint8 on the device; mean3 / stdinv3 are HOST pointers (three floats each); flip (N) int32 or NULL
    CPR_CHECK_ARG(img && mean3 && stdinv3 && out && N > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W);
    const long long total = (long long)N * Hp * Wp;
    const int grid = (int)(cdivll(total, 256) < 65536 ? cdivll(total, 256) : 65536);
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid), dim3(256), 0, stream, img, flip, mean3[0], mean3[1], mean3[2],
                       stdinv3[0], stdinv3[1], stdinv3[2], to_rgb, out, N, H, W, Hp, Wp);
    CPR_LAUNCH_STATUS();
}

// Box side of Resize (scale 1) -> RandomFlip: Resize._resize_bboxes clips every bbox field to the image
// (bbox_clip_border=True: x to [0, W], y to [0, H]; transforms.py:241-249) BEFORE RandomFlip.bbox_flip mirrors it
// (transforms.py:397-415).  boxes (n,4) xyxy of image `img_of[i]`; hw (N,2) int32 = img_shape[:2]; flipped when flip[img].
__global__ void clip_flip_boxes_kernel(float* __restrict__ boxes, const int* __restrict__ img_of,
                                       const int* __restrict__ flip, const int* __restrict__ hw, int n, int clip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int im = img_of[i];
    const float h = (float)hw[im * 2], w = (float)hw[im * 2 + 1];
    float x1 = boxes[i * 4], y1 = boxes[i * 4 + 1], x2 = boxes[i * 4 + 2], y2 = boxes[i * 4 + 3];
    if (clip) {
        x1 = fminf(fmaxf(x1, 0.f), w); x2 = fminf(fmaxf(x2, 0.f), w);
        y1 = fminf(fmaxf(y1, 0.f), h); y2 = fminf(fmaxf(y2, 0.f), h);
    }
    if (flip[im]) {
        const float t = x1;
        x1 = __fsub_rn(w, x2);
        x2 = __fsub_rn(w, t);
    }
    boxes[i * 4] = x1; boxes[i * 4 + 1] = y1; boxes[i * 4 + 2] = x2; boxes[i * 4 + 3] = y2;
}
extern "C" int cpr_clip_flip_boxes(float* boxes, const int* img_of, const int* flip, const int* img_hw, int n, int clip,
                                   hipStream_t stream) {
    CPR_CHECK_ARG(n >= 0);
    if (n == 0) return CPR_OK;
    CPR_CHECK_ARG(boxes && img_of && flip && img_hw);
    hipLaunchKernelGGL(clip_flip_boxes_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, boxes, img_of, flip, img_hw, n, clip);
    CPR_LAUNCH_STATUS();
}
