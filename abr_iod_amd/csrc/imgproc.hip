// Pixel kernels of the ABR data path: everything the reference does per sample on the HOST with Pillow and
// numpy -- resize of the rehearsal box crops and of the final image, mixup blend, mosaic paste, flip + ToTensor + BGR255 +
// mean/std normalisation, zero-padded batching -- on uint8 HWC images resident in HBM.
//
// Reference: maskrcnn_benchmark/data/datasets/voc_abr.py:512-816, data/transforms/transforms.py:64-165,
// structures/image_list.py:57-70.  Parity target is BIT-EXACT:
//   * resize = Pillow's 8-bit antialiased separable resampler (src/libImaging/Resample.c): the host precomputes the double
//     precision filter weights exactly as Pillow does and hands them over as 22-bit fixed point; the kernels do the integer
//     convolution, horizontal pass first, uint8 intermediate, clip8 = clamp((acc + 2^21) >> 22).
//   * blend = numpy's float64 expression assigned into a uint8 array (truncation) -- done in f64 here as well.
//   * normalise = the fp32 chain u8 -> /255 -> *255 -> -mean -> /std with IEEE division, contraction off.
// All of it is HBM-bound byte work (a 500x375 image is 0.56 MB); one thread per output pixel, 3 channels per thread.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// dst[y, xx, c] = clip8(2^21 + sum_t src[y, x0+t, c] * k[xx, t])
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, int H, int W, uint8_t* __restrict__ dst, int OW,
                                                         const int32_t* __restrict__ bounds, const int32_t* __restrict__ coeffs, int ksize) {
    const int64_t total = (int64_t)H * OW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / OW), xx = (int)(i % OW);
        const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
        const int32_t* k = coeffs + (size_t)xx * ksize;
        const uint8_t* s = src + ((size_t)y * W + x0) * 3;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int t = 0; t < n; t++) {
            const int kv = k[t];
            a0 += s[3 * t] * kv; a1 += s[3 * t + 1] * kv; a2 += s[3 * t + 2] * kv;
        }
        uint8_t* d = dst + i * 3;
        d[0] = clip8(a0); d[1] = clip8(a1); d[2] = clip8(a2);
    }
}

// dst[yy, x, c] = clip8(2^21 + sum_t src[y0+t, x, c] * k[yy, t])
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* __restrict__ src, int H, int W, uint8_t* __restrict__ dst, int OH,
                                                         const int32_t* __restrict__ bounds, const int32_t* __restrict__ coeffs, int ksize) {
    const int64_t total = (int64_t)OH * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int yy = (int)(i / W), x = (int)(i % W);
        const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
        const int32_t* k = coeffs + (size_t)yy * ksize;
        const uint8_t* s = src + ((size_t)y0 * W + x) * 3;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int t = 0; t < n; t++) {
            const int kv = k[t];
            const uint8_t* p = s + (size_t)t * W * 3;
            a0 += p[0] * kv; a1 += p[1] * kv; a2 += p[2] * kv;
        }
        uint8_t* d = dst + i * 3;
        d[0] = clip8(a0); d[1] = clip8(a1); d[2] = clip8(a2);
    }
}

// img[y0+j, x0+i] = (uint8)(lam * img[...] + (1 - lam) * crop[off_y+j, off_x+i])   -- voc_abr.py:664-683, float64 like numpy
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void blend_paste_kernel(uint8_t* __restrict__ img, int W, const uint8_t* __restrict__ crop, int CW,
                                                          int x0, int y0, int rw, int rh, int off_x, int off_y, double lam) {
    const int total = rw * rh;
    const double one_minus = 1.0 - lam;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i / rw, ii = i % rw;
        uint8_t* p = img + ((size_t)(y0 + j) * W + x0 + ii) * 3;
        const uint8_t* c = crop + ((size_t)(off_y + j) * CW + off_x + ii) * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const double a = lam * (double)p[ch];
            const double b = one_minus * (double)c[ch];
            p[ch] = (uint8_t)(a + b);
        }
    }
}

// dst[dy+j, dx+i] = src[sy+j, sx+i]  (mosaic tiles, voc_abr.py:765)
__global__ __launch_bounds__(256) void copy_rect_kernel(uint8_t* __restrict__ dst, int DW, const uint8_t* __restrict__ src, int SW, int dx,
                                                        int dy, int sx, int sy, int rw, int rh) {
    const int total = rw * rh;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i / rw, ii = i % rw;
        uint8_t* p = dst + ((size_t)(dy + j) * DW + dx + ii) * 3;
        const uint8_t* c = src + ((size_t)(sy + j) * SW + sx + ii) * 3;
        p[0] = c[0]; p[1] = c[1]; p[2] = c[2];
    }
}

__global__ __launch_bounds__(256) void fill_kernel(uint8_t* __restrict__ dst, int64_t n, uint8_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = v;
}

// out[c, y, x] (one [3,HP,WP] slot of the batch, zero outside [h,w]) = ((src[y, flip ? w-1-x : x, perm(c)] / 255) * 255 - mean[c]) / std[c]
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void normalize_kernel(const uint8_t* __restrict__ src, int h, int w, int flip, int to_bgr255, float m0,
                                                        float m1, float m2, float s0, float s1, float s2, float* __restrict__ out, int HP,
                                                        int WP) {
    const int64_t plane = (int64_t)HP * WP;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / WP), x = (int)(i % WP);
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (y < h && x < w) {
            const uint8_t* p = src + ((size_t)y * w + (flip ? w - 1 - x : x)) * 3;
            float r = __fdiv_rn((float)p[0], 255.f), g = __fdiv_rn((float)p[1], 255.f), b = __fdiv_rn((float)p[2], 255.f);
            float c0 = r, c1 = g, c2 = b;
            if (to_bgr255) { c0 = b * 255.f; c1 = g * 255.f; c2 = r * 255.f; }
            v0 = __fdiv_rn(c0 - m0, s0); v1 = __fdiv_rn(c1 - m1, s1); v2 = __fdiv_rn(c2 - m2, s2);
        }
        out[i] = v0; out[plane + i] = v1; out[2 * plane + i] = v2;
    }
}

inline unsigned grid_for(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, 16384); }


// ---- ColorJitter's pixel ops (transforms.py:132-150 -> torchvision -> Pillow), in place on a uint8 HWC RGB image.  Bit-exact restatements of
// Pillow's C arithmetic (Blend.c; Convert.c rgb2l / rgb2hsv_row / hsv2rgb_row), pinned against Pillow by oracle/abr_data_ref.py's tests:
//   blend(degenerate, image, f): float32, truncation; for f outside [0, 1] clipped to [0, 255] first
//   brightness: degenerate = 0;  contrast: degenerate = int(mean(L) + 0.5) (one scalar per image, summed exactly in 64-bit integers by
//   gray_sum_kernel and read from device memory: no host round trip);  saturation: degenerate = L of the pixel;  L = (19595 R + 38470 G + 7471 B + 2^15) >> 16
//   hue: RGB -> HSV (float quotients; the sector offset and the fmod in double), H += shift (mod 256), HSV -> RGB (double products, C round())
#pragma clang fp contract(off)
__device__ __forceinline__ uint8_t pil_blend(const int in1, const int in2, const float alpha, const bool inside) {
    const float t = (float)in1 + alpha * ((float)in2 - (float)in1);
    if (inside) return (uint8_t)(int)t;
    return t <= 0.f ? (uint8_t)0 : (t >= 255.f ? (uint8_t)255 : (uint8_t)(int)t);
}
__device__ __forceinline__ int pil_l(const int r, const int g, const int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

__global__ __launch_bounds__(256) void gray_sum_kernel(const uint8_t* __restrict__ img, int64_t npix, unsigned long long* __restrict__ sum) {
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x)
        acc += (unsigned)pil_l(img[3 * i], img[3 * i + 1], img[3 * i + 2]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(sum, acc);   // (integer sum: exact, order-free)
}

// op: 0 brightness, 1 contrast (gray_sum holds the image's L sum), 2 saturation
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void color_blend_kernel(uint8_t* __restrict__ img, int64_t npix, int op, float factor,
                                                          const unsigned long long* __restrict__ gray_sum) {
    const bool inside = factor >= 0.f && factor <= 1.f;
    int mean = 0;
    if (op == 1) mean = (int)((double)*gray_sum / (double)npix + 0.5);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        uint8_t* p = img + 3 * i;
        const int r = p[0], g = p[1], b = p[2];
        const int d = op == 0 ? 0 : (op == 1 ? mean : pil_l(r, g, b));
        p[0] = pil_blend(d, r, factor, inside); p[1] = pil_blend(d, g, factor, inside); p[2] = pil_blend(d, b, factor, inside);
    }
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void hue_shift_kernel(uint8_t* __restrict__ img, int64_t npix, int shift) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        uint8_t* px = img + 3 * i;
        const int r = px[0], g = px[1], b = px[2];
        const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
        int uh = 0, us = 0;
        const int uv = maxc;
        if (minc != maxc) {   // rgb2hsv_row
            const float cr = (float)(maxc - minc);
            const float s = cr / (float)maxc;
            const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
            float h;
            if (r == maxc) h = bc - gc;
            else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
            else h = (float)(4.0 + (double)gc - (double)rc);
            h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
            uh = (int)((double)h * 255.0); uh = uh < 0 ? 0 : (uh > 255 ? 255 : uh);
            us = (int)((double)s * 255.0); us = us < 0 ? 0 : (us > 255 ? 255 : us);
        }
        uh = (uh + shift) & 255;
        if (us == 0) { px[0] = px[1] = px[2] = (uint8_t)uv; continue; }   // hsv2rgb_row
        const double hh = (double)(float)uh * 6.0 / 255.0;
        const int sec = (int)floor(hh);
        const double f = (double)(float)(hh - (double)(float)sec);
        const double fs = (double)(float)((double)(float)us / 255.0);
        const double vf = (double)(float)uv;
        int pp = (int)round(vf * (1.0 - fs)), qq = (int)round(vf * (1.0 - fs * f)), tt = (int)round(vf * (1.0 - fs * (1.0 - f)));
        pp = pp < 0 ? 0 : (pp > 255 ? 255 : pp); qq = qq < 0 ? 0 : (qq > 255 ? 255 : qq); tt = tt < 0 ? 0 : (tt > 255 ? 255 : tt);
        int R, G, B;
        switch (sec % 6) {
            case 0: R = uv; G = tt; B = pp; break;
            case 1: R = qq; G = uv; B = pp; break;
            case 2: R = pp; G = uv; B = tt; break;
            case 3: R = pp; G = qq; B = uv; break;
            case 4: R = tt; G = pp; B = uv; break;
            default: R = uv; G = pp; B = qq; break;
        }
        px[0] = (uint8_t)R; px[1] = (uint8_t)G; px[2] = (uint8_t)B;
    }
}

}  // namespace

extern "C" int abr_img_resample_u8(const uint8_t* src, int H, int W, uint8_t* dst, int OH, int OW, const int32_t* bounds_h,
                                   const int32_t* coeffs_h, int ksize_h, const int32_t* bounds_v, const int32_t* coeffs_v, int ksize_v,
                                   uint8_t* tmp, void* stream) {
    ABR_REQUIRE(H > 0 && W > 0 && OH > 0 && OW > 0, "img_resample: bad shape");
    ABR_REQUIRE(src && dst, "img_resample: null pointer");
    hipStream_t st = abr::as_stream(stream);
    const bool need_h = OW != W, need_v = OH != H;
    if (!need_h && !need_v) {  // Image.resize returns a copy when nothing changes
        if (hipMemcpyAsync(dst, src, (size_t)H * W * 3, hipMemcpyDeviceToDevice, st) != hipSuccess) return ABR_E_LAUNCH;
        return ABR_OK;
    }
    ABR_REQUIRE(!need_h || (bounds_h && coeffs_h && ksize_h > 0), "img_resample: horizontal coefficients missing");
    ABR_REQUIRE(!need_v || (bounds_v && coeffs_v && ksize_v > 0), "img_resample: vertical coefficients missing");
    ABR_REQUIRE(!(need_h && need_v) || tmp, "img_resample: two passes need the [H,OW,3] intermediate");
    const uint8_t* cur = src;
    if (need_h) {
        uint8_t* o = need_v ? tmp : dst;
        resample_h_kernel<<<grid_for((int64_t)H * OW), 256, 0, st>>>(cur, H, W, o, OW, bounds_h, coeffs_h, ksize_h);
        cur = o;
    }
    if (need_v) resample_v_kernel<<<grid_for((int64_t)OH * OW), 256, 0, st>>>(cur, H, OW, dst, OH, bounds_v, coeffs_v, ksize_v);
    ABR_CHECK_LAUNCH("img_resample");
    return ABR_OK;
}

extern "C" int abr_img_blend_paste_u8(uint8_t* img, int H, int W, const uint8_t* crop, int CH, int CW, int x0, int y0, int rw, int rh,
                                      int off_x, int off_y, double lam, void* stream) {
    ABR_REQUIRE(rw >= 0 && rh >= 0, "img_blend_paste: bad rect");
    if (rw == 0 || rh == 0) return ABR_OK;
    ABR_REQUIRE(img && crop, "img_blend_paste: null pointer");
    ABR_REQUIRE(x0 >= 0 && y0 >= 0 && x0 + rw <= W && y0 + rh <= H, "img_blend_paste: target rectangle outside the image");
    ABR_REQUIRE(off_x >= 0 && off_y >= 0 && off_x + rw <= CW && off_y + rh <= CH, "img_blend_paste: source rectangle outside the crop");
    blend_paste_kernel<<<grid_for((int64_t)rw * rh), 256, 0, abr::as_stream(stream)>>>(img, W, crop, CW, x0, y0, rw, rh, off_x, off_y, lam);
    ABR_CHECK_LAUNCH("img_blend_paste");
    return ABR_OK;
}

extern "C" int abr_img_copy_rect_u8(uint8_t* dst, int DH, int DW, const uint8_t* src, int SH, int SW, int dx, int dy, int sx, int sy,
                                    int rw, int rh, void* stream) {
    ABR_REQUIRE(rw >= 0 && rh >= 0, "img_copy_rect: bad rect");
    if (rw == 0 || rh == 0) return ABR_OK;
    ABR_REQUIRE(dst && src, "img_copy_rect: null pointer");
    ABR_REQUIRE(dx >= 0 && dy >= 0 && dx + rw <= DW && dy + rh <= DH, "img_copy_rect: target rectangle outside the image");
    ABR_REQUIRE(sx >= 0 && sy >= 0 && sx + rw <= SW && sy + rh <= SH, "img_copy_rect: source rectangle outside the image");
    copy_rect_kernel<<<grid_for((int64_t)rw * rh), 256, 0, abr::as_stream(stream)>>>(dst, DW, src, SW, dx, dy, sx, sy, rw, rh);
    ABR_CHECK_LAUNCH("img_copy_rect");
    return ABR_OK;
}

extern "C" int abr_img_fill_u8(uint8_t* dst, int64_t n, int value, void* stream) {
    ABR_REQUIRE(n >= 0 && value >= 0 && value <= 255, "img_fill: bad args");
    if (n == 0) return ABR_OK;
    ABR_REQUIRE(dst, "img_fill: null pointer");
    fill_kernel<<<grid_for(n), 256, 0, abr::as_stream(stream)>>>(dst, n, (uint8_t)value);
    ABR_CHECK_LAUNCH("img_fill");
    return ABR_OK;
}

extern "C" int abr_img_normalize_to_batch(const uint8_t* src, int h, int w, int flip, int to_bgr255, const float* mean3_host,
                                          const float* std3_host, float* out_slot, int HP, int WP, void* stream) {
    ABR_REQUIRE(h > 0 && w > 0 && HP >= h && WP >= w, "img_normalize: bad shape");
    ABR_REQUIRE(src && mean3_host && std3_host && out_slot, "img_normalize: null pointer");
    normalize_kernel<<<grid_for((int64_t)HP * WP), 256, 0, abr::as_stream(stream)>>>(src, h, w, flip, to_bgr255, mean3_host[0], mean3_host[1],
                                                                                     mean3_host[2], std3_host[0], std3_host[1],
                                                                                     std3_host[2], out_slot, HP, WP);
    ABR_CHECK_LAUNCH("img_normalize");
    return ABR_OK;
}

/* ColorJitter's pixel ops, in place (include/abr_iod_hip.h section 7).  op: 0 brightness, 1 contrast, 2 saturation (factor = the enhancement factor),
 * 3 hue (factor = hue_factor in [-0.5, 0.5]).  scratch: 8 bytes of device memory (contrast's L sum). */
extern "C" int abr_img_color_jitter_u8(uint8_t* img, int H, int W, int op, double factor, void* scratch8, void* stream) {
    ABR_REQUIRE(H >= 0 && W >= 0 && op >= 0 && op <= 3, "img_color_jitter: bad args");
    const int64_t npix = (int64_t)H * W;
    if (npix == 0) return ABR_OK;
    ABR_REQUIRE(img, "img_color_jitter: null pointer");
    hipStream_t st = abr::as_stream(stream);
    if (op == 3) {
        ABR_REQUIRE(factor >= -0.5 && factor <= 0.5, "img_color_jitter: hue_factor (%g) is not in [-0.5, 0.5]", factor);
        const int shift = ((int)(factor * 255.0) % 256 + 256) % 256;   // uint8(hue_factor * 255): truncation toward zero, modulo 256
        hue_shift_kernel<<<grid_for(npix), 256, 0, st>>>(img, npix, shift);
    } else {
        ABR_REQUIRE(factor >= 0.0, "img_color_jitter: negative enhancement factor");
        unsigned long long* sum = reinterpret_cast<unsigned long long*>(scratch8);
        if (op == 1) {
            ABR_REQUIRE(sum, "img_color_jitter: contrast needs 8 bytes of device scratch");
            if (hipMemsetAsync(sum, 0, 8, st) != hipSuccess) { abr::set_error("img_color_jitter: memset failed"); return ABR_E_LAUNCH; }
            gray_sum_kernel<<<grid_for(npix), 256, 0, st>>>(img, npix, sum);
        }
        color_blend_kernel<<<grid_for(npix), 256, 0, st>>>(img, npix, op, (float)factor, sum);
    }
    ABR_CHECK_LAUNCH("img_color_jitter");
    return ABR_OK;
}
