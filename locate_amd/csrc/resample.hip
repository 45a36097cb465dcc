// Skip-branch resampling and indexing ops of libs/scale.py (all HBM-bound stencils / copies):
//   bilinear x2 upsample, align_corners=False   (scale.py:37-38)   fwd + bwd (adjoint, written as a gather)
//   2x2 average pool, stride 2                  (scale.py:40)      fwd + bwd
//   FeaturePooling                              (scale.py:7-16)    fwd + bwd - a RAW memory view: averages r
//       ADJACENT FLAT elements (neighbouring pixels along W), not channel groups.  Bit-exact indexing.
#include "common.h"

// source index / weight of ATen's upsample_bilinear2d for scale 2, align_corners = False:
//   src = 0.5 * (dst + 0.5) - 0.5, clamped at 0;  i0 = floor(src), i1 = i0 + (i0 < size-1), l1 = src - i0
__device__ __forceinline__ void bil_src(int dst, int size, int& i0, int& i1, float& l0, float& l1) {
    float src = 0.5f * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    i0 = (int)src;
    i1 = i0 + (i0 < size - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

// One output of the bilinear stencil, with the roundings spelled out (two products, two fused multiply-adds per row pair) so that
// every kernel variant below rounds the same way whatever the compiler would contract: they agree bit for bit.
__device__ __forceinline__ float bil_eval(float ly0, float ly1, float lx0, float lx1, float a0, float a1, float b0, float b1) {
    const float top = __fmaf_rn(lx1, a1, __fmul_rn(lx0, a0));
    const float bot = __fmaf_rn(lx1, b1, __fmul_rn(lx0, b0));
    return __fmaf_rn(ly1, bot, __fmul_rn(ly0, top));
}

// POOL (the generator's skip branch: FeaturePooling(r = 2) followed by the upsample, libs/scale.py:7-16,37-38): the kernel's input
// element i is the mean of the RAW elements 2 i, 2 i + 1 of x - what feature_pool_fwd_kernel would have written ((0 + a) + b) * 0.5,
// same bits - so the pooled map is never stored; the backward kernels write the pooled gradient times 0.5 to both raw elements.
template <bool POOL>
__device__ __forceinline__ float pool_ld(const float* __restrict__ x, int64_t i) {
    if (!POOL) return x[i];
    const float2 t = *reinterpret_cast<const float2*>(x + 2 * i);
    return (t.x + t.y) * 0.5f;
}
template <bool POOL>
__device__ __forceinline__ float4 pool_ld4(const float* __restrict__ x, int64_t i) {          // four consecutive elements, i % 4 == 0
    if (!POOL) return *reinterpret_cast<const float4*>(x + i);
    const float4 a = *reinterpret_cast<const float4*>(x + 2 * i), b = *reinterpret_cast<const float4*>(x + 2 * i + 4);
    return make_float4((a.x + a.y) * 0.5f, (a.z + a.w) * 0.5f, (b.x + b.y) * 0.5f, (b.z + b.w) * 0.5f);
}

template <bool POOL>
__global__ void __launch_bounds__(256) upsample2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int64_t planes, int H, int W) {
    const int OH = 2 * H, OW = 2 * W;
    const unsigned n = (unsigned)(planes * OH * OW);
    const unsigned stride = gridDim.x * blockDim.x;
    const DivU32 dw((unsigned)OW), dh((unsigned)OH);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned t, uox, p, uoy;
        dw.divmod(i, t, uox);
        dh.divmod(t, p, uoy);
        const int ox = (int)uox, oy = (int)uoy;
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        bil_src(oy, H, y0, y1, ly0, ly1);
        bil_src(ox, W, x0, x1, lx0, lx1);
        const int64_t xb = (int64_t)p * H * W;
        y[i] = bil_eval(ly0, ly1, lx0, lx1, pool_ld<POOL>(x, xb + y0 * W + x0), pool_ld<POOL>(x, xb + y0 * W + x1),
                        pool_ld<POOL>(x, xb + y1 * W + x0), pool_ld<POOL>(x, xb + y1 * W + x1));
    }
}

// W even: one thread per FOUR consecutive output pixels of a row (one 16-byte store).  Outputs 4j .. 4j+3 read the input
// columns 2j-1 .. 2j+2 (clamped) of two rows: eight L1-served loads for four results instead of sixteen for four, and a
// quarter of the store instructions.  Same expression per element as the scalar kernel: bit-identical results.
template <bool POOL>
__global__ void __launch_bounds__(256) upsample2x_fwd_vec_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 int64_t planes, int H, int W) {
    const int OH = 2 * H, OW4 = W / 2;                    // groups of four output columns per row
    const unsigned n = (unsigned)(planes * OH * OW4);
    const unsigned stride = gridDim.x * blockDim.x;
    const DivU32 dw((unsigned)OW4), dh((unsigned)OH);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned t, j, p, uoy;
        dw.divmod(i, t, j);
        dh.divmod(t, p, uoy);
        int y0, y1;
        float ly0, ly1;
        bil_src((int)uoy, H, y0, y1, ly0, ly1);
        const int64_t r0 = (int64_t)p * H * W + (int64_t)y0 * W, r1 = (int64_t)p * H * W + (int64_t)y1 * W;
        float out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int x0, x1;
            float lx0, lx1;
            bil_src(4 * (int)j + e, W, x0, x1, lx0, lx1);
            out[e] = bil_eval(ly0, ly1, lx0, lx1, pool_ld<POOL>(x, r0 + x0), pool_ld<POOL>(x, r0 + x1), pool_ld<POOL>(x, r1 + x0),
                              pool_ld<POOL>(x, r1 + x1));
        }
        reinterpret_cast<float4*>(y)[i] = make_float4(out[0], out[1], out[2], out[3]);
    }
}

// W a multiple of 4: one thread per 2 x 8 OUTPUT patch (rows 2k, 2k + 1; columns 8j .. 8j + 7), which reads input rows
// k - 1 .. k + 1 and columns 4j - 1 .. 4j + 4 (clamped): per row one aligned 16-byte load plus the two edge columns - nine load
// instructions for sixteen results (the four-wide kernel above: eight for four; the bilinear kernels were bound by their load
// ISSUE rate, 0.28 of the HBM peak at 64 x 64) and four 16-byte stores.  Same expression, weights and source elements per output
// as the scalar kernel: bit-identical results.
template <bool POOL>
__global__ void __launch_bounds__(256) upsample2x_fwd_tile_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  int64_t planes, int H, int W) {
    const int OW = 2 * W, W4 = W / 4;
    const unsigned n = (unsigned)(planes * H * W4);
    const unsigned stride = gridDim.x * blockDim.x;
    const DivU32 dw((unsigned)W4), dh((unsigned)H);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned t, uj, p, uk;
        dw.divmod(i, t, uj);
        dh.divmod(t, p, uk);
        const int j = (int)uj, k = (int)uk;
        const int64_t xb = (int64_t)p * H * W;
        float v[3][6];                       // v[s][c]: input row clamp(k - 1 + s), column clamp(4j - 1 + c)
        const int cl = 4 * j - 1 < 0 ? 0 : 4 * j - 1, cr = 4 * j + 4 > W - 1 ? W - 1 : 4 * j + 4;
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
            int r = k - 1 + s2;
            r = r < 0 ? 0 : (r > H - 1 ? H - 1 : r);
            const int64_t rp = xb + (int64_t)r * W;
            const float4 mid = pool_ld4<POOL>(x, rp + 4 * j);
            v[s2][0] = pool_ld<POOL>(x, rp + cl); v[s2][1] = mid.x; v[s2][2] = mid.y; v[s2][3] = mid.z; v[s2][4] = mid.w;
            v[s2][5] = pool_ld<POOL>(x, rp + cr);
        }
        const bool top = k == 0, bottom = k == H - 1, left = j == 0, right = 4 * j + 3 == W - 1;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            int y0, y1;
            float ly0, ly1;
            bil_src(2 * k + o, H, y0, y1, ly0, ly1);
            // local rows of (y0, y1): output row 2k reads rows (k - 1, k) - at the top edge (k, k + 1) with weight 0 on the
            // second -, output row 2k + 1 rows (k, k + 1) - at the bottom edge (k, k)
            float a[6], b[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                a[c] = o == 0 ? (top ? v[1][c] : v[0][c]) : v[1][c];
                b[c] = o == 0 ? (top ? v[2][c] : v[1][c]) : (bottom ? v[1][c] : v[2][c]);
            }
            float out[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int x0, x1;
                float lx0, lx1;
                bil_src(8 * j + e, W, x0, x1, lx0, lx1);
                // local columns: output column 8j + e reads columns 4j + floor((e - 1) / 2) and the next one; the first output of
                // the row (source clamped to column 0) and the last one (no column to the right) are the exceptions
                const int c0 = (e + 1) / 2;                      // e = 0 -> 0, 1,2 -> 1, 3,4 -> 2, 5,6 -> 3, 7 -> 4
                float a0 = a[c0], a1 = a[c0 + 1], b0 = b[c0], b1 = b[c0 + 1];
                if (e == 0) { a0 = left ? a[1] : a0; a1 = left ? a[2] : a1; b0 = left ? b[1] : b0; b1 = left ? b[2] : b1; }
                if (e == 7) { a1 = right ? a[4] : a1; b1 = right ? b[4] : b1; }
                out[e] = bil_eval(ly0, ly1, lx0, lx1, a0, a1, b0, b1);
            }
            float4* yp = reinterpret_cast<float4*>(y + ((int64_t)p * 2 * H + 2 * k + o) * OW + 8 * j);
            yp[0] = make_float4(out[0], out[1], out[2], out[3]);
            yp[1] = make_float4(out[4], out[5], out[6], out[7]);
        }
    }
}

// adjoint as a gather: input pixel (iy, ix) collects from output rows 2iy-1 .. 2iy+2 (and columns likewise)
__device__ __forceinline__ float bil_weight_to(int dst, int size, int target) {
    int i0, i1;
    float l0, l1;
    bil_src(dst, size, i0, i1, l0, l1);
    float w = 0.0f;
    if (i0 == target) w += l0;
    if (i1 == target) w += l1;
    return w;
}

template <bool POOL>
__global__ void __launch_bounds__(256) upsample2x_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                             int64_t planes, int H, int W, int accumulate) {
    const int OH = 2 * H, OW = 2 * W;
    const unsigned n = (unsigned)(planes * H * W);
    const unsigned stride = gridDim.x * blockDim.x;
    const DivU32 dw((unsigned)W), dh((unsigned)H);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned t, uix, p, uiy;
        dw.divmod(i, t, uix);
        dh.divmod(t, p, uiy);
        const int ix = (int)uix, iy = (int)uiy;
        const float* gp = gy + (int64_t)p * OH * OW;
        float wy[4], wx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oy = 2 * iy - 1 + k, ox = 2 * ix - 1 + k;
            wy[k] = (oy >= 0 && oy < OH) ? bil_weight_to(oy, H, iy) : 0.0f;
            wx[k] = (ox >= 0 && ox < OW) ? bil_weight_to(ox, W, ix) : 0.0f;
        }
        float acc = 0.0f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (wy[a] == 0.0f) continue;
            const int oy = 2 * iy - 1 + a;
            float row = 0.0f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ox = 2 * ix - 1 + b;
                if (wx[b] != 0.0f) row = fmaf(wx[b], gp[oy * OW + ox], row);
            }
            acc = fmaf(wy[a], row, acc);
        }
        if (POOL) {          // feature_pool_bwd_kernel's values: g * (1 / 2) to both raw elements, added to what is there on request
            float2* dst = reinterpret_cast<float2*>(gx) + i;
            const float o = acc * 0.5f;
            float2 w = make_float2(o, o);
            if (accumulate) { const float2 old = *dst; w.x = old.x + o; w.y = old.y + o; }
            *dst = w;
        } else {
            gx[i] = acc;
        }
    }
}

// W a multiple of 4: one thread per FOUR consecutive input-gradient pixels 4j .. 4j+3 of a row.  They collect from the
// output columns 8j-1 .. 8j+8 of four output rows: two aligned float4 plus the two border columns per row - sixteen load
// instructions for four results instead of sixty-four.  The per-element sums run in the scalar kernel's order (rows outer,
// columns inner, zero-weight taps skipped): bit-identical results.
template <bool POOL>
__global__ void __launch_bounds__(256) upsample2x_bwd_vec_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                 int64_t planes, int H, int W, int accumulate) {
    const int OH = 2 * H, OW = 2 * W, W4 = W / 4;
    const unsigned n = (unsigned)(planes * H * W4);
    const unsigned stride = gridDim.x * blockDim.x;
    const DivU32 dw((unsigned)W4), dh((unsigned)H);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned t, j, p, uiy;
        dw.divmod(i, t, j);
        dh.divmod(t, p, uiy);
        const int iy = (int)uiy, ix0 = 4 * (int)j, c0 = 8 * (int)j;      // c0: first of the eight aligned output columns
        const float* gp = gy + (int64_t)p * OH * OW;
        float wy[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oy = 2 * iy - 1 + k;
            wy[k] = (oy >= 0 && oy < OH) ? bil_weight_to(oy, H, iy) : 0.0f;
        }
        float wx[4][4];                                   // [element][tap]: output column c0 - 1 + 2 e + tap -> input column ix0 + e
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ox = c0 - 1 + 2 * e + k;
                wx[e][k] = (ox >= 0 && ox < OW) ? bil_weight_to(ox, W, ix0 + e) : 0.0f;
            }
        const bool has_left = c0 > 0, has_right = c0 + 8 < OW;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (wy[a] == 0.0f) continue;
            const float* rp = gp + (int64_t)(2 * iy - 1 + a) * OW + c0;
            const float4 q1 = *reinterpret_cast<const float4*>(rp), q2 = *reinterpret_cast<const float4*>(rp + 4);
            float v[10];
            v[0] = has_left ? rp[-1] : 0.0f;
            v[1] = q1.x; v[2] = q1.y; v[3] = q1.z; v[4] = q1.w; v[5] = q2.x; v[6] = q2.y; v[7] = q2.z; v[8] = q2.w;
            v[9] = has_right ? rp[8] : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float row = 0.0f;
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (wx[e][b] != 0.0f) row = fmaf(wx[e][b], v[2 * e + b], row);
                acc[e] = fmaf(wy[a], row, acc[e]);
            }
        }
        if (POOL) {
            float4* dst = reinterpret_cast<float4*>(gx) + 2 * (int64_t)i;
            float4 lo = make_float4(acc[0] * 0.5f, acc[0] * 0.5f, acc[1] * 0.5f, acc[1] * 0.5f);
            float4 hi = make_float4(acc[2] * 0.5f, acc[2] * 0.5f, acc[3] * 0.5f, acc[3] * 0.5f);
            if (accumulate) {
                const float4 a = dst[0], b = dst[1];
                lo.x = a.x + lo.x; lo.y = a.y + lo.y; lo.z = a.z + lo.z; lo.w = a.w + lo.w;
                hi.x = b.x + hi.x; hi.y = b.y + hi.y; hi.z = b.z + hi.z; hi.w = b.w + hi.w;
            }
            dst[0] = lo; dst[1] = hi;
        } else {
            reinterpret_cast<float4*>(gx)[i] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
}

__global__ void __launch_bounds__(256) avgpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t planes,
                                                           int H, int W) {
    const int OH = H / 2, OW = W / 2;
    const unsigned n = (unsigned)(planes * OH * OW);
    const unsigned stride = gridDim.x * blockDim.x;
    const DivU32 dw((unsigned)OW), dh((unsigned)OH);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned t, ox, p, oy;
        dw.divmod(i, t, ox);
        dh.divmod(t, p, oy);
        const float* xp = x + (int64_t)p * H * W + (2 * oy) * W + 2 * ox;
        y[i] = ((xp[0] + xp[1]) + (xp[W] + xp[W + 1])) * 0.25f;
    }
}

// W a multiple of 8: one thread per FOUR consecutive outputs of a row - two 32-byte row segments in, one 16-byte store out (the
// scalar kernel above issues four 4-byte loads per output); the same sum per output: bit-identical.
__global__ void __launch_bounds__(256) avgpool2_fwd_vec_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t planes,
                                                               int H, int W) {
    const int OH = H / 2, OW4 = W / 8;
    const unsigned n = (unsigned)(planes * OH * OW4);
    const unsigned stride = gridDim.x * blockDim.x;
    const DivU32 dw((unsigned)OW4), dh((unsigned)OH);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned t, j, p, oy;
        dw.divmod(i, t, j);
        dh.divmod(t, p, oy);
        const float* r0 = x + (int64_t)p * H * W + (int64_t)(2 * oy) * W + 8 * j;
        const float4 a0 = reinterpret_cast<const float4*>(r0)[0], a1 = reinterpret_cast<const float4*>(r0)[1];
        const float4 b0 = reinterpret_cast<const float4*>(r0 + W)[0], b1 = reinterpret_cast<const float4*>(r0 + W)[1];
        float4 o;
        o.x = ((a0.x + a0.y) + (b0.x + b0.y)) * 0.25f;
        o.y = ((a0.z + a0.w) + (b0.z + b0.w)) * 0.25f;
        o.z = ((a1.x + a1.y) + (b1.x + b1.y)) * 0.25f;
        o.w = ((a1.z + a1.w) + (b1.z + b1.w)) * 0.25f;
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

__global__ void __launch_bounds__(256) avgpool2_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                           int64_t planes, int H, int W, int accumulate) {
    const int OH = H / 2, OW = W / 2;
    const unsigned n = (unsigned)(planes * H * W);
    const unsigned stride = gridDim.x * blockDim.x;
    const DivU32 dw((unsigned)W), dh((unsigned)H);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned t, ix, p, iy;
        dw.divmod(i, t, ix);
        dh.divmod(t, p, iy);
        const int oy = (int)(iy >> 1), ox = (int)(ix >> 1);
        const float o = (oy < OH && ox < OW) ? gy[(int64_t)p * OH * OW + oy * OW + ox] * 0.25f : 0.0f;
        gx[i] = accumulate ? gx[i] + o : o;
    }
}

// H, W multiples of 2 and W of 4: one thread per four consecutive input-gradient pixels (two pooled values, one 16-byte store)
__global__ void __launch_bounds__(256) avgpool2_bwd_vec_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                               int64_t planes, int H, int W, int accumulate) {
    const int OW = W / 2, W4 = W / 4;
    const unsigned n = (unsigned)(planes * H * W4);
    const unsigned stride = gridDim.x * blockDim.x;
    const DivU32 dw((unsigned)W4), dh((unsigned)H);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned t, ux, p, iy;
        dw.divmod(i, t, ux);
        dh.divmod(t, p, iy);
        const float2 g = *reinterpret_cast<const float2*>(gy + ((int64_t)p * (H / 2) + (iy >> 1)) * OW + 2 * ux);
        const float a = g.x * 0.25f, b = g.y * 0.25f;
        float4 o = make_float4(a, a, b, b);
        if (accumulate) {
            const float4 c = reinterpret_cast<const float4*>(gx)[i];
            o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
        }
        reinterpret_cast<float4*>(gx)[i] = o;
    }
}

// y[k] = mean(x[r k .. r k + r - 1]) over the flat buffer;  dx[r k + j] = g[k] / r
__global__ void __launch_bounds__(256) feature_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int64_t n_out, int r) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float inv = 1.0f / (float)r;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += stride) {
        float acc = 0.0f;
        for (int j = 0; j < r; ++j) acc += x[i * r + j];
        y[i] = r == 2 ? acc * 0.5f : acc / (float)r;
        (void)inv;
    }
}

__global__ void __launch_bounds__(256) feature_pool_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                               int64_t n_out, int r, int accumulate) {
    const unsigned n = (unsigned)(n_out * r);
    const unsigned stride = gridDim.x * blockDim.x;
    const DivU32 dr((unsigned)r);
    const bool pow2 = dr.shift >= 0;
    const float inv = 1.0f / (float)r;         // exact for powers of two; other ratios keep the division (same rounding as ATen)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float g = gy[dr.div(i)];
        const float o = pow2 ? g * inv : g / (float)r;
        gx[i] = accumulate ? gx[i] + o : o;
    }
}

#define RESAMPLE_ENTRY(NAME, KERNEL, WORK)                                                                       \
    LOCATE_API int NAME(const float* a, float* b, int64_t planes, int H, int W, void* stream) {                  \
        LOCATE_REQUIRE(planes > 0 && H > 0 && W > 0 && planes * 4 * H * W < (1ll << 31), #NAME ": bad shape");   \
        KERNEL<<<stream_grid((WORK), 256), 256, 0, as_stream(stream)>>>(a, b, planes, H, W);                     \
        LOCATE_LAUNCH_CHECK(#NAME);                                                                              \
        return LOCATE_OK;                                                                                        \
    }

template <bool POOL>
static void launch_upsample_fwd(const float* a, float* b, int64_t planes, int H, int W, hipStream_t st) {
    const int al = POOL ? 31 : 15;          // (the pooled form reads 32-byte groups of raw elements)
    if ((W & 3) == 0 && W >= 8 && ((uintptr_t)a & al) == 0 && ((uintptr_t)b & 15) == 0)
        upsample2x_fwd_tile_kernel<POOL><<<stream_grid(planes * H * (W / 4), 256), 256, 0, st>>>(a, b, planes, H, W);
    else if ((W & 1) == 0 && ((uintptr_t)b & 15) == 0 && (!POOL || ((uintptr_t)a & 7) == 0))
        upsample2x_fwd_vec_kernel<POOL><<<stream_grid(planes * 2 * H * (W / 2), 256), 256, 0, st>>>(a, b, planes, H, W);
    else
        upsample2x_fwd_kernel<POOL><<<stream_grid(planes * 4 * H * W, 256), 256, 0, st>>>(a, b, planes, H, W);
}
template <bool POOL>
static void launch_upsample_bwd(const float* a, float* b, int64_t planes, int H, int W, int accumulate, hipStream_t st) {
    if ((W & 3) == 0 && W >= 8 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0)
        upsample2x_bwd_vec_kernel<POOL><<<stream_grid(planes * H * (W / 4), 256), 256, 0, st>>>(a, b, planes, H, W, accumulate);
    else
        upsample2x_bwd_kernel<POOL><<<stream_grid(planes * H * W, 256), 256, 0, st>>>(a, b, planes, H, W, accumulate);
}

// x: [planes, H, W] -> y: [planes, 2H, 2W]
LOCATE_API int locate_upsample2x_fwd(const float* a, float* b, int64_t planes, int H, int W, void* stream) {
    LOCATE_REQUIRE(planes > 0 && H > 0 && W > 0 && planes * 4 * H * W < (1ll << 31), "locate_upsample2x_fwd: bad shape");
    launch_upsample_fwd<false>(a, b, planes, H, W, as_stream(stream));
    LOCATE_LAUNCH_CHECK("locate_upsample2x_fwd");
    return LOCATE_OK;
}
// gy: [planes, 2H, 2W] -> gx: [planes, H, W]
LOCATE_API int locate_upsample2x_bwd(const float* a, float* b, int64_t planes, int H, int W, void* stream) {
    LOCATE_REQUIRE(planes > 0 && H > 0 && W > 0 && planes * 4 * H * W < (1ll << 31), "locate_upsample2x_bwd: bad shape");
    launch_upsample_bwd<false>(a, b, planes, H, W, 0, as_stream(stream));
    LOCATE_LAUNCH_CHECK("locate_upsample2x_bwd");
    return LOCATE_OK;
}
// FeaturePooling(r = 2) + bilinear x2 upsample in one launch (the generator's skip branch, libs/scale.py:7-16,37-38):
// x: the RAW tensor, 2 * planes * H * W elements (8-byte aligned) -> y: [planes, 2H, 2W]; bit for bit locate_feature_pool_fwd(r = 2)
// followed by locate_upsample2x_fwd
LOCATE_API int locate_pool2_upsample2x_fwd(const float* x, float* y, int64_t planes, int H, int W, void* stream) {
    LOCATE_REQUIRE(planes > 0 && H > 0 && W > 0 && planes * 4 * H * W < (1ll << 31) && ((uintptr_t)x & 7) == 0,
                   "locate_pool2_upsample2x_fwd: bad shape or unaligned input");
    launch_upsample_fwd<true>(x, y, planes, H, W, as_stream(stream));
    LOCATE_LAUNCH_CHECK("locate_pool2_upsample2x_fwd");
    return LOCATE_OK;
}
// its adjoint: gy [planes, 2H, 2W] -> gx, the RAW gradient of 2 * planes * H * W elements (accumulate != 0: added to what is there)
LOCATE_API int locate_pool2_upsample2x_bwd(const float* gy, float* gx, int64_t planes, int H, int W, int accumulate, void* stream) {
    LOCATE_REQUIRE(planes > 0 && H > 0 && W > 0 && planes * 4 * H * W < (1ll << 31) && ((uintptr_t)gx & 7) == 0,
                   "locate_pool2_upsample2x_bwd: bad shape or unaligned output");
    if ((W & 3) == 0 && W >= 8 && ((uintptr_t)gx & 31) != 0)          // (the four-wide kernel writes 32-byte groups)
        upsample2x_bwd_kernel<true><<<stream_grid(planes * H * W, 256), 256, 0, as_stream(stream)>>>(gy, gx, planes, H, W, accumulate);
    else
        launch_upsample_bwd<true>(gy, gx, planes, H, W, accumulate, as_stream(stream));
    LOCATE_LAUNCH_CHECK("locate_pool2_upsample2x_bwd");
    return LOCATE_OK;
}
// x: [planes, H, W] -> y: [planes, H/2, W/2]
LOCATE_API int locate_avgpool2_fwd(const float* a, float* b, int64_t planes, int H, int W, void* stream) {
    LOCATE_REQUIRE(planes > 0 && H > 0 && W > 0 && planes * 4 * H * W < (1ll << 31), "locate_avgpool2_fwd: bad shape");
    if ((W & 7) == 0 && (H & 1) == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0)
        avgpool2_fwd_vec_kernel<<<stream_grid(planes * (H / 2) * (W / 8), 256), 256, 0, as_stream(stream)>>>(a, b, planes, H, W);
    else
        avgpool2_fwd_kernel<<<stream_grid(planes * (H / 2) * (W / 2), 256), 256, 0, as_stream(stream)>>>(a, b, planes, H, W);
    LOCATE_LAUNCH_CHECK("locate_avgpool2_fwd");
    return LOCATE_OK;
}
// gy: [planes, H/2, W/2] -> gx: [planes, H, W]   (H, W are the INPUT sizes of the forward)
// accumulate != 0: gx += ... (the second backward kernel of a forked tensor, see ops.fork)
LOCATE_API int locate_avgpool2_bwd(const float* a, float* b, int64_t planes, int H, int W, int accumulate, void* stream) {
    LOCATE_REQUIRE(planes > 0 && H > 0 && W > 0 && planes * 4 * H * W < (1ll << 31), "locate_avgpool2_bwd: bad shape");
    if ((W & 3) == 0 && (H & 1) == 0 && (((uintptr_t)a & 7) | ((uintptr_t)b & 15)) == 0)
        avgpool2_bwd_vec_kernel<<<stream_grid(planes * H * (W / 4), 256), 256, 0, as_stream(stream)>>>(a, b, planes, H, W, accumulate);
    else
        avgpool2_bwd_kernel<<<stream_grid(planes * H * W, 256), 256, 0, as_stream(stream)>>>(a, b, planes, H, W, accumulate);
    LOCATE_LAUNCH_CHECK("locate_avgpool2_bwd");
    return LOCATE_OK;
}

LOCATE_API int locate_feature_pool_fwd(const float* x, float* y, int64_t n_out, int r, void* stream) {
    LOCATE_REQUIRE(n_out > 0 && r > 0, "locate_feature_pool_fwd: bad shape");
    feature_pool_fwd_kernel<<<stream_grid(n_out, 256), 256, 0, as_stream(stream)>>>(x, y, n_out, r);
    LOCATE_LAUNCH_CHECK("locate_feature_pool_fwd");
    return LOCATE_OK;
}

LOCATE_API int locate_feature_pool_bwd(const float* gy, float* gx, int64_t n_out, int r, int accumulate, void* stream) {
    LOCATE_REQUIRE(n_out > 0 && r > 0 && n_out * r < (1ll << 31), "locate_feature_pool_bwd: bad shape");
    feature_pool_bwd_kernel<<<stream_grid(n_out * r, 256), 256, 0, as_stream(stream)>>>(gy, gx, n_out, r, accumulate);
    LOCATE_LAUNCH_CHECK("locate_feature_pool_bwd");
    return LOCATE_OK;
}
