// Element-wise kernels of the LocAtE hot path (all HBM-bound, 16 B per lane, grid-strided):
//   RootTanh fwd/bwd        reference libs/activation.py:7-36
//   tanh fwd/bwd            reference libs/models.py:66 (generator output)
//   residual gate fwd/bwd   reference libs/merge.py:19-39 (incl. the x^2 gamma-gradient as coded)
//   channel-slice copy      reference libs/merge.py:15 (torch.cat along channels) and its backward
#include "common.h"

// ---------------------------------------------------------------------------------------------
// RootTanh:  y = (x^2+1)^(1/4) * tanh(x)
//   dy/dx = [2 (x^2+1) sech^2(x) + x tanh(x)] / (2 (x^2+1)^(3/4))          (ROOTTANH_GROWTH = 4)
// tanh and sech^2 come from one expm1: em = expm1(-2|x|) in (-1, 0]:
//   tanh|x| = -em / (2 + em),  sech^2 x = 4 (1 + em) / (2 + em)^2
// which is accurate for tiny |x| and saturates cleanly (sech^2 -> 0, tanh -> 1) for large |x|, where the
// reference's 1/cosh^2 overflows to 1/inf = 0 (activation.py:24-26) - same limit, finite result.
// ---------------------------------------------------------------------------------------------
template <int OP>  // 0 roottanh, 1 tanh
__global__ void __launch_bounds__(256) unary_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                        unsigned* __restrict__ absmax) {
    __shared__ float amax_scratch[16];
    float am = 0.0f;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = x4[i], o;
        if (OP == 0) {
            o.x = roottanh_f(v.x); o.y = roottanh_f(v.y); o.z = roottanh_f(v.z); o.w = roottanh_f(v.w);
        } else {
            o.x = tanhf(v.x); o.y = tanhf(v.y); o.z = tanhf(v.z); o.w = tanhf(v.w);
        }
        y4[i] = o;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float o = OP == 0 ? roottanh_f(x[i]) : tanhf(x[i]);
        y[i] = o;
        am = fmaxf(am, fabsf(o));
    }
    if (absmax) absmax_publish(am, amax_scratch, absmax);      // largest |y| for the fp16-piece contractions (common.h)
}

// OP 0: gx = g * roottanh'(a) with a = forward INPUT.   OP 1: gx = g * (1 - a^2) with a = forward OUTPUT.
template <int OP>
__global__ void __launch_bounds__(256) unary_bwd_kernel(const float* __restrict__ a, const float* __restrict__ g,
                                                        float* __restrict__ gx, int64_t n, int accumulate,
                                                        unsigned* __restrict__ absmax) {
    __shared__ float amax_scratch[16];
    float am = 0.0f;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* o4 = reinterpret_cast<float4*>(gx);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = a4[i], w = g4[i], o;
        if (OP == 0) {
            o.x = roottanh_grad_f(v.x, w.x); o.y = roottanh_grad_f(v.y, w.y);
            o.z = roottanh_grad_f(v.z, w.z); o.w = roottanh_grad_f(v.w, w.w);
        } else {
            o.x = w.x * (1.0f - v.x * v.x); o.y = w.y * (1.0f - v.y * v.y);
            o.z = w.z * (1.0f - v.z * v.z); o.w = w.w * (1.0f - v.w * v.w);
        }
        if (accumulate) {
            const float4 c = o4[i];
            o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
        }
        o4[i] = o;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float o = OP == 0 ? roottanh_grad_f(a[i], g[i]) : g[i] * (1.0f - a[i] * a[i]);
        o = accumulate ? gx[i] + o : o;
        gx[i] = o;
        am = fmaxf(am, fabsf(o));
    }
    if (absmax) absmax_publish(am, amax_scratch, absmax);      // largest |gx| (after accumulation) for the fp16-piece contractions
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

LOCATE_API int locate_roottanh_fwd(const float* x, float* y, int64_t n, void* absmax, void* stream) {
    LOCATE_REQUIRE(n >= 0 && aligned16(x) && aligned16(y), "locate_roottanh_fwd: bad size or unaligned pointer");
    if (n == 0) return LOCATE_OK;
    unary_fwd_kernel<0><<<stream_grid(n, 1024), 256, 0, as_stream(stream)>>>(x, y, n, static_cast<unsigned*>(absmax));
    LOCATE_LAUNCH_CHECK("locate_roottanh_fwd");
    return LOCATE_OK;
}

// accumulate != 0: gx += ... (the second backward kernel of a forked tensor, see ops.fork)
LOCATE_API int locate_roottanh_bwd(const float* x, const float* gy, float* gx, int64_t n, int accumulate, void* absmax, void* stream) {
    LOCATE_REQUIRE(n >= 0 && aligned16(x) && aligned16(gy) && aligned16(gx), "locate_roottanh_bwd: bad size or unaligned pointer");
    if (n == 0) return LOCATE_OK;
    unary_bwd_kernel<0><<<stream_grid(n, 1024), 256, 0, as_stream(stream)>>>(x, gy, gx, n, accumulate, static_cast<unsigned*>(absmax));
    LOCATE_LAUNCH_CHECK("locate_roottanh_bwd");
    return LOCATE_OK;
}

// ---------------------------------------------------------------------------------------------
// Style chain link (libs/block.py:119-125): the next style linear sees cat([latent, RootTanh(pre)], dim = 1).  One launch
// writes that concatenation - out[r, :z] = latent[r, :], out[r, z:] = RootTanh(pre[r, :]) - instead of an activation
// launch plus two copy launches; the backward takes the gradient's column slice in place (row stride = z + w).
// The tensors are tiny ([batch, <= 1000]); one thread per output element.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) act_cat_rows_kernel(const float* __restrict__ latent, const float* __restrict__ pre,
                                                           float* __restrict__ out, int rows, int z, int w) {
    const int total = rows * (z + w);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / (z + w), c = i - r * (z + w);
        out[i] = c < z ? latent[r * z + c] : roottanh_f(pre[r * w + (c - z)]);
    }
}

// gpre[r, j] = g[r * g_rs + j] * RootTanh'(pre[r, j])  (+ g_add[r, j]: the gradient pre receives as a norm's style scale -
// the sum autograd would otherwise form with a launch of its own; a separately rounded add, bit for bit the same sum)
__global__ void __launch_bounds__(256) act_rows_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ g, int g_rs,
                                                           const float* __restrict__ g_add, float* __restrict__ gpre, int rows, int w) {
    const int total = rows * w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / w, j = i - r * w;
        const float v = roottanh_grad_f(pre[i], g[(int64_t)r * g_rs + j]);
        gpre[i] = g_add ? __fadd_rn(g_add[i], v) : v;
    }
}

LOCATE_API int locate_act_cat_rows_fwd(const float* latent, const float* pre, float* out, int rows, int z, int w, void* stream) {
    LOCATE_REQUIRE(latent && pre && out && rows > 0 && z >= 0 && w > 0 && (int64_t)rows * (z + w) < (1ll << 31), "locate_act_cat_rows_fwd: bad arguments");
    act_cat_rows_kernel<<<stream_grid((int64_t)rows * (z + w), 256), 256, 0, as_stream(stream)>>>(latent, pre, out, rows, z, w);
    LOCATE_LAUNCH_CHECK("locate_act_cat_rows_fwd");
    return LOCATE_OK;
}

LOCATE_API int locate_act_rows_bwd(const float* pre, const float* g, int64_t g_row_stride, const float* g_add, float* gpre, int rows,
                                   int w, void* stream) {
    LOCATE_REQUIRE(pre && g && gpre && rows > 0 && w > 0 && g_row_stride >= w && g_row_stride < (1ll << 31), "locate_act_rows_bwd: bad arguments");
    act_rows_bwd_kernel<<<stream_grid((int64_t)rows * w, 256), 256, 0, as_stream(stream)>>>(pre, g, (int)g_row_stride, g_add, gpre, rows, w);
    LOCATE_LAUNCH_CHECK("locate_act_rows_bwd");
    return LOCATE_OK;
}

// out[b][i] = (a[b][i] + b_[b][i]) + c[b][i]: the three gradients that meet at a tensor with three consumers (the discriminator
// block's input: norm of the conv branch, identity half of the concatenation, the skip branch's 1x1 conv), in ONE pass and in a
// fixed order instead of autograd's two adds.  Each operand is contiguous inside a batch element and has its own batch stride
// (the concatenation's gradient is consumed as the channel slice it is).
template <bool VEC>
__global__ void __launch_bounds__(256) add3_kernel(const float* __restrict__ a, long long a_bs, const float* __restrict__ b, long long b_bs,
                                                   const float* __restrict__ c, long long c_bs, float* __restrict__ out, unsigned per,
                                                   long long total) {
    const DivU32 dv(per);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        unsigned q, r;
        dv.divmod((unsigned)i, q, r);
        if constexpr (VEC) {
            const float4 x = *reinterpret_cast<const float4*>(a + q * a_bs + 4ll * r);
            const float4 y = *reinterpret_cast<const float4*>(b + q * b_bs + 4ll * r);
            const float4 z = *reinterpret_cast<const float4*>(c + q * c_bs + 4ll * r);
            float4 o;
            o.x = __fadd_rn(__fadd_rn(x.x, y.x), z.x); o.y = __fadd_rn(__fadd_rn(x.y, y.y), z.y);
            o.z = __fadd_rn(__fadd_rn(x.z, y.z), z.z); o.w = __fadd_rn(__fadd_rn(x.w, y.w), z.w);
            *reinterpret_cast<float4*>(out + 4ll * i) = o;
        } else {
            out[i] = __fadd_rn(__fadd_rn(a[q * a_bs + r], b[q * b_bs + r]), c[q * c_bs + r]);
        }
    }
}

LOCATE_API int locate_add3(const float* a, int64_t a_bs, const float* b, int64_t b_bs, const float* c, int64_t c_bs, float* out, int batch,
                           int64_t per, void* stream) {
    LOCATE_REQUIRE(a && b && c && out && batch > 0 && per > 0 && a_bs >= per && b_bs >= per && c_bs >= per &&
                   (int64_t)batch * per < (1ll << 32), "locate_add3: bad arguments");
    const bool vec = (per & 3) == 0 && ((a_bs | b_bs | c_bs) & 3) == 0 && aligned16(a) && aligned16(b) && aligned16(c) && aligned16(out);
    if (vec) {
        const int64_t total = (int64_t)batch * (per / 4);
        add3_kernel<true><<<stream_grid(total, 2048), 256, 0, as_stream(stream)>>>(a, a_bs, b, b_bs, c, c_bs, out, (unsigned)(per / 4), total);
    } else {
        const int64_t total = (int64_t)batch * per;
        add3_kernel<false><<<stream_grid(total, 2048), 256, 0, as_stream(stream)>>>(a, a_bs, b, b_bs, c, c_bs, out, (unsigned)per, total);
    }
    LOCATE_LAUNCH_CHECK("locate_add3");
    return LOCATE_OK;
}

LOCATE_API int locate_absmax_words(void) { return AMAX_WORDS; }

// largest magnitude of x folded into the AMAX_WORDS words at slot (atomic max on the bit patterns; the caller zeroes them first)
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ slot) {
    __shared__ float scratch[16];
    float m = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
    absmax_publish(m, scratch, slot);
}

LOCATE_API int locate_absmax(const float* x, int64_t n, void* slot, void* stream) {
    LOCATE_REQUIRE(x && slot && n > 0, "locate_absmax: bad arguments");
    absmax_kernel<<<stream_grid(n, 2048), 256, 0, as_stream(stream)>>>(x, n, static_cast<unsigned*>(slot));
    LOCATE_LAUNCH_CHECK("locate_absmax");
    return LOCATE_OK;
}

LOCATE_API int locate_tanh_fwd(const float* x, float* y, int64_t n, void* stream) {
    LOCATE_REQUIRE(n >= 0 && aligned16(x) && aligned16(y), "locate_tanh_fwd: bad size or unaligned pointer");
    if (n == 0) return LOCATE_OK;
    unary_fwd_kernel<1><<<stream_grid(n, 1024), 256, 0, as_stream(stream)>>>(x, y, n, nullptr);
    LOCATE_LAUNCH_CHECK("locate_tanh_fwd");
    return LOCATE_OK;
}

LOCATE_API int locate_tanh_bwd(const float* y, const float* gy, float* gx, int64_t n, void* stream) {
    LOCATE_REQUIRE(n >= 0 && aligned16(y) && aligned16(gy) && aligned16(gx), "locate_tanh_bwd: bad size or unaligned pointer");
    if (n == 0) return LOCATE_OK;
    unary_bwd_kernel<1><<<stream_grid(n, 1024), 256, 0, as_stream(stream)>>>(y, gy, gx, n, 0, nullptr);
    LOCATE_LAUNCH_CHECK("locate_tanh_bwd");
    return LOCATE_OK;
}

// ---------------------------------------------------------------------------------------------
// Residual gate: out = (gamma * a + 1) * x     (merge.py:21-28)
//   a is either a full [planes, hw] tensor (a_per_plane = 0) or one value per (batch, channel) plane
//   (a_per_plane = 1: feature attention's [B, C, 1, 1] expanded with stride 0, util_modules.py:6-12).
//   gamma is a device scalar (the [1,1] Parameter).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gate_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                       const float* __restrict__ gamma, float* __restrict__ out,
                                                       int64_t n, int hw, int a_per_plane) {
    const float gm = gamma[0];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (!a_per_plane && (n & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* a4 = reinterpret_cast<const float4*>(a);
        float4* o4 = reinterpret_cast<float4*>(out);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n >> 2); i += stride) {
            float4 xv = x4[i], av = a4[i], o;
            o.x = fmaf(gm, av.x, 1.0f) * xv.x; o.y = fmaf(gm, av.y, 1.0f) * xv.y;
            o.z = fmaf(gm, av.z, 1.0f) * xv.z; o.w = fmaf(gm, av.w, 1.0f) * xv.w;
            o4[i] = o;
        }
        return;
    }
    if (a_per_plane && (hw & 3) == 0) {          // one gate value per plane: 16-byte accesses (see gate_fwd_stats_kernel)
        const float4* x4 = reinterpret_cast<const float4*>(x);
        float4* o4 = reinterpret_cast<float4*>(out);
        const unsigned hw4 = (unsigned)(hw >> 2);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n >> 2); i += stride) {
            const float m = fmaf(gm, a[(unsigned)i / hw4], 1.0f);
            const float4 xv = x4[i];
            o4[i] = make_float4(m * xv.x, m * xv.y, m * xv.z, m * xv.w);
        }
        return;
    }
    const DivU32 dhw((unsigned)hw);          // n < 2^31 (host entry)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float av = a_per_plane ? a[dhw.div((unsigned)i)] : a[i];
        out[i] = fmaf(gm, av, 1.0f) * x[i];
    }
}

// The same gate, and - in the epilogue of the kernel that produces the tensor - the partial sums (sum, sum of squares, fp64,
// fixed order) of the InPlaceNorm that almost always consumes a gate's output next (libs/block.py:44-52: every block output
// is a gate output, every block / attention branch starts with a norm).  Layout = stats_partial_kernel's (norm.hip):
// blockIdx.y = group of stacked calls, partial[group][2 * gridDim.x]; the norm then skips its own statistics pass - one
// full read of the tensor and one launch less.
__global__ void __launch_bounds__(256) gate_fwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                             const float* __restrict__ gamma, float* __restrict__ out,
                                                             int64_t n_group, int hw, int a_per_plane,
                                                             double* __restrict__ partial) {
    __shared__ double scratch[16];
    const float gm = gamma[0];
    const int64_t g0 = (int64_t)blockIdx.y * n_group;
    x += g0; out += g0;
    if (!a_per_plane) a += g0;
    partial += (int64_t)blockIdx.y * 2 * gridDim.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double s = 0.0, q = 0.0;
    if (!a_per_plane && (n_group & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* a4 = reinterpret_cast<const float4*>(a);
        float4* o4 = reinterpret_cast<float4*>(out);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n_group >> 2); i += stride) {
            float4 xv = x4[i], av = a4[i], o;
            o.x = fmaf(gm, av.x, 1.0f) * xv.x; o.y = fmaf(gm, av.y, 1.0f) * xv.y;
            o.z = fmaf(gm, av.z, 1.0f) * xv.z; o.w = fmaf(gm, av.w, 1.0f) * xv.w;
            o4[i] = o;
            const double a_ = o.x, b_ = o.y, c_ = o.z, d_ = o.w;
            s += (a_ + b_) + (c_ + d_);
            q += (a_ * a_ + b_ * b_) + (c_ * c_ + d_ * d_);
        }
    } else if (a_per_plane && (hw & 3) == 0 && (n_group & 3) == 0 && n_group < (1ll << 33)) {
        // one gate value per plane (the channel gate: a is [B, C, 1, 1]): 16-byte accesses, one 32-bit division per four elements
        // (this path used to be the scalar loop below: 52 us for the 50 MB map of the 64x64 stage, 19 us now)
        const float4* x4 = reinterpret_cast<const float4*>(x);
        float4* o4 = reinterpret_cast<float4*>(out);
        const unsigned hw4 = (unsigned)(hw >> 2);
        const float* ap = a + g0 / hw;                   // groups hold whole planes
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n_group >> 2); i += stride) {
            const float m = fmaf(gm, ap[(unsigned)i / hw4], 1.0f);
            const float4 xv = x4[i];
            float4 o;
            o.x = m * xv.x; o.y = m * xv.y; o.z = m * xv.z; o.w = m * xv.w;
            o4[i] = o;
            const double a_ = o.x, b_ = o.y, c_ = o.z, d_ = o.w;
            s += (a_ + b_) + (c_ + d_);
            q += (a_ * a_ + b_ * b_) + (c_ * c_ + d_ * d_);
        }
    } else {
        const DivU32 dhw((unsigned)hw);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_group; i += stride) {
            const float av = a_per_plane ? a[dhw.div((unsigned)(g0 + i))] : a[i];
            const float o = fmaf(gm, av, 1.0f) * x[i];
            out[i] = o;
            s += (double)o;
            q += (double)o * (double)o;
        }
    }
    s = block_sum<double>(s, scratch);
    q = block_sum<double>(q, scratch);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = s;
        partial[2 * blockIdx.x + 1] = q;
    }
}

// dx, da (full, or gamma * sum x*g per plane in broadcast mode) and, per BLOCK, the sum of x^2*g over the planes the
// block visited (fixed order: deterministic).  WPP waves share a plane: 1 for small planes (one wave per plane),
// 4 (the whole block) for planes of >= 1024 elements, 16-byte accesses when hw % 4 == 0.
template <int WPP>
__global__ void __launch_bounds__(256) gate_bwd_plane_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                             const float* __restrict__ gamma, const float* __restrict__ g,
                                                             float* __restrict__ dx, float* __restrict__ da_full,
                                                             float* __restrict__ da_plane, double* __restrict__ block_x2g,
                                                             int64_t planes, int hw, int a_per_plane, int accumulate_dx,
                                                             unsigned* __restrict__ da_absmax) {
    __shared__ double wsum[4], px2g[4];
    __shared__ float pxg[4];
    __shared__ float amax_scratch[16];
    float am = 0.0f;                    // largest |da| (full-map form): the branch's last conv consumes da in its fp16-piece form
    double wacc = 0.0;
    const float gm = gamma[0];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int sub = WPP == 1 ? lane : (int)threadIdx.x;              // position inside the plane's worker group
    const int nsub = WPP == 1 ? 64 : 256;
    const int64_t first = WPP == 1 ? (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) : (int64_t)blockIdx.x;
    const int64_t step = WPP == 1 ? (((int64_t)gridDim.x * blockDim.x) >> 6) : (int64_t)gridDim.x;
    const bool vec = (hw & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dx) |
                                        (a_per_plane ? 0 : (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(da_full)))) & 15) == 0;
    for (int64_t p = first; p < planes; p += step) {
        const int64_t base = p * hw;
        const float ap = a_per_plane ? a[p] : 0.0f;
        float sxg = 0.0f, sx2g = 0.0f;
        if (vec) {
            for (int i = sub; i < (hw >> 2); i += nsub) {
                const float4 xv = reinterpret_cast<const float4*>(x + base)[i], gv = reinterpret_cast<const float4*>(g + base)[i];
                float4 av = make_float4(ap, ap, ap, ap);
                if (!a_per_plane) av = reinterpret_cast<const float4*>(a + base)[i];
                const float4 xg = make_float4(xv.x * gv.x, xv.y * gv.y, xv.z * gv.z, xv.w * gv.w);
                float4 o = make_float4(fmaf(gm, av.x, 1.0f) * gv.x, fmaf(gm, av.y, 1.0f) * gv.y, fmaf(gm, av.z, 1.0f) * gv.z,
                                       fmaf(gm, av.w, 1.0f) * gv.w);
                if (accumulate_dx) {
                    const float4 old = reinterpret_cast<const float4*>(dx + base)[i];
                    o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                }
                reinterpret_cast<float4*>(dx + base)[i] = o;
                if (!a_per_plane) {
                    const float4 dav = make_float4(xg.x * gm, xg.y * gm, xg.z * gm, xg.w * gm);
                    reinterpret_cast<float4*>(da_full + base)[i] = dav;
                    am = fmaxf(fmaxf(am, fmaxf(fabsf(dav.x), fabsf(dav.y))), fmaxf(fabsf(dav.z), fabsf(dav.w)));
                }
                sxg += (xg.x + xg.y) + (xg.z + xg.w);
                sx2g += (xg.x * xv.x + xg.y * xv.y) + (xg.z * xv.z + xg.w * xv.w);
            }
        } else {
            for (int i = sub; i < hw; i += nsub) {
                const float xv = x[base + i], gv = g[base + i];
                const float av = a_per_plane ? ap : a[base + i];
                const float xg = xv * gv;
                const float o = fmaf(gm, av, 1.0f) * gv;
                dx[base + i] = accumulate_dx ? dx[base + i] + o : o;
                if (!a_per_plane) { da_full[base + i] = xg * gm; am = fmaxf(am, fabsf(xg * gm)); }
                sxg += xg;
                sx2g = fmaf(xg, xv, sx2g);
            }
        }
        sxg = wave_sum(sxg);
        const double s2 = wave_sum_d((double)sx2g);      // dgamma sums x^2 g with heavy cancellation: leave fp32 early
        if (WPP == 1) {
            wacc += s2;
            if (lane == 0 && a_per_plane) da_plane[p] = gm * sxg;
        } else {
            __syncthreads();
            if (lane == 0) { pxg[wid] = sxg; px2g[wid] = s2; }
            __syncthreads();
            if (threadIdx.x == 0) {
                wacc += (px2g[0] + px2g[1]) + (px2g[2] + px2g[3]);
                if (a_per_plane) da_plane[p] = gm * ((pxg[0] + pxg[1]) + (pxg[2] + pxg[3]));
            }
        }
    }
    if (WPP == 1) {
        if (lane == 0) wsum[wid] = wacc;
        __syncthreads();
        if (threadIdx.x == 0) block_x2g[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    } else if (threadIdx.x == 0) {
        block_x2g[blockIdx.x] = wacc;
    }
    if (da_absmax && !a_per_plane) absmax_publish(am, amax_scratch, da_absmax);
}

// The same for SMALL planes (hw a multiple of 4, at most 128 elements: the 2x2 ... 8x8 maps of the deep blocks, where B x C is
// tens of thousands of planes): LP lanes per plane, 64 / LP planes per wave and pass - one 16-byte access per lane.  With a whole
// wave per 4-element plane one lane worked, every plane paid two full wave reductions (the double one: 12 shuffles of 8 bytes) and
// a wave walked ~50 planes one after the other: 25 - 70 us for tensors of a few hundred KB.  Per-plane sums: the same additions
// in the same order as gate_bwd_plane_kernel<1> (whose idle lanes add exact zeros), hence the same bits.
template <int LP>
__global__ void __launch_bounds__(256) gate_bwd_small_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                             const float* __restrict__ gamma, const float* __restrict__ g,
                                                             float* __restrict__ dx, float* __restrict__ da_full,
                                                             float* __restrict__ da_plane, double* __restrict__ block_x2g,
                                                             int64_t planes, int hw, int a_per_plane, int accumulate_dx,
                                                             unsigned* __restrict__ da_absmax) {
    __shared__ double wsum[4];
    __shared__ float amax_scratch[16];
    constexpr int PW = 64 / LP;
    float am = 0.0f;
    double wacc = 0.0;
    const float gm = gamma[0];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int gid = lane / LP, li = lane % LP;
    const int q4 = hw >> 2;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t p0 = wave * PW; p0 < planes; p0 += nwaves * PW) {
        const int64_t p = p0 + gid;
        const bool live = p < planes;
        float sxg = 0.0f, sx2g = 0.0f;
        if (live && li < q4) {
            const int64_t base = p * hw;
            const float ap = a_per_plane ? a[p] : 0.0f;
            const float4 xv = reinterpret_cast<const float4*>(x + base)[li], gv = reinterpret_cast<const float4*>(g + base)[li];
            float4 av = make_float4(ap, ap, ap, ap);
            if (!a_per_plane) av = reinterpret_cast<const float4*>(a + base)[li];
            const float4 xg = make_float4(xv.x * gv.x, xv.y * gv.y, xv.z * gv.z, xv.w * gv.w);
            float4 o = make_float4(fmaf(gm, av.x, 1.0f) * gv.x, fmaf(gm, av.y, 1.0f) * gv.y, fmaf(gm, av.z, 1.0f) * gv.z,
                                   fmaf(gm, av.w, 1.0f) * gv.w);
            if (accumulate_dx) {
                const float4 old = reinterpret_cast<const float4*>(dx + base)[li];
                o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            reinterpret_cast<float4*>(dx + base)[li] = o;
            if (!a_per_plane) {
                const float4 dav = make_float4(xg.x * gm, xg.y * gm, xg.z * gm, xg.w * gm);
                reinterpret_cast<float4*>(da_full + base)[li] = dav;
                am = fmaxf(fmaxf(am, fmaxf(fabsf(dav.x), fabsf(dav.y))), fmaxf(fabsf(dav.z), fabsf(dav.w)));
            }
            sxg = (xg.x + xg.y) + (xg.z + xg.w);
            sx2g = (xg.x * xv.x + xg.y * xv.y) + (xg.z * xv.z + xg.w * xv.w);
        }
        double s2 = (double)sx2g;
#pragma unroll
        for (int o = LP / 2; o > 0; o >>= 1) {
            sxg += __shfl_xor(sxg, o, 64);
            s2 += __shfl_xor(s2, o, 64);
        }
        if (live && li == 0) {
            wacc += s2;
            if (a_per_plane) da_plane[p] = gm * sxg;
        }
    }
    wacc = wave_sum_d(wacc);          // the plane leaders' sums (other lanes hold 0), fixed order
    if (lane == 0) wsum[wid] = wacc;
    __syncthreads();
    if (threadIdx.x == 0) block_x2g[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    if (da_absmax && !a_per_plane) absmax_publish(am, amax_scratch, da_absmax);
}

// dgamma = sum over blocks of block_x2g (reference bug: x^2 g, merge.py:33-38)
__global__ void __launch_bounds__(256) gate_bwd_final_kernel(const double* __restrict__ block_x2g, float* __restrict__ dgamma,
                                                             int nblocks) {
    __shared__ double scratch[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) acc += block_x2g[i];
    acc = block_sum<double>(acc, scratch);
    if (threadIdx.x == 0) dgamma[0] = (float)acc;
}

LOCATE_API int locate_gate_fwd(const float* x, const float* a, int a_per_plane, const float* gamma, float* out,
                               int64_t planes, int hw, void* stream) {
    LOCATE_REQUIRE(planes >= 0 && hw > 0 && planes * hw < (1ll << 31), "locate_gate_fwd: bad shape");
    const int64_t n = planes * hw;
    if (n == 0) return LOCATE_OK;
    gate_fwd_kernel<<<stream_grid(n, 1024), 256, 0, as_stream(stream)>>>(x, a, gamma, out, n, hw, a_per_plane);
    LOCATE_LAUNCH_CHECK("locate_gate_fwd");
    return LOCATE_OK;
}

// out = (gamma a + 1) x AND the InPlaceNorm statistics partials of `out` for `groups` stacked calls:
// stats_partial = locate_norm_stats_workspace_bytes() bytes, to be handed to locate_norm_fwd as `pre_partial`.
LOCATE_API int locate_gate_fwd_stats(const float* x, const float* a, int a_per_plane, const float* gamma, float* out,
                                     int64_t planes, int hw, int groups, double* stats_partial, void* stream) {
    LOCATE_REQUIRE(planes > 0 && hw > 0 && planes * hw < (1ll << 31) && stats_partial, "locate_gate_fwd_stats: bad arguments");
    LOCATE_REQUIRE(groups >= 1 && groups <= 4 && planes % groups == 0, "locate_gate_fwd_stats: bad group count");
    const int64_t n_g = planes / groups * hw;
    LOCATE_REQUIRE(n_g > 1 && (groups == 1 || (n_g & 3) == 0), "locate_gate_fwd_stats: group size must be a multiple of 4");
    int np = stream_grid(n_g, 256 * 16);          // the SAME partial count locate_norm_fwd derives from the group size
    if (np > 512) np = 512;
    gate_fwd_stats_kernel<<<dim3(np, groups), 256, 0, as_stream(stream)>>>(x, a, gamma, out, n_g, hw, a_per_plane, stats_partial);
    LOCATE_LAUNCH_CHECK("locate_gate_fwd_stats");
    return LOCATE_OK;
}

LOCATE_API size_t locate_gate_bwd_workspace_bytes(int64_t planes) { (void)planes; return 4096 * sizeof(double); }

static int64_t gate_bwd_blocks(int64_t planes, int hw) {
    int64_t blocks = hw >= 1024 ? planes : cdiv64(planes, 4);
    return blocks > 4096 ? 4096 : blocks;
}
// how many doubles locate_gate_bwd leaves at the start of its workspace: their sum (in index order) is dgamma
LOCATE_API int locate_gate_bwd_partials(int64_t planes, int hw) { return (int)gate_bwd_blocks(planes, hw); }

// da: [planes*hw] when a_per_plane == 0, [planes] otherwise.  dgamma: one float (overwritten).
// (Tried and dropped: the kernel's last-arriving block reducing dgamma itself instead of the second launch - up to 4096
// blocks taking a ticket from ONE counter serialise at ~90 arrivals per microsecond: +0.7 ms per training step.)
LOCATE_API int locate_gate_bwd(const float* x, const float* a, int a_per_plane, const float* gamma, const float* g,
                               float* dx, float* da, float* dgamma, int64_t planes, int hw, void* workspace,
                               int accumulate_dx, void* da_absmax, void* stream) {
    LOCATE_REQUIRE(planes > 0 && hw > 0 && workspace, "locate_gate_bwd: bad shape or missing workspace");
    double* block_x2g = static_cast<double*>(workspace);
    const int64_t blocks = gate_bwd_blocks(planes, hw);          // (what the caller sized its partial sums for)
    if (hw == 1 && (planes & 63) == 0) {
        // a 1x1 map (either form of the gate value: a, da are [planes] both ways) IS the element-wise form over "planes" of 64
        // values each: the several-planes-per-wave kernel instead of one wave per single element (25 us for the 98 304 values of
        // the deepest block)
        if (a_per_plane) { a_per_plane = 0; da_absmax = nullptr; }
        planes >>= 6; hw = 64;
    }
    const bool whole_block = hw >= 1024;
    const bool vec = (hw & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dx) |
                                        (a_per_plane ? 0 : (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(da)))) & 15) == 0;
#define GATE_SMALL(LPV) gate_bwd_small_kernel<LPV><<<(int)blocks, 256, 0, as_stream(stream)>>>(x, a, gamma, g, dx, a_per_plane ? nullptr : da, \
        a_per_plane ? da : nullptr, block_x2g, planes, hw, a_per_plane, accumulate_dx, static_cast<unsigned*>(da_absmax))
    if (vec && hw <= 128) {
        const int q4 = hw >> 2;
        if (q4 <= 1) GATE_SMALL(1);
        else if (q4 <= 2) GATE_SMALL(2);
        else if (q4 <= 4) GATE_SMALL(4);
        else if (q4 <= 8) GATE_SMALL(8);
        else if (q4 <= 16) GATE_SMALL(16);
        else GATE_SMALL(32);
    } else if (whole_block)
        gate_bwd_plane_kernel<4><<<(int)blocks, 256, 0, as_stream(stream)>>>(x, a, gamma, g, dx, a_per_plane ? nullptr : da,
                                                                            a_per_plane ? da : nullptr, block_x2g, planes, hw,
                                                                            a_per_plane, accumulate_dx, static_cast<unsigned*>(da_absmax));
    else
        gate_bwd_plane_kernel<1><<<(int)blocks, 256, 0, as_stream(stream)>>>(x, a, gamma, g, dx, a_per_plane ? nullptr : da,
                                                                            a_per_plane ? da : nullptr, block_x2g, planes, hw,
                                                                            a_per_plane, accumulate_dx, static_cast<unsigned*>(da_absmax));
    LOCATE_LAUNCH_CHECK("locate_gate_bwd(plane)");
    if (!dgamma) return LOCATE_OK;       // the caller sums the locate_gate_bwd_partials() doubles in `workspace` itself, or needs none
    gate_bwd_final_kernel<<<1, 256, 0, as_stream(stream)>>>(block_x2g, dgamma, (int)blocks);
    LOCATE_LAUNCH_CHECK("locate_gate_bwd(final)");
    return LOCATE_OK;
}

// ---------------------------------------------------------------------------------------------
// Channel-slice copy: dst[b, c, :] = src[b, c, :] for c < C with independent batch strides (elements).
// Used for torch.cat([x, layer(x)], dim=1) (merge.py:15) and for slicing its gradient.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) copy_planes_kernel(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                          int64_t chw, int64_t src_bs, int64_t dst_bs, int accumulate) {
    // one batch element per blockIdx.y: no division, 16-byte accesses when everything is aligned
    const int b = blockIdx.y;
    const float* s = src + (int64_t)b * src_bs;
    float* d = dst + (int64_t)b * dst_bs;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec = ((chw | src_bs | dst_bs) & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    if (vec) {
        const float4* s4 = reinterpret_cast<const float4*>(s);
        float4* d4 = reinterpret_cast<float4*>(d);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (chw >> 2); i += stride) {
            float4 v = s4[i];
            if (accumulate) { const float4 o = d4[i]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            d4[i] = v;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < chw; i += stride) d[i] = accumulate ? d[i] + s[i] : s[i];
}

LOCATE_API int locate_copy_channels(const float* src, float* dst, int B, int C, int hw, int64_t src_batch_stride,
                                    int64_t dst_batch_stride, int accumulate, void* stream) {
    LOCATE_REQUIRE(B >= 0 && C >= 0 && hw >= 0, "locate_copy_channels: bad shape");
    const int64_t chw = (int64_t)C * hw;
    if ((int64_t)B * chw == 0) return LOCATE_OK;
    LOCATE_REQUIRE(src_batch_stride >= chw && dst_batch_stride >= chw, "locate_copy_channels: batch stride smaller than slice");
    int gx = stream_grid(chw, 1024);
    if (gx > 64) gx = 64;
    copy_planes_kernel<<<dim3(gx, B), 256, 0, as_stream(stream)>>>(src, dst, B, chw, src_batch_stride, dst_batch_stride,
                                                                   accumulate);
    LOCATE_LAUNCH_CHECK("locate_copy_channels");
    return LOCATE_OK;
}
