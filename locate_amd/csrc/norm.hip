// InPlaceNorm (reference libs/inplace_norm.py:4-45): GLOBAL scalar mean and unbiased std over the whole
// [B, C, H, W] tensor (no per-channel statistics, no eps, no running stats), then a per-channel weight
// [1,C,1,1] or a per-sample style scale [B,C,1,1], plus a per-channel bias:
//     out = (x - mu) * y / s + b
// Backward (closed form of the reference's MeanSubMulDivAdd.backward + ATen std backward):
//     dx = y g / s - mean(y g / s) + dz (x - mu) / ((N-1) s),   dz = -sum((x-mu) g y) / s^2
//     dy = sum_bcast (x - mu) g / s,    db = sum_bcast g
// GROUPS: the batch may hold `groups` independent forwards stacked along B (the discriminator's real / fake /
// augmented passes of one D-step run as ONE pass of batch 3B); statistics, and everything derived from them, are
// then per group (what three separate reference forwards compute), parameter gradients sum over all groups.
// All HBM-bound.  Statistics are accumulated in fp64 in a fixed order (no atomics: bit-reproducible).
#include "common.h"
#include "finrec.h"

#define NORM_MAX_PARTIALS 512
#define NORM_MAX_GROUPS 4

// ---------------------------------------------------------------------------------------------
// statistics: partial (sum, sum of squares) per block in double; blockIdx.y = group
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) stats_partial_kernel(const float* __restrict__ x, int64_t n_group,
                                                            double* __restrict__ partial) {
    __shared__ double scratch[16];
    x += (int64_t)blockIdx.y * n_group;
    partial += (int64_t)blockIdx.y * 2 * gridDim.x;
    double s = 0.0, q = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n_group >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = x4[i];
        const double a = v.x, b = v.y, c = v.z, d = v.w;
        s += (a + b) + (c + d);
        q += (a * a + b * b) + (c * c + d * d);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_group; i += stride) {
        const double a = x[i];
        s += a;
        q += a * a;
    }
    s = block_sum<double>(s, scratch);
    q = block_sum<double>(q, scratch);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = s;
        partial[2 * blockIdx.x + 1] = q;
    }
}

// mean / unbiased std from the per-block partial sums; every thread of the calling block gets the result
// (fixed summation order: bit-identical in every block and on every replay)
__device__ __forceinline__ void stats_from_partials(const double* __restrict__ partial, int nblocks, int64_t n, double* scratch,
                                                    float& mean_out, float& std_out) {
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        s += partial[2 * i];
        q += partial[2 * i + 1];
    }
    s = block_sum<double>(s, scratch);
    q = block_sum<double>(q, scratch);
    const double mean = s / (double)n;
    double var = (q - s * mean) / (double)(n - 1);
    if (var < 0.0) var = 0.0;
    mean_out = (float)mean;
    std_out = (float)sqrt(var);
}

__global__ void __launch_bounds__(256) stats_final_kernel(const double* __restrict__ partial, int nblocks, int64_t n,
                                                          float* __restrict__ stats) {
    __shared__ double scratch[16];
    float mu, sd;
    stats_from_partials(partial, nblocks, n, scratch, mu, sd);
    if (threadIdx.x == 0) {
        stats[0] = mu;
        stats[1] = sd;
    }
}

LOCATE_API size_t locate_norm_stats_workspace_bytes(void) { return (size_t)NORM_MAX_GROUPS * NORM_MAX_PARTIALS * 2 * sizeof(double); }

// stats = {mean, unbiased std} of x[0..n)
LOCATE_API int locate_norm_stats(const float* x, int64_t n, float* stats, void* workspace, void* stream) {
    LOCATE_REQUIRE(n > 1 && workspace && stats, "locate_norm_stats: needs >= 2 elements and both buffers");
    LOCATE_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "locate_norm_stats: x must be 16-byte aligned");
    int np = stream_grid(n, 256 * 16);
    if (np > NORM_MAX_PARTIALS) np = NORM_MAX_PARTIALS;
    double* partial = static_cast<double*>(workspace);
    stats_partial_kernel<<<np, 256, 0, as_stream(stream)>>>(x, n, partial);
    LOCATE_LAUNCH_CHECK("locate_norm_stats(partial)");
    stats_final_kernel<<<1, 256, 0, as_stream(stream)>>>(partial, np, n, stats);
    LOCATE_LAUNCH_CHECK("locate_norm_stats(final)");
    return LOCATE_OK;
}

// ---------------------------------------------------------------------------------------------
// forward: statistics partials (one launch) + apply (one launch; every block reduces its group's <= 512 partial
// pairs itself instead of waiting for a third, single-block launch).  blockIdx.y = group; the first block of a group
// publishes stats[group] = {mean, std} for backward.  ACT: store RootTanh(out) instead of out (the conv stage that
// follows a block-input norm starts with RootTanh, conv.py:22-24).
// ---------------------------------------------------------------------------------------------
template <bool ACT>
__global__ void __launch_bounds__(256) norm_apply_fused_kernel(const float* __restrict__ x, const double* __restrict__ partial,
                                                               int npartial, float* __restrict__ stats_out,
                                                               const float* __restrict__ scale, int scale_per_sample,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               int64_t planes_g, int C, int hw, unsigned* __restrict__ absmax) {
    __shared__ double scratch[16];
    __shared__ float amax_scratch[16];
    float am = 0.0f;
    const int grp = blockIdx.y;
    const int64_t n = planes_g * hw;                 // elements of one group
    float mu, s;
    stats_from_partials(partial + (int64_t)grp * 2 * npartial, npartial, n, scratch, mu, s);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stats_out[2 * grp] = mu;
        stats_out[2 * grp + 1] = s;
    }
    x += (int64_t)grp * n;
    out += (int64_t)grp * n;
    const float inv_s = 1.0f / s;
    const int64_t plane0 = (int64_t)grp * planes_g;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if ((hw & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        float4* o4 = reinterpret_cast<float4*>(out);
        const unsigned hw4 = (unsigned)(hw >> 2);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n >> 2); i += stride) {
            const int64_t p = plane0 + (unsigned)i / hw4;        // n / 4 < 2^32 (checked by the host entry): 32-bit division
            const int c = (int)((unsigned)p % (unsigned)C);
            const float y = scale[scale_per_sample ? p : c], b = bias[c];
            const float4 v = x4[i];
            float4 o;
            const float ys = y * inv_s;
            o.x = fmaf(v.x - mu, ys, b); o.y = fmaf(v.y - mu, ys, b);
            o.z = fmaf(v.z - mu, ys, b); o.w = fmaf(v.w - mu, ys, b);
            if (ACT) { o.x = roottanh_f(o.x); o.y = roottanh_f(o.y); o.z = roottanh_f(o.z); o.w = roottanh_f(o.w); }
            o4[i] = o;
            am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            const int64_t p = plane0 + i / hw;
            const int c = (int)(p % C);
            const float y = scale[scale_per_sample ? p : c];
            float o = fmaf(x[i] - mu, y * inv_s, bias[c]);
            o = ACT ? roottanh_f(o) : o;
            out[i] = o;
            am = fmaxf(am, fabsf(o));
        }
    }
    if (absmax) absmax_publish(am, amax_scratch, absmax);      // largest |out| over all groups, for the fp16-piece contractions
}

// out = (x - mean(x)) * scale / std(x) + bias  (with_act = 0), or RootTanh of it (with_act = 1; the plain value is not
// stored - backward recomputes it).  B = groups * (B / groups); statistics per group; stats_out: [groups][2].
// scale: [C] (scale_per_sample = 0) or [B*C].  Two launches.
// pre_partial (nullable): the statistics partials of x already produced by the kernel that wrote x (locate_gate_fwd_stats,
// same group count) - the statistics pass is then skipped.
LOCATE_API int locate_norm_fwd(const float* x, const float* scale, int scale_per_sample, const float* bias, float* out,
                               int with_act, float* stats_out, int B, int C, int hw, int groups, void* workspace,
                               const double* pre_partial, void* absmax, void* stream) {
    LOCATE_REQUIRE(B > 0 && C > 0 && hw > 0 && (workspace || pre_partial) && stats_out && out, "locate_norm_fwd: bad arguments");
    LOCATE_REQUIRE(groups >= 1 && groups <= NORM_MAX_GROUPS && B % groups == 0, "locate_norm_fwd: bad group count %d for batch %d", groups, B);
    LOCATE_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "locate_norm_fwd: x must be 16-byte aligned");
    const int64_t planes_g = (int64_t)(B / groups) * C, n_g = planes_g * hw;
    LOCATE_REQUIRE(n_g > 1 && n_g < (1ll << 33), "locate_norm_fwd: needs 2 .. 2^33 elements per group");
    LOCATE_REQUIRE(groups == 1 || (n_g & 3) == 0, "locate_norm_fwd: grouped tensors need a group size that is a multiple of 4");
    int np = stream_grid(n_g, 256 * 16);
    if (np > NORM_MAX_PARTIALS) np = NORM_MAX_PARTIALS;
    const double* partial = pre_partial;
    hipStream_t st = as_stream(stream);
    if (partial == nullptr) {
        stats_partial_kernel<<<dim3(np, groups), 256, 0, st>>>(x, n_g, static_cast<double*>(workspace));
        LOCATE_LAUNCH_CHECK("locate_norm_fwd(stats)");
        partial = static_cast<const double*>(workspace);
    }
    const dim3 grid(stream_grid(n_g, 1024), groups);
    if (with_act)
        norm_apply_fused_kernel<true><<<grid, 256, 0, st>>>(x, partial, np, stats_out, scale, scale_per_sample, bias, out, planes_g, C, hw,
                                                            static_cast<unsigned*>(absmax));
    else
        norm_apply_fused_kernel<false><<<grid, 256, 0, st>>>(x, partial, np, stats_out, scale, scale_per_sample, bias, out, planes_g, C, hw,
                                                             static_cast<unsigned*>(absmax));
    LOCATE_LAUNCH_CHECK("locate_norm_fwd(apply)");
    return LOCATE_OK;
}

// ---------------------------------------------------------------------------------------------
// backward
//   pass 1 (one wave per plane):   S1[p] = sum go,  S2[p] = sum (x - mu_g) go      (go = g, or g * RootTanh'(out) with ACT)
//   pass 2 (one block):            dbias[c], dscale, consts[g] = { mean(y go / s_g), dz_g / ((N_g - 1) s_g) }
//   pass 3 (element-wise):         dx = y[p] go / s_g - consts[g][0] + consts[g][1] (x - mu_g)
// ---------------------------------------------------------------------------------------------
template <bool ACT>
__device__ __forceinline__ float norm_go(float xv, float gv, float mu, float inv_s, float y, float b) {
    if (!ACT) return gv;
    const float o = fmaf(xv - mu, y * inv_s, b);          // the same expression as the forward's norm_apply_fused_kernel
    return roottanh_grad_f(o, gv);
}

template <bool ACT>
__global__ void __launch_bounds__(256) norm_bwd_plane_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                             const float* __restrict__ stats, const float* __restrict__ scale,
                                                             int scale_per_sample, const float* __restrict__ bias, int C,
                                                             float* __restrict__ S1, float* __restrict__ S2, int64_t planes,
                                                             int64_t planes_g, int hw, double* __restrict__ block_partial,
                                                             float* __restrict__ dscale_sample) {
    // block_partial (two-launch form, locate_norm_bwd_fused): this block's share of sum_p y[p] S1[p] and sum_p y[p] S2[p] per
    // group, [block][NORM_MAX_GROUPS][2] doubles - the dx kernel adds the blocks up itself, the middle launch is gone;
    // dscale_sample: the per-sample scale gradient S2[p] / std straight from here
    __shared__ double bp[4][NORM_MAX_GROUPS][2];
    double acc_a[NORM_MAX_GROUPS], acc_b[NORM_MAX_GROUPS];
#pragma unroll
    for (int q = 0; q < NORM_MAX_GROUPS; ++q) acc_a[q] = acc_b[q] = 0.0;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t p = wave; p < planes; p += nwaves) {
        const int grp = (int)(p / planes_g);
        const float mu = stats[2 * grp], inv_sd = 1.0f / stats[2 * grp + 1];
        const int64_t base = p * hw;
        const int c = (int)(p % C);
        const float y = ACT ? scale[scale_per_sample ? p : c] : 0.0f, b = ACT ? bias[c] : 0.0f;
        float s1 = 0.0f, s2 = 0.0f;
        if ((hw & 3) == 0) {
            const float4* x4 = reinterpret_cast<const float4*>(x + base);
            const float4* g4 = reinterpret_cast<const float4*>(g + base);
            for (int i = lane; i < (hw >> 2); i += 64) {
                const float4 xv = x4[i];
                float4 gv = g4[i];
                gv.x = norm_go<ACT>(xv.x, gv.x, mu, inv_sd, y, b); gv.y = norm_go<ACT>(xv.y, gv.y, mu, inv_sd, y, b);
                gv.z = norm_go<ACT>(xv.z, gv.z, mu, inv_sd, y, b); gv.w = norm_go<ACT>(xv.w, gv.w, mu, inv_sd, y, b);
                s1 += (gv.x + gv.y) + (gv.z + gv.w);
                s2 += ((xv.x - mu) * gv.x + (xv.y - mu) * gv.y) + ((xv.z - mu) * gv.z + (xv.w - mu) * gv.w);
            }
        } else {
            for (int i = lane; i < hw; i += 64) {
                const float xv = x[base + i];
                const float gv = norm_go<ACT>(xv, g[base + i], mu, inv_sd, y, b);
                s1 += gv;
                s2 = fmaf(xv - mu, gv, s2);
            }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {
            S1[p] = s1;
            S2[p] = s2;
            if (block_partial) {
                const float yv = scale[scale_per_sample ? p : c];
#pragma unroll
                for (int q = 0; q < NORM_MAX_GROUPS; ++q)
                    if (q == grp) {
                        acc_a[q] += (double)yv * (double)s1;
                        acc_b[q] += (double)yv * (double)s2;
                    }
                if (dscale_sample) dscale_sample[p] = s2 / stats[2 * grp + 1];
            }
        }
    }
    if (block_partial) {          // the four waves' lane-0 sums in wave order
        const int wid = threadIdx.x >> 6;
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < NORM_MAX_GROUPS; ++q) { bp[wid][q][0] = acc_a[q]; bp[wid][q][1] = acc_b[q]; }
        }
        __syncthreads();
        if (threadIdx.x < NORM_MAX_GROUPS * 2) {
            const int q = threadIdx.x >> 1, e = threadIdx.x & 1;
            block_partial[((int64_t)blockIdx.x * NORM_MAX_GROUPS + q) * 2 + e] = ((bp[0][q][e] + bp[1][q][e]) + bp[2][q][e]) + bp[3][q][e];
        }
    }
}

// Pass 1 for SMALL planes (hw a multiple of 4, at most 128 elements - see gate_bwd_small_kernel, elementwise.hip): LP lanes per
// plane, 64 / LP planes per wave and pass.  S1 / S2: the same additions in the same order as the wave-per-plane kernel.
template <bool ACT, int LP>
__global__ void __launch_bounds__(256) norm_bwd_plane_small_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                   const float* __restrict__ stats, const float* __restrict__ scale,
                                                                   int scale_per_sample, const float* __restrict__ bias, int C,
                                                                   float* __restrict__ S1, float* __restrict__ S2, int64_t planes,
                                                                   int64_t planes_g, int hw, double* __restrict__ block_partial,
                                                                   float* __restrict__ dscale_sample) {
    __shared__ double bp[4][NORM_MAX_GROUPS][2];
    // LP == 0: planes of ONE or TWO elements (1x1 maps) - a lane per plane, scalar accesses
    constexpr int LPE = LP == 0 ? 1 : LP;
    constexpr int PW = 64 / LPE;
    double acc_a[NORM_MAX_GROUPS], acc_b[NORM_MAX_GROUPS];
#pragma unroll
    for (int q = 0; q < NORM_MAX_GROUPS; ++q) acc_a[q] = acc_b[q] = 0.0;
    const int lane = threadIdx.x & 63;
    const int gid = lane / LPE, li = lane % LPE;
    const int q4 = hw >> 2;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t p0 = wave * PW; p0 < planes; p0 += nwaves * PW) {
        const int64_t p = p0 + gid;
        const bool live = p < planes;
        const int64_t pp = live ? p : planes - 1;
        const int grp = (int)(pp / planes_g);
        const float mu = stats[2 * grp], sd = stats[2 * grp + 1], inv_sd = 1.0f / sd;
        const int c = (int)(pp % C);
        const float yv = (ACT || block_partial != nullptr) ? scale[scale_per_sample ? pp : c] : 0.0f;
        const float y = ACT ? yv : 0.0f, b = ACT ? bias[c] : 0.0f;
        float s1 = 0.0f, s2 = 0.0f;
        if constexpr (LP == 0) {
            if (live) {
                for (int i = 0; i < hw; ++i) {          // (hw <= 2: e0 + e1, the wave-per-plane kernel's own order)
                    const float xv = x[pp * hw + i];
                    const float gv = norm_go<ACT>(xv, g[pp * hw + i], mu, inv_sd, y, b);
                    s1 += gv;
                    s2 = i == 0 ? (xv - mu) * gv : fmaf(xv - mu, gv, s2);
                }
            }
        } else if (live && li < q4) {
            const float4 xv = reinterpret_cast<const float4*>(x + pp * hw)[li];
            float4 gv = reinterpret_cast<const float4*>(g + pp * hw)[li];
            gv.x = norm_go<ACT>(xv.x, gv.x, mu, inv_sd, y, b); gv.y = norm_go<ACT>(xv.y, gv.y, mu, inv_sd, y, b);
            gv.z = norm_go<ACT>(xv.z, gv.z, mu, inv_sd, y, b); gv.w = norm_go<ACT>(xv.w, gv.w, mu, inv_sd, y, b);
            s1 = (gv.x + gv.y) + (gv.z + gv.w);
            s2 = ((xv.x - mu) * gv.x + (xv.y - mu) * gv.y) + ((xv.z - mu) * gv.z + (xv.w - mu) * gv.w);
        }
#pragma unroll
        for (int o = LPE / 2; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
        }
        if (live && li == 0) {
            S1[p] = s1;
            S2[p] = s2;
            if (block_partial) {
#pragma unroll
                for (int q = 0; q < NORM_MAX_GROUPS; ++q)
                    if (q == grp) {
                        acc_a[q] += (double)yv * (double)s1;
                        acc_b[q] += (double)yv * (double)s2;
                    }
                if (dscale_sample) dscale_sample[p] = s2 / sd;
            }
        }
    }
    if (block_partial) {          // the plane leaders' sums over the wave (fixed order), then the four waves in wave order
        const int wid = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < NORM_MAX_GROUPS; ++q) {
            const double a = wave_sum_d(acc_a[q]), bsum = wave_sum_d(acc_b[q]);
            if (lane == 0) { bp[wid][q][0] = a; bp[wid][q][1] = bsum; }
        }
        __syncthreads();
        if (threadIdx.x < NORM_MAX_GROUPS * 2) {
            const int q = threadIdx.x >> 1, e = threadIdx.x & 1;
            block_partial[((int64_t)blockIdx.x * NORM_MAX_GROUPS + q) * 2 + e] = ((bp[0][q][e] + bp[1][q][e]) + bp[2][q][e]) + bp[3][q][e];
        }
    }
}

template <bool ACT>
static bool launch_norm_plane_small(int blocks, hipStream_t st, const float* x, const float* g, const float* stats, const float* scale,
                                    int scale_per_sample, const float* bias, int C, float* S1, float* S2, int64_t planes,
                                    int64_t planes_g, int hw, double* block_partial, float* dscale_sample) {
    if (hw == 1) {
        norm_bwd_plane_small_kernel<ACT, 0><<<blocks, 256, 0, st>>>(x, g, stats, scale, scale_per_sample, bias, C, S1, S2, planes, planes_g, hw,
                                                                   block_partial, dscale_sample);
        return true;
    }
    if ((hw & 3) != 0 || hw > 128 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g)) & 15) != 0) return false;
    const int q4 = hw >> 2;
#define NORM_SMALL(LPV) norm_bwd_plane_small_kernel<ACT, LPV><<<blocks, 256, 0, st>>>(x, g, stats, scale, scale_per_sample, bias, C, S1, S2, \
        planes, planes_g, hw, block_partial, dscale_sample)
    if (q4 <= 1) NORM_SMALL(1);
    else if (q4 <= 2) NORM_SMALL(2);
    else if (q4 <= 4) NORM_SMALL(4);
    else if (q4 <= 8) NORM_SMALL(8);
    else if (q4 <= 16) NORM_SMALL(16);
    else NORM_SMALL(32);
    return true;
}

// Deterministic (fixed summation order - no atomics - so that eager and hipGraph replays agree bit for bit).
// Block = 16 channels x 16 batch slots; the slots meet through LDS in a fixed order.  Per block and group the partial
// sums of y*S1 and y*S2 go to `partial` [blocks][NORM_MAX_GROUPS][2] (double); the dx kernel adds them up.
constexpr int NF_CH = 16, NF_SLOTS = 16;
__global__ void __launch_bounds__(256) norm_bwd_final_kernel(const float* __restrict__ S1, const float* __restrict__ S2,
                                                             const float* __restrict__ stats, const float* __restrict__ scale,
                                                             int scale_per_sample, float* __restrict__ dscale,
                                                             float* __restrict__ dbias, double* __restrict__ partial, int B,
                                                             int C, int groups) {
    __shared__ double scratch[16];
    __shared__ float part_b[256], part_y[256];
    const int Bg = B / groups;
    const int cl = threadIdx.x & (NF_CH - 1), slot = threadIdx.x / NF_CH;
    const int c = blockIdx.x * NF_CH + cl;
    double sum_yg[NORM_MAX_GROUPS], sum_yxg[NORM_MAX_GROUPS];
#pragma unroll
    for (int g = 0; g < NORM_MAX_GROUPS; ++g) sum_yg[g] = sum_yxg[g] = 0.0;
    float db = 0.0f, dy = 0.0f;
    if (c < C) {
        for (int b = slot; b < B; b += NF_SLOTS) {
            const int grp = b / Bg;
            const float sf = stats[2 * grp + 1];
            const int64_t p = (int64_t)b * C + c;
            const float s1 = S1[p], s2 = S2[p];
            const float y = scale[scale_per_sample ? p : c];
            db += s1;
            if (scale_per_sample) dscale[p] = s2 / sf;
            else dy += s2 / sf;
#pragma unroll
            for (int g = 0; g < NORM_MAX_GROUPS; ++g)
                if (g == grp) {
                    sum_yg[g] += (double)y * (double)s1;
                    sum_yxg[g] += (double)y * (double)s2;
                }
        }
    }
    part_b[threadIdx.x] = db;
    part_y[threadIdx.x] = dy;
    __syncthreads();
    if (slot == 0 && c < C) {
        float tb = 0.0f, ty = 0.0f;
        for (int s2 = 0; s2 < NF_SLOTS; ++s2) { tb += part_b[s2 * NF_CH + cl]; ty += part_y[s2 * NF_CH + cl]; }
        dbias[c] = tb;
        if (!scale_per_sample) dscale[c] = ty;
    }
#pragma unroll
    for (int g = 0; g < NORM_MAX_GROUPS; ++g) {
        if (g >= groups) break;
        const double a = block_sum<double>(sum_yg[g], scratch);
        const double b2 = block_sum<double>(sum_yxg[g], scratch);
        if (threadIdx.x == 0) {
            partial[((int64_t)blockIdx.x * NORM_MAX_GROUPS + g) * 2] = a;
            partial[((int64_t)blockIdx.x * NORM_MAX_GROUPS + g) * 2 + 1] = b2;
        }
    }
}

template <bool ACT>
__global__ void __launch_bounds__(256) norm_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ stats, const float* __restrict__ scale,
                                                          int scale_per_sample, const float* __restrict__ bias,
                                                          const double* __restrict__ partial, int npartial,
                                                          float* __restrict__ dx, int64_t planes_g, int C, int hw, int accumulate) {
    __shared__ float consts[2];
    const int grp = blockIdx.y;
    const float mu = stats[2 * grp], s = stats[2 * grp + 1];
    const int64_t n = planes_g * hw;
    if (threadIdx.x < 64) {           // { mean(y go / s), dz / ((N - 1) s) } of this group from the per-block partials
        double a = 0.0, b2 = 0.0;
        for (int i = threadIdx.x; i < npartial; i += 64) {
            a += partial[((int64_t)i * NORM_MAX_GROUPS + grp) * 2];
            b2 += partial[((int64_t)i * NORM_MAX_GROUPS + grp) * 2 + 1];
        }
        a = wave_sum_d(a);
        b2 = wave_sum_d(b2);
        if (threadIdx.x == 0) {
            const double sd = (double)s, nn = (double)n;
            consts[0] = (float)(a / sd / nn);
            consts[1] = (float)(-b2 / (sd * sd) / ((nn - 1.0) * sd));
        }
    }
    __syncthreads();
    const float m = consts[0], k = consts[1];
    const float inv_s = 1.0f / s;
    x += (int64_t)grp * n; g += (int64_t)grp * n; dx += (int64_t)grp * n;
    const int64_t plane0 = (int64_t)grp * planes_g;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if ((hw & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* g4 = reinterpret_cast<const float4*>(g);
        float4* o4 = reinterpret_cast<float4*>(dx);
        const unsigned hw4 = (unsigned)(hw >> 2);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n >> 2); i += stride) {
            const int64_t p = plane0 + (unsigned)i / hw4;        // 32-bit division (n / 4 < 2^32, checked by the host entry)
            const int c = (int)((unsigned)p % (unsigned)C);
            const float y = scale[scale_per_sample ? p : c], b = ACT ? bias[c] : 0.0f;
            const float4 xv = x4[i];
            float4 gv = g4[i];
            gv.x = norm_go<ACT>(xv.x, gv.x, mu, inv_s, y, b); gv.y = norm_go<ACT>(xv.y, gv.y, mu, inv_s, y, b);
            gv.z = norm_go<ACT>(xv.z, gv.z, mu, inv_s, y, b); gv.w = norm_go<ACT>(xv.w, gv.w, mu, inv_s, y, b);
            float4 o;
            const float ys = y * inv_s;
            o.x = fmaf(ys, gv.x, fmaf(k, xv.x - mu, -m)); o.y = fmaf(ys, gv.y, fmaf(k, xv.y - mu, -m));
            o.z = fmaf(ys, gv.z, fmaf(k, xv.z - mu, -m)); o.w = fmaf(ys, gv.w, fmaf(k, xv.w - mu, -m));
            if (accumulate) { const float4 old = o4[i]; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
            o4[i] = o;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t p = plane0 + i / hw;
        const int c = (int)(p % C);
        const float y = scale[scale_per_sample ? p : c];
        const float gv = norm_go<ACT>(x[i], g[i], mu, inv_s, y, ACT ? bias[c] : 0.0f);
        const float o = fmaf(y * inv_s, gv, fmaf(k, x[i] - mu, -m));
        dx[i] = accumulate ? dx[i] + o : o;
    }
}

LOCATE_API size_t locate_norm_bwd_workspace_bytes(int B, int C) {
    return (size_t)((C + NF_CH - 1) / NF_CH) * NORM_MAX_GROUPS * 2 * sizeof(double) + (size_t)B * C * 2 * sizeof(float);
}

// dscale: [C] (scale_per_sample = 0) or [B*C];  dbias: [C].  Both overwritten.  stats: [groups][2] from the forward.
// with_act = 1: g is the gradient w.r.t. RootTanh(norm(x)) (the fused forward of locate_norm_fwd); the activation's
// derivative is applied on the fly from the recomputed norm output, so neither that output nor a separate
// RootTanh-backward pass exists.
// accumulate_dx != 0: dx += (the gradient) instead of dx = ...: the tensor x feeds a second consumer whose backward kernel
// has already written its share into the same buffer (ops.fork in the Python layer) - autograd's separate add launch and
// its extra pass over the tensor go away.
LOCATE_API int locate_norm_bwd(const float* x, const float* g, const float* stats, const float* scale,
                               int scale_per_sample, const float* bias, int with_act, float* dx, float* dscale, float* dbias,
                               int B, int C, int hw, int groups, void* workspace, int accumulate_dx, void* stream) {
    LOCATE_REQUIRE(B > 0 && C > 0 && hw > 0 && workspace, "locate_norm_bwd: bad shape or missing workspace");
    LOCATE_REQUIRE(groups >= 1 && groups <= NORM_MAX_GROUPS && B % groups == 0, "locate_norm_bwd: bad group count");
    LOCATE_REQUIRE(!with_act || bias, "locate_norm_bwd: with_act needs the bias");
    LOCATE_REQUIRE((int64_t)B * C / groups * hw < (1ll << 33), "locate_norm_bwd: more than 2^33 elements per group");
    const int64_t planes = (int64_t)B * C, planes_g = planes / groups;
    LOCATE_REQUIRE(groups == 1 || ((planes_g * hw) & 3) == 0, "locate_norm_bwd: grouped tensors need a group size that is a multiple of 4");
    const int nfb = (C + NF_CH - 1) / NF_CH;
    double* partial = static_cast<double*>(workspace);
    float* S1 = reinterpret_cast<float*>(partial + (size_t)nfb * NORM_MAX_GROUPS * 2);
    float* S2 = S1 + planes;
    int64_t blocks = cdiv64(planes, 4);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = as_stream(stream);
    const bool small = with_act ? launch_norm_plane_small<true>((int)blocks, st, x, g, stats, scale, scale_per_sample, bias, C, S1, S2, planes, planes_g, hw, nullptr, nullptr)
                                : launch_norm_plane_small<false>((int)blocks, st, x, g, stats, scale, scale_per_sample, bias, C, S1, S2, planes, planes_g, hw, nullptr, nullptr);
    if (small) {
    } else if (with_act)
        norm_bwd_plane_kernel<true><<<(int)blocks, 256, 0, st>>>(x, g, stats, scale, scale_per_sample, bias, C, S1, S2, planes, planes_g, hw, nullptr, nullptr);
    else
        norm_bwd_plane_kernel<false><<<(int)blocks, 256, 0, st>>>(x, g, stats, scale, scale_per_sample, bias, C, S1, S2, planes, planes_g, hw, nullptr, nullptr);
    LOCATE_LAUNCH_CHECK("locate_norm_bwd(plane)");
    norm_bwd_final_kernel<<<nfb, 256, 0, st>>>(S1, S2, stats, scale, scale_per_sample, dscale, dbias, partial, B, C, groups);
    LOCATE_LAUNCH_CHECK("locate_norm_bwd(final)");
    const dim3 grid(stream_grid(planes_g * hw, 1024), groups);
    if (with_act)
        norm_bwd_dx_kernel<true><<<grid, 256, 0, st>>>(x, g, stats, scale, scale_per_sample, bias, partial, nfb, dx, planes_g, C, hw, accumulate_dx);
    else
        norm_bwd_dx_kernel<false><<<grid, 256, 0, st>>>(x, g, stats, scale, scale_per_sample, bias, partial, nfb, dx, planes_g, C, hw, accumulate_dx);
    LOCATE_LAUNCH_CHECK("locate_norm_bwd(dx)");
    return LOCATE_OK;
}

// ---------------------------------------------------------------------------------------------
// Two-launch backward (round 3): the plane kernel also leaves, per block, its share of the two group sums the dx kernel
// needs (and the per-sample scale gradient), so the launch between them is gone from the pass's dependent chain.  What that
// launch also produced - dbias[c] and the per-channel dscale[c], PARAMETER gradients - is computed for all norms of a
// backward pass at its end (locate_fin_norm_channels) from the plane sums S1 / S2, which stay in the workspace.
//   workspace: [NBF_BLOCKS][NORM_MAX_GROUPS][2] doubles, then S1 [B*C], S2 [B*C] floats
// ---------------------------------------------------------------------------------------------
#define NBF_BLOCKS 512
LOCATE_API size_t locate_norm_bwd_fused_workspace_bytes(int B, int C) {
    return (size_t)NBF_BLOCKS * NORM_MAX_GROUPS * 2 * sizeof(double) + (size_t)B * C * 2 * sizeof(float);
}
// byte offset of S1 inside that workspace (S2 follows after B*C floats)
LOCATE_API size_t locate_norm_bwd_fused_plane_offset(void) { return (size_t)NBF_BLOCKS * NORM_MAX_GROUPS * 2 * sizeof(double); }

// dscale_sample: [B*C] when scale_per_sample (written), else must be null (the per-channel dscale comes from the finaliser)
LOCATE_API int locate_norm_bwd_fused(const float* x, const float* g, const float* stats, const float* scale, int scale_per_sample,
                                     const float* bias, int with_act, float* dx, float* dscale_sample, int B, int C, int hw,
                                     int groups, void* workspace, int accumulate_dx, void* stream) {
    LOCATE_REQUIRE(B > 0 && C > 0 && hw > 0 && workspace, "locate_norm_bwd_fused: bad shape or missing workspace");
    LOCATE_REQUIRE(groups >= 1 && groups <= NORM_MAX_GROUPS && B % groups == 0, "locate_norm_bwd_fused: bad group count");
    LOCATE_REQUIRE(!with_act || bias, "locate_norm_bwd_fused: with_act needs the bias");
    LOCATE_REQUIRE(!scale_per_sample == !dscale_sample, "locate_norm_bwd_fused: dscale_sample goes with scale_per_sample");
    LOCATE_REQUIRE((int64_t)B * C / groups * hw < (1ll << 33), "locate_norm_bwd_fused: more than 2^33 elements per group");
    const int64_t planes = (int64_t)B * C, planes_g = planes / groups;
    LOCATE_REQUIRE(groups == 1 || ((planes_g * hw) & 3) == 0, "locate_norm_bwd_fused: grouped tensors need a group size that is a multiple of 4");
    double* partial = static_cast<double*>(workspace);
    float* S1 = reinterpret_cast<float*>(partial + (size_t)NBF_BLOCKS * NORM_MAX_GROUPS * 2);
    float* S2 = S1 + planes;
    int64_t blocks = cdiv64(planes, 4);
    if (blocks > NBF_BLOCKS) blocks = NBF_BLOCKS;
    hipStream_t st = as_stream(stream);
    const bool small = with_act ? launch_norm_plane_small<true>((int)blocks, st, x, g, stats, scale, scale_per_sample, bias, C, S1, S2, planes, planes_g, hw, partial, dscale_sample)
                                : launch_norm_plane_small<false>((int)blocks, st, x, g, stats, scale, scale_per_sample, bias, C, S1, S2, planes, planes_g, hw, partial, dscale_sample);
    if (small) {
    } else if (with_act)
        norm_bwd_plane_kernel<true><<<(int)blocks, 256, 0, st>>>(x, g, stats, scale, scale_per_sample, bias, C, S1, S2, planes, planes_g, hw, partial, dscale_sample);
    else
        norm_bwd_plane_kernel<false><<<(int)blocks, 256, 0, st>>>(x, g, stats, scale, scale_per_sample, bias, C, S1, S2, planes, planes_g, hw, partial, dscale_sample);
    LOCATE_LAUNCH_CHECK("locate_norm_bwd_fused(plane)");
    const dim3 grid(stream_grid(planes_g * hw, 1024), groups);
    if (with_act)
        norm_bwd_dx_kernel<true><<<grid, 256, 0, st>>>(x, g, stats, scale, scale_per_sample, bias, partial, (int)blocks, dx, planes_g, C, hw, accumulate_dx);
    else
        norm_bwd_dx_kernel<false><<<grid, 256, 0, st>>>(x, g, stats, scale, scale_per_sample, bias, partial, (int)blocks, dx, planes_g, C, hw, accumulate_dx);
    LOCATE_LAUNCH_CHECK("locate_norm_bwd_fused(dx)");
    return LOCATE_OK;
}

// dbias[c] = sum_b S1[b, c];  dscale[c] = sum_b S2[b, c] / std(group of b)   - norm_bwd_final_kernel's arithmetic (16 channels x
// 16 batch slots per block, slots added in order), for all norms of a pass in one launch.
// records: p0 S1, p1 S2, p2 stats, p3 dscale [C] | 0, p4 dbias [C];  i0 B, i1 C, i2 groups, i5 first block, i6 block count
__global__ void __launch_bounds__(256) fin_norm_channels_kernel(const FinBatch batch, int n_rec) {
    __shared__ float part_b[256], part_y[256];
    int ri = 0;
    for (int k = 1; k < n_rec; ++k)
        if ((int)blockIdx.x >= batch.r[k].i[5]) ri = k;
    const FinRec& R = batch.r[ri];
    const int bx = (int)blockIdx.x - R.i[5];
    const float* __restrict__ S1 = static_cast<const float*>(R.p[0]);
    const float* __restrict__ S2 = static_cast<const float*>(R.p[1]);
    const float* __restrict__ stats = static_cast<const float*>(R.p[2]);
    float* __restrict__ dscale = static_cast<float*>(const_cast<void*>(R.p[3]));
    float* __restrict__ dbias = static_cast<float*>(const_cast<void*>(R.p[4]));
    const int B = R.i[0], C = R.i[1], groups = R.i[2];
    const int Bg = B / groups;
    const int cl = threadIdx.x & (NF_CH - 1), slot = threadIdx.x / NF_CH;
    const int c = bx * NF_CH + cl;
    float db = 0.0f, dy = 0.0f;
    if (c < C) {
        for (int b = slot; b < B; b += NF_SLOTS) {
            const float sf = stats[2 * (b / Bg) + 1];
            const int64_t p = (int64_t)b * C + c;
            db += S1[p];
            dy += S2[p] / sf;
        }
    }
    part_b[threadIdx.x] = db;
    part_y[threadIdx.x] = dy;
    __syncthreads();
    if (slot == 0 && c < C) {
        float tb = 0.0f, ty = 0.0f;
        for (int s2 = 0; s2 < NF_SLOTS; ++s2) { tb += part_b[s2 * NF_CH + cl]; ty += part_y[s2 * NF_CH + cl]; }
        if (dbias) dbias[c] = tb;
        if (dscale) dscale[c] = ty;
    }
}

LOCATE_API int locate_fin_norm_channels(const void* records, int n, void* stream) {
    LOCATE_REQUIRE(records && n > 0, "locate_fin_norm_channels: bad arguments");
    const FinRec* rec = static_cast<const FinRec*>(records);
    for (int at = 0; at < n; at += FIN_MAX) {
        FinBatch b = {};
        const int m = n - at < FIN_MAX ? n - at : FIN_MAX;
        int blocks = 0;
        for (int k = 0; k < m; ++k) {
            b.r[k] = rec[at + k];
            FinRec& R = b.r[k];
            LOCATE_REQUIRE(R.p[0] && R.p[1] && R.p[2] && (R.p[3] || R.p[4]) && R.i[0] > 0 && R.i[1] > 0 && R.i[2] >= 1 &&
                           R.i[2] <= NORM_MAX_GROUPS && R.i[0] % R.i[2] == 0, "locate_fin_norm_channels: bad record %d", at + k);
            R.i[5] = blocks;
            R.i[6] = (R.i[1] + NF_CH - 1) / NF_CH;
            blocks += R.i[6];
        }
        fin_norm_channels_kernel<<<blocks, 256, 0, as_stream(stream)>>>(b, m);
        LOCATE_LAUNCH_CHECK("locate_fin_norm_channels");
    }
    return LOCATE_OK;
}

// ---------------------------------------------------------------------------------------------
// per-channel sum over batch and space: out[c] = sum_{b,hw} g[b, c, hw]   (bias gradients of the 1x1 skip
// convs, scale.py:28-34, and of the style Linears, linear.py:10)
// ---------------------------------------------------------------------------------------------
// Blocks of 1024 threads: blockIdx.x = channel, blockIdx.y = batch slice (so that few channels still fill the chip),
// 16-byte loads, fixed summation order (deterministic); a second tiny kernel adds the slices.
#define CS_MAX_SLICES 64
__global__ void __launch_bounds__(1024) channel_sum_kernel(const float* __restrict__ g, float* __restrict__ out, int B, int C,
                                                           int hw, int64_t batch_stride, int per_slice) {
    __shared__ float scratch[16];
    const bool vec = (hw & 3) == 0 && (batch_stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
    if (channel_sum_small(per_slice, hw, vec)) {          // sixteen channels per block, a wave each (common.h: same bits)
        const int c = blockIdx.x * 16 + ((int)threadIdx.x >> 6);
        const int b0 = blockIdx.y * per_slice;
        const int nb = min(per_slice, B - b0);
        if (c < C) {
            const float t = nb > 0 ? channel_sum_wave(g, batch_stride, c, hw, b0, nb, vec, threadIdx.x & 63) : 0.0f;
            if ((threadIdx.x & 63) == 0) out[(int64_t)blockIdx.y * C + c] = t;
        }
        return;
    }
    const int c = blockIdx.x;
    const int b0 = blockIdx.y * per_slice;
    const int nb = min(per_slice, B - b0);
    float acc = 0.0f;
    if (nb > 0) {
        if (vec) {
            const int hw4 = hw >> 2;
            const int total = nb * hw4;
            for (int i = threadIdx.x; i < total; i += blockDim.x) {
                const int b = i / hw4, r = i - b * hw4;
                const float4 v = reinterpret_cast<const float4*>(g + (int64_t)(b0 + b) * batch_stride + (int64_t)c * hw)[r];
                acc += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            const int total = nb * hw;
            for (int i = threadIdx.x; i < total; i += blockDim.x) {
                const int b = i / hw, r = i - b * hw;
                acc += g[(int64_t)(b0 + b) * batch_stride + (int64_t)c * hw + r];
            }
        }
    }
    acc = block_sum<float>(acc, scratch);
    if (threadIdx.x == 0) out[(int64_t)blockIdx.y * C + c] = acc;
}

__global__ void __launch_bounds__(256) channel_sum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int C,
                                                                int slices) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float acc = 0.0f;
    for (int s = 0; s < slices; ++s) acc += part[(int64_t)s * C + c];
    out[c] = acc;
}

static int channel_sum_slices(int B, int C, int hw) {
    if (C >= 512 || (int64_t)B * hw < 8192) return 1;          // enough blocks already, or too little work to split
    int s = (1024 + C - 1) / C;
    if (s > B) s = B;
    if (s > CS_MAX_SLICES) s = CS_MAX_SLICES;
    return s < 1 ? 1 : s;
}

LOCATE_API size_t locate_channel_sum_workspace_bytes(int B, int C, int hw) {
    const int s = channel_sum_slices(B, C, hw);
    return s > 1 ? (size_t)s * C * sizeof(float) : 0;
}

// workspace: locate_channel_sum_workspace_bytes(B, C, hw) (may be null when that is 0)
LOCATE_API int locate_channel_sum(const float* g, float* out, int B, int C, int hw, int64_t batch_stride, void* workspace,
                                  void* stream) {
    LOCATE_REQUIRE(B > 0 && C > 0 && hw > 0 && batch_stride >= (int64_t)C * hw && (int64_t)B * hw < (1ll << 31),
                   "locate_channel_sum: bad shape");
    const int slices = channel_sum_slices(B, C, hw);
    const bool vec_h = (hw & 3) == 0 && (batch_stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
    if (slices == 1) {
        const int gx = channel_sum_small(B, hw, vec_h) ? (C + 15) / 16 : C;
        channel_sum_kernel<<<dim3(gx, 1), 1024, 0, as_stream(stream)>>>(g, out, B, C, hw, batch_stride, B);
        LOCATE_LAUNCH_CHECK("locate_channel_sum");
        return LOCATE_OK;
    }
    LOCATE_REQUIRE(workspace, "locate_channel_sum: missing workspace");
    float* part = static_cast<float*>(workspace);
    const int per_slice = (B + slices - 1) / slices;
    const int gxs = channel_sum_small(per_slice, hw, vec_h) ? (C + 15) / 16 : C;
    channel_sum_kernel<<<dim3(gxs, slices), 1024, 0, as_stream(stream)>>>(g, part, B, C, hw, batch_stride, per_slice);
    LOCATE_LAUNCH_CHECK("locate_channel_sum");
    channel_sum_final_kernel<<<(C + 255) / 256, 256, 0, as_stream(stream)>>>(part, out, C, slices);
    LOCATE_LAUNCH_CHECK("locate_channel_sum(final)");
    return LOCATE_OK;
}
