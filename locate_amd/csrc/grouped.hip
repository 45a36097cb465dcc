// Grouped convolutions of the SEPARABLE switch (libs/config.py:53) - every one of them is HBM-bound data movement with a
// handful of FMAs per element, so none of it goes near the matrix cores:
//   * depthwise k x k convs with a channel multiplier (libs/conv.py:14-18: groups = in_features, out = in * FEATURE_MULTIPLIER),
//     regular and transposed, forward / input gradient / weight gradient;
//   * feature attention's single full-size grouped conv (libs/attention.py:15-21: kernel = the whole S x S map,
//     groups = C / BOTTLENECK): each output is the dot product of L = (C/groups) * S * S CONTIGUOUS input values with one
//     weight row.
// 1/sigma of spectral norm is applied per batch element exactly like in conv.hip (one scalar, or one per stacked call).
#include "common.h"

struct DwGeom {
    int B, C, H, W, M, KH, KW, S, PH, PW, OH, OW;
};

static inline DwGeom dw_geom(const int* g) { return DwGeom{g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[11]}; }

static bool dw_geom_ok(const DwGeom& g) {
    if (g.B < 1 || g.C < 1 || g.M < 1 || g.H < 1 || g.W < 1 || g.OH < 1 || g.OW < 1 || g.S < 1 || g.KH < 1 || g.KW < 1) return false;
    if (g.PH < 0 || g.PW < 0) return false;
    const int wide = g.C > g.M ? g.C : g.M, narrow = g.C > g.M ? g.M : g.C;
    if (wide % narrow) return false;
    const int rem_h = g.H + 2 * g.PH - g.KH - (g.OH - 1) * g.S, rem_w = g.W + 2 * g.PW - g.KW - (g.OW - 1) * g.S;
    if (rem_h < 0 || rem_h >= g.S || rem_w < 0 || rem_w >= g.S) return false;      // OH = floor((H + 2 PH - KH) / S) + 1
    if ((int64_t)g.B * wide * g.H * g.W >= (1ll << 31) || (int64_t)g.B * wide * g.OH * g.OW >= (1ll << 31)) return false;
    return true;
}

__device__ __forceinline__ float batch_scale(const float* __restrict__ scale, int group_batch, int stride, unsigned b) {
    if (!scale) return 1.0f;
    return group_batch > 0 ? scale[(b / (unsigned)group_batch) * stride] : scale[0];
}

// One thread per output element.  TRANSPOSED = false: R (big [C,H,W] -> small [M,OH,OW]); true: R^T (small -> big).
// The wide side has `rows` channels (= weight rows), the narrow side rows / mult: an output channel on the wide side reads
// one input channel (co / mult) with weight row co; an output channel on the narrow side sums its mult rows.
template <bool TRANSPOSED>
__global__ void __launch_bounds__(256) dwconv_kernel(DwGeom g, const float* __restrict__ x, int64_t x_bs, const float* __restrict__ w,
                                                     const float* __restrict__ scale, int sgb, int sst, float* __restrict__ y,
                                                     int64_t y_bs) {
    const int Cin = TRANSPOSED ? g.M : g.C, Cout = TRANSPOSED ? g.C : g.M;
    const int IH = TRANSPOSED ? g.OH : g.H, IW = TRANSPOSED ? g.OW : g.W;
    const int OH = TRANSPOSED ? g.H : g.OH, OW = TRANSPOSED ? g.W : g.OW;
    const bool expand = Cout >= Cin;
    const int mult = expand ? Cout / Cin : Cin / Cout;
    const int terms = expand ? 1 : mult;
    const int KK = g.KH * g.KW;
    const unsigned n = (unsigned)g.B * Cout * OH * OW;
    const unsigned stride = gridDim.x * blockDim.x;
    const DivU32 dw((unsigned)OW), dh((unsigned)OH), dc((unsigned)Cout), dm((unsigned)mult);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned t, uox, p, uoy, b, co;
        dw.divmod(i, t, uox);
        dh.divmod(t, p, uoy);
        dc.divmod(p, b, co);
        const int oy = (int)uoy, ox = (int)uox;
        const unsigned row0 = expand ? co : co * mult;
        const unsigned ci0 = expand ? dm.div(co) : co * mult;
        float acc = 0.0f;
        for (int r = 0; r < terms; ++r) {
            const float* __restrict__ xp = x + (int64_t)b * x_bs + (int64_t)(ci0 + r) * IH * IW;
            const float* __restrict__ wp = w + (int64_t)(row0 + r) * KK;
            for (int ty = 0; ty < g.KH; ++ty) {
                int iy;
                if (TRANSPOSED) {
                    const int ny = oy + g.PH - ty;
                    if (ny < 0) continue;
                    iy = ny / g.S;
                    if (iy * g.S != ny || iy >= IH) continue;
                } else {
                    iy = oy * g.S - g.PH + ty;
                    if (iy < 0 || iy >= IH) continue;
                }
                for (int tx = 0; tx < g.KW; ++tx) {
                    int ix;
                    if (TRANSPOSED) {
                        const int nx = ox + g.PW - tx;
                        if (nx < 0) continue;
                        ix = nx / g.S;
                        if (ix * g.S != nx || ix >= IW) continue;
                    } else {
                        ix = ox * g.S - g.PW + tx;
                        if (ix < 0 || ix >= IW) continue;
                    }
                    acc = fmaf(wp[ty * g.KW + tx], xp[iy * IW + ix], acc);
                }
            }
        }
        y[(int64_t)b * y_bs + ((int64_t)co * OH + oy) * OW + ox] = acc * batch_scale(scale, sgb, sst, b);
    }
}

static int dwconv_launch(const int* geom, bool transposed, const float* x, int64_t x_bs, const float* w, const float* scale,
                         int sgb, int sst, float* y, int64_t y_bs, void* stream, const char* name) {
    LOCATE_REQUIRE(geom && x && w && y, "%s: null argument", name);
    const DwGeom g = dw_geom(geom);
    LOCATE_REQUIRE(dw_geom_ok(g), "%s: bad geometry", name);
    LOCATE_REQUIRE(sgb >= 0 && (sgb == 0 || g.B % sgb == 0), "%s: batch %d does not split into calls of %d", name, g.B, sgb);
    const int64_t in_plane = transposed ? (int64_t)g.M * g.OH * g.OW : (int64_t)g.C * g.H * g.W;
    const int64_t out_plane = transposed ? (int64_t)g.C * g.H * g.W : (int64_t)g.M * g.OH * g.OW;
    LOCATE_REQUIRE(x_bs >= in_plane && y_bs >= out_plane, "%s: batch strides smaller than one sample", name);
    const int grid = stream_grid((int64_t)g.B * out_plane, 256);
    if (transposed)
        dwconv_kernel<true><<<grid, 256, 0, as_stream(stream)>>>(g, x, x_bs, w, scale, sgb, sst, y, y_bs);
    else
        dwconv_kernel<false><<<grid, 256, 0, as_stream(stream)>>>(g, x, x_bs, w, scale, sgb, sst, y, y_bs);
    LOCATE_LAUNCH_CHECK(name);
    return LOCATE_OK;
}

LOCATE_API int locate_dwconv_fwd(const int* geom, const float* x, int64_t x_bs, const float* w, const float* scale,
                                 int scale_group_batch, int scale_stride, float* y, int64_t y_bs, void* stream) {
    return dwconv_launch(geom, false, x, x_bs, w, scale, scale_group_batch, scale_stride, y, y_bs, stream, "locate_dwconv_fwd");
}

LOCATE_API int locate_dwconv_dgrad(const int* geom, const float* gy, int64_t gy_bs, const float* w, const float* scale,
                                   int scale_group_batch, int scale_stride, float* gx, int64_t gx_bs, void* stream) {
    return dwconv_launch(geom, true, gy, gy_bs, w, scale, scale_group_batch, scale_stride, gx, gx_bs, stream, "locate_dwconv_dgrad");
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------
// gw[row][ty][tx] = sum_b s_b sum_{py,px} small[b, cs, py, px] * big[b, cb, py*S - PH + ty, px*S - PW + tx]
// grid (rows, chunks): a block reduces one chunk of the (b, py, px) items of one row into K*K partial sums (registers ->
// wave shuffles -> LDS); the finisher adds the chunks in a fixed order (deterministic, no atomics).
static int dw_wgrad_chunks(const DwGeom& g) {
    const int rows = g.C > g.M ? g.C : g.M;
    const int64_t items = (int64_t)g.B * g.OH * g.OW;
    int64_t c = (2048 + rows - 1) / rows;
    const int64_t most = (items + 1023) / 1024;
    if (c > most) c = most;
    if (c > 256) c = 256;
    if (c < 1) c = 1;
    return (int)c;
}

template <int K>
__global__ void __launch_bounds__(256) dw_wgrad_kernel(DwGeom g, const float* __restrict__ big, int64_t big_bs,
                                                       const float* __restrict__ small, int64_t small_bs,
                                                       const float* __restrict__ scale, int sgb, int sst, float* __restrict__ part,
                                                       int per_chunk) {
    constexpr int KK = K * K;
    __shared__ float red[4][KK];
    const unsigned row = blockIdx.x, chunk = blockIdx.y;
    const int rows = g.C > g.M ? g.C : g.M;
    const int mult = g.C > g.M ? g.C / g.M : g.M / g.C;
    const unsigned cb = g.C >= g.M ? row : row / (unsigned)mult;       // channel on the big side
    const unsigned cs = g.M >= g.C ? row : row / (unsigned)mult;       // channel on the small side
    const unsigned items = (unsigned)g.B * g.OH * g.OW;
    const unsigned lo = chunk * (unsigned)per_chunk;
    const unsigned hi = min(items, lo + (unsigned)per_chunk);
    const DivU32 dw((unsigned)g.OW), dh((unsigned)g.OH);
    float acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = 0.0f;
    for (unsigned i = lo + threadIdx.x; i < hi; i += 256) {
        unsigned t, upx, b, upy;
        dw.divmod(i, t, upx);
        dh.divmod(t, b, upy);
        const float sv = small[(int64_t)b * small_bs + ((int64_t)cs * g.OH + upy) * g.OW + upx] * batch_scale(scale, sgb, sst, b);
        const float* __restrict__ bp = big + (int64_t)b * big_bs + (int64_t)cb * g.H * g.W;
        const int y0 = (int)upy * g.S - g.PH, x0 = (int)upx * g.S - g.PW;
#pragma unroll
        for (int ty = 0; ty < K; ++ty) {
            const int iy = y0 + ty;
            const bool oky = iy >= 0 && iy < g.H;
#pragma unroll
            for (int tx = 0; tx < K; ++tx) {
                const int ix = x0 + tx;
                const bool ok = oky && ix >= 0 && ix < g.W;
                const float bv = ok ? bp[iy * g.W + ix] : 0.0f;
                acc[ty * K + tx] = fmaf(sv, bv, acc[ty * K + tx]);
            }
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < KK; ++t) {
        const float v = wave_sum(acc[t]);
        if (lane == 0) red[wid][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < KK)
        part[((int64_t)chunk * rows + row) * KK + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// one wave per row: gw = inv * sum_chunks part, partial[row] = <unscaled gw row, w_ref row> (double)
__global__ void __launch_bounds__(64) dw_wgrad_final_kernel(const float* __restrict__ part, int chunks, int rows, int KK,
                                                            const float* __restrict__ inv_scale, const float* __restrict__ w_ref,
                                                            float* __restrict__ gw, double* __restrict__ inner_partial) {
    const int row = blockIdx.x, t = threadIdx.x;
    float s = 0.0f;
    if (t < KK)
        for (int c = 0; c < chunks; ++c) s += part[((int64_t)c * rows + row) * KK + t];
    if (t < KK) gw[(int64_t)row * KK + t] = inv_scale ? s * inv_scale[0] : s;
    if (inner_partial) {
        const double d = wave_sum_d(t < KK ? (double)s * (double)w_ref[(int64_t)row * KK + t] : 0.0);
        if (t == 0) inner_partial[row] = d;
    }
}

LOCATE_API size_t locate_dwconv_wgrad_workspace_bytes(const int* geom) {
    if (!geom) return 0;
    const DwGeom g = dw_geom(geom);
    if (!dw_geom_ok(g)) return 0;
    const int rows = g.C > g.M ? g.C : g.M;
    return (size_t)dw_wgrad_chunks(g) * rows * g.KH * g.KW * sizeof(float);
}

LOCATE_API int locate_dwconv_wgrad_partials(const int* geom) {
    if (!geom) return 0;
    const DwGeom g = dw_geom(geom);
    return g.C > g.M ? g.C : g.M;
}

LOCATE_API int locate_dwconv_wgrad(const int* geom, const float* big, int64_t big_bs, const float* small, int64_t small_bs,
                                   float* gw, const float* w_ref, const float* inv_scale, int scale_group_batch, int scale_stride,
                                   double* inner_partial, void* workspace, void* stream) {
    LOCATE_REQUIRE(geom && big && small && gw && workspace, "locate_dwconv_wgrad: null argument");
    const DwGeom g = dw_geom(geom);
    LOCATE_REQUIRE(dw_geom_ok(g), "locate_dwconv_wgrad: bad geometry");
    LOCATE_REQUIRE(g.KH == g.KW && g.KH <= 5, "locate_dwconv_wgrad: square kernels up to 5 x 5 only (got %d x %d)", g.KH, g.KW);
    LOCATE_REQUIRE((w_ref != nullptr) == (inner_partial != nullptr), "locate_dwconv_wgrad: w_ref and inner_partial go together");
    LOCATE_REQUIRE(scale_group_batch == 0 || (inv_scale && !w_ref && g.B % scale_group_batch == 0),
                   "locate_dwconv_wgrad: per-call scales exclude w_ref / inner_partial");
    LOCATE_REQUIRE(big_bs >= (int64_t)g.C * g.H * g.W && small_bs >= (int64_t)g.M * g.OH * g.OW,
                   "locate_dwconv_wgrad: batch strides smaller than one sample");
    const int rows = g.C > g.M ? g.C : g.M;
    const int chunks = dw_wgrad_chunks(g);
    const int64_t items = (int64_t)g.B * g.OH * g.OW;
    const int per_chunk = (int)((items + chunks - 1) / chunks);
    float* part = static_cast<float*>(workspace);
    const bool per_call = scale_group_batch > 0;
    const float* load_scale = per_call ? inv_scale : nullptr;
    hipStream_t st = as_stream(stream);
    const dim3 grid(rows, chunks);
    switch (g.KH) {
        case 1: dw_wgrad_kernel<1><<<grid, 256, 0, st>>>(g, big, big_bs, small, small_bs, load_scale, scale_group_batch, scale_stride, part, per_chunk); break;
        case 2: dw_wgrad_kernel<2><<<grid, 256, 0, st>>>(g, big, big_bs, small, small_bs, load_scale, scale_group_batch, scale_stride, part, per_chunk); break;
        case 3: dw_wgrad_kernel<3><<<grid, 256, 0, st>>>(g, big, big_bs, small, small_bs, load_scale, scale_group_batch, scale_stride, part, per_chunk); break;
        case 4: dw_wgrad_kernel<4><<<grid, 256, 0, st>>>(g, big, big_bs, small, small_bs, load_scale, scale_group_batch, scale_stride, part, per_chunk); break;
        default: dw_wgrad_kernel<5><<<grid, 256, 0, st>>>(g, big, big_bs, small, small_bs, load_scale, scale_group_batch, scale_stride, part, per_chunk); break;
    }
    LOCATE_LAUNCH_CHECK("locate_dwconv_wgrad");
    dw_wgrad_final_kernel<<<rows, 64, 0, st>>>(part, chunks, rows, g.KH * g.KW, per_call ? nullptr : inv_scale, w_ref, gw, inner_partial);
    LOCATE_LAUNCH_CHECK("locate_dwconv_wgrad(final)");
    return LOCATE_OK;
}

// ---- full-size grouped conv == per-group dot products --------------------------------------------------------------------
// x [B, G*L] (batch stride x_bs), w [G, L], y [B, G]
__global__ void __launch_bounds__(256) groupdot_fwd_kernel(const float* __restrict__ x, int64_t x_bs, const float* __restrict__ w,
                                                           const float* __restrict__ scale, int sgb, int sst, float* __restrict__ y,
                                                           int G, int L) {
    __shared__ float scratch[16];
    const unsigned g = blockIdx.x, b = blockIdx.y;
    const float* __restrict__ xp = x + (int64_t)b * x_bs + (int64_t)g * L;
    const float* __restrict__ wp = w + (int64_t)g * L;
    float acc = 0.0f;
    if ((L & 3) == 0 && (x_bs & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        const float4* w4 = reinterpret_cast<const float4*>(wp);
        for (int i = threadIdx.x; i < L / 4; i += 256) {
            const float4 a = x4[i], c = w4[i];
            acc = fmaf(a.x, c.x, fmaf(a.y, c.y, fmaf(a.z, c.z, fmaf(a.w, c.w, acc))));
        }
    } else {
        for (int i = threadIdx.x; i < L; i += 256) acc = fmaf(xp[i], wp[i], acc);
    }
    const float total = block_sum(acc, scratch);
    if (threadIdx.x == 0) y[(int64_t)b * G + g] = total * batch_scale(scale, sgb, sst, b);
}

// gx[b, g, l] = s_b gy[b, g] w[g, l]
__global__ void __launch_bounds__(256) groupdot_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                             const float* __restrict__ scale, int sgb, int sst,
                                                             float* __restrict__ gx, int64_t gx_bs, int G, int L) {
    const unsigned g = blockIdx.y, b = blockIdx.z;
    const float f = gy[(int64_t)b * G + g] * batch_scale(scale, sgb, sst, b);
    const float* __restrict__ wp = w + (int64_t)g * L;
    float* __restrict__ op = gx + (int64_t)b * gx_bs + (int64_t)g * L;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) op[i] = f * wp[i];
}

// gw[g, l] = inv * sum_b s_b gy[b, g] x[b, g, l]; one thread per (g, l), batch loop in order (deterministic)
__global__ void __launch_bounds__(256) groupdot_wgrad_kernel(const float* __restrict__ x, int64_t x_bs, const float* __restrict__ gy,
                                                             const float* __restrict__ inv_scale, int sgb, int sst,
                                                             const float* __restrict__ w_ref, float* __restrict__ gw,
                                                             double* __restrict__ inner_partial, int B, int G, int L) {
    __shared__ double scratch[16];
    const unsigned g = blockIdx.y;
    const int l = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.0f;
    if (l < L) {
        const float* __restrict__ xp = x + (int64_t)g * L + l;
        for (int b = 0; b < B; ++b) {
            const float f = gy[(int64_t)b * G + g] * (sgb > 0 ? inv_scale[((unsigned)b / (unsigned)sgb) * sst] : 1.0f);
            acc = fmaf(f, xp[(int64_t)b * x_bs], acc);
        }
        gw[(int64_t)g * L + l] = (sgb == 0 && inv_scale) ? acc * inv_scale[0] : acc;
    }
    if (inner_partial) {
        const double d = block_sum<double>(l < L ? (double)acc * (double)w_ref[(int64_t)g * L + l] : 0.0, scratch);
        if (threadIdx.x == 0) inner_partial[(int64_t)g * gridDim.x + blockIdx.x] = d;
    }
}

static bool groupdot_ok(int B, int G, int L) {
    return B > 0 && G > 0 && L > 0 && B <= 65535 && G <= 65535 && (int64_t)B * G * L < (1ll << 31);
}

LOCATE_API int locate_groupdot_fwd(const float* x, int64_t x_bs, const float* w, const float* scale, int scale_group_batch,
                                   int scale_stride, float* y, int B, int G, int L, void* stream) {
    LOCATE_REQUIRE(x && w && y && groupdot_ok(B, G, L) && x_bs >= (int64_t)G * L, "locate_groupdot_fwd: bad argument");
    LOCATE_REQUIRE(scale_group_batch >= 0 && (scale_group_batch == 0 || B % scale_group_batch == 0), "locate_groupdot_fwd: bad call split");
    groupdot_fwd_kernel<<<dim3(G, B), 256, 0, as_stream(stream)>>>(x, x_bs, w, scale, scale_group_batch, scale_stride, y, G, L);
    LOCATE_LAUNCH_CHECK("locate_groupdot_fwd");
    return LOCATE_OK;
}

LOCATE_API int locate_groupdot_dgrad(const float* gy, const float* w, const float* scale, int scale_group_batch, int scale_stride,
                                     float* gx, int64_t gx_bs, int B, int G, int L, void* stream) {
    LOCATE_REQUIRE(gy && w && gx && groupdot_ok(B, G, L) && gx_bs >= (int64_t)G * L, "locate_groupdot_dgrad: bad argument");
    LOCATE_REQUIRE(scale_group_batch >= 0 && (scale_group_batch == 0 || B % scale_group_batch == 0), "locate_groupdot_dgrad: bad call split");
    int bx = (L + 1023) / 1024;
    if (bx > 64) bx = 64;
    groupdot_dgrad_kernel<<<dim3(bx, G, B), 256, 0, as_stream(stream)>>>(gy, w, scale, scale_group_batch, scale_stride, gx, gx_bs, G, L);
    LOCATE_LAUNCH_CHECK("locate_groupdot_dgrad");
    return LOCATE_OK;
}

LOCATE_API int locate_groupdot_wgrad_partials(int G, int L) { return G * ((L + 255) / 256); }

LOCATE_API int locate_groupdot_wgrad(const float* x, int64_t x_bs, const float* gy, float* gw, const float* w_ref,
                                     const float* inv_scale, int scale_group_batch, int scale_stride, double* inner_partial, int B,
                                     int G, int L, void* stream) {
    LOCATE_REQUIRE(x && gy && gw && groupdot_ok(B, G, L) && x_bs >= (int64_t)G * L, "locate_groupdot_wgrad: bad argument");
    LOCATE_REQUIRE((w_ref != nullptr) == (inner_partial != nullptr), "locate_groupdot_wgrad: w_ref and inner_partial go together");
    LOCATE_REQUIRE(scale_group_batch == 0 || (inv_scale && !w_ref && B % scale_group_batch == 0),
                   "locate_groupdot_wgrad: per-call scales exclude w_ref / inner_partial");
    groupdot_wgrad_kernel<<<dim3((L + 255) / 256, G), 256, 0, as_stream(stream)>>>(x, x_bs, gy, inv_scale, scale_group_batch, scale_stride,
                                                                                  w_ref, gw, inner_partial, B, G, L);
    LOCATE_LAUNCH_CHECK("locate_groupdot_wgrad");
    return LOCATE_OK;
}
