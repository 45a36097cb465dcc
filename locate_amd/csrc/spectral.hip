// Spectral normalisation state machine (reference libs/spectral_norm.py:8-32,57-59), pure HBM traffic:
// on EVERY forward, per wrapped weight W_bar viewed as [h, wd] (h = shape[0], i.e. C_in for ConvTranspose):
//     t = W^T u ;  v <- t / (|t| + 1e-12) ;  s = W v ;  u <- s / (|s| + 1e-12) ;  sigma = u . (W v)
// Three passes over W in the reference (two GEMVs + a third for sigma) plus a full-size W_bar/sigma write;
// here: two passes (column sums, row dots; W v = (W t)/(|t|+eps) by linearity, sigma = u . s from the same s)
// and no normalised copy (1/sigma is folded into the conv's weight panel, conv.hip).
//
// The kernels take a table of layers so that ALL spectral-norm layers of a network are advanced by the same
// four launches (blockIdx.y = layer; blocks beyond a layer's own work exit at once).  u, v are updated in
// place; sigma[0] = sigma, sigma[1] = 1/sigma; wv = W v is kept for the gradient of u.
//
// Backward of W_n = W_bar / sigma with sigma = u . (W_bar v)   (the reference's autograd sees
// the u, v left by the LATEST forward because they are overwritten through .data - oracle SigmaFn):
//     dsigma = -<G, W_bar> / sigma^2
//     dW_bar = G / sigma + dsigma * u v^T ;   du = dsigma * (W v)_k ;   dv = dsigma * W^T u
#include "common.h"

struct SnLayer {
    const float* w;   // [h, wd] row-major (W_bar)
    float* u;         // [h]
    float* v;         // [wd]
    float* sigma;     // [2]
    float* wv;        // [h]
    float* t;         // [wd]     scratch: W^T u
    float* s;         // [h]      scratch: W t
    float* tpart;     // [nchunk][wd] scratch: partial column sums
    int h, wd, nchunk, pad;
};

#define SN_ROWS 64      // rows per column-sum chunk
#define SN_COLS 256     // columns per block

template <bool BATCHED>
__device__ __forceinline__ SnLayer sn_get(const SnLayer& single, const SnLayer* table) {
    return BATCHED ? table[blockIdx.y] : single;
}

// tpart[chunk][col] = sum_{i in chunk} W[i][col] * u[i]
template <bool BATCHED>
__global__ void __launch_bounds__(SN_COLS) sn_colsum_kernel(const SnLayer single, const SnLayer* __restrict__ table) {
    const SnLayer L = sn_get<BATCHED>(single, table);
    const int nstrip = (L.wd + SN_COLS - 1) / SN_COLS;
    if ((int)blockIdx.x >= nstrip * L.nchunk) return;
    const int strip = blockIdx.x % nstrip, chunk = blockIdx.x / nstrip;
    const int col = strip * SN_COLS + threadIdx.x;
    const int i0 = chunk * SN_ROWS;
    const int i1 = min(i0 + SN_ROWS, L.h);
    if (col >= L.wd) return;
    float acc = 0.0f;
    int i = i0;
    for (; i + 8 <= i1; i += 8) {          // 8 independent loads in flight, summed in row order
        float wv8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wv8[k] = L.w[(int64_t)(i + k) * L.wd + col];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = fmaf(wv8[k], L.u[i + k], acc);
    }
    for (; i < i1; ++i) acc = fmaf(L.w[(int64_t)i * L.wd + col], L.u[i], acc);
    L.tpart[(int64_t)chunk * L.wd + col] = acc;
}

// t[col] = sum_chunk tpart[chunk][col]
template <bool BATCHED>
__global__ void __launch_bounds__(SN_COLS) sn_tsum_kernel(const SnLayer single, const SnLayer* __restrict__ table) {
    const SnLayer L = sn_get<BATCHED>(single, table);
    const int col = blockIdx.x * SN_COLS + threadIdx.x;
    if (col >= L.wd) return;
    float acc = 0.0f;
    for (int c = 0; c < L.nchunk; ++c) acc += L.tpart[(int64_t)c * L.wd + col];
    L.t[col] = acc;
}

// s[row] = W[row, :] . t.   Wide layers (wd >= SN_WIDE): one block per row; narrow layers: one wave per row.
#define SN_WIDE 1024
__device__ __forceinline__ float sn_row_partial(const float* __restrict__ wr, const float* __restrict__ t, int wd, int first,
                                                int step) {
    float acc = 0.0f;
    if ((wd & 3) == 0 && ((reinterpret_cast<uintptr_t>(wr) & 15) == 0)) {
        const float4* w4 = reinterpret_cast<const float4*>(wr);
        const bool t_al = (reinterpret_cast<uintptr_t>(t) & 15) == 0;
        for (int j = first; j < (wd >> 2); j += step) {
            const float4 a = w4[j];
            float4 b;
            if (t_al) b = reinterpret_cast<const float4*>(t)[j];
            else { b.x = t[4 * j]; b.y = t[4 * j + 1]; b.z = t[4 * j + 2]; b.w = t[4 * j + 3]; }
            acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        }
    } else {
        for (int j = first; j < wd; j += step) acc = fmaf(wr[j], t[j], acc);
    }
    return acc;
}

template <bool BATCHED>
__global__ void __launch_bounds__(256) sn_rowdot_kernel(const SnLayer single, const SnLayer* __restrict__ table) {
    __shared__ float wsum[4];
    const SnLayer L = sn_get<BATCHED>(single, table);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (L.wd >= SN_WIDE) {
        const int row = blockIdx.x;
        if (row >= L.h) return;
        float acc = sn_row_partial(L.w + (int64_t)row * L.wd, L.t, L.wd, threadIdx.x, 256);
        acc = wave_sum(acc);
        if (lane == 0) wsum[wid] = acc;
        __syncthreads();
        if (threadIdx.x == 0) L.s[row] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        return;
    }
    const int row = blockIdx.x * 4 + wid;
    if (row >= L.h) return;
    float acc = sn_row_partial(L.w + (int64_t)row * L.wd, L.t, L.wd, lane, 64);
    acc = wave_sum(acc);
    if (lane == 0) L.s[row] = acc;
}

// one block per layer: normalise and publish u, v, sigma, wv
// (1024 threads and four independent loads per pass: one block walks up to 19 200 columns three times, and at 256 threads with
// one load in flight each pass was a chain of 75 exposed round trips - 33 us for the widest layer, now 8)
template <bool BATCHED>
__global__ void __launch_bounds__(1024) sn_finalize_kernel(const SnLayer single, const SnLayer* __restrict__ table) {
    __shared__ double scratch[16];
    const SnLayer L = sn_get<BATCHED>(single, table);
    const float eps = 1e-12f;
    const float* __restrict__ tp = L.t;
    double a = 0.0;
    {
        int j = threadIdx.x;
        for (; j + 3 * (int)blockDim.x < L.wd; j += 4 * blockDim.x) {
            const float t0 = tp[j], t1 = tp[j + blockDim.x], t2 = tp[j + 2 * blockDim.x], t3 = tp[j + 3 * blockDim.x];
            a += (double)t0 * (double)t0;
            a += (double)t1 * (double)t1;
            a += (double)t2 * (double)t2;
            a += (double)t3 * (double)t3;
        }
        for (; j < L.wd; j += blockDim.x) a += (double)tp[j] * (double)tp[j];
    }
    a = block_sum<double>(a, scratch);
    const float nt = (float)sqrt(a);
    const float dt = nt + eps;
    float* __restrict__ vp = L.v;
#pragma unroll 4
    for (int j = threadIdx.x; j < L.wd; j += blockDim.x) vp[j] = tp[j] / dt;
    double b = 0.0;
    for (int i = threadIdx.x; i < L.h; i += blockDim.x) {
        const float si = L.s[i] / dt;     // (W v)[i]
        b += (double)si * (double)si;
    }
    b = block_sum<double>(b, scratch);
    const float ns = (float)sqrt(b);
    const float ds = ns + eps;
    double sg = 0.0;
    for (int i = threadIdx.x; i < L.h; i += blockDim.x) {
        const float si = L.s[i] / dt;
        const float ui = si / ds;
        L.u[i] = ui;
        L.wv[i] = si;
        sg += (double)ui * (double)si;
    }
    sg = block_sum<double>(sg, scratch);
    if (threadIdx.x == 0) {
        L.sigma[0] = (float)sg;
        L.sigma[1] = (float)(1.0 / sg);
    }
}

static inline int sn_nchunk(int h) { return (h + SN_ROWS - 1) / SN_ROWS; }

// scratch floats needed for one layer: t[wd] + s[h] + tpart[nchunk*wd]
LOCATE_API size_t locate_sn_workspace_bytes(int h, int wd) {
    return ((size_t)wd + h + (size_t)sn_nchunk(h) * wd) * sizeof(float);
}

static SnLayer sn_make(const float* w, float* u, float* v, float* sigma, float* wv, int h, int wd, float* ws) {
    SnLayer L;
    L.w = w; L.u = u; L.v = v; L.sigma = sigma; L.wv = wv;
    L.h = h; L.wd = wd; L.nchunk = sn_nchunk(h); L.pad = 0;
    L.t = ws; L.s = ws + wd; L.tpart = ws + wd + h;
    return L;
}

// One power iteration for one layer.  sigma: 2 floats, wv: h floats, workspace: locate_sn_workspace_bytes(h, wd).
LOCATE_API int locate_sn_power_iter(const float* w, float* u, float* v, float* sigma, float* wv, int h, int wd,
                                    void* workspace, void* stream) {
    LOCATE_REQUIRE(h > 0 && wd > 0 && w && u && v && sigma && wv && workspace, "locate_sn_power_iter: bad arguments");
    hipStream_t st = as_stream(stream);
    const SnLayer L = sn_make(w, u, v, sigma, wv, h, wd, static_cast<float*>(workspace));
    const int nstrip = (wd + SN_COLS - 1) / SN_COLS;
    sn_colsum_kernel<false><<<nstrip * L.nchunk, SN_COLS, 0, st>>>(L, nullptr);
    LOCATE_LAUNCH_CHECK("locate_sn_power_iter(colsum)");
    sn_tsum_kernel<false><<<nstrip, SN_COLS, 0, st>>>(L, nullptr);
    LOCATE_LAUNCH_CHECK("locate_sn_power_iter(tsum)");
    sn_rowdot_kernel<false><<<wd >= SN_WIDE ? h : (h + 3) / 4, 256, 0, st>>>(L, nullptr);
    LOCATE_LAUNCH_CHECK("locate_sn_power_iter(rowdot)");
    sn_finalize_kernel<false><<<1, 1024, 0, st>>>(L, nullptr);
    LOCATE_LAUNCH_CHECK("locate_sn_power_iter(finalize)");
    return LOCATE_OK;
}

// Batched form: `table` is a DEVICE array of `n_layers` records of 9 x 8 bytes:
//   { w, u, v, sigma, wv, t, s, tpart (pointers), (h | wd << 32), (nchunk) } - see locate_sn_table_record_bytes().
// max_h / max_wd: maxima over the table (grid sizing).  Four launches advance every layer.
LOCATE_API size_t locate_sn_table_record_bytes(void) { return sizeof(SnLayer); }

LOCATE_API int locate_sn_power_iter_batched(const void* table, int n_layers, int max_h, int max_wd, void* stream) {
    LOCATE_REQUIRE(table && n_layers > 0 && max_h > 0 && max_wd > 0, "locate_sn_power_iter_batched: bad arguments");
    hipStream_t st = as_stream(stream);
    const SnLayer* tab = static_cast<const SnLayer*>(table);
    SnLayer dummy = {};
    const int nstrip = (max_wd + SN_COLS - 1) / SN_COLS;
    sn_colsum_kernel<true><<<dim3(nstrip * sn_nchunk(max_h), n_layers), SN_COLS, 0, st>>>(dummy, tab);
    LOCATE_LAUNCH_CHECK("locate_sn_power_iter_batched(colsum)");
    sn_tsum_kernel<true><<<dim3(nstrip, n_layers), SN_COLS, 0, st>>>(dummy, tab);
    LOCATE_LAUNCH_CHECK("locate_sn_power_iter_batched(tsum)");
    sn_rowdot_kernel<true><<<dim3(max_wd >= SN_WIDE ? max_h : (max_h + 3) / 4, n_layers), 256, 0, st>>>(dummy, tab);
    LOCATE_LAUNCH_CHECK("locate_sn_power_iter_batched(rowdot)");
    sn_finalize_kernel<true><<<dim3(1, n_layers), 1024, 0, st>>>(dummy, tab);
    LOCATE_LAUNCH_CHECK("locate_sn_power_iter_batched(finalize)");
    return LOCATE_OK;
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// gw holds G / sigma (written by the weight-gradient pass, which also left the partial sums of <G, W_bar>):
//   dsigma = -(sum partial) / sigma^2 ;  gw += dsigma * u v^T ;  block 0: du = dsigma * wv, dsigma_out = dsigma
__global__ void __launch_bounds__(256) sn_rank1_kernel(const double* __restrict__ partial, int npartial,
                                                       const float* __restrict__ u, const float* __restrict__ v,
                                                       const float* __restrict__ sigma, const float* __restrict__ wv,
                                                       float* __restrict__ gw, float* __restrict__ du,
                                                       float* __restrict__ dsigma_out, int h, int wd) {
    __shared__ double scratch[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < npartial; i += blockDim.x) acc += partial[i];
    acc = block_sum<double>(acc, scratch);
    const float sg = sigma[0];
    const float dsg = (float)(-acc / ((double)sg * (double)sg));
    const int64_t n = (int64_t)h * wd;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / wd), c = (int)(i - (int64_t)r * wd);
        gw[i] = fmaf(dsg * u[r], v[c], gw[i]);
    }
    if (blockIdx.x == 0) {
        if (du)
            for (int i = threadIdx.x; i < h; i += blockDim.x) du[i] = dsg * wv[i];
        if (threadIdx.x == 0 && dsigma_out) dsigma_out[0] = dsg;
    }
}

// ---------------------------------------------------------------------------------------------
// Stacked forwards ("groups" along the batch, each with its own sigma_k): <G_k, W_bar> is taken on the activation
// side, <G_k, W_bar> = <gy_k, conv(x_k, W_bar)> = sigma_k <gy_k, y_k - bias>, so
//     dsigma_k = -<G_k, W_bar> / sigma_k^2 = -(1 / sigma_k) * sum_{b in k, m, p} gy[b,m,p] (y[b,m,p] - bias[m])
// ---------------------------------------------------------------------------------------------
#define SGD_CHUNK 4096      // elements per block visit
__global__ void __launch_bounds__(256) sn_group_dot_kernel(const float* __restrict__ gy, int64_t gy_bs, const float* __restrict__ y,
                                                           int64_t y_bs, const float* __restrict__ bias, int Bg, int M, int plane,
                                                           double* __restrict__ partial) {
    // One batch element is a contiguous run of M * plane values: blocks walk (batch element, 4096-element chunk) pairs
    // with 16-byte loads; the channel (for the bias) costs one 32-bit division per four elements.
    __shared__ double scratch[16];
    const int grp = blockIdx.y;
    const int per_b = M * plane;
    const int nchunk = (per_b + SGD_CHUNK - 1) / SGD_CHUNK;
    const int work = Bg * nchunk;
    const bool vec = (plane & 3) == 0 && (gy_bs & 3) == 0 && (y_bs & 3) == 0 && ((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    double acc = 0.0;
    for (int w = blockIdx.x; w < work; w += gridDim.x) {
        const int bl = w / nchunk, ch = w - bl * nchunk;
        const int64_t b = (int64_t)grp * Bg + bl;
        const float* gp = gy + b * gy_bs;
        const float* yp = y + b * y_bs;
        const int j0 = ch * SGD_CHUNK, j1 = min(j0 + SGD_CHUNK, per_b);
        float part = 0.0f;
        if (vec) {
            for (int j = j0 + 4 * (int)threadIdx.x; j < j1; j += 1024) {
                const float bv = bias ? bias[j / plane] : 0.0f;
                const float4 g4 = *reinterpret_cast<const float4*>(gp + j), y4 = *reinterpret_cast<const float4*>(yp + j);
                part += (g4.x * (y4.x - bv) + g4.y * (y4.y - bv)) + (g4.z * (y4.z - bv) + g4.w * (y4.w - bv));
            }
        } else {
            for (int j = j0 + (int)threadIdx.x; j < j1; j += 256) part = fmaf(gp[j], yp[j] - (bias ? bias[j / plane] : 0.0f), part);
        }
        acc += (double)part;
    }
    acc = block_sum<double>(acc, scratch);
    if (threadIdx.x == 0) partial[(int64_t)grp * gridDim.x + blockIdx.x] = acc;
}

// gw += (sum_k dsigma_k) u v^T;  block 0: du = sum_k dsigma_k wv_k, dsigma_total_out = sum_k dsigma_k
__global__ void __launch_bounds__(256) sn_rank1_grouped_kernel(const double* __restrict__ partial, int npartial, int groups,
                                                               const float* __restrict__ sigma_tab, int sigma_stride,
                                                               const float* __restrict__ u, const float* __restrict__ v,
                                                               const float* __restrict__ wv, int64_t wv_stride,
                                                               float* __restrict__ gw, float* __restrict__ du,
                                                               float* __restrict__ dsigma_total_out, int h, int wd) {
    __shared__ double scratch[16];
    float dsg[4] = {0.f, 0.f, 0.f, 0.f};
    float total = 0.0f;
    for (int k = 0; k < groups; ++k) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < npartial; i += blockDim.x) acc += partial[(int64_t)k * npartial + i];
        acc = block_sum<double>(acc, scratch);
        dsg[k] = (float)(-acc * (double)sigma_tab[k * sigma_stride + 1]);     // [k][1] = 1 / sigma_k
        total += dsg[k];
    }
    const int64_t n = (int64_t)h * wd;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / wd), c = (int)(i - (int64_t)r * wd);
        gw[i] = fmaf(total * u[r], v[c], gw[i]);
    }
    if (blockIdx.x == 0) {
        if (du)
            for (int i = threadIdx.x; i < h; i += blockDim.x) {
                float a = 0.0f;
                for (int k = 0; k < groups; ++k) a = fmaf(dsg[k], wv[(int64_t)k * wv_stride + i], a);
                du[i] = a;
            }
        if (threadIdx.x == 0 && dsigma_total_out) dsigma_total_out[0] = total;
    }
}

LOCATE_API size_t locate_sn_group_workspace_bytes(void) { return 4 * 256 * sizeof(double); }

// Spectral-norm backward for `groups` (<= 4) forwards stacked along the batch of one layer call.
//   gw (in/out): sum_k G_k / sigma_k (from locate_conv_wgrad with group scaling) -> dW_bar (rank-1 correction added)
//   gy / y: gradient and output of the layer call [groups*Bg, M, plane] (batch strides in elements), bias nullable
//   sigma_tab: {sigma_k, 1/sigma_k} pairs sigma_stride floats apart; wv: W v_k rows wv_stride floats apart
//   du (nullable) = sum_k dsigma_k W v_k;  dsigma_total_out (nullable) = sum_k dsigma_k
LOCATE_API int locate_sn_weight_bwd_grouped(const float* gy, int64_t gy_bs, const float* y, int64_t y_bs, const float* bias,
                                            int groups, int Bg, int M, int plane, const float* sigma_tab, int sigma_stride,
                                            const float* u, const float* v, const float* wv, int64_t wv_stride, float* gw,
                                            float* du, float* dsigma_total_out, int h, int wd, void* workspace, void* stream) {
    LOCATE_REQUIRE(gy && y && sigma_tab && u && v && gw && workspace && groups >= 1 && groups <= 4 && Bg > 0 && M > 0 && plane > 0,
                   "locate_sn_weight_bwd_grouped: bad arguments");
    LOCATE_REQUIRE(!du || wv, "locate_sn_weight_bwd_grouped: du requested without the saved W v");
    hipStream_t st = as_stream(stream);
    double* partial = static_cast<double*>(workspace);
    const int64_t work = (int64_t)Bg * (((int64_t)M * plane + SGD_CHUNK - 1) / SGD_CHUNK);
    LOCATE_REQUIRE((int64_t)M * plane < (1ll << 31), "locate_sn_weight_bwd_grouped: layer output too large");
    const int nb = work < 256 ? (int)work : 256;      // the workspace holds 4 x 256 partial sums
    sn_group_dot_kernel<<<dim3(nb, groups), 256, 0, st>>>(gy, gy_bs, y, y_bs, bias, Bg, M, plane, partial);
    LOCATE_LAUNCH_CHECK("locate_sn_weight_bwd_grouped(dot)");
    const int64_t n = (int64_t)h * wd;
    sn_rank1_grouped_kernel<<<stream_grid(n, 1024), 256, 0, st>>>(partial, nb, groups, sigma_tab, sigma_stride, u, v, wv, wv_stride,
                                                                 gw, du, dsigma_total_out, h, wd);
    LOCATE_LAUNCH_CHECK("locate_sn_weight_bwd_grouped(rank1)");
    return LOCATE_OK;
}

// dv[col] = (sum of the layer's dsigma slots) * t[col];  the slots are cleared for the next backward pass
__global__ void __launch_bounds__(SN_COLS) sn_dv_batched_kernel(const SnLayer* __restrict__ table) {
    const SnLayer L = table[blockIdx.y];
    const int col = blockIdx.x * SN_COLS + threadIdx.x;
    // record reuse: L.v = dv output, L.sigma = dsigma slots [4]
    const float total = (L.sigma[0] + L.sigma[1]) + (L.sigma[2] + L.sigma[3]);
    if (col < L.wd) L.v[col] = total * L.t[col];
}

__global__ void __launch_bounds__(64) sn_dsig_clear_kernel(const SnLayer* __restrict__ table, int n_layers) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_layers * 4) table[i >> 2].sigma[i & 3] = 0.0f;
}

// Spectral-norm backward after locate_conv_wgrad(..., w_ref = W_bar, inv_scale = 1/sigma, inner_partial):
//   gw (in/out) enters as G / sigma and leaves as dW_bar = G / sigma + dsigma u v^T,  dsigma = -<G, W_bar> / sigma^2,
//   du = dsigma (W v)_k (nullable), dsigma_out[0] = dsigma (nullable; feeds locate_sn_dv_batched).
//   sigma: the [2] buffer and wv the W v of the forward being differentiated; u, v: CURRENT (latest) state, as the
//   reference's autograd sees them.
LOCATE_API int locate_sn_weight_bwd(const double* inner_partial, int n_partial, const float* u, const float* v,
                                    const float* sigma, const float* wv, float* gw, float* du, float* dsigma_out, int h,
                                    int wd, void* stream) {
    LOCATE_REQUIRE(h > 0 && wd > 0 && inner_partial && n_partial > 0 && u && v && sigma && gw, "locate_sn_weight_bwd: bad arguments");
    LOCATE_REQUIRE(!du || wv, "locate_sn_weight_bwd: du requested without the saved W v");
    const int64_t n = (int64_t)h * wd;
    sn_rank1_kernel<<<stream_grid(n, 1024), 256, 0, as_stream(stream)>>>(inner_partial, n_partial, u, v, sigma, wv, gw, du,
                                                                         dsigma_out, h, wd);
    LOCATE_LAUNCH_CHECK("locate_sn_weight_bwd(rank1)");
    return LOCATE_OK;
}

// dv = (sum_k dsigma_k) * W^T u_latest for every layer of `table` in three launches (same record layout as
// locate_sn_power_iter_batched with two fields reused: v = dv output [wd], sigma = the layer's 4 dsigma slots, which
// are zeroed afterwards).  Called once at the end of a backward pass instead of three launches per layer and graph.
LOCATE_API int locate_sn_dv_batched(const void* table, int n_layers, int max_h, int max_wd, void* stream) {
    LOCATE_REQUIRE(table && n_layers > 0 && max_h > 0 && max_wd > 0, "locate_sn_dv_batched: bad arguments");
    hipStream_t st = as_stream(stream);
    const SnLayer* tab = static_cast<const SnLayer*>(table);
    SnLayer dummy = {};
    const int nstrip = (max_wd + SN_COLS - 1) / SN_COLS;
    sn_colsum_kernel<true><<<dim3(nstrip * sn_nchunk(max_h), n_layers), SN_COLS, 0, st>>>(dummy, tab);
    LOCATE_LAUNCH_CHECK("locate_sn_dv_batched(colsum)");
    sn_tsum_kernel<true><<<dim3(nstrip, n_layers), SN_COLS, 0, st>>>(dummy, tab);
    LOCATE_LAUNCH_CHECK("locate_sn_dv_batched(tsum)");
    sn_dv_batched_kernel<<<dim3(nstrip, n_layers), SN_COLS, 0, st>>>(tab);
    LOCATE_LAUNCH_CHECK("locate_sn_dv_batched(dv)");
    sn_dsig_clear_kernel<<<(n_layers * 4 + 63) / 64, 64, 0, st>>>(tab, n_layers);
    LOCATE_LAUNCH_CHECK("locate_sn_dv_batched(clear)");
    return LOCATE_OK;
}
