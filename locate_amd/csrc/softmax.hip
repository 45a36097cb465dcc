// Row softmax, forward and backward.  Used for
//   * SelfAttention: softmax over the N = H*W positions of every (batch, channel) row
//     (reference libs/attention.py:47, rows of 64 ... 65 536);
//   * feature attention: softmax over channels of a [B, C, 1, 1] tensor (attention.py:35) = rows of C.
// Short rows (n <= 1024): one wave per row, the row lives in registers, reductions by wave shuffles.
// Long rows: one 256-thread block per row, three streaming passes (the row stays in L2 between passes;
// a 65 536-element row is 256 KiB, more than the 160 KiB LDS, so it is not staged there).
#include "common.h"

template <int EPT>  // elements per lane; n <= 64 * EPT
__global__ void __launch_bounds__(256) softmax_fwd_wave_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int64_t rows, int n) {
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= rows) return;  // whole wave exits together
    const float* xr = x + row * n;
    float* yr = y + row * n;
    float v[EPT];
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = e * 64 + lane;
        v[e] = i < n ? xr[i] : -INFINITY;
        mx = fmaxf(mx, v[e]);
    }
    mx = wave_max(mx);
    float sum = 0.0f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        v[e] = __expf(v[e] - mx);   // exp(-inf) = 0 for the padding lanes
        sum += v[e];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = e * 64 + lane;
        if (i < n) yr[i] = v[e] * inv;
    }
}

template <int EPT>
__global__ void __launch_bounds__(256) softmax_bwd_wave_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                               float* __restrict__ gx, int64_t rows, int n) {
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= rows) return;
    const float* yr = y + row * n;
    const float* gr = g + row * n;
    float yv[EPT], gv[EPT];
    float dot = 0.0f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = e * 64 + lane;
        yv[e] = i < n ? yr[i] : 0.0f;
        gv[e] = i < n ? gr[i] : 0.0f;
        dot = fmaf(yv[e], gv[e], dot);
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = e * 64 + lane;
        if (i < n) gx[row * n + i] = yv[e] * (gv[e] - dot);
    }
}

__global__ void __launch_bounds__(256) softmax_fwd_block_kernel(const float* __restrict__ x, float* __restrict__ y, int n) {
    __shared__ float scratch[16];
    const float* xr = x + (int64_t)blockIdx.x * n;
    float* yr = y + (int64_t)blockIdx.x * n;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) mx = fmaxf(mx, xr[i]);
    mx = block_max(mx, scratch);
    float sum = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) sum += __expf(xr[i] - mx);
    sum = block_sum<float>(sum, scratch);
    const float inv = 1.0f / sum;
    for (int i = threadIdx.x; i < n; i += blockDim.x) yr[i] = __expf(xr[i] - mx) * inv;
}

__global__ void __launch_bounds__(256) softmax_bwd_block_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                                float* __restrict__ gx, int n) {
    __shared__ float scratch[16];
    const int64_t base = (int64_t)blockIdx.x * n;
    float dot = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) dot = fmaf(y[base + i], g[base + i], dot);
    dot = block_sum<float>(dot, scratch);
    for (int i = threadIdx.x; i < n; i += blockDim.x) gx[base + i] = y[base + i] * (g[base + i] - dot);
}

LOCATE_API int locate_softmax_fwd(const float* x, float* y, int64_t rows, int n, void* stream) {
    LOCATE_REQUIRE(rows > 0 && n > 0, "locate_softmax_fwd: bad shape");
    hipStream_t st = as_stream(stream);
    const int grid_w = (int)cdiv64(rows, 4);
    if (n <= 64) softmax_fwd_wave_kernel<1><<<grid_w, 256, 0, st>>>(x, y, rows, n);
    else if (n <= 256) softmax_fwd_wave_kernel<4><<<grid_w, 256, 0, st>>>(x, y, rows, n);
    else if (n <= 1024) softmax_fwd_wave_kernel<16><<<grid_w, 256, 0, st>>>(x, y, rows, n);
    else {
        LOCATE_REQUIRE(rows <= 0x7fffffff, "locate_softmax_fwd: too many rows");
        softmax_fwd_block_kernel<<<(int)rows, 256, 0, st>>>(x, y, n);
    }
    LOCATE_LAUNCH_CHECK("locate_softmax_fwd");
    return LOCATE_OK;
}

LOCATE_API int locate_softmax_bwd(const float* y, const float* gy, float* gx, int64_t rows, int n, void* stream) {
    LOCATE_REQUIRE(rows > 0 && n > 0, "locate_softmax_bwd: bad shape");
    hipStream_t st = as_stream(stream);
    const int grid_w = (int)cdiv64(rows, 4);
    if (n <= 64) softmax_bwd_wave_kernel<1><<<grid_w, 256, 0, st>>>(y, gy, gx, rows, n);
    else if (n <= 256) softmax_bwd_wave_kernel<4><<<grid_w, 256, 0, st>>>(y, gy, gx, rows, n);
    else if (n <= 1024) softmax_bwd_wave_kernel<16><<<grid_w, 256, 0, st>>>(y, gy, gx, rows, n);
    else {
        LOCATE_REQUIRE(rows <= 0x7fffffff, "locate_softmax_bwd: too many rows");
        softmax_bwd_block_kernel<<<(int)rows, 256, 0, st>>>(y, gy, gx, n);
    }
    LOCATE_LAUNCH_CHECK("locate_softmax_bwd");
    return LOCATE_OK;
}
