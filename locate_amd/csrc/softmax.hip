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

// rows of 1025 .. 4096 elements (the self-attention gate at 64x64: N = 4096), n % 4 == 0: the row lives in registers
// (V4 float4 per thread), so it is read once and every exponential is evaluated once
template <int V4>
__global__ void __launch_bounds__(256) softmax_fwd_row_kernel(const float* __restrict__ x, float* __restrict__ y, int n) {
    __shared__ float scratch[16];
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)blockIdx.x * n);
    float4* yr = reinterpret_cast<float4*>(y + (int64_t)blockIdx.x * n);
    const int n4 = n >> 2;
    float4 v[V4];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < V4; ++k) {
        const int i = threadIdx.x + 256 * k;
        v[k] = i < n4 ? xr[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        mx = fmaxf(fmaxf(mx, fmaxf(v[k].x, v[k].y)), fmaxf(v[k].z, v[k].w));
    }
    mx = block_max(mx, scratch);
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < V4; ++k) {
        v[k].x = __expf(v[k].x - mx); v[k].y = __expf(v[k].y - mx); v[k].z = __expf(v[k].z - mx); v[k].w = __expf(v[k].w - mx);
        sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    sum = block_sum<float>(sum, scratch);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int k = 0; k < V4; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < n4) yr[i] = make_float4(v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv);
    }
}

template <int V4>
__global__ void __launch_bounds__(256) softmax_bwd_row_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                              float* __restrict__ gx, int n) {
    __shared__ float scratch[16];
    const int64_t base = (int64_t)blockIdx.x * n;
    const float4* yr = reinterpret_cast<const float4*>(y + base);
    const float4* gr = reinterpret_cast<const float4*>(g + base);
    float4* xr = reinterpret_cast<float4*>(gx + base);
    const int n4 = n >> 2;
    float4 a[V4], b[V4];
    float dot = 0.0f;
#pragma unroll
    for (int k = 0; k < V4; ++k) {
        const int i = threadIdx.x + 256 * k;
        a[k] = i < n4 ? yr[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        b[k] = i < n4 ? gr[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        dot = fmaf(a[k].x, b[k].x, fmaf(a[k].y, b[k].y, fmaf(a[k].z, b[k].z, fmaf(a[k].w, b[k].w, dot))));
    }
    dot = block_sum<float>(dot, scratch);
#pragma unroll
    for (int k = 0; k < V4; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < n4) xr[i] = make_float4(a[k].x * (b[k].x - dot), a[k].y * (b[k].y - dot), a[k].z * (b[k].z - dot), a[k].w * (b[k].w - dot));
    }
}

static bool softmax_row_ok(const void* p0, const void* p1, const void* p2, int n) {
    return n > 1024 && n <= 4096 && (n & 3) == 0 &&
           (((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)(p2 ? p2 : p0)) & 15) == 0;
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
        if (softmax_row_ok(x, y, nullptr, n)) {
            if (n <= 2048) softmax_fwd_row_kernel<2><<<(int)rows, 256, 0, st>>>(x, y, n);
            else if (n <= 3072) softmax_fwd_row_kernel<3><<<(int)rows, 256, 0, st>>>(x, y, n);
            else softmax_fwd_row_kernel<4><<<(int)rows, 256, 0, st>>>(x, y, n);
        } else {
            softmax_fwd_block_kernel<<<(int)rows, 256, 0, st>>>(x, y, n);
        }
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
        if (softmax_row_ok(y, gy, gx, n)) {
            if (n <= 2048) softmax_bwd_row_kernel<2><<<(int)rows, 256, 0, st>>>(y, gy, gx, n);
            else if (n <= 3072) softmax_bwd_row_kernel<3><<<(int)rows, 256, 0, st>>>(y, gy, gx, n);
            else softmax_bwd_row_kernel<4><<<(int)rows, 256, 0, st>>>(y, gy, gx, n);
        } else {
            softmax_bwd_block_kernel<<<(int)rows, 256, 0, st>>>(y, gy, gx, n);
        }
    }
    LOCATE_LAUNCH_CHECK("locate_softmax_bwd");
    return LOCATE_OK;
}
