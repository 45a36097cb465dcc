// Shared helpers for the gfx950 (MI355X / CDNA4) kernels of the LocAtE hot path.
// Wave = 64 lanes; every block size in this library is a multiple of 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define LOCATE_OK 0
#define LOCATE_ERR_ARG 1
#define LOCATE_ERR_LAUNCH 2

#define LOCATE_API extern "C" __attribute__((visibility("default")))

void locate_set_error(const char* fmt, ...);

#define LOCATE_REQUIRE(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            locate_set_error(__VA_ARGS__);        \
            return LOCATE_ERR_ARG;                \
        }                                         \
    } while (0)

#define LOCATE_LAUNCH_CHECK(name)                                                   \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            locate_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return LOCATE_ERR_LAUNCH;                                               \
        }                                                                           \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// memory-bound kernels: cap the grid and grid-stride the rest (256 CUs x 8 blocks)
static inline int stream_grid(int64_t work_items, int per_block) {
    int64_t g = cdiv64(work_items, per_block);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

#ifdef __HIPCC__
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// A tensor's largest magnitude for the contractions' fp16-piece scaling (conv.hip): every producing block folds its maximum into
// a small VECTOR of device words with an atomic max on the bit pattern (non-negative floats order like unsigned integers, and a
// maximum does not depend on the order of arrival: deterministic); the consumer takes the maximum of the words.  One word would
// do, but atomics on one CACHE LINE serialise at ~90 per microsecond - a 2048-block element-wise launch whose blocks all
// finish together paid 12-20 us for it, more than the kernel itself - so block b goes to word AMAX_STRIDE * (b % AMAX_LINES):
// 32 words on 32 different 128-byte lines (<= 64 arrivals each), AMAX_WORDS = 1024 words (4 KiB) per tensor, of which only those
// 32 are ever touched.  The words must be zero before the producer starts.  `scratch`: >= 16 floats.
#define AMAX_LINES 32
#define AMAX_STRIDE 32
#define AMAX_WORDS (AMAX_LINES * AMAX_STRIDE)
__device__ __forceinline__ void absmax_publish(float m, float* scratch, unsigned* slot) {
    m = wave_max(m);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[wid] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = scratch[0];
        for (int i = 1; i < nw; ++i) t = fmaxf(t, scratch[i]);
        unsigned* word = slot + AMAX_STRIDE * ((blockIdx.x + blockIdx.y * 7u) & (AMAX_LINES - 1));
        const unsigned bits = __float_as_uint(t);
        if (bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            (void)__hip_atomic_fetch_max(word, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // result unused: no return trip
    }
}
// the same from a kernel whose waves finish on their own (the contractions' fused epilogues): one atomic per wave, no block barrier
__device__ __forceinline__ void absmax_publish_wave(float m, unsigned* slot) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) {
        const unsigned lin = blockIdx.x + blockIdx.y * 7u + blockIdx.z * 13u + (threadIdx.x >> 6) * 5u;
        unsigned* word = slot + AMAX_STRIDE * (lin & (AMAX_LINES - 1));
        const unsigned bits = __float_as_uint(m);
        if (bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            (void)__hip_atomic_fetch_max(word, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// the consumer's side: maximum of the AMAX_LINES words (wave-uniform)
__device__ __forceinline__ unsigned absmax_read(const unsigned* slot) {
    unsigned v = slot[AMAX_STRIDE * (threadIdx.x & (AMAX_LINES - 1))];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned w = (unsigned)__shfl_xor((int)v, o, 64);
        v = w > v ? w : v;
    }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

// Block-wide sum for blockDim.x <= 1024 (a multiple of 64). `scratch` holds >= 16 values.
// Every thread returns the total.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();  // scratch may still be read from a previous call
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    T t = 0;
    for (int i = 0; i < nw; ++i) t += scratch[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float t = scratch[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, scratch[i]);
    return t;
}
// Sum of channel c over the batch slice [b0, b0 + nb) of g [B, C, hw] by ONE wave, for slices of at most 1024 (16-byte) elements:
// the arithmetic of a 1024-thread block in which every thread holds at most one element - a butterfly per 64 elements, the sixteen
// wave results added in order (block_sum) - so a kernel that gives such a slice to a wave instead of a block returns the same bits.
// (Bias gradients on 1x1 ... 4x4 maps: 64 ... 768 elements per channel; a block of 1024 threads per channel was 4 400 blocks of
// mostly idle lanes and two barriers each.)
__device__ __forceinline__ float channel_sum_wave(const float* __restrict__ g, int64_t batch_stride, int c, int hw, int b0, int nb,
                                                  bool vec, int lane) {
    float total = 0.0f;
    if (vec) {
        const int hw4 = hw >> 2, n = nb * hw4;
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int i = k0 + lane;
            float v = 0.0f;
            if (i < n) {
                const int b = i / hw4, r = i - b * hw4;
                const float4 q = reinterpret_cast<const float4*>(g + (int64_t)(b0 + b) * batch_stride + (int64_t)c * hw)[r];
                v = 0.0f + ((q.x + q.y) + (q.z + q.w));
            }
            total += wave_sum(v);
        }
    } else {
        const int n = nb * hw;
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int i = k0 + lane;
            float v = 0.0f;
            if (i < n) {
                const int b = i / hw, r = i - b * hw;
                v = 0.0f + g[(int64_t)(b0 + b) * batch_stride + (int64_t)c * hw + r];
            }
            total += wave_sum(v);
        }
    }
    return total;
}
// whether a slice of nb batch elements of one channel takes that form
__host__ __device__ __forceinline__ bool channel_sum_small(int nb, int hw, bool vec) { return (long long)nb * (vec ? hw >> 2 : hw) <= 1024; }

// q = i / d, r = i % d for 32-bit unsigned operands: shift/mask when d is a power of two (every spatial size and most
// channel counts of this model), one 32-bit division otherwise.  The kernels' flat element indices stay below 2^31
// (checked by the host entries), so none of them needs the ~100-instruction 64-bit division in its inner loop.
struct DivU32 {
    unsigned d, mask;
    int shift;          // >= 0: d == 1 << shift
    __device__ __forceinline__ explicit DivU32(unsigned dd) : d(dd), mask(dd - 1), shift((dd & (dd - 1)) == 0 ? __ffs((int)dd) - 1 : -1) {}
    __device__ __forceinline__ void divmod(unsigned i, unsigned& q, unsigned& r) const {
        if (shift >= 0) { q = i >> shift; r = i & mask; }
        else { q = i / d; r = i - q * d; }
    }
    __device__ __forceinline__ unsigned div(unsigned i) const { return shift >= 0 ? i >> shift : i / d; }
};

// RootTanh pieces shared by the element-wise kernels and the fused norm kernels (see elementwise.hip).
// These kernels are ALU-bound with libm's expm1f / IEEE sqrt and division (~75 instructions per element against 8-12
// bytes of traffic), so they use the hardware approximations (v_exp_f32, v_rcp_f32, v_sqrt_f32: 1 ulp each) and a short
// series where exp(t) - 1 would cancel: ~30 instructions, errors of a few 1e-7 relative (tests: 2e-6 against fp64).
// Every operation is spelled out (fmaf / __fmul_rn / __fadd_rn): with the compiler free to contract a * b + c its own way per
// call site, the same formula inlined into two kernels (an element-wise launch, a contraction's fused epilogue) rounds
// differently - and the fused and the separate form of an activation are held to the same bits (tests/test_gpu_ops.py).
__device__ __forceinline__ float expm1_nonpos(float t) {          // t <= 0
    float s = fmaf(t, 0.0001984127f, 0.0013888889f);
    s = fmaf(t, s, 0.0083333338f);
    s = fmaf(t, s, 0.041666668f);
    s = fmaf(t, s, 0.16666667f);
    s = fmaf(t, s, 0.5f);
    s = fmaf(t, s, 1.0f);
    const float series = __fmul_rn(t, s);          // |t| <= 0.35: truncation < 2e-8 |t|
    const float direct = __fsub_rn(__builtin_amdgcn_exp2f(__fmul_rn(t, 1.44269504f)), 1.0f);
    return t > -0.35f ? series : direct;
}

__device__ __forceinline__ void tanh_sech2(float x, float& th, float& sech2) {
    const float ax = fabsf(x);
    const float em = expm1_nonpos(__fmul_rn(-2.0f, ax));
    const float r = __builtin_amdgcn_rcpf(__fadd_rn(2.0f, em));
    th = copysignf(__fmul_rn(-em, r), x);
    sech2 = __fmul_rn(__fmul_rn(__fmul_rn(4.0f, __fadd_rn(1.0f, em)), r), r);
}

__device__ __forceinline__ float roottanh_f(float x) {
    float th, s2;
    tanh_sech2(x, th, s2);
    return __fmul_rn(__builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(fmaf(x, x, 1.0f))), th);
}

__device__ __forceinline__ float roottanh_grad_f(float x, float g) {
    float th, s2;
    tanh_sech2(x, th, s2);
    const float q = fmaf(x, x, 1.0f);
    const float r = __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(q));          // q^(1/4)
    const float q34 = __fmul_rn(__fmul_rn(r, r), r);              // q^(3/4)
    const float inner = fmaf(x, th, __fmul_rn(__fmul_rn(2.0f, q), s2));
    return __fmul_rn(__fmul_rn(g, inner), __fmul_rn(0.5f, __builtin_amdgcn_rcpf(q34)));
}

#endif
