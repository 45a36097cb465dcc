// Gradient bucket pack / unpack for the data-parallel exchange (locate_amd/parallel.py): the ~140 gradient tensors of a
// network are copied into a few flat buckets before the RCCL all-reduce and back (scaled by 1 / world when the collective
// sums instead of averaging) afterwards - one launch per bucket and direction, driven by a device table, instead of an
// ATen multi-tensor copy.  The reference has no counterpart (single process, single device: libs/config.py:10-11).
#include "common.h"

struct CopyTensor {
    float* grad;        // the parameter's gradient
    float* flat;        // its place inside the bucket
    long long n;
};

#define MC_CHUNK 4096

// direction 0: flat <- grad (pack);  1: grad <- flat * scale (unpack)
__global__ void __launch_bounds__(256) multi_copy_kernel(const CopyTensor* __restrict__ tensors, const int2* __restrict__ chunks,
                                                         int direction, float scale) {
    const int2 ch = chunks[blockIdx.x];
    const CopyTensor T = tensors[ch.x];
    const long long begin = (long long)ch.y * MC_CHUNK;
    long long end = begin + MC_CHUNK;
    if (end > T.n) end = T.n;
    const float* __restrict__ src = direction == 0 ? T.grad : T.flat;
    float* __restrict__ dst = direction == 0 ? T.flat : T.grad;
    const bool vec = ((reinterpret_cast<uintptr_t>(src + begin) | reinterpret_cast<uintptr_t>(dst + begin)) & 15) == 0;
    if (vec) {
        const long long n4 = (end - begin) >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(src + begin);
        float4* d4 = reinterpret_cast<float4*>(dst + begin);
        for (long long i = threadIdx.x; i < n4; i += blockDim.x) {
            float4 v = s4[i];
            if (direction == 1) { v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; }
            d4[i] = v;
        }
        for (long long i = begin + (n4 << 2) + threadIdx.x; i < end; i += blockDim.x) dst[i] = direction == 1 ? src[i] * scale : src[i];
        return;
    }
    for (long long i = begin + threadIdx.x; i < end; i += blockDim.x) dst[i] = direction == 1 ? src[i] * scale : src[i];
}

LOCATE_API size_t locate_multi_copy_record_bytes(void) { return sizeof(CopyTensor); }
LOCATE_API int locate_multi_copy_chunk_elems(void) { return MC_CHUNK; }

// tensors: DEVICE array of records {grad, flat, n}; chunks: DEVICE array of n_chunks (tensor index, chunk index) int pairs
// covering every tensor in locate_multi_copy_chunk_elems() pieces.  direction 0 packs (flat <- grad), 1 unpacks
// (grad <- flat * scale; scale = 1 for a collective that already averaged).
LOCATE_API int locate_multi_copy(const void* tensors, const void* chunks, int n_chunks, int direction, float scale, void* stream) {
    LOCATE_REQUIRE(tensors && chunks && n_chunks > 0 && (direction == 0 || direction == 1), "locate_multi_copy: bad arguments");
    multi_copy_kernel<<<n_chunks, 256, 0, as_stream(stream)>>>(static_cast<const CopyTensor*>(tensors), static_cast<const int2*>(chunks),
                                                              direction, scale);
    LOCATE_LAUNCH_CHECK("locate_multi_copy");
    return LOCATE_OK;
}
