// Fused multi-tensor Nadam step (reference libs/nadam.py:31-89; hyper-parameters libs/config.py:70-73).
// The reference loops over ~100 parameter tensors in Python with ~10 ATen calls each; here one tiny
// "schedule" kernel advances every tensor's (step, m_schedule) state ON DEVICE (so the whole optimizer step
// can live inside a captured hipGraph) and one streaming kernel updates all tensors.
//
// Per tensor (state is per tensor because tensors whose gradient is None are skipped, nadam.py:44-45):
//   t <- t + 1
//   mc_t   = b1 (1 - 0.5 * 0.96^(t sd)),  mc_t1 = b1 (1 - 0.5 * 0.96^((t+1) sd))
//   ms_new = m_schedule * mc_t,  ms_next = ms_new * mc_t1,  m_schedule <- ms_new
//   g <- g + weight_decay p   (nadam.py:65-66; 0 in the reference's training loop)
//   m <- b1 m + (1-b1) g ;  v <- b2 v + (1-b2) g^2 ;  denom = sqrt(v / (1 - b2^t)) + eps
//   p <- p - lr (1-mc_t)/(1-ms_new) * g / denom ;  p <- p - lr mc_t1/(1-ms_next) * m / denom
#include "common.h"

struct NadamTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    double* sched;     // [2]: step count, m_schedule
    long long n;
    unsigned* absmax;  // nullable: AMAX_WORDS words that receive the largest magnitude of the UPDATED tensor (bit pattern,
                       // spread like every other absmax slot, common.h) - what the fp16-piece weight panels are scaled by;
                       // re-packing them then needs no pass of its own over the weights (conv.hip, direct packing)
};

struct NadamCoef {
    float c_grad, c_mom, bias2, pad;
};

#define NADAM_CHUNK 4096

__global__ void __launch_bounds__(64) nadam_schedule_kernel(const NadamTensor* __restrict__ tensors, NadamCoef* __restrict__ coef,
                                                            int n_tensors, double lr, double b1, double b2, double sd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tensors) return;
    double* st = tensors[i].sched;
    const double t = st[0] + 1.0;
    const double ms = st[0] == 0.0 ? 1.0 : st[1];
    const double mc_t = b1 * (1.0 - 0.5 * pow(0.96, t * sd));
    const double mc_t1 = b1 * (1.0 - 0.5 * pow(0.96, (t + 1.0) * sd));
    const double ms_new = ms * mc_t;
    const double ms_next = ms_new * mc_t1;
    st[0] = t;
    st[1] = ms_new;
    if (tensors[i].absmax)
        for (int k = 0; k < AMAX_LINES; ++k) tensors[i].absmax[k * AMAX_STRIDE] = 0u;      // the update kernel folds the new maximum in
    NadamCoef c;
    c.c_grad = (float)(-lr * (1.0 - mc_t) / (1.0 - ms_new));
    c.c_mom = (float)(-lr * mc_t1 / (1.0 - ms_next));
    c.bias2 = (float)(1.0 - pow(b2, t));
    c.pad = 0.0f;
    coef[i] = c;
}

__global__ void __launch_bounds__(256) nadam_update_kernel(const NadamTensor* __restrict__ tensors,
                                                           const NadamCoef* __restrict__ coef,
                                                           const int2* __restrict__ chunks, float b1, float b2, float eps, float wd) {
    const int2 ch = chunks[blockIdx.x];
    const NadamTensor T = tensors[ch.x];
    const NadamCoef c = coef[ch.x];
    const long long begin = (long long)ch.y * NADAM_CHUNK;
    long long end = begin + NADAM_CHUNK;
    if (end > T.n) end = T.n;
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    __shared__ float scratch[16];
    float amax = 0.0f;
    for (long long i = begin + threadIdx.x; i < end; i += blockDim.x) {
        float p = T.p[i];
        const float g = wd != 0.0f ? fmaf(wd, p, T.g[i]) : T.g[i];          // nadam.py:65-66: grad + weight_decay * p
        float m = T.m[i] * b1;
        m = m + omb1 * g;
        float v = T.v[i] * b2;
        v = v + omb2 * g * g;
        const float denom = sqrtf(v / c.bias2) + eps;
        p = p + c.c_grad * (g / denom);
        p = p + c.c_mom * (m / denom);
        T.m[i] = m;
        T.v[i] = v;
        T.p[i] = p;
        amax = fmaxf(amax, fabsf(p));
    }
    if (T.absmax) absmax_publish(amax, scratch, T.absmax);          // wave-uniform condition: every thread of the block takes it
}

LOCATE_API size_t locate_nadam_tensor_record_bytes(void) { return sizeof(NadamTensor); }
LOCATE_API int locate_nadam_chunk_elems(void) { return NADAM_CHUNK; }

// tensors: DEVICE array of n_tensors records {p, g, m, v, sched, n, absmax (nullable)}; coef: DEVICE scratch of n_tensors * 16 bytes;
// chunks: DEVICE array of n_chunks (tensor index, chunk index) int pairs covering every tensor in
// locate_nadam_chunk_elems() pieces.
LOCATE_API int locate_nadam_step(const void* tensors, void* coef, const void* chunks, int n_tensors, int n_chunks, double lr,
                                 double beta1, double beta2, double eps, double schedule_decay, double weight_decay, void* stream) {
    LOCATE_REQUIRE(tensors && coef && chunks && n_tensors > 0 && n_chunks > 0, "locate_nadam_step: bad arguments");
    hipStream_t st = as_stream(stream);
    nadam_schedule_kernel<<<(n_tensors + 63) / 64, 64, 0, st>>>(static_cast<const NadamTensor*>(tensors),
                                                               static_cast<NadamCoef*>(coef), n_tensors, lr, beta1, beta2,
                                                               schedule_decay);
    LOCATE_LAUNCH_CHECK("locate_nadam_step(schedule)");
    nadam_update_kernel<<<n_chunks, 256, 0, st>>>(static_cast<const NadamTensor*>(tensors), static_cast<const NadamCoef*>(coef),
                                                  static_cast<const int2*>(chunks), (float)beta1, (float)beta2, (float)eps,
                                                  (float)weight_decay);
    LOCATE_LAUNCH_CHECK("locate_nadam_step(update)");
    return LOCATE_OK;
}
