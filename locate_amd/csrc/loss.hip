// Loss glue of the training step (reference main.py:149-156,164-169, libs/utils.py:133-134,
// libs/grad_penalty.py:1-2): scalars from [B] vectors, one small block.  Values AND the gradients with
// respect to the discriminator outputs come out of the same launch, so the step driver can start the
// backward passes directly from them.
//   D-step:  d_error = mean( relu(1 - t_i) + relu(1 + f_i) )        t = D(real), f = D(G(z).detach())
//            penalty = gamma * (mean t - mean a)^2                   a = D(augmented real)   (not a gradient penalty)
//   G-step:  g_error = mean( relu(1 - f_i) )
// clamp(min=0) passes the gradient where its argument is >= 0 (ATen clamp backward mask).
#include "common.h"

__global__ void __launch_bounds__(256) d_loss_kernel(const float* __restrict__ t, const float* __restrict__ f,
                                                     const float* __restrict__ a, int B, float gamma, float* __restrict__ losses,
                                                     float* __restrict__ gt, float* __restrict__ gf, float* __restrict__ ga) {
    __shared__ double scratch[16];
    double hs = 0.0, ts = 0.0, as = 0.0;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        hs += (double)fmaxf(1.0f - t[i], 0.0f) + (double)fmaxf(1.0f + f[i], 0.0f);
        ts += t[i];
        as += a[i];
    }
    hs = block_sum<double>(hs, scratch);
    ts = block_sum<double>(ts, scratch);
    as = block_sum<double>(as, scratch);
    const float invB = 1.0f / (float)B;
    const float delta = (float)(ts / B) - (float)(as / B);
    const float pen_g = 2.0f * gamma * delta * invB;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        gt[i] = ((1.0f - t[i]) >= 0.0f ? -invB : 0.0f) + pen_g;
        gf[i] = (1.0f + f[i]) >= 0.0f ? invB : 0.0f;
        ga[i] = -pen_g;
    }
    if (threadIdx.x == 0) {
        const float d_error = (float)(hs / B);
        const float pen = gamma * delta * delta;
        losses[0] = d_error;
        losses[1] = pen;
        losses[2] = d_error + pen;
    }
}

__global__ void __launch_bounds__(256) g_loss_kernel(const float* __restrict__ f, int B, float* __restrict__ loss,
                                                     float* __restrict__ gf) {
    __shared__ double scratch[16];
    double hs = 0.0;
    const float invB = 1.0f / (float)B;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        hs += (double)fmaxf(1.0f - f[i], 0.0f);
        gf[i] = (1.0f - f[i]) >= 0.0f ? -invB : 0.0f;
    }
    hs = block_sum<double>(hs, scratch);
    if (threadIdx.x == 0) loss[0] = (float)(hs / B);
}

// losses: 3 floats {d_error, penalty, d_error + penalty};  g_*: [B] gradients of the total w.r.t. each D output
LOCATE_API int locate_d_loss(const float* d_true, const float* d_fake, const float* d_aug, int B, float gamma, float* losses,
                             float* g_true, float* g_fake, float* g_aug, void* stream) {
    LOCATE_REQUIRE(B > 0 && d_true && d_fake && d_aug && losses && g_true && g_fake && g_aug, "locate_d_loss: bad arguments");
    d_loss_kernel<<<1, 256, 0, as_stream(stream)>>>(d_true, d_fake, d_aug, B, gamma, losses, g_true, g_fake, g_aug);
    LOCATE_LAUNCH_CHECK("locate_d_loss");
    return LOCATE_OK;
}

LOCATE_API int locate_g_loss(const float* d_fake, int B, float* loss, float* g_fake, void* stream) {
    LOCATE_REQUIRE(B > 0 && d_fake && loss && g_fake, "locate_g_loss: bad arguments");
    g_loss_kernel<<<1, 256, 0, as_stream(stream)>>>(d_fake, B, loss, g_fake);
    LOCATE_LAUNCH_CHECK("locate_g_loss");
    return LOCATE_OK;
}
