// Probe: what limits the fp32-MFMA implicit-GEMM main loop?  Builds the loop up piece by piece.
//   stage 0: MFMAs only (operands in registers)
//   stage 1: + LDS fragment reads (ds_read) per k-pair
//   stage 2: + one __syncthreads per K step (16 k = 8 k-pairs = 32 MFMAs per wave)
//   stage 3: + LDS tile writes per K step (from registers)
//   stage 4: + global loads per K step (coalesced), written to LDS
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/mfma_probe.hip -o tools/mfma_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int STAGE>
__global__ void __launch_bounds__(256) probe_kernel(const float* __restrict__ src, float* __restrict__ out, int steps) {
    __shared__ __attribute__((aligned(16))) float As[2][16][128];
    __shared__ __attribute__((aligned(16))) float Bs[2][16][128];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, lrow = lane >> 5, lcol = lane & 31;
    for (int i = tid; i < 2 * 16 * 128; i += 256) {
        (&As[0][0][0])[i] = 0.001f * (float)(i & 255);
        (&Bs[0][0][0])[i] = 0.002f * (float)(i & 127);
    }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float a[2] = {1.0f + lane, 2.0f}, b[2] = {0.5f, 0.25f + lane};
    float4 areg[2];
    float breg[8];
    const float* gp = src + (size_t)blockIdx.x * 4096 + tid * 4;
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (STAGE >= 4) {
            areg[0] = *reinterpret_cast<const float4*>(gp + (size_t)(s & 63) * 1048576);
            areg[1] = *reinterpret_cast<const float4*>(gp + (size_t)(s & 63) * 1048576 + 1024);
#pragma unroll
            for (int j = 0; j < 8; ++j) breg[j] = gp[(size_t)(s & 63) * 1048576 + 2048 + j * 256];
        }
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            if (STAGE >= 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = As[buf][k2 * 2 + lrow][(wm * 2 + i) * 32 + lcol];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = Bs[buf][k2 * 2 + lrow][(wn * 2 + j) * 32 + lcol];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (STAGE >= 3) {
            if (STAGE < 4) {
                areg[0] = make_float4(a[0], a[1], b[0], b[1]);
                areg[1] = areg[0];
#pragma unroll
                for (int j = 0; j < 8; ++j) breg[j] = b[j & 1];
            }
            const int row = tid >> 5, c4 = tid & 31;
            *reinterpret_cast<float4*>(&As[buf ^ 1][row][c4 * 4]) = areg[0];
            *reinterpret_cast<float4*>(&As[buf ^ 1][row + 8][c4 * 4]) = areg[1];
#pragma unroll
            for (int j = 0; j < 8; ++j) Bs[buf ^ 1][(tid >> 7) * 8 + j][tid & 127] = breg[j];
        }
        if (STAGE >= 2) __syncthreads();
    }
    float t = 0.0f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + tid] = t;
}

extern "C" int probe_launch(int stage, const float* src, float* out, int blocks, int steps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (stage) {
        case 0: probe_kernel<0><<<blocks, 256, 0, st>>>(src, out, steps); break;
        case 1: probe_kernel<1><<<blocks, 256, 0, st>>>(src, out, steps); break;
        case 2: probe_kernel<2><<<blocks, 256, 0, st>>>(src, out, steps); break;
        case 3: probe_kernel<3><<<blocks, 256, 0, st>>>(src, out, steps); break;
        default: probe_kernel<4><<<blocks, 256, 0, st>>>(src, out, steps); break;
    }
    return (int)hipGetLastError();
}
