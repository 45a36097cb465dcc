// Probe: what limits the bf16x3 implicit-GEMM main loop (128x128x32 stage, 4 waves, 2x2 tiles of 32x32 per wave)?
//   stage 0: MFMAs only (operands in registers)                   stage 1: + LDS fragment reads (ds_read_b128)
//   stage 2: + barrier per stage                                   stage 3: + B conversion (fp32 -> hi/lo) + LDS writes
//   stage 4: + global loads per stage, `foot` bytes footprint per block (small = L2 resident, large = streaming)
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/bx3_probe.hip -o tools/bx3_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int STAGE>
__global__ void __launch_bounds__(256) probe_kernel(const float* __restrict__ src, float* __restrict__ out, int steps, long long foot) {
    __shared__ uint4 Ah[2][4][128], Al[2][4][128], Bh[2][4][128], Bl[2][4][128];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, lrow = lane >> 5, lcol = lane & 31;
    for (int i = tid; i < 2 * 4 * 128; i += 256) {
        const uint4 v = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3c003c00u, 0x3c003c00u);
        (&Ah[0][0][0])[i] = v; (&Al[0][0][0])[i] = v; (&Bh[0][0][0])[i] = v; (&Bl[0][0][0])[i] = v;
    }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    bf16x8 ah[2], al[2], bh[2], bl[2];
    for (int i = 0; i < 2; ++i)
        for (int e = 0; e < 8; ++e) { ah[i][e] = (__bf16)(1.0f + lane); al[i][e] = (__bf16)0.5f; bh[i][e] = (__bf16)0.25f; bl[i][e] = (__bf16)(0.125f + lane); }
    uint4 areg[4];
    float breg[2][8];
    const long long fmask = foot / 4 - 1;     // floats, power of two
    const float* base = src + (long long)blockIdx.x * (foot / 4);
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (STAGE >= 4) {
            const long long o = ((long long)s * 8192) & fmask;
#pragma unroll
            for (int i = 0; i < 4; ++i) areg[i] = *reinterpret_cast<const uint4*>(base + ((o + i * 1024 + tid * 4) & fmask));
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int j = 0; j < 8; ++j) breg[f][j] = base[(o + 4096 + (f * 8 + j) * 256 + tid) & fmask];
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (STAGE >= 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = *reinterpret_cast<const bf16x8*>(&Ah[buf][kk * 2 + lrow][(wm * 2 + i) * 32 + lcol]);
                    al[i] = *reinterpret_cast<const bf16x8*>(&Al[buf][kk * 2 + lrow][(wm * 2 + i) * 32 + lcol]);
                    bh[i] = *reinterpret_cast<const bf16x8*>(&Bh[buf][kk * 2 + lrow][(wn * 2 + i) * 32 + lcol]);
                    bl[i] = *reinterpret_cast<const bf16x8*>(&Bl[buf][kk * 2 + lrow][(wn * 2 + i) * 32 + lcol]);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (STAGE >= 3) {
            if (STAGE < 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) areg[i] = make_uint4(s, tid, s + 1, tid + 1);
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int j = 0; j < 8; ++j) breg[f][j] = (float)(s + j) * 0.001f;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * 256;
                Ah[buf ^ 1][idx >> 7][idx & 127] = areg[i];
                Al[buf ^ 1][idx >> 7][idx & 127] = areg[2 + i];
            }
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                bf16x8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    hi[j] = (__bf16)breg[f][j];
                    lo[j] = (__bf16)(breg[f][j] - (float)hi[j]);
                }
                Bh[buf ^ 1][(tid >> 7) * 2 + f][tid & 127] = *reinterpret_cast<uint4*>(&hi);
                Bl[buf ^ 1][(tid >> 7) * 2 + f][tid & 127] = *reinterpret_cast<uint4*>(&lo);
            }
        }
        if (STAGE >= 2) __syncthreads();
    }
    float t = 0.0f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + tid] = t;
}

extern "C" int probe_launch(int stage, const float* src, float* out, int blocks, int steps, long long foot, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (stage) {
        case 0: probe_kernel<0><<<blocks, 256, 0, st>>>(src, out, steps, foot); break;
        case 1: probe_kernel<1><<<blocks, 256, 0, st>>>(src, out, steps, foot); break;
        case 2: probe_kernel<2><<<blocks, 256, 0, st>>>(src, out, steps, foot); break;
        case 3: probe_kernel<3><<<blocks, 256, 0, st>>>(src, out, steps, foot); break;
        default: probe_kernel<4><<<blocks, 256, 0, st>>>(src, out, steps, foot); break;
    }
    return (int)hipGetLastError();
}
