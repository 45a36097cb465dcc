// fp8 contractions (BASELINE.json configs[4]: "fp8 weights + activations on CDNA4 fp8 MFMA"; the reference itself is fp32 only,
// libs/config.py:10-11,75-78 - the oracle for this variant is the fp32 / float64 record with a stated tolerance,
// tests/test_gpu_fp8.py).  The same regular convolution R and data adjoint as conv.hip (libs/conv.py:14-20, libs/attention.py:18-46,
// libs/scale.py:25-34), as an implicit GEMM on v_mfma_f32_32x32x16_fp8_fp8:
//
//   * both operands are OCP e4m3 (gfx950's native fp8: 1 + 4 + 3 bits, largest finite value 448), each TENSOR scaled by a power
//     of two that puts its largest magnitude into [2^7, 2^8) - the weights when their panel is packed (format 2: one byte per
//     weight, the header word carries the largest magnitude), the gathered activations while their tile is written to LDS (from the
//     producer's absmax words, like the fp16-piece form) - rounded to nearest even (v_cvt_pk_fp8_f32);
//   * fp32 accumulation; the two exact inverse scales go back in after the K loop; storage stays fp32 on both sides.
//
// Tiling as conv_igemm_bx6_kernel (256 threads, BM x 128, double-buffered LDS, register prefetch, one barrier per stage), with
// stages of 32 reduction elements: a 16-byte LDS chunk holds 16 consecutive k of one row / column, lane (r, h) of the 32x32x16
// MFMA reads chunk h and feeds its low and its high 8 bytes to two MFMAs (any k order is fine as long as both operands use the
// same one).  Every thread gathers one chunk - 16 elements - per stage.
#include "igemm.h"

// 16 fp32 values times sc -> 16 e4m3 bytes (k order: element j in byte j)
__device__ __forceinline__ uint4 quant_fp8x16(const float (&v)[16], float sc) {
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int t = 0;
        t = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * q + 0] * sc, v[4 * q + 1] * sc, t, false);
        t = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * q + 2] * sc, v[4 * q + 3] * sc, t, true);
        w[q] = (unsigned)t;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <int WGM, int WGN, int TM, int TN>
__global__ void __launch_bounds__(256, (WGM * TM > 4 ? 2 : 3)) conv_igemm_fp8_kernel(const IgParams p) {
    constexpr int BK = 32, KB = 2, NT = 256, NW = 4;
    constexpr int BM = WGM * TM * 32;
    constexpr int BN = WGN * TN * 32;
    static_assert(WGM * WGN == NW && BN == 128 && KB * BN == NT, "one gathered chunk (16 k) per thread and stage");
    static_assert(KB * BM <= 2 * NT, "at most two weight chunks per thread and stage");
    constexpr bool A2 = KB * BM > NT;
    static_assert(2 * BK <= IG_TAIL, "panel tail shorter than the prefetch distance");
    constexpr int A_U4 = 2 * KB * BM, B_U4 = 2 * KB * BN;
    constexpr int EPI_U4 = (NW * 32 * 33 + 8 + 3) / 4;
    constexpr int SMEM_U4 = A_U4 + B_U4 > EPI_U4 ? A_U4 + B_U4 : EPI_U4;
    __shared__ uint4 smem[SMEM_U4];
    uint4 (*As)[KB][BM] = reinterpret_cast<uint4 (*)[KB][BM]>(smem);
    uint4 (*Bs)[KB][BN] = reinterpret_cast<uint4 (*)[KB][BN]>(smem + A_U4);

    int bx, by, bz;
    xcd_tile(bx, by, bz, p.tile_nphase);
    const int zphase = bz / p.ksplit, zsplit = bz - zphase * p.ksplit;
    const IgPhase& ph = p.ph[zphase];
    const int N = p.B * ph.QH * ph.QW;
    const int n0 = bx * BN;
    const int m0 = by * BM;
    if (n0 >= N) return;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int ncol = tid % BN;
    const int kgrp = __builtin_amdgcn_readfirstlane(tid / BN);   // wave-uniform 16-deep k-block of this thread's chunk
    const GatherCol gc = gather_setup(p, ph, n0 + ncol, N);
    float col_scale[TN];
    igemm_col_scales<WGM, WGN, TM, TN>(p, ph, col_scale, N, n0, wn, lane);
    const int kb_ = f8_scale_exp(absmax_read(p.b_absmax));
    const int ka_ = f8_scale_exp(__builtin_amdgcn_readfirstlane(*ph.a_absmax));
    const float b_scale = pow2f(kb_), a_unscale = pow2f(-ka_), b_unscale = pow2f(-kb_);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // K range of this block (split-K), in 32-deep stages; the host's split accounting counts 16-deep steps (Kpad % 32 == 0)
    const int total_steps = ph.Kpad / BK;
    const int per_split = (total_steps + p.ksplit - 1) / p.ksplit;
    const int step0 = zsplit * per_split;
    int nsteps = total_steps - step0;
    if (nsteps > per_split) nsteps = per_split;
    if (nsteps < 0) nsteps = 0;

    const bool a_thread = KB * BM == NT || tid < KB * BM;
    const int a_kb = a_thread ? tid / BM : 0, a_m = a_thread ? tid % BM : 0;
    const uint4* ap = ph.w3 + (long long)(step0 * KB + a_kb) * ph.ld + (m0 + a_m);
    const long long a_step = (long long)KB * ph.ld;
    const bool b_thread = A2 && tid + NT < KB * BM;
    const int b_kb = b_thread ? (tid + NT) / BM : 0, b_m = b_thread ? (tid + NT) % BM : 0;
    const uint4* bp = ph.w3 + (long long)(step0 * KB + b_kb) * ph.ld + (m0 + b_m);
    int kidx = step0 * BK + kgrp * 16;

    uint4 areg0, areg1;
    float breg[16];
    i32x8 offs0 = *reinterpret_cast<const i32x8*>(ph.koff + __builtin_amdgcn_readfirstlane(kidx));
    i32x8 offs1 = *reinterpret_cast<const i32x8*>(ph.koff + __builtin_amdgcn_readfirstlane(kidx) + 8);
    u32x4 taps = *reinterpret_cast<const u32x4*>(ph.ktap + __builtin_amdgcn_readfirstlane(kidx));
    auto issue_loads = [&]() {
        areg0 = ap[0];
        ap += a_step;
        if constexpr (A2) {
            areg1 = bp[0];
            bp += a_step;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) breg[j] = gather_load(gc, offs0[j], (taps[j >> 2] >> (8 * (j & 3))) & 31u);
#pragma unroll
        for (int j = 0; j < 8; ++j) breg[8 + j] = gather_load(gc, offs1[j], (taps[2 + (j >> 2)] >> (8 * (j & 3))) & 31u);
        kidx += BK;
        const int ks = __builtin_amdgcn_readfirstlane(kidx);
        offs0 = *reinterpret_cast<const i32x8*>(ph.koff + ks);
        offs1 = *reinterpret_cast<const i32x8*>(ph.koff + ks + 8);
        taps = *reinterpret_cast<const u32x4*>(ph.ktap + ks);
    };
    auto store_tiles = [&](int buf) {
        if (a_thread) As[buf][a_kb][a_m] = areg0;
        if constexpr (A2) {
            if (b_thread) As[buf][b_kb][b_m] = areg1;
        }
        Bs[buf][kgrp][ncol] = quant_fp8x16(breg, b_scale);
    };

    if (nsteps > 0) {
        issue_loads();
        store_tiles(0);
        issue_loads();
    }
    __syncthreads();
    const int lrow = lane >> 5, lcol = lane & 31;
    constexpr int NMF = TM * TN * 2;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        uint4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[buf][lrow][(wm * TM + i) * 32 + lcol];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[buf][lrow][(wn * TN + j) * 32 + lcol];
        auto mfmas = [&](int lo, int hi) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int base = (i * TN + j) * 2;
                    const long alo = (long)(((unsigned long long)a[i].y << 32) | a[i].x), ahi = (long)(((unsigned long long)a[i].w << 32) | a[i].z);
                    const long blo = (long)(((unsigned long long)b[j].y << 32) | b[j].x), bhi = (long)(((unsigned long long)b[j].w << 32) | b[j].z);
                    if (base + 0 >= lo && base + 0 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(alo, blo, acc[i][j], 0, 0, 0);
                    if (base + 1 >= lo && base + 1 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(ahi, bhi, acc[i][j], 0, 0, 0);
                }
        };
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, NMF / 2);
        __builtin_amdgcn_sched_barrier(0);
        store_tiles(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        issue_loads();
        mfmas(NMF / 2, NMF);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] * a_unscale) * b_unscale;

    float* const stage = reinterpret_cast<float*>(smem) + wid * (32 * 33);
    bool combined = false;
    if (p.ksplit > 1 && p.combine) {
        // split-K combined inside the launch: the protocol of conv_igemm_bx6_kernel
        constexpr int FR = TM * TN * 4;
        const int tile = (zphase * (int)gridDim.y + by) * (int)gridDim.x + bx;
        const int ntiles = p.nphase * (int)gridDim.y * (int)gridDim.x;
        const unsigned tile_bytes = FR * NT * 16;
        __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, (int)((unsigned)p.ksplit * ntiles * tile_bytes), 0x00020000);
        const unsigned my = ((unsigned)(zsplit * ntiles + tile)) * tile_bytes + (unsigned)tid * 16u;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u32x4 v;
                    v[0] = __float_as_uint(acc[i][j][4 * q + 0]); v[1] = __float_as_uint(acc[i][j][4 * q + 1]);
                    v[2] = __float_as_uint(acc[i][j][4 * q + 2]); v[3] = __float_as_uint(acc[i][j][4 * q + 3]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, srs, (int)(my + (unsigned)(((i * TN + j) * 4 + q) * NT * 16)), 0, 16);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* const flag = reinterpret_cast<unsigned*>(smem) + NW * 32 * 33 + 4;
        if (tid == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(p.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = ticket == (unsigned)(p.ksplit - 1);
            if (last) {
                __hip_atomic_store(p.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *flag = last ? 1u : 0u;
        }
        __syncthreads();
        if (*flag == 0u) return;
        const unsigned t0 = (unsigned)tile * tile_bytes + (unsigned)tid * 16u;
        const unsigned zstride = (unsigned)ntiles * tile_bytes;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        constexpr int HB = 4, ZB = 4;
#pragma unroll
        for (int h0 = 0; h0 < FR; h0 += HB) {
            for (int z0 = 0; z0 < p.ksplit; z0 += ZB) {
                u32x4 v[ZB][HB];
#pragma unroll
                for (int zz = 0; zz < ZB; ++zz)
#pragma unroll
                    for (int f = 0; f < HB; ++f)
                        v[zz][f] = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)(t0 + (unsigned)(z0 + zz) * zstride + (unsigned)((h0 + f) * NT * 16)), 0, 16);
#pragma unroll
                for (int zz = 0; zz < ZB; ++zz)
#pragma unroll
                    for (int f = 0; f < HB; ++f) {
                        const int g = h0 + f, i = g / (TN * 4), j = (g / 4) % TN, q = g % 4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += __uint_as_float(v[zz][f][e]);
                    }
            }
        }
        __syncthreads();
        combined = true;
    }
    igemm_epilogue<WGM, WGN, TM, TN>(p, ph, acc, col_scale, N, n0, m0, combined ? 0 : zsplit, wm, wn, lane, stage, p.ksplit > 1 && !combined);
}

void launch_fp8_igemm(const IgParams& p, dim3 grid, int bm, hipStream_t st) {
    if (bm == 192) conv_igemm_fp8_kernel<2, 2, 3, 2><<<grid, 256, 0, st>>>(p);
    else if (bm == 128) conv_igemm_fp8_kernel<2, 2, 2, 2><<<grid, 256, 0, st>>>(p);
    else if (bm == 96) conv_igemm_fp8_kernel<1, 4, 3, 1><<<grid, 256, 0, st>>>(p);
    else if (bm == 64) conv_igemm_fp8_kernel<1, 4, 2, 1><<<grid, 256, 0, st>>>(p);
    else conv_igemm_fp8_kernel<1, 4, 1, 1><<<grid, 256, 0, st>>>(p);
}
