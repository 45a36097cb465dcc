// Dense contractions of the LocAtE hot path as LDS-tiled implicit GEMMs on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32, 157 TFLOP/s dense peak on MI355X).  No im2col buffer is ever
// materialised: the activation operand is gathered straight from NCHW into LDS.
//
// Everything is expressed through ONE regular convolution R
//     out[b, m, oh, ow] = sum_{c, kh, kw} w[m, c, kh, kw] * in[b, c, oh*s - ph + kh, ow*s - pw + kw]
// and its two adjoints:
//     locate_conv_fwd    R            Conv2d / Conv1d(k=1) / Linear forward  (reference libs/conv.py:14-20,
//                                     attention.py:18-46, scale.py:25-34, linear.py:10); ConvTranspose2d dgrad
//     locate_conv_dgrad  R^T (data)   Conv2d dgrad; ConvTranspose2d FORWARD (conv.py:49-52: both convs of a
//                                     transposed stage are ConvTranspose2d, weights [C_in, C_out, k, k])
//     locate_conv_wgrad  R^T (weight) weight gradient of either (roles of the two activations swapped by the
//                                     caller for the transposed case)
// A stride-s adjoint is decomposed into s*s sub-pixel phases, each a stride-1 gather with its own tap subset
// (4x4 s2 p1 ConvTranspose = four 2x2 convolutions; 5x5 s2 p2 dgrad = 3x3 + 3x2 + 2x3 + 2x2 taps).
//
// The weight operand is re-laid out once per weight update into a K-major [Kpad][Mpad] panel (zero padded);
// the spectral-norm factor 1/sigma multiplies the accumulator in the epilogue (reference
// libs/spectral_norm.py:31-32 materialises W_bar/sigma as a separate full-size tensor on every forward).
//
// Tiling: 256 threads = 4 waves, block tile BM x 128 (BM in {128, 96, 64, 32}), K step 16, double-buffered
// LDS with register prefetch of the next K step, one barrier per step.
#include "igemm.h"



struct PackBatch {
    PackArgs ph[4];
};

__device__ __forceinline__ void pack_tables(const PackArgs& a, int k) {
    int* koff = reinterpret_cast<int*>(a.out + (size_t)a.rows * a.ld);
    unsigned char* ktap = reinterpret_cast<unsigned char*>(koff + a.rows);
    const int T = a.TH * a.TW;
    int off = 0, tap = 31;
    if (k < a.K && T > 0) {
        const int c = k / T, t = k - c * T;
        const int th = t / a.TW, tw = t - th * a.TW;
        off = 4 * (c * a.gHW + (a.dy0 + a.dys * th) * a.gW + a.dx0 + a.dxs * tw - a.dmin);
        tap = t;
    }
    koff[k] = off;
    ktap[k] = (unsigned char)tap;
}

// fp16-piece panels: a packing block folds the largest magnitude of the weights it handled into the header word of its
// phase(s) (atomic max on the bit pattern, skipped when the word already holds as much: after the first few blocks almost
// every one).  A block of the adjoint packer handles all sub-pixel phases of its weights and reports to each of them the
// maximum over ALL its taps - an upper bound of the phase's own, which is all the scale exponent needs.
__device__ __forceinline__ void pack_publish_absmax(float m, const PackArgs* phases, int nphase, float* red) {
    m = wave_max(m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned bits = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
        for (int i = 0; i < nphase; ++i) {
            const PackArgs& a = phases[i];
            if (!a.fmt) continue;
            unsigned* word = reinterpret_cast<unsigned*>(a.out + panel_split_offset_dev(a.rows, a.ld));
            if (bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                (void)__hip_atomic_fetch_max(word, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#define PACK_MAX_TAPS 32
#define PACK_SMEM (256 * (PACK_MAX_TAPS + 1))      // floats; also holds the 64 x 65 transpose tile
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2_f16x8(const float (&v)[8], float sc, uint4& h, uint4& l);
__device__ __forceinline__ void split3_bf16x8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l);

// direct form: one chunk of 8 consecutive k for one column -> the panel's planes (two scaled fp16 pieces, or three bf16 pieces)
__device__ __forceinline__ void pack_emit_chunk(const PackArgs& a, const float (&v)[8], float sc, int64_t i) {
    const int64_t plane = (int64_t)(a.rows / 8) * a.ld;
    uint4* w3 = reinterpret_cast<uint4*>(a.out + panel_split_offset_dev(a.rows, a.ld) + (a.fmt ? PANEL_HDR : 0));
    if (a.fmt == 2) {          // fp8 plane: 16-k chunks - this 8-k chunk is one half (8 bytes) of chunk (kb8 / 2, column)
        const int64_t kb8 = i / a.ld, col = i - kb8 * a.ld;
        int t0 = 0, t1 = 0;
        t0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * sc, v[1] * sc, t0, false);
        t0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * sc, v[3] * sc, t0, true);
        t1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * sc, v[5] * sc, t1, false);
        t1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * sc, v[7] * sc, t1, true);
        reinterpret_cast<uint2*>(w3)[2 * ((kb8 >> 1) * a.ld + col) + (kb8 & 1)] = make_uint2((unsigned)t0, (unsigned)t1);
    } else if (a.fmt) {
        uint4 h, l;
        split2_f16x8(v, sc, h, l);
        w3[i] = h;
        w3[plane + i] = l;
    } else {
        bf16x8 h, m, l;
        split3_bf16x8(v, h, m, l);
        w3[i] = *reinterpret_cast<uint4*>(&h);
        w3[plane + i] = *reinterpret_cast<uint4*>(&m);
        w3[2 * plane + i] = *reinterpret_cast<uint4*>(&l);
    }
}

// generic element-wise form (any tap count): virtual grid (nbx, nphase)
__device__ __forceinline__ void pack_generic_body(const PackBatch& batch, int bx, int by, int nbx, float* smem) {
    const PackArgs& a = batch.ph[by];
    const int64_t total = (int64_t)a.rows * a.ld;
    const int64_t stride = (int64_t)nbx * 256;
    const int T = a.TH * a.TW;
    float am = 0.0f;
    for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < total; i += stride) {
        const int k = (int)(i / a.ld), col = (int)(i - (int64_t)k * a.ld);
        float v = 0.0f;
        if (k < a.K) {
            if (a.mode == 0) {
                if (col < a.M) {
                    const int c = k / T, t = k - c * T;
                    const int th = t / a.TW, tw = t - th * a.TW;
                    v = a.w[(((int64_t)col * a.C + c) * a.KH + a.kh0 + th) * a.KW + a.kw0 + tw];
                }
            } else {
                if (col < a.C) {
                    const int m = k / T, r = k - m * T;
                    const int th = r / a.TW, tw = r - th * a.TW;
                    const int kh = a.kh0 + a.s * th, kw = a.kw0 + a.s * tw;
                    v = a.w[(((int64_t)m * a.C + col) * a.KH + kh) * a.KW + kw];
                }
            }
        }
        a.out[i] = v;
        am = fmaxf(am, fabsf(v));
    }
    for (int64_t k = (int64_t)bx * 256 + threadIdx.x; k < a.rows; k += stride) pack_tables(a, (int)k);
    if (a.fmt) pack_publish_absmax(am, &a, 1, smem);
}

// mode 0 (R forward): the panel is the transpose of W viewed as [M][K]: 64 x 64 tiles through LDS, both the read
// (along k) and the write (along m) are coalesced.  virtual grid (rows / 64, ld / 64).
__device__ __forceinline__ void pack_transpose_body(const PackArgs& a, int bx, int by, float* smem) {
    float (*tile)[65] = reinterpret_cast<float (*)[65]>(smem);
    const int k0 = bx * 64, m0 = by * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float am = 0.0f;
    // this thread's reduction index k = (c, th, tw) inside a weight row [C][KH][KW]: k itself when the panel holds every tap, else
    // the tap sub-rectangle's position (layers whose maps mostly see padding: the reads then stay inside each channel's window)
    const int k = k0 + tx, T = a.TH * a.TW;
    int src = k;
    if (T != a.KH * a.KW && k < a.K) {
        const int c = k / T, t = k - c * T;
        const int th = t / a.TW, tw = t - th * a.TW;
        src = (c * a.KH + a.kh0 + th) * a.KW + a.kw0 + tw;
    }
    const int64_t wrow = (int64_t)a.C * a.KH * a.KW;
    for (int j = ty; j < 64; j += 4) {
        const int m = m0 + j;
        const float v = (m < a.M && k < a.K) ? a.w[(int64_t)m * wrow + src] : 0.0f;
        tile[j][tx] = v;
        am = fmaxf(am, fabsf(v));
    }
    __syncthreads();
    if (a.direct) {
        // direct form: the tile's 8 k-blocks x 64 columns as piece chunks, two per thread; consecutive threads write
        // consecutive 16-byte chunks of a plane row
        const unsigned bits = a.fmt ? absmax_read(a.wmax) : 0u;
        const float sc = pow2f(a.fmt == 2 ? f8_scale_exp(bits) : f16_scale_exp(bits));
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int kb = (threadIdx.x >> 6) + 4 * q, m = m0 + tx;
            if (k0 + kb * 8 < a.rows && m < a.ld) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tile[tx][kb * 8 + e];
                pack_emit_chunk(a, v, sc, (int64_t)(k0 / 8 + kb) * a.ld + m);
            }
        }
        if (a.fmt && bx == 0 && by == 0 && threadIdx.x == 0) *reinterpret_cast<unsigned*>(a.out + panel_split_offset_dev(a.rows, a.ld)) = bits;
        if (!a.keep_f32) return;
    }
    for (int j = ty; j < 64; j += 4) {
        const int k = k0 + j, m = m0 + tx;
        if (k < a.rows && m < a.ld) a.out[(int64_t)k * a.ld + m] = tile[tx][j];
    }
    if (a.direct) return;               // tables and header: from the panel's first (two-pass) packing / written above
    if (by == 0 && threadIdx.x < 64 && k0 + (int)threadIdx.x < a.rows) pack_tables(a, k0 + threadIdx.x);
    if (a.fmt) pack_publish_absmax(am, &a, 1, smem + 64 * 65);       // PACK_SMEM floats: room behind the tile
}

// Direct form of the adjoint packer (see PackArgs::wmax): a block stages eight weight rows m x 32 channels (all taps) in LDS
// - eight contiguous reads - and writes, for every sub-pixel phase, the chunks of 8 consecutive k = (m, tap) it now holds for
// its 32 columns: 8 T / 8 = T chunks per phase and column, 512-byte runs per plane row.  virtual grid (ld / 32, ceil(M / 8)).
#define PACKD_MB 8
static_assert(8704 >= PACK_SMEM, "direct packing reuses the packers' LDS block");
#define PACKD_SMEM 8704           // floats: 8 rows x 64 channels x up to 16 (+1) taps, or x 32 channels for up to 32 taps (34 KB:
                                  // four blocks per CU - the packers are bandwidth kernels, a larger block cost them occupancy)
static inline __host__ __device__ int packd_cs(int KK) { return PACKD_MB * 64 * (KK | 1) <= PACKD_SMEM ? 64 : 32; }
__device__ __forceinline__ void pack_adjoint_direct_body(const PackBatch& batch, int nphase, int bx, int by, float* lds) {
    const PackArgs& a0 = batch.ph[0];
    const int KK = a0.KH * a0.KW, S = KK | 1;
    const int CS = packd_cs(KK);
    const int m0 = by * PACKD_MB, c0 = bx * CS;
    const int cn = min(CS, a0.C - c0);
    const DivU32 dk((unsigned)KK);
    if (cn > 0) {
        const int run = cn * KK;
#pragma unroll
        for (int mm = 0; mm < PACKD_MB; ++mm) {
            if (m0 + mm >= a0.M) break;
            const float* src = a0.w + ((int64_t)(m0 + mm) * a0.C + c0) * KK;
            for (int idx = threadIdx.x; idx < run; idx += 256) {
                unsigned cl, t;
                dk.divmod((unsigned)idx, cl, t);
                lds[(mm * CS + (int)cl) * S + (int)t] = src[idx];
            }
        }
    }
    __syncthreads();
    const unsigned bits = a0.fmt ? absmax_read(a0.wmax) : 0u;
    const float sc = pow2f(a0.fmt == 2 ? f8_scale_exp(bits) : f16_scale_exp(bits));
    // work items (phase, chunk, column), columns fastest: T chunks of 8 rows k = (m, tap) per phase for this block's 8 m
    int tsum = 0;
    for (int ph = 0; ph < nphase; ++ph) tsum += batch.ph[ph].TH * batch.ph[ph].TW;
    const int csh = CS == 64 ? 6 : 5;
    for (int it = threadIdx.x; it < (tsum << csh); it += 256) {
        const int cl = it & (CS - 1);
        int ch = it >> csh, ph = 0;
        while (ch >= batch.ph[ph].TH * batch.ph[ph].TW) { ch -= batch.ph[ph].TH * batch.ph[ph].TW; ++ph; }
        const PackArgs& a = batch.ph[ph];
        const int T = a.TH * a.TW, col = c0 + cl;
        if (col >= a.ld || (by * T + ch) * 8 >= a.rows) continue;          // (the last row group may reach past the zero tail)
        int mm = (ch * 8) / T, r = ch * 8 - mm * T;
        int th = r / a.TW, tw = r - th * a.TW;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = (a.kh0 + a.s * th) * a.KW + a.kw0 + a.s * tw;
            v[e] = (m0 + mm < a.M && cl < cn) ? lds[(mm * CS + cl) * S + t] : 0.0f;
            if (++tw == a.TW) { tw = 0; if (++th == a.TH) { th = 0; ++mm; } }
        }
        pack_emit_chunk(a, v, sc, (int64_t)(by * T + ch) * a.ld + col);
        if (a.keep_f32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a.out[(int64_t)((by * T + ch) * 8 + e) * a.ld + col] = v[e];
        }
    }
    if (a0.fmt && bx == 0 && by == 0 && threadIdx.x < nphase)
        *reinterpret_cast<unsigned*>(batch.ph[threadIdx.x].out + panel_split_offset_dev(batch.ph[threadIdx.x].rows, batch.ph[threadIdx.x].ld)) = bits;
}

// mode 1 (data adjoint, all sub-pixel phases at once): for one m, W[m] is a [C][KH*KW] matrix; a block stages 256
// channels of it in LDS (contiguous read) and writes, per tap, one 256-wide piece of the row (m, tap) of the phase
// that owns the tap.  by == M: zero tail rows and the offset tables.  virtual grid (ld / 256, M + 1).
__device__ __forceinline__ void pack_adjoint_body(const PackBatch& batch, int nphase, int bx, int by, float* lds) {
    const PackArgs& a0 = batch.ph[0];
    const int KK = a0.KH * a0.KW, S = KK | 1;
    const int m = by, c0 = bx * 256;
    const int col = c0 + threadIdx.x;
    float am = 0.0f;
    if (m < a0.M) {
        const int cn = min(256, a0.C - c0);
        if (cn > 0) {
            const float* src = a0.w + ((int64_t)m * a0.C + c0) * KK;
            const int total = cn * KK;
            for (int idx = threadIdx.x; idx < total; idx += 256) {
                const int cl = idx / KK, t = idx - cl * KK;
                const float v = src[idx];
                lds[cl * S + t] = v;
                am = fmaxf(am, fabsf(v));
            }
        }
        __syncthreads();
        for (int ph = 0; ph < nphase; ++ph) {
            const PackArgs& a = batch.ph[ph];
            if (col >= a.ld) continue;
            const int T = a.TH * a.TW;
            for (int r = 0; r < T; ++r) {
                const int th = r / a.TW, tw = r - th * a.TW;
                const int t = (a.kh0 + a.s * th) * a.KW + a.kw0 + a.s * tw;
                a.out[((int64_t)m * T + r) * a.ld + col] = (int)threadIdx.x < cn ? lds[threadIdx.x * S + t] : 0.0f;
            }
        }
        if (a0.fmt) pack_publish_absmax(am, batch.ph, nphase, lds + 256 * (PACK_MAX_TAPS + 1) - 8);
        return;
    }
    for (int ph = 0; ph < nphase; ++ph) {
        const PackArgs& a = batch.ph[ph];
        if (col < a.ld)
            for (int k = a.K; k < a.rows; ++k) a.out[(int64_t)k * a.ld + col] = 0.0f;
        if (bx == 0)
            for (int k = threadIdx.x; k < a.rows; k += 256) pack_tables(a, k);
    }
}


// second packing pass: the fp32 K-major rows of a panel -> its three bf16 planes (16-byte chunks of 8 consecutive k), or
// its two scaled fp16 planes
__device__ __forceinline__ void pack_split_body(const PackArgs& a, int bx, int nbx) {
    if (a.direct || a.win) return;     // direct form / window panels: the packing blocks wrote the planes
    const float* w = a.out;
    uint4* w3 = reinterpret_cast<uint4*>(a.out + panel_split_offset_dev(a.rows, a.ld) + (a.fmt ? PANEL_HDR : 0));
    const int64_t stride = (int64_t)nbx * 256;
    if (a.fmt == 2) {          // fp8: chunks of 16 consecutive k, e4m3 bytes of w * 2^k(absmax)
        const float sc = pow2f(f8_scale_exp(*reinterpret_cast<const unsigned*>(a.out + panel_split_offset_dev(a.rows, a.ld))));
        const int64_t total16 = (int64_t)(a.rows / 16) * a.ld;
        for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < total16; i += stride) {
            const int kb = (int)(i / a.ld), col = (int)(i - (int64_t)kb * a.ld);
            unsigned wq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = w[(int64_t)(kb * 16 + 4 * q + j) * a.ld + col] * sc;
                int t = 0;
                t = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], t, false);
                t = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], t, true);
                wq[q] = (unsigned)t;
            }
            w3[i] = make_uint4(wq[0], wq[1], wq[2], wq[3]);
        }
        return;
    }
    const int64_t total = (int64_t)(a.rows / 8) * a.ld;
    if (a.fmt) {
        const float sc = pow2f(f16_scale_exp(*reinterpret_cast<const unsigned*>(a.out + panel_split_offset_dev(a.rows, a.ld))));
        for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < total; i += stride) {
            const int kb = (int)(i / a.ld), col = (int)(i - (int64_t)kb * a.ld);
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = w[(int64_t)(kb * 8 + j) * a.ld + col];
            uint4 h, l;
            split2_f16x8(v, sc, h, l);
            w3[i] = h;
            w3[total + i] = l;
        }
        return;
    }
    for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < total; i += stride) {
        const int kb = (int)(i / a.ld), col = (int)(i - (int64_t)kb * a.ld);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w[(int64_t)(kb * 8 + j) * a.ld + col];
        bf16x8 h, m, l;
        split3_bf16x8(v, h, m, l);
        w3[i] = *reinterpret_cast<uint4*>(&h);
        w3[total + i] = *reinterpret_cast<uint4*>(&m);
        w3[2 * total + i] = *reinterpret_cast<uint4*>(&l);
    }
}

// ---------------------------------------------------------------------------------------------
// Window panels (PackArgs::win; convwin.hip): chunk rows in unit order u = c8g * Tp + t - the 8 reduction channels of group c8g
// at tap t - piece planes only.  A block stages RG reduction channels x CS columns x all taps of the weight tensor in LDS
// (contiguous runs: RG KK floats per column in the regular direction, CS KK floats per reduction channel in the adjoint one)
// and writes, for every phase, the chunks (t, c8) of its columns; the last row of blocks also writes the zero rows behind the
// last unit.  Always one pass: the scale of fp16-piece planes comes from the optimizer's absmax words or from the panel's own
// header (win_absmax_jobs_kernel ran first).  virtual grid (ceil(ld / CS), ceil(reduction channels / RG)).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_win_emit(const PackArgs& a, const float (&v)[8], float sc, int64_t i) {
    const int64_t plane = (int64_t)a.urows * a.ld;
    uint4* w3 = reinterpret_cast<uint4*>(a.out + PANEL_HDR);
    if (a.fmt) {
        uint4 h, l;
        split2_f16x8(v, sc, h, l);
        w3[i] = h;
        w3[plane + i] = l;
    } else {
        bf16x8 h, m, l;
        split3_bf16x8(v, h, m, l);
        w3[i] = *reinterpret_cast<uint4*>(&h);
        w3[plane + i] = *reinterpret_cast<uint4*>(&m);
        w3[2 * plane + i] = *reinterpret_cast<uint4*>(&l);
    }
}

__device__ __forceinline__ void pack_win_body(const PackBatch& batch, int nphase, int bx, int by, int gy, float* lds) {
    const PackArgs& a0 = batch.ph[0];
    const int KK = a0.KH * a0.KW, S = KK | 1;
    const int CS = KK <= 4 ? 64 : (KK <= 16 ? 32 : 16), RG = win_pack_rg(KK);
    const int Cred = a0.mode == 0 ? a0.C : a0.M, ncol = a0.mode == 0 ? a0.M : a0.C;
    const int r0 = by * RG, col0 = bx * CS;
    const int rn = min(RG, Cred - r0), cn = min(CS, ncol - col0);
    if (rn > 0 && cn > 0) {
        const DivU32 dk((unsigned)KK);
        if (a0.mode == 0) {          // w[col][r][t]: per column a run of rn KK floats
            const int run = rn * KK;
            const DivU32 dr((unsigned)run);
            for (int idx = threadIdx.x; idx < cn * run; idx += 256) {
                unsigned col, rem, r, t;
                dr.divmod((unsigned)idx, col, rem);
                dk.divmod(rem, r, t);
                lds[((int)r * CS + (int)col) * S + (int)t] = a0.w[((int64_t)(col0 + (int)col) * a0.C + r0) * KK + rem];
            }
        } else {                     // w[r][col][t]: per reduction channel a run of cn KK floats
            const int run = cn * KK;
            const DivU32 dr((unsigned)run);
            for (int idx = threadIdx.x; idx < rn * run; idx += 256) {
                unsigned r, rem, col, t;
                dr.divmod((unsigned)idx, r, rem);
                dk.divmod(rem, col, t);
                lds[((int)r * CS + (int)col) * S + (int)t] = a0.w[((int64_t)(r0 + (int)r) * a0.C + col0) * KK + rem];
            }
        }
    }
    __syncthreads();
    unsigned bits = 0u;
    if (a0.fmt) bits = a0.wmax_single ? (unsigned)__builtin_amdgcn_readfirstlane((int)*a0.wmax) : absmax_read(a0.wmax);
    const float sc = pow2f(f16_scale_exp(bits));
    const int csh = CS == 64 ? 6 : (CS == 32 ? 5 : 4);
    const int ng8 = RG / 8;
    for (int ph = 0; ph < nphase; ++ph) {
        const PackArgs& a = batch.ph[ph];
        const int T = a.TH * a.TW;
        // work items (unit row of this block, column), columns fastest
        const int nrow = ng8 * a.Tp;
        for (int it = threadIdx.x; it < (nrow << csh); it += 256) {
            const int cl = it & (CS - 1), row = it >> csh;
            const int c8 = row / a.Tp, t = row - c8 * a.Tp;
            const int col = col0 + cl;
            const int u = (r0 / 8 + c8) * a.Tp + t;
            if (col >= a.ld || u >= a.urows) continue;
            float v[8];
            const bool tap_ok = t < T && cl < cn;
            int tapidx = 0;
            if (tap_ok) {
                const int th = t / a.TW, tw = t - th * a.TW;
                tapidx = (a.kh0 + a.s * th) * a.KW + a.kw0 + a.s * tw;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (tap_ok && c8 * 8 + e < rn) ? lds[((c8 * 8 + e) * CS + cl) * S + tapidx] : 0.0f;
            pack_win_emit(a, v, sc, (int64_t)u * a.ld + col);
        }
        if (by == gy - 1) {          // zero rows behind this block's last unit (padding groups of single-tap layers, the tail)
            const int u0 = (r0 / 8 + ng8) * a.Tp;
            const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int it = threadIdx.x; it < ((a.urows - u0) << csh); it += 256) {
                const int cl = it & (CS - 1), u = u0 + (it >> csh);
                if (col0 + cl < a.ld) pack_win_emit(a, z, 1.0f, (int64_t)u * a.ld + col0 + cl);
            }
        }
        if (a.fmt && !a.wmax_single && bx == 0 && by == 0 && threadIdx.x == 0) *reinterpret_cast<unsigned*>(a.out) = bits;
    }
}

// One packing job = all phases of one panel; `kind` selects the body, (gx, gy) is its virtual grid.
struct PackJob {
    PackBatch batch;
    int nphase, kind;          // kind 0: transpose, 1: adjoint, 2: generic, 3: adjoint, direct form
    int gx, gy;
    int block_start, pad;      // first block of this job inside a batched launch
};

static PackJob make_pack_job(const PackBatch& b, int nphase) {
    PackJob j;
    j.batch = b; j.nphase = nphase; j.block_start = 0; j.pad = 0;
    const PackArgs& a0 = b.ph[0];
    if (a0.win) {
        const int KK = a0.KH * a0.KW;
        j.kind = 4; j.gx = (a0.ld + win_pack_cs(KK) - 1) / win_pack_cs(KK);
        j.gy = ((a0.mode == 0 ? a0.C : a0.M) + win_pack_rg(KK) - 1) / win_pack_rg(KK);
    } else if (a0.mode == 0 && nphase == 1) {
        j.kind = 0; j.gx = (a0.rows + 63) / 64; j.gy = (a0.ld + 63) / 64;
    } else if (a0.mode == 1 && a0.KH * a0.KW <= PACK_MAX_TAPS && a0.direct) {
        j.kind = 3; j.gx = (a0.ld + packd_cs(a0.KH * a0.KW) - 1) / packd_cs(a0.KH * a0.KW); j.gy = (a0.M + PACKD_MB - 1) / PACKD_MB;
    } else if (a0.mode == 1 && a0.KH * a0.KW <= PACK_MAX_TAPS) {
        j.kind = 1; j.gx = (a0.ld + 255) / 256; j.gy = a0.M + 1;
    } else {
        int64_t big = 1;
        for (int i = 0; i < nphase; ++i) {
            const int64_t t = (int64_t)b.ph[i].rows * b.ph[i].ld;
            if (t > big) big = t;
        }
        j.kind = 2; j.gx = stream_grid(big, 256); j.gy = nphase;
    }
    return j;
}

__device__ __forceinline__ void pack_job_body(const PackJob& j, int local, float* smem) {
    const int bx = local % j.gx, by = local / j.gx;
    if (j.kind == 0) pack_transpose_body(j.batch.ph[0], bx, by, smem);
    else if (j.kind == 1) pack_adjoint_body(j.batch, j.nphase, bx, by, smem);
    else if (j.kind == 3) pack_adjoint_direct_body(j.batch, j.nphase, bx, by, smem);
    else if (j.kind == 4) pack_win_body(j.batch, j.nphase, bx, by, j.gy, smem);
    else pack_generic_body(j.batch, bx, by, j.gx, smem);
}

__global__ void __launch_bounds__(256) pack_job_kernel(const PackJob job) {
    __shared__ float smem[PACKD_SMEM];
    pack_job_body(job, blockIdx.x, smem);
}

#define PACK_SPLIT_BLOCKS 256
__global__ void __launch_bounds__(256) pack_split_kernel(const PackJob job) {
    if ((int)blockIdx.y < job.nphase) pack_split_body(job.batch.ph[blockIdx.y], blockIdx.x, gridDim.x);
}

// fp16-piece panels only: zero the absmax words (one thread per (job, phase)), then take the maxima
__device__ __forceinline__ void pack_clear_one(const PackJob& j, int ph) {
    if (ph < j.nphase && j.batch.ph[ph].fmt && !j.batch.ph[ph].direct) {
        const PackArgs& a = j.batch.ph[ph];
        unsigned* hdr = reinterpret_cast<unsigned*>(a.win ? a.out : a.out + panel_split_offset_dev(a.rows, a.ld));
        hdr[0] = 0u; hdr[1] = 0u; hdr[2] = 0u; hdr[3] = 0u;
    }
}
// window panels of fp16 pieces packed without the optimizer's absmax words: the largest weight magnitude goes into every phase's
// header word first (the packing blocks read it from there).  blockIdx.y = job.
__device__ __forceinline__ void win_absmax_body(const PackJob& j, int bx, int nbx) {
    const PackArgs& a0 = j.batch.ph[0];
    if (j.kind != 4 || !a0.fmt || !a0.wmax_single) return;
    __shared__ float red[16];
    const int64_t total = (int64_t)a0.M * a0.C * a0.KH * a0.KW;
    float am = 0.0f;
    for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < total; i += (int64_t)nbx * 256) am = fmaxf(am, fabsf(a0.w[i]));
    am = block_max(am, red);
    if (threadIdx.x == 0) {
        const unsigned bits = __float_as_uint(am);
        for (int ph = 0; ph < j.nphase; ++ph) {
            unsigned* word = reinterpret_cast<unsigned*>(j.batch.ph[ph].out);
            if (bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                (void)__hip_atomic_fetch_max(word, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
#define WIN_ABSMAX_BLOCKS 64
__global__ void __launch_bounds__(256) win_absmax_kernel(const PackJob job) { win_absmax_body(job, blockIdx.x, gridDim.x); }
__global__ void __launch_bounds__(256) win_absmax_jobs_kernel(const PackJob* __restrict__ jobs) { win_absmax_body(jobs[blockIdx.y], blockIdx.x, gridDim.x); }
__global__ void __launch_bounds__(64) pack_clear_kernel(const PackJob job) { if (threadIdx.x < 4) pack_clear_one(job, threadIdx.x); }
__global__ void __launch_bounds__(64) pack_clear_jobs_kernel(const PackJob* __restrict__ jobs, int n_jobs) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < 4 * n_jobs) pack_clear_one(jobs[i >> 2], i & 3);
}
// blockIdx.y = job, blockIdx.z = phase
__global__ void __launch_bounds__(256) pack_split_jobs_kernel(const PackJob* __restrict__ jobs) {
    const PackJob& j = jobs[blockIdx.y];
    if ((int)blockIdx.z < j.nphase) pack_split_body(j.batch.ph[blockIdx.z], blockIdx.x, gridDim.x);
}

// many panels in one launch: `jobs` (device) sorted by block_start; a block finds its job by bisection
__global__ void __launch_bounds__(256) pack_jobs_kernel(const PackJob* __restrict__ jobs, int n_jobs) {
    __shared__ float smem[PACKD_SMEM];
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block_start <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const PackJob& j = jobs[lo];
    pack_job_body(j, blockIdx.x - j.block_start, smem);
}


template <int WGM, int WGN, int TM, int TN>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const IgParams p) {
    constexpr int BK = IG_BK;
    constexpr int BM = WGM * TM * 32;
    constexpr int BN = WGN * TN * 32;
    constexpr int KPT = BK * BN / 256;                // gathered elements per thread per K step
    constexpr int A_F4 = BK * BM / 4;                 // float4 per A tile
    constexpr int A_PT = (A_F4 + 255) / 256;
    static_assert(WGM * WGN == 4, "four waves");
    static_assert(BN % 64 == 0 && KPT == 8, "one 8-row group of the offset table per thread and step");
    static_assert(BK <= IG_TAIL, "panel tail shorter than the prefetch distance (one stage beyond the last)");

    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    int bx, by, bz;
    xcd_tile(bx, by, bz, p.tile_nphase);
    const int zphase = bz / p.ksplit, zsplit = bz - zphase * p.ksplit;
    const IgPhase& ph = p.ph[zphase];
    const int N = p.B * ph.QH * ph.QW;
    const int n0 = bx * BN;
    const int m0 = by * BM;
    if (n0 >= N) return;   // phases can have different extents; uniform per block

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;

    // ---- per-thread gather column (fixed for the whole K loop): base pointer and the set of taps inside the input
    const int ncol = tid % BN;
    const int kgrp = __builtin_amdgcn_readfirstlane(tid / BN);   // wave-uniform
    const GatherCol gc = gather_setup(p, ph, n0 + ncol, N);
    float col_scale[TN];
    igemm_col_scales<WGM, WGN, TM, TN>(p, ph, col_scale, N, n0, wn, lane);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- K range of this block (split-K) and the running addresses of its prefetch
    const int total_steps = ph.Kpad / BK;
    const int per_split = (total_steps + p.ksplit - 1) / p.ksplit;
    const int step0 = zsplit * per_split;
    int nsteps = total_steps - step0;
    if (nsteps > per_split) nsteps = per_split;
    if (nsteps < 0) nsteps = 0;

    const float4* aptr[A_PT];      // weight-panel rows: advance BK rows per step; the panel has IG_TAIL spare rows
    bool a_ok[A_PT];
    int a_row[A_PT], a_c4[A_PT];
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
        const int idx = tid + i * 256;
        const int idc = ((i + 1) * 256 <= A_F4 || idx < A_F4) ? idx : 0;
        a_row[i] = idc / (BM / 4);
        a_c4[i] = idc - a_row[i] * (BM / 4);
        const int col = m0 + a_c4[i] * 4;
        a_ok[i] = col < ph.ld;
        aptr[i] = reinterpret_cast<const float4*>(ph.wp + (long long)(step0 * BK + a_row[i]) * ph.ld + (a_ok[i] ? col : 0));
    }
    const long long a_step = (long long)BK * ph.ld / 4;            // float4 units
    int kidx = step0 * BK + kgrp * KPT;                           // wave-uniform first table row of the next load

    float breg[KPT];
    float4 areg[A_PT];
    auto issue_loads = [&]() {
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            areg[i] = *aptr[i];
            aptr[i] += a_step;
        }
        const int ks = __builtin_amdgcn_readfirstlane(kidx);
        const i32x8 offs = *reinterpret_cast<const i32x8*>(ph.koff + ks);                      // s_load_dwordx8
        const unsigned long long taps = *reinterpret_cast<const unsigned long long*>(ph.ktap + ks);   // s_load_dwordx2
#pragma unroll
        for (int j = 0; j < KPT; ++j) breg[j] = gather_load(gc, offs[j], (unsigned)(taps >> (8 * j)) & 31u);
        kidx += BK;
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            const int idx = tid + i * 256;
            if ((i + 1) * 256 <= A_F4 || idx < A_F4) {
                float4 v = areg[i];
                v.x = a_ok[i] ? v.x : 0.f; v.y = a_ok[i] ? v.y : 0.f; v.z = a_ok[i] ? v.z : 0.f; v.w = a_ok[i] ? v.w : 0.f;
                *reinterpret_cast<float4*>(&As[buf][a_row[i]][a_c4[i] * 4]) = v;
            }
        }
#pragma unroll
        for (int j = 0; j < KPT; ++j) Bs[buf][kgrp * KPT + j][ncol] = breg[j];
    };

    if (nsteps > 0) {
        issue_loads();
        store_tiles(0);
    }
    __syncthreads();
    const int lrow = lane >> 5, lcol = lane & 31;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        // One basic block per iteration: the loads of step s+1 first (pinned by the fence), the MFMAs of step s, the
        // LDS writes of step s+1, one barrier.  The last iteration's prefetch is redundant but branch-free (zero tail
        // rows of the panel and of the offset table).
        issue_loads();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k2 = 0; k2 < BK / 2; ++k2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[buf][k2 * 2 + lrow][(wm * TM + i) * 32 + lcol];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[buf][k2 * 2 + lrow][(wn * TN + j) * 32 + lcol];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        store_tiles(buf ^ 1);
        __syncthreads();
    }

    igemm_epilogue<WGM, WGN, TM, TN>(p, ph, acc, col_scale, N, n0, m0, zsplit, wm, wn, lane, nullptr, p.ksplit > 1);
}

// ---------------------------------------------------------------------------------------------
// "bf16 x 6": the same GEMM with BOTH operands split into three bf16 pieces x = h + m + l (exact: 3 x 8 significant
// bits cover fp32's 24) and the six products of combined order <= 2^-16 kept: h*h, h*m, m*h, m*m, h*l, l*h.  The
// dropped m*l, l*m, l*l terms are O(2^-24) of the product - fp32-level accuracy (the results agree with the fp32
// MFMA path to summation order) at 192 instead of 512 matrix-core cycles per 32x32x16 slice.  The weights are split once
// per optimizer step (panel planes w3), the gathered activations while their tile is written to LDS.
// LDS images are [piece][k / 8][column][8 bf16]: one ds_read_b128 per lane delivers an MFMA fragment (lane (r, h)
// holds k = 8h .. 8h + 7 of row / column r), one ds_write_b128 per gathered fragment, conflict-free both ways.
// ---------------------------------------------------------------------------------------------
#define B6_BK 16

// NP = 3: the exact three-piece form above.  NP = 1: "bf16 operands" - both operands rounded (to nearest even) to ONE bf16
// piece, one MFMA per slice, fp32 accumulation: the precision of a bf16 mixed-precision training step (BASELINE configs[1]),
// a third of the LDS traffic and a sixth of the matrix work.  Storage stays fp32 on both sides of the kernel.
// NW = 8: eight waves on the same 128-column tile with K steps of 32 - every thread still gathers ONE fragment per stage, so a
// stage covers twice the reduction depth at the same per-thread work, each wave owns half the rows of a four-wave tile, and two
// waves per SIMD cover each other's waits: for launches that cannot put more than one block on a CU anyway (mid-sized layers:
// a hundred-odd tiles), where a lone four-wave block walks its K loop at ~1.2 us per 16-deep stage.
// Resident blocks per CU: one eight-wave block; two of the tall tiles; FOUR of the 96 x 128 and 64 x 128 two-piece tiles (117 - 128 VGPRs, at most 8 bytes of
// scratch: the 96-channel stage is 2048 blocks of 24 short K steps - more waves in flight are worth more there than registers:
// 0.108 - 0.112 -> 0.104 - 0.105 ms forward, same call, alternating); three otherwise (the other small tiles spill at 128)
#define IG_FOUR_BLOCKS_TILES 3
#define IG_FOUR_BLOCKS_MIN 2
template <int WGM, int WGN, int TM, int TN, int NP, int NW = 4>
__global__ void __launch_bounds__(64 * NW, (NW == 8 ? 1 : (WGM * TM > 4 ? 2 : ((WGM * TM * TN <= IG_FOUR_BLOCKS_TILES && WGM * TM * TN >= IG_FOUR_BLOCKS_MIN && NP <= 2) ? 4 : 3)))) conv_igemm_bx6_kernel(const IgParams p) {
    constexpr int BK = 4 * NW, KB = BK / 8, KS = BK / 16, NT = 64 * NW;
    constexpr int BM = WGM * TM * 32;
    constexpr int BN = WGN * TN * 32;
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    static_assert(WGM * WGN == NW, "wave grid");
    static_assert(BN == 128 && KB * BN == NT, "one gathered fragment (k-block) per thread and stage");
    static_assert(KB * BM <= 2 * NT, "at most two weight fragments per thread and stage");
    static_assert(B6_BK == 16, "the host's split-K accounting counts 16-deep steps");
    constexpr bool A2 = KB * BM > NT;       // tall tiles (BM = 192: 64 x 96 per wave, twice the MFMAs per gathered element)
    // the pipeline issues the loads of two stages beyond the last one (tiles nsteps and nsteps + 1): their weight chunks and
    // offset-table rows must lie inside the panel's zero tail
    static_assert(2 * BK <= IG_TAIL && (IG_TAIL % 8) == 0, "panel tail shorter than the prefetch distance");

    // ONE LDS array (a second __shared__ object beside it can cost a vmcnt(0) per stage, cdna_hip_programming.md section 5):
    // operand images of the K loop, then - they are dead after its last barrier - four 32 x 33 float patches of the staged
    // epilogue and the split-K "I am last" word
    constexpr int A_U4 = 2 * NP * KB * BM, B_U4 = 2 * NP * KB * BN;
    constexpr int EPI_U4 = (NW * 32 * 33 + 8 + 3) / 4;
    constexpr int SMEM_U4 = A_U4 + B_U4 > EPI_U4 ? A_U4 + B_U4 : EPI_U4;
    __shared__ uint4 smem[SMEM_U4];
    uint4 (*As)[NP][KB][BM] = reinterpret_cast<uint4 (*)[NP][KB][BM]>(smem);
    uint4 (*Bs)[NP][KB][BN] = reinterpret_cast<uint4 (*)[NP][KB][BN]>(smem + A_U4);

    int bx, by, bz;
    xcd_tile(bx, by, bz, p.tile_nphase);
    const int zphase = bz / p.ksplit, zsplit = bz - zphase * p.ksplit;
    const IgPhase& ph = p.ph[zphase];
    const int N = p.B * ph.QH * ph.QW;
    const int n0 = bx * BN;
    const int m0 = by * BM;
    if (n0 >= N) return;   // phases can have different extents; uniform per block

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;

    // ---- per-thread gather column (fixed for the whole K loop): base pointer and the set of taps inside the input
    const int ncol = tid % BN;
    const int kgrp = __builtin_amdgcn_readfirstlane(tid / BN);   // wave-uniform k-block of this thread's fragment
    const GatherCol gc = gather_setup(p, ph, n0 + ncol, N);
    float col_scale[TN];
    igemm_col_scales<WGM, WGN, TM, TN>(p, ph, col_scale, N, n0, wn, lane);
    // fp16 pieces: both operands are scaled by powers of two into fp16's range (the weights when their planes were packed,
    // the gathered tensor below); the exact inverse factors go back in after the K loop
    float b_scale = 1.0f, a_unscale = 1.0f, b_unscale = 1.0f;
    if constexpr (NP == 2) {
        const int kb_ = f16_scale_exp(absmax_read(p.b_absmax));
        const int ka_ = f16_scale_exp(__builtin_amdgcn_readfirstlane(*ph.a_absmax));
        b_scale = pow2f(kb_);
        a_unscale = pow2f(-ka_);
        b_unscale = pow2f(-kb_);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- K range of this block (split-K)
    const int total_steps = ph.Kpad / BK;
    const int per_split = (total_steps + p.ksplit - 1) / p.ksplit;
    const int step0 = zsplit * per_split;
    int nsteps = total_steps - step0;
    if (nsteps > per_split) nsteps = per_split;
    if (nsteps < 0) nsteps = 0;

    // weight fragment of this thread: chunk (k-block a_kb, column m0 + a_m) of the three pre-split planes
    const bool a_thread = KB * BM == NT || tid < KB * BM;
    const int a_kb = a_thread ? tid / BM : 0, a_m = a_thread ? tid % BM : 0;
    // every tile column lies inside the panel: ld = round_up(M, 32) = round_up(M, BM) for the BM pick_bm() chooses (checked by
    // launch_igemm), and the columns M .. ld - 1 are zero - no masking in the loop
    const uint4* ap = ph.w3 + (long long)(step0 * KB + a_kb) * ph.ld + (m0 + a_m);
    const long long a_step = (long long)KB * ph.ld, a_plane = ph.w3_plane;
    // second weight fragment of tall tiles: chunk index tid + NT
    const bool b_thread = A2 && tid + NT < KB * BM;
    const int b_kb = b_thread ? (tid + NT) / BM : 0, b_m = b_thread ? (tid + NT) % BM : 0;
    const uint4* bp = ph.w3 + (long long)(step0 * KB + b_kb) * ph.ld + (m0 + b_m);
    int kidx = step0 * BK + kgrp * 8;                             // wave-uniform first table row of the next load

    uint4 areg0, areg1, areg2;     // three named registers: as an array this spills to scratch (clang keeps it in memory)
    uint4 areg3, areg4, areg5;
    float breg[8];
    // the offset-table rows are fetched one stage ahead of the gather that uses them: the scalar load's round trip is
    // then off the per-stage critical path (it used to sit in front of every stage's buffer loads)
    i32x8 offs = *reinterpret_cast<const i32x8*>(ph.koff + __builtin_amdgcn_readfirstlane(kidx));                       // s_load_dwordx8
    unsigned long long taps = *reinterpret_cast<const unsigned long long*>(ph.ktap + __builtin_amdgcn_readfirstlane(kidx));   // s_load_dwordx2
    auto issue_loads = [&]() {
        areg0 = ap[0];
        if constexpr (NP >= 2) areg1 = ap[a_plane];
        if constexpr (NP == 3) areg2 = ap[2 * a_plane];
        ap += a_step;
        if constexpr (A2) {
            areg3 = bp[0];
            if constexpr (NP >= 2) areg4 = bp[a_plane];
            if constexpr (NP == 3) areg5 = bp[2 * a_plane];
            bp += a_step;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) breg[j] = gather_load(gc, offs[j], (unsigned)(taps >> (8 * j)) & 31u);
        kidx += BK;
        const int ks = __builtin_amdgcn_readfirstlane(kidx);
        offs = *reinterpret_cast<const i32x8*>(ph.koff + ks);
        taps = *reinterpret_cast<const unsigned long long*>(ph.ktap + ks);
    };
    auto store_tiles = [&](int buf) {
        if (a_thread) {
            As[buf][0][a_kb][a_m] = areg0;
            if constexpr (NP >= 2) As[buf][NP >= 2 ? 1 : 0][a_kb][a_m] = areg1;
            if constexpr (NP == 3) As[buf][NP - 1][a_kb][a_m] = areg2;
        }
        if constexpr (A2) {
            if (b_thread) {
                As[buf][0][b_kb][b_m] = areg3;
                if constexpr (NP >= 2) As[buf][NP >= 2 ? 1 : 0][b_kb][b_m] = areg4;
                if constexpr (NP == 3) As[buf][NP - 1][b_kb][b_m] = areg5;
            }
        }
        if constexpr (NP == 2) {
            uint4 h, l;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = breg[j];
            split2_f16x8(v, b_scale, h, l);
            Bs[buf][0][kgrp][ncol] = h;
            Bs[buf][NP - 1][kgrp][ncol] = l;
        } else if constexpr (NP == 3) {
            uint4 h, m, l;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = breg[j];
            split3_trunc_x8(v, h, m, l);
            Bs[buf][0][kgrp][ncol] = h;
            Bs[buf][NP - 2][kgrp][ncol] = m;
            Bs[buf][NP - 1][kgrp][ncol] = l;
        } else {
            bf16x8 h;
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = (__bf16)breg[j];          // round to nearest even (v_cvt_pk_bf16_f32)
            Bs[buf][0][kgrp][ncol] = *reinterpret_cast<uint4*>(&h);
        }
    };

    // Software pipeline: while the matrix pipe works through the first half of a stage's MFMAs, the wave converts and
    // writes the NEXT tile (its global loads were issued half a stage earlier) and immediately re-issues the loads for the
    // tile after that into the registers it just freed; the second half of the MFMAs follows.  Global-load latency and
    // the fp32 -> 3 x bf16 split are then in the shadow of the wave's own MFMAs, with no additional registers.
    if (nsteps > 0) {
        issue_loads();                       // tile 0
        store_tiles(0);
        issue_loads();                       // tile 1 (past the end: inside the panel's zero tail, IG_TAIL rows)
    }
    __syncthreads();
    const int lrow = lane >> 5, lcol = lane & 31;
    constexpr int PROD = NP == 3 ? 6 : (NP == 2 ? 3 : 1);
    constexpr int NMF = TM * TN * PROD, HALF = KS == 1 ? NMF / 2 : NMF;
    using frag_t = typename std::conditional<NP == 2, f16x8, bf16x8>::type;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        frag_t a[TM][NP], b[TN][NP];
        auto fragments = [&](int ksub) {          // the 16-deep slice `ksub` of the stage
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < NP; ++q) a[i][q] = *reinterpret_cast<const frag_t*>(&As[buf][q][2 * ksub + lrow][(wm * TM + i) * 32 + lcol]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < NP; ++q) b[j][q] = *reinterpret_cast<const frag_t*>(&Bs[buf][q][2 * ksub + lrow][(wn * TN + j) * 32 + lcol]);
        };
        fragments(0);
        auto mfmas = [&](int lo, int hi) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int base = (i * TN + j) * PROD;
                    if constexpr (NP == 2) {          // smallest terms first: l h, h l, h h (l l, 2^-22 of the product, is dropped)
                        if (base + 0 >= lo && base + 0 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                        if (base + 1 >= lo && base + 1 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                        if (base + 2 >= lo && base + 2 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                    } else if constexpr (NP == 3) {
                        if (base + 0 >= lo && base + 0 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][NP - 1], b[j][0], acc[i][j], 0, 0, 0);   // l h
                        if (base + 1 >= lo && base + 1 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][NP - 1], acc[i][j], 0, 0, 0);   // h l
                        if (base + 2 >= lo && base + 2 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][NP - 2], b[j][NP - 2], acc[i][j], 0, 0, 0);   // m m
                        if (base + 3 >= lo && base + 3 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][NP - 2], b[j][0], acc[i][j], 0, 0, 0);   // m h
                        if (base + 4 >= lo && base + 4 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][NP - 2], acc[i][j], 0, 0, 0);   // h m
                        if (base + 5 >= lo && base + 5 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);   // h h
                    } else {
                        if (base >= lo && base < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                    }
                }
        };
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, HALF);
        __builtin_amdgcn_sched_barrier(0);
        store_tiles(buf ^ 1);                // tile s + 1
        __builtin_amdgcn_sched_barrier(0);
        issue_loads();                       // tile s + 2
        if constexpr (KS == 1) {
            mfmas(HALF, NMF);
        } else {
            fragments(1);                    // the stage's second 16-deep slice
            mfmas(0, NMF);
        }
        // one vector-memory instruction (and its address arithmetic) in the shadow of every MFMA: the memory pipe takes
        // ~100 cycles per wave instruction when twelve waves queue on it, the matrix pipe 32 per MFMA
        if constexpr (NP >= 2) {
#pragma unroll
            for (int k = 0; k < (KS == 1 ? NMF - HALF : NMF); ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
        }
        __syncthreads();
    }
    if constexpr (NP == 2) {       // undo the two power-of-two scales, one after the other (each exact; their product may not be
                                   // a normal fp32 number): whatever leaves this block - output or split-K partial - is unscaled
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] * a_unscale) * b_unscale;
    }
    // the operand tiles are dead after the loop's last barrier: each wave takes a 32 x 33 float patch of them
    float* const stage = reinterpret_cast<float*>(smem) + wid * (32 * 33);
    bool combined = false;
    if (p.ksplit > 1 && p.combine) {
        // ---- split-K combined inside the launch (no reduction kernel, no slab in the output's layout) ----
        // Every block of a tile stores its partial accumulators as they sit in its registers - thread t, fragment f at
        // [z][tile][f][t][4]: one contiguous KiB per store instruction - write-through (sc1), drains them, and one lane
        // takes a ticket from the tile's counter.  The block that draws the last ticket sums all ksplit partials IN z
        // ORDER (its own included, re-read: the result does not depend on which block came last - bit-reproducible),
        // and runs the ordinary epilogue.  Protocol (cdna_hip_programming.md, Guideline 16, counter form): sc1 payload
        // stores -> every storing wave s_waitcnt vmcnt(0) -> workgroup barrier -> one relaxed agent-scope fetch_add;
        // last arriver: agent-scope acquire -> s_waitcnt vmcnt(0) -> barrier -> sc1 loads.  The counter is reset by the last
        // arriver, so the counter block stays zero between launches (it must be zero before its first use).
        constexpr int FR = TM * TN * 4;                                   // 16-byte fragments per thread
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
                    __builtin_amdgcn_raw_buffer_store_b128(v, srs, (int)(my + (unsigned)(((i * TN + j) * 4 + q) * NT * 16)), 0, 16);   // aux 16 = sc1
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* const flag = reinterpret_cast<unsigned*>(smem) + NW * 32 * 33 + 4;    // beyond the waves' 32 x 33 patches
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
        // the accumulators (stored above) become the sum; one z at a time, its fragments in two batches of loads in flight
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        // four fragments of four consecutive z in flight (16 loads: this block is the launch's tail, every round trip it
        // waits for is exposed); a z beyond ksplit lies beyond the descriptor's extent and reads as 0
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
        __syncthreads();      // the flag word's patch neighbours are about to be reused by the staged epilogue
        combined = true;
    }
    igemm_epilogue<WGM, WGN, TM, TN>(p, ph, acc, col_scale, N, n0, m0, combined ? 0 : zsplit, wm, wn, lane, stage, p.ksplit > 1 && !combined);
}

// out[b, m, :] = bias[m] + scale * sum_z slab[z][b, m, :]
__global__ void __launch_bounds__(256) igemm_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                                const float* __restrict__ bias, const float* __restrict__ scale,
                                                                int scale_bg, int scale_stride, int B, int M, int plane,
                                                                long long out_bs, long long slab_stride, int ksplit,
                                                                float* __restrict__ act_out, long long act_bs,
                                                                const float* __restrict__ mul_pre, long long mul_bs,
                                                                unsigned* __restrict__ out_absmax) {
    // flat indices stay below 2^31 (host entry): power-of-two shifts or one 32-bit division instead of 64-bit ones
    const unsigned per_b = (unsigned)M * (unsigned)plane;
    const unsigned total = (unsigned)B * per_b;
    const unsigned stride = gridDim.x * blockDim.x;
    const DivU32 dpb(per_b), dpl((unsigned)plane);
    float am = 0.0f;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        unsigned b, r;
        dpb.divmod(i, b, r);
        // eight slabs' loads in flight, added in z order (one load - wait - add per slab is ksplit exposed round trips)
        float acc = 0.0f;
        const float* __restrict__ sp = slab + i;
        int z = 0;
        for (; z + 8 <= ksplit; z += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = sp[(long long)(z + q) * slab_stride];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        }
        {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = z + q < ksplit ? sp[(long long)(z + q) * slab_stride] : 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];          // + 0.0f beyond ksplit: exact
        }
        if (scale) acc *= scale[(scale_bg ? (int)b / scale_bg : 0) * scale_stride];
        if (bias) acc += bias[dpl.div(r)];
        // the fused forms of igemm_epilogue (IgParams::act_out / mul_pre)
        if (mul_pre) { acc = roottanh_grad_f(mul_pre[(long long)b * mul_bs + r], acc); am = fmaxf(am, fabsf(acc)); }
        out[(long long)b * out_bs + r] = acc;
        if (act_out) { const float a = roottanh_f(acc); act_out[(long long)b * act_bs + r] = a; am = fmaxf(am, fabsf(a)); }
    }
    if (out_absmax) absmax_publish_wave(am, out_absmax);
}

// ---------------------------------------------------------------------------------------------
// Pointwise (1x1, stride 1, no padding) contraction with few channels on both sides (K, M <= 64) over many pixels:
// an HBM-bound stream (the self-attention gates at 64x64 read and write 50 MB each for ~1 GFLOP), where the
// tile machinery of the MFMA kernel costs more than the arithmetic.  One thread = 2 adjacent pixels x all output
// channels; the weight row of each k is wave-uniform and comes through scalar loads; 8-byte coalesced loads / stores.
//   out[b, m, q] = scale_g(b) * sum_k wp[k][m] * in[b, k, q] (+ bias[m])
// ---------------------------------------------------------------------------------------------
template <int MT, int PX>
__global__ void __launch_bounds__(256) conv_pointwise_kernel(const IgParams p) {
    const IgPhase& ph = p.ph[0];
    const int HW = p.H * p.W;
    const long long npx = (long long)p.B * HW / PX;
    const long long i_ = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i_ < npx;          // (no early return: the fused epilogue's largest-magnitude word is a wave reduction)
    const long long i = live ? i_ : 0;
    const long long n = PX * i;
    const int b = (int)(n / HW), q = (int)(n - (long long)b * HW);
    const float* ip = p.in + (long long)b * p.in_bs + q;
    const float* __restrict__ wp = ph.wp;
    const int ld = ph.ld, K = ph.K;
    float acc[MT][PX];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < PX; ++e) acc[m][e] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 4) {
        float xv[4][PX];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = min(k0 + kk, K - 1);          // rows >= K of the panel are zero
            const float* xp = ip + (long long)k * HW;
            if (PX == 2) {
                const float2 t = *reinterpret_cast<const float2*>(xp);
                xv[kk][0] = t.x; xv[kk][PX - 1] = t.y;
            } else {
                xv[kk][0] = *xp;
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float* wr = wp + (long long)(k0 + kk) * ld;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float wv = wr[m];
#pragma unroll
                for (int e = 0; e < PX; ++e) acc[m][e] = fmaf(wv, xv[kk][e], acc[m][e]);
            }
        }
    }
    const float sc = p.scale ? p.scale[(p.scale_bg ? b / p.scale_bg : 0) * p.scale_stride] : 1.0f;
    float* op = p.out + (long long)b * p.out_bs + q;
    if (p.act_out == nullptr && p.mul_pre == nullptr) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m < p.M && live) {
                const float bv = p.bias ? p.bias[m] : 0.0f;
                const float v0 = fmaf(acc[m][0], sc, bv), v1 = fmaf(acc[m][PX - 1], sc, bv);
                if (PX == 2) *reinterpret_cast<float2*>(op + (long long)m * HW) = make_float2(v0, v1);
                else op[(long long)m * HW] = v0;
            }
        }
        return;
    }
    // the fused forms of igemm_epilogue (IgParams::act_out / mul_pre); pointwise_ok has checked their alignment
    float am = 0.0f;
    const float* zp = p.mul_pre ? p.mul_pre + (long long)b * p.mul_bs + q : nullptr;
    float* ap = p.act_out ? p.act_out + (long long)b * p.act_bs + q : nullptr;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m < p.M && live) {
            const float bv = p.bias ? p.bias[m] : 0.0f;
            float v0 = fmaf(acc[m][0], sc, bv), v1 = fmaf(acc[m][PX - 1], sc, bv);
            if (zp) {
                float z0, z1;
                if (PX == 2) { const float2 t = *reinterpret_cast<const float2*>(zp + (long long)m * HW); z0 = t.x; z1 = t.y; }
                else { z0 = zp[(long long)m * HW]; z1 = z0; }
                v0 = roottanh_grad_f(z0, v0); v1 = roottanh_grad_f(z1, v1);
                am = fmaxf(am, fmaxf(fabsf(v0), fabsf(v1)));
            }
            if (PX == 2) *reinterpret_cast<float2*>(op + (long long)m * HW) = make_float2(v0, v1);
            else op[(long long)m * HW] = v0;
            if (ap) {
                const float a0 = roottanh_f(v0), a1 = roottanh_f(v1);
                if (PX == 2) *reinterpret_cast<float2*>(ap + (long long)m * HW) = make_float2(a0, a1);
                else ap[(long long)m * HW] = a0;
                am = fmaxf(am, fmaxf(fabsf(a0), fabsf(a1)));
            }
        }
    }
    if (p.out_absmax) absmax_publish_wave(am, p.out_absmax);
}


// ---------------------------------------------------------------------------------------------
// Contractions on 1x1 maps (the style linears, the squeeze convs of the channel gates, the discriminator's head: a batch of
// 64 ... 192 rows times a [C, M] matrix, <= 0.1 GFLOP) - 45 launches per step that the tile machinery above runs in
// 8 ... 25 us each: one or two 128-wide tiles, a dozen dependent memory round trips of prologue, K loop, split-K combine and
// staged store, for microseconds' worth of arithmetic.  They get a direct form in plain fp32 FMAs (exact fp32 products):
//   out[n][j] = scale(n) * sum_r in[n][r] * P[r][j] + bias[j]
// lane = output column j (the K-major fp32 panel row P[r][.] is one coalesced load), the activations in[n][r] are
// wave-uniform and come through the scalar cache, the eight waves of a block take interleaved 32-row chunks of the reduction
// and their partial sums are added in wave order through LDS (deterministic).  Forward (adjoint-0 panel: r = c, j = m) and
// input gradient (adjoint-1 panel: r = m, j = c) are the same kernel.
// ---------------------------------------------------------------------------------------------
#define SK_NT 4           // batch rows per block
#define SK_RC 32          // reduction rows per wave and chunk
#define SK_WAVES 8        // waves per block: they take interleaved chunks of the reduction

__global__ void __launch_bounds__(64 * SK_WAVES) skinny_rows_kernel(const IgParams p) {
    const IgPhase& ph = p.ph[0];
    const int R = ph.K, J = p.M, N = p.B, ld = ph.ld;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int j = blockIdx.x * 64 + lane;
    const int n0 = blockIdx.y * SK_NT;
    const bool jok = j < ld;                                   // panel columns J .. ld - 1 are zero
    const float* __restrict__ P = ph.wp + (jok ? j : 0);
    const float* __restrict__ in = p.in;
    float acc[SK_NT];
#pragma unroll
    for (int t = 0; t < SK_NT; ++t) acc[t] = 0.0f;
    const int nchunk = (R + SK_RC - 1) / SK_RC;               // rows up to round_up(R, 32) - 1 exist in the panel (zero beyond R)
    float cur[SK_RC], nxt[SK_RC];
    if (wid < nchunk) {
#pragma unroll
        for (int i = 0; i < SK_RC; ++i) cur[i] = P[(long long)(wid * SK_RC + i) * ld];
    }
    for (int c = wid; c < nchunk; c += SK_WAVES) {
        const int r0 = c * SK_RC;
        if (c + SK_WAVES < nchunk) {
#pragma unroll
            for (int i = 0; i < SK_RC; ++i) nxt[i] = P[(long long)(r0 + SK_WAVES * SK_RC + i) * ld];
        }
        if (r0 + SK_RC <= R) {
#pragma unroll
            for (int t = 0; t < SK_NT; ++t) {
                if (n0 + t < N) {
                    const float* __restrict__ xr = in + (long long)(n0 + t) * p.in_bs + r0;       // wave-uniform: scalar loads
#pragma unroll
                    for (int i = 0; i < SK_RC; ++i) acc[t] = fmaf(xr[i], cur[i], acc[t]);
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < SK_NT; ++t) {
                if (n0 + t < N) {
                    const float* __restrict__ xr = in + (long long)(n0 + t) * p.in_bs + r0;
#pragma unroll
                    for (int i = 0; i < SK_RC; ++i) {
                        const float xv = r0 + i < R ? xr[i] : 0.0f;      // never read past the row (the last row ends the tensor)
                        acc[t] = fmaf(xv, cur[i], acc[t]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < SK_RC; ++i) cur[i] = nxt[i];
    }
    __shared__ float red[SK_WAVES][SK_NT][64];
#pragma unroll
    for (int t = 0; t < SK_NT; ++t) red[wid][t][lane] = acc[t];
    __syncthreads();
    // wave t finishes row t of the tile: the partial sums in wave order, then scale and bias
    if (wid < SK_NT) {
        const int n = n0 + wid;
        float am = 0.0f;
        if (n < N && j < J) {
            float s = red[0][wid][lane];
#pragma unroll
            for (int w = 1; w < SK_WAVES; ++w) s += red[w][wid][lane];
            const float sc = p.scale ? p.scale[(p.scale_bg ? n / p.scale_bg : 0) * p.scale_stride] : 1.0f;
            float v = fmaf(s, sc, p.bias ? p.bias[j] : 0.0f);
            if (p.mul_pre) { v = roottanh_grad_f(p.mul_pre[(long long)n * p.mul_bs + j], v); am = fabsf(v); }
            p.out[(long long)n * p.out_bs + j] = v;
            if (p.act_out) { const float a = roottanh_f(v); p.act_out[(long long)n * p.act_bs + j] = a; am = fabsf(a); }
        }
        if (p.out_absmax) absmax_publish_wave(am, p.out_absmax);
        if (p.lat && blockIdx.x == 0 && n < N)          // the latent columns of the next link's input row n
            for (int c = lane; c < p.lat_z; c += 64) p.act_out[(long long)n * p.act_bs - p.lat_z + c] = p.lat[(long long)n * p.lat_bs + c];
    }
}

static bool skinny_ok(const IgParams& p) {
    if (p.nphase != 1 || path_disabled("skinny")) return false;
    const IgPhase& ph = p.ph[0];
    return ph.T == 1 && p.H == 1 && p.W == 1 && p.OH == 1 && p.OW == 1 && ph.dy0 == 0 && ph.dx0 == 0 && ph.K >= 1;
}

static void launch_skinny(const IgParams& p, hipStream_t st) {
    skinny_rows_kernel<<<dim3((p.M + 63) / 64, (p.B + SK_NT - 1) / SK_NT), 64 * SK_WAVES, 0, st>>>(p);
}

static bool pointwise_ok(const IgParams& p) {
    if (p.nphase != 1 || path_disabled("pointwise")) return false;
    const IgPhase& ph = p.ph[0];
    const int HW = p.H * p.W;
    return ph.T == 1 && p.istride == 1 && p.ostep == 1 && ph.dy0 == 0 && ph.dx0 == 0 && p.H == p.OH && p.W == p.OW &&
           ph.K >= 1 && ph.K <= 64 && p.M <= 64 && ph.ld >= ((p.M + 15) / 16) * 16 && (HW & 1) == 0 && (p.in_bs & 1) == 0 &&
           (p.out_bs & 1) == 0 && (reinterpret_cast<uintptr_t>(p.in) & 7) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 7) == 0 &&
           (long long)p.B * HW >= 131072 &&
           (!p.act_out || ((p.act_bs & 1) == 0 && (reinterpret_cast<uintptr_t>(p.act_out) & 7) == 0)) &&
           (!p.mul_pre || ((p.mul_bs & 1) == 0 && (reinterpret_cast<uintptr_t>(p.mul_pre) & 7) == 0));
}

template <int PX>
static void launch_pointwise_px(const IgParams& p, hipStream_t st) {
    const long long npx = (long long)p.B * p.H * p.W / PX;
    const int blocks = (int)((npx + 255) / 256);
    const int mt = (p.M + 15) / 16;
    if (mt == 1) conv_pointwise_kernel<16, PX><<<blocks, 256, 0, st>>>(p);
    else if (mt == 2) conv_pointwise_kernel<32, PX><<<blocks, 256, 0, st>>>(p);
    else if (mt == 3) conv_pointwise_kernel<48, PX><<<blocks, 256, 0, st>>>(p);
    else conv_pointwise_kernel<64, PX><<<blocks, 256, 0, st>>>(p);
}

static void launch_pointwise(const IgParams& p, hipStream_t st) {
    // two pixels per thread only when that still leaves >= 4 blocks per CU
    if ((long long)p.B * p.H * p.W >= (long long)knob_int("LOCATE_PX2_MIN", 2 * 256 * 1024)) launch_pointwise_px<2>(p, st);
    else launch_pointwise_px<1>(p, st);
}

// Tile height of the implicit-GEMM launch.  Tall 192 x 128 tiles (each wave 96 x 64: twice the MFMAs per gathered, split and
// LDS-written activation element, two blocks per CU) for the wide layers when the launch still fills the chip with them.
static int pick_bm(int M);
static int igemm_bm(int M, int nmax, int nphase) {
    if (M % 192 == 0 && M >= 192 && !path_disabled("tall") && !path_disabled("bx6")) {
        // only where the tall tiling alone fills its two blocks per CU (512 slots): with split-K on top, or on launches of a
        // hundred-odd blocks, the 128-row tiles at three blocks per CU measured faster (profiles/r02_igemm_tiles.txt)
        const long long tiles = (long long)((nmax + 127) / 128) * (M / 192) * nphase;
        if (tiles >= 512) return 192;
    }
    return pick_bm(M);
}

static int pick_bm(int M) {
    const int cands[4] = {128, 96, 64, 32};
    int best = 128, best_pad = 1 << 30;
    for (int i = 0; i < 4; ++i) {
        const int pad = round_up(M, cands[i]);
        if (pad < best_pad) { best_pad = pad; best = cands[i]; }
    }
    return best;
}

// Split K over extra blocks when the (M, N) tiling alone cannot fill 256 CUs (deep discriminator layers and the
// first generator stages: N = B*OH*OW is only 64 ... 1024 there while K = C*KH*KW is up to 12 800).
static int igemm_ksplit(int M, int nmax, int nphase, int min_kpad) {
    const int bm = igemm_bm(M, nmax, nphase);
    const long long tiles = (long long)((nmax + 127) / 128) * ((M + bm - 1) / bm) * nphase;
    const int slots = bm == 192 ? 512 : 768;        // resident blocks: 256 CUs x 2 (tall tiles) or x 3
    // launches of >= one full round of resident blocks, or an exact multiple of 256 from 512 up, are balanced
    if (tiles >= slots || (tiles >= 512 && tiles % 256 == 0)) return 1;
    const int steps = min_kpad / IG_BK;
    long long want = (slots + tiles - 1) / tiles;
    // at least 8 K steps (128 reduction elements) per block - 2 when the output is tiny (the style linears, the deep
    // discriminator layers at batch-sized N): those launches are a latency chain of K steps on a handful of blocks and
    // their partial tiles cost next to nothing
    const int min_steps = (long long)M * nmax <= (1 << 16) ? 2 : 8;
    if (steps < 8) return 1;                                   // a K step of a lone block is ~1.2 us, the reduction launch ~3
    const int max_split = steps / min_steps > 0 ? steps / min_steps : 1;
    if (want > max_split) want = max_split;
    if (want > 64) want = 64;
    if (want > knob_int("LOCATE_KS_MAX", 64)) want = knob_int("LOCATE_KS_MAX", 64);
    return want < 1 ? 1 : (int)want;
}

// How one launch splits K: the split count, whether the partial tiles are combined inside the launch (see the kernel) or by
// igemm_slab_reduce_kernel, and the slab space either way.  In-launch combining reads ksplit x tile bytes serially in the
// tile's last block, so it is taken only while that stays small (<= 512 KiB: ~5 us) and the tile count fits the counter
// block; deep splits of tiny outputs (the style linears, the discriminator's 1x1 ... 4x4 tail) keep the reduction kernel,
// which spreads the same bytes over the whole chip.
#define IG_MAX_COUNTERS 1024
struct SplitPlan {
    int ksplit, combine, gx, gy;
    size_t slab_floats;
};

static SplitPlan igemm_split_plan(const IgParams& p, int nmax, bool have_counters) {
    SplitPlan sp;
    const int bm = igemm_bm(p.M, nmax, p.nphase);
    int min_kpad = 1 << 30;
    for (int i = 0; i < p.nphase; ++i) min_kpad = p.ph[i].Kpad < min_kpad ? p.ph[i].Kpad : min_kpad;
    int ks = igemm_ksplit(p.M, nmax, p.nphase, min_kpad);
    if (ks > min_kpad / IG_BK) ks = min_kpad / IG_BK;
    if (ks < 1) ks = 1;
    sp.ksplit = ks;
    sp.gx = (nmax + 127) / 128;
    sp.gy = (p.M + bm - 1) / bm;
    const long long ntiles = (long long)sp.gx * sp.gy * p.nphase;
    const long long tile_floats = (long long)bm * 128;
    const long long legacy = ks > 1 ? (long long)ks * p.B * p.M * p.OH * p.OW : 0;
    const long long fused = ks > 1 ? (long long)ks * ntiles * tile_floats : 0;
    sp.combine = have_counters && ks > 1 && (long long)ks * tile_floats * 4 <= (512 << 10) && ntiles <= IG_MAX_COUNTERS &&
                 fused * 4 < (1ll << 31) && (p.precision == 1 || !path_disabled("bx6")) && !path_disabled("combine");
    sp.slab_floats = (size_t)(sp.combine ? fused : legacy);
    return sp;
}

void launch_slab_reduce(const IgParams& p, hipStream_t st) {
    igemm_slab_reduce_kernel<<<stream_grid(p.slab_stride, 256), 256, 0, st>>>(p.slab, p.out, p.bias, p.scale, p.scale_bg, p.scale_stride, p.B, p.M, p.OH * p.OW,
                                                                              p.out_bs, p.slab_stride, p.ksplit, p.act_out, p.act_bs, p.mul_pre, p.mul_bs, p.out_absmax);
}

static int launch_igemm(IgParams& p, int nmax, void* slab_ws, unsigned* counters, hipStream_t st, const char* who) {
    // phase-fastest tile order (xcd_tile) where the output map is large enough for its lines to matter (same-box A/B, gather kernels:
    // 64x64 maps -5 ... -9 %, 32x32 -3 ... -11 %, 16x16 and 8x8 even; on 4x4 maps the four phases' weight panels thrash the
    // XCD's L2 instead: +25 %)
    p.tile_nphase = (path_disabled("phasefast") || (long long)p.OH * p.OW < 1024) ? 1 : p.nphase;
    if (p.win) return launch_win_igemm(p, nmax, slab_ws, counters, st, who);
    if (skinny_ok(p)) {
        p.ksplit = 1;
        launch_skinny(p, st);
        LOCATE_LAUNCH_CHECK(who);
        return LOCATE_OK;
    }
    if (pointwise_ok(p)) {
        p.ksplit = 1;
        launch_pointwise(p, st);
        LOCATE_LAUNCH_CHECK(who);
        return LOCATE_OK;
    }
    const int bm = igemm_bm(p.M, nmax, p.nphase);
    const SplitPlan sp = igemm_split_plan(p, nmax, counters != nullptr);
    p.ksplit = sp.ksplit;
    p.combine = sp.combine;
    p.counters = counters;
    p.slab = static_cast<float*>(slab_ws);
    p.slab_stride = (long long)p.B * p.M * p.OH * p.OW;
    dim3 grid(sp.gx, sp.gy, p.nphase * p.ksplit);
    for (int i = 0; i < p.nphase; ++i)
        LOCATE_REQUIRE(round_up(p.M, bm) <= p.ph[i].ld, "%s: tile height %d does not divide the panel width %d", who, bm, p.ph[i].ld);
    if (p.precision == 3) {          // fp8 operands (convfp8.hip): the same tiling and split plan, 32-deep stages
        launch_fp8_igemm(p, grid, bm, st);
        LOCATE_LAUNCH_CHECK(who);
        if (p.ksplit > 1 && !p.combine) {
            LOCATE_REQUIRE(p.slab_stride < (1ll << 31), "%s: split-K output of %lld elements exceeds the 32-bit index range", who, p.slab_stride);
            launch_slab_reduce(p, st);
            LOCATE_LAUNCH_CHECK(who);
        }
        return LOCATE_OK;
    }
    // launches of at most ~one block per CU: the eight-wave form (see the kernel)
    const bool w8 = !path_disabled("w8") && !path_disabled("bx6") && (bm == 128 || bm == 64) &&
                    (long long)grid.x * grid.y * grid.z <= knob_int("LOCATE_W8_MAX", 320);
    if (p.precision == 2) LOCATE_REQUIRE(!path_disabled("bx6"), "%s: fp16 pieces need the bf16/fp16 MFMA kernels", who);
    if (w8) {
        if (p.precision == 1) {
            if (bm == 128) conv_igemm_bx6_kernel<2, 4, 2, 1, 1, 8><<<grid, 512, 0, st>>>(p);
            else conv_igemm_bx6_kernel<2, 4, 1, 1, 1, 8><<<grid, 512, 0, st>>>(p);
        } else if (p.precision == 2) {
            if (bm == 128) conv_igemm_bx6_kernel<2, 4, 2, 1, 2, 8><<<grid, 512, 0, st>>>(p);
            else conv_igemm_bx6_kernel<2, 4, 1, 1, 2, 8><<<grid, 512, 0, st>>>(p);
        } else {
            if (bm == 128) conv_igemm_bx6_kernel<2, 4, 2, 1, 3, 8><<<grid, 512, 0, st>>>(p);
            else conv_igemm_bx6_kernel<2, 4, 1, 1, 3, 8><<<grid, 512, 0, st>>>(p);
        }
    } else if (p.precision == 1) {
        if (bm == 192) conv_igemm_bx6_kernel<2, 2, 3, 2, 1><<<grid, 256, 0, st>>>(p);
        else if (bm == 128) conv_igemm_bx6_kernel<2, 2, 2, 2, 1><<<grid, 256, 0, st>>>(p);
        else if (bm == 96) conv_igemm_bx6_kernel<1, 4, 3, 1, 1><<<grid, 256, 0, st>>>(p);
        else if (bm == 64) conv_igemm_bx6_kernel<1, 4, 2, 1, 1><<<grid, 256, 0, st>>>(p);
        else conv_igemm_bx6_kernel<1, 4, 1, 1, 1><<<grid, 256, 0, st>>>(p);
    } else if (p.precision == 2) {
        if (bm == 192) conv_igemm_bx6_kernel<2, 2, 3, 2, 2><<<grid, 256, 0, st>>>(p);
        else if (bm == 128) conv_igemm_bx6_kernel<2, 2, 2, 2, 2><<<grid, 256, 0, st>>>(p);
        else if (bm == 96) conv_igemm_bx6_kernel<1, 4, 3, 1, 2><<<grid, 256, 0, st>>>(p);
        else if (bm == 64) conv_igemm_bx6_kernel<1, 4, 2, 1, 2><<<grid, 256, 0, st>>>(p);
        else conv_igemm_bx6_kernel<1, 4, 1, 1, 2><<<grid, 256, 0, st>>>(p);
    } else if (!path_disabled("bx6")) {
        if (bm == 192) conv_igemm_bx6_kernel<2, 2, 3, 2, 3><<<grid, 256, 0, st>>>(p);
        else if (bm == 128) conv_igemm_bx6_kernel<2, 2, 2, 2, 3><<<grid, 256, 0, st>>>(p);
        else if (bm == 96) conv_igemm_bx6_kernel<1, 4, 3, 1, 3><<<grid, 256, 0, st>>>(p);
        else if (bm == 64) conv_igemm_bx6_kernel<1, 4, 2, 1, 3><<<grid, 256, 0, st>>>(p);
        else conv_igemm_bx6_kernel<1, 4, 1, 1, 3><<<grid, 256, 0, st>>>(p);
    } else {
        if (bm == 128) conv_igemm_kernel<2, 2, 2, 2><<<grid, 256, 0, st>>>(p);
        else if (bm == 96) conv_igemm_kernel<1, 4, 3, 1><<<grid, 256, 0, st>>>(p);
        else if (bm == 64) conv_igemm_kernel<1, 4, 2, 1><<<grid, 256, 0, st>>>(p);
        else conv_igemm_kernel<1, 4, 1, 1><<<grid, 256, 0, st>>>(p);
    }
    LOCATE_LAUNCH_CHECK(who);
    if (p.ksplit > 1 && !p.combine) {
        const long long total = p.slab_stride;
        LOCATE_REQUIRE(total < (1ll << 31), "%s: split-K output of %lld elements exceeds the 32-bit index range", who, total);
        launch_slab_reduce(p, st);
        LOCATE_LAUNCH_CHECK(who);
    }
    return LOCATE_OK;
}

// all phases of one panel in a single launch
static int launch_pack(const PackBatch& b, int nphase, hipStream_t st, const char* who) {
    const PackJob j = make_pack_job(b, nphase);
    if (b.ph[0].fmt) pack_clear_kernel<<<1, 64, 0, st>>>(j);      // the absmax words the packing blocks fold their maxima into
    if (b.ph[0].win) {                 // window panels: (largest magnitude into the headers,) one packing pass
        if (b.ph[0].fmt) win_absmax_kernel<<<WIN_ABSMAX_BLOCKS, 256, 0, st>>>(j);
        pack_job_kernel<<<j.gx * j.gy, 256, 0, st>>>(j);
        LOCATE_LAUNCH_CHECK(who);
        return LOCATE_OK;
    }
    pack_job_kernel<<<j.gx * j.gy, 256, 0, st>>>(j);
    LOCATE_LAUNCH_CHECK(who);
    int64_t big = 1;
    for (int i = 0; i < nphase; ++i) {
        const int64_t t = (int64_t)(b.ph[i].rows / 8) * b.ph[i].ld;
        if (t > big) big = t;
    }
    pack_split_kernel<<<dim3(stream_grid(big, 256), nphase), 256, 0, st>>>(j);
    LOCATE_LAUNCH_CHECK(who);
    return LOCATE_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static ConvGeom make_geom(const int* g) {
    ConvGeom c;
    c.B = g[0]; c.C = g[1]; c.H = g[2]; c.W = g[3]; c.M = g[4]; c.KH = g[5]; c.KW = g[6];
    c.stride = g[7]; c.pad_h = g[8]; c.pad_w = g[9]; c.OH = g[10]; c.OW = g[11];
    return c;
}

// geom = {B, C, H, W, M, KH, KW, stride, pad_h, pad_w, OH, OW} of the regular convolution R
static size_t slab_floats(const IgParams& p, int nmax) {
    if (p.win) return win_slab_floats(p, nmax);
    if (skinny_ok(p)) return 0;
    // the caller may or may not pass counters: room for whichever form the launch then takes
    const size_t a = igemm_split_plan(p, nmax, false).slab_floats, b = igemm_split_plan(p, nmax, true).slab_floats;
    return a > b ? a : b;
}

// Taps that can ever meet the input.  On the deepest maps most of a kernel only ever sees padding (a 5x5 s2 p2 conv that
// maps 2x2 -> 1x1 reads 4 of its 25 taps, a 3x3 p1 conv on a 1x1 map one of 9): the contraction then runs over the bounding
// range of useful taps only - fewer K rows to stream, pack and multiply - with identical results (the dropped taps
// contribute exact zeros).
//   regular direction: tap k reads input i = o*s - pad + k for outputs o in [0, out): useful iff some i lands in [0, in)
static void useful_taps(int in, int out, int k, int s, int pad, int* lo, int* n) {
    if (path_disabled("taps")) { *lo = 0; *n = k; return; }
    int first = -1, last = -1;
    for (int t = 0; t < k; ++t) {
        bool any = false;
        for (int o = 0; o < out && !any; ++o) {
            const int i = o * s - pad + t;
            any = i >= 0 && i < in;
        }
        if (any) { if (first < 0) first = t; last = t; }
    }
    if (first < 0) { first = 0; last = 0; }          // nothing useful: keep one tap (it reads zeros)
    *lo = first; *n = last - first + 1;
}
//   adjoint phase: tap th reads the gathered map at q + d0 - th for phase rows q in [0, Q): useful iff that lands in [0, out)
static void useful_phase_taps(int Q, int out, int d0, int T, int* lo, int* n) {
    if (path_disabled("taps")) { *lo = 0; *n = T > 0 ? T : 1; return; }
    int first = T > 0 ? d0 - out + 1 : 0, last = d0 + Q - 1;
    if (first < 0) first = 0;
    if (last > T - 1) last = T - 1;
    if (first > last) { first = 0; last = 0; }
    *lo = first; *n = last - first + 1;
}

static void phase_taps(int parity, int pad, int K, int s, int* k0, int* d0, int* T) {
    *k0 = (parity + pad) % s;
    *d0 = (parity + pad - *k0) / s;
    *T = *k0 < K ? (K - *k0 + s - 1) / s : 0;
}

// most negative displacement dy*W + dx over the taps of a phase (0 if none is negative)
static int tap_dmin(const PackArgs& a) {
    int best = 0;
    for (int th = 0; th < a.TH; ++th)
        for (int tw = 0; tw < a.TW; ++tw) {
            const int d = (a.dy0 + a.dys * th) * a.gW + a.dx0 + a.dxs * tw;
            if (d < best) best = d;
        }
    return best;
}

static void phase_finish(IgPhase& ph, const PackArgs& pa, float* panel_base) {
    ph.wp = panel_base;
    ph.koff = reinterpret_cast<const int*>(panel_base + (size_t)pa.rows * pa.ld);
    ph.ktap = reinterpret_cast<const unsigned char*>(ph.koff + pa.rows);
    ph.w3 = reinterpret_cast<const uint4*>(panel_base + panel_split_offset(pa.rows, pa.ld) + (pa.fmt ? PANEL_HDR : 0));
    ph.a_absmax = pa.fmt ? reinterpret_cast<const unsigned*>(panel_base + panel_split_offset(pa.rows, pa.ld)) : nullptr;
    ph.w3_plane = (long long)(pa.rows / 8) * pa.ld;
    ph.dmin = pa.dmin;
    ph.K = pa.K; ph.Kpad = pa.rows - IG_TAIL; ph.ld = pa.ld;
    ph.TW = pa.TW > 0 ? pa.TW : 1; ph.tw_magic = (65536 + ph.TW - 1) / ph.TW;
    ph.dy0 = pa.dy0; ph.dys = pa.dys; ph.dx0 = pa.dx0; ph.dxs = pa.dxs;
}

// Window form of a planned contraction (convwin.hip): turns the phase table conv_plan built into the window kernels' - panel
// layout in unit order, tile structure, LDS window geometry, per-tap slot displacements - or returns false when this geometry
// has none (the caller then keeps the gather kernels).  What the window kernels need:
//   * >= 16 reduction channels, a multiple of 8; an even map width and even plane size (8-byte loads of pixel pairs; a power of
//     two for stride-2 gathers);
//   * column tiles (128 or 256 wide) that are whole rows of one image (BN % QW == 0) or whole images (BN % (QH QW) == 0), in every phase;
//   * every phase with more than one tap, or a single phase with one tap (then the map is treated as flat pixel runs);
//   * operand images that fit the LDS budget below.
// The caller's alignment conditions on the tensor itself (base address, batch stride) are checked at launch.
#define WIN_LDS_BUDGET (128 * 1024)
static bool win_finish(IgParams& p, PackBatch& batch, int fmt, float* panel, size_t* off_out) {
    if (path_disabled("win") || p.nphase < 1 || p.C < 16 || (p.C & 7)) return false;
    bool any1 = false;
    for (int i = 0; i < p.nphase; ++i) {
        if (p.ph[i].T < 1) return false;
        any1 = any1 || p.ph[i].T == 1;
    }
    if (any1 && (p.nphase > 1 || p.istride != 1 || p.ostep != 1 || p.ph[0].dy0 != 0 || p.ph[0].dx0 != 0 || p.H != p.OH || p.W != p.OW)) return false;
    if (any1) {
        // single tap: the map is a flat run of pixels; re-shape it into rows of min(HW, tile width) pixels
        const long long HW = (long long)p.H * p.W;
        if (HW < 2 || (HW & (HW - 1)) != 0) return false;          // powers of two only (every map of the model is)
        int bn1 = 128;
        if (knob_int("LOCATE_WIN_BM", 0) > 0 && p.M % knob_int("LOCATE_WIN_BM", 0) == 0) bn1 = win_pick_bn(knob_int("LOCATE_WIN_BM", 0));
        if (knob_int("LOCATE_WIN_BN", 0) > 0) bn1 = knob_int("LOCATE_WIN_BN", 0);
        const int Wc = HW >= bn1 ? bn1 : (int)HW;
        p.W = p.OW = Wc; p.H = p.OH = (int)(HW / Wc);
        p.ph[0].QW = Wc; p.ph[0].QH = p.H;
        p.ph[0].TW = 1; p.ph[0].tw_magic = 65536;
    }
    if ((p.W & 1) || (((long long)p.H * p.W) & 1)) return false;
    int bm = win_pick_bm(p.M), bn = win_pick_bn(bm);
    if (any1) {
        // single-tap layers (plain GEMMs of a few GFLOP): 64 x 128 tiles while that gives every CU a block, else 32 x 128
        // (measured: profiles/notes_r04_experiments.md)
        const long long N1 = (long long)p.B * p.H * p.W;
        bn = 128;
        bm = ((N1 + 127) / 128) * ((p.M + 63) / 64) >= 256 ? 64 : 32;
        if (round_up(p.M, 64) != round_up(p.M, 32)) bm = 32;          // (the panel is round_up(M, 32) columns wide: M = 96 has no 64-row tiling)
    }
    {   // (debug library: force a tile - LOCATE_WIN_BM rows, LOCATE_WIN_BN columns)
        const int fbm = knob_int("LOCATE_WIN_BM", 0), fbn = knob_int("LOCATE_WIN_BN", 0);
        if (fbm > 0 && p.M % fbm == 0) { bm = fbm; bn = win_pick_bn(bm); }
        if (fbn > 0) bn = fbn;
    }
    const int C8G = (p.C + 7) / 8;
    int X[4], nx = 0;
    for (int i = 0; i < p.nphase; ++i) X[nx++] = any1 ? C8G : p.ph[i].T;
    const int SL = win_pick_sl(X, nx), U = 2 * SL;
    const int NP = fmt ? 2 : 3;
    int slotsp = 0, bpt = 1;
    size_t off = 0;
    for (int i = 0; i < p.nphase; ++i) {
        IgPhase& ph = p.ph[i];
        PackArgs& pa = batch.ph[i];
        const int QHW = ph.QH * ph.QW;
        if (QHW >= bn) {
            if (QHW % bn != 0 || bn % ph.QW != 0) return false;
            ph.win_NI = 1; ph.win_TR = bn / ph.QW;
        } else {
            if (bn % QHW != 0) return false;
            ph.win_NI = bn / QHW; ph.win_TR = ph.QH;
        }
        const int T = ph.T, TH = T / ph.TW;
        if (TH * ph.TW != T || T > 25) return false;
        const int dyA = ph.dy0, dyB = ph.dy0 + ph.dys * (TH - 1);
        const int dymin = dyA < dyB ? dyA : dyB, dymax = dyA < dyB ? dyB : dyA;
        int WR = (ph.win_TR - 1) * p.istride + (dymax - dymin) + 1;
        if (WR > p.H) WR = p.H;
        ph.win_WR = WR;
        fastdiv_make((unsigned)(WR * p.W), &ph.win_wrw_mul, &ph.win_wrw_s1, &ph.win_wrw_s2);
        ph.win_Tp = T == 1 ? 1 : round_up(T, U);
        ph.win_NG = T == 1 ? 1 : ph.win_Tp / U;
        const int slots = ph.win_NI * WR * p.W;
        if (slots + 2 > slotsp) slotsp = slots + 2;          // + the trash slot (masked items' chunks) and the zero slot (taps outside the input)
        // the loader's capacity: at most two items (pixel pairs x 8 channels) per thread and stage
        const long long items = (long long)(T == 1 ? U : 1) * (slots / 2);
        if (items > 2ll * 256 * ph.win_NG) return false;
        if (items > 256ll * ph.win_NG) bpt = 2;
        for (int t = 0; t < 32; ++t) {
            int c = 0;
            if (t < T) {
                const int th = t / ph.TW, tw = t - th * ph.TW;
                const int dy = ph.dy0 + ph.dys * th, dx = ph.dx0 + ph.dxs * tw;
                c = dy * p.W + (p.istride == 2 ? (dx & 1) * (p.W / 2) + (dx >> 1) : dx);
            }
            ph.tapc[t] = c;
        }
        const int C8Gp = T == 1 ? round_up(C8G, U) : C8G;
        pa.win = 1; pa.Tp = ph.win_Tp; pa.urows = C8Gp * ph.win_Tp + WIN_TAIL_UNITS;
        pa.out = panel ? panel + off : nullptr;
        // the panel is header + planes: the K-major fp32 rows, offset tables and their pointers do not exist
        ph.wp = nullptr; ph.koff = nullptr; ph.ktap = nullptr;
        ph.a_absmax = (fmt && pa.out) ? reinterpret_cast<const unsigned*>(pa.out) : nullptr;
        ph.w3 = pa.out ? reinterpret_cast<const uint4*>(pa.out + PANEL_HDR) : nullptr;
        ph.w3_plane = (long long)pa.urows * pa.ld;
        ph.Kpad = C8Gp * ph.win_Tp * 8;
        // non-direct packing reads its scale from the panel's own header (win_absmax_jobs_kernel)
        pa.wmax = (fmt && pa.out) ? reinterpret_cast<const unsigned*>(pa.out) : nullptr;
        pa.wmax_single = 1;
        off += win_panel_floats(pa.urows, pa.ld, fmt);
    }
    if (p.istride == 2 && (p.W & (p.W - 1))) return false;          // (the parity de-interleave masks with W - 1)
    int ngm = 1;
    for (int i = 0; i < p.nphase; ++i) ngm = p.ph[i].win_NG > ngm ? p.ph[i].win_NG : ngm;
    const size_t lds = ((size_t)2 * NP * U * bm + (size_t)2 * NP * (any1 ? U : 1) * slotsp) * 16 + (size_t)ngm * bpt * 256 * 8;
    if (lds > WIN_LDS_BUDGET) return false;
    if (any1)
        for (int t = 0; t < 32; ++t) p.ph[0].tapc[t] = t < U ? t * slotsp : 0;
    p.win = 1; p.win_U = U; p.win_slotsp = slotsp; p.win_bm = bm; p.win_bn = bn;
    *off_out = off;
    return true;
}

// Fills the phase table of R (adjoint = 0) or of its data adjoint (adjoint = 1: one phase per sub-pixel).
// `panel` is the packed-weight buffer (may be null when only sizes are wanted); with `pack` the packing kernels
// are launched.  Returns the panel size in floats and the largest per-phase N.
static int conv_plan(const ConvGeom& g, int adjoint_fmt, const float* w, float* panel, IgParams& p, int* nmax_out,
                     size_t* panel_floats_out, bool pack, hipStream_t st, PackBatch* batch_out = nullptr) {
    // adjoint_fmt: bit 0 = direction (0: R, 1: its data adjoint), bit 1 = panel format (0: bf16 planes, 1: fp16-piece planes),
    // bit 2 = window panel (chunk rows in unit order for the LDS-window kernels of convwin.hip; win_finish below)
    // bit 3 = fp8 panel (one e4m3 plane, convfp8.hip)
    const int adjoint = adjoint_fmt & 1, fmt = (adjoint_fmt & 8) ? 2 : ((adjoint_fmt >> 1) & 1);
    size_t off = 0;
    int nmax = 0;
    p.nphase = 0;
    PackBatch batch;
    if (!adjoint) {
        p.B = g.B; p.C = g.C; p.H = g.H; p.W = g.W; p.M = g.M; p.OH = g.OH; p.OW = g.OW;
        p.istride = g.stride; p.ostep = 1; p.nphase = 1;
        PackArgs pa;
        pa.w = w; pa.out = panel;
        pa.M = g.M; pa.C = g.C; pa.KH = g.KH; pa.KW = g.KW; pa.mode = 0;
        pa.s = 1;
        useful_taps(g.H, g.OH, g.KH, g.stride, g.pad_h, &pa.kh0, &pa.TH);
        useful_taps(g.W, g.OW, g.KW, g.stride, g.pad_w, &pa.kw0, &pa.TW);
        pa.K = g.C * pa.TH * pa.TW; pa.rows = round_up(pa.K, IG_KPAD) + IG_TAIL; pa.ld = round_up(g.M, 32);
        pa.gHW = g.H * g.W; pa.gW = g.W; pa.dy0 = pa.kh0 - g.pad_h; pa.dys = 1; pa.dx0 = pa.kw0 - g.pad_w; pa.dxs = 1;
        pa.dmin = tap_dmin(pa);
        pa.fmt = fmt; pa.wmax = nullptr; pa.direct = 0; pa.keep_f32 = 1;
        pa.win = 0; pa.Tp = 0; pa.urows = 0; pa.wmax_single = 0;
        batch.ph[0] = pa;
        IgPhase& ph = p.ph[0];
        phase_finish(ph, pa, panel);
        ph.T = pa.TH * pa.TW;
        ph.oy0 = ph.ox0 = 0; ph.QH = g.OH; ph.QW = g.OW;
        off = panel_floats(pa.rows, pa.ld, fmt);
        nmax = g.B * g.OH * g.OW;
    } else {
        p.B = g.B; p.C = g.M; p.H = g.OH; p.W = g.OW; p.M = g.C; p.OH = g.H; p.OW = g.W;
        p.istride = 1; p.ostep = g.stride;
        for (int py = 0; py < g.stride; ++py)
            for (int px = 0; px < g.stride; ++px) {
                int kh0, dy0, TH, kw0, dx0, TW;
                phase_taps(py, g.pad_h, g.KH, g.stride, &kh0, &dy0, &TH);
                phase_taps(px, g.pad_w, g.KW, g.stride, &kw0, &dx0, &TW);
                const int QH = py < g.H ? (g.H - py + g.stride - 1) / g.stride : 0;
                const int QW = px < g.W ? (g.W - px + g.stride - 1) / g.stride : 0;
                if (QH == 0 || QW == 0) continue;
                {
                    int lo, n;
                    useful_phase_taps(QH, g.OH, dy0, TH, &lo, &n);
                    kh0 += g.stride * lo; dy0 -= lo; TH = TH > 0 ? n : 0;
                    useful_phase_taps(QW, g.OW, dx0, TW, &lo, &n);
                    kw0 += g.stride * lo; dx0 -= lo; TW = TW > 0 ? n : 0;
                }
                IgPhase& ph = p.ph[p.nphase++];
                PackArgs pa;
                pa.w = w; pa.out = panel ? panel + off : nullptr;
                pa.M = g.M; pa.C = g.C; pa.KH = g.KH; pa.KW = g.KW; pa.mode = 1;
                pa.kh0 = kh0; pa.kw0 = kw0; pa.s = g.stride; pa.TH = TH; pa.TW = TW;
                pa.K = g.M * TH * TW; pa.rows = round_up(pa.K > 0 ? pa.K : 1, IG_KPAD) + IG_TAIL; pa.ld = round_up(g.C, 32);
                pa.gHW = g.OH * g.OW; pa.gW = g.OW; pa.dy0 = dy0; pa.dys = -1; pa.dx0 = dx0; pa.dxs = -1;
                pa.dmin = tap_dmin(pa);
                pa.fmt = fmt; pa.wmax = nullptr; pa.direct = 0; pa.keep_f32 = 1;
                pa.win = 0; pa.Tp = 0; pa.urows = 0; pa.wmax_single = 0;
                batch.ph[p.nphase - 1] = pa;
                phase_finish(ph, pa, pa.out);
                ph.T = TH * TW;
                ph.oy0 = py; ph.ox0 = px; ph.QH = QH; ph.QW = QW;
                off += panel_floats(pa.rows, pa.ld, fmt);
                const int nph = g.B * QH * QW;
                if (nph > nmax) nmax = nph;
            }
    }
    p.win = 0;
    if (adjoint_fmt & 4) {
        LOCATE_REQUIRE(win_finish(p, batch, fmt, panel, &off), "conv: this geometry has no window form (ask locate_conv_win_ok first)");
    }
    if (nmax_out) *nmax_out = nmax;
    if (panel_floats_out) *panel_floats_out = off;
    if (batch_out) *batch_out = batch;
    if (pack && p.nphase > 0)
        if (int e = launch_pack(batch, p.nphase, st, "locate_conv_pack_panel")) return e;
    return LOCATE_OK;
}

// adjoint = 0: panel for locate_conv_fwd; adjoint = 1: panels (one per sub-pixel phase) for locate_conv_dgrad
LOCATE_API size_t locate_conv_panel_bytes(const int* geom, int adjoint) {
    IgParams p;
    size_t n = 0;
    conv_plan(make_geom(geom), adjoint, nullptr, nullptr, p, nullptr, &n, false, nullptr);
    return n * sizeof(float);
}

// Re-lays W [M, C, KH, KW] out as the K-major, zero-padded panel(s) the implicit GEMM streams, followed by the
// gather offset table of this geometry.  Only needs to be redone when W changes (once per optimizer step), not per
// forward: the spectral-norm 1/sigma is applied in the GEMM epilogue instead of being baked into the weights.
LOCATE_API int locate_conv_pack_panel(const int* geom, int adjoint, const float* w, float* panel, void* stream) {
    const ConvGeom g = make_geom(geom);
    if (int e = geom_check(g, "locate_conv_pack_panel")) return e;
    LOCATE_REQUIRE(w && panel, "locate_conv_pack_panel: null pointer");
    IgParams p;
    return conv_plan(g, adjoint, w, panel, p, nullptr, nullptr, true, as_stream(stream));
}

// Batched form (all panels of a network in ONE launch after an optimizer step): the caller fills one host record per
// panel with locate_conv_pack_job (block_start = running sum of the returned block counts), uploads the array and
// calls locate_conv_pack_panels.  Records hold raw pointers: rebuild them when a weight or panel buffer moves.
LOCATE_API size_t locate_conv_pack_job_bytes(void) { return sizeof(PackJob); }

// direct != 0: the job RE-packs a panel that has been packed in full before (its offset tables and zero tails are kept) in its
// direct form - the piece planes in one pass, straight from the weights, the K-major fp32 rows only where a kernel reads them.
// A fp16-piece panel needs weight_absmax for that: locate_absmax_words() device words holding the largest magnitude of w as it
// is NOW (the optimizer kernel leaves them, locate_nadam_step); without them - and for geometries the direct bodies do not
// cover - the job silently takes the two-pass form.
LOCATE_API int locate_conv_pack_job(const int* geom, int adjoint, const float* w, float* panel, int block_start, void* job_out,
                                    int* blocks_out, int direct, const void* weight_absmax) {
    const ConvGeom g = make_geom(geom);
    if (int e = geom_check(g, "locate_conv_pack_job")) return e;
    LOCATE_REQUIRE(w && panel && job_out && blocks_out && block_start >= 0, "locate_conv_pack_job: bad arguments");
    IgParams p;
    PackBatch batch;
    if (int e = conv_plan(g, adjoint, w, panel, p, nullptr, nullptr, false, nullptr, &batch)) return e;
    LOCATE_REQUIRE(p.nphase > 0, "locate_conv_pack_job: empty panel");
    if (batch.ph[0].win) {
        // window panels are always packed in one pass; "direct" = their scale is at hand (bf16 pieces need none, fp16 pieces take
        // the optimizer's absmax words) - otherwise the launch runs the absmax pre-pass into the panel headers first
        for (int i = 0; i < p.nphase; ++i) {
            if (weight_absmax && (adjoint & 2)) { batch.ph[i].wmax = static_cast<const unsigned*>(weight_absmax); batch.ph[i].wmax_single = 0; }
            batch.ph[i].direct = (weight_absmax || !(adjoint & 2)) ? 1 : 0;
        }
    } else if (direct && (weight_absmax || !(adjoint & 10))) {          // (fp16-piece and fp8 panels need the weights' absmax words)
        const PackArgs& a0 = batch.ph[0];
        const bool transpose = a0.mode == 0 && p.nphase == 1;
        const bool adj = a0.mode == 1 && a0.KH * a0.KW <= PACK_MAX_TAPS;
        if (transpose || adj) {
            // single-tap panels feed the pointwise / 1x1-map kernels, which read the fp32 rows; the debug library's fp32-MFMA
            // fallback reads them for every panel
            bool keep = path_disabled("bx6") || path_disabled("direct_planes_only");
            for (int i = 0; i < p.nphase; ++i) keep = keep || (p.nphase == 1 && batch.ph[i].TH * batch.ph[i].TW == 1);
            for (int i = 0; i < p.nphase; ++i) {
                batch.ph[i].wmax = static_cast<const unsigned*>(weight_absmax);
                batch.ph[i].direct = 1;
                batch.ph[i].keep_f32 = keep ? 1 : 0;
            }
        }
    }
    PackJob j = make_pack_job(batch, p.nphase);
    j.block_start = block_start;
    memcpy(job_out, &j, sizeof(PackJob));
    *blocks_out = j.gx * j.gy;
    return LOCATE_OK;
}

// any_f16: some job is a TWO-PASS fp16-piece panel (its absmax header is cleared first); any_two_pass: some job is in the
// two-pass form at all (the split launch is needed) - both 0 when every job was built in the direct form: one launch.
// passes: bit 0 = some gather-kernel panel is in the two-pass form (split launch), bit 1 = some WINDOW panel of fp16 pieces has no
// absmax words (absmax pre-pass into the panel headers).
LOCATE_API int locate_conv_pack_panels(const void* jobs, int n_jobs, int total_blocks, int any_f16, int passes, void* stream) {
    LOCATE_REQUIRE(jobs && n_jobs > 0 && total_blocks > 0, "locate_conv_pack_panels: bad arguments");
    const int any_two_pass = passes & 1;
    if (any_f16)        // fp16-piece panels: zero the absmax words the packing blocks fold their maxima into
        pack_clear_jobs_kernel<<<(4 * n_jobs + 63) / 64, 64, 0, as_stream(stream)>>>(static_cast<const PackJob*>(jobs), n_jobs);
    if (passes & 2)
        win_absmax_jobs_kernel<<<dim3(WIN_ABSMAX_BLOCKS, n_jobs), 256, 0, as_stream(stream)>>>(static_cast<const PackJob*>(jobs));
    pack_jobs_kernel<<<total_blocks, 256, 0, as_stream(stream)>>>(static_cast<const PackJob*>(jobs), n_jobs);
    LOCATE_LAUNCH_CHECK("locate_conv_pack_panels");
    if (any_two_pass) {
        pack_split_jobs_kernel<<<dim3(PACK_SPLIT_BLOCKS, n_jobs, 4), 256, 0, as_stream(stream)>>>(static_cast<const PackJob*>(jobs));
        LOCATE_LAUNCH_CHECK("locate_conv_pack_panels(split)");
    }
    return LOCATE_OK;
}
// whether a job built by locate_conv_pack_job took the direct form (then it needs neither the clearing nor the split launch)
// 1 for a window panel's job (its non-direct form needs pass bit 1 of locate_conv_pack_panels, never the split launch)
LOCATE_API int locate_conv_pack_job_is_window(const void* job) {
    if (!job) return 0;
    PackJob j;
    memcpy(&j, job, sizeof(j));
    return j.kind == 4;
}
LOCATE_API int locate_conv_pack_job_is_direct(const void* job) {
    if (!job) return 0;
    PackJob j;
    memcpy(&j, job, sizeof(j));
    return j.batch.ph[0].direct;
}

// optional activated second output of locate_conv_fwd (HOST struct, see IgParams::act_out)
struct LocateActEpilogue {
    void* act_out;
    long long act_bs;
    const void* lat;
    long long lat_bs;
    int lat_z, pad;
    const void* mul_pre;
    long long mul_bs;
    void* out_absmax;
};

static int run_igemm(const ConvGeom& g, int adjoint, const float* in, int64_t in_bs, const float* panel, const float* scale,
                     int scale_bg, int scale_stride, const float* bias, float* out, int64_t out_bs, float* ws, unsigned* counters,
                     int precision, const unsigned* in_absmax, hipStream_t st, const char* who, const LocateActEpilogue* epi = nullptr) {
    const int win = (precision >> 4) & 1;          // bit 4: the panel is a window panel (locate_conv_win_ok)
    precision &= 15;
    LOCATE_REQUIRE(precision >= 0 && precision <= 3, "%s: precision must be 0 (fp32-faithful, bf16 pieces), 1 (bf16 operands), 2 (fp32-faithful, fp16 pieces) or 3 (fp8 operands)", who);
    LOCATE_REQUIRE(precision < 2 || in_absmax, "%s: precisions 2 and 3 need the gathered tensor's absmax words", who);
    LOCATE_REQUIRE(!(win && precision == 3), "%s: no window form of the fp8 contractions", who);
    IgParams p;
    p.precision = precision;
    p.b_absmax = in_absmax;
    p.act_out = nullptr; p.act_bs = 0; p.lat = nullptr; p.lat_bs = 0; p.lat_z = 0;
    p.mul_pre = nullptr; p.mul_bs = 0; p.out_absmax = nullptr;
    int nmax = 0;
    if (int e = conv_plan(g, (adjoint & 1) | (win ? 4 : 0) | (precision == 2 ? 2 : 0) | (precision == 3 ? 8 : 0), nullptr, const_cast<float*>(panel), p, &nmax, nullptr, false, st)) return e;
    LOCATE_REQUIRE(p.nphase > 0, "%s: empty output", who);
    LOCATE_REQUIRE(!win || ((in_bs & 1) == 0 && (reinterpret_cast<uintptr_t>(in) & 7) == 0), "%s: the window form loads pixel pairs: 8-byte aligned tensor, even batch stride", who);
    p.in = in; p.out = out; p.bias = bias; p.scale = scale; p.in_bs = in_bs; p.out_bs = out_bs;
    p.scale_bg = scale_bg; p.scale_stride = scale_stride;
    {
        const long long extent = 4ll * ((long long)(p.B - 1) * in_bs + (long long)p.C * p.H * p.W);
        LOCATE_REQUIRE(in_bs >= 0 && extent > 0 && extent < (1ll << 31) - (1 << 20), "%s: gathered tensor of %lld bytes exceeds the 2 GiB a buffer descriptor addresses", who, extent);
        p.in_bytes = (unsigned)extent;
    }
    LOCATE_REQUIRE(scale_bg >= 0 && (scale_bg == 0 || g.B % scale_bg == 0), "%s: batch %d is not a multiple of the scale group %d", who, g.B, scale_bg);
    LOCATE_REQUIRE(ws || slab_floats(p, nmax) == 0, "%s: split-K needs a workspace", who);
    if (epi && epi->act_out) {
        LOCATE_REQUIRE(!epi->lat || skinny_ok(p), "%s: the latent prefix of the activated second output exists for 1x1-map layers only", who);
        LOCATE_REQUIRE(!epi->lat || (epi->lat_z > 0 && epi->act_bs >= epi->lat_z + g.M), "%s: bad latent prefix", who);
        LOCATE_REQUIRE(!epi->mul_pre, "%s: an epilogue either activates or multiplies by the activation's derivative", who);
        LOCATE_REQUIRE(epi->act_bs >= (long long)p.M * p.OH * p.OW, "%s: activated output's batch stride is smaller than one element of the batch", who);
        p.act_out = static_cast<float*>(epi->act_out); p.act_bs = epi->act_bs;
        p.lat = static_cast<const float*>(epi->lat); p.lat_bs = epi->lat_bs; p.lat_z = epi->lat ? epi->lat_z : 0;
    }
    if (epi && epi->mul_pre) {
        LOCATE_REQUIRE(epi->mul_bs >= (long long)p.M * p.OH * p.OW, "%s: pre-activation's batch stride is smaller than one element of the batch", who);
        p.mul_pre = static_cast<const float*>(epi->mul_pre); p.mul_bs = epi->mul_bs;
    }
    if (epi && (epi->act_out || epi->mul_pre)) p.out_absmax = static_cast<unsigned*>(epi->out_absmax);
    return launch_igemm(p, nmax, ws, counters, st, who);
}

static size_t igemm_ws_bytes(const int* geom, int adjoint) {
    IgParams p;
    p.precision = 0;
    int nmax = 0;
    if (conv_plan(make_geom(geom), adjoint, nullptr, nullptr, p, &nmax, nullptr, false, nullptr)) return 0;
    return (p.nphase > 0 ? slab_floats(p, nmax) : 0) * sizeof(float);
}

// Whether this geometry and direction (bit 0 of adjoint_fmt; bit 1 = fp16-piece planes) has a WINDOW form (convwin.hip): the
// caller then packs the panel with format bit 2 set (adjoint_fmt | 4, here and in locate_conv_panel_bytes / _pack_panel / _pack_job)
// and passes adjoint-direction entry points the same panel as before; locate_conv_fwd / locate_conv_dgrad take the flag through
// their `precision` argument's bit 4 (precision | 16).  x_bs / x: the gathered tensor's batch stride and address (pixel pairs are
// loaded 8 bytes wide).  1x1 maps and the narrow streaming layers keep their own kernels: 0 for them.  Returns 1 where the window
// form is also the faster one on MI355X (what the Python layer takes by default), 2 where it merely exists.
LOCATE_API int locate_conv_win_ok(const int* geom, int adjoint_fmt, int64_t x_bs, const void* x) {
    const ConvGeom g = make_geom(geom);
    if (geom_check(g, "locate_conv_win_ok")) return 0;
    if ((x_bs & 1) || (reinterpret_cast<uintptr_t>(x) & 7)) return 0;
    IgParams p;
    p.precision = 0;
    int nmax = 0;
    if (conv_plan(g, adjoint_fmt & 1, nullptr, nullptr, p, &nmax, nullptr, false, nullptr) || p.nphase < 1) return 0;
    p.in = static_cast<const float*>(x); p.out = nullptr; p.in_bs = x_bs; p.out_bs = 2;
    if (skinny_ok(p)) return 0;
    {   // the narrow pointwise stream (geometry part of pointwise_ok)
        const IgPhase& ph = p.ph[0];
        if (p.nphase == 1 && ph.T == 1 && p.istride == 1 && p.ostep == 1 && p.H == p.OH && p.W == p.OW && ph.K <= 64 && p.M <= 64 &&
            (long long)p.B * p.H * p.W >= 131072 && !path_disabled("pointwise")) return 0;
    }
    if (conv_plan(g, (adjoint_fmt & 3) | 4, nullptr, nullptr, p, &nmax, nullptr, false, nullptr) || !p.win) return 0;
    // 1: the window form is the measured choice (profiles/notes_r04_experiments.md: the weight-streaming layers - a single tap over
    // >= 384 reduction channels, the four 2x2-tap phases of a transposed 4x4 stride-2 conv over >= 512); 2: it exists and is
    // correct, the gather kernels measured as fast or faster
    bool taps4 = p.nphase == 4;
    for (int i = 0; i < p.nphase; ++i) taps4 = taps4 && p.ph[i].T == 4;
    const bool best = p.M >= 128 && ((p.nphase == 1 && p.ph[0].T == 1 && p.C >= 384) || (taps4 && p.C >= 512));
    return best ? 1 : 2;
}
// split-K workspace of the window form (0 when the launch already fills the chip)
LOCATE_API size_t locate_conv_win_workspace_bytes(const int* geom, int adjoint_fmt) {
    IgParams p;
    p.precision = (adjoint_fmt & 2) ? 2 : 0;
    int nmax = 0;
    if (conv_plan(make_geom(geom), (adjoint_fmt & 3) | 4, nullptr, nullptr, p, &nmax, nullptr, false, nullptr) || !p.win) return 0;
    return win_slab_floats(p, nmax) * sizeof(float);
}

// split-K slab space (0 when the launch already fills the chip)
LOCATE_API size_t locate_conv_fwd_workspace_bytes(const int* geom) { return igemm_ws_bytes(geom, 0); }
LOCATE_API size_t locate_conv_dgrad_workspace_bytes(const int* geom) { return igemm_ws_bytes(geom, 1); }

// y[b, m, oh, ow] = bias[m] + scale_g(b) * sum w[m, c, kh, kw] x[b, c, oh*s-ph+kh, ow*s-pw+kw]     (panel: adjoint = 0)
// x_bs / y_bs: batch strides in elements (channel-sliced views of a contiguous NCHW tensor are allowed).
// scale (nullable): scale_group_batch = 0 -> one device scalar; > 0 -> batch element b uses
// scale[(b / scale_group_batch) * scale_stride] (several forwards stacked along the batch, each with its own sigma).
LOCATE_API size_t locate_conv_counter_bytes(void) { return IG_MAX_COUNTERS * sizeof(unsigned); }

// counters (nullable): locate_conv_counter_bytes() bytes of device memory, ZERO before the first call that uses them and
// left zero by every completed call, not shared by launches that may run concurrently.  With counters the split-K partial
// tiles of mid-sized launches are combined inside the launch instead of by a second kernel.
LOCATE_API int locate_conv_fwd(const int* geom, const float* x, int64_t x_bs, const float* panel, const float* scale,
                               int scale_group_batch, int scale_stride, const float* bias, float* y, int64_t y_bs,
                               void* workspace, void* counters, int precision, const void* x_absmax, const void* act_epilogue,
                               void* stream) {
    const ConvGeom g = make_geom(geom);
    if (int e = geom_check(g, "locate_conv_fwd")) return e;
    LOCATE_REQUIRE(x && panel && y, "locate_conv_fwd: null pointer");
    return run_igemm(g, 0, x, x_bs, panel, scale, scale_group_batch, scale_stride, bias, y, y_bs, static_cast<float*>(workspace),
                     static_cast<unsigned*>(counters), precision, static_cast<const unsigned*>(x_absmax), as_stream(stream), "locate_conv_fwd",
                     static_cast<const LocateActEpilogue*>(act_epilogue));
}

// gx[b, c, i, j] = bias[c] + scale * sum_{m, kh, kw} gy[b, m, oh, ow] w[m, c, kh, kw],  i = oh*s - ph + kh, j = ow*s - pw + kw
// (data adjoint of R; also the FORWARD of ConvTranspose2d with weight [C_in = M, C_out = C, KH, KW]; panel: adjoint = 1).
// Every element of gx [B, C, H, W] is written.
LOCATE_API int locate_conv_dgrad(const int* geom, const float* gy, int64_t gy_bs, const float* panel, const float* scale,
                                 int scale_group_batch, int scale_stride, const float* bias, float* gx, int64_t gx_bs,
                                 void* workspace, void* counters, int precision, const void* gy_absmax, const void* act_epilogue,
                                 void* stream) {
    const ConvGeom g = make_geom(geom);
    if (int e = geom_check(g, "locate_conv_dgrad")) return e;
    LOCATE_REQUIRE(gy && panel && gx, "locate_conv_dgrad: null pointer");
    return run_igemm(g, 1, gy, gy_bs, panel, scale, scale_group_batch, scale_stride, bias, gx, gx_bs,
                     static_cast<float*>(workspace), static_cast<unsigned*>(counters), precision, static_cast<const unsigned*>(gy_absmax),
                     as_stream(stream), "locate_conv_dgrad", static_cast<const LocateActEpilogue*>(act_epilogue));
}

// ---------------------------------------------------------------------------------------------
// weight gradient:  gw[m, c, kh, kw] = sum_{b, oh, ow} gy[b, m, oh, ow] * x[b, c, oh*s-ph+kh, ow*s-pw+kw]
// GEMM rows = m, columns r = (c, kh, kw), reduction over n = (b, oh, ow) split over blockIdx.z into slabs
// (deterministic: slabs are summed in a fixed order by a second kernel).
// ---------------------------------------------------------------------------------------------
#define WG_BK 32

struct WgParams {
    const float* x;     // gathered activation [B, C, H, W]
    const float* gy;    // dense activation    [B, M, OH, OW]
    float* slab;        // [nsplit][M * R]
    long long x_bs, gy_bs;
    unsigned x_bytes;   // extent of the x view in bytes (buffer descriptor bound of the bf16 path)
    int B, C, H, W, M, OH, OW, KH, KW, stride, pad_h, pad_w;
    int R;              // C * KH * KW
    int N;              // B * OH * OW
    int chunk;          // reduction elements per split (multiple of WG_BK)
    int zper, Ng;       // splits per stacked call and reduction elements per call: split z covers elements
                        // [(z / zper) Ng + (z % zper) chunk, ...) and never crosses a call boundary (one call: zper = nsplit, Ng = N)
    unsigned q_mul, ow_mul;   // division by Q = OH*OW and by OW as multiply-high + shifts (see fastdiv)
    int q_s1, q_s2, ow_s1, ow_s2;
    // single-split launches finish in the epilogue (no slab, no reduce kernel):
    int gscale_bg, gscale_stride;   // > 0: gy of batch element b is multiplied by inv_scale[(b / gscale_bg) * gscale_stride]
                              // while it is loaded (stacked forwards with different sigma); inv_scale then is NOT
                              // applied in the epilogue
    float* direct_out;        // gw, or null when slabs are used
    const float* w_ref;       // W_bar for the fused <G, W_bar> partial sums (nullable)
    const float* inv_scale;   // device scalar 1/sigma (nullable)
    double* partial;          // one double per block (nullable)
    const unsigned* x_absmax; // fp16 pieces (NP = 2): largest magnitudes of x and of gy, AMAX_WORDS words of bit patterns each
    const unsigned* g_absmax;
};


// Shared epilogue of the weight-gradient kernels.
template <int WGM, int WGN, int TM, int TN>
__device__ __forceinline__ void wgrad_epilogue(const WgParams& p, f32x16 (&acc)[TM][TN], int r0, int m0, int wm, int wn, int lane,
                                               int wid, int tid) {
    const int lrow = lane >> 5, lcol = lane & 31;
    // epilogue.  With a single split the result is final: scale by 1/sigma, write the gradient in the weight's own
    // layout and reduce this block's share of <G, W_bar> (spectral-norm backward needs it) - no slab round trip.
    __shared__ double red[4];
    const bool direct = p.direct_out != nullptr;
    float* dst = direct ? p.direct_out : p.slab + (long long)blockIdx.z * p.M * p.R;
    const float sc = (direct && p.inv_scale && p.gscale_bg == 0) ? p.inv_scale[0] : 1.0f;
    double dot = 0.0;
    const bool want_dot = direct && p.w_ref != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = r0 + (wn * TN + j) * 32 + lcol;
        const bool r_ok = r < p.R;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // this lane's 16 rows of the 32x32 tile: W_bar values first (branch-free, all loads in flight), then the
            // products in fp32 per tile and the running sum in fp64
            float wref[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lrow;
                const bool ok = want_dot && r_ok && m < p.M;
                const float* wp_ = ok ? p.w_ref + (long long)m * p.R + r : p.gy;      // always a mapped address
                wref[e] = ok ? *wp_ : 0.0f;
            }
            float part = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lrow;
                const float v = acc[i][j][e];
                part = fmaf(v, wref[e], part);
                if (r_ok && m < p.M) dst[(long long)m * p.R + r] = v * sc;
            }
            dot += (double)part;
        }
    }
    if (direct && p.partial) {
        dot = wave_sum_d(dot);
        if (lane == 0) red[wid] = dot;
        __syncthreads();
        if (tid == 0) p.partial[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

template <int WGM, int WGN, int TM, int TN>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(const WgParams p) {
    constexpr int BM = WGM * TM * 32;
    constexpr int BR = WGN * TN * 32;
    constexpr int G_PT = WG_BK * BM / 256;
    constexpr int X_PT = WG_BK * BR / 256;
    static_assert(WGM * WGN == 4, "four waves");

    __shared__ float Gs[2][WG_BK][BM + 1];
    __shared__ float Xs[2][WG_BK][BR + 1];
    __shared__ int rt_off[BR];       // c*H*W + dy*W + dx, or INT_MIN for r >= R
    __shared__ signed char rt_dy[BR], rt_dx[BR];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int r0 = blockIdx.x * BR, m0 = blockIdx.y * BM;
    const int taps = p.KH * p.KW;
    for (int i = tid; i < BR; i += 256) {
        const int r = r0 + i;
        if (r < p.R) {
            const int c = r / taps, t = r - c * taps;
            const int kh = t / p.KW, kw = t - kh * p.KW;
            rt_dy[i] = (signed char)(kh - p.pad_h);
            rt_dx[i] = (signed char)(kw - p.pad_w);
            rt_off[i] = c * p.H * p.W + (kh - p.pad_h) * p.W + (kw - p.pad_w);
        } else {
            rt_dy[i] = rt_dx[i] = 0;
            rt_off[i] = -2147483647 - 1;
        }
    }
    __syncthreads();

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int zgrp = (int)blockIdx.z / p.zper;
    const int n_begin = zgrp * p.Ng + ((int)blockIdx.z - zgrp * p.zper) * p.chunk;
    int n_end = n_begin + p.chunk;
    if (n_end > (zgrp + 1) * p.Ng) n_end = (zgrp + 1) * p.Ng;
    const int nl = tid & 31, sub = tid >> 5;   // reduction lane, row/column subgroup (0..7)
    const int Q = p.OH * p.OW;

    float greg[G_PT], xreg[X_PT];
    unsigned gmask = 0, xmask = 0;   // validity bits, applied when the tiles are written to LDS (loads are unconditional
                                     // from clamped addresses so that they all issue back to back, see conv_igemm_kernel)
    // per-thread invariants of the gathered operand: this thread always loads the same X_PT im2col columns
    int xoff[X_PT], xdyx[X_PT];
#pragma unroll
    for (int i = 0; i < X_PT; ++i) {
        const int rl = sub + 8 * i;
        xoff[i] = rt_off[rl];
        xdyx[i] = ((int)rt_dy[rl] & 0xffff) | ((int)rt_dx[rl] << 16);
    }
    // group scales (at most 4 groups) live in registers; gsc = the scale of the tile currently held in greg[]
    float gs0 = 1.0f, gs1 = 1.0f, gs2 = 1.0f, gs3 = 1.0f, gsc = 1.0f;
    if (p.gscale_bg > 0) {
        const int ng = p.B / p.gscale_bg;
        gs0 = p.inv_scale[0];
        gs1 = ng > 1 ? p.inv_scale[p.gscale_stride] : 1.0f;
        gs2 = ng > 2 ? p.inv_scale[2 * p.gscale_stride] : 1.0f;
        gs3 = ng > 3 ? p.inv_scale[3 * p.gscale_stride] : 1.0f;
    }
    auto load_tiles = [&](int nb) {
        const int n = nb + nl;
        const bool ok = n < n_end;
        const int nn = ok ? n : 0;
        const int b = fastdiv(nn, p.q_mul, p.q_s1, p.q_s2), q = nn - b * Q;
        const int oh = fastdiv(q, p.ow_mul, p.ow_s1, p.ow_s2), ow = q - oh * p.OW;
        const float* gp = p.gy + (long long)b * p.gy_bs + q;
        if (p.gscale_bg > 0) {       // group of batch element b (at most 4 groups): compares, no division in the hot loop
            const int bg = p.gscale_bg;
            gsc = gs0;
            gsc = b >= bg ? gs1 : gsc;
            gsc = b >= 2 * bg ? gs2 : gsc;
            gsc = b >= 3 * bg ? gs3 : gsc;
        }
        gmask = 0;
#pragma unroll
        for (int i = 0; i < G_PT; ++i) {
            const int m = m0 + sub + 8 * i;
            const bool v = ok && m < p.M;
            greg[i] = gp[v ? (long long)m * Q : 0];
            gmask |= (v ? 1u : 0u) << i;
        }
        const int iy0 = oh * p.stride, ix0 = ow * p.stride;
        const int base = iy0 * p.W + ix0;
        const float* xp = p.x + (long long)b * p.x_bs;       // start of batch image b
        xmask = 0;
#pragma unroll
        for (int i = 0; i < X_PT; ++i) {
            const int dy = (short)(xdyx[i] & 0xffff), dx = xdyx[i] >> 16;
            const bool v = ok && xoff[i] != (-2147483647 - 1) && (unsigned)(iy0 + dy) < (unsigned)p.H &&
                           (unsigned)(ix0 + dx) < (unsigned)p.W;
            xreg[i] = xp[v ? base + xoff[i] : 0];            // masked lanes read element 0 of the image (always valid)
            xmask |= (v ? 1u : 0u) << i;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < G_PT; ++i) Gs[buf][nl][sub + 8 * i] = ((gmask >> i) & 1u) ? greg[i] * gsc : 0.0f;
#pragma unroll
        for (int i = 0; i < X_PT; ++i) Xs[buf][nl][sub + 8 * i] = ((xmask >> i) & 1u) ? xreg[i] : 0.0f;
    };

    const int nsteps = (n_end - n_begin + WG_BK - 1) / WG_BK;
    const int lrow = lane >> 5, lcol = lane & 31;
    if (nsteps > 0) {
        load_tiles(n_begin);
        store_tiles(0);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) load_tiles(n_begin + (s + 1) * WG_BK);
#pragma unroll
        for (int k2 = 0; k2 < WG_BK / 2; ++k2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Gs[buf][k2 * 2 + lrow][(wm * TM + i) * 32 + lcol];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Xs[buf][k2 * 2 + lrow][(wn * TN + j) * 32 + lcol];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nsteps) store_tiles(buf ^ 1);
        __syncthreads();
    }

    wgrad_epilogue<WGM, WGN, TM, TN>(p, acc, r0, m0, wm, wn, lane, wid, tid);
}

// ---------------------------------------------------------------------------------------------
// The weight gradient on the bf16 matrix cores with exact three-way splits (see conv_igemm_bx6_kernel).  The reduction
// index n = (b, oh, ow) is the MFMA's k: every thread loads PAIRS of adjacent n (coalesced along n), splits them and
// writes the packed bf16 pairs to LDS images [piece][row][n] (n contiguous, 32-byte rows with an XOR swizzle of the two
// halves: conflict-free ds_read_b128 fragments of 8 consecutive n; 48 KiB per block, three blocks per CU).  Needs even OH*OW and OW (a pair never straddles an image or a row).
// ---------------------------------------------------------------------------------------------
#define WB_BK 16                      // reduction elements per stage = one MFMA k
#define WB_PITCH 8                    // dwords per LDS row (16 bf16, no padding): the two 16-byte halves of a row are
                                      // swapped on rows with bit 3 set, which makes both the ds_read_b128 fragment reads
                                      // (16-lane groups = 16 consecutive rows) and the dword writes conflict-free

// NP = 4: fp8 operands (BASELINE configs[4], the arithmetic of convfp8.hip): both operands scaled into e4m3's range, rounded to e4m3
// (v_cvt_pk_fp8_f32, a pair per instruction) and multiplied on v_mfma_f32_32x32x16_fp8_fp8.  LDS rows are 16 bytes (16 consecutive
// n of one row / column); lane (r, h) reads the 8 bytes k = 8h .. 8h + 7, the two halves swapped on rows with bit 4 set (conflict-free
// ds_read_b64 over 32 rows); a thread's pair goes in as one 16-bit store.
template <int WGM, int WGN, int TM, int TN, int NP>       // NP = 3: exact splits; NP = 1: bf16 operands; NP = 2: two scaled fp16 pieces (see conv_igemm_bx6_kernel)
__global__ void __launch_bounds__(256, (WGM * TM > 4 ? 2 : 3)) conv_wgrad_bx6_kernel(const WgParams p) {
    constexpr int BM = WGM * TM * 32;
    constexpr int BR = WGN * TN * 32;
    constexpr int G_PT = BM / 32;          // row groups per thread: rows sub + 32 i
    constexpr int X_PT = BR / 32;
    static_assert(WGM * WGN == 4, "four waves");

    constexpr int NPL = NP == 4 ? 1 : NP;          // piece planes in LDS
    __shared__ unsigned Gs[2][NPL][NP == 4 ? 1 : BM][WB_PITCH];
    __shared__ unsigned Xs[2][NPL][NP == 4 ? 1 : BR][WB_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short G8[2][NP == 4 ? BM : 1][8];          // fp8: 16 bytes per row, addressed in pairs
    __shared__ __attribute__((aligned(16))) unsigned short X8[2][NP == 4 ? BR : 1][8];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int r0 = blockIdx.x * BR, m0 = blockIdx.y * BM;
    const int taps = p.KH * p.KW;
    const int np = tid & 7, sub = tid >> 3;      // pair index inside the stage (n = nb + 2 np), row/column subgroup 0..31
    const int Q = p.OH * p.OW;

    // per-thread invariants of the gathered operand: this thread always loads the same X_PT im2col columns
    int xoff[X_PT], xdy[X_PT], xdx[X_PT];
#pragma unroll
    for (int i = 0; i < X_PT; ++i) {
        const int r = r0 + sub + 32 * i;
        if (r < p.R) {
            const int c = r / taps, t = r - c * taps;
            const int kh = t / p.KW, kw = t - kh * p.KW;
            xdy[i] = kh - p.pad_h;
            xdx[i] = kw - p.pad_w;
            xoff[i] = 4 * (c * p.H * p.W + xdy[i] * p.W + xdx[i]);
        } else {
            xdy[i] = -(1 << 20);               // never inside the input
            xdx[i] = 0;
            xoff[i] = 0;
        }
    }
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int zgrp = (int)blockIdx.z / p.zper;
    const int n_begin = zgrp * p.Ng + ((int)blockIdx.z - zgrp * p.zper) * p.chunk;
    int n_end = n_begin + p.chunk;
    if (n_end > (zgrp + 1) * p.Ng) n_end = (zgrp + 1) * p.Ng;

    // group scales (at most 4 groups) live in registers; gsc = the scale of the pairs currently held in greg[]
    float gs0 = 1.0f, gs1 = 1.0f, gs2 = 1.0f, gs3 = 1.0f, gsc = 1.0f;
    if (p.gscale_bg > 0) {
        const int ng = p.B / p.gscale_bg;
        gs0 = p.inv_scale[0];
        gs1 = ng > 1 ? p.inv_scale[p.gscale_stride] : 1.0f;
        gs2 = ng > 2 ? p.inv_scale[2 * p.gscale_stride] : 1.0f;
        gs3 = ng > 3 ? p.inv_scale[3 * p.gscale_stride] : 1.0f;
    }

    // fp16 pieces: both operands go through powers of two into fp16's range (gy after its per-call 1/sigma, whose largest
    // value bounds the product), the exact inverses are applied to the accumulators after the loop
    float g_scale = 1.0f, x_scale = 1.0f, g_unscale = 1.0f, x_unscale = 1.0f;
    if constexpr (NP == 2) {
        const float gmax = __uint_as_float(absmax_read(p.g_absmax)) * fmaxf(fmaxf(gs0, gs1), fmaxf(gs2, gs3));
        const int kg_ = f16_scale_exp(__float_as_uint(gmax) + (p.gscale_bg > 0 ? 0x00800000u : 0u));    // (product rounded: one binade of slack)
        const int kx_ = f16_scale_exp(absmax_read(p.x_absmax));
        g_scale = pow2f(kg_); g_unscale = pow2f(-kg_);
        x_scale = pow2f(kx_); x_unscale = pow2f(-kx_);
    }
    if constexpr (NP == 4) {
        const float gmax = __uint_as_float(absmax_read(p.g_absmax)) * fmaxf(fmaxf(gs0, gs1), fmaxf(gs2, gs3));
        const int kg_ = f8_scale_exp(__float_as_uint(gmax) + (p.gscale_bg > 0 ? 0x00800000u : 0u));
        const int kx_ = f8_scale_exp(absmax_read(p.x_absmax));
        g_scale = pow2f(kg_); g_unscale = pow2f(-kg_);
        x_scale = pow2f(kx_); x_unscale = pow2f(-kx_);
    }
    // a pair of values rounded to e4m3: two bytes
    auto q8_pair = [](float v0, float v1) { return (unsigned short)(__builtin_amdgcn_cvt_pk_fp8_f32(v0, v1, 0, false) & 0xffff); };
    // pair np of a row: halfword (np & 3) of the row's 8-byte half np >> 2, the halves swapped on rows with bit 4 set
    auto h8 = [](int row, int pair) { return ((((pair >> 2) ^ (row >> 4)) & 1) << 2) | (pair & 3); };

    float2 greg[G_PT], xreg[X_PT];
    auto load_tiles = [&](int nb) {
        const int n = nb + 2 * np;                 // even; n + 1 is in the same image and output row
        const bool ok = n < n_end;                 // n_end is even as well
        const int nn = ok ? n : 0;
        const int b = fastdiv(nn, p.q_mul, p.q_s1, p.q_s2), q = nn - b * Q;
        const int oh = fastdiv(q, p.ow_mul, p.ow_s1, p.ow_s2), ow = q - oh * p.OW;
        const float* gp = p.gy + (long long)b * p.gy_bs + q;
        if (p.gscale_bg > 0) {       // group of batch element b (at most 4 groups): compares, no division in the hot loop
            const int bg = p.gscale_bg;
            gsc = gs0;
            gsc = b >= bg ? gs1 : gsc;
            gsc = b >= 2 * bg ? gs2 : gsc;
            gsc = b >= 3 * bg ? gs3 : gsc;
        }
#pragma unroll
        for (int i = 0; i < G_PT; ++i) {
            const int m = m0 + sub + 32 * i;
            const bool v = ok && m < p.M;
            const float2 t = *reinterpret_cast<const float2*>(gp + (v ? (long long)m * Q : 0));
            greg[i] = v ? t : make_float2(0.0f, 0.0f);
        }
        const int iy0 = oh * p.stride, ix0 = ow * p.stride;
        const unsigned base = (unsigned)(4 * ((long long)b * p.x_bs + (long long)iy0 * p.W + ix0));
#pragma unroll
        for (int i = 0; i < X_PT; ++i) {
            const bool vy = ok && (unsigned)(iy0 + xdy[i]) < (unsigned)p.H;
            const bool v0 = vy && (unsigned)(ix0 + xdx[i]) < (unsigned)p.W;
            const bool v1 = vy && (unsigned)(ix0 + p.stride + xdx[i]) < (unsigned)p.W;
            const unsigned o = base + (unsigned)xoff[i];
            // an out-of-range voffset makes the buffer load return 0 without touching memory (zero padding)
            xreg[i].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)(v0 ? o : 0x80000000u), 0, 0));
            xreg[i].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)(v1 ? o + 4u * (unsigned)p.stride : 0x80000000u), 0, 0));
        }
    };
    // dword column of this thread's pair inside its rows: rows sub + 32 i all have bit 3 of `sub`
    const int wcol = (((np >> 2) ^ ((sub >> 3) & 1)) << 2) | (np & 3);
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < G_PT; ++i) {
            if constexpr (NP == 2) {
                unsigned h, l;
                split2_f16_pair(greg[i].x * gsc * g_scale, greg[i].y * gsc * g_scale, h, l);
                Gs[buf][0][sub + 32 * i][wcol] = h;
                Gs[buf][NP - 1][sub + 32 * i][wcol] = l;
            } else if constexpr (NP == 3) {
                unsigned h, m, l;
                split3_trunc_pair(greg[i].x * gsc, greg[i].y * gsc, h, m, l);
                Gs[buf][0][sub + 32 * i][wcol] = h;
                Gs[buf][NP - 2][sub + 32 * i][wcol] = m;
                Gs[buf][NP - 1][sub + 32 * i][wcol] = l;
            } else if constexpr (NP == 4) {
                G8[buf][sub + 32 * i][h8(sub + 32 * i, np)] = q8_pair(greg[i].x * gsc * g_scale, greg[i].y * gsc * g_scale);
            } else {
                Gs[buf][0][sub + 32 * i][wcol] = round_bf16_pair(greg[i].x * gsc, greg[i].y * gsc);
            }
        }
#pragma unroll
        for (int i = 0; i < X_PT; ++i) {
            if constexpr (NP == 2) {
                unsigned h, l;
                split2_f16_pair(xreg[i].x * x_scale, xreg[i].y * x_scale, h, l);
                Xs[buf][0][sub + 32 * i][wcol] = h;
                Xs[buf][NP - 1][sub + 32 * i][wcol] = l;
            } else if constexpr (NP == 3) {
                unsigned h, m, l;
                split3_trunc_pair(xreg[i].x, xreg[i].y, h, m, l);
                Xs[buf][0][sub + 32 * i][wcol] = h;
                Xs[buf][NP - 2][sub + 32 * i][wcol] = m;
                Xs[buf][NP - 1][sub + 32 * i][wcol] = l;
            } else if constexpr (NP == 4) {
                X8[buf][sub + 32 * i][h8(sub + 32 * i, np)] = q8_pair(xreg[i].x * x_scale, xreg[i].y * x_scale);
            } else {
                Xs[buf][0][sub + 32 * i][wcol] = round_bf16_pair(xreg[i].x, xreg[i].y);
            }
        }
    };

    const int nsteps = (n_end - n_begin + WB_BK - 1) / WB_BK;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int rhalf = lrow ^ ((lcol >> 3) & 1);          // fragment rows are tile_row0 + lcol with tile_row0 % 32 == 0
    // same software pipeline as conv_igemm_bx6_kernel: the next tile is split and written, and the loads of the one after
    // it re-issued, between the two halves of a stage's MFMAs
    if (nsteps > 0) {
        load_tiles(n_begin);
        store_tiles(0);
        load_tiles(n_begin + WB_BK);                 // beyond n_end: every lane masked, nothing is read
    }
    __syncthreads();
    constexpr int PROD = NP == 3 ? 6 : (NP == 2 ? 3 : 1);
    constexpr int NMF = TM * TN * PROD, HALF = NMF / 2;
    using frag_t = typename std::conditional<NP == 2, f16x8, bf16x8>::type;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        frag_t a[TM][NPL], b[TN][NPL];
        long a8[TM], b8[TN];
        if constexpr (NP == 4) {
            const int half8 = (lrow ^ (lcol >> 4)) & 1;          // (tile rows start at multiples of 32: bit 4 of the row = bit 4 of lcol)
#pragma unroll
            for (int i = 0; i < TM; ++i) a8[i] = *reinterpret_cast<const long*>(&G8[buf][(wm * TM + i) * 32 + lcol][half8 * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b8[j] = *reinterpret_cast<const long*>(&X8[buf][(wn * TN + j) * 32 + lcol][half8 * 4]);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < NPL; ++q) a[i][q] = *reinterpret_cast<const frag_t*>(&Gs[buf][q][(wm * TM + i) * 32 + lcol][rhalf * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < NPL; ++q) b[j][q] = *reinterpret_cast<const frag_t*>(&Xs[buf][q][(wn * TN + j) * 32 + lcol][rhalf * 4]);
        }
        auto mfmas = [&](int lo, int hi) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int base = (i * TN + j) * PROD;
                    if constexpr (NP == 2) {
                        if (base + 0 >= lo && base + 0 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);   // l h
                        if (base + 1 >= lo && base + 1 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);   // h l
                        if (base + 2 >= lo && base + 2 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);   // h h
                    } else if constexpr (NP == 3) {
                        if (base + 0 >= lo && base + 0 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][NP - 1], b[j][0], acc[i][j], 0, 0, 0);   // l h
                        if (base + 1 >= lo && base + 1 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][NP - 1], acc[i][j], 0, 0, 0);   // h l
                        if (base + 2 >= lo && base + 2 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][NP - 2], b[j][NP - 2], acc[i][j], 0, 0, 0);   // m m
                        if (base + 3 >= lo && base + 3 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][NP - 2], b[j][0], acc[i][j], 0, 0, 0);   // m h
                        if (base + 4 >= lo && base + 4 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][NP - 2], acc[i][j], 0, 0, 0);   // h m
                        if (base + 5 >= lo && base + 5 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);   // h h
                    } else if constexpr (NP == 4) {
                        if (base >= lo && base < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a8[i], b8[j], acc[i][j], 0, 0, 0);
                    } else {
                        if (base >= lo && base < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                    }
                }
        };
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, HALF);
        __builtin_amdgcn_sched_barrier(0);
        store_tiles(buf ^ 1);                               // tile s + 1
        load_tiles(n_begin + (s + 2) * WB_BK);              // tile s + 2
        __builtin_amdgcn_sched_barrier(0);
        mfmas(HALF, NMF);
        __syncthreads();
    }
    if constexpr (NP == 2 || NP == 4) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] * g_unscale) * x_unscale;
    }
    wgrad_epilogue<WGM, WGN, TM, TN>(p, acc, r0, m0, wm, wn, lane, wid, tid);
}

// out = (sum_z slab[z]) * inv_scale;  partial[block] = this block's share of <sum_z slab[z], w_ref>.
// ZP = 1: one thread per element walks all slabs.  ZP = 4: four z-groups per element (many slabs, few elements: the
// 1x1 / attention layers), combined through LDS in a fixed order - results stay bit-reproducible.
// Stacked calls (groups > 1; the slabs of call k are z = k zper ... (k + 1) zper - 1, each already weighted by 1 / sigma_k):
// the block emits one partial of <G_k / sigma_k, W_bar> PER CALL (partial[k * nblocks + bid]) - what the spectral-norm backward
// of stacked calls needs for d(sigma_k), from slab values this pass reads anyway (the activation-side dots <gy_k, y_k - b>
// it replaces read both activations of every layer once more).
// V = 4: four consecutive elements per thread (16-byte accesses, n % 4 == 0) - the same additions per element in the same order
// as V = 1 (whose 4-byte accesses in 64-byte runs reached 1.5 TB/s on the 150 MB of a generator pass's slabs).
template <int V>
__device__ __forceinline__ void slab_load(const float* p, float (&v)[V]) {
    if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else v[0] = *p;
}

template <int ZP, int V>
__device__ __forceinline__ void slab_reduce_groups(const float* __restrict__ slab, float* __restrict__ out, int64_t n, int nsplit,
                                                   const float* __restrict__ w_ref, double* __restrict__ partial, int bid, int nblocks,
                                                   int groups, int zper, float* zbuf, double* gscratch) {
    float (*gzsum)[ZP][256 / ZP][V] = reinterpret_cast<float (*)[ZP][256 / ZP][V]>(zbuf);          // [4][ZP][256 / ZP][V]
    constexpr int TPB = 256 / ZP, EPB = TPB * V;
    const int ex = threadIdx.x % TPB, ez = threadIdx.x / TPB;
    const int64_t stride = (int64_t)nblocks * EPB;
    double dot[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i0 = (int64_t)bid * EPB; i0 < n; i0 += stride) {
        const int64_t i = i0 + V * ex;
        float acc[4][V];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < V; ++e) acc[k][e] = 0.0f;
        if (i < n) {
            // z walks ALL slabs, call-major, eight loads in flight; slab z belongs to call z / zper and is added to that call's sum
            // (the other calls' sums take + 0.0f: exact), in z order within each call
            const float* __restrict__ sp = slab + i;
            for (int z = ez; z < nsplit; z += 8 * ZP) {
                float v[8][V];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (z + q * ZP < nsplit) slab_load<V>(sp + (int64_t)(z + q * ZP) * n, v[q]);
                    else
#pragma unroll
                        for (int e = 0; e < V; ++e) v[q][e] = 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int zz = z + q * ZP;
                    const int k = (zz >= zper) + (zz >= 2 * zper) + (zz >= 3 * zper);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        acc[0][e] += k == 0 ? v[q][e] : 0.0f;
                        acc[1][e] += k == 1 ? v[q][e] : 0.0f;
                        acc[2][e] += k == 2 ? v[q][e] : 0.0f;
                        acc[3][e] += k == 3 ? v[q][e] : 0.0f;
                    }
                }
            }
        }
        if (ZP > 1) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < V; ++e) gzsum[k][ez][ex][e] = acc[k][e];
            __syncthreads();
            if (ez == 0)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        float a = 0.0f;
#pragma unroll
                        for (int g = 0; g < ZP; ++g) a += gzsum[k][g][ex][e];
                        acc[k][e] = a;
                    }
        }
        if (ez == 0 && i < n) {
            float o[V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const double w = w_ref ? (double)w_ref[i + e] : 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) dot[k] += (double)acc[k][e] * w;
                o[e] = ((acc[0][e] + acc[1][e]) + acc[2][e]) + acc[3][e];          // calls beyond `groups` contribute + 0.0f: exact
            }
            if constexpr (V == 4) *reinterpret_cast<float4*>(out + i) = make_float4(o[0], o[1], o[2], o[3]);
            else out[i] = o[0];
        }
    }
    if (partial) {
        for (int k = 0; k < groups; ++k) {
            const double t = block_sum<double>(dot[k], gscratch);
            if (threadIdx.x == 0) partial[(int64_t)k * nblocks + bid] = t;
        }
    }
}

template <int ZP, int V>
__device__ __forceinline__ void slab_reduce_body_v(const float* __restrict__ slab, float* __restrict__ out, int64_t n, int nsplit,
                                                   const float* __restrict__ w_ref, const float* __restrict__ inv_scale,
                                                   double* __restrict__ partial, int bid, int nblocks, int groups, int zper,
                                                   float* zbuf, double* scratch) {
    if (groups > 1) {
        slab_reduce_groups<ZP, V>(slab, out, n, nsplit, w_ref, partial, bid, nblocks, groups, zper, zbuf, scratch);
        return;
    }
    float (*zsum)[256 / ZP][V] = reinterpret_cast<float (*)[256 / ZP][V]>(zbuf);          // [ZP][256 / ZP][V]
    const float sc = inv_scale ? inv_scale[0] : 1.0f;
    constexpr int TPB = 256 / ZP, EPB = TPB * V;        // threads / elements per block pass
    const int ex = threadIdx.x % TPB, ez = threadIdx.x / TPB;
    const int64_t stride = (int64_t)nblocks * EPB;
    double dot = 0.0;
    for (int64_t i0 = (int64_t)bid * EPB; i0 < n; i0 += stride) {
        const int64_t i = i0 + V * ex;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.0f;
        if (i < n) {
            // eight slabs' loads in flight, added in z order
            const float* __restrict__ sp = slab + i;
            for (int z = ez; z < nsplit; z += 8 * ZP) {
                float v[8][V];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (z + q * ZP < nsplit) slab_load<V>(sp + (int64_t)(z + q * ZP) * n, v[q]);
                    else
#pragma unroll
                        for (int e = 0; e < V; ++e) v[q][e] = 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += v[q][e];      // + 0.0f beyond nsplit: exact
            }
        }
        if (ZP > 1) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < V; ++e) zsum[ez][ex][e] = acc[e];
            __syncthreads();
            if (ez == 0)
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    float a = 0.0f;
#pragma unroll
                    for (int g = 0; g < ZP; ++g) a += zsum[g][ex][e];
                    acc[e] = a;
                }
        }
        if (ez == 0 && i < n) {
            if (w_ref)
#pragma unroll
                for (int e = 0; e < V; ++e) dot += (double)acc[e] * (double)w_ref[i + e];
            if constexpr (V == 4) *reinterpret_cast<float4*>(out + i) = make_float4(acc[0] * sc, acc[1] * sc, acc[2] * sc, acc[3] * sc);
            else out[i] = acc[0] * sc;
        }
    }
    if (partial) {
        dot = block_sum<double>(dot, scratch);
        if (threadIdx.x == 0) partial[bid] = dot;
    }
}

template <int ZP>
__device__ __forceinline__ void slab_reduce_body(const float* __restrict__ slab, float* __restrict__ out, int64_t n, int nsplit,
                                                 const float* __restrict__ w_ref, const float* __restrict__ inv_scale,
                                                 double* __restrict__ partial, int bid, int nblocks, int groups = 0, int zper = 0) {
    // one LDS area for whichever form runs: [4 calls][256 threads][4 values]
    __shared__ __attribute__((aligned(16))) float zbuf[4 * 256 * 4];
    __shared__ double scratch[16];
    const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(slab) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    // (ZP = 16 - a handful of elements under hundreds of slabs - stays scalar: its sixteen-way LDS sums times four values spill)
    if constexpr (ZP <= 4) {
        if (vec) {
            slab_reduce_body_v<ZP, 4>(slab, out, n, nsplit, w_ref, inv_scale, partial, bid, nblocks, groups, zper, zbuf, scratch);
            return;
        }
    }
    slab_reduce_body_v<ZP, 1>(slab, out, n, nsplit, w_ref, inv_scale, partial, bid, nblocks, groups, zper, zbuf, scratch);
}

template <int ZP>
__global__ void __launch_bounds__(256, 4) slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int64_t n,
                                                          int nsplit, const float* __restrict__ w_ref,
                                                          const float* __restrict__ inv_scale, double* __restrict__ partial,
                                                          int groups, int zper) {
    slab_reduce_body<ZP>(slab, out, n, nsplit, w_ref, inv_scale, partial, blockIdx.x, gridDim.x, groups, zper);
}

// The split reductions of ALL weight gradients of one backward pass in one launch: a weight gradient only feeds a parameter
// gradient, so its slab sum can wait for the end of the pass like the other finalisers (finalise.hip) - each of the ~30 per
// iteration is a launch-floor-sized kernel behind its GEMM.  Records by value; per layer the same blocks, the same z order and
// the same <G, W_bar> partials as the single launch.
struct SlabRec {
    const float* slab; float* out; const float* w_ref; const float* inv_scale; double* partial;
    long long n;
    int nsplit, zp, grid, block0, groups, zper;
};
#define SLAB_MAX 32
struct SlabBatch {
    SlabRec r[SLAB_MAX];
};

__global__ void __launch_bounds__(256, 4) slab_reduce_batch_kernel(const SlabBatch b, int nrec) {
    int k = 0;
    for (int i = 1; i < nrec; ++i)
        if ((int)blockIdx.x >= b.r[i].block0) k = i;          // block0 ascending
    const SlabRec& r = b.r[k];
    const int bid = (int)blockIdx.x - r.block0;
    if (r.zp == 16) slab_reduce_body<16>(r.slab, r.out, r.n, r.nsplit, r.w_ref, r.inv_scale, r.partial, bid, r.grid, r.groups, r.zper);
    else if (r.zp == 4) slab_reduce_body<4>(r.slab, r.out, r.n, r.nsplit, r.w_ref, r.inv_scale, r.partial, bid, r.grid, r.groups, r.zper);
    else slab_reduce_body<1>(r.slab, r.out, r.n, r.nsplit, r.w_ref, r.inv_scale, r.partial, bid, r.grid, r.groups, r.zper);
}

// groups > 1 (stacked calls whose per-call <G_k, W_bar> the reduction is to emit): every call's share of the reduction is split on
// its own - nsplit = groups x zper slabs, none crossing a call boundary, at least one slab per call
static void wgrad_plan(const ConvGeom& g, int* bm, int* nsplit, int* chunk, int* tiles_out, int groups = 0, int* zper_out = nullptr) {
    *bm = pick_bm(g.M);
    const int R = g.C * g.KH * g.KW;
    // tall 192 x 128 tiles (96 x 64 per wave: a third more MFMAs per gathered and split element, two blocks per CU) for the wide
    // layers, as in the forward kernels (same-box A/B of the step: 9.031 / 9.039 -> 9.010 / 8.997 ms); paired-load kernels only
    if (g.M % 192 == 0 && ((g.OH * g.OW) & 1) == 0 && (g.OW & 1) == 0 && knob_int("LOCATE_WG_TALL", 1) && !path_disabled("wbx6") && (int64_t)(g.M / 192) * ((R + 127) / 128) >= knob_int("LOCATE_WG_TALL_MIN_TILES", 24)) *bm = 192;
    const int64_t slots = *bm == 192 ? 512 : 768;
    const int64_t tiles = (int64_t)((g.M + *bm - 1) / *bm) * ((R + 127) / 128);
    if (groups > 1) {
        const int64_t Ng = (int64_t)(g.B / groups) * g.OH * g.OW;
        const int wg_min = knob_int("LOCATE_WG_MIN_CHUNK", 64);
        const int64_t max_split = Ng >= 2 * wg_min ? Ng / wg_min : 1;
        int64_t best_s = 1;
        double best_cost = 1e300;
        for (int64_t s_ = 1; s_ <= max_split && s_ * groups <= 512; ++s_) {
            int64_t ch = (Ng + s_ - 1) / s_;
            ch = (ch + WG_BK - 1) / WG_BK * WG_BK;
            const int64_t ns = groups * ((Ng + ch - 1) / ch);
            const int64_t rounds = (tiles * ns + slots - 1) / slots;
            const double cost = (double)rounds * (double)ch + 96.0 * (double)ns * (double)tiles / (double)slots;
            if (cost < best_cost * 0.999) { best_cost = cost; best_s = s_; }
        }
        int64_t ch = (Ng + best_s - 1) / best_s;
        ch = (ch + WG_BK - 1) / WG_BK * WG_BK;
        *chunk = (int)ch;
        const int zper = (int)((Ng + ch - 1) / ch);
        *nsplit = groups * zper;
        if (zper_out) *zper_out = zper;
        if (tiles_out) *tiles_out = (int)tiles;
        return;
    }
    const int64_t N = (int64_t)g.B * g.OH * g.OW;
    // Split the reduction over s blocks per tile so that the launch fills whole rounds of the 768 resident blocks
    // (256 CUs x 3): cost(s) = rounds(s) x reduction elements per block, plus the slab traffic of s > 1 expressed in
    // the same unit (one output tile written and re-read ~ 96 reduction elements of MFMA time).
    const int wg_min = knob_int("LOCATE_WG_MIN_CHUNK", 64);
    const int64_t max_split = N >= 2 * wg_min ? N / wg_min : 1;     // at least 64 reduction elements per block (the split
    // reductions of a pass run as ONE batched launch at its end, so a deeper split costs slab traffic only: the deep layers' 8 - 24
    // tiles x 3 splits of 256 were latency chains of 16 steps on a tenth of the chip; same-box A/B of the step: 256 -> 9.13 / 9.12,
    // 128 -> 9.06 / 9.05, 64 -> 9.05 / 9.03, 32 -> 9.06 / 9.06 ms)
    int64_t best_s = 1;
    double best_cost = 1e300;
    for (int64_t s_ = 1; s_ <= max_split && s_ <= 512; ++s_) {
        int64_t ch = (N + s_ - 1) / s_;
        ch = (ch + WG_BK - 1) / WG_BK * WG_BK;
        const int64_t ns = (N + ch - 1) / ch;
        const int64_t rounds = (tiles * ns + slots - 1) / slots;
        const double cost = (double)rounds * (double)ch + (ns > 1 ? 96.0 * (double)ns * (double)tiles / (double)slots : 0.0);
        if (cost < best_cost * 0.999) { best_cost = cost; best_s = s_; }
    }
    int64_t ch = (N + best_s - 1) / best_s;
    ch = (ch + WG_BK - 1) / WG_BK * WG_BK;
    *chunk = (int)ch;
    *nsplit = (int)((N + ch - 1) / ch);
    if (zper_out) *zper_out = *nsplit;
    if (tiles_out) *tiles_out = (int)tiles;
}

static int wgrad_reduce_zp(int nsplit, int64_t n) {
    // few outputs, many slabs: 16 threads share one output element - below 4096 elements only: from there on the four-way form with
    // its 16-byte accesses is faster (same-call A/B of the step, threshold 65536 / 16384 / 4096 / 1024: 8.97, 8.96 / 8.95, 8.96 / 8.92,
    // 8.94 / 8.94, 8.96 ms)
    if (nsplit >= 64 && n < knob_int("LOCATE_ZP16_MAX_N", 1 << 12)) return 16;
    return (nsplit >= 16 && n < (1 << 20)) ? 4 : 1;
}
static int wgrad_reduce_grid(int64_t n, int nsplit) {
    const int epb = 256 / wgrad_reduce_zp(nsplit, n);
    int g = stream_grid(n, epb);
    return g > 2048 ? 2048 : g;
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of 1x1 stride-1 layers with few channels on both sides (<= 128: the attention gates' convs, the skip
// branches' 1x1 convs, the generator's head) over many pixels:  gw[m][c] = sum_{b, p} gy[b][m][p] x[b][c][p].
// Both operands are contiguous along the reduction index p, which is exactly the MFMA fragment layout (lane (r, h) holds
// k = 8h .. 8h + 7 of row r): every wave loads its fragments straight from global memory - two 16-byte loads per fragment
// row - splits them in registers and multiplies; no LDS image, no barrier, no gather tables in the loop.  These launches
// are HBM streams (a 64 x 64 output tile per wave against 8 KB of operands per 16 pixels); the general kernel above, built
// for wide layers, ran them at 15-25 % of that.  A wave owns one (row tile, column tile) and a contiguous run of 16-pixel
// steps; the four waves of a block add their tiles in wave order through LDS and write one slab, summed (with 1/sigma and
// the <G, W_bar> partials) by slab_reduce_kernel like every split weight gradient.
// ---------------------------------------------------------------------------------------------
struct PwParams {
    const float* x;
    const float* gy;
    float* slab;
    const float* inv_scale;
    long long x_bs, gy_bs;
    int B, C, M, P;            // P = H * W
    int N;                     // B * P
    int steps, chunk;          // 16-pixel steps in all, steps per wave
    int tiles_c;
    int gscale_bg, gscale_stride;
    const unsigned* x_absmax;  // NP = 2 (two scaled fp16 pieces, three MFMAs - conv_igemm_bx6_kernel's form): largest magnitudes of
    const unsigned* g_absmax;  // x and of gy, AMAX_WORDS words each
};

template <int TM, int TN, int NP>
__global__ void __launch_bounds__(256) pw_wgrad_kernel(const PwParams p) {
    __shared__ float red[3][TM * TN * 16][64];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lrow = lane >> 5, lcol = lane & 31;
    const int tm = blockIdx.y / p.tiles_c, tc = blockIdx.y - tm * p.tiles_c;
    const int m0 = tm * (TM * 32), c0 = tc * (TN * 32);
    const int z = blockIdx.x * 4 + wid;
    const int s_begin = z * p.chunk;
    int s_end = s_begin + p.chunk;
    if (s_end > p.steps) s_end = p.steps;

    float gs0 = 1.0f, gs1 = 1.0f, gs2 = 1.0f, gs3 = 1.0f;
    if (p.gscale_bg > 0) {
        const int ng = p.B / p.gscale_bg;
        gs0 = p.inv_scale[0];
        gs1 = ng > 1 ? p.inv_scale[p.gscale_stride] : 1.0f;
        gs2 = ng > 2 ? p.inv_scale[2 * p.gscale_stride] : 1.0f;
        gs3 = ng > 3 ? p.inv_scale[3 * p.gscale_stride] : 1.0f;
    }
    // rows of this lane's fragments (clamped to a valid row; masked when beyond the tensor)
    long long arow[TM], brow[TN];
    bool aok[TM], bok[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 32 + lcol;
        aok[i] = m < p.M;
        arow[i] = (long long)(aok[i] ? m : 0) * p.P;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = c0 + j * 32 + lcol;
        bok[j] = c < p.C;
        brow[j] = (long long)(bok[j] ? c : 0) * p.P;
    }
    const DivU32 dp((unsigned)p.P);
    float x_scale = 1.0f, g_scale = 1.0f, x_unscale = 1.0f, g_unscale = 1.0f;
    if constexpr (NP == 2) {          // powers of two into fp16's range; the exact inverses go back in after the loop
        const int kx = f16_scale_exp(absmax_read(p.x_absmax)), kg = f16_scale_exp(absmax_read(p.g_absmax));
        x_scale = pow2f(kx); g_scale = pow2f(kg);
        x_unscale = pow2f(-kx); g_unscale = pow2f(-kg);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ca[TM][2], cb[TN][2], na[TM][2], nb[TN][2];
    float csc = 1.0f, nsc = 1.0f;
    auto load = [&](int s, float4 (&fa)[TM][2], float4 (&fb)[TN][2], float& sc) {
        // this lane's eight pixels n .. n + 7 of step s (P % 8 == 0: they lie in one image)
        const unsigned n = (unsigned)s * 16u + 8u * (unsigned)lrow;
        const bool ok = s < s_end && n < (unsigned)p.N;
        unsigned b, q;
        dp.divmod(ok ? n : 0u, b, q);
        const float* gp = p.gy + (long long)b * p.gy_bs + q;
        const float* xp = p.x + (long long)b * p.x_bs + q;
        sc = 1.0f;
        if (p.gscale_bg > 0) {
            const int bg = p.gscale_bg;
            sc = gs0;
            sc = (int)b >= bg ? gs1 : sc;
            sc = (int)b >= 2 * bg ? gs2 : sc;
            sc = (int)b >= 3 * bg ? gs3 : sc;
        }
        if (!ok) sc = 0.0f;                                  // beyond this wave's run: the loads below are valid, the values dropped
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float4* g4 = reinterpret_cast<const float4*>(gp + arow[i]);
            fa[i][0] = g4[0];
            fa[i][1] = g4[1];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float4* x4 = reinterpret_cast<const float4*>(xp + brow[j]);
            fb[j][0] = x4[0];
            fb[j][1] = x4[1];
        }
    };
    if (s_begin < s_end) load(s_begin, ca, cb, csc);
    for (int s = s_begin; s < s_end; ++s) {
        load(s + 1, na, nb, nsc);
        using pfrag_t = typename std::conditional<NP == 2, f16x8, bf16x8>::type;
        pfrag_t a[TM][NP], b[TN][NP];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float w = aok[i] ? csc : 0.0f;
            float v[8] = {ca[i][0].x * w, ca[i][0].y * w, ca[i][0].z * w, ca[i][0].w * w,
                          ca[i][1].x * w, ca[i][1].y * w, ca[i][1].z * w, ca[i][1].w * w};
            if constexpr (NP == 2) {
                uint4 h, l;
                split2_f16x8(v, g_scale, h, l);
                a[i][0] = *reinterpret_cast<pfrag_t*>(&h);
                a[i][NP - 1] = *reinterpret_cast<pfrag_t*>(&l);
            } else if constexpr (NP == 3) {
                uint4 h, m, l;
                split3_trunc_x8(v, h, m, l);
                a[i][0] = *reinterpret_cast<pfrag_t*>(&h);
                a[i][NP - 2] = *reinterpret_cast<pfrag_t*>(&m);
                a[i][NP - 1] = *reinterpret_cast<pfrag_t*>(&l);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) a[i][0][e] = (__bf16)v[e];
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float w = (bok[j] && csc != 0.0f) ? 1.0f : 0.0f;
            float v[8] = {cb[j][0].x * w, cb[j][0].y * w, cb[j][0].z * w, cb[j][0].w * w,
                          cb[j][1].x * w, cb[j][1].y * w, cb[j][1].z * w, cb[j][1].w * w};
            if constexpr (NP == 2) {
                uint4 h, l;
                split2_f16x8(v, x_scale, h, l);
                b[j][0] = *reinterpret_cast<pfrag_t*>(&h);
                b[j][NP - 1] = *reinterpret_cast<pfrag_t*>(&l);
            } else if constexpr (NP == 3) {
                uint4 h, m, l;
                split3_trunc_x8(v, h, m, l);
                b[j][0] = *reinterpret_cast<pfrag_t*>(&h);
                b[j][NP - 2] = *reinterpret_cast<pfrag_t*>(&m);
                b[j][NP - 1] = *reinterpret_cast<pfrag_t*>(&l);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) b[j][0][e] = (__bf16)v[e];
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (NP == 2) {          // smallest terms first: l h, h l, h h
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][NP - 1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][NP - 1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                } else if constexpr (NP == 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][NP - 1], b[j][0], acc[i][j], 0, 0, 0);        // l h
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][NP - 1], acc[i][j], 0, 0, 0);        // h l
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][NP - 2], b[j][NP - 2], acc[i][j], 0, 0, 0);   // m m
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][NP - 2], b[j][0], acc[i][j], 0, 0, 0);        // m h
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][NP - 2], acc[i][j], 0, 0, 0);        // h m
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);             // h h
                } else {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
            }
#pragma unroll
        for (int i = 0; i < TM; ++i) { ca[i][0] = na[i][0]; ca[i][1] = na[i][1]; }
#pragma unroll
        for (int j = 0; j < TN; ++j) { cb[j][0] = nb[j][0]; cb[j][1] = nb[j][1]; }
        csc = nsc;
    }
    if constexpr (NP == 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] * g_unscale) * x_unscale;
    }
    // the block's four tiles, added in wave order; wave 0 writes the slab
    if (wid > 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wid - 1][(i * TN + j) * 16 + r][lane] = acc[i][j][r];
    }
    __syncthreads();
    if (wid != 0) return;
    float* out = p.slab + (long long)blockIdx.x * p.M * p.C;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = c0 + j * 32 + lcol;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                const int e = (i * TN + j) * 16 + r;
                const float v = ((acc[i][j][r] + red[0][e][lane]) + red[1][e][lane]) + red[2][e][lane];
                if (m < p.M && c < p.C) out[(long long)m * p.C + c] = v;
            }
        }
}

struct PwPlan {
    bool ok;
    int tm, tn, tiles_m, tiles_c, steps, chunk, nslab, zper;
};

// groups > 1: the slabs are to stay inside one stacked call each (see wgrad_plan); q.zper slabs per call, or q.zper = 0 when the
// call length does not divide into whole blocks of four wave runs (the caller then takes the dots on the activation side)
static PwPlan pw_plan(const ConvGeom& g, int groups = 0) {
    PwPlan q;
    q.zper = 0;
    const long long P = (long long)g.H * g.W;
    q.ok = !path_disabled("pwgrad") && g.KH == 1 && g.KW == 1 && g.stride == 1 && g.pad_h == 0 && g.pad_w == 0 && g.OH == g.H &&
           g.OW == g.W && (P % 8) == 0 && g.M <= 128 && g.C <= 128 && (long long)g.B * P >= 4096 && (long long)g.B * P < (1ll << 31);
    q.tm = g.M <= 32 ? 1 : 2;
    q.tn = g.C <= 32 ? 1 : 2;
    q.tiles_m = (g.M + q.tm * 32 - 1) / (q.tm * 32);
    q.tiles_c = (g.C + q.tn * 32 - 1) / (q.tn * 32);
    const long long N = (long long)g.B * P;
    q.steps = (int)((N + 15) / 16);
    // ~2048 waves (two per SIMD) over all tiles, at least 8 steps each
    const int tiles = q.tiles_m * q.tiles_c;
    int waves = 2048 / tiles;
    if (waves < 4) waves = 4;
    int chunk = (q.steps + waves - 1) / waves;
    if (chunk < 8) chunk = 8;
    if (groups > 1 && q.ok) {
        const long long per = ((long long)(g.B / groups) * P) / 16;          // 16-pixel steps per call
        if (((long long)(g.B / groups) * P) % 64 != 0) return q;             // zper stays 0
        while (chunk > 4 && per % (4ll * chunk) != 0) --chunk;
        if (per % (4ll * chunk) != 0) return q;
        q.chunk = chunk;
        q.zper = (int)(per / (4ll * chunk));
        q.nslab = groups * q.zper;
        return q;
    }
    q.chunk = chunk;
    const int nw = (q.steps + chunk - 1) / chunk;
    q.nslab = (nw + 3) / 4;
    q.zper = q.nslab;
    return q;
}

// Weight gradient of the same 1x1-map layers (skinny_rows_kernel): gw[m][c] = inv_scale * sum_n gy[n][m] x[n][c], an outer-product
// sum over the 64 ... 192 batch rows.  lane = c (x[n][.] is one coalesced load), a block owns eight rows m (gy[n][m .. m + 7] is
// wave-uniform: one scalar load) and its four waves a quarter of the batch each, plain fp32 FMAs, no slab; one partial of
// <unscaled gw, W_bar> per block.
// OnePix (hw > 0): the layer maps its whole H x W input to ONE output pixel (the discriminator's last 5x5 s2 conv on a 2 x 2
// map, its 3x3 head on a 1 x 1 map: 6.5 M of D's 11.6 M parameters).  Only the taps that meet the input carry a gradient -
// 4 of 25, 1 of 9 - and that gradient is the same outer-product sum over the batch with x viewed as [B, C H W]: the kernel
// below with its columns scattered to their taps and the other taps zeroed (the general kernel multiplies through all 25 taps'
// columns on the fp32 MFMA - these layers' OW is odd - for 21 exact zeros out of 25).
struct OnePix {
    int hw, W, KH, KW, pad_h, pad_w, Cw;
};

#define SKW_MT 8          // gradient rows per block (narrow layers)
#define SKW_MT_WIDE 32    // ... of layers with >= SKW_WIDE_M rows: x is re-read by a quarter as many blocks
#define SKW_WIDE_M 128
static inline int skw_mt(int M) { return M >= SKW_WIDE_M ? SKW_MT_WIDE : SKW_MT; }
#define SKW_NC 32         // batch rows per load batch

// one finished element (row m, column j) of the gradient: scattered to its tap for a one-pixel layer; its <G, W_bar> term
__device__ __forceinline__ void skw_store(const OnePix& op, float* __restrict__ gw, const float* __restrict__ w_ref, int C, int m, int j,
                                          float v, float sc, double& dot) {
    long long o = (long long)m * C + j;
    bool inside = true;
    if (op.hw > 0) {           // column j = (c, iy, ix) of a whole input map: tap (iy + pad_h, ix + pad_w) of weight row (m, c)
        const int c = j / op.hw, pix = j - c * op.hw;
        const int iy = pix / op.W, ix = pix - iy * op.W;
        const int kh = iy + op.pad_h, kw = ix + op.pad_w;
        inside = kh < op.KH && kw < op.KW;           // pixels no tap of the single output position reaches
        o = (((long long)m * op.Cw + c) * op.KH + kh) * op.KW + kw;
    }
    if (inside) {
        if (w_ref) dot += (double)v * (double)w_ref[o];
        gw[o] = v * sc;
    }
}

// The taps no input pixel reaches get their zeros here: the block owns rows i0 .. i0 + mt - 1 of the channels its 64 columns
// span - contiguous runs of gw - and walks them with consecutive lanes on consecutive addresses, skipping the taps skw_store
// wrote (disjoint addresses: no ordering needed).  A channel whose pixels straddle two blocks is zeroed by the block that
// holds its pixel 0.
__device__ __forceinline__ void skw_zero_taps(const OnePix& op, float* __restrict__ gw, int M, int i0, int mt, int bx) {
    const int H = op.hw / op.W, taps = op.KH * op.KW;
    const int first_col = bx * 64;
    const int c_first = (first_col + op.hw - 1) / op.hw;
    int c_last = (first_col + 63) / op.hw;
    if (c_last > op.Cw - 1) c_last = op.Cw - 1;
    const int span = (c_last - c_first + 1) * taps;
    for (int t = 0; t < mt; ++t) {
        if (i0 + t >= M) break;
        float* row = gw + ((long long)(i0 + t) * op.Cw + c_first) * taps;
        for (int e = threadIdx.x; e < span; e += blockDim.x) {
            const int tap = e % taps, kh = tap / op.KW, kw = tap - kh * op.KW;
            if (!(kh >= op.pad_h && kh - op.pad_h < H && kw >= op.pad_w && kw - op.pad_w < op.W)) row[e] = 0.0f;
        }
    }
}

template <int MT>
__device__ __forceinline__ void skinny_wgrad_body(const float* __restrict__ x, long long x_bs, const float* __restrict__ gy,
                                                  long long gy_bs, float* __restrict__ gw, const float* __restrict__ w_ref,
                                                  const float* __restrict__ inv_scale, int scale_bg, int scale_stride,
                                                  double* __restrict__ partial, int N, int M, int C, const OnePix& op, int bx, int by,
                                                  int grid_x) {
    __shared__ double scratch[16];
    __shared__ float red[4][MT][64];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int j = bx * 64 + lane;
    const int i0 = by * MT;
    const bool jok = j < C;
    const float* __restrict__ xc = x + (jok ? j : 0);
    float acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = 0.0f;
    // the four waves take a quarter of the batch rows each; their partial sums are added in wave order below
    const int nq = (N + 3) / 4, nlo = wid * nq, nhi = nlo + nq < N ? nlo + nq : N;
    for (int nb = nlo; nb < nhi; nb += SKW_NC) {
        float xv[SKW_NC];
#pragma unroll
        for (int q = 0; q < SKW_NC; ++q) xv[q] = (jok && nb + q < nhi) ? xc[(long long)(nb + q) * x_bs] : 0.0f;
#pragma unroll
        for (int q = 0; q < SKW_NC; ++q) {
            if (nb + q < nhi) {
                const float* __restrict__ g = gy + (long long)(nb + q) * gy_bs + i0;          // wave-uniform: scalar loads
                // a stacked call's 1 / sigma_k goes onto the x value (one multiply per batch row instead of one per row and m)
                const float xs = scale_bg ? xv[q] * inv_scale[((nb + q) / scale_bg) * scale_stride] : xv[q];
                if (i0 + MT <= M) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t] = fmaf(g[t], xs, acc[t]);
                } else {
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t] = fmaf(i0 + t < M ? g[t] : 0.0f, xs, acc[t]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) red[wid][t][lane] = acc[t];
    __syncthreads();
    const float sc = (!scale_bg && inv_scale) ? inv_scale[0] : 1.0f;
    double dot = 0.0;
#pragma unroll
    for (int q = 0; q < MT / 4; ++q) {          // wave w finishes rows (MT / 4) w ... (MT / 4) (w + 1) - 1
        const int t = wid * (MT / 4) + q;
        if (jok && i0 + t < M) {
            const float v = ((red[0][t][lane] + red[1][t][lane]) + red[2][t][lane]) + red[3][t][lane];
            skw_store(op, gw, w_ref, C, i0 + t, j, v, sc, dot);
        }
    }
    if (op.hw > 0) skw_zero_taps(op, gw, M, i0, MT, bx);
    if (partial) {
        dot = block_sum<double>(dot, scratch);
        if (threadIdx.x == 0) partial[by * grid_x + bx] = dot;
    }
}

// Layers of SKW_WIDE_M rows or more: a 32 x 64 tile of the gradient per block, both operands staged through LDS in runs of
// SKW_WN batch rows with coalesced loads (the narrow form's per-row scalar loads of gy cost a round trip per batch row and block),
// 2 x 4 results per thread, the batch rows summed in order by ONE thread per result (no cross-wave combination).
#define SKW_WN 96
__device__ __forceinline__ void skinny_wgrad_wide(const float* __restrict__ x, long long x_bs, const float* __restrict__ gy,
                                                  long long gy_bs, float* __restrict__ gw, const float* __restrict__ w_ref,
                                                  const float* __restrict__ inv_scale, int scale_bg, int scale_stride,
                                                  double* __restrict__ partial, int N, int M, int C, const OnePix& op, int bx, int by,
                                                  int grid_x) {
    __shared__ double wscratch[16];
    __shared__ __attribute__((aligned(16))) float wbuf[SKW_WN * (SKW_MT_WIDE + 64)];          // operand stages, then the output rows of a one-pixel layer
    float (*gs)[SKW_MT_WIDE] = reinterpret_cast<float (*)[SKW_MT_WIDE]>(wbuf);
    float (*xs)[64] = reinterpret_cast<float (*)[64]>(wbuf + SKW_WN * SKW_MT_WIDE);
    const int tid = threadIdx.x;
    const int i0 = by * SKW_MT_WIDE, j0 = bx * 64;
    const int tm = tid >> 4, tc = tid & 15;              // rows i0 + 2 tm + {0, 1}, columns j0 + 4 tc + {0 .. 3}
    const int gm = tid & 31, gr = tid >> 5;              // staging: gy column / first row of this thread
    const int xc = tid & 63, xr = tid >> 6;
    const bool gok = i0 + gm < M, xok = j0 + xc < C;
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int n0 = 0; n0 < N; n0 += SKW_WN) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SKW_WN / 8; ++r) {
            const int n = n0 + gr + 8 * r;
            float v = 0.0f;
            if (gok && n < N) {
                v = gy[(long long)n * gy_bs + i0 + gm];
                if (scale_bg) v *= inv_scale[(n / scale_bg) * scale_stride];
            }
            gs[gr + 8 * r][gm] = v;
        }
#pragma unroll
        for (int r = 0; r < SKW_WN / 4; ++r) {
            const int n = n0 + xr + 4 * r;
            xs[xr + 4 * r][xc] = (xok && n < N) ? x[(long long)n * x_bs + j0 + xc] : 0.0f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < SKW_WN; ++k) {
            const float2 g = *reinterpret_cast<const float2*>(&gs[k][2 * tm]);
            const float4 xv = *reinterpret_cast<const float4*>(&xs[k][4 * tc]);
            acc[0][0] = fmaf(g.x, xv.x, acc[0][0]); acc[0][1] = fmaf(g.x, xv.y, acc[0][1]);
            acc[0][2] = fmaf(g.x, xv.z, acc[0][2]); acc[0][3] = fmaf(g.x, xv.w, acc[0][3]);
            acc[1][0] = fmaf(g.y, xv.x, acc[1][0]); acc[1][1] = fmaf(g.y, xv.y, acc[1][1]);
            acc[1][2] = fmaf(g.y, xv.z, acc[1][2]); acc[1][3] = fmaf(g.y, xv.w, acc[1][3]);
        }
    }
    const float sc = (!scale_bg && inv_scale) ? inv_scale[0] : 1.0f;
    double dot = 0.0;
    const int taps = op.KH * op.KW;
    // One-pixel layer whose 64 columns are whole channels (hw | 64) and whose weight rows take 16-byte stores: the block's output
    // - 64 / hw channels x taps floats per row, contiguous in gw - is assembled in LDS (zeros, then the useful taps scattered in)
    // and streamed out with full-width stores, a few rows per pass: scattered 4-byte stores plus a separate zeroing walk cost
    // more than the arithmetic (26 MB of D's last 5x5 layer: 61 -> measured below).
    const int cpb = op.hw > 0 && (64 % op.hw) == 0 ? 64 / op.hw : 0;
    const int c0 = bx * cpb;
    const int nch = cpb > 0 ? (c0 + cpb <= op.Cw ? cpb : op.Cw - c0) : 0;
    const int rowlen = nch * taps;
    const bool staged = cpb > 0 && nch > 0 && (rowlen & 3) == 0 && (((long long)op.Cw * taps) & 3) == 0 && (((long long)c0 * taps) & 3) == 0 &&
                        rowlen * 2 <= SKW_WN * (SKW_MT_WIDE + 64) && (reinterpret_cast<uintptr_t>(gw) & 15) == 0;
    if (staged) {
        int rpp = (SKW_WN * (SKW_MT_WIDE + 64)) / rowlen;          // rows per pass: even (a thread's two rows stay together)
        rpp = rpp > SKW_MT_WIDE ? SKW_MT_WIDE : (rpp & ~1);
        const DivU32 dq((unsigned)(rowlen / 4));
        for (int r0 = 0; r0 < SKW_MT_WIDE && i0 + r0 < M; r0 += rpp) {
            __syncthreads();
            for (int e = tid; e < rpp * rowlen / 4; e += 256) reinterpret_cast<float4*>(wbuf)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int row = 2 * tm + a;
                if (row >= r0 && row < r0 + rpp && i0 + row < M) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int jl = 4 * tc + b, cl = jl / op.hw, pix = jl - cl * op.hw;
                        const int iy = pix / op.W, ix = pix - iy * op.W;
                        const int kh = iy + op.pad_h, kw = ix + op.pad_w;
                        if (cl < nch && kh < op.KH && kw < op.KW) {
                            const int t = cl * taps + kh * op.KW + kw;
                            const float v = acc[a][b];
                            if (w_ref) dot += (double)v * (double)w_ref[((long long)(i0 + row) * op.Cw + c0) * taps + t];
                            wbuf[(row - r0) * rowlen + t] = v * sc;
                        }
                    }
                }
            }
            __syncthreads();
            int rows = M - (i0 + r0);
            if (rows > rpp) rows = rpp;
            if (rows > SKW_MT_WIDE - r0) rows = SKW_MT_WIDE - r0;
            for (int e = tid; e < rows * rowlen / 4; e += 256) {
                unsigned r, q;
                dq.divmod((unsigned)e, r, q);
                reinterpret_cast<float4*>(gw + ((long long)(i0 + r0 + (int)r) * op.Cw + c0) * taps)[q] = reinterpret_cast<const float4*>(wbuf)[e];
            }
        }
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int m = i0 + 2 * tm + a, j = j0 + 4 * tc + b;
                if (m < M && j < C) skw_store(op, gw, w_ref, C, m, j, acc[a][b], sc, dot);
            }
        if (op.hw > 0) skw_zero_taps(op, gw, M, i0, SKW_MT_WIDE, bx);
    }
    if (partial) {
        dot = block_sum<double>(dot, wscratch);
        if (threadIdx.x == 0) partial[by * grid_x + bx] = dot;
    }
}

__global__ void __launch_bounds__(256) skinny_wgrad_kernel(const float* __restrict__ x, long long x_bs, const float* __restrict__ gy,
                                                           long long gy_bs, float* __restrict__ gw, const float* __restrict__ w_ref,
                                                           const float* __restrict__ inv_scale, int scale_bg, int scale_stride,
                                                           double* __restrict__ partial, int N, int M, int C, OnePix op) {
    if (M >= SKW_WIDE_M)
        skinny_wgrad_wide(x, x_bs, gy, gy_bs, gw, w_ref, inv_scale, scale_bg, scale_stride, partial, N, M, C, op, blockIdx.x, blockIdx.y,
                          gridDim.x);
    else
        skinny_wgrad_body<SKW_MT>(x, x_bs, gy, gy_bs, gw, w_ref, inv_scale, scale_bg, scale_stride, partial, N, M, C, op, blockIdx.x,
                                  blockIdx.y, gridDim.x);
}

// The same for ALL such layers of one backward pass in ONE launch (the style chain's links, the channel gates' squeeze convs,
// the discriminator's 1x1-map layers, its last 5x5 conv and its head): records by value in the kernel arguments like the other
// end-of-pass finalisers (finalise.hip) - every one of these launches is a ten-microsecond walk over the batch rows by a
// handful of blocks; together they fill the chip once.  Arithmetic and summation order per layer are those of the single
// launch (same body, same block shape).
struct SkwRec {
    const float* x; const float* gy; float* gw; const float* w_ref; const float* inv_scale; double* partial;
    long long x_bs, gy_bs;
    int scale_bg, scale_stride, N, M, C, grid_x, block0;
    OnePix op;
};
#define SKW_MAX 24
struct SkwBatch {
    SkwRec r[SKW_MAX];
};

__global__ void __launch_bounds__(256) skinny_wgrad_batch_kernel(const SkwBatch b, int n) {
    int k = 0;
    for (int i = 1; i < n; ++i)
        if ((int)blockIdx.x >= b.r[i].block0) k = i;          // block0 ascending
    const SkwRec& r = b.r[k];
    const int local = (int)blockIdx.x - r.block0;
    const int by = local / r.grid_x, bx = local - by * r.grid_x;
    if (r.M >= SKW_WIDE_M)
        skinny_wgrad_wide(r.x, r.x_bs, r.gy, r.gy_bs, r.gw, r.w_ref, r.inv_scale, r.scale_bg, r.scale_stride, r.partial, r.N, r.M, r.C,
                          r.op, bx, by, r.grid_x);
    else
        skinny_wgrad_body<SKW_MT>(r.x, r.x_bs, r.gy, r.gy_bs, r.gw, r.w_ref, r.inv_scale, r.scale_bg, r.scale_stride, r.partial, r.N, r.M,
                                  r.C, r.op, bx, by, r.grid_x);
}

static bool skinny_wgrad_ok(const ConvGeom& g) {
    return !path_disabled("skinny") && g.KH == 1 && g.KW == 1 && g.stride == 1 && g.pad_h == 0 && g.pad_w == 0 && g.H == 1 &&
           g.W == 1 && g.OH == 1 && g.OW == 1;
}
static dim3 skinny_wgrad_grid(const ConvGeom& g) { return dim3((g.C + 63) / 64, (g.M + skw_mt(g.M) - 1) / skw_mt(g.M)); }
// one output pixel, a kernel larger than 1x1 (see OnePix); the input map is small by construction (it fits under the kernel)
static bool onepix_wgrad_ok(const ConvGeom& g) {
    return !path_disabled("skinny") && !skinny_wgrad_ok(g) && g.OH == 1 && g.OW == 1 && (long long)g.C * g.H * g.W < (1 << 24);
}
static dim3 onepix_wgrad_grid(const ConvGeom& g) { return dim3((g.C * g.H * g.W + 63) / 64, (g.M + skw_mt(g.M) - 1) / skw_mt(g.M)); }

LOCATE_API size_t locate_conv_wgrad_workspace_bytes(const int* geom) {
    const ConvGeom g = make_geom(geom);
    if (skinny_wgrad_ok(g) || onepix_wgrad_ok(g)) return 0;
    {
        const PwPlan q = pw_plan(g);
        if (q.ok) return (size_t)q.nslab * g.M * g.C * sizeof(float);
    }
    int bm, nsplit, chunk, tiles;
    wgrad_plan(g, &bm, &nsplit, &chunk, &tiles);
    return nsplit > 1 ? (size_t)nsplit * g.M * g.C * g.KH * g.KW * sizeof(float) : 0;
}

// number of doubles written to `inner_partial` by locate_conv_wgrad for this geometry
LOCATE_API int locate_conv_wgrad_partials(const int* geom) {
    const ConvGeom g = make_geom(geom);
    if (skinny_wgrad_ok(g)) {
        const dim3 grid = skinny_wgrad_grid(g);
        return (int)(grid.x * grid.y);
    }
    if (onepix_wgrad_ok(g)) {
        const dim3 grid = onepix_wgrad_grid(g);
        return (int)(grid.x * grid.y);
    }
    {
        const PwPlan q = pw_plan(g);
        if (q.ok) return wgrad_reduce_grid((int64_t)g.M * g.C, q.nslab);
    }
    int bm, nsplit, chunk, tiles;
    wgrad_plan(g, &bm, &nsplit, &chunk, &tiles);
    return nsplit > 1 ? wgrad_reduce_grid((int64_t)g.M * g.C * g.KH * g.KW, nsplit) : tiles;
}

// ---- the small weight gradients of a pass in one launch (SkwRec above) ----
LOCATE_API size_t locate_wgrad_batch_record_bytes(void) { return sizeof(SkwRec); }
LOCATE_API int locate_wgrad_batch_max(void) { return SKW_MAX; }
// Fills `record` (locate_wgrad_batch_record_bytes() bytes, host memory) with the launch of locate_conv_wgrad for this geometry
// and these operands and returns its number of blocks - or 0 when the geometry is not one of the small-map layers (1x1 maps,
// one output pixel), which the caller then launches on its own.  Same argument meaning as locate_conv_wgrad.
LOCATE_API int locate_wgrad_batch_record(const int* geom, const float* x, int64_t x_bs, const float* gy, int64_t gy_bs, float* gw,
                                         const float* w_ref, const float* inv_scale, int scale_group_batch, int scale_stride,
                                         double* inner_partial, void* record) {
    const ConvGeom g = make_geom(geom);
    if (geom_check(g, "locate_wgrad_batch_record") || !record || !x || !gy || !gw) return 0;
    if (inner_partial && !w_ref) return 0;
    if (scale_group_batch < 0 || (scale_group_batch > 0 && (!inv_scale || g.B % scale_group_batch != 0 || g.B / scale_group_batch > 4 ||
                                                            w_ref || inner_partial))) return 0;
    const bool skinny = skinny_wgrad_ok(g), onepix = !skinny && onepix_wgrad_ok(g);
    if (!skinny && !onepix) return 0;
    const dim3 grid = skinny ? skinny_wgrad_grid(g) : onepix_wgrad_grid(g);
    SkwRec r;
    r.x = x; r.gy = gy; r.gw = gw; r.w_ref = w_ref; r.inv_scale = inv_scale; r.partial = inner_partial;
    r.x_bs = x_bs; r.gy_bs = gy_bs;
    r.scale_bg = scale_group_batch; r.scale_stride = scale_stride; r.N = g.B; r.M = g.M; r.C = skinny ? g.C : g.C * g.H * g.W;
    r.grid_x = (int)grid.x; r.block0 = 0;
    r.op = skinny ? OnePix{0, 0, 0, 0, 0, 0, 0} : OnePix{g.H * g.W, g.W, g.KH, g.KW, g.pad_h, g.pad_w, g.C};
    memcpy(record, &r, sizeof(r));
    return (int)(grid.x * grid.y);
}
// Launches n records (filled by locate_wgrad_batch_record, in host memory, packed) in one grid.
LOCATE_API int locate_wgrad_batch(const void* records, int n, void* stream) {
    LOCATE_REQUIRE(records && n > 0 && n <= SKW_MAX, "locate_wgrad_batch: 1 .. locate_wgrad_batch_max() records");
    SkwBatch b;
    memcpy(b.r, records, (size_t)n * sizeof(SkwRec));
    long long blocks = 0;
    for (int i = 0; i < n; ++i) {
        const SkwRec& r = b.r[i];
        LOCATE_REQUIRE(r.x && r.gy && r.gw && r.grid_x > 0 && r.M > 0 && r.C > 0 && r.N > 0, "locate_wgrad_batch: bad record");
        b.r[i].block0 = (int)blocks;
        blocks += (long long)r.grid_x * ((r.M + skw_mt(r.M) - 1) / skw_mt(r.M));
    }
    LOCATE_REQUIRE(blocks < (1ll << 31), "locate_wgrad_batch: too many blocks");
    skinny_wgrad_batch_kernel<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(b, n);
    LOCATE_LAUNCH_CHECK("locate_wgrad_batch");
    return LOCATE_OK;
}

// Stacked calls with the per-call <G_k / sigma_k, W_bar> partials out of the split reduction (locate_conv_wgrad with
// scale_group_batch > 0 AND w_ref + inner_partial): partials PER CALL for this geometry split into `groups` calls - inner_partial
// then holds groups x that many doubles, [call][partial] - or 0 when this geometry cannot emit them (layers on 1x1 maps / with one
// output pixel, call lengths that do not divide into whole slabs: take <gy_k, y_k - bias> on the activation side instead,
// locate_fin_sn_dots).  The workspace of that mode has its own size.
static bool wgrad_group_dots_ok(const ConvGeom& g, int groups) {
    if (groups < 2 || groups > 4 || g.B % groups != 0 || skinny_wgrad_ok(g) || onepix_wgrad_ok(g)) return false;
    const PwPlan q = pw_plan(g, groups);
    if (q.ok) return q.zper > 0;
    return (long long)g.B * g.OH * g.OW < (1ll << 31);
}
LOCATE_API int locate_conv_wgrad_group_partials(const int* geom, int groups) {
    const ConvGeom g = make_geom(geom);
    if (geom_check(g, "locate_conv_wgrad_group_partials") || !wgrad_group_dots_ok(g, groups)) return 0;
    const PwPlan q = pw_plan(g, groups);
    if (q.ok) return wgrad_reduce_grid((int64_t)g.M * g.C, q.nslab);
    int bm, nsplit, chunk, tiles;
    wgrad_plan(g, &bm, &nsplit, &chunk, &tiles, groups);
    return wgrad_reduce_grid((int64_t)g.M * g.C * g.KH * g.KW, nsplit);
}
LOCATE_API size_t locate_conv_wgrad_group_workspace_bytes(const int* geom, int groups) {
    const ConvGeom g = make_geom(geom);
    if (geom_check(g, "locate_conv_wgrad_group_workspace_bytes") || !wgrad_group_dots_ok(g, groups)) return 0;
    const PwPlan q = pw_plan(g, groups);
    if (q.ok) return (size_t)q.nslab * g.M * g.C * sizeof(float);
    int bm, nsplit, chunk, tiles;
    wgrad_plan(g, &bm, &nsplit, &chunk, &tiles, groups);
    return (size_t)nsplit * g.M * g.C * g.KH * g.KW * sizeof(float);
}

// gw[m,c,kh,kw] = inv_scale * sum_{b,oh,ow} gy[b,m,oh,ow] x[b,c,oh*s-ph+kh,ow*s-pw+kw]          (overwritten)
// With w_ref (= W_bar, same layout as gw) and inner_partial: the partial sums of <UNSCALED gw, W_bar> the
// spectral-norm backward needs come out of the same pass (locate_conv_wgrad_partials(geom) doubles).
// scale_group_batch > 0: gy of batch element b is weighted by inv_scale[(b / scale_group_batch) * scale_stride] instead
// (stacked forwards; at most 4 groups; w_ref / inner_partial must then be null - see locate_sn_group_dsigma).
// deferred_reduce (nullable, host memory of locate_slab_reduce_record_bytes() bytes): the split reduction - when this geometry has
// one - is NOT launched; its launch is written there instead and gw / inner_partial are complete only after
// locate_slab_reduce_batch() has run that record (the workspace must stay untouched until then).  The record's block count is
// locate_slab_reduce_record_blocks(record): 0 = nothing pending (gw is complete when this launch is).
LOCATE_API int locate_conv_wgrad(const int* geom, const float* x, int64_t x_bs, const float* gy, int64_t gy_bs, float* gw,
                                 const float* w_ref, const float* inv_scale, int scale_group_batch, int scale_stride,
                                 double* inner_partial, void* workspace, int precision, const void* x_absmax, const void* gy_absmax,
                                 void* deferred_reduce, void* stream) {
    const ConvGeom g = make_geom(geom);
    if (int e = geom_check(g, "locate_conv_wgrad")) return e;
    LOCATE_REQUIRE(precision >= 0 && precision <= 3, "locate_conv_wgrad: precision must be 0 (fp32-faithful, bf16 pieces), 1 (bf16 operands), 2 (fp32-faithful, fp16 pieces) or 3 (fp8 operands)");
    LOCATE_REQUIRE(precision < 2 || (x_absmax && gy_absmax), "locate_conv_wgrad: precisions 2 and 3 need the absmax words of x and gy");
    LOCATE_REQUIRE(x && gy && gw, "locate_conv_wgrad: null pointer");
    LOCATE_REQUIRE(!inner_partial || w_ref, "locate_conv_wgrad: inner_partial needs w_ref");
    // stacked calls with w_ref + inner_partial: the per-call dots come out of the split reduction (locate_conv_wgrad_group_partials)
    const int gd = (scale_group_batch > 0 && w_ref && inner_partial && g.B % scale_group_batch == 0) ? g.B / scale_group_batch : 0;
    LOCATE_REQUIRE(scale_group_batch >= 0 && (scale_group_batch == 0 || (inv_scale && g.B % scale_group_batch == 0 &&
                   g.B / scale_group_batch <= 4 && ((!w_ref && !inner_partial) || wgrad_group_dots_ok(g, gd)))),
                   "locate_conv_wgrad: bad group scaling arguments (per-call partials: see locate_conv_wgrad_group_partials)");
    hipStream_t st = as_stream(stream);
    if (deferred_reduce) memset(deferred_reduce, 0, sizeof(SlabRec));
    auto reduce = [&](const float* slab, int64_t n, int nsplit, const float* scale, int zper, const char* who) -> int {
        const int rg = wgrad_reduce_grid(n, nsplit);
        const int zp = wgrad_reduce_zp(nsplit, n);
        if (deferred_reduce) {
            SlabRec r;
            r.slab = slab; r.out = gw; r.w_ref = w_ref; r.inv_scale = scale; r.partial = inner_partial;
            r.n = n; r.nsplit = nsplit; r.zp = zp; r.grid = rg; r.block0 = 0; r.groups = gd; r.zper = zper;
            memcpy(deferred_reduce, &r, sizeof(r));
            return LOCATE_OK;
        }
        if (zp == 16) slab_reduce_kernel<16><<<rg, 256, 0, st>>>(slab, gw, n, nsplit, w_ref, scale, inner_partial, gd, zper);
        else if (zp == 4) slab_reduce_kernel<4><<<rg, 256, 0, st>>>(slab, gw, n, nsplit, w_ref, scale, inner_partial, gd, zper);
        else slab_reduce_kernel<1><<<rg, 256, 0, st>>>(slab, gw, n, nsplit, w_ref, scale, inner_partial, gd, zper);
        LOCATE_LAUNCH_CHECK(who);
        return LOCATE_OK;
    };
    if (skinny_wgrad_ok(g)) {          // 1x1 maps: plain fp32 FMAs at either precision setting (see skinny_rows_kernel)
        skinny_wgrad_kernel<<<skinny_wgrad_grid(g), 256, 0, st>>>(x, x_bs, gy, gy_bs, gw, w_ref, inv_scale, scale_group_batch, scale_stride,
                                                                  inner_partial, g.B, g.M, g.C, OnePix{0, 0, 0, 0, 0, 0, 0});
        LOCATE_LAUNCH_CHECK("locate_conv_wgrad(1x1 map)");
        return LOCATE_OK;
    }
    if (onepix_wgrad_ok(g)) {          // one output pixel: the useful taps only; the kernel zeroes the others itself (see OnePix)
        const OnePix op = {g.H * g.W, g.W, g.KH, g.KW, g.pad_h, g.pad_w, g.C};
        skinny_wgrad_kernel<<<onepix_wgrad_grid(g), 256, 0, st>>>(x, x_bs, gy, gy_bs, gw, w_ref, inv_scale, scale_group_batch, scale_stride,
                                                                  inner_partial, g.B, g.M, g.C * g.H * g.W, op);
        LOCATE_LAUNCH_CHECK("locate_conv_wgrad(one output pixel)");
        return LOCATE_OK;
    }
    const PwPlan pq = pw_plan(g, gd);
    if (pq.ok) {
        // the size queries (workspace bytes, partial count) decide on the geometry alone, so the pointwise plan is binding here:
        // operands it cannot take are an error, never a silent switch to the general plan with its different workspace layout
        LOCATE_REQUIRE((x_bs & 3) == 0 && (gy_bs & 3) == 0 &&
                       ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0,
                       "locate_conv_wgrad: narrow 1x1 layers need 16-byte aligned x / gy and batch strides that are multiples of 4");
        LOCATE_REQUIRE(workspace, "locate_conv_wgrad: the split reduction needs a workspace");
        PwParams q;
        q.x = x; q.gy = gy; q.slab = static_cast<float*>(workspace); q.x_bs = x_bs; q.gy_bs = gy_bs;
        q.B = g.B; q.C = g.C; q.M = g.M; q.P = g.H * g.W; q.N = g.B * g.H * g.W; q.steps = pq.steps; q.chunk = pq.chunk;
        q.tiles_c = pq.tiles_c;
        const bool grouped = scale_group_batch > 0;
        q.inv_scale = grouped ? inv_scale : nullptr; q.gscale_bg = scale_group_batch; q.gscale_stride = scale_stride;
        const dim3 grid(pq.nslab, pq.tiles_m * pq.tiles_c);
        const int key = (pq.tm - 1) * 2 + (pq.tn - 1);
        q.x_absmax = static_cast<const unsigned*>(x_absmax);
        q.g_absmax = static_cast<const unsigned*>(gy_absmax);
        if (precision == 2 && x_absmax && gy_absmax && !grouped) {
            // (stacked calls weight gy by 1 / sigma_k while it is loaded: its largest magnitude no longer bounds the scaled value -
            // they keep the three-piece form, which needs no range)
            if (key == 0) pw_wgrad_kernel<1, 1, 2><<<grid, 256, 0, st>>>(q);
            else if (key == 1) pw_wgrad_kernel<1, 2, 2><<<grid, 256, 0, st>>>(q);
            else if (key == 2) pw_wgrad_kernel<2, 1, 2><<<grid, 256, 0, st>>>(q);
            else pw_wgrad_kernel<2, 2, 2><<<grid, 256, 0, st>>>(q);
        } else if (precision == 1) {
            if (key == 0) pw_wgrad_kernel<1, 1, 1><<<grid, 256, 0, st>>>(q);
            else if (key == 1) pw_wgrad_kernel<1, 2, 1><<<grid, 256, 0, st>>>(q);
            else if (key == 2) pw_wgrad_kernel<2, 1, 1><<<grid, 256, 0, st>>>(q);
            else pw_wgrad_kernel<2, 2, 1><<<grid, 256, 0, st>>>(q);
        } else {
            if (key == 0) pw_wgrad_kernel<1, 1, 3><<<grid, 256, 0, st>>>(q);
            else if (key == 1) pw_wgrad_kernel<1, 2, 3><<<grid, 256, 0, st>>>(q);
            else if (key == 2) pw_wgrad_kernel<2, 1, 3><<<grid, 256, 0, st>>>(q);
            else pw_wgrad_kernel<2, 2, 3><<<grid, 256, 0, st>>>(q);
        }
        LOCATE_LAUNCH_CHECK("locate_conv_wgrad(pointwise)");
        return reduce(q.slab, (int64_t)g.M * g.C, pq.nslab, grouped ? nullptr : inv_scale, pq.zper, "locate_conv_wgrad(pointwise reduce)");
    }
    int bm, nsplit, chunk, tiles, zper;
    wgrad_plan(g, &bm, &nsplit, &chunk, &tiles, gd, &zper);
    LOCATE_REQUIRE(nsplit == 1 || workspace, "locate_conv_wgrad: split reduction needs a workspace");
    WgParams p;
    p.x = x; p.gy = gy; p.slab = static_cast<float*>(workspace); p.x_bs = x_bs; p.gy_bs = gy_bs;
    p.B = g.B; p.C = g.C; p.H = g.H; p.W = g.W; p.M = g.M; p.OH = g.OH; p.OW = g.OW; p.KH = g.KH; p.KW = g.KW;
    p.stride = g.stride; p.pad_h = g.pad_h; p.pad_w = g.pad_w;
    p.R = g.C * g.KH * g.KW; p.N = g.B * g.OH * g.OW; p.chunk = chunk;
    p.zper = zper; p.Ng = gd > 1 ? p.N / gd : p.N;
    fastdiv_make((unsigned)(g.OH * g.OW), &p.q_mul, &p.q_s1, &p.q_s2);
    fastdiv_make((unsigned)g.OW, &p.ow_mul, &p.ow_s1, &p.ow_s2);
    const bool direct = nsplit == 1;
    const bool grouped = scale_group_batch > 0;
    p.gscale_bg = scale_group_batch; p.gscale_stride = scale_stride;
    p.x_absmax = static_cast<const unsigned*>(x_absmax);
    p.g_absmax = static_cast<const unsigned*>(gy_absmax);
    p.direct_out = direct ? gw : nullptr;
    p.w_ref = direct ? w_ref : nullptr;
    p.inv_scale = (direct || grouped) ? inv_scale : nullptr;
    p.partial = direct ? inner_partial : nullptr;
    dim3 grid((p.R + 127) / 128, (g.M + bm - 1) / bm, nsplit);
    const long long x_extent = 4ll * ((long long)(g.B - 1) * x_bs + (long long)g.C * g.H * g.W);
    p.x_bytes = (unsigned)x_extent;
    // pairs of adjacent reduction elements: same image and same output row, 8-byte aligned in gy
    const bool pairs_ok = ((g.OH * g.OW) & 1) == 0 && (g.OW & 1) == 0 && (gy_bs & 1) == 0 && (chunk & 1) == 0 &&
                          (reinterpret_cast<uintptr_t>(gy) & 7) == 0 && x_extent > 0 && x_extent < (1ll << 31) - (1 << 20);
    LOCATE_REQUIRE(bm != 192 || pairs_ok, "locate_conv_wgrad: layers with M %% 192 == 0 on even maps take the paired-load kernels - gy must be 8-byte aligned with an even batch stride");
    if (precision == 3 && pairs_ok) {          // (odd output maps: the exact fp32-MFMA kernel below, at every precision setting)
        if (bm == 192) conv_wgrad_bx6_kernel<2, 2, 3, 2, 4><<<grid, 256, 0, st>>>(p);
        else if (bm == 128) conv_wgrad_bx6_kernel<2, 2, 2, 2, 4><<<grid, 256, 0, st>>>(p);
        else if (bm == 96) conv_wgrad_bx6_kernel<1, 4, 3, 1, 4><<<grid, 256, 0, st>>>(p);
        else if (bm == 64) conv_wgrad_bx6_kernel<1, 4, 2, 1, 4><<<grid, 256, 0, st>>>(p);
        else conv_wgrad_bx6_kernel<1, 4, 1, 1, 4><<<grid, 256, 0, st>>>(p);
    } else if (pairs_ok && precision == 1) {
        if (bm == 192) conv_wgrad_bx6_kernel<2, 2, 3, 2, 1><<<grid, 256, 0, st>>>(p);
        else if (bm == 128) conv_wgrad_bx6_kernel<2, 2, 2, 2, 1><<<grid, 256, 0, st>>>(p);
        else if (bm == 96) conv_wgrad_bx6_kernel<1, 4, 3, 1, 1><<<grid, 256, 0, st>>>(p);
        else if (bm == 64) conv_wgrad_bx6_kernel<1, 4, 2, 1, 1><<<grid, 256, 0, st>>>(p);
        else conv_wgrad_bx6_kernel<1, 4, 1, 1, 1><<<grid, 256, 0, st>>>(p);
    } else if (pairs_ok && precision == 2 && !path_disabled("wbx6")) {
        if (bm == 192) conv_wgrad_bx6_kernel<2, 2, 3, 2, 2><<<grid, 256, 0, st>>>(p);
        else if (bm == 128) conv_wgrad_bx6_kernel<2, 2, 2, 2, 2><<<grid, 256, 0, st>>>(p);
        else if (bm == 96) conv_wgrad_bx6_kernel<1, 4, 3, 1, 2><<<grid, 256, 0, st>>>(p);
        else if (bm == 64) conv_wgrad_bx6_kernel<1, 4, 2, 1, 2><<<grid, 256, 0, st>>>(p);
        else conv_wgrad_bx6_kernel<1, 4, 1, 1, 2><<<grid, 256, 0, st>>>(p);
    } else if (pairs_ok && !path_disabled("wbx6")) {
        if (bm == 192) conv_wgrad_bx6_kernel<2, 2, 3, 2, 3><<<grid, 256, 0, st>>>(p);
        else if (bm == 128) conv_wgrad_bx6_kernel<2, 2, 2, 2, 3><<<grid, 256, 0, st>>>(p);
        else if (bm == 96) conv_wgrad_bx6_kernel<1, 4, 3, 1, 3><<<grid, 256, 0, st>>>(p);
        else if (bm == 64) conv_wgrad_bx6_kernel<1, 4, 2, 1, 3><<<grid, 256, 0, st>>>(p);
        else conv_wgrad_bx6_kernel<1, 4, 1, 1, 3><<<grid, 256, 0, st>>>(p);
    } else if (bm == 128) conv_wgrad_kernel<2, 2, 2, 2><<<grid, 256, 0, st>>>(p);
    else if (bm == 96) conv_wgrad_kernel<1, 4, 3, 1><<<grid, 256, 0, st>>>(p);
    else if (bm == 64) conv_wgrad_kernel<1, 4, 2, 1><<<grid, 256, 0, st>>>(p);
    else conv_wgrad_kernel<1, 4, 1, 1><<<grid, 256, 0, st>>>(p);
    LOCATE_LAUNCH_CHECK("locate_conv_wgrad(gemm)");
    if (!direct) return reduce(p.slab, (int64_t)g.M * p.R, nsplit, grouped ? nullptr : inv_scale, zper, "locate_conv_wgrad(reduce)");
    return LOCATE_OK;
}

LOCATE_API size_t locate_slab_reduce_record_bytes(void) { return sizeof(SlabRec); }
LOCATE_API int locate_slab_reduce_max(void) { return SLAB_MAX; }
LOCATE_API int locate_slab_reduce_record_blocks(const void* record) {
    if (!record) return 0;
    SlabRec r;
    memcpy(&r, record, sizeof(r));
    return r.grid;
}
// Runs n deferred split reductions (records written by locate_conv_wgrad(deferred_reduce), packed, host memory) in one grid.
LOCATE_API int locate_slab_reduce_batch(const void* records, int n, void* stream) {
    LOCATE_REQUIRE(records && n > 0 && n <= SLAB_MAX, "locate_slab_reduce_batch: 1 .. locate_slab_reduce_max() records");
    SlabBatch b;
    memcpy(b.r, records, (size_t)n * sizeof(SlabRec));
    long long blocks = 0;
    for (int i = 0; i < n; ++i) {
        const SlabRec& r = b.r[i];
        LOCATE_REQUIRE(r.slab && r.out && r.n > 0 && r.nsplit > 0 && r.grid > 0 && (r.zp == 1 || r.zp == 4 || r.zp == 16),
                       "locate_slab_reduce_batch: bad record");
        b.r[i].block0 = (int)blocks;
        blocks += r.grid;
    }
    slab_reduce_batch_kernel<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(b, n);
    LOCATE_LAUNCH_CHECK("locate_slab_reduce_batch");
    return LOCATE_OK;
}
