// End-of-backward finalisers, batched: what every spectral-normalised layer and every residual gate still owes its
// parameters' gradients once the backward pass's last data gradient is out -
//     * <gy_k, y_k - bias> per stacked call k            (activation-side <G_k, W_bar>, spectral.hip)
//     * dW_bar += (sum_k dsigma_k) u v^T, du, dsigma     (rank-1 term of d(W_bar / sigma), libs/spectral_norm.py:31-32)
//     * dgamma = sum of the gate kernel's block partials (libs/merge.py:33-38)
// - for ALL layers of the pass in one launch each, instead of two or three small launches per layer in the middle of the
// pass's dependent chain (84 + 34 of the ~870 launches of a step at 64x64).
//
// The records travel BY VALUE in the kernel arguments (<= FIN_MAX records of 112 bytes per launch): no device table, no
// host-to-device copy - the operands are this pass's transient gradient buffers, whose addresses differ from pass to pass,
// and a hipGraph capture records the arguments themselves.  Arithmetic and summation order are those of the per-layer
// kernels (spectral.hip, elementwise.hip): results are bit-identical to the unbatched path, which the data-parallel eager
// mode still takes (its reducer hooks need complete gradients from autograd).
#include "common.h"
#include "finrec.h"

LOCATE_API size_t locate_fin_record_bytes(void) { return sizeof(FinRec); }

// ---- activation-side dots of stacked calls -------------------------------------------------------------------------
// p0 gy, p1 y, p2 bias (nullable), p3 partial out [groups][nb];  l0 gy_bs, l1 y_bs;  i0 groups, i1 Bg, i2 M, i3 plane, i4 nb
#define FIN_DOT_CHUNK 4096
#define FIN_DOT_BLOCKS 256
__global__ void __launch_bounds__(256) fin_sn_dots_kernel(const FinBatch batch) {
    __shared__ double scratch[16];
    const FinRec& R = batch.r[blockIdx.y >> 2];
    const int grp = blockIdx.y & 3;
    const int groups = R.i[0], Bg = R.i[1], M = R.i[2], plane = R.i[3], nb = R.i[4];
    if (grp >= groups || (int)blockIdx.x >= nb) return;
    const float* __restrict__ gy = static_cast<const float*>(R.p[0]);
    const float* __restrict__ y = static_cast<const float*>(R.p[1]);
    const float* __restrict__ bias = static_cast<const float*>(R.p[2]);
    double* __restrict__ partial = static_cast<double*>(const_cast<void*>(R.p[3]));
    const int64_t gy_bs = R.l[0], y_bs = R.l[1];
    const int per_b = M * plane;
    const int nchunk = (per_b + FIN_DOT_CHUNK - 1) / FIN_DOT_CHUNK;
    const int work = Bg * nchunk;
    const bool vec = (plane & 3) == 0 && (gy_bs & 3) == 0 && (y_bs & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    double acc = 0.0;
    for (int w = blockIdx.x; w < work; w += nb) {
        const int bl = w / nchunk, ch = w - bl * nchunk;
        const int64_t b = (int64_t)grp * Bg + bl;
        const float* gp = gy + b * gy_bs;
        const float* yp = y + b * y_bs;
        const int j0 = ch * FIN_DOT_CHUNK, j1 = min(j0 + FIN_DOT_CHUNK, per_b);
        float part = 0.0f;
        if (vec) {
            for (int j = j0 + 4 * (int)threadIdx.x; j < j1; j += 1024) {
                const float bv = bias ? bias[j / plane] : 0.0f;
                const float4 g4 = *reinterpret_cast<const float4*>(gp + j), y4 = *reinterpret_cast<const float4*>(yp + j);
                part += (g4.x * (y4.x - bv) + g4.y * (y4.y - bv)) + (g4.z * (y4.z - bv) + g4.w * (y4.w - bv));
            }
        } else {
            for (int j = j0 + (int)threadIdx.x; j < j1; j += 256) part = fmaf(gp[j], yp[j] - (bias ? bias[j / plane] : 0.0f), part);
        }
        acc += (double)part;
    }
    acc = block_sum<double>(acc, scratch);
    if (threadIdx.x == 0) partial[(int64_t)grp * nb + blockIdx.x] = acc;
}

// ---- rank-1 term, du, dsigma ------------------------------------------------------------------------------------------
// p0 partial (double), p1 sigma table ({sigma, 1/sigma} pairs), p2 u, p3 v, p4 wv, p5 gw (in/out), p6 du (nullable),
// p7 dsigma out (nullable);  l0 wv_stride;  i0 npartial (per group), i1 groups (0: one un-stacked call, dsigma from the
// weight-side partials = -sum / sigma^2; k >= 1: stacked, dsigma_k = -sum_k / sigma_k), i2 sigma_stride, i3 h, i4 wd,
// i5 first block of this record in the launch, i6 its block count
__global__ void __launch_bounds__(256) fin_sn_rank1_kernel(const FinBatch batch, int n_rec) {
    __shared__ double scratch[16];
    int ri = 0;
    for (int k = 1; k < n_rec; ++k)
        if ((int)blockIdx.x >= batch.r[k].i[5]) ri = k;
    const FinRec& R = batch.r[ri];
    const int bx = (int)blockIdx.x - R.i[5], nblk = R.i[6];
    const double* __restrict__ partial = static_cast<const double*>(R.p[0]);
    const float* __restrict__ sigma_tab = static_cast<const float*>(R.p[1]);
    const float* __restrict__ u = static_cast<const float*>(R.p[2]);
    const float* __restrict__ v = static_cast<const float*>(R.p[3]);
    const float* __restrict__ wv = static_cast<const float*>(R.p[4]);
    float* __restrict__ gw = static_cast<float*>(const_cast<void*>(R.p[5]));
    float* __restrict__ du = static_cast<float*>(const_cast<void*>(R.p[6]));
    float* __restrict__ dsigma_out = static_cast<float*>(const_cast<void*>(R.p[7]));
    const int npartial = R.i[0], groups = R.i[1], sigma_stride = R.i[2], h = R.i[3], wd = R.i[4];
    const int64_t wv_stride = R.l[0];
    float dsg[4] = {0.f, 0.f, 0.f, 0.f};
    float total = 0.0f;
    if (groups == 0) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < npartial; i += blockDim.x) acc += partial[i];
        acc = block_sum<double>(acc, scratch);
        const float sg = sigma_tab[0];
        dsg[0] = (float)(-acc / ((double)sg * (double)sg));
        total = dsg[0];
    } else {
        for (int k = 0; k < groups; ++k) {
            double acc = 0.0;
            for (int i = threadIdx.x; i < npartial; i += blockDim.x) acc += partial[(int64_t)k * npartial + i];
            acc = block_sum<double>(acc, scratch);
            dsg[k] = (float)(-acc * (double)sigma_tab[k * sigma_stride + 1]);     // [k][1] = 1 / sigma_k
            total += dsg[k];
        }
    }
    const int64_t n = (int64_t)h * wd;
    const int64_t stride = (int64_t)nblk * blockDim.x;
    if ((wd & 3) == 0 && n < (1ll << 33) && ((reinterpret_cast<uintptr_t>(gw) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        // 16-byte accesses, one 32-bit division per four elements (the scalar loop below pays a 64-bit division per element:
        // 49 us for the discriminator's 47 MB of weight gradients, 2 TB/s); the same fma per element
        const unsigned wd4 = (unsigned)(wd >> 2);
        const float4* v4p = reinterpret_cast<const float4*>(v);
        float4* g4p = reinterpret_cast<float4*>(gw);
        for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < (n >> 2); i += stride) {
            const unsigned r = (unsigned)i / wd4, c4 = (unsigned)i - r * wd4;
            const float tu = total * u[r];
            const float4 vv = v4p[c4];
            float4 gv = g4p[i];
            gv.x = fmaf(tu, vv.x, gv.x); gv.y = fmaf(tu, vv.y, gv.y); gv.z = fmaf(tu, vv.z, gv.z); gv.w = fmaf(tu, vv.w, gv.w);
            g4p[i] = gv;
        }
    } else {
        for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < n; i += stride) {
            const int r = (int)(i / wd), c = (int)(i - (int64_t)r * wd);
            gw[i] = fmaf(total * u[r], v[c], gw[i]);
        }
    }
    if (bx == 0) {
        if (du) {
            const int ng = groups == 0 ? 1 : groups;
            for (int i = threadIdx.x; i < h; i += blockDim.x) {
                if (groups == 0) {
                    du[i] = dsg[0] * wv[i];
                } else {
                    float a = 0.0f;
                    for (int k = 0; k < ng; ++k) a = fmaf(dsg[k], wv[(int64_t)k * wv_stride + i], a);
                    du[i] = a;
                }
            }
        }
        if (threadIdx.x == 0 && dsigma_out) dsigma_out[0] = total;
    }
}

// ---- plain sums of double partials into one float (the gates' dgamma) ---------------------------------------------------
// p0 partials (double), p1 out (float);  i0 count
__global__ void __launch_bounds__(256) fin_sums_kernel(const FinBatch batch) {
    __shared__ double scratch[16];
    const FinRec& R = batch.r[blockIdx.x];
    const double* __restrict__ part = static_cast<const double*>(R.p[0]);
    double acc = 0.0;
    for (int i = threadIdx.x; i < R.i[0]; i += blockDim.x) acc += part[i];
    acc = block_sum<double>(acc, scratch);
    if (threadIdx.x == 0) static_cast<float*>(const_cast<void*>(R.p[1]))[0] = (float)acc;
}

// ---- per-channel sums over batch and space (the bias gradients of the 1x1 skip convs and of the style linears) ---------
// The arithmetic of norm.hip's channel_sum_kernel / channel_sum_final_kernel (same thread-to-element map, same order), for all
// biases of a backward pass in two launches: blocks of 1024 threads over (channel, batch slice) pairs of every record, then one
// block per record adding its slices.
// p0 g, p1 out [C], p2 slice partials [slices][C] (unused when slices = 1);  l0 batch stride;  i0 B, i1 C, i2 hw, i3 slices,
// i4 batch elements per slice, i5 first block of this record in the launch, i6 its block count (= C * slices)
__global__ void __launch_bounds__(1024) fin_channel_sums_kernel(const FinBatch batch, int n_rec) {
    __shared__ float scratch[16];
    int ri = 0;
    for (int k = 1; k < n_rec; ++k)
        if ((int)blockIdx.x >= batch.r[k].i[5]) ri = k;
    const FinRec& R = batch.r[ri];
    const int bx = (int)blockIdx.x - R.i[5];
    const int B = R.i[0], C = R.i[1], hw = R.i[2], slices = R.i[3], per_slice = R.i[4];
    const float* __restrict__ g = static_cast<const float*>(R.p[0]);
    float* __restrict__ out = static_cast<float*>(const_cast<void*>(slices > 1 ? R.p[2] : R.p[1]));
    const int64_t batch_stride = R.l[0];
    const bool vec = (hw & 3) == 0 && (batch_stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
    if (channel_sum_small(per_slice, hw, vec)) {          // sixteen (channel, slice) pairs per block, a wave each (channel_sum_kernel's form)
        const int C16 = (C + 15) / 16;
        const int sl = bx / C16, c = (bx - sl * C16) * 16 + ((int)threadIdx.x >> 6);
        const int b0 = sl * per_slice;
        const int nb = min(per_slice, B - b0);
        if (c < C) {
            const float t = nb > 0 ? channel_sum_wave(g, batch_stride, c, hw, b0, nb, vec, threadIdx.x & 63) : 0.0f;
            if ((threadIdx.x & 63) == 0) out[(int64_t)sl * C + c] = t;
        }
        return;
    }
    const int c = bx % C, sl = bx / C;
    const int b0 = sl * per_slice;
    const int nb = min(per_slice, B - b0);
    float acc = 0.0f;
    if (nb > 0) {
        if (vec) {
            const int hw4 = hw >> 2;
            const int total = nb * hw4;
            for (int i = threadIdx.x; i < total; i += blockDim.x) {
                const int b = i / hw4, r = i - b * hw4;
                const float4 v = reinterpret_cast<const float4*>(g + (int64_t)(b0 + b) * batch_stride + (int64_t)c * hw)[r];
                acc += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            const int total = nb * hw;
            for (int i = threadIdx.x; i < total; i += blockDim.x) {
                const int b = i / hw, r = i - b * hw;
                acc += g[(int64_t)(b0 + b) * batch_stride + (int64_t)c * hw + r];
            }
        }
    }
    acc = block_sum<float>(acc, scratch);
    if (threadIdx.x == 0) out[(int64_t)sl * C + c] = acc;
}

__global__ void __launch_bounds__(256) fin_channel_final_kernel(const FinBatch batch) {
    const FinRec& R = batch.r[blockIdx.x];
    const int C = R.i[1], slices = R.i[3];
    if (slices <= 1) return;
    const float* __restrict__ part = static_cast<const float*>(R.p[2]);
    float* __restrict__ out = static_cast<float*>(const_cast<void*>(R.p[1]));
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.0f;
        for (int s = 0; s < slices; ++s) acc += part[(int64_t)s * C + c];
        out[c] = acc;
    }
}

// number of partial sums per stacked call that locate_fin_sn_dots writes for a layer output of Bg x M x plane per call
LOCATE_API int locate_fin_sn_dot_partials(int Bg, int M, int plane) {
    const int64_t work = (int64_t)Bg * (((int64_t)M * plane + FIN_DOT_CHUNK - 1) / FIN_DOT_CHUNK);
    return work < FIN_DOT_BLOCKS ? (int)work : FIN_DOT_BLOCKS;
}

static int fin_blocks_for(int64_t n) { return stream_grid(n, 1024); }

// `records`: HOST array of n FinRec (layouts above; i[4] of a dots record and i[5], i[6] of a rank-1 record are filled in
// here).  Any n: the launches take FIN_MAX records each.
LOCATE_API int locate_fin_sn_dots(const void* records, int n, void* stream) {
    LOCATE_REQUIRE(records && n > 0, "locate_fin_sn_dots: bad arguments");
    const FinRec* rec = static_cast<const FinRec*>(records);
    for (int at = 0; at < n; at += FIN_MAX) {
        FinBatch b = {};
        const int m = n - at < FIN_MAX ? n - at : FIN_MAX;
        for (int k = 0; k < m; ++k) {
            b.r[k] = rec[at + k];
            FinRec& R = b.r[k];
            LOCATE_REQUIRE(R.p[0] && R.p[1] && R.p[3] && R.i[0] >= 1 && R.i[0] <= 4 && R.i[1] > 0 && R.i[2] > 0 && R.i[3] > 0 &&
                           (int64_t)R.i[2] * R.i[3] < (1ll << 31), "locate_fin_sn_dots: bad record %d", at + k);
            R.i[4] = locate_fin_sn_dot_partials(R.i[1], R.i[2], R.i[3]);
        }
        fin_sn_dots_kernel<<<dim3(FIN_DOT_BLOCKS, 4 * m), 256, 0, as_stream(stream)>>>(b);
        LOCATE_LAUNCH_CHECK("locate_fin_sn_dots");
    }
    return LOCATE_OK;
}

LOCATE_API int locate_fin_sn_rank1(const void* records, int n, void* stream) {
    LOCATE_REQUIRE(records && n > 0, "locate_fin_sn_rank1: bad arguments");
    const FinRec* rec = static_cast<const FinRec*>(records);
    for (int at = 0; at < n; at += FIN_MAX) {
        FinBatch b = {};
        const int m = n - at < FIN_MAX ? n - at : FIN_MAX;
        int blocks = 0;
        for (int k = 0; k < m; ++k) {
            b.r[k] = rec[at + k];
            FinRec& R = b.r[k];
            LOCATE_REQUIRE(R.p[0] && R.p[1] && R.p[2] && R.p[3] && R.p[5] && R.i[0] > 0 && R.i[1] >= 0 && R.i[1] <= 4 && R.i[3] > 0 &&
                           R.i[4] > 0 && (!R.p[6] || R.p[4]), "locate_fin_sn_rank1: bad record %d", at + k);
            R.i[5] = blocks;
            R.i[6] = fin_blocks_for((int64_t)R.i[3] * R.i[4]);
            blocks += R.i[6];
        }
        fin_sn_rank1_kernel<<<blocks, 256, 0, as_stream(stream)>>>(b, m);
        LOCATE_LAUNCH_CHECK("locate_fin_sn_rank1");
    }
    return LOCATE_OK;
}

// slices of the batch a channel sum of [B, C, hw] is split into (so that few channels still fill the chip) - the rule of
// locate_channel_sum; the record's partial buffer p2 needs slices * C floats when this is > 1
LOCATE_API int locate_fin_channel_slices(int B, int C, int hw) {
    if (C >= 512 || (int64_t)B * hw < 8192) return 1;
    int s = (1024 + C - 1) / C;
    if (s > B) s = B;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

LOCATE_API int locate_fin_channel_sums(const void* records, int n, void* stream) {
    LOCATE_REQUIRE(records && n > 0, "locate_fin_channel_sums: bad arguments");
    const FinRec* rec = static_cast<const FinRec*>(records);
    for (int at = 0; at < n; at += FIN_MAX) {
        FinBatch b = {};
        const int m = n - at < FIN_MAX ? n - at : FIN_MAX;
        int blocks = 0;
        bool any_split = false;
        for (int k = 0; k < m; ++k) {
            b.r[k] = rec[at + k];
            FinRec& R = b.r[k];
            LOCATE_REQUIRE(R.p[0] && R.p[1] && R.i[0] > 0 && R.i[1] > 0 && R.i[2] > 0, "locate_fin_channel_sums: bad record %d", at + k);
            const int slices = locate_fin_channel_slices(R.i[0], R.i[1], R.i[2]);
            LOCATE_REQUIRE(slices == 1 || R.p[2], "locate_fin_channel_sums: record %d needs a partial buffer", at + k);
            R.i[3] = slices;
            R.i[4] = (R.i[0] + slices - 1) / slices;
            R.i[5] = blocks;
            {
                const float* g = static_cast<const float*>(R.p[0]);
                const bool vec = (R.i[2] & 3) == 0 && (R.l[0] & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
                R.i[6] = (channel_sum_small(R.i[4], R.i[2], vec) ? (R.i[1] + 15) / 16 : R.i[1]) * slices;
            }
            blocks += R.i[6];
            any_split = any_split || slices > 1;
        }
        fin_channel_sums_kernel<<<blocks, 1024, 0, as_stream(stream)>>>(b, m);
        LOCATE_LAUNCH_CHECK("locate_fin_channel_sums");
        if (any_split) {
            fin_channel_final_kernel<<<m, 256, 0, as_stream(stream)>>>(b);
            LOCATE_LAUNCH_CHECK("locate_fin_channel_sums(final)");
        }
    }
    return LOCATE_OK;
}

LOCATE_API int locate_fin_sums(const void* records, int n, void* stream) {
    LOCATE_REQUIRE(records && n > 0, "locate_fin_sums: bad arguments");
    const FinRec* rec = static_cast<const FinRec*>(records);
    for (int at = 0; at < n; at += FIN_MAX) {
        FinBatch b = {};
        const int m = n - at < FIN_MAX ? n - at : FIN_MAX;
        for (int k = 0; k < m; ++k) {
            b.r[k] = rec[at + k];
            LOCATE_REQUIRE(b.r[k].p[0] && b.r[k].p[1] && b.r[k].i[0] > 0, "locate_fin_sums: bad record %d", at + k);
        }
        fin_sums_kernel<<<m, 256, 0, as_stream(stream)>>>(b);
        LOCATE_LAUNCH_CHECK("locate_fin_sums");
    }
    return LOCATE_OK;
}
