// Shared declarations of the dense-contraction kernels (conv.hip: implicit-GEMM gather kernels, panel packing, weight
// gradients; convwin.hip: the LDS-window kernels): panel layout, launch parameters, operand splits, the common epilogue.
#pragma once
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <stdlib.h>
#include <type_traits>

// Kernel-flavour switch for A/B timing and for the in-tree cross-check of the bf16 x 6 kernels against the fp32-MFMA ones
// (LOCATE_DISABLE=bx6,wbx6,pointwise).  Compiled ONLY into the debug variant of the library (liblocate_hip_dbg.so,
// -DLOCATE_DEBUG_KNOBS): the shipped liblocate_hip.so reads no environment variable, its dispatch depends on its arguments alone.
#ifdef LOCATE_DEBUG_KNOBS
static bool path_disabled(const char* name) {
    static const char* env = getenv("LOCATE_DISABLE");
    return env != nullptr && strstr(env, name) != nullptr;
}
static int knob_int(const char* name, int fallback) {
    const char* v = getenv(name);
    return v != nullptr ? atoi(v) : fallback;
}
#else
static constexpr bool path_disabled(const char*) { return false; }
static constexpr int knob_int(const char*, int fallback) { return fallback; }
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define IG_BK 16            // K elements per main-loop step
#define IG_KPAD 32          // panels are zero-padded to a multiple of this many K rows
#define IG_TAIL 112         // extra zero rows after Kpad: the branch-free prefetch of the last steps stays in bounds
#define IG_MAXT 32

struct ConvGeom {
    int B, C, H, W;            // input of R
    int M, KH, KW;             // weight [M, C, KH, KW]
    int stride, pad_h, pad_w;
    int OH, OW;                // output of R
};

static int geom_check(const ConvGeom& g, const char* who) {
    LOCATE_REQUIRE(g.B > 0 && g.C > 0 && g.H > 0 && g.W > 0 && g.M > 0 && g.KH > 0 && g.KW > 0 && g.stride > 0,
                   "%s: non-positive dimension", who);
    LOCATE_REQUIRE(g.KH * g.KW <= IG_MAXT - 7, "%s: kernel %dx%d has more than %d taps", who, g.KH, g.KW, IG_MAXT - 7);
    LOCATE_REQUIRE(g.stride <= 2, "%s: stride %d unsupported (1 or 2)", who, g.stride);
    LOCATE_REQUIRE(g.OH == (g.H + 2 * g.pad_h - g.KH) / g.stride + 1 && g.OW == (g.W + 2 * g.pad_w - g.KW) / g.stride + 1,
                   "%s: output size %dx%d does not match the geometry", who, g.OH, g.OW);
    LOCATE_REQUIRE((int64_t)g.B * g.C * g.H * g.W < (1ll << 31) && (int64_t)g.B * g.M * g.OH * g.OW < (1ll << 31),
                   "%s: tensor larger than 2^31 elements", who);
    return LOCATE_OK;
}

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------------
// Weight panel of one phase (all sections in 4-byte units, one buffer):
//   weights [rows][ld]   K-major, rows = Kpad + IG_TAIL, zero beyond K and beyond M
//   koff    [rows]       int: BYTE offset of gathered row k = (c, t) inside one batch image of the gathered tensor,
//                        4 (c*H*W + dy(t)*W + dx(t) - dmin), dmin = the most negative tap displacement, so that the
//                        offsets are >= 0 (they go into the scalar offset of a buffer load whose descriptor starts
//                        dmin elements before the tensor; 0 beyond K)
//   ktap    [rows / 4]   bytes: tap index t of row k (31 beyond K: a tap that is never valid)
//   w3      [3][rows / 8][ld][8]  bf16: the weights as three bf16 pieces w = h + m + l (exact), 8 consecutive k per
//                        16-byte chunk = the A fragment of v_mfma_f32_32x32x16_bf16
// The offset table turns the gather's per-element mixed-radix arithmetic (~230 ALU instructions per K step, which
// cost a quarter of the kernel's throughput) into two wave-uniform scalar loads per step.
// ---------------------------------------------------------------------------------------------
struct PackArgs {
    const float* w;       // [M, C, KH, KW]
    float* out;           // phase panel
    int M, C, KH, KW;
    int mode;             // 0: rows k=(c,kh,kw), cols m          (R forward)
                          // 1: rows k=(m,th,tw), cols c, taps kh = kh0 + s*th, kw = kw0 + s*tw   (R data-adjoint phase)
    int kh0, kw0, s, TH, TW;
    int K, rows, ld;
    int gHW, gW, dy0, dys, dx0, dxs;   // geometry of the gathered tensor and of the tap grid
    int dmin;                          // min over the taps of dy*gW + dx (<= 0)
    int fmt;                           // split section: 0 = three bf16 planes, 1 = header + two fp16 planes
    // DIRECT re-packing (fmt 1 only): the weights' largest magnitude is already known (AMAX_WORDS words left by the optimizer
    // kernel, nadam.hip), so the packing blocks write the two scaled fp16 planes themselves - no second pass over an fp32
    // intermediate - and, where no kernel reads the K-major fp32 rows (every panel but single-tap ones: pointwise / 1x1-map
    // kernels), do not write those either.  The offset tables and zero tails of an earlier full packing stay as they are.
    const unsigned* wmax;              // fmt 1, direct form: the weights' absmax words
    int direct;                        // 0: the two-pass form (fp32 rows, maximum folded in while packing, split pass)
    int keep_f32;                      // direct form: also refresh the fp32 rows
    // WINDOW panels (convwin.hip; panel format bit 2): no fp32 rows, no offset tables - header + piece planes only, the chunk
    // rows in "unit" order u = c8g * Tp + t: the 8 reduction channels 8 c8g .. 8 c8g + 7 at tap t (zero chunks for t >= T and
    // beyond the last channel), urows rows in all (zero tail included).  wmax_single: wmax points at ONE word (the panel's own
    // header, filled by win_absmax_jobs_kernel) instead of AMAX_WORDS words.
    int win, Tp, urows, wmax_single;
};

// fmt 0: the split section holds the three bf16 planes; fmt 1 ("fp16 pieces", see conv_igemm_bx6_kernel NP = 2): a 4-dword
// header {absmax bits of this phase's weights, 0, 0, 0} followed by TWO fp16 planes of the weights times 2^k(absmax)
#define PANEL_HDR 4
// fmt 2 ("fp8", convfp8.hip): the same header followed by ONE plane of e4m3 bytes, 16 consecutive k per 16-byte chunk
static inline size_t panel_floats(int rows, int ld, int fmt) {
    if (fmt == 2) return (size_t)rows * ld + rows + rows / 4 + PANEL_HDR + (size_t)rows * ld / 4;
    return (size_t)rows * ld + rows + rows / 4 + (fmt ? PANEL_HDR + (size_t)rows * ld / 2 * 2 : (size_t)rows * ld / 2 * 3);
}
static inline size_t panel_split_offset(int rows, int ld) { return (size_t)rows * ld + rows + rows / 4; }   // floats
__device__ __forceinline__ size_t panel_split_offset_dev(int rows, int ld) { return (size_t)rows * ld + rows + rows / 4; }

// Scale exponent of a tensor whose largest magnitude has the fp32 bit pattern `bits`: k with absmax * 2^k in [2^14, 2^15)
// - one binade below fp16's largest finite value, so that the two fp16 pieces of every element (11 + 11 significant bits)
// stay normal numbers down to elements 2^-17 of the largest (below that the low piece goes subnormal: absolute error
// <= 2^-25 on the scaled tensor, i.e. 2^-39 of its largest element).  Zero / denormal tensors: k = 0.  Clamped so that
// 2^k and 2^-k are normal fp32 numbers.
__host__ __device__ __forceinline__ int f16_scale_exp(unsigned bits) {
    const int e = (int)((bits >> 23) & 0xffu) - 127;
    if (e == -127) return 0;
    const int k = 14 - e;
    return k > 100 ? 100 : (k < -100 ? -100 : k);
}
// the same for e4m3 operands (largest finite value 448): absmax * 2^k in [2^7, 2^8)
__host__ __device__ __forceinline__ int f8_scale_exp(unsigned bits) {
    const int e = (int)((bits >> 23) & 0xffu) - 127;
    if (e == -127) return 0;
    const int k = 7 - e;
    return k > 100 ? 100 : (k < -100 ? -100 : k);
}
__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((unsigned)(k + 127) << 23); }      // -126 <= k <= 127

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x = h + m + l exactly (3 x 8 significant bits cover fp32's 24): h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
__device__ __forceinline__ void split3_bf16x8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h[j] = (__bf16)v[j];
        const float r1 = v[j] - (float)h[j];
        m[j] = (__bf16)r1;
        l[j] = (__bf16)(r1 - (float)m[j]);
    }
}

// The same exact decomposition by TRUNCATION for the in-loop splits: h = the top 16 bits of x (8 significant bits), r = x - h
// (exact), m = the top 16 bits of r, l = r - m (at most 8 significant bits left, so its top 16 bits hold all of it).  Bit masks,
// two subtractions and one byte permute per bf16 pair instead of three conversions and two shifts per element.
__device__ __forceinline__ void split3_trunc_pair(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __builtin_bit_cast(unsigned, v0), u1 = __builtin_bit_cast(unsigned, v1);
    const float r0 = v0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = v1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const unsigned q0 = __builtin_bit_cast(unsigned, r0), q1 = __builtin_bit_cast(unsigned, r1);
    const float l0 = r0 - __builtin_bit_cast(float, q0 & 0xffff0000u), l1 = r1 - __builtin_bit_cast(float, q1 & 0xffff0000u);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);            // (hi16(v1) << 16) | hi16(v0)
    m = __builtin_amdgcn_perm(q1, q0, 0x07060302u);
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, l1), __builtin_bit_cast(unsigned, l0), 0x07060302u);
}

// two fp32 values rounded to nearest-even bf16, packed (v0 in the low half)
__device__ __forceinline__ unsigned round_bf16_pair(float v0, float v1) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t r;
    r[0] = (__bf16)v0;
    r[1] = (__bf16)v1;
    return *reinterpret_cast<unsigned*>(&r);
}

__device__ __forceinline__ void split3_trunc_x8(const float (&v)[8], uint4& h, uint4& m, uint4& l) {
    split3_trunc_pair(v[0], v[1], h.x, m.x, l.x);
    split3_trunc_pair(v[2], v[3], h.y, m.y, l.y);
    split3_trunc_pair(v[4], v[5], h.z, m.z, l.z);
    split3_trunc_pair(v[6], v[7], h.w, m.w, l.w);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// x * 2^k = h + l up to 2^-22 |x| with h = fp16(x 2^k), l = fp16(x 2^k - h), both rounded to nearest even
__device__ __forceinline__ void split2_f16x8(const float (&v)[8], float sc, uint4& h, uint4& l) {
    f16x8 hh, ll;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = v[j] * sc;
        hh[j] = (_Float16)x;
        ll[j] = (_Float16)(x - (float)hh[j]);
    }
    h = *reinterpret_cast<uint4*>(&hh);
    l = *reinterpret_cast<uint4*>(&ll);
}

// the same for one pair of values, packed (v0 in the low halves)
__device__ __forceinline__ void split2_f16_pair(float v0, float v1, unsigned& h, unsigned& l) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    f16x2 hh, ll;
    hh[0] = (_Float16)v0; hh[1] = (_Float16)v1;
    ll[0] = (_Float16)(v0 - (float)hh[0]); ll[1] = (_Float16)(v1 - (float)hh[1]);
    h = *reinterpret_cast<unsigned*>(&hh);
    l = *reinterpret_cast<unsigned*>(&ll);
}

// ---------------------------------------------------------------------------------------------
// implicit-GEMM gather kernel:
//   out[b, m, oy, ox] = scale * sum_k wp[k][m] * in[b, :, qy*istride, qx*istride][koff[k]]  (+ bias[m])
//   with (oy, ox) = (oy0 + qy*ostep, ox0 + qx*ostep); taps that fall outside the input contribute zero.
// ---------------------------------------------------------------------------------------------
struct IgPhase {
    const float* wp;             // packed weights [rows][ld]
    const int* koff;             // [rows]
    const unsigned char* ktap;   // [rows]
    const uint4* w3;             // [3][rows / 8][ld] chunks of 8 bf16, or [2][rows / 8][ld] chunks of 8 fp16 (see the panel layout)
    const unsigned* a_absmax;    // fp16-piece panels: bit pattern of the largest weight magnitude of this phase
    long long w3_plane;          // chunks per plane
    int K, Kpad, ld, T;
    int dmin;                    // the buffer descriptor of the gather starts dmin (<= 0) elements before the tensor
    int oy0, ox0, QH, QW;
    int TW, tw_magic;            // tap t = th*TW + tw, th = (t * tw_magic) >> 16  (exact for t < 32)
    int dy0, dys, dx0, dxs;      // tap (th, tw) reads input (qy*istride + dy0 + dys*th, qx*istride + dx0 + dxs*tw)
    // window kernels (convwin.hip): Tp = taps padded to the stage's unit count, NG = stages per window (one window = 8 reduction
    // channels, every tap), tile structure (TR output rows of NI images per 128-column tile), window rows WR per image and
    // tapc[t] = LDS slot displacement of tap t relative to a column's own slot
    int win_Tp, win_NG, win_TR, win_NI, win_WR;
    unsigned win_wrw_mul;        // division by WR * W (fastdiv)
    int win_wrw_s1, win_wrw_s2;
    int tapc[32] __attribute__((aligned(32)));      // (single-tap windows: tapc[i] = i * win_slotsp, the channel groups of a stage)
};

struct IgParams {
    const float* in;
    float* out;
    const float* bias;   // [M] or null
    const float* scale;  // multiplies the contraction (1/sigma of spectral norm) or null; one value, or one per GROUP of
    int scale_bg;        //   scale_bg consecutive batch elements (scale_bg = 0: a single value), scale_stride floats apart
    int scale_stride;
    long long in_bs, out_bs;
    unsigned in_bytes;   // extent of the gathered tensor view in bytes (< 2^31): bound of the gather's buffer descriptor
    int B, C, H, W;      // gathered tensor: C = reduction channels
    int M, OH, OW;       // produced tensor
    int istride, ostep, nphase;
    int ksplit;          // > 1: K is split over blockIdx.z; partial tiles go to `slab`
    float* slab;         //   combine == 0: dense [ksplit][B, M, OH, OW], summed by igemm_slab_reduce_kernel
    long long slab_stride;
    int precision;       // 0: fp32-faithful (three bf16 pieces per operand, six MFMAs per slice); 1: bf16 operands (one piece);
                         // 2: fp32-faithful with two scaled fp16 pieces per operand, three MFMAs per slice (panel format 1)
    const unsigned* b_absmax;   // precision 2: largest magnitude of the gathered tensor as AMAX_WORDS words of bit patterns (common.h)
    // activated second output (1x1-map layers, skinny_rows_kernel only): act_out[n, j] = RootTanh(out[n, j]) with its own row
    // stride, and - a style-chain link writing the NEXT link's input [latent | activation] (libs/block.py:119-125) - the
    // lat_z latent columns copied in front of it: act_out - lat_z is then the start of that row
    // On any other map (no latent): act_out[b, m, pixel] = RootTanh(out[b, m, pixel]), batch stride act_bs (igemm_epilogue,
    // conv_pointwise_kernel, igemm_slab_reduce_kernel) - the activation between the two convs of a stage, without a launch and
    // a read of its own.
    float* act_out = nullptr;
    long long act_bs = 0;
    const float* lat = nullptr;
    long long lat_bs = 0;
    int lat_z = 0;
    // The mirror image for an input-gradient launch whose input tensor WAS such an activation a = RootTanh(pre): the epilogue
    // multiplies by RootTanh'(pre) - out[b, c, pixel] = roottanh_grad(mul_pre[b, c, pixel], contraction) - so what leaves the launch
    // is the gradient of `pre` (batch stride mul_bs).
    const float* mul_pre = nullptr;
    long long mul_bs = 0;
    // AMAX_WORDS words (zero before the launch) that receive the largest magnitude of act_out (or, with mul_pre, of out)
    unsigned* out_absmax = nullptr;
    int combine;         //   combine == 1: [ksplit][tile][fragment][thread][4] (every store / load instruction of the block is
    unsigned* counters;  //   one contiguous KiB), summed INSIDE this launch by the tile's last-arriving block (counters[tile])
    int tile_nphase;     // nphase, or 1: the phase-fastest tile order switched off (debug library, LOCATE_DISABLE=phasefast)
    int win;             // window panel / window kernel (convwin.hip); win_U = chunk rows (units) per stage, win_slotsp = LDS slots
    int win_dbg;         // (debug library: experiment switches of the window kernel)
    int win_U, win_slotsp, win_bm, win_bn;   // per 8-channel window image (largest phase, the zero slot included), win_bm x win_bn = tile
    IgPhase ph[4];
};

// Shared epilogue of the implicit-GEMM kernels (both MFMA flavours have the same 32x32 accumulator layout).
template <int WGM, int WGN, int TM, int TN>
__device__ __forceinline__ void igemm_col_scales(const IgParams& p, const IgPhase& ph, float (&col_scale)[TN], int N, int n0,
                                                 int wn, int lane) {
    // 1/sigma of each of this lane's output columns (the batch may stack several forwards).  Loaded BEFORE the K loop:
    // at its end it would be one more dependent memory round trip on the critical path of every block.
    const int lcol = lane & 31;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nj = n0 + (wn * TN + j) * 32 + lcol;
        const int nn = nj < N ? nj : 0;
        const int b = nn / (ph.QH * ph.QW);
        col_scale[j] = p.scale ? p.scale[(p.scale_bg ? b / p.scale_bg : 0) * p.scale_stride] : 1.0f;
    }
}

template <int WGM, int WGN, int TM, int TN>
__device__ __forceinline__ void igemm_epilogue(const IgParams& p, const IgPhase& ph, f32x16 (&acc)[TM][TN],
                                               const float (&col_scale)[TN], int N, int n0, int m0, int zsplit, int wm,
                                               int wn, int lane, float* stage, bool split) {
    const int lrow = lane >> 5, lcol = lane & 31;
    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const long long plane = (long long)p.OH * p.OW;
    // Small output planes (linears and the deep layers: 1x1 ... 4x4 maps): the lanes of an accumulator register hold 32
    // different columns n = (b, pixel), whose addresses are a whole channel stack apart - 64 scattered 4-byte stores per
    // instruction (measured: 5-11 us of a 10 us launch).  There the 32x32 tile goes through a wave-private LDS patch and
    // is written out along (m, pixel), which is contiguous inside one batch element.
    const int iplane = (int)plane;
    if (stage != nullptr && p.nphase == 1 && p.ostep == 1 && iplane <= 16 && (iplane & (iplane - 1)) == 0 &&
        ph.QH * ph.QW == iplane) {
        const int lp = __ffs(iplane) - 1;
        const bool fused_s = !split && (p.act_out != nullptr || p.mul_pre != nullptr);
        float am_s = 0.0f;
        float* out_base = split ? p.slab + (long long)zsplit * p.slab_stride : p.out;
        const long long obs = split ? (long long)p.M * plane : p.out_bs;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mt0 = m0 + (wm * TM + i) * 32;
            float bias_v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt0 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                const bool use = p.bias != nullptr && !split && m < p.M;
                const float* bp = use ? p.bias + m : p.in;
                bias_v[r] = use ? *bp : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nt0 = n0 + (wn * TN + j) * 32;          // multiple of 32, hence of the plane size
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = (r & 3) + 8 * (r >> 2) + 4 * lrow;
                    stage[lcol * 33 + ml] = split ? acc[i][j][r] : fmaf(acc[i][j][r], col_scale[j], bias_v[r]);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                const int b0 = nt0 >> lp;
                if (!fused_s) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int flat = e * 64 + lane;
                        const int bl = flat >> (5 + lp), rem = flat & ((32 << lp) - 1);
                        const int ml = rem >> lp, pix = rem & (iplane - 1);
                        const int nl = (bl << lp) + pix;
                        const float v = stage[nl * 33 + ml];
                        if (nt0 + nl < N && mt0 + ml < p.M) out_base[(long long)(b0 + bl) * obs + (long long)(mt0 + ml) * plane + pix] = v;
                    }
                } else if (WGM * WGN == 8 && TM * TN <= 2 && p.mul_pre != nullptr) {
                    // the fused forms (see below).  (The eight-wave instantiations - the ones small maps are launched
                    // with; the large tiles' register budget has no room for this and takes the rolled loop at the end.)
                    // The sixteen pre-activation loads of the tile go out together (one after the other
                    // they were sixteen exposed round trips: 33 us for a 16-block launch that takes 17 without the epilogue); the
                    // fences keep the compiler from hoisting every tile's loads to the top (the tall tiles then spill)
                    __builtin_amdgcn_sched_barrier(0);
                    float z[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int flat = e * 64 + lane;
                        const int bl = flat >> (5 + lp), rem = flat & ((32 << lp) - 1);
                        const int ml = rem >> lp, pix = rem & (iplane - 1);
                        const int nl = (bl << lp) + pix;
                        const bool ok = nt0 + nl < N && mt0 + ml < p.M;
                        z[e] = ok ? p.mul_pre[(long long)(b0 + bl) * p.mul_bs + (long long)(mt0 + ml) * plane + pix] : 0.0f;
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int flat = e * 64 + lane;
                        const int bl = flat >> (5 + lp), rem = flat & ((32 << lp) - 1);
                        const int ml = rem >> lp, pix = rem & (iplane - 1);
                        const int nl = (bl << lp) + pix;
                        if (nt0 + nl < N && mt0 + ml < p.M) {
                            const float v = roottanh_grad_f(z[e], stage[nl * 33 + ml]);
                            am_s = fmaxf(am_s, fabsf(v));
                            out_base[(long long)(b0 + bl) * obs + (long long)(mt0 + ml) * plane + pix] = v;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll 1
                    for (int e = 0; e < 16; ++e) {
                        const int flat = e * 64 + lane;
                        const int bl = flat >> (5 + lp), rem = flat & ((32 << lp) - 1);
                        const int ml = rem >> lp, pix = rem & (iplane - 1);
                        const int nl = (bl << lp) + pix;
                        float v = stage[nl * 33 + ml];
                        if (nt0 + nl < N && mt0 + ml < p.M) {
                            const long long rel = (long long)(mt0 + ml) * plane + pix;
                            if (p.mul_pre != nullptr) {
                                v = roottanh_grad_f(p.mul_pre[(long long)(b0 + bl) * p.mul_bs + rel], v);
                                am_s = fmaxf(am_s, fabsf(v));
                            }
                            out_base[(long long)(b0 + bl) * obs + rel] = v;
                            if (p.mul_pre == nullptr) {
                                const float a = roottanh_f(v);
                                p.act_out[(long long)(b0 + bl) * p.act_bs + rel] = a;
                                am_s = fmaxf(am_s, fabsf(a));
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
        }
        if (fused_s && p.out_absmax != nullptr) absmax_publish_wave(am_s, p.out_absmax);
        return;
    }
    const float* optr[TN];         // per column tile: address of (row 0, this lane's column); null beyond N
    long long xoff[TN];            // fused forms: offset of the same element in act_out / mul_pre relative to its offset in out
    const bool fused = !split && (p.act_out != nullptr || p.mul_pre != nullptr);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nj = n0 + (wn * TN + j) * 32 + lcol;
        const int nn = nj < N ? nj : 0;
        const int qhw = ph.QH * ph.QW;
        const int b = nn / qhw, q = nn - b * qhw;
        const int qy = q / ph.QW, qx = q - qy * ph.QW;
        const long long pix = (long long)(ph.oy0 + qy * p.ostep) * p.OW + (ph.ox0 + qx * p.ostep);
        const float* o = split ? p.slab + (long long)zsplit * p.slab_stride + (long long)b * p.M * plane + pix
                               : p.out + (long long)b * p.out_bs + pix;
        optr[j] = nj < N ? o : nullptr;
        xoff[j] = (long long)b * ((p.mul_pre ? p.mul_bs : p.act_bs) - p.out_bs);
    }
    float am = 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float bias_v[16];          // this lane's 16 rows of the row tile: loaded together, branch-free
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
            const bool use = p.bias != nullptr && !split && m < p.M;
            const float* bp = use ? p.bias + m : p.in;          // always a valid address; value discarded when unused
            bias_v[r] = use ? *bp : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float* o = const_cast<float*>(optr[j]);
            if (o == nullptr) continue;
            if (!fused) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                    if (m < p.M) o[(long long)m * plane] = split ? acc[i][j][r] : fmaf(acc[i][j][r], col_scale[j], bias_v[r]);
                }
            } else if (p.mul_pre != nullptr) {
                // (scheduling fences: without them the compiler hoists every tile's loads to the top and the tall tiles spill)
                const float* pre = p.mul_pre + ((o - p.out) + xoff[j]);
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    float z[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int m = m0 + (wm * TM + i) * 32 + ((r0 + r) & 3) + 8 * ((r0 + r) >> 2) + 4 * lrow;
                        z[r] = m < p.M ? pre[(long long)m * plane] : 0.0f;
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int m = m0 + (wm * TM + i) * 32 + ((r0 + r) & 3) + 8 * ((r0 + r) >> 2) + 4 * lrow;
                        const float v = roottanh_grad_f(z[r], fmaf(acc[i][j][r0 + r], col_scale[j], bias_v[r0 + r]));
                        if (m < p.M) { o[(long long)m * plane] = v; am = fmaxf(am, fabsf(v)); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                float* ao = p.act_out + ((o - p.out) + xoff[j]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                    const float v = fmaf(acc[i][j][r], col_scale[j], bias_v[r]);
                    const float a = roottanh_f(v);
                    if (m < p.M) { o[(long long)m * plane] = v; ao[(long long)m * plane] = a; am = fmaxf(am, fabsf(a)); }
                    if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if (fused && p.out_absmax != nullptr) absmax_publish_wave(am, p.out_absmax);
}

// XCD-aware tile order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs (private 4 MiB L2s), so
// neighbouring tiles - which share a weight-panel slice (same m tile) or a gathered slice (same n tile) - would each
// fetch it into a different L2.  Remap so that every XCD walks a contiguous chunk of the x-fastest tile order
// (bijective for any grid size): the co-resident blocks of an XCD then share their operand slices through its L2.
// nphase > 1 (the sub-pixel phases of a stride-2 adjoint): the PHASE is the fastest index of that order - the 2 x 2 phases of one
// tile write interleaved pixels of the same output lines (each phase alone leaves every other 4-byte word of a line: half-written
// lines cost the memory side twice their bytes, profiles/r03_conv_pmc_mem.json: WRITE_SIZE = 2.0 x the output) and gather the same
// input slice; as neighbours on one XCD their stores meet in its L2 before the lines leave and the slice is fetched once.
// bz = phase * ksplit + split as before.
__device__ __forceinline__ void xcd_tile(int& bx, int& by, int& bz, int nphase = 1) {
    const int nx = gridDim.x, ny = gridDim.y;
    const int nwg = nx * ny * (int)gridDim.z;
    const int lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const int q = nwg >> 3, r = nwg & 7, xcd = lin & 7;
    int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    int phase = 0;
    if (nphase > 1) { phase = swz % nphase; swz /= nphase; }
    bx = swz % nx;
    const int t = swz / nx;
    by = t % ny;
    bz = t / ny;
    if (nphase > 1) bz += phase * ((int)gridDim.z / nphase);
}

// Per-thread state of the implicit-GEMM gather.  The gathered element of (column n, row k = (c, t)) sits at
//   in + [b(n) in_bs + iy0(n) W + ix0(n)]  +  [c H W + dy(t) W + dx(t)]
// = a lane part (fixed for the whole K loop) + a wave-uniform part (from the offset table): exactly the
// voffset + soffset of a buffer load, whose descriptor also does the zero padding - a lane whose tap falls outside the
// input passes an out-of-range voffset and gets 0 back without touching memory.  Per element that leaves three vector
// instructions (mask bit -> voffset select) instead of the ~9 of 64-bit address arithmetic + two selects.
struct GatherCol {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned lane_off;   // bytes
    unsigned outside;    // bit t: tap t reads OUTSIDE the input for this column (bit 31 is always set)
};

__device__ __forceinline__ GatherCol gather_setup(const IgParams& p, const IgPhase& ph, int n, int N) {
    GatherCol g;
    g.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in + ph.dmin), 0, (int)(p.in_bytes - 4 * ph.dmin), 0x00020000);
    const bool n_ok = n < N;
    const int nn = n_ok ? n : 0;
    const int qhw = ph.QH * ph.QW;
    const int b = nn / qhw, q = nn - b * qhw;
    const int qy = q / ph.QW, qx = q - qy * ph.QW;
    const int iy0 = qy * p.istride, ix0 = qx * p.istride;
    g.lane_off = (unsigned)(4 * ((long long)b * p.in_bs + (long long)iy0 * p.W + ix0));
    unsigned inside = 0;
    if (n_ok) {
        for (int t = 0; t < ph.T; ++t) {
            const int th = (t * ph.tw_magic) >> 16, tw = t - th * ph.TW;
            const bool ok = (unsigned)(iy0 + ph.dy0 + ph.dys * th) < (unsigned)p.H &&
                            (unsigned)(ix0 + ph.dx0 + ph.dxs * tw) < (unsigned)p.W;
            inside |= (ok ? 1u : 0u) << t;
        }
    }
    g.outside = ~inside;
    return g;
}

// soff: table entry of row k (bytes, wave-uniform); tap: its tap index (wave-uniform; 31 = never valid)
__device__ __forceinline__ float gather_load(const GatherCol& g, int soff, unsigned tap) {
    // two VALU ops per element: the tap's "outside" bit moves to bit 31 of the offset (lane_off < 2^31), which puts the
    // access beyond num_records - the buffer load then returns 0 without touching memory (zero padding)
    const unsigned vo = ((g.outside << (31u - tap)) & 0x80000000u) | g.lane_off;
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g.rsrc, (int)vo, soff, 0));
}

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// NOTE (hipcc 7.2 / clang 22): __builtin_bit_cast applied DIRECTLY to an element of an ext_vector (`bit_cast(unsigned, v[k])`)
// compiles to element 0 for every k.  Use __float_as_uint / __uint_as_float on vector elements, or copy to a scalar first.


// n / d for 0 <= n < 2^31 as t = mulhi(n, mul); (t + ((n - t) >> s1)) >> s2   (Granlund-Montgomery)
__device__ __forceinline__ int fastdiv(int n, unsigned mul, int s1, int s2) {
    const unsigned t = __umulhi((unsigned)n, mul);
    return (int)((t + (((unsigned)n - t) >> s1)) >> s2);
}
static void fastdiv_make(unsigned d, unsigned* mul, int* s1, int* s2) {
    if (d <= 1) { *mul = 0; *s1 = 0; *s2 = 0; return; }
    int l = 0;
    while ((1ull << l) < d) ++l;
    *mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    *s1 = 1; *s2 = l - 1;
}

// ---------------------------------------------------------------------------------------------
// Window kernels (convwin.hip): shared planning (conv.hip packs the panels and fills the phase tables, convwin.hip launches)
// ---------------------------------------------------------------------------------------------
#define WIN_TAIL_UNITS 16          // zero chunk rows behind the last unit: the A prefetch runs two stages (<= 2 x 8 rows) ahead

// Tile of the window kernels: rows BM x columns BN of four waves.  Every wave tile is at least 64 columns wide (two B fragments per
// A fragment read): 192 x 128 (waves 96 x 64) and 128 x 128 (64 x 64) for the wide layers, 96 / 64 / 32 rows x 256 columns below.
static inline int win_pick_bm(int M) {
    if (M >= 192 && M % 192 == 0) return 192;
    const int cands[4] = {128, 96, 64, 32};
    int best = 128, best_pad = 1 << 30;
    for (int i = 0; i < 4; ++i) {
        const int pad = (M + cands[i] - 1) / cands[i] * cands[i];
        if (pad < best_pad) { best_pad = pad; best = cands[i]; }
    }
    return best;
}
static inline int win_pick_bn(int bm) { return bm >= 128 ? 128 : 256; }
// 16-deep slices per stage: two while the padding of the unit count (taps, or 8-channel groups of a single-tap layer) to a multiple
// of four stays under 15 %, else one
static inline int win_pick_sl(const int* X, int nx) {
    long long real = 0, padded = 0;
    for (int i = 0; i < nx; ++i) { real += X[i]; padded += (X[i] + 3) / 4 * 4; }
    return (padded - real) * 100 <= 15 * real ? 2 : 1;
}

static inline int win_pack_cs(int KK) { return KK <= 4 ? 64 : (KK <= 16 ? 32 : 16); }
static inline __host__ __device__ int win_pack_rg(int KK) { return KK == 1 ? 64 : 16; }
static inline size_t win_panel_floats(int urows, int ld, int fmt) { return PANEL_HDR + (size_t)(fmt ? 2 : 3) * urows * ld * 4; }

int launch_win_igemm(IgParams& p, int nmax, void* slab_ws, unsigned* counters, hipStream_t st, const char* who);
void launch_fp8_igemm(const IgParams& p, dim3 grid, int bm, hipStream_t st);
size_t win_slab_floats(const IgParams& p, int nmax);
void launch_slab_reduce(const IgParams& p, hipStream_t st);
