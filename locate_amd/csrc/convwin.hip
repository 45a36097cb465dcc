// LDS-window contractions: the dense convolutions of the LocAtE hot path (reference libs/conv.py:14-20, libs/attention.py:44-46,
// libs/scale.py:25-34 - the same regular convolution R and its data adjoint as conv.hip) with the activation operand staged in
// LDS ONCE per block and 8 reduction channels, as the raw window of the input map the block's 128 output columns can see:
//
//     window image  [piece][slot][8 halves]      slot = (image, input row, input column) of the window, 8 channels per chunk
//
// The MFMA B fragment of output column n at tap (dy, dx) is then simply the chunk at  slot(n) + displacement(dy, dx)  - a
// per-lane base plus a wave-uniform constant per tap - so every gathered, split and LDS-written activation element serves ALL
// taps of the kernel (4 for the sub-pixel phases of a 4x4 stride-2 transposed conv, 9 for a 3x3, 16 / 25 for the stride-2
// convs) instead of one: the implicit-GEMM kernels of conv.hip gather an im2col image, i.e. load, split and write every element
// once PER TAP, and are bound by exactly that (the vector-memory pipe takes ~34 cycles per 4-byte-per-lane gather instruction;
// profiles/notes_r04_experiments.md).  Taps that fall outside the input read a zero chunk instead (per-column tap masks), so the
// window holds in-bounds pixels only - full rows of the map, which makes the global side of the copy contiguous runs loaded 8
// bytes per lane.  For stride-2 gathers the window's columns are stored de-interleaved by parity (even columns, then odd), which
// keeps the fragment reads of neighbouring output columns on neighbouring chunks (conflict-free ds_read_b128).
//
// The weight operand comes from a WINDOW PANEL (conv.hip, pack_win_body): chunk rows in "unit" order u = c8g * Tp + t, so that
// the 16-deep MFMA slice (units 2i, 2i + 1 - lane half h takes unit 2i + h) pairs two taps of the same 8 channels (or, for a
// single-tap layer, two 8-channel groups).  One stage = U = 2 SL units; one window = 8 channels x every tap = NG stages (a
// single-tap layer: U channel groups, one stage).  The next window's loads are spread over the current window's stages.
//
// Numerics are those of conv.hip's kernels: NP = 2 (two scaled fp16 pieces, three MFMAs per slice), NP = 3 (three bf16 pieces,
// six MFMAs), NP = 1 (bf16 operands).
#include "igemm.h"

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int WGM, int WGN, int TM, int TN, int NP, int SL, int BPT>
__global__ void __launch_bounds__(256, 2) conv_win_kernel(const IgParams p) {
    constexpr int NT = 256, NW = 4, U = 2 * SL;
    constexpr int BM = WGM * TM * 32;
    constexpr int BN = WGN * TN * 32;
    static_assert(WGM * WGN == NW && (BN == 128 || BN == 256), "four waves on a 128- or 256-column tile");
    constexpr int A_CH = U * BM;                         // weight chunks per piece and stage
    constexpr int A_PT = (A_CH + NT - 1) / NT;           // ... per thread
    constexpr int A_U4 = 2 * NP * A_CH;                  // both stage buffers
    static_assert(2 * U <= WIN_TAIL_UNITS, "panel tail shorter than the prefetch distance");

    extern __shared__ uint4 smem[];                      // As[2][NP][U][BM] | Bs[2][NP][WINC][SP]; then the epilogue's patches

    int bx, by, bz;
    xcd_tile(bx, by, bz, p.tile_nphase);
    const int zphase = bz / p.ksplit, zsplit = bz - zphase * p.ksplit;
    const IgPhase& ph = p.ph[zphase];
    const int QHW = ph.QH * ph.QW;
    const int N = p.B * QHW;
    const int n0 = bx * BN;
    const int m0 = by * BM;
    if (n0 >= N) return;   // phases can have different extents; uniform per block

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int lrow = lane >> 5, lcol = lane & 31;

    const int T = ph.T, Tp = ph.win_Tp, NG = ph.win_NG;
    const int WINC = Tp == 1 ? U : 1;                    // 8-channel groups per window image
    const int SP = p.win_slotsp;
    const int HW = p.H * p.W;
    const int WR = ph.win_WR, WRW = WR * p.W, NI = ph.win_NI;
    const int C8G = (p.C + 7) >> 3;
    // ---- the tile's images / rows and its window
    int b0, qy0;
    if (NI == 1) { b0 = n0 / QHW; qy0 = (n0 - b0 * QHW) / ph.QW; }
    else { b0 = bx * NI; qy0 = 0; }
    const int TH = T / ph.TW;
    const int dyA = ph.dy0, dyB = ph.dy0 + ph.dys * (TH - 1);
    const int dymin = dyA < dyB ? dyA : dyB;
    int iy_lo = qy0 * p.istride + dymin;
    if (iy_lo < 0) iy_lo = 0;
    const int slots = NI * WRW, SV = slots >> 1;

    // ---- per-lane columns: slot of the column's own pixel and the taps that fall outside the input
    int lbase[TN];
    unsigned outs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + lcol;
        const bool ok = n < N;
        const int nn = ok ? n : n0;
        const int b = nn / QHW, q = nn - b * QHW;
        const int qy = q / ph.QW, qx = q - qy * ph.QW;
        const int iy0 = qy * p.istride, ix0 = qx * p.istride;
        unsigned inside = 0;
        for (int t = 0; t < T; ++t) {
            const int th = (t * ph.tw_magic) >> 16, tw = t - th * ph.TW;
            const bool in = (unsigned)(iy0 + ph.dy0 + ph.dys * th) < (unsigned)p.H && (unsigned)(ix0 + ph.dx0 + ph.dxs * tw) < (unsigned)p.W;
            inside |= (in ? 1u : 0u) << t;
        }
        outs[j] = ok ? (Tp == 1 ? 0u : ~inside) : ~0u;          // (a single-tap window: bit i = channel group i, all inside)
        lbase[j] = ((b - b0) * WR + iy0 - iy_lo) * p.W + qx;
    }
    float col_scale[TN];
    igemm_col_scales<WGM, WGN, TM, TN>(p, ph, col_scale, N, n0, wn, lane);
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

    // ---- K range of this block: whole windows
    const int nwin_total = Tp == 1 ? (C8G + U - 1) / U : C8G;
    const int per_split = (nwin_total + p.ksplit - 1) / p.ksplit;
    const int win0 = zsplit * per_split;
    int nwin = nwin_total - win0;
    if (nwin > per_split) nwin = per_split;
    if (nwin < 0) nwin = 0;
    const int nsteps = nwin * NG;

    // ---- weight chunks of this thread: a constant per-lane offset into the panel's planes; the stage and the piece plane travel
    //      in the buffer load's SCALAR offset (no address arithmetic in the loop)
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(ph.w3), 0, (int)((unsigned)NP * (unsigned)ph.w3_plane * 16u), 0x00020000);
    unsigned a_voff[A_PT];
    int a_dst[A_PT];
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
        // (a thread beyond the tile's chunk count repeats an earlier thread's chunk: same value to the same LDS address - no
        // predicate, hence no control flow in the loop)
        const int idc = (tid + i * NT) % A_CH;
        const int kb = idc / BM, m = idc - kb * BM;
        a_dst[i] = kb * BM + m;
        a_voff[i] = 16u * (unsigned)(kb * ph.ld + m0 + m);
    }
    const unsigned a_step = 16u * (unsigned)(U * ph.ld), a_plane = 16u * (unsigned)ph.w3_plane;
    unsigned a_soff = (unsigned)(win0 * NG) * a_step;          // wave-uniform: the stage whose tile is loaded next
    // named registers: as an array this operand stage is kept in scratch memory by clang (see conv_igemm_bx6_kernel)
    static_assert(A_PT <= 3, "at most three weight chunks per thread, piece and stage");
    uint4 a00, a01, a02, a10, a11, a12, a20, a21, a22;
    auto aload = [&](unsigned vo, unsigned so) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)vo, (int)so, 0);
        return make_uint4(v[0], v[1], v[2], v[3]);
    };
    auto issue_A = [&]() {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)a_soff);
        a00 = aload(a_voff[0], so);
        if constexpr (NP >= 2) a01 = aload(a_voff[0], so + a_plane);
        if constexpr (NP == 3) a02 = aload(a_voff[0], so + 2 * a_plane);
        if constexpr (A_PT >= 2) {
            a10 = aload(a_voff[A_PT >= 2 ? 1 : 0], so);
            if constexpr (NP >= 2) a11 = aload(a_voff[A_PT >= 2 ? 1 : 0], so + a_plane);
            if constexpr (NP == 3) a12 = aload(a_voff[A_PT >= 2 ? 1 : 0], so + 2 * a_plane);
        }
        if constexpr (A_PT == 3) {
            a20 = aload(a_voff[A_PT - 1], so);
            if constexpr (NP >= 2) a21 = aload(a_voff[A_PT - 1], so + a_plane);
            if constexpr (NP == 3) a22 = aload(a_voff[A_PT - 1], so + 2 * a_plane);
        }
        a_soff += a_step;
    };
    auto store_A = [&](int buf) {
        uint4* As = smem + buf * NP * A_CH;
        As[a_dst[0]] = a00;
        if constexpr (NP >= 2) As[A_CH + a_dst[0]] = a01;
        if constexpr (NP == 3) As[2 * A_CH + a_dst[0]] = a02;
        if constexpr (A_PT >= 2) {
            As[a_dst[A_PT >= 2 ? 1 : 0]] = a10;
            if constexpr (NP >= 2) As[A_CH + a_dst[A_PT >= 2 ? 1 : 0]] = a11;
            if constexpr (NP == 3) As[2 * A_CH + a_dst[A_PT >= 2 ? 1 : 0]] = a12;
        }
        if constexpr (A_PT == 3) {
            As[a_dst[A_PT - 1]] = a20;
            if constexpr (NP >= 2) As[A_CH + a_dst[A_PT - 1]] = a21;
            if constexpr (NP == 3) As[2 * A_CH + a_dst[A_PT - 1]] = a22;
        }
    };

    // ---- window loader: item = (8-channel group wi of the window, pixel pair sv); part g of a window = items
    //      g BPT NT + i NT + tid.  The per-lane offset is the pixel pair's position (sign bit set = masked: an out-of-range voffset
    //      reads 0 without touching memory); window and channel travel in the scalar offset, which the descriptor's range check
    //      does not cover - hence C % 8 == 0 (win_finish) and the explicit masks on channel groups past the last one.
    //      An item's offset and LDS slot do not depend on the window: they are derived once (divisions, the stride-2 de-interleave)
    //      into a table in LDS, [part][item] -> {voffset, slot | group << 16}; the loop reads its entry back - no control flow and a
    //      handful of vector instructions per stage.  Masked items write their chunks to a trash slot (SP - 2).
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
    u32x2 breg[BPT][8];
    int bphys[BPT];               // chunk index (inside one piece plane of the window image) of the pair's first pixel (or the trash slot)
    bool bon[BPT];
    const unsigned HW4 = 4u * (unsigned)HW;
    uint2* const itab = reinterpret_cast<uint2*>(smem + A_U4 + 2 * NP * WINC * SP);          // [NG BPT NT] behind the images
    for (int e = tid; e < NG * BPT * NT; e += NT) {
        const int it = e;
        int wi = 0, sv = it;
        if (WINC > 1) { wi = it / SV; sv = it - wi * SV; }
        bool valid = it < WINC * SV;
        const int s2 = 2 * sv;
        int bi = 0, rem = s2;
        if (NI > 1) { bi = fastdiv(valid ? s2 : 0, ph.win_wrw_mul, ph.win_wrw_s1, ph.win_wrw_s2); rem = s2 - bi * WRW; }
        const int pix = iy_lo * p.W + rem;
        valid = valid && (b0 + bi) < p.B && pix < HW;
        const unsigned voff = ((unsigned)(4 * ((long long)(b0 + bi) * p.in_bs + pix)) + (unsigned)wi * 8u * HW4) | (valid ? 0u : 0x80000000u);
        int f = s2;
        if (p.istride == 2) {          // de-interleave the window's columns by parity: x -> (x & 1) (W / 2) + x / 2
            const int x = s2 & (p.W - 1);          // (W is a power of two: win_finish)
            f = (s2 - x) + (x >> 1);
        }
        itab[e] = make_uint2(voff, (unsigned)(wi * SP + f) | ((unsigned)wi << 16));
    }
    const int odd_off = p.istride == 2 ? p.W >> 1 : 1;          // chunk distance between the two pixels of a pair
    auto issue_B = [&](int wrel, int g) {
        const int wglob = __builtin_amdgcn_readfirstlane(win0 + wrel);
        const int gleft = __builtin_amdgcn_readfirstlane(C8G - wglob * WINC);      // channel groups from this window's first one on (<= 0: past the end)
        const unsigned wso = (unsigned)(wglob * WINC) * 8u * HW4;
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const uint2 ent = itab[(g * BPT + i) * NT + tid];
            const bool ok = (int)(ent.y >> 16) < gleft && (int)ent.x >= 0;
            const unsigned voff = ok ? ent.x : 0x80000000u;
            bphys[i] = (int)(ent.y & 0xffffu);
            bon[i] = ok;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                breg[i][e] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)(wso + (unsigned)e * HW4), 0));
        }
    };
    auto store_B = [&](int buf) {
        uint4* Bs = smem + A_U4 + buf * NP * WINC * SP;
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = __uint_as_float(v == 0 ? breg[i][e].x : breg[i][e].y);
                uint4* dst = Bs + (bon[i] ? bphys[i] + v * odd_off : SP - 2);
                if constexpr (NP == 2) {
                    uint4 h, l;
                    split2_f16x8(f, b_scale, h, l);
                    dst[0] = h;
                    dst[WINC * SP] = l;
                } else if constexpr (NP == 3) {
                    uint4 h, m, l;
                    split3_trunc_x8(f, h, m, l);
                    dst[0] = h;
                    dst[WINC * SP] = m;
                    dst[2 * WINC * SP] = l;
                } else {
                    bf16x8 h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (__bf16)f[e];
                    dst[0] = *reinterpret_cast<uint4*>(&h);
                }
            }
        }
    };

    // ---- prologue: the zero chunks, window 0 in full, weight tile 0; then the loads of weight tile 1 and of window 1's first part
    if (Tp == 1 && (C8G % U) != 0) {
        // a single-tap window whose last stage holds fewer than U channel groups: the missing groups' chunks are never written (their
        // items go to the trash slot) but ARE read - against zero weight rows - and what an untouched LDS word holds need not be a
        // finite number (0 x NaN: the 24 -> 12 channel 1x1 layer of the tiny test network came out NaN): clear both images once
        for (int e = tid; e < 2 * NP * WINC * SP; e += NT) smem[A_U4 + e] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    }
    if (tid < 2 * NP) smem[A_U4 + tid * WINC * SP + SP - 1] = make_uint4(0u, 0u, 0u, 0u);       // slot SP - 1 of group 0, every buffer / piece
    __syncthreads();               // the item table
    if (nsteps > 0) {
        for (int g = 0; g < NG; ++g) {
            issue_B(0, g);
            store_B(0);
        }
        issue_A();
        store_A(0);
        issue_A();
        issue_B(1, 0);
    }
    __syncthreads();

    constexpr int PROD = NP == 3 ? 6 : (NP == 2 ? 3 : 1);
    using frag_t = typename std::conditional<NP == 2, f16x8, bf16x8>::type;
    int g = 0, wrel = 0;                                  // stage s = wrel NG + g
    for (int s = 0; s < nsteps; ++s) {
        const int abuf = s & 1, bbuf = wrel & 1;
#ifdef LOCATE_DEBUG_KNOBS
        const int dbg = p.win_dbg;
#endif
        const uint4* Asb = smem + abuf * NP * A_CH;
        const uint4* Bsb = smem + A_U4 + bbuf * NP * WINC * SP;
        // the stage's unit displacements (wave-uniform, one scalar load): tap constants, or the channel groups of a single-tap
        // window (the host fills tapc[i] = i SP for those)
        const int t0 = Tp == 1 ? 0 : g * U;
        int uo[U];
        {
            const int* tc = ph.tapc + __builtin_amdgcn_readfirstlane(t0);
            if constexpr (U == 2) { const i32x2 v = *reinterpret_cast<const i32x2*>(tc); uo[0] = v[0]; uo[U - 1] = v[1]; }
            else if constexpr (U == 4) { const i32x4 v = *reinterpret_cast<const i32x4*>(tc); uo[0] = v[0]; uo[1] = v[1]; uo[U - 2] = v[2]; uo[U - 1] = v[3]; }
            else { const i32x8 v = *reinterpret_cast<const i32x8*>(tc);
#pragma unroll
                   for (int i = 0; i < U; ++i) uo[i] = v[i & 7]; }
        }
        unsigned mj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) mj[j] = outs[j] >> t0;
        // fragments of slice sl into register set `set`: all reads issued together, ahead of the matrix work that uses them
        frag_t fa[2][TM][NP], fb[2][TN][NP];
        auto load_frags = [&](int sl, int set) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    fa[set][i][q] = *reinterpret_cast<const frag_t*>(&Asb[q * A_CH + (2 * sl + lrow) * BM + (wm * TM + i) * 32 + lcol]);
            const int uoff = lrow ? uo[2 * sl + 1] : uo[2 * sl];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const bool out = (mj[j] >> (2 * sl + lrow)) & 1u;
                const int addr = out ? SP - 1 : lbase[j] + uoff;
#pragma unroll
                for (int q = 0; q < NP; ++q) fb[set][j][q] = *reinterpret_cast<const frag_t*>(&Bsb[q * WINC * SP + addr]);
            }
        };
        auto mfmas = [&](int set, int lo, int hi) {
#ifdef LOCATE_DEBUG_KNOBS
            if (dbg & 1) return;             // experiment (debug library): no matrix work
#endif
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int base = (i * TN + j) * PROD;
                    frag_t (&a)[NP] = fa[set][i];
                    frag_t (&b)[NP] = fb[set][j];
                    if constexpr (NP == 2) {          // smallest terms first: l h, h l, h h
                        if (base + 0 >= lo && base + 0 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc[i][j], 0, 0, 0);
                        if (base + 1 >= lo && base + 1 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc[i][j], 0, 0, 0);
                        if (base + 2 >= lo && base + 2 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc[i][j], 0, 0, 0);
                    } else if constexpr (NP == 3) {
                        if (base + 0 >= lo && base + 0 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NP - 1], b[0], acc[i][j], 0, 0, 0);   // l h
                        if (base + 1 >= lo && base + 1 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[NP - 1], acc[i][j], 0, 0, 0);   // h l
                        if (base + 2 >= lo && base + 2 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NP - 2], b[NP - 2], acc[i][j], 0, 0, 0);   // m m
                        if (base + 3 >= lo && base + 3 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NP - 2], b[0], acc[i][j], 0, 0, 0);   // m h
                        if (base + 4 >= lo && base + 4 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[NP - 2], acc[i][j], 0, 0, 0);   // h m
                        if (base + 5 >= lo && base + 5 < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[i][j], 0, 0, 0);   // h h
                    } else {
                        if (base >= lo && base < hi) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[i][j], 0, 0, 0);
                    }
                }
        };
        auto next_tiles = [&]() {
#ifdef LOCATE_DEBUG_KNOBS
            if (dbg & 2) return;             // experiment (debug library): no loads / LDS writes in the loop
#endif
            // the NEXT tiles go to LDS (their loads were issued a stage ago) and the loads of the tiles after those are issued
            // into the registers just freed
            store_A(abuf ^ 1);                              // weight tile s + 1
            store_B(bbuf ^ 1);                              // part g of window wrel + 1
            issue_A();                                      // weight tile s + 2
            int g2 = g + 1, w2 = wrel + 1;
            if (g2 == NG) { g2 = 0; ++w2; }
            issue_B(w2, g2);                                // the part stored by the next iteration
        };
        constexpr int NMF = TM * TN * PROD;
        load_frags(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SL == 1) {
            mfmas(0, 0, NMF / 2);
            next_tiles();
            mfmas(0, NMF / 2, NMF);
        } else {
            // ONE scheduling region: slice 0's matrix work with slice 1's fragment reads and the next tiles' conversion / LDS writes
            // in its shadow, slice 1's with the loads of the tiles after those (one MFMA : one LDS op : a few VALU : one VMEM)
            load_frags(1, 1);
            mfmas(0, 0, NMF);
            next_tiles();
            mfmas(1, 0, NMF);
#pragma unroll
            for (int k = 0; k < 2 * NMF; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        if (++g == NG) { g = 0; ++wrel; }
        __syncthreads();
    }
    if constexpr (NP == 2) {       // undo the two power-of-two scales, one after the other (each exact)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] * a_unscale) * b_unscale;
    }
    // the operand images are dead after the loop's last barrier: each wave takes a 32 x 33 float patch of them
    float* const stage = reinterpret_cast<float*>(smem) + wid * (32 * 33);
    bool combined = false;
    if (p.ksplit > 1 && p.combine) {
        // split-K combined inside the launch: the protocol of conv_igemm_bx6_kernel (sc1 partial tiles, drained, one ticket per
        // block; the last arriver sums all partials in z order - bit-reproducible - and runs the epilogue)
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
                        const int gg = h0 + f, i = gg / (TN * 4), j = (gg / 4) % TN, q = gg % 4;
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

// ---- host side --------------------------------------------------------------------------------------------------------------
struct WinSplit {
    int ksplit, combine, gx, gy;
    size_t slab_floats;
};

static int win_nwin(const IgParams& p) {
    const int C8G = (p.C + 7) / 8;
    return p.ph[0].win_Tp == 1 ? (C8G + p.win_U - 1) / p.win_U : C8G;
}

// K is split over whole windows when the (M, N) tiling alone leaves most of the 512 resident blocks (two per CU) empty; at least
// two windows per block, partial tiles combined inside the launch while that stays a short serial read (<= 8 partials of one tile)
static WinSplit win_split_plan(const IgParams& p, int nmax, bool have_counters) {
    WinSplit sp;
    const int bm = p.win_bm;
    sp.gx = (nmax + p.win_bn - 1) / p.win_bn;
    sp.gy = (p.M + bm - 1) / bm;
    const long long tiles = (long long)sp.gx * sp.gy * p.nphase;
    const int nwin = win_nwin(p);
    int ks = 1;
    if (tiles < 384) {
        long long want = (512 + tiles - 1) / tiles;
        const int max_split = nwin / 2 > 0 ? nwin / 2 : 1;
        if (want > max_split) want = max_split;
        if (want > 16) want = 16;
        ks = want < 1 ? 1 : (int)want;
        // no empty splits: every block gets ceil(nwin / ks) windows
        const int per = (nwin + ks - 1) / ks;
        ks = (nwin + per - 1) / per;
    }
    sp.ksplit = ks;
    const long long tile_floats = (long long)bm * p.win_bn;
    const long long legacy = ks > 1 ? (long long)ks * p.B * p.M * p.OH * p.OW : 0;
    const long long fused = ks > 1 ? (long long)ks * tiles * tile_floats : 0;
    sp.combine = have_counters && ks > 1 && (long long)ks * tile_floats * 4 <= (512 << 10) && tiles <= 1024 && fused * 4 < (1ll << 31);
    sp.slab_floats = (size_t)(sp.combine ? fused : legacy);
    return sp;
}

size_t win_slab_floats(const IgParams& p, int nmax) {
    const size_t a = win_split_plan(p, nmax, false).slab_floats, b = win_split_plan(p, nmax, true).slab_floats;
    return a > b ? a : b;
}

template <int WGM, int WGN, int TM, int TN, int NP, int SL>
static void win_launch_bpt(const IgParams& p, dim3 grid, size_t lds, int bpt, hipStream_t st) {
    // (dynamic LDS beyond the default limit is enabled once per kernel, for the whole budget win_finish admits)
    if (bpt <= 1) {
        auto k = conv_win_kernel<WGM, WGN, TM, TN, NP, SL, 1>;
        static bool enabled = false;
        if (!enabled) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); enabled = true; }
        k<<<grid, 256, lds, st>>>(p);
    } else {
        auto k = conv_win_kernel<WGM, WGN, TM, TN, NP, SL, 2>;
        static bool enabled = false;
        if (!enabled) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); enabled = true; }
        k<<<grid, 256, lds, st>>>(p);
    }
}

template <int NP, int SL>
static void win_launch_bm(const IgParams& p, dim3 grid, size_t lds, int bpt, hipStream_t st) {
    const int bm = p.win_bm;
    if (p.win_bn == 128 && bm == 96) win_launch_bpt<1, 4, 3, 1, NP, SL>(p, grid, lds, bpt, st);
    else if (p.win_bn == 128 && bm == 64) win_launch_bpt<1, 4, 2, 1, NP, SL>(p, grid, lds, bpt, st);
    else if (p.win_bn == 128 && bm == 32) win_launch_bpt<1, 4, 1, 1, NP, SL>(p, grid, lds, bpt, st);
    else if (bm == 192) win_launch_bpt<2, 2, 3, 2, NP, SL>(p, grid, lds, bpt, st);
    else if (bm == 128) win_launch_bpt<2, 2, 2, 2, NP, SL>(p, grid, lds, bpt, st);
    else if (bm == 96) win_launch_bpt<1, 4, 3, 2, NP, SL>(p, grid, lds, bpt, st);
    else if (bm == 64) win_launch_bpt<1, 4, 2, 2, NP, SL>(p, grid, lds, bpt, st);
    else win_launch_bpt<1, 4, 1, 2, NP, SL>(p, grid, lds, bpt, st);
}

template <int NP>
static void win_launch_sl(const IgParams& p, dim3 grid, size_t lds, int bpt, hipStream_t st) {
    if (p.win_U == 2) win_launch_bm<NP, 1>(p, grid, lds, bpt, st);
    else win_launch_bm<NP, 2>(p, grid, lds, bpt, st);
}

int launch_win_igemm(IgParams& p, int nmax, void* slab_ws, unsigned* counters, hipStream_t st, const char* who) {
    const WinSplit sp = win_split_plan(p, nmax, counters != nullptr);
    p.ksplit = sp.ksplit;
    p.combine = sp.combine;
    p.counters = counters;
    p.slab = static_cast<float*>(slab_ws);
    p.slab_stride = (long long)p.B * p.M * p.OH * p.OW;
    LOCATE_REQUIRE(p.ksplit == 1 || slab_ws, "%s: split-K needs a workspace", who);
    p.win_dbg = knob_int("LOCATE_WIN_DBG", 0);
    const int NP = p.precision == 2 ? 2 : (p.precision == 1 ? 1 : 3);
    const int U = p.win_U, bm = p.win_bm;
    const bool single = p.ph[0].win_Tp == 1;
    for (int i = 0; i < p.nphase; ++i)
        LOCATE_REQUIRE(round_up(p.M, bm) <= p.ph[i].ld, "%s: tile height %d does not divide the panel width %d", who, bm, p.ph[i].ld);
    // items per thread and stage of the window loader (win_finish admitted at most two)
    int bpt = 1;
    for (int i = 0; i < p.nphase; ++i) {
        const IgPhase& ph = p.ph[i];
        const long long items = (long long)(single ? U : 1) * (ph.win_NI * ph.win_WR * p.W / 2);
        const int need = (int)((items + 256ll * ph.win_NG - 1) / (256ll * ph.win_NG));
        if (need > bpt) bpt = need;
    }
    LOCATE_REQUIRE(bpt <= 2, "%s: window of %d items per thread", who, bpt);
    int ng_max = 1;
    for (int i = 0; i < p.nphase; ++i) ng_max = p.ph[i].win_NG > ng_max ? p.ph[i].win_NG : ng_max;
    size_t lds = ((size_t)2 * NP * U * bm + (size_t)2 * NP * (single ? U : 1) * p.win_slotsp) * 16 + (size_t)ng_max * bpt * 256 * 8;
    const size_t epi = (size_t)4 * 32 * 33 * 4 + 64;
    if (lds < epi) lds = epi;
    LOCATE_REQUIRE(lds <= 160 * 1024, "%s: window images of %zu bytes exceed the LDS", who, lds);
    dim3 grid(sp.gx, sp.gy, p.nphase * p.ksplit);
    if (NP == 2) win_launch_sl<2>(p, grid, lds, bpt, st);
    else if (NP == 3) win_launch_sl<3>(p, grid, lds, bpt, st);
    else win_launch_sl<1>(p, grid, lds, bpt, st);
    LOCATE_LAUNCH_CHECK(who);
    if (p.ksplit > 1 && !p.combine) {
        LOCATE_REQUIRE(p.slab_stride < (1ll << 31), "%s: split-K output of %lld elements exceeds the 32-bit index range", who, p.slab_stride);
        launch_slab_reduce(p, st);
        LOCATE_LAUNCH_CHECK(who);
    }
    return LOCATE_OK;
}
