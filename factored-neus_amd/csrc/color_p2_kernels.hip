// Forward of the colour network (RenderingNetwork.forward, mode 'idr': reference models/fields.py:150-175 via renderer.py:278)
// in the two-pass pipelined form (p2_engine.h, p2_relu.h): the maths, operands and per-accumulator summation order of
// color_fwd_tph_kernel; chip-filling launches only (the RefColor heads and small launches stay on color_kernels.hip).
//   layer 0: [feature 256 | pts, PE4(view), normal: 33 -> 48] -> 256 ReLU (19 k-steps), layers 1..3: 256 -> 256 ReLU,
//   layer 4: 256 -> 3 as vector dot products on the accumulators + a fixed-order sum over the waves, sigmoid.
// A unit = 128 samples = 4 tiles in two sets A = {0, 1}, B = {2, 3}; wave w of 8 owns output tile w.
//   L0.A || tail of the previous unit (ReLU 3 of set B -> dots)      L0.B || ReLU 0 A
//   Ll.A || ReLU l-1 B                                                Ll.B || ReLU l A       (l = 1..3; ReLU 3 -> dots)
// The 19 input k-steps of a set are written when the set's last reader of the running unit has finished: the feature rows
// (fp32 [n][256]) are split into hi / lo fragments by all 8 waves (k-steps 2w, 2w + 1 of both tiles), the side inputs by two.
#include <stdlib.h>
#ifndef FNEUS_P2_DEPTH
#define FNEUS_P2_DEPTH 2            // weight-prefetch distance of this file's passes (p2_engine.h: 3).  30 spilled registers at depth 3, 8 at depth 2: 2 us
#endif                              // (tools/runs/r04_ab.sh k2d2 / k2d1 / cold2)
#include "p2_relu.h"
#include "fneus_kernels.h"
#include "color_p2.h"

namespace fneus {

constexpr int kColP2Red = kP2LdsTotal;                       // float [2 tiles][3 rows][8 waves][32]
constexpr int kColP2LdsTotal = kColP2Red + 2 * 3 * 8 * 32 * 4;

template <int PREC, int MODE>
__global__ void __launch_bounds__(512, 1) color_fwd_p2_kernel(const unsigned char* blob, PointSrc src, long N,
                                                              const float* __restrict__ dirs, const float* __restrict__ normal,
                                                              const float* __restrict__ feat, ColStash st,
                                                              float* __restrict__ rgb_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    constexpr int TN = 1, NW = 8;
    constexpr int NPL = PREC == 3 ? 2 : 1;
    constexpr bool TRAIN = MODE != 0;
    constexpr bool LO = MODE == 3 && PREC == 3;
    float* red = reinterpret_cast<float*>(lds_ + kColP2Red);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int t0 = wave, r = lane & 31, h = lane >> 5;
    constexpr auto& LY = kColLayout;
    const long units = (N + 127) / 128;
    const long tiles = pp_tiles(N);
    const PPLane pl = pp_lane(lane);
    const unsigned voff_mask = (unsigned)lane * 16u + (unsigned)(wave >> 1) * 4u + (unsigned)(wave & 1) * 2u;
    // ---- the 19 input k-steps of tiles ta, ta + 1 of a unit
    auto encode = [&](long unit, int ta) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {                    // feature rows -> k-steps 2w, 2w + 1 of both tiles
            const long tile = unit * 4 + ta + k;
            const long n = tile * 32 + r;
            const long nc = n < N ? n : N - 1;
            unsigned char* dst = lds_ + (ta + k) * kP2Half + lane * 16;
            if (feat == nullptr) {
                // round 6: the SDF kernel's feature PLANES (hi + lo fragments, what this loop would build from the fp32 rows -- the same
                // split of the same values): 16-byte fragment loads straight into the k-steps, no conversion
                const long tl = tile < tiles ? tile : tiles - 1;        // (a tile beyond the launch: any block, its samples are masked)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int ks = 2 * wave + q;
                    *reinterpret_cast<bf16x8*>(dst + (ks * NPL) * kFragBytes) = pp_load(st.feat_hi + (size_t)tl * kPPBlock, ks, pl);
                    if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(dst + (ks * NPL + 1) * kFragBytes) = pp_load(st.feat_lo + (size_t)tl * kPPBlock, ks, pl);
                }
                continue;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ks = 2 * wave + q;
                const f32x4 a = *reinterpret_cast<const f32x4*>(feat + nc * 256 + 16 * ks + 4 * h);
                const f32x4 b = *reinterpret_cast<const f32x4*>(feat + nc * 256 + 16 * ks + 8 + 4 * h);
                bf16x8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = j < 4 ? a[j] : b[j - 4];
                    if constexpr (PREC == 3) {
                        __bf16 x, y;
                        split_bf16(v, x, y);
                        hi[j] = x;
                        lo[j] = y;
                    } else {
                        hi[j] = (__bf16)v;
                    }
                }
                *reinterpret_cast<bf16x8*>(dst + (ks * NPL) * kFragBytes) = hi;
                if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(dst + (ks * NPL + 1) * kFragBytes) = lo;
            }
        }
        if (wave < 2) {                                  // side inputs of tile ta + wave: pts | PE4(view) | normal -> k-steps 16..18
            const long tile = unit * 4 + ta + wave;
            const long n = tile * 32 + r;
            const long nc = n < N ? n : N - 1;
            float x[3], d[3], side[33], pe[27], jc[27];
            load_point(src, nc, x);
            if (dirs) {
#pragma unroll
                for (int c = 0; c < 3; ++c) d[c] = dirs[nc * 3 + c];
            } else {
                const long ray = nc / src.m;
#pragma unroll
                for (int c = 0; c < 3; ++c) d[c] = src.rays_d[ray * 3 + c];
            }
            posenc<4, false>(d, pe, jc);
#pragma unroll
            for (int c = 0; c < 3; ++c) side[c] = x[c];
#pragma unroll
            for (int f = 0; f < 27; ++f) side[3 + f] = pe[f];
#pragma unroll
            for (int c = 0; c < 3; ++c) side[30 + c] = normal[nc * 3 + c];
            BFrag<PREC> bs[kMaxKS];
            vec_to_bfrag<PREC, 33, 3, 16>(side, bs, h);
            frags_to_lds<PREC, 3>(lds_ + (ta + wave) * kP2Half, lane, 16, &bs[16]);
            if constexpr (TRAIN) {
                if (tile < tiles)
                    frags_to_plane<PREC, 3>(&bs[16], 0, st.side_hi + (size_t)tile * 4 * kFragBytes,
                                            (LO && st.side_lo) ? st.side_lo + (size_t)tile * 4 * kFragBytes : nullptr, pl, n < N);
            }
        }
    };
    float dot[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
    auto put_dot = [&]() {                               // the set's partial sums (both lane halves added) -> red
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float p = dot[k][c] + xor32(dot[k][c]);
                if (lane < 32) red[((k * 3 + c) * NW + wave) * 32 + lane] = p;
                dot[k][c] = 0.0f;
            }
    };
    auto finish = [&](long unit, int hb0) {              // waves 0, 1: rgb of tile hb0 + wave (fields.py:173-174 sigmoid)
        if (wave < 2 && lane < 32) {
            f32x16 b4[1];
            load_accvec<1, 0, 1>(blob, LY.L[4].bias, b4, lane);
            const long n = (unit * 4 + hb0 + wave) * 32 + r;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s = b4[0][c];
#pragma unroll
                for (int k = 0; k < NW; ++k) s += red[((wave * 3 + c) * NW + k) * 32 + lane];
                if (n < N) rgb_out[n * 3 + c] = 1.0f / (1.0f + expf(-s));
            }
        }
    };
    auto outputs = [&](long unit, int lV, int hbV) {     // where the vector work of a pass stores: layer lV of set hbV of `unit`
        P2St so;
        const long tile = unit * 4 + hbV;
        const bool ok = tile < tiles;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) so.vmask[hb] = (tile + hb) * 32 + r < N;
        if constexpr (TRAIN) {
            so.sig = p2_out_rsrc(reinterpret_cast<unsigned char*>(st.mask) + ((size_t)tile * 4 + lV) * 1024, ok ? 8192u : 0u);
            so.hi = p2_out_rsrc(st.u_hi + ((size_t)lV * tiles + tile) * kPPBlock, ok ? 2u * (unsigned)kPPBlock : 0u);
            // (gradient precision 2 -- side_lo NULL -- keeps the lo plane of slot 3 alone: the stores of the other slots fall outside an
            //  empty buffer and are dropped)
            if constexpr (LO) so.lo = p2_out_rsrc(st.u_lo + ((size_t)lV * tiles + tile) * kPPBlock, (ok && (st.side_lo || lV == 3)) ? 2u * (unsigned)kPPBlock : 0u);
        }
        return so;
    };
    f32x16 accA[TN][2], accB[TN][2], cw[3];
    auto load_cw = [&]() {                               // the 3 rows of the output layer in accumulator layout
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f32x16 one[1];
            load_accvec<8, 0, 1>(blob, LY.extra + c * 8 * 2 * 16 * 4, one, lane, t0);
            cw[c] = one[0];
        }
    };
    const __amdgpu_buffer_rsrc_t rsrc = p2_rsrc(blob);
    auto next_of = [&](int l) { return P2Next{LY.L[l].fwd_hi, LY.L[l].fwd_lo, LY.L[l].bias, 8}; };
    P2Prime<FNEUS_P2_DEPTH, TN> pr;
    p2_prime_all<PREC, FNEUS_P2_DEPTH, TN>(pr, blob, rsrc, lane, t0, next_of(0));
    if ((long)blockIdx.x < units) {
        encode(blockIdx.x, 0);
        encode(blockIdx.x, 2);
    }
    p2_barrier();
    bool first = true;
#define COL_PASS(KS, ACT, L_, NX, ACCM, HBM, ACCV, HBV, SO)                                                                  \
    p2_pass_relu<PREC, KS, 8, ACT, MODE>(blob, rsrc, LY.L[L_].fwd_hi, LY.L[L_].fwd_lo, pr, NX, lds_, lane, t0, ACCM, HBM, ACCV, HBV, \
                                         cw, dot, SO, pl.even, pl.odd, voff_mask)
    for (long unit = blockIdx.x; unit < units; unit += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const bool more = unit + gridDim.x < units;
        if (first) {
            const P2St none = outputs(unit, 0, 0);
            COL_PASS(19, 0, 0, next_of(0), accA, 0, accB, 2, none);
        } else {
            const P2St so = outputs(unit - gridDim.x, 3, 2);
            load_cw();
            COL_PASS(19, 8, 0, next_of(0), accA, 0, accB, 2, so);
        }
        p2_barrier();
        // `red` serves both sets: set B's sums are written only BEHIND this barrier (waves 0, 1 read set A's sums of the previous
        // unit up to it) and read behind the next one
        if (!first) put_dot();
        {
            const P2St so = outputs(unit, 0, 0);
            COL_PASS(19, 7, 0, next_of(1), accB, 2, accA, 0, so);
        }
        p2_barrier();
        if (!first) finish(unit - gridDim.x, 2);
        first = false;
#pragma unroll 1
        for (int l = 1; l <= 3; ++l) {
            asm volatile("" : "+s"(blob));
            const P2Next same = next_of(l), following = next_of(l == 3 ? 0 : l + 1);
            {
                const P2St so = outputs(unit, l - 1, 2);
                COL_PASS(16, 7, l, same, accA, 0, accB, 2, so);
            }
            p2_barrier();
            {
                const P2St so = outputs(unit, l, 0);
                if (l == 3) {
                    load_cw();
                    COL_PASS(16, 8, l, following, accB, 2, accA, 0, so);
                    put_dot();
                    if (more) encode(unit + gridDim.x, 0);          // set A's slots: last read by pass A of this layer
                } else {
                    COL_PASS(16, 7, l, following, accB, 2, accA, 0, so);
                }
            }
            p2_barrier();
        }
        finish(unit, 0);
        if (more) encode(unit + gridDim.x, 2);                      // set B's slots: last read by pass B of layer 3
        // (the barrier behind the next L0.A orders these writes before L0.B reads them; red is rewritten only behind it too)
    }
#undef COL_PASS
    if (!first) {       // tail of the last unit: ReLU 3 of set B -> plane, mask, dots
        long last = blockIdx.x;
        while (last + gridDim.x < units) last += gridDim.x;
        const P2St so = outputs(last, 3, 2);
        load_cw();
        uint32_t mbits = 0u;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                bf16x8 ph, plo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = fmaxf(accB[0][hb][8 * sh + e], 0.0f);
                    mbits |= (y > 0.0f ? 1u : 0u) << (8 * sh + e);
#pragma unroll
                    for (int c = 0; c < 3; ++c) dot[hb][c] = fmaf(y, cw[c][8 * sh + e], dot[hb][c]);
                    const float ys = so.vmask[hb] ? y : 0.0f;
                    __bf16 a, b2;
                    split_bf16(ys, a, b2);
                    ph[e] = a;
                    plo[e] = b2;
                }
                if constexpr (TRAIN) {
                    const int ks = 2 * t0 + sh;
                    const unsigned vo = (ks & 1) ? pl.odd : pl.even;
                    p2_store128<true>(__builtin_bit_cast(p2_u32x4, ph), so.hi, vo, hb * (int)kPPBlock + ks * kFragBytes);
                    if constexpr (LO) p2_store128<true>(__builtin_bit_cast(p2_u32x4, plo), so.lo, vo, hb * (int)kPPBlock + ks * kFragBytes);
                }
            }
            if constexpr (TRAIN) p2_store16(mbits, so.sig, voff_mask, hb * 4096);
            mbits = 0u;
        }
        p2_barrier();                   // waves 0, 1 have read set A's sums of this unit
        put_dot();
        p2_barrier();
        finish(last, 2);
    }
}

template <int PREC, int MODE>
static int launch_col_p2(const unsigned char* b, const PointSrc& src, long n_pts, const float* dirs, const float* normal,
                         const float* feat, const ColStash& st, float* rgb_out, hipStream_t stream) {
    static bool done = false;
    if (!done) {
        allow_big_lds(color_fwd_p2_kernel<PREC, MODE>);
        done = true;
    }
    const long units = (n_pts + 127) / 128;
    hipLaunchKernelGGL((color_fwd_p2_kernel<PREC, MODE>), dim3((unsigned)(units < 256 ? units : 256)), dim3(512), kColP2LdsTotal, stream,
                       b, src, n_pts, dirs, normal, feat, st, rgb_out);
    return launch_status();
}

int color_fwd_p2(const unsigned char* b, const PointSrc& src, long n_pts, const float* dirs, const float* normal, const float* feat,
                 const ColStash& st, float* rgb_out, int prec, int mode, hipStream_t stream) {
    if (prec == 3 && mode == 0) return launch_col_p2<3, 0>(b, src, n_pts, dirs, normal, feat, st, rgb_out, stream);
    if (prec == 3 && mode == 1) return launch_col_p2<3, 1>(b, src, n_pts, dirs, normal, feat, st, rgb_out, stream);
    if (prec == 3 && mode == 3) return launch_col_p2<3, 3>(b, src, n_pts, dirs, normal, feat, st, rgb_out, stream);
    if (prec == 1 && mode == 0) return launch_col_p2<1, 0>(b, src, n_pts, dirs, normal, feat, st, rgb_out, stream);
    if (prec == 1 && mode == 1) return launch_col_p2<1, 1>(b, src, n_pts, dirs, normal, feat, st, rgb_out, stream);
    return -2;
}

}  // namespace fneus
