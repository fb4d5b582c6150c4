// Colour-network kernels (reference models/fields.py:150-175 RenderingNetwork.forward, mode 'idr', and its autograd):
//   K4 color_fwd : [pts3 | PE4(view)27 | normal3 | feature256] -> 4 x (Linear 256 + ReLU) -> Linear 3 -> sigmoid
//   K4 color_bwd : descending chain; returns d feature, d normal and writes the weight-gradient GEMM operands
// Same wave-local MFMA engine as the SDF kernels.  K-slot order of layer 0: 256 feature slots, then the 33 "side"
// inputs in the reference's column order (pts, PE(view), normal).
//
// The two MLPs of the surface head RefColor (fields.py:271-335) have the same shape -- 256 features + <= 33 side
// inputs -> 4 x 256 ReLU -> <= 3 sigmoid outputs -- and run on the same kernels (template parameter VAR):
//   VAR 1  net_cd               : [pts3 | PE4(n)27 | feature]           -> diffuse rgb (3)
//   VAR 2  viewdir_mlp + net_cs : [n3 | pts3 | PE4(reflect(-d, n^))27 | feature] -> specular (1; rows 1,2 of the output
//          tile are padding with zero weights)
// They differ in how the side inputs are built and in how the side gradients map back to d normal.
// weight-prefetch depth of this file's kernels (stages; see dense() in mlp_engine.h): measured best at 4 / 8
#define FNEUS_PREFETCH_X3 4
#define FNEUS_PREFETCH_X1 8
#include <stdlib.h>
#include "pp_engine.h"
#include "fneus_kernels.h"

#ifndef FNEUS_COL_OCC
#define FNEUS_COL_OCC 2      // workgroups per CU the tensor-parallel kernels of this file are compiled for (experiments: 3)
#endif

#include "color_p2.h"
#include "color_r8.h"

#ifndef FNEUS_COL_P2_DEFAULT
#define FNEUS_COL_P2_DEFAULT 1
#endif

namespace fneus {

template <int TN>
FN_DEV void relu_inplace(f32x16 (&acc)[TN]) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = fmaxf(acc[t][r], 0.0f);
}

// value of feature `idx` (compile-time) of an accumulator-layout vector, valid in every lane of the sample pair
template <int TN, int IDX>
FN_DEV float acc_extract(const f32x16 (&acc)[TN], int h) {
    constexpr int t = IDX / 32, row = IDX % 32;
    constexpr int hh = (row >> 2) & 1;
    constexpr int reg = (row & 3) + 4 * (row >> 3);
    static_assert(acc_row(reg, hh) == row, "accumulator row mapping");
    const float v = (h == hh) ? acc[t][reg] : 0.0f;
    return v + xor32(v);
}

enum { VAR_COLOR = 0, VAR_REF_DIFFUSE = 1, VAR_REF_SPECULAR = 2 };

// n^ = n / max(|n|, sqrt(eps)) and the reflected view direction 2 (n^ . (-d)) n^ + d   (fields.py:277-283, 305-308)
FN_DEV void reflect_dir(const float (&d)[3], const float (&nrm)[3], float (&nh)[3], float (&ref)[3], float& inv_len,
                        float& s) {
    const float eps = 1.1920928955078125e-07f;
    const float len2 = fmaxf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2], eps);
    inv_len = 1.0f / sqrtf(len2);
    s = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        nh[c] = nrm[c] * inv_len;
        s -= d[c] * nh[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) ref[c] = 2.0f * s * nh[c] + d[c];
}

// the <= 33 inputs of layer 0 besides the feature vector, in the reference's column order
template <int VAR>
FN_DEV void make_side(const float (&x)[3], const float (&d)[3], const float (&nrm)[3], float (&side)[33]) {
    float pe[27], jc[27];
    if constexpr (VAR == VAR_COLOR) {             // pts | PE4(view) | normal
        posenc<4, false>(d, pe, jc);
#pragma unroll
        for (int c = 0; c < 3; ++c) side[c] = x[c];
#pragma unroll
        for (int f = 0; f < 27; ++f) side[3 + f] = pe[f];
#pragma unroll
        for (int c = 0; c < 3; ++c) side[30 + c] = nrm[c];
    } else if constexpr (VAR == VAR_REF_DIFFUSE) {  // pts | PE4(n)   (the raw normal is encoded, fields.py:306)
        posenc<4, false>(nrm, pe, jc);
#pragma unroll
        for (int c = 0; c < 3; ++c) side[c] = x[c];
#pragma unroll
        for (int f = 0; f < 27; ++f) side[3 + f] = pe[f];
#pragma unroll
        for (int c = 0; c < 3; ++c) side[30 + c] = 0.0f;
    } else {                                        // n | pts | PE4(reflected direction)
        float nh[3], ref[3], inv_len, sdot;
        reflect_dir(d, nrm, nh, ref, inv_len, sdot);
        posenc<4, false>(ref, pe, jc);
#pragma unroll
        for (int c = 0; c < 3; ++c) side[c] = nrm[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) side[3 + c] = x[c];
#pragma unroll
        for (int f = 0; f < 27; ++f) side[6 + f] = pe[f];
    }
}

// both RefColor heads in one launch (blockIdx.y = head): at 2 samples per ray a head is 32 workgroups, i.e. the launch
// is as long as ONE tile's chain whatever it contains; side by side the two heads cost one such latency instead of two
struct HeadArgs {
    const unsigned char* blob;
    ColStash st;
    float* out;            // forward: [N][3] output;  backward: d_feat [N][256]
    const float* d_out;    // backward only: cotangent of the head's output [N][3]
    const float* fwd_out;  // backward only: the head's forward output [N][3]
    float* d_normal;       // backward only: [N][3]
};

// gradient of the 33 side inputs (2 accumulator tiles) -> d normal of the sample, for the three input layouts
template <int VAR>
FN_DEV void side_grad_to_dnormal(const f32x16 (&s2)[2], long n, long nc, bool valid, int lane, int h,
                                 float* __restrict__ d_normal, const float* __restrict__ normal,
                                 const float* __restrict__ dirs, const float* __restrict__ rays_d, int m) {
    if constexpr (VAR == VAR_COLOR) {
        const float g0 = acc_extract<2, 30>(s2, h), g1 = acc_extract<2, 31>(s2, h), g2 = acc_extract<2, 32>(s2, h);
        if (valid && lane < 32) {
            d_normal[n * 3 + 0] = g0;
            d_normal[n * 3 + 1] = g1;
            d_normal[n * 3 + 2] = g2;
        }
    } else {
        float nrm[3], d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) nrm[c] = normal[nc * 3 + c];
        if (dirs) {
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = dirs[nc * 3 + c];
        } else {
            const long ray = nc / m;
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = rays_d[ray * 3 + c];
        }
        float pe[27], jc[27], dn[3];
        if constexpr (VAR == VAR_REF_DIFFUSE) {    // side = pts | PE4(n):  d n = J_PE(n)^T g[3..29]
            posenc<4, true>(nrm, pe, jc);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float coef[33];
#pragma unroll
                for (int f = 0; f < 33; ++f) coef[f] = (f >= 3 && f < 30 && ((f - 3) % 3) == c) ? jc[f - 3] : 0.0f;
                const float part = acc_dot_partial<2, 33>(s2, coef, h);
                dn[c] = part + xor32(part);
            }
        } else {                                   // side = n | pts | PE4(ref),  ref = 2 (n^ . -d) n^ + d
            float nh[3], ref[3], inv_len, sdot, dref[3], gdir[3];
            reflect_dir(d, nrm, nh, ref, inv_len, sdot);
            posenc<4, true>(ref, pe, jc);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float coef[33], one[33];
#pragma unroll
                for (int f = 0; f < 33; ++f) {
                    coef[f] = (f >= 6 && ((f - 6) % 3) == c) ? jc[f - 6] : 0.0f;
                    one[f] = (f == c) ? 1.0f : 0.0f;
                }
                const float p1 = acc_dot_partial<2, 33>(s2, coef, h);
                const float p2 = acc_dot_partial<2, 33>(s2, one, h);
                dref[c] = p1 + xor32(p1);
                gdir[c] = p2 + xor32(p2);
            }
            // d n^_j = 2 s dref_j - 2 d_j (dref . n^);  n^ = n / |n|  ->  d n = (d n^ - n^ (n^ . d n^)) / |n|
            const float dr_n = dref[0] * nh[0] + dref[1] * nh[1] + dref[2] * nh[2];
            float dnh[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) dnh[c] = 2.0f * sdot * dref[c] - 2.0f * d[c] * dr_n;
            const float proj = dnh[0] * nh[0] + dnh[1] * nh[1] + dnh[2] * nh[2];
            const float eps = 1.1920928955078125e-07f;
            const bool clamped = (nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]) < eps;   // torch.clamp: zero slope
#pragma unroll
            for (int c = 0; c < 3; ++c) dn[c] = gdir[c] + (clamped ? dnh[c] : dnh[c] - nh[c] * proj) * inv_len;
        }
        if (valid && lane < 32) {
#pragma unroll
            for (int c = 0; c < 3; ++c) d_normal[n * 3 + c] = dn[c];
        }
    }
}

// ---- tensor-parallel workgroups (tp_engine.h): 4 waves share a 32-sample tile, wave w owns output tiles 2w, 2w+1 ----------
// A wave needs < 256 registers, so two workgroups share a CU and one's barrier phases overlap the other's MFMAs; the
// 32-tile launches of the RefColor heads (latency-bound: a launch lasts as long as one tile's chain) get a chain that is
// four times shorter.  Everything the weight-gradient GEMM needs leaves as fragment planes (fneus_pp.h, pp_engine.h):
// the B fragments a wave publishes for the exchange are stored to global memory as they are.
constexpr int kColTpFrag = 19 * 2 * kFragBytes;          // layer 0 has 19 k-steps (hi, lo)
constexpr int kColTpLds = kColTpFrag;

FN_DEV void tp_barrier_pair() {      // what a wave without a tile to publish does while the others run tp_exchange
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int PREC, int KS>
FN_DEV void tp_write_frags(unsigned char* frag, int lane, int ks0, const BFrag<PREC> (&b)[kMaxKS], int src0) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        *reinterpret_cast<bf16x8*>(frag + ((ks0 + i) * NPL) * kFragBytes + lane * 16) = b[src0 + i].hi;
        if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(frag + ((ks0 + i) * NPL + 1) * kFragBytes + lane * 16) = b[src0 + i].lo;
    }
}

// ReLU in place on this wave's 2 tiles; returns the 32 sign bits (word `wave` of the tile's 128-bit mask)
FN_DEV uint32_t relu_mask2(f32x16 (&acc)[2]) {
    uint32_t m = 0u;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool pos = acc[t][r] > 0.0f;
            acc[t][r] = pos ? acc[t][r] : 0.0f;
            m |= (pos ? 1u : 0u) << (t * 16 + r);
        }
    return m;
}

template <int PREC, bool TRAIN, int VAR>
FN_DEV void color_fwd_tp_body(unsigned char* lds, const unsigned char* blob, const PointSrc& src, long N,
                              const float* __restrict__ dirs, const float* __restrict__ normal,
                              const float* __restrict__ feat, const ColStash& st, float* __restrict__ rgb_out) {
    unsigned char* frag = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    constexpr auto& LY = kColLayout;
    const long tiles = pp_tiles(N);
    // lo planes: all of them (gradient precision 3: side_lo given), or -- gradient precision 2, side_lo NULL -- slot 3 of u alone: the
    // operand of the output layer's weight gradient, the one product of this network whose bf16 rounding shows (include/fneus.h)
    const bool lo_planes = TRAIN && PREC == 3 && st.u_lo != nullptr;
    for (long tile = blockIdx.x; tile * 32 < N; tile += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const long n = tile * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        BFrag<PREC> bf[kMaxKS];
        if (wave == 0) {        // the 33 side inputs: k-steps 16..18 of layer 0, published by wave 0
            float side[33];
            float x[3], d[3], nrm[3];
            load_point(src, nc, x);
            if (dirs) {
#pragma unroll
                for (int c = 0; c < 3; ++c) d[c] = dirs[nc * 3 + c];
            } else {
                const long ray = nc / src.m;
#pragma unroll
                for (int c = 0; c < 3; ++c) d[c] = src.rays_d[ray * 3 + c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) nrm[c] = normal[nc * 3 + c];
            make_side<VAR>(x, d, nrm, side);
            vec_to_bfrag<PREC, 33, 3, 16>(side, bf, h);
            tp_write_frags<PREC, 3>(frag, lane, 16, bf, 16);
            if constexpr (TRAIN)    // side plane [tiles][4 fragments] (fragment 3 stays zero)
                frags_to_plane<PREC, 3>(&bf[16], 0, st.side_hi + (size_t)tile * 4 * kFragBytes,
                                        (lo_planes && st.side_lo) ? st.side_lo + (size_t)tile * 4 * kFragBytes : nullptr, pl, valid);
        }
        f32x16 acc[2];
        // the 256 features: every wave brings its two tiles (k-steps 4w .. 4w+3 of layer 0)
        load_f32<2>(acc, feat + 32 * t0, 256, nc, h);
        constexpr bool FEAT_PLANE = TRAIN && VAR != VAR_COLOR;     // the surface head keeps its own copy of the features
        tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, FEAT_PLANE ? st.feat_hi + (size_t)tile * kPPBlock : nullptr,
                                      (FEAT_PLANE && lo_planes && st.feat_lo) ? st.feat_lo + (size_t)tile * kPPBlock : nullptr, pl, valid);
        tp_operands<PREC, 19>(frag, lane, bf);
        // layer 0 (19 k-steps), layers 1..3
#pragma unroll 1
        for (int l = 0; l <= 3; ++l) {
            asm volatile("" : "+s"(blob));
            load_accvec<8, 0, 2>(blob, LY.L[l].bias, acc, lane, t0);
            if (l == 0)
                tp_dense<PREC, 19, 8, 0, 2>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, frag, bf, acc, lane, t0);
            else
                tp_dense<PREC, 16, 8, 0, 2>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, frag, bf, acc, lane, t0);
            const uint32_t m = relu_mask2(acc);
            if constexpr (TRAIN) reinterpret_cast<uint32_t*>(st.mask + ((size_t)tile * 4 + l) * 64 + lane)[wave] = m;
            tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, TRAIN ? st.u_hi + ((size_t)l * tiles + tile) * kPPBlock : nullptr,
                                          (lo_planes && (st.side_lo || l == 3)) ? st.u_lo + ((size_t)l * tiles + tile) * kPPBlock : nullptr, pl, valid);
            tp_operands<PREC, 16>(frag, lane, bf);
        }
        if (wave == 0) {        // output layer: one tile, rows 0..2
            f32x16 o[1];
            load_accvec<1, 0, 1>(blob, LY.L[4].bias, o, lane);
            tp_dense<PREC, 16, 1, 0, 1>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, frag, bf, o, lane);
            if (valid && lane < 32) {
#pragma unroll
                for (int c = 0; c < 3; ++c) rgb_out[n * 3 + c] = 1.0f / (1.0f + expf(-o[0][c]));   // fields.py:173-174
            }
        }
    }
}

template <int PREC, int VAR>
FN_DEV void color_bwd_tp_body(unsigned char* lds, const unsigned char* blob, long N, const float* __restrict__ d_rgb,
                              const float* __restrict__ rgb, const ColStash& st, float* __restrict__ d_feat,
                              float* __restrict__ d_normal, const float* __restrict__ normal,
                              const float* __restrict__ dirs, const float* __restrict__ rays_d, int m) {
    unsigned char* frag = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    constexpr auto& LY = kColLayout;
    const long tiles = pp_tiles(N);
    const bool lo_planes = PREC == 3 && st.zbar_lo != nullptr;
    for (long tile = blockIdx.x; tile * 32 < N; tile += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const long n = tile * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        BFrag<PREC> bf[kMaxKS];
        // zbar_4 = d rgb * sigmoid' (3 rows of one tile): wave 0 publishes k-steps 0, 1 and stores them as the zout plane
        if (wave == 0) {
            f32x16 z[1];
            zero_acc(z);
            if (h == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float y = rgb[nc * 3 + c];
                    z[0][c] = valid ? d_rgb[nc * 3 + c] * y * (1.0f - y) : 0.0f;
                }
            }
            tp_exchange_pp<PREC, 1, true>(frag, lane, 0, z, st.zout_hi + (size_t)tile * 2 * kFragBytes,
                                          (PREC == 3 && st.zout_lo) ? st.zout_lo + (size_t)tile * 2 * kFragBytes : nullptr, pl, valid);
        } else {
            tp_barrier_pair();
        }
        tp_operands<PREC, 2>(frag, lane, bf);
        f32x16 acc[2];
        zero_acc(acc);
        tp_dense<PREC, 2, 8, 0, 2>(blob, LY.L[4].rev_hi, LY.L[4].rev_lo, frag, bf, acc, lane, t0);
#pragma unroll 1
        for (int l = 3; l >= 0; --l) {
            asm volatile("" : "+s"(blob));
            const uint32_t msk = reinterpret_cast<const uint32_t*>(st.mask + ((size_t)tile * 4 + l) * 64 + lane)[wave];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const bool pos = (msk >> (t * 16 + rr)) & 1u;
                    acc[t][rr] = (pos && valid) ? acc[t][rr] : 0.0f;
                }
            tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, st.zbar_hi + ((size_t)l * tiles + tile) * kPPBlock,
                                          lo_planes ? st.zbar_lo + ((size_t)l * tiles + tile) * kPPBlock : nullptr, pl, valid);
            tp_operands<PREC, 16>(frag, lane, bf);
            if (l > 0) {
                zero_acc(acc);
                tp_dense<PREC, 16, 8, 0, 2>(blob, LY.L[l].rev_hi, LY.L[l].rev_lo, frag, bf, acc, lane, t0);
            }
        }
        // layer 0 reverse: 10 row tiles -- the 8 feature tiles (two per wave) and the 2 side tiles (wave 0)
        zero_acc(acc);
        tp_dense<PREC, 16, 10, 0, 2>(blob, LY.L[0].rev_hi, LY.L[0].rev_lo, frag, bf, acc, lane, t0);
        store_f32<2>(acc, d_feat + 32 * t0, 256, nc, h, valid);
        if (wave == 0) {
            f32x16 s2[2];
            zero_acc(s2);
            tp_dense<PREC, 16, 10, 8, 2>(blob, LY.L[0].rev_hi, LY.L[0].rev_lo, frag, bf, s2, lane);
            side_grad_to_dnormal<VAR>(s2, n, nc, valid, lane, h, d_normal, normal, dirs, rays_d, m);
        }
    }
}

// ---- the colour network on workgroups of HB sample halves (HB = 2: 64 samples share one pass over the weight fragments) ------
// The tensor-parallel kernels are bound by the L2 weight stream (every 32-sample tile streams the layer's fragments: ~52 TB/s
// wanted chip-wide at the full MFMA rate, ~18 delivered); dense_ldsb_h feeds two tiles from one pass.  Same maths, planes and
// masks as color_fwd_tp_body / color_bwd_tp_body<VAR_COLOR>; used for launches of >= 1024 tiles.
constexpr int kColHalf = kColTpFrag;

template <int PREC, bool TRAIN, int HB>
__global__ void __launch_bounds__(256, 2) color_fwd_tph_kernel(const unsigned char* blob, PointSrc src, long N,
                                                               const float* __restrict__ dirs, const float* __restrict__ normal,
                                                               const float* __restrict__ feat, ColStash st,
                                                               float* __restrict__ rgb_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    constexpr int HALF = kColHalf;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    constexpr auto& LY = kColLayout;
    const long tiles = pp_tiles(N);
    const long groups = (N + 32 * HB - 1) / (32 * HB);
    const bool lo_planes = TRAIN && PREC == 3 && st.u_lo != nullptr;
    for (long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        asm volatile("" : "+s"(blob));
        long tile[HB], n[HB], nc[HB];
        bool valid[HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            tile[hb] = grp * HB + hb;
            n[hb] = tile[hb] * 32 + r;
            valid[hb] = n[hb] < N;
            nc[hb] = valid[hb] ? n[hb] : N - 1;
        }
        if (wave < HB) {        // the 33 side inputs of half `wave`: k-steps 16..18 of layer 0 (not read since the previous group's layer 0)
            const int hb = wave;
            float side[33], x[3], d[3], nrm[3];
            load_point(src, nc[hb], x);
            if (dirs) {
#pragma unroll
                for (int c = 0; c < 3; ++c) d[c] = dirs[nc[hb] * 3 + c];
            } else {
                const long ray = nc[hb] / src.m;
#pragma unroll
                for (int c = 0; c < 3; ++c) d[c] = src.rays_d[ray * 3 + c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) nrm[c] = normal[nc[hb] * 3 + c];
            make_side<VAR_COLOR>(x, d, nrm, side);
            BFrag<PREC> bs[kMaxKS];
            vec_to_bfrag<PREC, 33, 3, 16>(side, bs, h);
            tp_write_frags<PREC, 3>(frag + hb * HALF, lane, 16, bs, 16);
            if constexpr (TRAIN)
                frags_to_plane<PREC, 3>(&bs[16], 0, st.side_hi + (size_t)tile[hb] * 4 * kFragBytes,
                                        (lo_planes && st.side_lo) ? st.side_lo + (size_t)tile[hb] * 4 * kFragBytes : nullptr, pl, valid[hb]);
        }
        f32x16 acc[2][HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            f32x16 a2[2];
            load_f32<2>(a2, feat + 32 * t0, 256, nc[hb], h);
            acc[0][hb] = a2[0];
            acc[1][hb] = a2[1];
        }
        unsigned char* none[HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) none[hb] = nullptr;
        tph_exchange<PREC, 2, true, HB, HALF>(frag, lane, t0, acc, none, none, pl, valid);
#pragma unroll 1
        for (int l = 0; l <= 3; ++l) {
            asm volatile("" : "+s"(blob));
            {
                f32x16 b2[2];
                load_accvec<8, 0, 2>(blob, LY.L[l].bias, b2, lane, t0);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) acc[t][hb] = b2[t];
            }
            if (l == 0)
                tph_dense<PREC, 19, 8, 0, 2, true, HB, HALF>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, frag, acc, lane, t0);
            else
                tph_dense<PREC, 16, 8, 0, 2, true, HB, HALF>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, frag, acc, lane, t0);
            unsigned char *uh[HB], *ul[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                uint32_t m = 0u;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const bool pos = acc[t][hb][e] > 0.0f;
                        acc[t][hb][e] = pos ? acc[t][hb][e] : 0.0f;
                        m |= (pos ? 1u : 0u) << (t * 16 + e);
                    }
                if constexpr (TRAIN) reinterpret_cast<uint32_t*>(st.mask + ((size_t)tile[hb] * 4 + l) * 64 + lane)[wave] = m;
                const size_t off = ((size_t)l * tiles + tile[hb]) * kPPBlock;
                uh[hb] = TRAIN ? st.u_hi + off : nullptr;
                ul[hb] = (lo_planes && (st.side_lo || l == 3)) ? st.u_lo + off : nullptr;
            }
            tph_exchange<PREC, 2, true, HB, HALF>(frag, lane, t0, acc, uh, ul, pl, valid);
        }
        if (wave == 0) {        // output layer: one tile per half, rows 0..2
            f32x16 o[1][HB], b1[1];
            load_accvec<1, 0, 1>(blob, LY.L[4].bias, b1, lane);
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) o[0][hb] = b1[0];
            tph_dense<PREC, 16, 1, 0, 1, true, HB, HALF>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, frag, o, lane);
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
                if (valid[hb] && lane < 32) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) rgb_out[n[hb] * 3 + c] = 1.0f / (1.0f + expf(-o[0][hb][c]));   // fields.py:173-174
                }
        }
    }
}

template <int PREC, int HB>
__global__ void __launch_bounds__(256, 2) color_bwd_tph_kernel(const unsigned char* blob, long N, const float* __restrict__ d_rgb,
                                                               const float* __restrict__ rgb, ColStash st,
                                                               float* __restrict__ d_feat, float* __restrict__ d_normal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    constexpr int HALF = kColHalf;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    constexpr auto& LY = kColLayout;
    const long tiles = pp_tiles(N);
    const long groups = (N + 32 * HB - 1) / (32 * HB);
    const bool lo_planes = PREC == 3 && st.zbar_lo != nullptr;
    for (long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        asm volatile("" : "+s"(blob));
        long tile[HB], n[HB], nc[HB];
        bool valid[HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            tile[hb] = grp * HB + hb;
            n[hb] = tile[hb] * 32 + r;
            valid[hb] = n[hb] < N;
            nc[hb] = valid[hb] ? n[hb] : N - 1;
        }
        // zbar_4 = d rgb * sigmoid' (3 rows of one tile per half): wave 0 publishes k-steps 0, 1 and stores the zout planes
        {
            f32x16 z[1][HB];
            unsigned char *zh[HB], *zl[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) z[0][hb][e] = 0.0f;
                if (wave == 0 && h == 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float y = rgb[nc[hb] * 3 + c];
                        z[0][hb][c] = valid[hb] ? d_rgb[nc[hb] * 3 + c] * y * (1.0f - y) : 0.0f;
                    }
                }
                zh[hb] = wave == 0 ? st.zout_hi + (size_t)tile[hb] * 2 * kFragBytes : nullptr;
                zl[hb] = (wave == 0 && PREC == 3 && st.zout_lo) ? st.zout_lo + (size_t)tile[hb] * 2 * kFragBytes : nullptr;
            }
            if (wave == 0) {
                tph_exchange<PREC, 1, true, HB, HALF>(frag, lane, 0, z, zh, zl, pl, valid);
            } else {
                tp_barrier_pair();
            }
        }
        f32x16 acc[2][HB];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][hb][e] = 0.0f;
        tph_dense<PREC, 2, 8, 0, 2, true, HB, HALF>(blob, LY.L[4].rev_hi, LY.L[4].rev_lo, frag, acc, lane, t0);
#pragma unroll 1
        for (int l = 3; l >= 0; --l) {
            asm volatile("" : "+s"(blob));
            unsigned char *zh[HB], *zl[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const uint32_t msk = reinterpret_cast<const uint32_t*>(st.mask + ((size_t)tile[hb] * 4 + l) * 64 + lane)[wave];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        acc[t][hb][e] = (((msk >> (t * 16 + e)) & 1u) && valid[hb]) ? acc[t][hb][e] : 0.0f;
                const size_t off = ((size_t)l * tiles + tile[hb]) * kPPBlock;
                zh[hb] = st.zbar_hi + off;
                zl[hb] = lo_planes ? st.zbar_lo + off : nullptr;
            }
            tph_exchange<PREC, 2, true, HB, HALF>(frag, lane, t0, acc, zh, zl, pl, valid);
            if (l > 0) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[t][hb][e] = 0.0f;
                tph_dense<PREC, 16, 8, 0, 2, true, HB, HALF>(blob, LY.L[l].rev_hi, LY.L[l].rev_lo, frag, acc, lane, t0);
            }
        }
        // layer 0 reverse: 10 row tiles -- the 8 feature tiles (two per wave) and the 2 side tiles (wave 0)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][hb][e] = 0.0f;
        tph_dense<PREC, 16, 10, 0, 2, true, HB, HALF>(blob, LY.L[0].rev_hi, LY.L[0].rev_lo, frag, acc, lane, t0);
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            f32x16 a2[2] = {acc[0][hb], acc[1][hb]};
            store_f32<2>(a2, d_feat + 32 * t0, 256, nc[hb], h, valid[hb]);
        }
        if (wave == 0) {
            f32x16 s2[2][HB];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) s2[t][hb][e] = 0.0f;
            tph_dense<PREC, 16, 10, 8, 2, true, HB, HALF>(blob, LY.L[0].rev_hi, LY.L[0].rev_lo, frag, s2, lane);
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const f32x16 one[2] = {s2[0][hb], s2[1][hb]};
                side_grad_to_dnormal<VAR_COLOR>(one, n[hb], nc[hb], valid[hb], lane, h, d_normal, nullptr, nullptr, nullptr, 1);
            }
        }
    }
}

template <int PREC, bool TRAIN, int VAR>
__global__ void __launch_bounds__(256, FNEUS_COL_OCC) color_fwd_tp_kernel(const unsigned char* blob, PointSrc src, long N,
                                                              const float* __restrict__ dirs, const float* __restrict__ normal,
                                                              const float* __restrict__ feat, ColStash st,
                                                              float* __restrict__ rgb_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    color_fwd_tp_body<PREC, TRAIN, VAR>(lds_, blob, src, N, dirs, normal, feat, st, rgb_out);
}

template <int PREC, bool TRAIN>
__global__ void __launch_bounds__(256, FNEUS_COL_OCC) refcolor_fwd_both_tp_kernel(HeadArgs cd, HeadArgs vd, PointSrc src, long N,
                                                                      const float* __restrict__ dirs,
                                                                      const float* __restrict__ normal,
                                                                      const float* __restrict__ feat) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    if (blockIdx.y == 0)
        color_fwd_tp_body<PREC, TRAIN, VAR_REF_DIFFUSE>(lds_, cd.blob, src, N, dirs, normal, feat, cd.st, cd.out);
    else
        color_fwd_tp_body<PREC, TRAIN, VAR_REF_SPECULAR>(lds_, vd.blob, src, N, dirs, normal, feat, vd.st, vd.out);
}

template <int PREC, int VAR>
__global__ void __launch_bounds__(256, FNEUS_COL_OCC) color_bwd_tp_kernel(const unsigned char* blob, long N, const float* __restrict__ d_rgb,
                                                              const float* __restrict__ rgb, ColStash st,
                                                              float* __restrict__ d_feat, float* __restrict__ d_normal,
                                                              const float* __restrict__ normal, const float* __restrict__ dirs,
                                                              const float* __restrict__ rays_d, int m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    color_bwd_tp_body<PREC, VAR>(lds_, blob, N, d_rgb, rgb, st, d_feat, d_normal, normal, dirs, rays_d, m);
}

template <int PREC>
__global__ void __launch_bounds__(256, FNEUS_COL_OCC) refcolor_bwd_both_tp_kernel(HeadArgs cd, HeadArgs vd, long N,
                                                                      const float* __restrict__ normal,
                                                                      const float* __restrict__ dirs,
                                                                      const float* __restrict__ rays_d, int m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    if (blockIdx.y == 0)
        color_bwd_tp_body<PREC, VAR_REF_DIFFUSE>(lds_, cd.blob, N, cd.d_out, cd.fwd_out, cd.st, cd.out, cd.d_normal, normal, dirs, rays_d, m);
    else
        color_bwd_tp_body<PREC, VAR_REF_SPECULAR>(lds_, vd.blob, N, vd.d_out, vd.fwd_out, vd.st, vd.out, vd.d_normal, normal, dirs, rays_d, m);
}

}  // namespace fneus

using namespace fneus;

static inline unsigned tp_grid(long n_tiles) {
    const long cap = 256 * 2 * 4;
    return (unsigned)(n_tiles < 1 ? 1 : (n_tiles < cap ? n_tiles : cap));
}
#define FNEUS_TP_LAUNCH(KERNEL, GRID, ...)                                                                            \
    do {                                                                                                              \
        static bool attr_done = false;                                                                                \
        if (!attr_done) {                                                                                             \
            fneus::allow_big_lds(KERNEL);                                                                             \
            attr_done = true;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((KERNEL), GRID, dim3(256), fneus::kColTpLds, stream, __VA_ARGS__);                         \
    } while (0)

#define FNEUS_TPH_LAUNCH(KERNEL, GRID, ...)                                                                           \
    do {                                                                                                              \
        static bool attr_done = false;                                                                                \
        if (!attr_done) {                                                                                             \
            fneus::allow_big_lds(KERNEL);                                                                             \
            attr_done = true;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((KERNEL), GRID, dim3(256), 2 * fneus::kColHalf, stream, __VA_ARGS__);                      \
    } while (0)

// 64-sample workgroups for launches that fill the chip (>= 1024 tiles); FNEUS_COL_HB=1 keeps the 32-sample kernels
static inline bool col_use_hb2(long n_tiles) {
    const char* e = getenv("FNEUS_COL_HB");
    if (e && e[0] == '1') return false;
    return n_tiles >= 1024;
}

template <int VAR>
static int launch_fwd(const void* blob, const float* pts, const float* rays_o, const float* rays_d, const float* t, int m,
                      long n_pts, const float* dirs, const float* normal, const float* feat, const FneusColStash* stash,
                      float* out, int prec, int train, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    PointSrc src{pts, rays_o, rays_d, t, m > 0 ? m : 1};
    if (!pts && !rays_d) return -2;
    if (pts && !dirs && !rays_d) return -2;
    if (train && !stash) return -2;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(blob);
    ColStash st = stash ? ColStash(*stash) : ColStash();
    if (train && VAR != VAR_COLOR && !st.feat_hi) return -2;
    if (VAR == VAR_COLOR && (n_pts + 31) / 32 >= 1024 && (prec == 3 || prec == 1)) {
        // chip-filling launches: the two-pass pipelined kernel (color_p2_kernels.hip); FNEUS_COL_P2=0 keeps the 4-wave kernels
        const char* p2_env = getenv("FNEUS_COL_P2");
        if ((p2_env ? atoi(p2_env) : FNEUS_COL_P2_DEFAULT) != 0) {
            // feat NULL (round 6): the features come as the planes stash.feat_hi / feat_lo (the SDF kernel's: fneus_sdf_fwd_grad)
            if (feat == nullptr && !(st.feat_hi && (prec == 1 || st.feat_lo))) return -2;
            const int mode = !train ? 0 : ((st.u_lo != nullptr && prec == 3) ? 3 : 1);
            return fneus::color_fwd_p2(b, src, n_pts, dirs, normal, feat, st, out, prec, mode, stream);
        }
    }
    if (feat == nullptr) {
        fneus::set_last_error("fneus_color_fwd: feat may be NULL only for launches of >= 1024 tiles on the two-pass kernel with feature planes in the stash");
        return -2;
    }
    if (VAR == VAR_COLOR && col_use_hb2((n_pts + 31) / 32)) {
        dim3 g2(tp_grid((n_pts + 63) / 64));
        if (prec == 3 && train) FNEUS_TPH_LAUNCH((color_fwd_tph_kernel<3, true, 2>), g2, b, src, n_pts, dirs, normal, feat, st, out);
        else if (prec == 3) FNEUS_TPH_LAUNCH((color_fwd_tph_kernel<3, false, 2>), g2, b, src, n_pts, dirs, normal, feat, st, out);
        else if (prec == 1 && train) FNEUS_TPH_LAUNCH((color_fwd_tph_kernel<1, true, 2>), g2, b, src, n_pts, dirs, normal, feat, st, out);
        else if (prec == 1) FNEUS_TPH_LAUNCH((color_fwd_tph_kernel<1, false, 2>), g2, b, src, n_pts, dirs, normal, feat, st, out);
        else return -2;
        return fneus::launch_status();
    }
    {
        dim3 g2(tp_grid((n_pts + 31) / 32));
        if (prec == 3 && train) FNEUS_TP_LAUNCH((color_fwd_tp_kernel<3, true, VAR>), g2, b, src, n_pts, dirs, normal, feat, st, out);
        else if (prec == 3) FNEUS_TP_LAUNCH((color_fwd_tp_kernel<3, false, VAR>), g2, b, src, n_pts, dirs, normal, feat, st, out);
        else if (prec == 1 && train) FNEUS_TP_LAUNCH((color_fwd_tp_kernel<1, true, VAR>), g2, b, src, n_pts, dirs, normal, feat, st, out);
        else if (prec == 1) FNEUS_TP_LAUNCH((color_fwd_tp_kernel<1, false, VAR>), g2, b, src, n_pts, dirs, normal, feat, st, out);
        else return -2;
        return fneus::launch_status();
    }
}

template <int VAR>
static int launch_bwd(const void* blob, long n_pts, const float* d_out, const float* out, const FneusColStash* stash,
                      float* d_feat, float* d_normal, const float* normal, const float* dirs, const float* rays_d, int m,
                      int prec, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    if (!stash) return -2;
    if (VAR != VAR_COLOR && (!normal || (!dirs && !rays_d))) return -2;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(blob);
    ColStash st(*stash);
    if (VAR == VAR_COLOR && (n_pts + 31) / 32 >= 1024 && (prec == 3 || prec == 1)) {
        // chip-filling launches: resident-weight 8-wave workgroups (color_r8_kernels.hip, round 6); FNEUS_COL_BWD_R8=0 keeps the
        // 4-wave kernels below, which also take launches whose planes exceed the r8 kernel's 32-bit buffer offsets
        const char* r8_env = getenv("FNEUS_COL_BWD_R8");
        const long tiles_pp = 2 * ((n_pts + 63) / 64);
        if ((r8_env ? atoi(r8_env) : 1) != 0 && tiles_pp * 4 * (long)kPPBlock < (1L << 31))
            return fneus::color_bwd_r8(b, n_pts, d_out, out, st, d_feat, d_normal, prec, stream);
    }
    if (d_feat == nullptr || st.dnormal_add) return -2;       // fragments out (stash->dfeat_hi) / d_normal added into: the resident-weight kernel only
    if (VAR == VAR_COLOR && col_use_hb2((n_pts + 31) / 32)) {
        dim3 g2(tp_grid((n_pts + 63) / 64));
        if (prec == 3) FNEUS_TPH_LAUNCH((color_bwd_tph_kernel<3, 2>), g2, b, n_pts, d_out, out, st, d_feat, d_normal);
        else if (prec == 1) FNEUS_TPH_LAUNCH((color_bwd_tph_kernel<1, 2>), g2, b, n_pts, d_out, out, st, d_feat, d_normal);
        else return -2;
        return fneus::launch_status();
    }
    {
        dim3 g2(tp_grid((n_pts + 31) / 32));
        const int mm = m > 0 ? m : 1;
        if (prec == 3) FNEUS_TP_LAUNCH((color_bwd_tp_kernel<3, VAR>), g2, b, n_pts, d_out, out, st, d_feat, d_normal, normal, dirs, rays_d, mm);
        else if (prec == 1) FNEUS_TP_LAUNCH((color_bwd_tp_kernel<1, VAR>), g2, b, n_pts, d_out, out, st, d_feat, d_normal, normal, dirs, rays_d, mm);
        else return -2;
        return fneus::launch_status();
    }
}

extern "C" int fneus_color_fwd(const void* blob, const float* pts, const float* rays_o, const float* rays_d,
                               const float* t, int m, long n_pts, const float* dirs, const float* normal,
                               const float* feat, const FneusColStash* stash, float* rgb_out, int prec, int train,
                               fneus_stream_t stream) {
    return launch_fwd<VAR_COLOR>(blob, pts, rays_o, rays_d, t, m, n_pts, dirs, normal, feat, stash, rgb_out, prec, train, stream);
}

extern "C" int fneus_color_bwd(const void* blob, long n_pts, const float* d_rgb, const float* rgb,
                               const FneusColStash* stash, float* d_feat, float* d_normal, int prec,
                               fneus_stream_t stream) {
    return launch_bwd<VAR_COLOR>(blob, n_pts, d_rgb, rgb, stash, d_feat, d_normal, nullptr, nullptr, nullptr, 1, prec, stream);
}

extern "C" int fneus_refcolor_fwd(const void* blob, int head, const float* pts, const float* rays_o, const float* rays_d,
                                  const float* t, int m, long n_pts, const float* dirs, const float* normal,
                                  const float* feat, const FneusColStash* stash, float* out, int prec, int train,
                                  fneus_stream_t stream) {
    if (head == VAR_REF_DIFFUSE)
        return launch_fwd<VAR_REF_DIFFUSE>(blob, pts, rays_o, rays_d, t, m, n_pts, dirs, normal, feat, stash, out, prec, train, stream);
    if (head == VAR_REF_SPECULAR)
        return launch_fwd<VAR_REF_SPECULAR>(blob, pts, rays_o, rays_d, t, m, n_pts, dirs, normal, feat, stash, out, prec, train, stream);
    return -2;
}

extern "C" int fneus_refcolor_bwd(const void* blob, int head, long n_pts, const float* rays_d, int m, const float* dirs,
                                  const float* normal, const float* d_out, const float* out, const FneusColStash* stash,
                                  float* d_feat, float* d_normal, int prec, fneus_stream_t stream) {
    if (head == VAR_REF_DIFFUSE)
        return launch_bwd<VAR_REF_DIFFUSE>(blob, n_pts, d_out, out, stash, d_feat, d_normal, normal, dirs, rays_d, m, prec, stream);
    if (head == VAR_REF_SPECULAR)
        return launch_bwd<VAR_REF_SPECULAR>(blob, n_pts, d_out, out, stash, d_feat, d_normal, normal, dirs, rays_d, m, prec, stream);
    return -2;
}

extern "C" int fneus_refcolor_fwd_both(const void* blob_cd, const void* blob_vd, const float* pts, const float* rays_o,
                                       const float* rays_d, const float* t, int m, long n_pts, const float* dirs,
                                       const float* normal, const float* feat, const FneusColStash* stash_cd,
                                       const FneusColStash* stash_vd, float* diffuse_out, float* spec_out, int prec,
                                       int train, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    PointSrc src{pts, rays_o, rays_d, t, m > 0 ? m : 1};
    if (!pts && !rays_d) return -2;
    if (pts && !dirs && !rays_d) return -2;
    if (train && (!stash_cd || !stash_vd)) return -2;
    HeadArgs cd{reinterpret_cast<const unsigned char*>(blob_cd), stash_cd ? ColStash(*stash_cd) : ColStash(), diffuse_out,
                nullptr, nullptr, nullptr};
    HeadArgs vd{reinterpret_cast<const unsigned char*>(blob_vd), stash_vd ? ColStash(*stash_vd) : ColStash(), spec_out,
                nullptr, nullptr, nullptr};
    if (train && (!cd.st.feat_hi || !vd.st.feat_hi)) return -2;
    {
        dim3 g2(tp_grid((n_pts + 31) / 32), 2);
        if (prec == 3 && train) FNEUS_TP_LAUNCH((refcolor_fwd_both_tp_kernel<3, true>), g2, cd, vd, src, n_pts, dirs, normal, feat);
        else if (prec == 3) FNEUS_TP_LAUNCH((refcolor_fwd_both_tp_kernel<3, false>), g2, cd, vd, src, n_pts, dirs, normal, feat);
        else if (prec == 1 && train) FNEUS_TP_LAUNCH((refcolor_fwd_both_tp_kernel<1, true>), g2, cd, vd, src, n_pts, dirs, normal, feat);
        else if (prec == 1) FNEUS_TP_LAUNCH((refcolor_fwd_both_tp_kernel<1, false>), g2, cd, vd, src, n_pts, dirs, normal, feat);
        else return -2;
        return fneus::launch_status();
    }
}

extern "C" int fneus_refcolor_bwd_both(const void* blob_cd, const void* blob_vd, long n_pts, const float* rays_d, int m,
                                       const float* dirs, const float* normal, const float* d_diffuse, const float* d_spec,
                                       const float* diffuse, const float* spec, const FneusColStash* stash_cd,
                                       const FneusColStash* stash_vd, float* d_feat2, float* d_normal2, int prec,
                                       fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    if (!stash_cd || !stash_vd || !normal || (!dirs && !rays_d)) return -2;
    HeadArgs cd{reinterpret_cast<const unsigned char*>(blob_cd), ColStash(*stash_cd), d_feat2, d_diffuse, diffuse, d_normal2};
    HeadArgs vd{reinterpret_cast<const unsigned char*>(blob_vd), ColStash(*stash_vd), d_feat2 + n_pts * 256, d_spec, spec,
                d_normal2 + n_pts * 3};
    {
        dim3 g2(tp_grid((n_pts + 31) / 32), 2);
        const int mm = m > 0 ? m : 1;
        if (prec == 3) FNEUS_TP_LAUNCH((refcolor_bwd_both_tp_kernel<3>), g2, cd, vd, n_pts, normal, dirs, rays_d, mm);
        else if (prec == 1) FNEUS_TP_LAUNCH((refcolor_bwd_both_tp_kernel<1>), g2, cd, vd, n_pts, normal, dirs, rays_d, mm);
        else return -2;
        return fneus::launch_status();
    }
}
