// SDF-network kernels (reference models/fields.py:74-111 SDFNetwork.forward/.sdf/.gradient and their autograd).
//   K1 sdf_fwd        : PE -> 9 dense layers (Softplus beta=100) -> sdf                      (no-grad sampler path)
//   K2 sdf_fwd_grad   : sdf, feature[256], normal = d sdf/dx (analytic reverse sweep) + bf16 stash for backward
//   K3 sdf_bwd_chain  : double-backward chains (ascending + descending); writes the
//                       operand matrices of the weight-gradient GEMM (dw_gemm.hip)
// One wavefront = 32 samples, whole chain register resident; weights come pre-packed from pack.hip (L2 resident).
#include <stdlib.h>
#include "mlp_engine.h"
#include "tp_engine.h"
#include "fneus_kernels.h"

namespace fneus {

// softplus in place; sigma'(z) goes to the lane-private stash block `ps` of this (tile, layer)
template <int PREC, int TN>
FN_DEV void softplus_ps(f32x16 (&acc)[TN], unsigned char* __restrict__ ps, int lane) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float sv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float hh;
                softplus_sig(acc[t][4 * g + e], hh, sv[e]);
                acc[t][4 * g + e] = hh;
            }
            sig_put(ps, t * 4 + g, lane, sv);
        }
}

template <int TN>
FN_DEV void softplus_inplace(f32x16 (&acc)[TN]) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = softplus100(acc[t][r]);
}

// ---- forward chain shared by K1/K2 -----------------------------------------------------------------------
// On return acc9 holds z_8: tiles 0..7 = feature (natural order), tile 8 row 0 (reg 0 of lane half 0) = sdf.
// If SDF_ONLY only tile 8 is computed (acc9[8]).  If STASH, H_{l+1} (l = 0..7) and PE are written.
// prefetch depth of the kernels that carry stash traffic (K2, K3)
template <int PREC> constexpr int kDeep = PREC == 3 ? 4 : 8;

template <int PREC, bool SDF_ONLY, bool STASH>
FN_DEV void sdf_forward_chain(const unsigned char* __restrict__ blob, const float (&pe)[39],
                              BFrag<PREC> (&bf)[kMaxKS], f32x16 (&acc)[9], const SdfStash& st, long N, long n,
                              int lane, bool valid, unsigned char* scr, long tile) {
    const int h = lane >> 5;
    const long n0 = tile * 32;
    unsigned char* psb = STASH ? st.ps + (size_t)tile * 8 * kSigBlockBytes : nullptr;
    constexpr auto& LY = kSdfLayout;
    constexpr int DD = STASH ? kDeep<PREC> : 0;
    BFrag<PREC> pef[3];
    vec_to_bfrag<PREC, 39, 3, 0>(pe, bf, h);
#pragma unroll
    for (int i = 0; i < 3; ++i) pef[i] = bf[i];
    if constexpr (STASH) {
        if (valid) {   // PE rows [N][48]; lane half h writes k-step features phi(ks,h,*) -> 8-byte pieces
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int col = 16 * ks + 8 * g + 4 * h;
                    bf16x4 vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vh[e] = bf[ks].hi[4 * g + e];
                        if constexpr (PREC == 3) vl[e] = bf[ks].lo[4 * g + e];
                    }
                    *reinterpret_cast<bf16x4*>(st.pe_hi + n * 48 + col) = vh;
                    if constexpr (PREC == 3) *reinterpret_cast<bf16x4*>(st.pe_lo + n * 48 + col) = vl;
                }
        }
    }
    f32x16(&a8)[8] = reinterpret_cast<f32x16(&)[8]>(acc);
    // layer 0
    load_accvec<8, 0, 8>(blob, LY.L[0].bias, a8, lane);
    dense<PREC, 3, 8, 0, 8, 0, DD>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, bf, a8, lane);
    if constexpr (STASH) {
        softplus_ps<PREC, 8>(a8, psb, lane);
        store_stash<PREC, 8>(scr, lane, a8, st.h_hi, st.h_lo, 256, n0, N, 256);
    } else {
        softplus_inplace(a8);
    }
    acc_to_bfrag<PREC, 8>(a8, bf);
    // layers 1, 2
    for (int l = 1; l <= 2; ++l) {
        load_accvec<8, 0, 8>(blob, LY.L[l].bias, a8, lane);
        dense<PREC, 16, 8, 0, 8, 0, DD>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, a8, lane);
        if constexpr (STASH) {
            softplus_ps<PREC, 8>(a8, psb + (size_t)l * kSigBlockBytes, lane);
            store_stash<PREC, 8>(scr, lane, a8, st.h_hi + (size_t)l * N * 256, st.h_lo + (size_t)l * N * 256, 256, n0, N, 256);
        } else {
            softplus_inplace(a8);
        }
        acc_to_bfrag<PREC, 8>(a8, bf);
    }
    // layer 3: 256 -> 217 (7 tiles); its output + PE is the input of layer 4 (skip connection, fields.py:83-84)
    {
        f32x16(&a7)[7] = reinterpret_cast<f32x16(&)[7]>(acc);
        load_accvec<7, 0, 7>(blob, LY.L[3].bias, a7, lane);
        dense<PREC, 16, 7, 0, 7, 0, DD>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, bf, a7, lane);
        if constexpr (STASH) {
            softplus_ps<PREC, 7>(a7, psb + (size_t)3 * kSigBlockBytes, lane);
            store_stash<PREC, 7>(scr, lane, a7, st.h_hi + (size_t)3 * N * 256, st.h_lo + (size_t)3 * N * 256, 256, n0, N, 224);
        } else {
            softplus_inplace(a7);
        }
        acc_to_bfrag<PREC, 7>(a7, bf);
#pragma unroll
        for (int i = 0; i < 3; ++i) bf[14 + i] = pef[i];
    }
    // layer 4 (17 k-steps; 1/sqrt2 folded into the pack)
    load_accvec<8, 0, 8>(blob, LY.L[4].bias, a8, lane);
    dense<PREC, 17, 8, 0, 8, 0, DD>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, bf, a8, lane);
    if constexpr (STASH) {
        softplus_ps<PREC, 8>(a8, psb + (size_t)4 * kSigBlockBytes, lane);
        store_stash<PREC, 8>(scr, lane, a8, st.h_hi + (size_t)4 * N * 256, st.h_lo + (size_t)4 * N * 256, 256, n0, N, 256);
    } else {
        softplus_inplace(a8);
    }
    acc_to_bfrag<PREC, 8>(a8, bf);
    // layers 5, 6, 7
    for (int l = 5; l <= 7; ++l) {
        load_accvec<8, 0, 8>(blob, LY.L[l].bias, a8, lane);
        dense<PREC, 16, 8, 0, 8, 0, DD>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, a8, lane);
        if constexpr (STASH) {
            softplus_ps<PREC, 8>(a8, psb + (size_t)l * kSigBlockBytes, lane);
            store_stash<PREC, 8>(scr, lane, a8, st.h_hi + (size_t)l * N * 256, st.h_lo + (size_t)l * N * 256, 256, n0, N, 256);
        } else {
            softplus_inplace(a8);
        }
        acc_to_bfrag<PREC, 8>(a8, bf);
    }
    // layer 8 (linear)
    if constexpr (SDF_ONLY) {
        f32x16(&a1)[1] = reinterpret_cast<f32x16(&)[1]>(acc[8]);
        load_accvec<9, 8, 1>(blob, LY.L[8].bias, a1, lane);
        dense<PREC, 16, 9, 8, 1, 0, DD>(blob, LY.L[8].fwd_hi, LY.L[8].fwd_lo, bf, a1, lane);
    } else {
        load_accvec<9, 0, 9>(blob, LY.L[8].bias, acc, lane);
        dense<PREC, 16, 9, 0, 9, 0, DD>(blob, LY.L[8].fwd_hi, LY.L[8].fwd_lo, bf, acc, lane);
    }
}

// ---- K1 ----------------------------------------------------------------------------------------------------
template <int PREC>
__global__ void __launch_bounds__(64, 1) sdf_fwd_kernel(const unsigned char* blob, PointSrc src, long N,
                                                        float* __restrict__ sdf_out) {
    const int lane = threadIdx.x;
    const int r = lane & 31;
    SdfStash st{};
    for (long tile = blockIdx.x; tile * 32 < N; tile += gridDim.x) {
        // launder the blob pointer: otherwise LICM hoists every statically addressed weight load out of the tile loop
        // (hundreds of VGPRs -> scratch spills)
        asm volatile("" : "+s"(blob));
        const long n = tile * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        const long n0 = tile * 32;
        float x[3], pe[39], jc[39];
        load_point(src, nc, x);
        posenc<6, false>(x, pe, jc);
        BFrag<PREC> bf[kMaxKS];
        f32x16 acc[9];
        sdf_forward_chain<PREC, true, false>(blob, pe, bf, acc, st, N, nc, lane, valid, nullptr, tile);
        if (valid && lane < 32) sdf_out[n] = acc[8][0];
    }
}

// NOTE on the LDS pointers of the tensor-parallel helpers below: no __restrict__ -- fragments written through one helper
// are read through another.
// ---- K1, small launches -------------------------------------------------------------------------------------
// The hierarchical sampler evaluates only the 16 NEW depths of every ray three times per step (renderer.py:430): 8192
// points = 256 tiles, a quarter of the wave slots, so such a launch lasts as long as ONE wave needs for a tile (72 us:
// 42 us of back-to-back MFMAs + activation work, all serial).  Here the 4 wavefronts of a workgroup share one
// 32-sample tile instead: wave w computes output tiles 2w, 2w+1 of every layer (a quarter of the MFMAs, of the
// activation work and of the weight stream), the activated tiles are exchanged through LDS as ready-made B fragments
// (k-step 2t+s of the next layer = half s of tile t), two LDS-only barriers per layer.
constexpr int kTpLds = 17 * 2 * kFragBytes;      // up to 17 k-steps x (hi, lo) fragments
#ifndef FNEUS_TP_WAVES
#define FNEUS_TP_WAVES 1
#define FNEUS_TP_MAX_TILES 256
#endif

// WAVES = 4: wave w owns output tiles 2w, 2w+1;  WAVES = 8: wave w owns tile w (half the MFMAs and the activation work
// per wave again, the same two barriers per layer)
template <int PREC, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 1) sdf_fwd_tp_kernel(const unsigned char* blob, PointSrc src, long N,
                                                                    float* __restrict__ sdf_out) {
    __shared__ __attribute__((aligned(16))) unsigned char frag[kTpLds];
    constexpr int TN = 8 / WAVES;                    // output tiles per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    constexpr auto& LY = kSdfLayout;
    for (long tile = blockIdx.x; tile * 32 < N; tile += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const long n = tile * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        float x[3], pe[39], jc[39];
        load_point(src, nc, x);
        posenc<6, false>(x, pe, jc);
        BFrag<PREC> bf[kMaxKS], pef[3];
        vec_to_bfrag<PREC, 39, 3, 0>(pe, bf, h);       // every wave encodes the (same) 32 points itself
#pragma unroll
        for (int i = 0; i < 3; ++i) pef[i] = bf[i];
        f32x16 acc[TN];
        f32x16(&a1)[1] = reinterpret_cast<f32x16(&)[1]>(acc);
        const int t0 = TN * wave;
        // layer 0
        load_accvec<8, 0, TN>(blob, LY.L[0].bias, acc, lane, t0);
        dense<PREC, 3, 8, 0, TN>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, bf, acc, lane, t0);
        softplus_inplace(acc);
        tp_publish<PREC, TN>(frag, lane, t0, acc);
        tp_gather<PREC, 16>(frag, lane, bf);
        for (int l = 1; l <= 2; ++l) {
            load_accvec<8, 0, TN>(blob, LY.L[l].bias, acc, lane, t0);
            dense<PREC, 16, 8, 0, TN>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, acc, lane, t0);
            softplus_inplace(acc);
            tp_publish<PREC, TN>(frag, lane, t0, acc);
            tp_gather<PREC, 16>(frag, lane, bf);
        }
        // layer 3: 7 output tiles (217 features): the last tile slot is empty
        if (t0 + TN <= 7) {
            load_accvec<7, 0, TN>(blob, LY.L[3].bias, acc, lane, t0);
            dense<PREC, 16, 7, 0, TN>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, bf, acc, lane, t0);
            softplus_inplace(acc);
            tp_publish<PREC, TN>(frag, lane, t0, acc);
        } else if (t0 < 7) {          // (WAVES = 4, wave 3: tile 6 only)
            load_accvec<7, 0, 1>(blob, LY.L[3].bias, a1, lane, t0);
            dense<PREC, 16, 7, 0, 1>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, bf, a1, lane, t0);
            softplus_inplace(a1);
            tp_publish<PREC, 1>(frag, lane, t0, a1);
        } else {                      // (WAVES = 8, wave 7: nothing to publish, keep the barrier count)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        tp_gather<PREC, 14>(frag, lane, bf);
#pragma unroll
        for (int i = 0; i < 3; ++i) bf[14 + i] = pef[i];      // skip connection (fields.py:83-84)
        // layer 4 (17 k-steps)
        load_accvec<8, 0, TN>(blob, LY.L[4].bias, acc, lane, t0);
        dense<PREC, 17, 8, 0, TN>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, bf, acc, lane, t0);
        softplus_inplace(acc);
        tp_publish<PREC, TN>(frag, lane, t0, acc);
        tp_gather<PREC, 16>(frag, lane, bf);
        for (int l = 5; l <= 7; ++l) {
            load_accvec<8, 0, TN>(blob, LY.L[l].bias, acc, lane, t0);
            dense<PREC, 16, 8, 0, TN>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, acc, lane, t0);
            softplus_inplace(acc);
            tp_publish<PREC, TN>(frag, lane, t0, acc);
            tp_gather<PREC, 16>(frag, lane, bf);
        }
        // layer 8: only the sdf row (tile 8 of 9) is needed; wave 0 computes it
        if (wave == 0) {
            load_accvec<9, 8, 1>(blob, LY.L[8].bias, a1, lane);
            dense<PREC, 16, 9, 8, 1>(blob, LY.L[8].fwd_hi, LY.L[8].fwd_lo, bf, a1, lane);
            if (valid && lane < 32) sdf_out[n] = a1[0][0];
        }
    }
}

// ---- K2 ----------------------------------------------------------------------------------------------------
// g[t] *= sigma'(z_l) from the lane-private stash block; with TRAIN the product a_l also goes to its private block
template <int PREC, int TN, bool TRAIN>
FN_DEV void mul_sig_priv(f32x16 (&g)[TN], const unsigned char* __restrict__ ps, unsigned char* __restrict__ pa, int lane) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sv[4], av[4];
            sig_get(ps, t * 4 + q, lane, sv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                av[e] = g[t][4 * q + e] * sv[e];
                g[t][4 * q + e] = av[e];
            }
            if constexpr (TRAIN) priv_put<PREC>(pa, t * 4 + q, lane, av);
        }
}

template <int PREC, bool TRAIN>
__global__ void __launch_bounds__(64, 1) sdf_fwd_grad_kernel(const unsigned char* blob, PointSrc src, long N,
                                                             SdfStash st, float* __restrict__ sdf_out,
                                                             float* __restrict__ feat_out, float* __restrict__ normal_out) {
    __shared__ __attribute__((aligned(16))) unsigned char scr[kWaveScr];
    const int lane = threadIdx.x;
    const int r = lane & 31, h = lane >> 5;
    constexpr auto& LY = kSdfLayout;
    for (long tile = blockIdx.x; tile * 32 < N; tile += gridDim.x) {
        // launder the blob pointer: otherwise LICM hoists every statically addressed weight load out of the tile loop
        // (hundreds of VGPRs -> scratch spills)
        asm volatile("" : "+s"(blob));
        const long n = tile * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        const long n0 = tile * 32;
        float x[3], pe[39], jc[39];
        load_point(src, nc, x);
        posenc<6, true>(x, pe, jc);
        BFrag<PREC> bf[kMaxKS];
        f32x16 acc[9];
        sdf_forward_chain<PREC, false, true>(blob, pe, bf, acc, st, N, nc, lane, valid, scr, tile);
        constexpr size_t PB = priv_block_bytes<PREC>();
        const unsigned char* psb = st.ps + (size_t)tile * 8 * kSigBlockBytes;
        unsigned char* pab = TRAIN ? st.pa + (size_t)tile * 8 * PB : nullptr;
        if (valid && lane < 32) sdf_out[n] = acc[8][0];
        {
            f32x16(&a8)[8] = reinterpret_cast<f32x16(&)[8]>(acc);
            store_f32<8>(a8, feat_out, 256, nc, h, valid);
            if constexpr (TRAIN) store_stash<PREC, 8>(scr, lane, a8, st.feat_hi, st.feat_lo, 256, n0, N, 256);
        }
        // ---- reverse sweep: g = d sdf / d u_l   ----
        f32x16(&g8)[8] = reinterpret_cast<f32x16(&)[8]>(acc);
        load_accvec<8, 0, 8>(blob, LY.extra, g8, lane);                      // g_hat(h_8) = row 0 of W_8
        for (int l = 7; l >= 5; --l) {
            mul_sig_priv<PREC, 8, TRAIN>(g8, psb + (size_t)(l) * kSigBlockBytes, pab + (size_t)(l) * PB, lane);   // a_l
            if constexpr (TRAIN)
                store_stash<PREC, 8>(scr, lane, g8, st.a_hi + (size_t)l * N * 256, st.a_lo + (size_t)l * N * 256, 256, n0, N, 256);
            acc_to_bfrag<PREC, 8>(g8, bf);
            zero_acc(g8);
            dense<PREC, 16, 8, 0, 8, 0, kDeep<PREC>>(blob, LY.L[l].rev_hi, LY.L[l].rev_lo, bf, g8, lane);
        }
        // layer 4: outputs 9 row tiles: 0..6 -> g_hat(h_4), 7..8 -> q_skip (PE part of the skip input)
        f32x16 qskip[2];
        {
            mul_sig_priv<PREC, 8, TRAIN>(g8, psb + (size_t)(4) * kSigBlockBytes, pab + (size_t)(4) * PB, lane);
            if constexpr (TRAIN)
                store_stash<PREC, 8>(scr, lane, g8, st.a_hi + (size_t)4 * N * 256, st.a_lo + (size_t)4 * N * 256, 256, n0, N, 256);
            acc_to_bfrag<PREC, 8>(g8, bf);
            zero_acc(acc);
            dense<PREC, 16, 9, 0, 9, 0, kDeep<PREC>>(blob, LY.L[4].rev_hi, LY.L[4].rev_lo, bf, acc, lane);
            qskip[0] = acc[7];
            qskip[1] = acc[8];
        }
        // layer 3 (7 tiles of outputs -> 14 k-steps)
        {
            f32x16(&g7)[7] = reinterpret_cast<f32x16(&)[7]>(acc);
            mul_sig_priv<PREC, 7, TRAIN>(g7, psb + (size_t)(3) * kSigBlockBytes, pab + (size_t)(3) * PB, lane);
            if constexpr (TRAIN)
                store_stash<PREC, 7>(scr, lane, g7, st.a_hi + (size_t)3 * N * 256, st.a_lo + (size_t)3 * N * 256, 256, n0, N, 224);
            acc_to_bfrag<PREC, 7>(g7, bf);
            zero_acc(g8);
            dense<PREC, 14, 8, 0, 8, 0, kDeep<PREC>>(blob, LY.L[3].rev_hi, LY.L[3].rev_lo, bf, g8, lane);
        }
        for (int l = 2; l >= 1; --l) {
            mul_sig_priv<PREC, 8, TRAIN>(g8, psb + (size_t)(l) * kSigBlockBytes, pab + (size_t)(l) * PB, lane);
            if constexpr (TRAIN)
                store_stash<PREC, 8>(scr, lane, g8, st.a_hi + (size_t)l * N * 256, st.a_lo + (size_t)l * N * 256, 256, n0, N, 256);
            acc_to_bfrag<PREC, 8>(g8, bf);
            zero_acc(g8);
            dense<PREC, 16, 8, 0, 8, 0, kDeep<PREC>>(blob, LY.L[l].rev_hi, LY.L[l].rev_lo, bf, g8, lane);
        }
        // layer 0: 2 row tiles (39 PE inputs)
        f32x16 q[2];
        {
            mul_sig_priv<PREC, 8, TRAIN>(g8, psb, pab, lane);
            if constexpr (TRAIN) store_stash<PREC, 8>(scr, lane, g8, st.a_hi, st.a_lo, 256, n0, N, 256);
            acc_to_bfrag<PREC, 8>(g8, bf);
            zero_acc(q);
            dense<PREC, 16, 2, 0, 2, 0, kDeep<PREC>>(blob, LY.L[0].rev_hi, LY.L[0].rev_lo, bf, q, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) q[t][rr] += qskip[t][rr];
        }
        // normal = J^T q
        float nrm[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float coef[39];
#pragma unroll
            for (int f = 0; f < 39; ++f) coef[f] = ((f % 3) == c) ? jc[f] : 0.0f;
            const float part = acc_dot_partial<2, 39>(q, coef, h);
            nrm[c] = part + xor32(part);
        }
        if (valid && lane < 32) {
#pragma unroll
            for (int c = 0; c < 3; ++c) normal_out[n * 3 + c] = nrm[c];
        }
    }
}

// ---- K2, tensor-parallel workgroups --------------------------------------------------------------------------------
// Same maths, same stash layouts and the same results as sdf_fwd_grad_kernel, organised like sdf_fwd_tp_kernel: the 4
// wavefronts of a workgroup share one 32-sample tile, wave w owns output tiles 2w, 2w+1 of every layer.  A wave then
// needs ~250 registers instead of ~450, so TWO workgroups fit a CU (2 waves per SIMD): while one workgroup sits in a
// barrier or in its stash-store phase the other one feeds the matrix pipe -- the overlap the one-wave-per-SIMD kernel
// cannot have (measured there: 650 of 1111 us were un-overlapped store phases).
// Per layer: barrier, own tiles -> LDS (B fragments for the exchange + the [32][256] row image of the stash), barrier,
// every wave fetches all k-steps of the next layer and stores a quarter of the image rows.
constexpr int kTp2Lds = kTpLds + kWaveScr;       // fragments + row image (hi and lo planes): 68 096 bytes

// positional encoding of x as B fragments KS0..KS0+2, recomputed where needed (the point is laundered so that the
// evaluations are not merged and kept in 24 registers)
template <int PREC, int KS0>
FN_DEV void pe_frags_tp(const float (&x)[3], BFrag<PREC> (&bf)[kMaxKS], int h) {
    float xx[3] = {x[0], x[1], x[2]};
    asm volatile("" : "+v"(xx[0]), "+v"(xx[1]), "+v"(xx[2]));
    float pe[39], jc[39];
    posenc<6, false>(xx, pe, jc);
    vec_to_bfrag<PREC, 39, 3, KS0>(pe, bf, h);
}

constexpr int kTp2LdsTotal = kTp2Lds + 2 * 64 * 64;     // + q_skip of wave 0 (2 tiles x 64 lanes x 16 floats)

template <int PREC, bool TRAIN>
__global__ void __launch_bounds__(256, 2) sdf_fwd_grad_tp_kernel(const unsigned char* blob, PointSrc src, long N,
                                                                 SdfStash st, float* __restrict__ sdf_out,
                                                                 float* __restrict__ feat_out,
                                                                 float* __restrict__ normal_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    unsigned char* img = lds_ + kTpLds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    constexpr auto& LY = kSdfLayout;
    constexpr size_t PB = priv_block_bytes<PREC>();
    const size_t LS = (size_t)N * 256;
    for (long tile = blockIdx.x; tile * 32 < N; tile += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const long n = tile * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        const long n0 = tile * 32;
        // this wave's part of the lane-private blocks of the tile: slots 4*t0 .. 4*t0+7 of each layer
        unsigned char* psb = st.ps + (size_t)tile * 8 * kSigBlockBytes + (size_t)t0 * (kSigBlockBytes / 8);
        unsigned char* pab = TRAIN ? st.pa + (size_t)tile * 8 * PB + (size_t)t0 * (PB / 8) : nullptr;
        float x[3];
        load_point(src, nc, x);
        BFrag<PREC> bf[kMaxKS];
        pe_frags_tp<PREC, 0>(x, bf, h);             // every wave encodes the (same) 32 points itself

        if (wave == 0 && valid) {   // PE rows [N][48]
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int col = 16 * ks + 8 * g + 4 * h;
                    bf16x4 vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vh[e] = bf[ks].hi[4 * g + e];
                        if constexpr (PREC == 3) vl[e] = bf[ks].lo[4 * g + e];
                    }
                    *reinterpret_cast<bf16x4*>(st.pe_hi + nc * 48 + col) = vh;
                    if constexpr (PREC == 3) *reinterpret_cast<bf16x4*>(st.pe_lo + nc * 48 + col) = vl;
                }
        }
        f32x16 acc[2];
        f32x16(&a1)[1] = reinterpret_cast<f32x16(&)[1]>(acc);
        // ---------------- forward chain ----------------
#pragma unroll 1
        for (int l = 0; l <= 7; ++l) {
            asm volatile("" : "+s"(blob));      // per layer as well: keeps the static-offset addresses of the branches
                                                // below from being hoisted out of this loop (and spilled)
            unsigned char* ps_l = psb + (size_t)l * kSigBlockBytes;
            __bf16* h_hi = st.h_hi + l * LS;
            __bf16* h_lo = st.h_lo + l * LS;
            if (l == 3) {               // 7 output tiles (217 features): wave 3 owns tile 6 only
                if (wave < 3) {
                    load_accvec<7, 0, 2>(blob, LY.L[3].bias, acc, lane, t0);
                    tp_dense<PREC, 16, 7, 0, 2>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, frag, bf, acc, lane, t0);
                    softplus_ps<PREC, 2>(acc, ps_l, lane);
                    tp_exchange<PREC, 2, true, true>(frag, img, lane, t0, acc);
                } else {
                    load_accvec<7, 0, 1>(blob, LY.L[3].bias, a1, lane, t0);
                    tp_dense<PREC, 16, 7, 0, 1>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, frag, bf, a1, lane, t0);
                    softplus_ps<PREC, 1>(a1, ps_l, lane);
                    BFrag<PREC>* skip = nullptr;
                    if constexpr (kTpLdsB<PREC>) {      // the owner of the short tile also publishes the skip input
                        pe_frags_tp<PREC, 14>(x, bf, h);
                        skip = &bf[14];
                    }
                    tp_exchange<PREC, 1, true, true>(frag, img, lane, t0, a1, skip);
                }
                tp_operands<PREC, 14>(frag, lane, bf);
                if constexpr (!kTpLdsB<PREC>) pe_frags_tp<PREC, 14>(x, bf, h);   // skip connection (fields.py:83-84)
                tp_store_rows<PREC, 224>(img, lane, wave, h_hi, h_lo, n0, N);
            } else {
                load_accvec<8, 0, 2>(blob, LY.L[l].bias, acc, lane, t0);
                if (l == 0)
                    dense<PREC, 3, 8, 0, 2>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, bf, acc, lane, t0);
                else if (l == 4)
                    tp_dense<PREC, 17, 8, 0, 2>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, frag, bf, acc, lane, t0);
                else
                    tp_dense<PREC, 16, 8, 0, 2>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, frag, bf, acc, lane, t0);
                softplus_ps<PREC, 2>(acc, ps_l, lane);
                tp_exchange<PREC, 2, true, true>(frag, img, lane, t0, acc);
                tp_operands<PREC, 16>(frag, lane, bf);
                tp_store_rows<PREC, 256>(img, lane, wave, h_hi, h_lo, n0, N);
            }
        }
        // layer 8 (linear): feature tiles 0..7 (two per wave) and the sdf row (tile 8, wave 0)
        load_accvec<9, 0, 2>(blob, LY.L[8].bias, acc, lane, t0);
        tp_dense<PREC, 16, 9, 0, 2>(blob, LY.L[8].fwd_hi, LY.L[8].fwd_lo, frag, bf, acc, lane, t0);
        store_f32<2>(acc, feat_out + 32 * t0, 256, nc, h, valid);
        if constexpr (TRAIN) {
            tp_exchange<PREC, 2, false, true>(frag, img, lane, t0, acc);
            tp_store_rows<PREC, 256>(img, lane, wave, st.feat_hi, st.feat_lo, n0, N);
        }
        if (wave == 0) {
            f32x16 s1[1];
            load_accvec<9, 8, 1>(blob, LY.L[8].bias, s1, lane);
            tp_dense<PREC, 16, 9, 8, 1>(blob, LY.L[8].fwd_hi, LY.L[8].fwd_lo, frag, bf, s1, lane);
            if (valid && lane < 32) sdf_out[n] = s1[0][0];
        }
        // ---------------- reverse sweep: g = d sdf / d u_l   ----------------
        load_accvec<8, 0, 2>(blob, LY.extra, acc, lane, t0);                 // g_hat(h_8) = row 0 of W_8
        f32x16* qskip_lds = reinterpret_cast<f32x16*>(lds_ + kTp2Lds);     // [2][64] accumulator registers of wave 0

#pragma unroll 1
        for (int l = 7; l >= 1; --l) {
            asm volatile("" : "+s"(blob));
            const unsigned char* ps_l = psb + (size_t)l * kSigBlockBytes;
            unsigned char* pa_l = TRAIN ? pab + (size_t)l * PB : nullptr;
            __bf16* a_hi = st.a_hi + l * LS;
            __bf16* a_lo = st.a_lo + l * LS;
            if (l == 3) {   // g_hat(h_4): 7 tiles
                if (wave < 3) {
                    mul_sig_priv<PREC, 2, TRAIN>(acc, ps_l, pa_l, lane);
                    tp_exchange<PREC, 2, true, TRAIN>(frag, img, lane, t0, acc);
                } else {
                    mul_sig_priv<PREC, 1, TRAIN>(a1, ps_l, pa_l, lane);
                    tp_exchange<PREC, 1, true, TRAIN>(frag, img, lane, t0, a1);
                }
                tp_operands<PREC, 14>(frag, lane, bf);
                if constexpr (TRAIN) tp_store_rows<PREC, 224>(img, lane, wave, a_hi, a_lo, n0, N);
                zero_acc(acc);
                tp_dense<PREC, 14, 8, 0, 2>(blob, LY.L[3].rev_hi, LY.L[3].rev_lo, frag, bf, acc, lane, t0);
            } else {
                mul_sig_priv<PREC, 2, TRAIN>(acc, ps_l, pa_l, lane);                                     // a_l
                tp_exchange<PREC, 2, true, TRAIN>(frag, img, lane, t0, acc);
                tp_operands<PREC, 16>(frag, lane, bf);
                if constexpr (TRAIN) tp_store_rows<PREC, 256>(img, lane, wave, a_hi, a_lo, n0, N);
                if (l == 4) {   // 9 row tiles: 0..6 -> g_hat(h_4), 7..8 -> q_skip (PE part of the skip input; wave 0)
                    if (wave == 0) {   // parked in LDS until the end of the sweep (32 registers for 4 layers otherwise)
                        f32x16 qs[2];
                        zero_acc(qs);
                        tp_dense<PREC, 16, 9, 7, 2>(blob, LY.L[4].rev_hi, LY.L[4].rev_lo, frag, bf, qs, lane);
                        qskip_lds[lane] = qs[0];
                        qskip_lds[64 + lane] = qs[1];
                    }
                    if (wave < 3) {
                        zero_acc(acc);
                        tp_dense<PREC, 16, 9, 0, 2>(blob, LY.L[4].rev_hi, LY.L[4].rev_lo, frag, bf, acc, lane, t0);
                    } else {
                        zero_acc(a1);
                        tp_dense<PREC, 16, 9, 0, 1>(blob, LY.L[4].rev_hi, LY.L[4].rev_lo, frag, bf, a1, lane, t0);
                    }
                } else {
                    zero_acc(acc);
                    tp_dense<PREC, 16, 8, 0, 2>(blob, LY.L[l].rev_hi, LY.L[l].rev_lo, frag, bf, acc, lane, t0);
                }
            }
        }
        // layer 0: a_0, then the 2 row tiles of the 39 PE inputs (wave 0) and normal = J^T q
        mul_sig_priv<PREC, 2, TRAIN>(acc, psb, pab, lane);
        tp_exchange<PREC, 2, true, TRAIN>(frag, img, lane, t0, acc);
        tp_operands<PREC, 16>(frag, lane, bf);
        if constexpr (TRAIN) tp_store_rows<PREC, 256>(img, lane, wave, st.a_hi, st.a_lo, n0, N);
        if (wave == 0) {
            f32x16 q[2];
            zero_acc(q);
            tp_dense<PREC, 16, 2, 0, 2>(blob, LY.L[0].rev_hi, LY.L[0].rev_lo, frag, bf, q, lane);
            q[0] += qskip_lds[lane];
            q[1] += qskip_lds[64 + lane];
            float pe[39], jc[39];       // Jacobian coefficients of the encoding, recomputed (39 registers otherwise)
            {
                float xx[3] = {x[0], x[1], x[2]};
                asm volatile("" : "+v"(xx[0]), "+v"(xx[1]), "+v"(xx[2]));
                posenc<6, true>(xx, pe, jc);
            }
            float nrm[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float coef[39];
#pragma unroll
                for (int f = 0; f < 39; ++f) coef[f] = ((f % 3) == c) ? jc[f] : 0.0f;
                const float part = acc_dot_partial<2, 39>(q, coef, h);
                nrm[c] = part + xor32(part);
            }
            if (valid && lane < 32) {
#pragma unroll
                for (int c = 0; c < 3; ++c) normal_out[n * 3 + c] = nrm[c];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // all four waves finish the tile together
    }
}

// ---- K3 ----------------------------------------------------------------------------------------------------
// Backward of (sdf, feature, normal) w.r.t. the SDF-network weights: the two chains of SURVEY.md Appendix A.
//   ascending  (tangent of the reverse sweep): adj_0 = J nbar;  abar_l = W_l adj_l;  adj_{l+1} = s_l * abar_l;
//               coupling c_l = beta (1 - s_l) a_l abar_l   (= softplus'' * g_hat * abar)
//   descending (ordinary backprop):            zbar_8 = [fbar ; sbar];  ubar_l = W_l^T zbar_l;
//               zbar_{l-1} = s_{l-1} * ubar_l + c_{l-1}
// The operand matrices of dW_l = zbar_l^T u_l + a_l^T adj_l are written as bf16 planes for dw_gemm.hip.
template <int PREC, int TN>
FN_DEV void asc_post(f32x16 (&acc)[TN], const unsigned char* __restrict__ ps, const unsigned char* __restrict__ pa,
                     f32x4* __restrict__ cs, int lane) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sv[4], av[4];
            sig_get<true>(ps, t * 4 + q, lane, sv);
            priv_get<PREC>(pa, t * 4 + q, lane, av);
            f32x4 c;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float abar = acc[t][4 * q + e];
                c[e] = kBeta * (1.0f - sv[e]) * av[e] * abar;      // softplus'' * g_hat * abar  (a = s * g_hat)
                acc[t][4 * q + e] = sv[e] * abar;
            }
            stream_store<2>(cs + (t * 4 + q) * 64 + lane, c);
        }
}

template <int PREC, int TN>
FN_DEV void desc_post(f32x16 (&acc)[TN], const unsigned char* __restrict__ ps, const f32x4* __restrict__ cs, int lane) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sv[4];
            sig_get<true>(ps, t * 4 + q, lane, sv);
            const f32x4 c = stream_load<3>(cs + (t * 4 + q) * 64 + lane);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][4 * q + e] = sv[e] * acc[t][4 * q + e] + c[e];
        }
}

template <int PREC>
__global__ void __launch_bounds__(64, 1) sdf_bwd_kernel(const unsigned char* blob, PointSrc src, long N, SdfStash st,
                                                        SdfBwdBufs bb, const float* __restrict__ d_sdf,
                                                        const float* __restrict__ d_feat,
                                                        const float* __restrict__ d_normal) {
    __shared__ __attribute__((aligned(16))) unsigned char scr[kWaveScr];
    const int lane = threadIdx.x;
    const int r = lane & 31, h = lane >> 5;
    constexpr auto& LY = kSdfLayout;
    const size_t LS = (size_t)N * 256;   // layer stride of the [L][N][256] planes
    for (long tile = blockIdx.x; tile * 32 < N; tile += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const long n = tile * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        const long n0 = tile * 32;
        f32x4* cs = bb.cscratch + (size_t)tile * 8 * 32 * 64;
        constexpr size_t PB = priv_block_bytes<PREC>();
        const unsigned char* psb = st.ps + (size_t)tile * 8 * kSigBlockBytes;
        const unsigned char* pab = st.pa + (size_t)tile * 8 * PB;
        BFrag<PREC> bf[kMaxKS];
        BFrag<PREC> qf[3];
        f32x16 acc[9];
        f32x16(&a8)[8] = reinterpret_cast<f32x16(&)[8]>(acc);
        f32x16(&a7)[7] = reinterpret_cast<f32x16(&)[7]>(acc);
        // ---- qbar = J nbar ----
        {
            float x[3], pe[39], jc[39], qb[39];
            load_point(src, nc, x);
            posenc<6, true>(x, pe, jc);
            float nb[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) nb[c] = valid ? d_normal[nc * 3 + c] : 0.0f;
#pragma unroll
            for (int f = 0; f < 39; ++f) qb[f] = jc[f] * nb[f % 3];
            vec_to_bfrag<PREC, 39, 3, 0>(qb, bf, h);
#pragma unroll
            for (int i = 0; i < 3; ++i) qf[i] = bf[i];
            if (valid) {
#pragma unroll
                for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int col = 16 * ks + 8 * g + 4 * h;
                        bf16x4 vh, vl;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            vh[e] = bf[ks].hi[4 * g + e];
                            if constexpr (PREC == 3) vl[e] = bf[ks].lo[4 * g + e];
                        }
                        *reinterpret_cast<bf16x4*>(bb.qbar_hi + nc * 48 + col) = vh;
                        if constexpr (PREC == 3) *reinterpret_cast<bf16x4*>(bb.qbar_lo + nc * 48 + col) = vl;
                    }
            }
        }
        // ---- ascending chain ----
        zero_acc(a8);
        dense<PREC, 3, 8, 0, 8, 0, kDeep<PREC>>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, bf, a8, lane);
        asc_post<PREC, 8>(a8, psb, pab, cs, lane);
        store_stash<PREC, 8>(scr, lane, a8, bb.adj_hi, bb.adj_lo, 256, n0, N, 256);
        acc_to_bfrag<PREC, 8>(a8, bf);
        for (int l = 1; l <= 2; ++l) {
            zero_acc(a8);
            dense<PREC, 16, 8, 0, 8, 0, kDeep<PREC>>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, a8, lane);
            asc_post<PREC, 8>(a8, psb + (size_t)(l) * kSigBlockBytes, pab + (size_t)(l) * PB, cs + (size_t)l * 32 * 64, lane);
            store_stash<PREC, 8>(scr, lane, a8, bb.adj_hi + l * LS, bb.adj_lo + l * LS, 256, n0, N, 256);
            acc_to_bfrag<PREC, 8>(a8, bf);
        }
        {
            zero_acc(a7);
            dense<PREC, 16, 7, 0, 7, 0, kDeep<PREC>>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, bf, a7, lane);
            asc_post<PREC, 7>(a7, psb + (size_t)(3) * kSigBlockBytes, pab + (size_t)(3) * PB, cs + (size_t)3 * 32 * 64, lane);
            store_stash<PREC, 7>(scr, lane, a7, bb.adj_hi + 3 * LS, bb.adj_lo + 3 * LS, 256, n0, N, 224);
            acc_to_bfrag<PREC, 7>(a7, bf);
#pragma unroll
            for (int i = 0; i < 3; ++i) bf[14 + i] = qf[i];
        }
        {
            zero_acc(a8);
            dense<PREC, 17, 8, 0, 8, 0, kDeep<PREC>>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, bf, a8, lane);
            asc_post<PREC, 8>(a8, psb + (size_t)(4) * kSigBlockBytes, pab + (size_t)(4) * PB, cs + (size_t)4 * 32 * 64, lane);
            store_stash<PREC, 8>(scr, lane, a8, bb.adj_hi + 4 * LS, bb.adj_lo + 4 * LS, 256, n0, N, 256);
            acc_to_bfrag<PREC, 8>(a8, bf);
        }
        for (int l = 5; l <= 7; ++l) {
            zero_acc(a8);
            dense<PREC, 16, 8, 0, 8, 0, kDeep<PREC>>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, a8, lane);
            asc_post<PREC, 8>(a8, psb + (size_t)(l) * kSigBlockBytes, pab + (size_t)(l) * PB, cs + (size_t)l * 32 * 64, lane);
            store_stash<PREC, 8>(scr, lane, a8, bb.adj_hi + l * LS, bb.adj_lo + l * LS, 256, n0, N, 256);
            acc_to_bfrag<PREC, 8>(a8, bf);
        }
        // ---- descending chain ----
        load_f32<8>(a8, d_feat, 256, nc, h);
        if (!valid) zero_acc(a8);
        zero_acc(reinterpret_cast<f32x16(&)[1]>(acc[8]));
        if (h == 0 && valid) acc[8][0] = d_sdf[nc];
        store_stash<PREC, 8>(scr, lane, a8, bb.zbar_hi + 8 * LS, bb.zbar_lo + 8 * LS, 256, n0, N, 256);
        store_stash<PREC, 1>(scr, lane, reinterpret_cast<f32x16(&)[1]>(acc[8]), bb.zsdf_hi, bb.zsdf_lo, 32, n0, N, 32);
        acc_to_bfrag<PREC, 9>(acc, bf);
        zero_acc(a8);
        dense<PREC, 18, 8, 0, 8, 0, kDeep<PREC>>(blob, LY.L[8].rev_hi, LY.L[8].rev_lo, bf, a8, lane);
        for (int l = 7; l >= 5; --l) {
            // here a8 = ubar_{l+1} = hbar_{l+1};  zbar_l = s_l * hbar_{l+1} + c_l
            desc_post<PREC, 8>(a8, psb + (size_t)(l) * kSigBlockBytes, cs + (size_t)l * 32 * 64, lane);
            store_stash<PREC, 8>(scr, lane, a8, bb.zbar_hi + l * LS, bb.zbar_lo + l * LS, 256, n0, N, 256);
            acc_to_bfrag<PREC, 8>(a8, bf);
            zero_acc(a8);
            dense<PREC, 16, 8, 0, 8, 0, kDeep<PREC>>(blob, LY.L[l].rev_hi, LY.L[l].rev_lo, bf, a8, lane);
        }
        {   // zbar_4, then ubar_4 restricted to the h_4 rows (7 tiles of the 9-tile reverse pack)
            desc_post<PREC, 8>(a8, psb + (size_t)(4) * kSigBlockBytes, cs + (size_t)4 * 32 * 64, lane);
            store_stash<PREC, 8>(scr, lane, a8, bb.zbar_hi + 4 * LS, bb.zbar_lo + 4 * LS, 256, n0, N, 256);
            acc_to_bfrag<PREC, 8>(a8, bf);
            zero_acc(a7);
            dense<PREC, 16, 9, 0, 7, 0, kDeep<PREC>>(blob, LY.L[4].rev_hi, LY.L[4].rev_lo, bf, a7, lane);
        }
        {   // zbar_3 (7 tiles), ubar_3
            desc_post<PREC, 7>(a7, psb + (size_t)(3) * kSigBlockBytes, cs + (size_t)3 * 32 * 64, lane);
            store_stash<PREC, 7>(scr, lane, a7, bb.zbar_hi + 3 * LS, bb.zbar_lo + 3 * LS, 256, n0, N, 224);
            acc_to_bfrag<PREC, 7>(a7, bf);
            zero_acc(a8);
            dense<PREC, 14, 8, 0, 8, 0, kDeep<PREC>>(blob, LY.L[3].rev_hi, LY.L[3].rev_lo, bf, a8, lane);
        }
        for (int l = 2; l >= 1; --l) {
            desc_post<PREC, 8>(a8, psb + (size_t)(l) * kSigBlockBytes, cs + (size_t)l * 32 * 64, lane);
            store_stash<PREC, 8>(scr, lane, a8, bb.zbar_hi + l * LS, bb.zbar_lo + l * LS, 256, n0, N, 256);
            acc_to_bfrag<PREC, 8>(a8, bf);
            zero_acc(a8);
            dense<PREC, 16, 8, 0, 8, 0, kDeep<PREC>>(blob, LY.L[l].rev_hi, LY.L[l].rev_lo, bf, a8, lane);
        }
        desc_post<PREC, 8>(a8, psb, cs, lane);
        store_stash<PREC, 8>(scr, lane, a8, bb.zbar_hi, bb.zbar_lo, 256, n0, N, 256);
    }
}

}  // namespace fneus

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
using namespace fneus;

static inline int grid_for(long n_tiles) {
    long g = n_tiles;
    const long cap = 256 * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int fneus_sdf_fwd(const void* blob, const float* pts, const float* rays_o, const float* rays_d,
                             const float* t, int m, long n_pts, float* sdf_out, int prec, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    PointSrc src{pts, rays_o, rays_d, t, m > 0 ? m : 1};
    const long tiles = (n_pts + 31) / 32;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(blob);
    // up to one workgroup per CU: 4 waves share a tile (latency-bound launches); beyond that one wave per tile
    static const bool no_tp = getenv("FNEUS_K1_NO_TP") != nullptr;
    const bool tp = tiles <= FNEUS_TP_MAX_TILES && !no_tp;
    static const bool tp8 = getenv("FNEUS_K1_TP_WAVES") ? atoi(getenv("FNEUS_K1_TP_WAVES")) == 8 : false;
    if (prec == 3 && tp)
        if (tp8) hipLaunchKernelGGL((sdf_fwd_tp_kernel<3, 8>), dim3((unsigned)(tiles < 512 ? tiles : 512)), dim3(512), 0, stream, b, src, n_pts, sdf_out);
        else hipLaunchKernelGGL((sdf_fwd_tp_kernel<3, 4>), dim3((unsigned)(tiles < 512 ? tiles : 512)), dim3(256), 0, stream, b, src, n_pts, sdf_out);
    else if (prec == 1 && tp)
        if (tp8) hipLaunchKernelGGL((sdf_fwd_tp_kernel<1, 8>), dim3((unsigned)(tiles < 512 ? tiles : 512)), dim3(512), 0, stream, b, src, n_pts, sdf_out);
        else hipLaunchKernelGGL((sdf_fwd_tp_kernel<1, 4>), dim3((unsigned)(tiles < 512 ? tiles : 512)), dim3(256), 0, stream, b, src, n_pts, sdf_out);
    else if (prec == 3)
        hipLaunchKernelGGL(sdf_fwd_kernel<3>, dim3(grid_for(tiles)), dim3(64), 0, stream, b, src, n_pts, sdf_out);
    else if (prec == 1)
        hipLaunchKernelGGL(sdf_fwd_kernel<1>, dim3(grid_for(tiles)), dim3(64), 0, stream, b, src, n_pts, sdf_out);
    else
        return -2;
    return fneus::launch_status();
}

extern "C" int fneus_sdf_fwd_grad(const void* blob, const float* pts, const float* rays_o, const float* rays_d,
                                  const float* t, int m, long n_pts, const FneusSdfStash* stash, float* sdf_out,
                                  float* feat_out, float* normal_out, int prec, int train, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    PointSrc src{pts, rays_o, rays_d, t, m > 0 ? m : 1};
    const long tiles = (n_pts + 31) / 32;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(blob);
    SdfStash st = *stash;
    // tensor-parallel workgroups by default; FNEUS_K2_TP=0 selects the one-wave-per-tile kernel (kept for comparison).
    // N = 65 536, parity mode: train 845 vs 840 us standalone but 0.64 vs 0.79 ms inside the step (two workgroups per CU
    // interleave with their neighbours' traffic), inference 620 vs 650 us; bf16 mode 405 vs 565 us.
    static const int tp_mode = getenv("FNEUS_K2_TP") ? atoi(getenv("FNEUS_K2_TP")) : 1;
    if (tp_mode) {
#ifndef FNEUS_K2_GRID_CAP
#define FNEUS_K2_GRID_CAP (256 * 2 * 4)
#endif
        const long cap = FNEUS_K2_GRID_CAP;
        dim3 g2((unsigned)(tiles < cap ? tiles : cap)), b2(256);
#define FNEUS_K2TP(P, T)                                                                                              \
    do {                                                                                                              \
        static bool attr_done = false;                                                                                \
        if (!attr_done) {                                                                                             \
            allow_big_lds(sdf_fwd_grad_tp_kernel<P, T>);                                                              \
            attr_done = true;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((sdf_fwd_grad_tp_kernel<P, T>), g2, b2, kTp2LdsTotal, stream, b, src, n_pts, st, sdf_out,       \
                           feat_out, normal_out);                                                                     \
    } while (0)
        if (prec == 3 && train) FNEUS_K2TP(3, true);
        else if (prec == 3) FNEUS_K2TP(3, false);
        else if (prec == 1 && train) FNEUS_K2TP(1, true);
        else if (prec == 1) FNEUS_K2TP(1, false);
        else return -2;
#undef FNEUS_K2TP
        return fneus::launch_status();
    }
    dim3 grid(grid_for(tiles)), blk(64);
    if (prec == 3 && train)
        hipLaunchKernelGGL((sdf_fwd_grad_kernel<3, true>), grid, blk, 0, stream, b, src, n_pts, st, sdf_out, feat_out, normal_out);
    else if (prec == 3)
        hipLaunchKernelGGL((sdf_fwd_grad_kernel<3, false>), grid, blk, 0, stream, b, src, n_pts, st, sdf_out, feat_out, normal_out);
    else if (prec == 1 && train)
        hipLaunchKernelGGL((sdf_fwd_grad_kernel<1, true>), grid, blk, 0, stream, b, src, n_pts, st, sdf_out, feat_out, normal_out);
    else if (prec == 1)
        hipLaunchKernelGGL((sdf_fwd_grad_kernel<1, false>), grid, blk, 0, stream, b, src, n_pts, st, sdf_out, feat_out, normal_out);
    else
        return -2;
    return fneus::launch_status();
}

extern "C" int fneus_sdf_bwd(const void* blob, const float* pts, const float* rays_o, const float* rays_d,
                             const float* t, int m, long n_pts, const FneusSdfStash* stash, const FneusSdfBwdBufs* bufs,
                             const float* d_sdf, const float* d_feat, const float* d_normal, int prec,
                             fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    PointSrc src{pts, rays_o, rays_d, t, m > 0 ? m : 1};
    const unsigned char* b = reinterpret_cast<const unsigned char*>(blob);
    SdfStash st = *stash;
    SdfBwdBufs bb = *bufs;
    dim3 grid(grid_for((n_pts + 31) / 32)), blk(64);
    if (prec == 3)
        hipLaunchKernelGGL(sdf_bwd_kernel<3>, grid, blk, 0, stream, b, src, n_pts, st, bb, d_sdf, d_feat, d_normal);
    else if (prec == 1)
        hipLaunchKernelGGL(sdf_bwd_kernel<1>, grid, blk, 0, stream, b, src, n_pts, st, bb, d_sdf, d_feat, d_normal);
    else
        return -2;
    return fneus::launch_status();
}
