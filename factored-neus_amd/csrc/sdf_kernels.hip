// SDF-network kernels (reference models/fields.py:74-111 SDFNetwork.forward/.sdf/.gradient and their autograd).
//   K1 sdf_fwd        : PE -> 9 dense layers (Softplus beta=100) -> sdf                      (no-grad sampler path)
//   K2 sdf_fwd_grad   : sdf, feature[256], normal = d sdf/dx (analytic reverse sweep) + bf16 stash for backward
//   K3 sdf_bwd_chain  : double-backward chains (ascending + descending); writes the
//                       operand planes of the weight-gradient GEMM (dw_gemm_pp.hip)
// One wavefront = 32 samples, whole chain register resident; weights come pre-packed from pack.hip (L2 resident).
#include <stdlib.h>
#include "pp_engine.h"
#include "fneus_kernels.h"
#include "sdf_w8.h"
#include "sdf_r8.h"
#include "s_prefetch.h"

#ifndef FNEUS_K1_W8_BIG_DEFAULT
#define FNEUS_K1_W8_BIG_DEFAULT 31        // round 3: two-pass pipelined kernel, 8 waves (sdf_p2_kernels.hip)
#define FNEUS_K1_W8_SMALL_DEFAULT 2       // round 3: one tile per 8-wave workgroup, whole layers primed in registers
#endif
#ifndef FNEUS_K2_P2_DEFAULT
#define FNEUS_K2_P2_DEFAULT 1
#endif
#ifndef FNEUS_K3_R8_DEFAULT
#define FNEUS_K3_R8_DEFAULT 1
#endif
#ifndef FNEUS_K2_REV8_DEFAULT
#define FNEUS_K2_REV8_DEFAULT 1
#endif
#ifndef FNEUS_K2_CHUNKS_DEFAULT
#define FNEUS_K2_CHUNKS_DEFAULT 1
#endif
#ifndef FNEUS_K2_OCC
#define FNEUS_K2_OCC 2      // workgroups per CU the tensor-parallel kernels of this file are compiled for (experiments: 3)
#endif

namespace fneus {

// prefetch depth of the kernels that carry stash traffic (K3)
template <int PREC> constexpr int kDeep = PREC == 3 ? 4 : 8;

template <int TN>
FN_DEV void softplus_inplace(f32x16 (&acc)[TN]) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = softplus100(acc[t][r]);
}

// ---- forward chain of K1 (one wave per tile) ------------------------------------------------------------------
// On return acc[8] row 0 (reg 0 of lane half 0) = sdf: only tile 8 of the last layer is computed.
template <int PREC>
FN_DEV void sdf_forward_chain(const unsigned char* __restrict__ blob, const float (&pe)[39], BFrag<PREC> (&bf)[kMaxKS],
                              f32x16 (&acc)[9], int lane) {
    const int h = lane >> 5;
    constexpr auto& LY = kSdfLayout;
    BFrag<PREC> pef[3];
    vec_to_bfrag<PREC, 39, 3, 0>(pe, bf, h);
#pragma unroll
    for (int i = 0; i < 3; ++i) pef[i] = bf[i];
    f32x16(&a8)[8] = reinterpret_cast<f32x16(&)[8]>(acc);
    load_accvec<8, 0, 8>(blob, LY.L[0].bias, a8, lane);
    dense<PREC, 3, 8, 0, 8>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, bf, a8, lane);
    softplus_inplace(a8);
    acc_to_bfrag<PREC, 8>(a8, bf);
    for (int l = 1; l <= 2; ++l) {
        load_accvec<8, 0, 8>(blob, LY.L[l].bias, a8, lane);
        dense<PREC, 16, 8, 0, 8>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, a8, lane);
        softplus_inplace(a8);
        acc_to_bfrag<PREC, 8>(a8, bf);
    }
    {   // layer 3: 256 -> 217 (7 tiles); its output + PE is the input of layer 4 (skip connection, fields.py:83-84)
        f32x16(&a7)[7] = reinterpret_cast<f32x16(&)[7]>(acc);
        load_accvec<7, 0, 7>(blob, LY.L[3].bias, a7, lane);
        dense<PREC, 16, 7, 0, 7>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, bf, a7, lane);
        softplus_inplace(a7);
        acc_to_bfrag<PREC, 7>(a7, bf);
#pragma unroll
        for (int i = 0; i < 3; ++i) bf[14 + i] = pef[i];
    }
    load_accvec<8, 0, 8>(blob, LY.L[4].bias, a8, lane);       // layer 4 (17 k-steps; 1/sqrt2 folded into the pack)
    dense<PREC, 17, 8, 0, 8>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, bf, a8, lane);
    softplus_inplace(a8);
    acc_to_bfrag<PREC, 8>(a8, bf);
    for (int l = 5; l <= 7; ++l) {
        load_accvec<8, 0, 8>(blob, LY.L[l].bias, a8, lane);
        dense<PREC, 16, 8, 0, 8>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, a8, lane);
        softplus_inplace(a8);
        acc_to_bfrag<PREC, 8>(a8, bf);
    }
    f32x16(&a1)[1] = reinterpret_cast<f32x16(&)[1]>(acc[8]);    // layer 8 (linear): the sdf row only
    load_accvec<9, 8, 1>(blob, LY.L[8].bias, a1, lane);
    dense<PREC, 16, 9, 8, 1>(blob, LY.L[8].fwd_hi, LY.L[8].fwd_lo, bf, a1, lane);
}

// ---- K1 ----------------------------------------------------------------------------------------------------
template <int PREC>
__global__ void __launch_bounds__(64, 1) sdf_fwd_kernel(const unsigned char* blob, PointSrc src, long N,
                                                        float* __restrict__ sdf_out) {
    const int lane = threadIdx.x;
    const int r = lane & 31;
    for (long tile = blockIdx.x; tile * 32 < N; tile += gridDim.x) {
        // launder the blob pointer: otherwise LICM hoists every statically addressed weight load out of the tile loop
        // (hundreds of VGPRs -> scratch spills)
        asm volatile("" : "+s"(blob));
        const long n = tile * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        float x[3], pe[39], jc[39];
        load_point(src, nc, x);
        posenc<6, false>(x, pe, jc);
        BFrag<PREC> bf[kMaxKS];
        f32x16 acc[9];
        sdf_forward_chain<PREC>(blob, pe, bf, acc, lane);
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
constexpr int kTpLds = 18 * 2 * kFragBytes;      // up to 18 k-steps x (hi, lo) fragments
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

// ---- K2: tensor-parallel workgroups ----------------------------------------------------------------------------------
// sdf, feature[256], normal = d sdf/dx (analytic reverse sweep) + the stash for the backward.
// The 4 wavefronts of a workgroup share one 32-sample tile, wave w owns output tiles 2w, 2w+1 of every layer (like
// sdf_fwd_tp_kernel); a wave needs ~250 registers, so TWO workgroups fit a CU (2 waves per SIMD): while one sits in a
// barrier the other feeds the matrix pipe.  Per layer: barrier, own tiles -> LDS as B fragments for the exchange AND, in
// training, the same fragments -> global memory as the layer's plane (fneus_pp.h: one 16-byte store per lane and
// fragment, hi part; lo part only in the exact-gradient mode), barrier.  No row image, no transposition: the
// weight-gradient GEMM reads the fragments as they are.
constexpr int kTp2Lds = kTpLds;                  // the B fragments of the layer in flight

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
__global__ void __launch_bounds__(256, FNEUS_K2_OCC) sdf_fwd_grad_tp_kernel(const unsigned char* blob, PointSrc src, long N,
                                                                 SdfStash st, float* __restrict__ sdf_out,
                                                                 float* __restrict__ feat_out,
                                                                 float* __restrict__ normal_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    constexpr auto& LY = kSdfLayout;
    const long tiles = pp_tiles(N);
    const bool lo_planes = PREC == 3 && st.h_lo != nullptr;
    for (long tile = blockIdx.x; tile * 32 < N; tile += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const long n = tile * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        unsigned char* sig_t = st.ps + (size_t)tile * 8 * kPPBlock;      // sigma' blocks of this tile, [layer]
        float x[3];
        load_point(src, nc, x);
        BFrag<PREC> bf[kMaxKS];
        pe_frags_tp<PREC, 0>(x, bf, h);             // every wave encodes the (same) 32 points itself
        if constexpr (TRAIN) {
            if (wave == 0)                           // PE plane [tiles][4 fragments] (fragment 3 stays zero)
                frags_to_plane<PREC, 3>(bf, 0, st.pe_hi + (size_t)tile * 4 * kFragBytes,
                                        lo_planes ? st.pe_lo + (size_t)tile * 4 * kFragBytes : nullptr, pl, valid);
        }
        f32x16 acc[2];
        f32x16(&a1)[1] = reinterpret_cast<f32x16(&)[1]>(acc);
        // ---------------- forward chain ----------------
#pragma unroll 1
        for (int l = 0; l <= 7; ++l) {
            asm volatile("" : "+s"(blob));      // per layer as well: keeps the static-offset addresses of the branches
                                                // below from being hoisted out of this loop (and spilled)
            unsigned char* sblk = sig_t + (size_t)l * kPPBlock;
            unsigned char* hb_hi = TRAIN ? st.h_hi + ((size_t)l * tiles + tile) * kPPBlock : nullptr;
            unsigned char* hb_lo = (TRAIN && lo_planes) ? st.h_lo + ((size_t)l * tiles + tile) * kPPBlock : nullptr;
            if (l == 3) {               // 7 output tiles (217 features): wave 3 owns tile 6 only
                if (wave < 3) {
                    load_accvec<7, 0, 2>(blob, LY.L[3].bias, acc, lane, t0);
                    tp_dense<PREC, 16, 7, 0, 2>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, frag, bf, acc, lane, t0);
                    softplus_sig8<2>(acc, sblk, t0, lane);
                    tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, hb_hi, hb_lo, pl, valid);
                } else {
                    load_accvec<7, 0, 1>(blob, LY.L[3].bias, a1, lane, t0);
                    tp_dense<PREC, 16, 7, 0, 1>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, frag, bf, a1, lane, t0);
                    softplus_sig8<1>(a1, sblk, t0, lane);
                    BFrag<PREC>* skip = nullptr;
                    if constexpr (kTpLdsB<PREC>) {      // the owner of the short tile also publishes the skip input
                        pe_frags_tp<PREC, 14>(x, bf, h);
                        skip = &bf[14];
                    }
                    tp_exchange_pp<PREC, 1, true>(frag, lane, t0, a1, hb_hi, hb_lo, pl, valid, skip);
                }
                tp_operands<PREC, 14>(frag, lane, bf);
                if constexpr (!kTpLdsB<PREC>) pe_frags_tp<PREC, 14>(x, bf, h);   // skip connection (fields.py:83-84)
            } else {
                load_accvec<8, 0, 2>(blob, LY.L[l].bias, acc, lane, t0);
                if (l == 0)
                    dense<PREC, 3, 8, 0, 2>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, bf, acc, lane, t0);
                else if (l == 4)
                    tp_dense<PREC, 17, 8, 0, 2>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, frag, bf, acc, lane, t0);
                else
                    tp_dense<PREC, 16, 8, 0, 2>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, frag, bf, acc, lane, t0);
                softplus_sig8<2>(acc, sblk, t0, lane);
                tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, hb_hi, hb_lo, pl, valid);
                tp_operands<PREC, 16>(frag, lane, bf);
            }
        }
        // layer 8 (linear): feature tiles 0..7 (two per wave) and the sdf row (tile 8, wave 0)
        load_accvec<9, 0, 2>(blob, LY.L[8].bias, acc, lane, t0);
        tp_dense<PREC, 16, 9, 0, 2>(blob, LY.L[8].fwd_hi, LY.L[8].fwd_lo, frag, bf, acc, lane, t0);
        store_f32<2>(acc, feat_out + 32 * t0, 256, nc, h, valid);
        if constexpr (TRAIN)        // the feature plane: the colour network's weight-gradient operand
            tp_exchange_pp<PREC, 2, false>(frag, lane, t0, acc, st.feat_hi + (size_t)tile * kPPBlock,
                                           (PREC == 3 && st.feat_lo) ? st.feat_lo + (size_t)tile * kPPBlock : nullptr, pl, valid);      // (hi + lo whatever the gradient precision: the colour network's input, round 6)
        if (wave == 0) {
            f32x16 s1[1];
            load_accvec<9, 8, 1>(blob, LY.L[8].bias, s1, lane);
            tp_dense<PREC, 16, 9, 8, 1>(blob, LY.L[8].fwd_hi, LY.L[8].fwd_lo, frag, bf, s1, lane);
            if (valid && lane < 32) sdf_out[n] = s1[0][0];
        }
        // ---------------- reverse sweep: g = d sdf / d u_l   ----------------
#ifdef FNEUS_DBG_K2_NO_REVERSE      // timing experiments only
        if (N > 0) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); continue; }
#endif
        load_accvec<8, 0, 2>(blob, LY.extra, acc, lane, t0);                 // g_hat(h_8) = row 0 of W_8
        f32x16* qskip_lds = reinterpret_cast<f32x16*>(lds_ + kTp2Lds);     // [2][64] accumulator registers of wave 0

#pragma unroll 1
        for (int l = 7; l >= 1; --l) {
            asm volatile("" : "+s"(blob));
            const unsigned char* sblk = sig_t + (size_t)l * kPPBlock;
            unsigned char* ab_hi = TRAIN ? st.a_hi + ((size_t)l * tiles + tile) * kPPBlock : nullptr;
            unsigned char* ab_lo = (TRAIN && lo_planes) ? st.a_lo + ((size_t)l * tiles + tile) * kPPBlock : nullptr;
            if (l == 3) {   // g_hat(h_4): 7 tiles
                if (wave < 3) {
                    mul_sig8<2, false>(acc, sblk, t0, lane);
                    tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, ab_hi, ab_lo, pl, valid);
                } else {
                    mul_sig8<1, false>(a1, sblk, t0, lane);
                    tp_exchange_pp<PREC, 1, true>(frag, lane, t0, a1, ab_hi, ab_lo, pl, valid);
                }
                tp_operands<PREC, 14>(frag, lane, bf);
                zero_acc(acc);
                tp_dense<PREC, 14, 8, 0, 2>(blob, LY.L[3].rev_hi, LY.L[3].rev_lo, frag, bf, acc, lane, t0);
            } else {
                mul_sig8<2, false>(acc, sblk, t0, lane);                                     // a_l
                tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, ab_hi, ab_lo, pl, valid);
                tp_operands<PREC, 16>(frag, lane, bf);
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
        mul_sig8<2, false>(acc, sig_t, t0, lane);
        tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, TRAIN ? st.a_hi + (size_t)tile * kPPBlock : nullptr,
                                      (TRAIN && lo_planes) ? st.a_lo + (size_t)tile * kPPBlock : nullptr, pl, valid);
        tp_operands<PREC, 16>(frag, lane, bf);
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

// ---- K2 on workgroups of HB sample halves ----------------------------------------------------------------------------
// sdf_fwd_grad_tp_kernel generalised like K3 (sdf_bwd_tph_kernel): HB 32-sample tiles share one pass over the weight
// fragments.  The one-off jobs of a tile are spread over the waves: wave 0 encodes the points (k-steps 0..2 of layer 0, the
// PE plane, and a copy parked in k-steps 16..18 that layer 4 takes as its skip input), wave hb computes the sdf row, the
// q_skip tiles and the normal of half hb.  q_skip waits for the end of the reverse sweep in a global scratch line per lane
// (st.qs: written and read back by the same lane; L2 resident).
constexpr int kK2Half = 19 * 2 * kFragBytes;

template <int TN, int HB>
FN_DEV void bias_h(const unsigned char* __restrict__ blob, uint32_t off, f32x16 (&acc)[TN][HB], int lane, int t0_rt) {
    const f32x16 FN_GLOBAL* __restrict__ p = reinterpret_cast<const f32x16 FN_GLOBAL*>((gblob_t)blob + off);
    const int h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const f32x16 v = p[(t0_rt + i) * 2 + h];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) acc[i][hb] = v;
    }
}

// ---- K1 on 64-sample workgroups (chip-filling launches: the 32 768 coarse depths of a step, the 1 M-point march of stage 2) ----
// The chain kernels stream every weight fragment from L2 once per 32-sample tile -- at the full MFMA rate ~52 TB/s chip-wide
// against the ~18 TB/s the L2s deliver, which is what holds them at 35-40 % MFMA busy.  Two tiles per workgroup share one pass
// over the fragments (dense_ldsb_h): half the stream per sample.  Tensor-parallel form (wave w owns tiles 2w, 2w+1), B fragments
// in LDS, two workgroups per CU; the encoding's fragments are built by waves 0, 1 (one half each), used by layer 0 and parked
// in k-steps 16..18 for the skip input of layer 4.
template <int PREC>
__global__ void __launch_bounds__(256, 2) sdf_fwd_tph_kernel(const unsigned char* blob, PointSrc src, long N,
                                                             float* __restrict__ sdf_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    constexpr int HB = 2, HALF = kK2Half;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    constexpr auto& LY = kSdfLayout;
    const long groups = (N + 32 * HB - 1) / (32 * HB);
    unsigned char* none[HB] = {nullptr, nullptr};
    for (long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        asm volatile("" : "+s"(blob));
        long n[HB], nc[HB];
        bool valid[HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            n[hb] = (grp * HB + hb) * 32 + r;
            valid[hb] = n[hb] < N;
            nc[hb] = valid[hb] ? n[hb] : N - 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the previous group's fragments are consumed
        if (wave < HB) {
            const int hb = wave;
            float x[3], pe[39], jc[39];
            load_point(src, nc[hb], x);
            posenc<6, false>(x, pe, jc);
            BFrag<PREC> pf[kMaxKS];
            vec_to_bfrag<PREC, 39, 3, 0>(pe, pf, h);
            frags_to_lds<PREC, 3>(frag + hb * HALF, lane, 0, pf);
            frags_to_lds<PREC, 3>(frag + hb * HALF, lane, 16, pf);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        f32x16 acc[2][HB];
#pragma unroll 1
        for (int l = 0; l <= 7; ++l) {
            asm volatile("" : "+s"(blob));
            if (l == 3) {               // 7 output tiles (217 features): wave 3 owns tile 6 only
                if (wave < 3) {
                    bias_h<2, HB>(blob, LY.L[3].bias, acc, lane, t0);
                    tph_dense<PREC, 16, 7, 0, 2, true, HB, HALF>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, frag, acc, lane, t0);
                } else {
                    f32x16(&a1)[1][HB] = reinterpret_cast<f32x16(&)[1][HB]>(acc);
                    bias_h<1, HB>(blob, LY.L[3].bias, a1, lane, t0);
                    tph_dense<PREC, 16, 7, 0, 1, true, HB, HALF>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, frag, a1, lane, t0);
                }
            } else {
                bias_h<2, HB>(blob, LY.L[l].bias, acc, lane, t0);
                if (l == 0)
                    tph_dense<PREC, 3, 8, 0, 2, true, HB, HALF>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, frag, acc, lane, t0);
                else if (l == 4)
                    tph_dense<PREC, 17, 8, 0, 2, true, HB, HALF>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, frag, acc, lane, t0);
                else
                    tph_dense<PREC, 16, 8, 0, 2, true, HB, HALF>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, frag, acc, lane, t0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) {
                    f32x16(&one)[1] = reinterpret_cast<f32x16(&)[1]>(acc[i][hb]);
                    softplus_inplace(one);
                }
            if (l == 3 && wave == 3) {
                f32x16(&a1)[1][HB] = reinterpret_cast<f32x16(&)[1][HB]>(acc);
                BFrag<PREC> ex[HB * 3];        // skip connection (fields.py:83-84): k-steps 14..16 of layer 4 = the parked encoding
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) lds_to_frags<PREC, 3>(frag + hb * HALF, lane, 16, &ex[hb * 3]);
                tph_exchange<PREC, 1, true, HB, HALF>(frag, lane, t0, a1, none, none, pl, valid, ex, 14, 3);
            } else {
                tph_exchange<PREC, 2, true, HB, HALF>(frag, lane, t0, acc, none, none, pl, valid);
            }
        }
        // layer 8: only the sdf row (tile 8 of 9) is needed; wave 0 computes it for both halves
        if (wave == 0) {
            f32x16 a1[1][HB];
            bias_h<1, HB>(blob, LY.L[8].bias, a1, lane, 8);
            tph_dense<PREC, 16, 9, 8, 1, true, HB, HALF>(blob, LY.L[8].fwd_hi, LY.L[8].fwd_lo, frag, a1, lane);
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
                if (valid[hb] && lane < 32) sdf_out[n[hb]] = a1[0][hb][0];
        }
    }
}

// REV: the reverse sweep alone (normal, a_l planes) on the sigma' blocks a forward launch has written (sdf_p2_train_kernels.hip)
template <int PREC, bool TRAIN, int HB, int GP, bool REV = false>
__global__ void __launch_bounds__(256, 2) sdf_fwd_grad_tph_kernel(const unsigned char* blob, PointSrc src, long N, SdfStash st,
                                                                  float* __restrict__ sdf_out, float* __restrict__ feat_out,
                                                                  float* __restrict__ normal_out, long grp_begin = 0, long grp_end = -1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    constexpr int HALF = kK2Half;
    constexpr bool LO = TRAIN && PREC == 3 && GP == 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    constexpr auto& LY = kSdfLayout;
    const long tiles = pp_tiles(N);
    const long groups = grp_end >= 0 ? grp_end : (N + 32 * HB - 1) / (32 * HB);      // this launch: groups grp_begin .. groups - 1
    uint32_t spf = 0;                    // scalar prefetch in flight (s_prefetch.h)
    for (long grp = grp_begin + blockIdx.x; grp < groups; grp += gridDim.x) {
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
        auto blk = [&](unsigned char* base, int slot, int hb) { return base + ((size_t)slot * tiles + tile[hb]) * kPPBlock; };
        auto sig = [&](int l, int hb) { return st.ps + ((size_t)tile[hb] * 8 + l) * kPPBlock; };
        f32x16 acc[2][HB];
        if constexpr (!REV) {
        // ---- positional encoding (wave 0) ----
        if (wave == 0) {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                float x[3], pe[39], jc[39];
                load_point(src, nc[hb], x);
                posenc<6, false>(x, pe, jc);
                BFrag<PREC> pf[kMaxKS];
                vec_to_bfrag<PREC, 39, 3, 0>(pe, pf, h);
                frags_to_lds<PREC, 3>(frag + hb * HALF, lane, 0, pf);
                frags_to_lds<PREC, 3>(frag + hb * HALF, lane, 16, pf);
                if constexpr (TRAIN)       // PE plane [tiles][4 fragments] (fragment 3 stays zero)
                    frags_to_plane<PREC, 3>(pf, 0, st.pe_hi + (size_t)tile[hb] * 4 * kFragBytes,
                                            LO ? st.pe_lo + (size_t)tile[hb] * 4 * kFragBytes : nullptr, pl, valid[hb]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---------------- forward chain ----------------
#pragma unroll 1
        for (int l = 0; l <= 7; ++l) {
            asm volatile("" : "+s"(blob));
            unsigned char *hb_hi[HB], *hb_lo[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                hb_hi[hb] = TRAIN ? blk(st.h_hi, l, hb) : nullptr;
                hb_lo[hb] = LO ? blk(st.h_lo, l, hb) : nullptr;
            }
            if (l == 3) {               // 7 output tiles (217 features): wave 3 owns tile 6 only
                if (wave < 3) {
                    bias_h<2, HB>(blob, LY.L[3].bias, acc, lane, t0);
                    tph_dense<PREC, 16, 7, 0, 2, true, HB, HALF>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, frag, acc, lane, t0);
                } else {
                    f32x16(&a1)[1][HB] = reinterpret_cast<f32x16(&)[1][HB]>(acc);
                    bias_h<1, HB>(blob, LY.L[3].bias, a1, lane, t0);
                    tph_dense<PREC, 16, 7, 0, 1, true, HB, HALF>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, frag, a1, lane, t0);
                }
            } else {
                bias_h<2, HB>(blob, LY.L[l].bias, acc, lane, t0);
                if (l == 0)
                    tph_dense<PREC, 3, 8, 0, 2, true, HB, HALF>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, frag, acc, lane, t0);
                else if (l == 4)
                    tph_dense<PREC, 17, 8, 0, 2, true, HB, HALF>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, frag, acc, lane, t0);
                else
                    tph_dense<PREC, 16, 8, 0, 2, true, HB, HALF>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, frag, acc, lane, t0);
            }
            const int tn_l = (l == 3 && wave == 3) ? 1 : 2;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (i < tn_l) {
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) {
                        f32x16(&one)[1] = reinterpret_cast<f32x16(&)[1]>(acc[i][hb]);
                        softplus_sig8<1>(one, sig(l, hb), t0 + i, lane);
                    }
                }
            if (l == 3 && wave == 3) {
                f32x16(&a1)[1][HB] = reinterpret_cast<f32x16(&)[1][HB]>(acc);
                BFrag<PREC> ex[HB * 3];        // skip connection (fields.py:83-84): k-steps 14..16 of layer 4 = the parked encoding
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) lds_to_frags<PREC, 3>(frag + hb * HALF, lane, 16, &ex[hb * 3]);
                tph_exchange<PREC, 1, true, HB, HALF>(frag, lane, t0, a1, hb_hi, hb_lo, pl, valid, ex, 14, 3);
            } else {
                tph_exchange<PREC, 2, true, HB, HALF>(frag, lane, t0, acc, hb_hi, hb_lo, pl, valid);
            }
        }
        // layer 8 (linear): feature tiles 0..7 (two per wave) and the sdf row (tile 8: wave hb for half hb)
        bias_h<2, HB>(blob, LY.L[8].bias, acc, lane, t0);
        tph_dense<PREC, 16, 9, 0, 2, true, HB, HALF>(blob, LY.L[8].fwd_hi, LY.L[8].fwd_lo, frag, acc, lane, t0);
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x16(&one)[1] = reinterpret_cast<f32x16(&)[1]>(acc[i][hb]);
                store_f32<1>(one, feat_out + 32 * (t0 + i), 256, nc[hb], h, valid[hb]);
            }
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
            if (wave == hb) {
                f32x16 s1[1][1];
                bias_h<1, 1>(blob, LY.L[8].bias, s1, lane, 8);
                tph_dense<PREC, 16, 9, 8, 1, true, 1, HALF>(blob, LY.L[8].fwd_hi, LY.L[8].fwd_lo, frag + hb * HALF, s1, lane);
                if (valid[hb] && lane < 32) sdf_out[n[hb]] = s1[0][0][0];
            }
        if constexpr (TRAIN) {      // the feature plane: the colour network's weight-gradient operand
            unsigned char *f_hi[HB], *f_lo[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                f_hi[hb] = st.feat_hi + (size_t)tile[hb] * kPPBlock;
                f_lo[hb] = (PREC == 3 && st.feat_lo) ? st.feat_lo + (size_t)tile[hb] * kPPBlock : nullptr;
            }
            tph_exchange<PREC, 2, false, HB, HALF>(frag, lane, t0, acc, f_hi, f_lo, pl, valid);
        }
        }       // !REV
        // ---------------- reverse sweep: g = d sdf / d u_l   ----------------
        bias_h<2, HB>(blob, LY.extra, acc, lane, t0);                  // g_hat(h_8) = row 0 of W_8
#pragma unroll 1
        for (int l = 7; l >= 0; --l) {
            asm volatile("" : "+s"(blob));
            unsigned char *a_hi[HB], *a_lo[HB];
            u16x8 sg[2][HB][2];          // sigma'(z_l) of this wave's tiles: one batch of loads
            const int tn_l = (l == 3 && wave == 3) ? 1 : 2;
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                a_hi[hb] = TRAIN ? blk(st.a_hi, l, hb) : nullptr;
                a_lo[hb] = LO ? blk(st.a_lo, l, hb) : nullptr;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int sh = 0; sh < 2; ++sh)
#ifdef FNEUS_DBG_NO_SIGLOAD             // timing experiments only
                        for (int e = 0; e < 8; ++e) sg[i][hb][sh][e] = (unsigned short)(lane * 257 + e + l);
#else
                        sg[i][hb][sh] = *reinterpret_cast<const u16x8*>(sig(l, hb) + (size_t)(2 * (t0 + i) + sh) * kFragBytes + lane * 16);
#endif
            }
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int sh = 0; sh < 2; ++sh)
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[i][hb][8 * sh + e] *= (float)sg[i][hb][sh][e] * (1.0f / 65535.0f);   // a_l
            if (tn_l == 1) {
                f32x16(&a1)[1][HB] = reinterpret_cast<f32x16(&)[1][HB]>(acc);
                tph_exchange<PREC, 1, true, HB, HALF>(frag, lane, t0, a1, a_hi, a_lo, pl, valid);
            } else {
                tph_exchange<PREC, 2, true, HB, HALF>(frag, lane, t0, acc, a_hi, a_lo, pl, valid);
            }
#ifdef FNEUS_K2_SPF      // experiment (s_prefetch.h): 307-316 us against 284 without -- every LDS wait of the dense phase
                         // (lgkmcnt(N)) then also waits for the 64 scalar loads, i.e. for HBM, at the phase's start
            if constexpr (REV && HB == 2) {
                s_prefetch_done(spf);                     // (the exchange above ended with lgkmcnt(0) + barrier)
                // sigma' of the NEXT layer of the sweep (or of the next group's first one) -> L2 while this layer's MFMAs run:
                // wave w takes 8 KiB of the 2 x 16 KiB
                const int wv = __builtin_amdgcn_readfirstlane(wave);
                const long nt = l > 0 ? tile[wv >> 1] : tile[wv >> 1] + (long)gridDim.x * HB;
                if (nt < tiles) spf = s_prefetch_8k(st.ps + ((size_t)nt * 8 + (l > 0 ? l - 1 : 7)) * kPPBlock + (wv & 1) * 8192);
            }
#endif
            if (l == 0) break;
            if (l == 4) {   // 9 row tiles: 0..6 -> g_hat(h_4), 7..8 -> q_skip (PE part of the skip input; wave hb for half hb)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb)
                    if (wave == hb) {
                        f32x16 qs[2][1];
                        zero_acc(qs[0]);
                        zero_acc(qs[1]);
                        tph_dense<PREC, 16, 9, 7, 2, true, 1, HALF>(blob, LY.L[4].rev_hi, LY.L[4].rev_lo, frag + hb * HALF, qs, lane);
                        f32x16* q = st.qs + ((size_t)tile[hb] * 2) * 64 + lane;
                        q[0] = qs[0][0];
                        q[64] = qs[1][0];
                    }
#pragma unroll
                for (int i = 0; i < 2; ++i) zero_acc(acc[i]);
                if (wave < 3) {
                    tph_dense<PREC, 16, 9, 0, 2, true, HB, HALF>(blob, LY.L[4].rev_hi, LY.L[4].rev_lo, frag, acc, lane, t0);
                } else {
                    f32x16(&a1)[1][HB] = reinterpret_cast<f32x16(&)[1][HB]>(acc);
                    tph_dense<PREC, 16, 9, 0, 1, true, HB, HALF>(blob, LY.L[4].rev_hi, LY.L[4].rev_lo, frag, a1, lane, t0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) zero_acc(acc[i]);
                if (l == 3)
                    tph_dense<PREC, 14, 8, 0, 2, true, HB, HALF>(blob, LY.L[3].rev_hi, LY.L[3].rev_lo, frag, acc, lane, t0);
                else
                    tph_dense<PREC, 16, 8, 0, 2, true, HB, HALF>(blob, LY.L[l].rev_hi, LY.L[l].rev_lo, frag, acc, lane, t0);
            }
        }
        // the 2 row tiles of the 39 PE inputs and normal = J^T q: wave hb for half hb
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
            if (wave == hb) {
                f32x16 qq[2][1];
                zero_acc(qq[0]);
                zero_acc(qq[1]);
                tph_dense<PREC, 16, 2, 0, 2, true, 1, HALF>(blob, LY.L[0].rev_hi, LY.L[0].rev_lo, frag + hb * HALF, qq, lane);
                const f32x16* qsk = st.qs + ((size_t)tile[hb] * 2) * 64 + lane;
                f32x16 q[2];
                q[0] = qq[0][0] + qsk[0];
                q[1] = qq[1][0] + qsk[64];
                float x[3], pe[39], jc[39];       // Jacobian coefficients of the encoding, recomputed
                load_point(src, nc[hb], x);
                posenc<6, true>(x, pe, jc);
                float nrm[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float coef[39];
#pragma unroll
                    for (int f = 0; f < 39; ++f) coef[f] = ((f % 3) == c) ? jc[f] : 0.0f;
                    const float part = acc_dot_partial<2, 39>(q, coef, h);
                    nrm[c] = part + xor32(part);
                }
                if (valid[hb] && lane < 32) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) normal_out[n[hb] * 3 + c] = nrm[c];
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // all four waves finish the group together
    }
}

// ---- K3 ----------------------------------------------------------------------------------------------------
// Backward of (sdf, feature, normal) w.r.t. the SDF-network weights: the two chains of SURVEY.md Appendix A.
//   ascending  (tangent of the reverse sweep): adj_0 = J nbar;  abar_l = W_l adj_l;  adj_{l+1} = s_l * abar_l;
//               coupling c_l = beta (1 - s_l) a_l abar_l   (= softplus'' * g_hat * abar)
//   descending (ordinary backprop):            zbar_8 = [fbar ; sbar];  ubar_l = W_l^T zbar_l;
//               zbar_{l-1} = s_{l-1} * ubar_l + c_{l-1}
// The operands of dW_l = zbar_l^T u_l + a_l^T adj_l leave as fragment planes (fneus_pp.h) for dw_gemm_pp.hip: the B
// fragments every layer converts its accumulators into anyway, stored straight from registers.
// ---- K3, tensor-parallel workgroups --------------------------------------------------------------------------------
// Same maths, buffers and results as sdf_bwd_kernel, organised like sdf_fwd_grad_tp_kernel: wave w owns tiles 2w, 2w+1 of
// every layer, two workgroups per CU.  With the fragment planes the kernel has no store phase of its own any more (each
// wave stores the fragments it has just published, 4 per layer), so what the split buys is the overlap of one
// workgroup's exchange / activation phases with the other's MFMAs.
// qbar = J(x) nbar as B fragments KS0..KS0+2, recomputed where needed
template <int PREC, int KS0>
FN_DEV void qbar_frags_tp(const float (&x)[3], const float (&nb)[3], BFrag<PREC> (&bf)[kMaxKS], int h) {
    float xx[3] = {x[0], x[1], x[2]};
    asm volatile("" : "+v"(xx[0]), "+v"(xx[1]), "+v"(xx[2]));
    float pe[39], jc[39], qb[39];
    posenc<6, true>(xx, pe, jc);
#pragma unroll
    for (int f = 0; f < 39; ++f) qb[f] = jc[f] * nb[f % 3];
    vec_to_bfrag<PREC, 39, 3, KS0>(qb, bf, h);
}

// ---- post-phase operands fetched ONE LAYER AHEAD ----------------------------------------------------------------------
// The activation work behind a layer's MFMAs needs sigma'(z_l) and a second lane-private operand (a_l for the ascending
// chain, the coupling c_l for the descending one) from HBM.  Loaded where they are used, their latency (~2 us under load)
// is exposed 17 times per tile -- 29 % of K3 (measured with the loads compiled out).  A wave's vector-memory operations
// retire in order, so an HBM load issued just ahead of a dense phase would stall that phase's first weight fragment
// instead; the loads are therefore issued right behind the PREVIOUS layer's MFMAs and have that layer's activation work,
// exchange and barriers to land.  LO: the second operand has a lo plane (gradient precision 3).
template <int HB, int TN, bool LO>
struct PostData {
    u16x8 sg[HB][TN][2];
    bf16x8 hi[HB][TN][2];
    bf16x8 lo[HB][TN][LO ? 2 : 1];
};

// PLANE: the second operand is a fragment plane (slot-permuted: a_l) or lane-linear scratch (c_l)
template <int HB, int TN, bool LO, bool PLANE>
FN_DEV void post_fetch(PostData<HB, TN, LO>& d, int T0, const unsigned char* const (&sblk)[HB], const unsigned char* const (&hi)[HB],
                       const unsigned char* const (&lo)[HB], int lane, const PPLane& pl) {
#pragma unroll
    for (int hb = 0; hb < HB; ++hb)
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ks = 2 * (T0 + t) + s;
                d.sg[hb][t][s] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(sblk[hb] + (size_t)ks * kFragBytes + lane * 16));
                const size_t off = (size_t)ks * kFragBytes + (PLANE ? ((ks & 1) ? pl.odd : pl.even) : (unsigned)lane * 16u);
                d.hi[hb][t][s] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(hi[hb] + off));
                if constexpr (LO) d.lo[hb][t][s] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(lo[hb] + off));
            }
}

// ascending chain: adj_{l+1} = s_l * abar_l in place, coupling c_l = beta (1 - s_l) a_l abar_l to the scratch
template <int PREC, int HB, int TN, bool LO>
FN_DEV void asc_apply(f32x16 (&acc)[TN][HB], const PostData<HB, TN, LO>& d, int T0, unsigned char* const (&cb_hi)[HB],
                      unsigned char* const (&cb_lo)[HB], int lane) {
#pragma unroll
    for (int hb = 0; hb < HB; ++hb)
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ks = 2 * (T0 + t) + s;
                bf16x8 chi, clo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sv = (float)d.sg[hb][t][s][e] * (1.0f / 65535.0f);
                    float av = (float)d.hi[hb][t][s][e];
                    if constexpr (LO) av += (float)d.lo[hb][t][s][e];
                    const float abar = acc[t][hb][8 * s + e];
                    const float c = kBeta * (1.0f - sv) * av * abar;      // softplus'' * g_hat * abar  (a = s * g_hat)
                    acc[t][hb][8 * s + e] = sv * abar;
                    if constexpr (PREC == 3) {
                        __bf16 x, y;
                        split_bf16(c, x, y);
                        chi[e] = x;
                        clo[e] = y;
                    } else {
                        chi[e] = (__bf16)c;
                    }
                }
                __builtin_nontemporal_store(chi, reinterpret_cast<bf16x8*>(cb_hi[hb] + (size_t)ks * kFragBytes + lane * 16));
                if constexpr (PREC == 3 && LO)
                    __builtin_nontemporal_store(clo, reinterpret_cast<bf16x8*>(cb_lo[hb] + (size_t)ks * kFragBytes + lane * 16));
            }
}

// descending chain: zbar_l = s_l * hbar_{l+1} + c_l
template <int HB, int TN, bool LO>
FN_DEV void desc_apply(f32x16 (&acc)[TN][HB], const PostData<HB, TN, LO>& d) {
#pragma unroll
    for (int hb = 0; hb < HB; ++hb)
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sv = (float)d.sg[hb][t][s][e] * (1.0f / 65535.0f);
                    float c = (float)d.hi[hb][t][s][e];
                    if constexpr (LO) c += (float)d.lo[hb][t][s][e];
                    acc[t][hb][8 * s + e] = sv * acc[t][hb][8 * s + e] + c;
                }
}

constexpr int kK3Half = 19 * 2 * kFragBytes;      // B fragments of one half: up to 18 k-steps x (hi, lo) + 1 (parking, see below)

#ifndef FNEUS_K3_FETCH
#define FNEUS_K3_FETCH 0          // 0: post-phase operands fetched in ONE batch where they are used; 1: one layer ahead (measured: no gain)
#endif
template <int PREC, bool WLO, int HB, int GP>
__global__ void __launch_bounds__(256, 2) sdf_bwd_tph_kernel(const unsigned char* blob, PointSrc src, long N, SdfStash st,
                                                             SdfBwdBufs bb, const float* __restrict__ d_sdf,
                                                             const float* __restrict__ d_feat,
                                                             const float* __restrict__ d_normal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    constexpr int HALF = kK3Half;
    constexpr bool LO = PREC == 3 && GP == 3;          // lo planes exist (exact-gradient mode)
    constexpr bool AHEAD = FNEUS_K3_FETCH != 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    constexpr auto& LY = kSdfLayout;
    const long tiles = pp_tiles(N);
    const long groups = (N + 32 * HB - 1) / (32 * HB);
    for (long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        asm volatile("" : "+s"(blob));
        long tile[HB], nc[HB];
        bool valid[HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            tile[hb] = grp * HB + hb;
            const long n = tile[hb] * 32 + r;
            valid[hb] = n < N;
            nc[hb] = valid[hb] ? n : N - 1;
        }
        auto blk = [&](unsigned char* base, int slot, int hb) { return base + ((size_t)slot * tiles + tile[hb]) * kPPBlock; };
        auto priv = [&](unsigned char* base, int l, int hb) { return base + ((size_t)tile[hb] * 8 + l) * kPPBlock; };
        PostData<HB, 2, LO> pd;
        auto fetch_asc = [&](int l) {
            const unsigned char *sb[HB], *ah[HB], *al[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                sb[hb] = priv(st.ps, l, hb);
                ah[hb] = blk(st.a_hi, l, hb);
                al[hb] = LO ? blk(st.a_lo, l, hb) : nullptr;
            }
            post_fetch<HB, 2, LO, true>(pd, t0, sb, ah, al, lane, pl);
        };
        auto fetch_desc = [&](int l) {
            const unsigned char *sb[HB], *ch[HB], *cl[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                sb[hb] = priv(st.ps, l, hb);
                ch[hb] = priv(bb.c_hi, l, hb);
                cl[hb] = LO ? priv(bb.c_lo, l, hb) : nullptr;
            }
            post_fetch<HB, 2, LO, false>(pd, t0, sb, ch, cl, lane, pl);
        };
        if constexpr (AHEAD) fetch_asc(0);
        // ---- qbar = J nbar (wave 0): k-steps 0..2 of the first layer, the qbar plane, and a second copy parked in k-steps
        // 16..18 of the region: layer 4 takes it as its k-steps 14..16 (tangent of the skip input) -- re-encoding the point
        // there, with the accumulators live, costs ~120 registers
        if (wave == 0) {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                float x[3], nb[3];
                load_point(src, nc[hb], x);
#pragma unroll
                for (int c = 0; c < 3; ++c) nb[c] = valid[hb] ? d_normal[nc[hb] * 3 + c] : 0.0f;
                BFrag<PREC> qf[kMaxKS];
                qbar_frags_tp<PREC, 0>(x, nb, qf, h);
                frags_to_lds<PREC, 3>(frag + hb * HALF, lane, 0, qf);
                frags_to_lds<PREC, 3>(frag + hb * HALF, lane, 16, qf);
                frags_to_plane<PREC, 3>(qf, 0, bb.qbar_hi + (size_t)tile[hb] * 4 * kFragBytes,
                                        LO ? bb.qbar_lo + (size_t)tile[hb] * 4 * kFragBytes : nullptr, pl, valid[hb]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        f32x16 acc[2][HB];
        // ---- ascending chain: adj_{l+1} = s_l * (W_l adj_l), coupling c_l to the scratch ----
#pragma unroll 1
        for (int l = 0; l <= 7; ++l) {
            asm volatile("" : "+s"(blob));
            unsigned char *cb_hi[HB], *cb_lo[HB], *o_hi[HB], *o_lo[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                cb_hi[hb] = priv(bb.c_hi, l, hb);
                cb_lo[hb] = LO ? priv(bb.c_lo, l, hb) : nullptr;
                o_hi[hb] = blk(bb.adj_hi, l, hb);
                o_lo[hb] = LO ? blk(bb.adj_lo, l, hb) : nullptr;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) zero_acc(acc[i]);
            // layer 3 has 7 output tiles: wave 3 computes tile 6 and a padding tile whose weights are zero fragments (the
            // pack pads every layer to whole tiles), so all waves run the same code
            if (l == 0)
                tph_dense<PREC, 3, 8, 0, 2, WLO, HB, HALF>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, frag, acc, lane, t0);
            else if (l == 3) {
                if (wave < 3) {
                    tph_dense<PREC, 16, 7, 0, 2, WLO, HB, HALF>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, frag, acc, lane, t0);
                } else {
                    f32x16(&a1)[1][HB] = reinterpret_cast<f32x16(&)[1][HB]>(acc);
                    tph_dense<PREC, 16, 7, 0, 1, WLO, HB, HALF>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, frag, a1, lane, t0);
                }
            } else if (l == 4)
                tph_dense<PREC, 17, 8, 0, 2, WLO, HB, HALF>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, frag, acc, lane, t0);
            else
                tph_dense<PREC, 16, 8, 0, 2, WLO, HB, HALF>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, frag, acc, lane, t0);
            if constexpr (!AHEAD) fetch_asc(l);
            asc_apply<PREC, HB, 2, LO>(acc, pd, t0, cb_hi, cb_lo, lane);       // (wave 3, layer 3: its second tile is padding)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (AHEAD) {       // the next post phase's operands: in flight during this exchange and the next MFMAs
                if (l < 7) fetch_asc(l + 1);
                else fetch_desc(7);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (l == 3) {
                if (wave < 3) {
                    tph_exchange<PREC, 2, true, HB, HALF>(frag, lane, t0, acc, o_hi, o_lo, pl, valid);
                } else {
                    f32x16(&a1)[1][HB] = reinterpret_cast<f32x16(&)[1][HB]>(acc);
                    BFrag<PREC> ex[HB * 3];     // tangent of the skip input: k-steps 14..16 of layer 4 = the parked copy of qbar
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) lds_to_frags<PREC, 3>(frag + hb * HALF, lane, 16, &ex[hb * 3]);
                    tph_exchange<PREC, 1, true, HB, HALF>(frag, lane, t0, a1, o_hi, o_lo, pl, valid, ex, 14, 3);
                }
            } else if (l < 7) {
                tph_exchange<PREC, 2, true, HB, HALF>(frag, lane, t0, acc, o_hi, o_lo, pl, valid);
            } else {
                tph_exchange<PREC, 2, false, HB, HALF>(frag, lane, t0, acc, o_hi, o_lo, pl, valid);     // adj_8: plane only
            }
        }
        // ---- descending chain: zbar_8 = [fbar ; sbar], ubar_l = W_l^T zbar_l, zbar_{l-1} = s_{l-1} * ubar_l + c_{l-1} ----
        {
            unsigned char *o_hi[HB], *o_lo[HB];
            BFrag<PREC> sf[HB * 3];     // tile 8 of zbar_8: row 0 (register 0 of lane half 0) = d sdf, everything else zero
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                {
                    f32x16 t2[2];
                    load_f32<2>(t2, d_feat + 32 * t0, 256, nc[hb], h);
                    if (!valid[hb]) zero_acc(t2);
                    acc[0][hb] = t2[0];
                    acc[1][hb] = t2[1];
                }
                o_hi[hb] = blk(bb.zbar_hi, 8, hb);
                o_lo[hb] = LO ? blk(bb.zbar_lo, 8, hb) : nullptr;
                if (wave == 0) {
                    const float sv = (h == 0 && valid[hb]) ? d_sdf[nc[hb]] : 0.0f;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        sf[hb * 3 + i].hi = zero_bf16x8();
                        if constexpr (PREC == 3) sf[hb * 3 + i].lo = zero_bf16x8();
                    }
                    if constexpr (PREC == 3) {
                        __bf16 shi, slo;
                        split_bf16(sv, shi, slo);
                        sf[hb * 3].hi[0] = shi;
                        sf[hb * 3].lo[0] = slo;
                    } else {
                        sf[hb * 3].hi[0] = (__bf16)sv;
                    }
                    frags_to_plane<PREC, 2>(&sf[hb * 3], 0, bb.zsdf_hi + (size_t)tile[hb] * 2 * kFragBytes,
                                            LO ? bb.zsdf_lo + (size_t)tile[hb] * 2 * kFragBytes : nullptr, pl, valid[hb]);
                }
            }
            // k-steps 0..15: the feature tiles, 16, 17: the sdf tile
            tph_exchange<PREC, 2, true, HB, HALF>(frag, lane, t0, acc, o_hi, o_lo, pl, valid, wave == 0 ? sf : nullptr, 16, 2);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) zero_acc(acc[i]);
        tph_dense<PREC, 18, 8, 0, 2, WLO, HB, HALF>(blob, LY.L[8].rev_hi, LY.L[8].rev_lo, frag, acc, lane, t0);
#pragma unroll 1
        for (int l = 7; l >= 0; --l) {
            asm volatile("" : "+s"(blob));
            unsigned char *o_hi[HB], *o_lo[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                o_hi[hb] = blk(bb.zbar_hi, l, hb);
                o_lo[hb] = LO ? blk(bb.zbar_lo, l, hb) : nullptr;
            }
            // here acc = ubar_{l+1} = hbar_{l+1};  zbar_l = s_l * hbar_{l+1} + c_l
            if constexpr (!AHEAD) fetch_desc(l);
            desc_apply<HB, 2, LO>(acc, pd);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (AHEAD) {
                if (l > 0) fetch_desc(l - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (l == 0) {
                tph_exchange<PREC, 2, false, HB, HALF>(frag, lane, t0, acc, o_hi, o_lo, pl, valid);
                break;
            }
            if (l == 3 && wave == 3) {   // zbar_3 has 7 tiles: wave 3 owns tile 6 only
                f32x16(&a1)[1][HB] = reinterpret_cast<f32x16(&)[1][HB]>(acc);
                tph_exchange<PREC, 1, true, HB, HALF>(frag, lane, t0, a1, o_hi, o_lo, pl, valid);
            } else {
                tph_exchange<PREC, 2, true, HB, HALF>(frag, lane, t0, acc, o_hi, o_lo, pl, valid);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) zero_acc(acc[i]);
            if (l == 4) {   // ubar_4 restricted to the h_4 rows: tiles 0..6 of the 9-tile reverse pack
                if (wave < 3) {
                    tph_dense<PREC, 16, 9, 0, 2, WLO, HB, HALF>(blob, LY.L[4].rev_hi, LY.L[4].rev_lo, frag, acc, lane, t0);
                } else {
                    f32x16(&a1)[1][HB] = reinterpret_cast<f32x16(&)[1][HB]>(acc);
                    tph_dense<PREC, 16, 9, 0, 1, WLO, HB, HALF>(blob, LY.L[4].rev_hi, LY.L[4].rev_lo, frag, a1, lane, t0);
                }
            } else if (l == 3) {
                tph_dense<PREC, 14, 8, 0, 2, WLO, HB, HALF>(blob, LY.L[3].rev_hi, LY.L[3].rev_lo, frag, acc, lane, t0);
            } else {
                tph_dense<PREC, 16, 8, 0, 2, WLO, HB, HALF>(blob, LY.L[l].rev_hi, LY.L[l].rev_lo, frag, acc, lane, t0);
            }
        }
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

extern "C" int fneus_sdf_fwd_rays(const void* blob, const float* rays_o, const float* rays_d, const float* t, int m, long n_pts,
                                  const unsigned char* ray_mask, float fill, int32_t* work, float* sdf_out, int prec,
                                  fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    if (!blob || !rays_o || !rays_d || !t || !ray_mask || !work || !sdf_out || m <= 0 || m % 128 != 0 || n_pts % m != 0 ||
        (n_pts + 31) / 32 < 1024 || (prec != 1 && prec != 3)) {
        fneus::set_last_error("fneus_sdf_fwd_rays: rays of m = k x 128 samples, >= 32 768 samples in all, every buffer");
        return -2;
    }
    PointSrc src{nullptr, rays_o, rays_d, t, m};
    return fneus::sdf_fwd_p2_rays(reinterpret_cast<const unsigned char*>(blob), src, n_pts, ray_mask, fill, work, sdf_out, prec, stream);
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
    // Product forms (read at every call so that tests can switch): launches of >= 1024 tiles -- two-pass pipelined layers
    // (sdf_p2_kernels.hip; FNEUS_K1_W8_BIG = 31: 8 waves, the default; 3: 4 waves; 0: the 4-wave kernels below); smaller launches -- one
    // tile per 8-wave workgroup with primed layers (sdf_w8_kernels.hip; FNEUS_K1_W8_SMALL = 2, the default; 0: the kernels below).
    // (Round 6 removed the forms no path selected: 8 waves in lockstep, staggered halves, 64-sample two-pass workgroups, h6 products --
    // DESIGN.md 4.1b / 4.4b keep their measurements.)
    {
        const char* e_big = getenv("FNEUS_K1_W8_BIG");
        const char* e_small = getenv("FNEUS_K1_W8_SMALL");
        const int w8_big = e_big ? atoi(e_big) : FNEUS_K1_W8_BIG_DEFAULT;
        const int w8_small = e_small ? atoi(e_small) : FNEUS_K1_W8_SMALL_DEFAULT;
        if (tiles >= 1024 && (w8_big == 3 || w8_big == 31)) return fneus::sdf_fwd_p2(b, src, n_pts, sdf_out, prec, w8_big == 31 ? 1 : 2, stream);   // two-pass pipelined
        if (tiles < 1024 && w8_small == 2) return fneus::sdf_fwd_w8p(b, src, n_pts, sdf_out, prec, stream);               // primed layers
    }
    // chip-filling launches: 64-sample workgroups (FNEUS_K1_HB=1 keeps one wave per tile)
    static const bool no_hb = getenv("FNEUS_K1_HB") && getenv("FNEUS_K1_HB")[0] == '1';
    if (tiles >= 1024 && !no_hb && (prec == 3 || prec == 1)) {
        const long groups = (n_pts + 63) / 64;
        const dim3 g2((unsigned)(groups < 2048 ? groups : 2048));
        if (prec == 3) {
            static bool done = false;
            if (!done) { fneus::allow_big_lds(sdf_fwd_tph_kernel<3>); done = true; }
            hipLaunchKernelGGL(sdf_fwd_tph_kernel<3>, g2, dim3(256), 2 * fneus::kK2Half, stream, b, src, n_pts, sdf_out);
        } else {
            static bool done = false;
            if (!done) { fneus::allow_big_lds(sdf_fwd_tph_kernel<1>); done = true; }
            hipLaunchKernelGGL(sdf_fwd_tph_kernel<1>, g2, dim3(256), 2 * fneus::kK2Half, stream, b, src, n_pts, sdf_out);
        }
        return fneus::launch_status();
    }
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
    // FNEUS_K2_TPH=1 / 2 (experiment, off; read at every call so that tests can switch it): the HB-generic kernel with 1 / 2
    // tiles per workgroup.  Measured at N = 65 536, parity mode: standalone 686 / 644 us against 691 us for the 32-sample
    // kernel below, but 0.55 ms against 0.50 ms inside the training step -- unlike K3, K2 has hardly any loads to batch (its
    // reverse sweep reads sigma' only), and the serial point encoding of wave 0 lengthens every group's start.
    // Chip-filling launches (the 65 536 samples of a training step): the forward chain in the two-pass pipelined form with
    // the stash written on the way (sdf_p2_train_kernels.hip), then the reverse sweep as a launch of its own on 64-sample
    // workgroups.  FNEUS_K2_P2=0 keeps the fused 32-sample kernel below (read at every call so that tests can switch it).
    const char* p2_env = getenv("FNEUS_K2_P2");
    const int k2p2 = p2_env ? atoi(p2_env) : FNEUS_K2_P2_DEFAULT;
    const bool two_launches = k2p2 && tiles >= 1024 && st.qs && (prec == 3 || prec == 1);
    // feat_out NULL (round 6): the caller takes the feature vector as the stash's hi + lo planes -- written by the two-launch form only
    if (feat_out == nullptr && !(two_launches && train && st.feat_hi && (prec == 1 || st.feat_lo))) {
        fneus::set_last_error("fneus_sdf_fwd_grad: feat_out may be NULL only for training launches of >= 1024 tiles with feature planes (hi + lo in parity mode)");
        return -2;
    }
    if (two_launches) {
        const int gp = (train && st.h_lo != nullptr && prec == 3) ? 3 : 1;
        // FNEUS_K2_CHUNKS (experiment): forward and reverse launches alternate over C chunks of the samples, so that a chunk's
        // sigma' blocks are still in the memory-side cache when its reverse sweep reads them
        // (timing experiments: FNEUS_K2_P2 = 2: the forward launches alone, 3: the reverse sweeps alone)
        const char* ch_env = getenv("FNEUS_K2_CHUNKS");
        const long units = (n_pts + 127) / 128, groups = (n_pts + 63) / 64, cap = 256 * 2 * 4;
        long chunks = ch_env ? atol(ch_env) : FNEUS_K2_CHUNKS_DEFAULT;
        chunks = chunks < 1 ? 1 : (chunks > units / 256 ? (units / 256 > 0 ? units / 256 : 1) : chunks);
        dim3 b2(256);
#define FNEUS_K2REV(P, T, G)                                                                                                  \
    do {                                                                                                                      \
        static bool attr_done = false;                                                                                        \
        if (!attr_done) {                                                                                                     \
            allow_big_lds(sdf_fwd_grad_tph_kernel<P, T, 2, G, true>);                                                         \
            attr_done = true;                                                                                                 \
        }                                                                                                                     \
        hipLaunchKernelGGL((sdf_fwd_grad_tph_kernel<P, T, 2, G, true>), g2, b2, 2 * kK2Half, stream, b, src, n_pts, st,       \
                           sdf_out, feat_out, normal_out, g_begin, g_end);                                                    \
    } while (0)
        for (long c = 0; c < chunks; ++c) {
            const long u_begin = units * c / chunks, u_end = units * (c + 1) / chunks;
            const int rc = k2p2 == 3 ? 0 : fneus::sdf_fwd_stash_p2(b, src, n_pts, st, sdf_out, feat_out, prec, train ? gp : 0, u_begin, u_end, stream);
            if (rc) return rc;
            if (k2p2 == 2) continue;
            const long g_begin = 2 * u_begin, g_end = 2 * u_end < groups ? 2 * u_end : groups;
            if (g_end <= g_begin) continue;
            // reverse sweep: resident-weight 8-wave workgroups (sdf_r8_kernels.hip); FNEUS_K2_REV8=0 keeps the 4-wave kernel
            // below (read at every call so that tests can switch it)
            const char* r8_env = getenv("FNEUS_K2_REV8");
            if (r8_env ? atoi(r8_env) != 0 : FNEUS_K2_REV8_DEFAULT) {
                const int rc8 = fneus::sdf_grad_rev_r8(b, src, n_pts, st, normal_out, prec, train, gp, g_begin, g_end, stream);
                if (rc8) return rc8;
                continue;
            }
            dim3 g2((unsigned)(g_end - g_begin < cap ? g_end - g_begin : cap));
            if (prec == 3 && train && gp == 3) FNEUS_K2REV(3, true, 3);
            else if (prec == 3 && train) FNEUS_K2REV(3, true, 1);
            else if (prec == 3) FNEUS_K2REV(3, false, 1);
            else if (train) FNEUS_K2REV(1, true, 1);
            else FNEUS_K2REV(1, false, 1);
        }
#undef FNEUS_K2REV
        return fneus::launch_status();
    }
    const char* tph_env = getenv("FNEUS_K2_TPH");
    const int hbs = tph_env ? atoi(tph_env) : 0;
    if (hbs == 1 || hbs == 2) {
        if (!st.qs) return -2;
        const long groups = (n_pts + 32 * hbs - 1) / (32 * hbs), cap = 256 * 2 * 4;
        dim3 g2((unsigned)(groups < cap ? groups : cap)), b2(256);
        const int gp = (train && st.h_lo != nullptr) ? 3 : 1;
#define FNEUS_K2H(P, T, H, G)                                                                                                \
    do {                                                                                                                     \
        static bool attr_done = false;                                                                                       \
        if (!attr_done) {                                                                                                    \
            allow_big_lds(sdf_fwd_grad_tph_kernel<P, T, H, G>);                                                              \
            attr_done = true;                                                                                                \
        }                                                                                                                    \
        hipLaunchKernelGGL((sdf_fwd_grad_tph_kernel<P, T, H, G>), g2, b2, H * kK2Half, stream, b, src, n_pts, st, sdf_out,  \
                           feat_out, normal_out);                                                                            \
    } while (0)
#define FNEUS_K2H_HB(P, T, G)                                                                                                \
    do {                                                                                                                     \
        if (hbs == 2) FNEUS_K2H(P, T, 2, G);                                                                                 \
        else FNEUS_K2H(P, T, 1, G);                                                                                          \
    } while (0)
        if (prec == 3 && train && gp == 3) FNEUS_K2H_HB(3, true, 3);
        else if (prec == 3 && train) FNEUS_K2H_HB(3, true, 1);
        else if (prec == 3) FNEUS_K2H_HB(3, false, 1);
        else if (prec == 1 && train) FNEUS_K2H_HB(1, true, 1);
        else if (prec == 1) FNEUS_K2H_HB(1, false, 1);
        else return -2;
#undef FNEUS_K2H_HB
#undef FNEUS_K2H
        return fneus::launch_status();
    }
    // one 4-wave workgroup per 32-sample tile, two workgroups per CU
    {
#ifndef FNEUS_K2_GRID_CAP
#define FNEUS_K2_GRID_CAP (256 * 2 * 4)
#endif
        const long cap = FNEUS_K2_GRID_CAP;
        dim3 g2((unsigned)(tiles < cap ? tiles : cap)), b2(256);
        // FNEUS_K2_LDS_PAD (experiment): extra dynamic LDS so that fewer workgroups share a CU (90000: one per CU)
        const char* pad_s = getenv("FNEUS_K2_LDS_PAD");
        const int lds_pad = pad_s ? atoi(pad_s) : 0;
#define FNEUS_K2TP(P, T)                                                                                              \
    do {                                                                                                              \
        static bool attr_done = false;                                                                                \
        if (!attr_done) {                                                                                             \
            allow_big_lds(sdf_fwd_grad_tp_kernel<P, T>);                                                              \
            attr_done = true;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((sdf_fwd_grad_tp_kernel<P, T>), g2, b2, kTp2LdsTotal + lds_pad, stream, b, src, n_pts, st, sdf_out, \
                           feat_out, normal_out);                                                                     \
    } while (0)
        if (prec == 3 && train) FNEUS_K2TP(3, true);
        else if (prec == 3) FNEUS_K2TP(3, false);
        else if (prec == 1 && train) FNEUS_K2TP(1, true);
        else if (prec == 1) FNEUS_K2TP(1, false);
        else return -2;
#undef FNEUS_K2TP
    }
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
    // 4-wave workgroups, two per CU, carrying HB sample tiles through every layer on one pass over the weights: 64 samples
    // (HB = 2) once the launch fills the chip that way, 32 for small launches; FNEUS_K3_HB=1 / 2 forces (read at every call)
    const char* hb_s = getenv("FNEUS_K3_HB");
    const int hb_env = hb_s ? atoi(hb_s) : 0;
    const long n_tiles32 = (n_pts + 31) / 32;
    const int hbs = (hb_env == 1 || hb_env == 2) ? hb_env : (n_tiles32 >= 1024 ? 2 : 1);
    const long groups = (n_pts + 32 * hbs - 1) / (32 * hbs), cap = 256 * 2 * 4;
    dim3 g2((unsigned)(groups < cap ? groups : cap)), b2(256);
    const char* pad3_s = getenv("FNEUS_K3_LDS_PAD");            // (experiment) as FNEUS_K2_LDS_PAD
    const int lds_pad3 = pad3_s ? atoi(pad3_s) : 0;
    // FNEUS_BWD_WHI=1 (experiment, off): with bf16 gradient planes the two chains take the weights as their bf16 hi part (2
    // MFMAs per product, half the weight stream): K3 0.65 -> 0.50 ms in its 32-sample form, but the weight gradients move
    // from 3.5e-3 to 5.5e-3 of their norm (tests/test_hip_backward.py) and miss the 5e-3 bound of the golden gradient test
    const char* whi_s = getenv("FNEUS_BWD_WHI");
    const bool whi = whi_s != nullptr && atoi(whi_s) != 0;
    const bool exact = bb.adj_lo != nullptr;            // gradient precision 3: lo planes everywhere
    if (exact && (!st.a_lo || !bb.c_lo || !bb.zbar_lo || !bb.qbar_lo || !bb.zsdf_lo)) return -2;
    // Chip-filling launches: resident-weight 8-wave workgroups (sdf_r8_kernels.hip).  FNEUS_K3_R8=0 keeps the 4-wave kernels
    // below (read at every call so that tests can switch it); they also take launches whose planes exceed the 32-bit buffer
    // offsets of the r8 kernel (2 GiB per array: 14 563 sample tiles of 9 x 16 KiB).
    {
        const char* r8_env = getenv("FNEUS_K3_R8");
        const bool r8 = r8_env ? atoi(r8_env) != 0 : FNEUS_K3_R8_DEFAULT;
        const long tiles_pp = 2 * ((n_pts + 63) / 64);
        if (r8 && !whi && hb_env == 0 && n_tiles32 >= 1024 && tiles_pp * 9 * (long)kPPBlock < (1L << 31) && (prec == 3 || prec == 1))
            return fneus::sdf_bwd_r8(b, src, n_pts, st, bb, d_sdf, d_feat, d_normal, prec, exact ? 3 : 1, stream);
    }
    if (d_feat == nullptr) return -2;       // the seed's feature rows as fragments in bufs->zbar_hi slot 8: the resident-weight kernel only
#define FNEUS_K3H(P, W, H, G)                                                                                                \
    do {                                                                                                                     \
        static bool attr_done = false;                                                                                       \
        if (!attr_done) {                                                                                                    \
            allow_big_lds(sdf_bwd_tph_kernel<P, W, H, G>);                                                                   \
            attr_done = true;                                                                                                \
        }                                                                                                                    \
        hipLaunchKernelGGL((sdf_bwd_tph_kernel<P, W, H, G>), g2, b2, H * kK3Half + lds_pad3, stream, b, src, n_pts, st, bb, d_sdf, d_feat, \
                           d_normal);                                                                                        \
    } while (0)
    if (prec == 3 && exact) { if (hbs == 2) FNEUS_K3H(3, true, 2, 3); else FNEUS_K3H(3, true, 1, 3); }
    else if (prec == 3 && !whi) { if (hbs == 2) FNEUS_K3H(3, true, 2, 1); else FNEUS_K3H(3, true, 1, 1); }
    else if (prec == 3) { if (hbs == 2) FNEUS_K3H(3, false, 2, 1); else FNEUS_K3H(3, false, 1, 1); }
    else if (prec == 1) { if (hbs == 2) FNEUS_K3H(1, true, 2, 1); else FNEUS_K3H(1, true, 1, 1); }
    else return -2;
#undef FNEUS_K3H
    return fneus::launch_status();
}
