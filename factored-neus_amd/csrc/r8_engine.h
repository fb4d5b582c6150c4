// Resident-weight layers on 8-wave workgroups ("r8"): the chain-kernel form of round 4 for the kernels whose chain has HBM
// operands inside it -- the reverse sweep of K2 (sigma' blocks) and both chains of K3 (sigma' twice, a_l, the coupling planes).
//
// What stood in their way: a wave's vector-memory operations retire IN ORDER, so a weight fragment requested
// behind an HBM load (or behind a plane store waiting for its acknowledgement) is not usable before that load has returned --
// ~2 us under load, once per layer and phase, whatever the issue point of the load.  The round-2 kernels (4 waves x 2 output tiles
// x 2 sample tiles, a register ring 2 k-steps deep) therefore sat at 30-35 % MFMA busy, and a deeper ring did not fit 256 registers.
//
// The form that takes the weight stream out of that queue:
//   * workgroup = 8 waves (one per CU, 2 waves per SIMD), 64 samples = two 32-sample halves h0, h1; wave w owns output tile w of
//     every layer, so ONE tile's fragments of a whole layer are 16 k-steps x (hi, lo) x 4 registers = 128 registers: they stay
//     RESIDENT for the layer, serve both halves, and stage s of the NEXT layer is requested into the registers stage s vacates
//     during the second half's MFMAs -- a full half-layer (~1.5 us) before its first use.  No weight fragment is ever waited for
//     behind an operand load issued in the same layer;
//   * the HBM operands of a post phase are requested one layer ahead, right behind the phase that consumed their registers, and
//     the plane stores go out behind the weights they could delay;
//   * a layer is  D0: MFMAs of h0 . barrier . P0: post phase of h0 (operands, hi / lo split, fragments -> LDS, planes -> HBM)
//                 D1: MFMAs of h1 (+ the next layer's weight requests) . barrier . P1: post phase of h1
//     so only ONE accumulator tile is live at a time, and the SIMD partners (waves w, w + 4) are in the same phase (an MFMA-only
//     wave beside a vector-only wave on one SIMD slows both).  Two barriers per layer:
//       P_h overwrites region h (read by every wave's D_h just before the barrier in front of it); D_h of the next layer reads
//       what every wave's P_h wrote before the other half's barrier.
// LDS: region of half h = B fragments of the running layer's input, [k-step][hi, lo] x 1 KiB, 19 k-steps (16 + parking).
#pragma once
#include "p2_engine.h"

namespace fneus {

constexpr int kR8Half = 19 * 2 * kFragBytes;
constexpr int kR8Lds = 2 * kR8Half;
constexpr int kR8MaxKS = 18;

struct R8W {                                    // this wave's tile of one layer: fragment s = k-step s (hi, lo)
    bf16x8 hi[kR8MaxKS], lo[kR8MaxKS];
};
struct R8Layer {                                // where a layer's pack lies in the blob (uniform): fragment order [ks][tile]
    uint32_t off_hi, off_lo;
    int nt;
};

// every stage of a layer at once (the first layer of a group)
template <int PREC, int KS>
FN_DEV void r8_wload_all(R8W& W, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, const R8Layer& ly, const unsigned char* blob) {
    static_for<0, KS>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        const uint32_t f = (uint32_t)(s * ly.nt * 64) * 16u;
        W.hi[s] = p2_wload(rsrc, voff, ly.off_hi + f, blob);
        if constexpr (PREC == 3) W.lo[s] = p2_wload(rsrc, voff, ly.off_lo + f, blob);
    });
}

// acc += sum_{s < KS} W[s] . B[SLOT0 + s] for ONE half (fl = its LDS region + lane * 16).  KSN > 0: stage s of the next layer
// (nx) is requested into W[s] right behind the MFMAs that read it (stages KS .. KSN-1, if any, at the end).
// Same operands and per-accumulator summation order (lo.hi, hi.lo, hi.hi per k-step) as dense_ldsb_h.
template <int PREC, int KS, int KSN, int SLOT0 = 0>
FN_DEV void r8_dense(R8W& W, const unsigned char* fl, f32x16& acc, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, const R8Layer& nx,
                     const unsigned char* blob) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
    bf16x8 bh[3], bl[3];               // HAZARD (mlp_engine.h dense_ldsb): the prefetch of k-step s + 1 is pinned in front of the
    bh[0] = *reinterpret_cast<const bf16x8*>(fl + (SLOT0 * NPL) * kFragBytes);      // MFMAs of k-step s (distinct registers)
    if constexpr (PREC == 3) bl[0] = *reinterpret_cast<const bf16x8*>(fl + (SLOT0 * NPL + 1) * kFragBytes);
    static_for<0, KS>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        if constexpr (s + 1 < KS) {
            bh[(s + 1) % 3] = *reinterpret_cast<const bf16x8*>(fl + ((SLOT0 + s + 1) * NPL) * kFragBytes);
            if constexpr (PREC == 3) bl[(s + 1) % 3] = *reinterpret_cast<const bf16x8*>(fl + ((SLOT0 + s + 1) * NPL + 1) * kFragBytes);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PREC == 3) {
            acc = mfma32(W.lo[s], bh[s % 3], acc);
            acc = mfma32(W.hi[s], bl[s % 3], acc);
        }
        acc = mfma32(W.hi[s], bh[s % 3], acc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s < KSN) {
            const uint32_t f = (uint32_t)(s * nx.nt * 64) * 16u;
            W.hi[s] = p2_wload(rsrc, voff, nx.off_hi + f, blob);
            if constexpr (PREC == 3) W.lo[s] = p2_wload(rsrc, voff, nx.off_lo + f, blob);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<KS, KSN>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        const uint32_t f = (uint32_t)(s * nx.nt * 64) * 16u;
        W.hi[s] = p2_wload(rsrc, voff, nx.off_hi + f, blob);
        if constexpr (PREC == 3) W.lo[s] = p2_wload(rsrc, voff, nx.off_lo + f, blob);
    });
    // the caller may reuse the B registers at once (LDS loads): let the last MFMAs read them first
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
}

FN_DEV void r8_zero(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.0f;
}

// this wave's tile (already mapped in place) -> B fragments 2 w, 2 w + 1 of a half's LDS region, and -- where a block is given --
// the same fragments -> plane block in HBM (fneus_pp.h; samples beyond N as zeros)
template <int PREC>
FN_DEV void r8_publish(const f32x16& acc, unsigned char* region, int lane, int w, unsigned char* __restrict__ blk_hi,
                       unsigned char* __restrict__ blk_lo, const PPLane& pl, bool valid, bool to_lds = true) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
        bf16x8 hi, lo;
        split_half<PREC>(acc, sh, hi, lo);
        const int ks = 2 * w + sh;
        if (to_lds) {
            *reinterpret_cast<bf16x8*>(region + (ks * NPL) * kFragBytes + lane * 16) = hi;
            if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(region + (ks * NPL + 1) * kFragBytes + lane * 16) = lo;
        }
        if (blk_hi != nullptr) pp_store(blk_hi, ks, pl, valid ? hi : zero_bf16x8());
        if constexpr (PREC == 3) {
            if (blk_lo != nullptr) pp_store(blk_lo, ks, pl, valid ? lo : zero_bf16x8());
        }
    }
}

}  // namespace fneus
