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
constexpr int kR8MaxKS = 17;

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
// (nx) is requested into W[s] FNEUS_R8_WLAG k-steps behind the MFMAs that read it (the last ones behind the phase).
// Same operands and per-accumulator summation order (lo.hi, hi.lo, hi.hi per k-step) as dense_ldsb_h.
// LMAP (p2_slot): 0: k-step s is slot s; 2: the skip layer's input -- 14 k-steps of h_4, then the encoding's 3 parked in slots
// 16..18 (fields.py:83-84).  voff: of the NEXT layer's tile (a wave may own another tile there).
#ifndef FNEUS_R8_WLAG
#define FNEUS_R8_WLAG 2             // the next layer's stage s is requested behind the MFMAs of k-step s + WLAG - 1
#endif
#ifndef FNEUS_R8_BDIST
#define FNEUS_R8_BDIST 2            // B fragments are requested this many k-steps ahead of their MFMAs
#endif
#ifndef FNEUS_R8_BDIST1
#define FNEUS_R8_BDIST1 2           // the same for the chains on bf16 cotangents (XP 1)
#endif
#ifndef FNEUS_R8_WSPLIT
#define FNEUS_R8_WSPLIT 10          // stages of the next layer requested inside the dense phase; the others behind its barrier
#endif
// how many of the next layer's KSN stages the dense phase itself requests: eight waves x 32 requests of 1 KiB are 4096 cycles of
// the CU's 64 B / clk vector-memory path -- more than the phase's 3400 cycles of MFMAs, and a wave whose request does not get
// into that path stalls its MFMAs behind it (FNEUS_R8_STAMPS: the requesting half's dense phase took 5860 cycles).  The first
// FNEUS_R8_WSPLIT stages fit beside the MFMAs; the rest goes out at the start of the post phase (r8_request_rest), when the
// matrix pipe idles anyway, and is back before the next dense phase reaches those k-steps.
template <int KSN> constexpr int r8_inside = KSN < FNEUS_R8_WSPLIT ? KSN : FNEUS_R8_WSPLIT;

// XP (activation products): PREC = the B fragments are hi + lo in parity mode; 1 with PREC == 3 = the B fragments of the half's
// region are bf16 values ([k-step] x 1 KiB, no lo plane): W.lo . b, W.hi . b -- the cotangent chains of gradient precision 1 / 2,
// whose activations are the values their planes hold.
// SIDE: side(integral_constant<int, s>) is called once per k-step in front of its MFMAs: a slice of the post phase of the half
// BEFORE (another accumulator), so that its vector work runs beside this phase's MFMAs instead of beside an idle matrix pipe
// (color_bwd_r8_kernel, PIPE).
struct R8NoSide {
    template <class S> FN_DEV void operator()(S) const {}
};
template <int PREC, int KS, int KSN, int LMAP = 0, int XP = PREC, class SIDE = R8NoSide>
FN_DEV void r8_dense(R8W& W, const unsigned char* fl, f32x16& acc, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, const R8Layer& nx,
                     const unsigned char* blob, SIDE&& side = SIDE{}) {
    constexpr int NPL = XP == 3 ? 2 : 1;
    // B ring.  A k-step is only 3 MFMAs here (96 cycles of this wave's matrix time), less than an LDS round trip with eight waves
    // reading: requested one k-step ahead (as the 12-MFMA k-steps of the other engines do) every k-step waited for its fragments
    // -- 286 cycles per k-step of a SIMD's two waves instead of 192 (FNEUS_R8_STAMPS).  Distance D, ring of D + 2 buffers:
    // HAZARD (mlp_engine.h dense_ldsb): an LDS load must not land in the operand registers of an MFMA that is still queued; the
    // buffer written at k-step s last fed the MFMAs of k-step s - 2, and the requests are pinned in front of the MFMAs of k-step s.
    // (bf16 regions, XP 1: a k-step is 2 MFMAs = 64 cycles of this wave's matrix time and a fragment 4 registers: a deeper ring)
    constexpr int BD = (XP == 1 && PREC == 3) ? FNEUS_R8_BDIST1 : FNEUS_R8_BDIST;
    constexpr int D = BD < KS ? BD : KS - 1, NB = D + 2;
    constexpr int LAG = FNEUS_R8_WLAG;
    bf16x8 bh[NB], bl[NB];
    static_for<0, D>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        bh[s % NB] = *reinterpret_cast<const bf16x8*>(fl + (p2_slot<LMAP>(s) * NPL) * kFragBytes);
        if constexpr (XP == 3) bl[s % NB] = *reinterpret_cast<const bf16x8*>(fl + (p2_slot<LMAP>(s) * NPL + 1) * kFragBytes);
    });
    static_for<0, KS>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        if constexpr (s + D < KS) {
            bh[(s + D) % NB] = *reinterpret_cast<const bf16x8*>(fl + (p2_slot<LMAP>(s + D) * NPL) * kFragBytes);
            if constexpr (XP == 3) bl[(s + D) % NB] = *reinterpret_cast<const bf16x8*>(fl + (p2_slot<LMAP>(s + D) * NPL + 1) * kFragBytes);
        }
        // The request for stage s' of the next layer goes out LAG k-steps after the MFMAs that read W[s'] were issued: a load
        // whose destination a queued MFMA still has to read stalls at issue until that MFMA has started (write-after-read), and
        // with it everything behind it -- requested right behind their last reader the 32 requests of a phase cost ~76 cycles each
        // (FNEUS_R8_STAMPS: 4355 cycles for the requesting half's dense phase against 1930 for the others).
        if constexpr (s >= LAG && s - LAG < r8_inside<KSN>) {
            constexpr int q = s - LAG;
            const uint32_t f = (uint32_t)(q * nx.nt * 64) * 16u;
            W.hi[q] = p2_wload(rsrc, voff, nx.off_hi + f, blob);
            if constexpr (PREC == 3) W.lo[q] = p2_wload(rsrc, voff, nx.off_lo + f, blob);
        }
        side(S_);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PREC == 3) acc = mfma32(W.lo[s], bh[s % NB], acc);
        if constexpr (XP == 3) acc = mfma32(W.hi[s], bl[s % NB], acc);
        acc = mfma32(W.hi[s], bh[s % NB], acc);
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<(KS > LAG ? KS - LAG : 0), r8_inside<KSN>>([&](auto S_) {
        constexpr int q = decltype(S_)::value;
        const uint32_t f = (uint32_t)(q * nx.nt * 64) * 16u;
        W.hi[q] = p2_wload(rsrc, voff, nx.off_hi + f, blob);
        if constexpr (PREC == 3) W.lo[q] = p2_wload(rsrc, voff, nx.off_lo + f, blob);
    });
    // the caller may reuse the B registers at once (LDS loads): let the last MFMAs read them first
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
}

// stages r8_inside<KSN> .. KSN-1 of the next layer: behind the barrier of the requesting dense phase
template <int PREC, int KSN>
FN_DEV void r8_request_rest(R8W& W, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, const R8Layer& nx, const unsigned char* blob) {
    static_for<r8_inside<KSN>, KSN>([&](auto S_) {
        constexpr int q = decltype(S_)::value;
        const uint32_t f = (uint32_t)(q * nx.nt * 64) * 16u;
        W.hi[q] = p2_wload(rsrc, voff, nx.off_hi + f, blob);
        if constexpr (PREC == 3) W.lo[q] = p2_wload(rsrc, voff, nx.off_lo + f, blob);
    });
}

// the next layer's stages alone (a wave without a tile in the running layer)
template <int PREC, int KSN>
FN_DEV void r8_request(R8W& W, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, const R8Layer& nx, const unsigned char* blob) {
    r8_wload_all<PREC, KSN>(W, rsrc, voff, nx, blob);
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
