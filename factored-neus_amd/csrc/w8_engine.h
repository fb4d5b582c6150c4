// 8-wave workgroups ("w8"): ONE 512-thread workgroup per CU carries HB 32-sample tiles (HB = 4: 128 samples) through a
// network; wave w owns output tile w of every layer, so each weight fragment that enters the CU is used for all HB
// tiles (HB x 3 MFMAs in parity mode).  Why: the tensor-parallel kernels of rounds 1 / 2 (4 waves x 2 tiles, two
// workgroups per CU) are held at 35-46 % MFMA busy by the L2 -> CU weight stream: a CU takes in
// ~20-30 B / clk, two 64-sample workgroups ask for 42 B / clk at the full MFMA rate.  LDS holds the B fragments of 128
// samples (4 x 38 KiB), so 128 samples per pass over the weights is what a CU can do: 21 B / clk.
//
// The per-wave state equals that of the 64-sample kernels (one tile x 4 halves instead of two tiles x 2 halves), and a
// ring stage is ONE fragment pair (8 registers), so the ring can look 3-8 k-steps ahead where the 4-wave kernels had
// registers for 2.
#pragma once
#include "pp_engine.h"

namespace fneus {

constexpr int kW8Half = 19 * 2 * kFragBytes;      // B fragments of one half: 16 k-steps + 3 (skip input / parking) x (hi, lo)
constexpr int kW8Waves = 8;

// weight-prefetch distance in k-steps: a stage is HB x 3 MFMAs (96 cycles each) deep
template <int HB> constexpr int kW8Depth = HB >= 4 ? 3 : (HB == 2 ? 4 : 8);

template <int HB>
FN_DEV void w8_bias(const unsigned char* __restrict__ blob, uint32_t off, f32x16 (&acc)[1][HB], int lane, int tile) {
    const f32x16 FN_GLOBAL* __restrict__ p = reinterpret_cast<const f32x16 FN_GLOBAL*>((gblob_t)blob + off);
    const f32x16 v = p[tile * 2 + (lane >> 5)];
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) acc[0][hb] = v;
}

template <int HB>
FN_DEV void w8_zero(f32x16 (&acc)[1][HB]) {
#pragma unroll
    for (int hb = 0; hb < HB; ++hb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][hb][r] = 0.0f;
}

template <int PREC, int KS, int NT_TOTAL, bool WLO, int HB>
FN_DEV void w8_dense(const unsigned char* __restrict__ blob, uint32_t off_hi, uint32_t off_lo, const unsigned char* frag,
                     f32x16 (&acc)[1][HB], int lane, int tile) {
    dense_ldsb_h<PREC, KS, NT_TOTAL, 0, 1, kW8Depth<HB>, WLO, HB, kW8Half>(blob, off_hi, off_lo, frag, acc, lane, tile);
}

FN_DEV void w8_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// this wave's activated tile -> B fragments 2 tile, 2 tile + 1 of every half in LDS (no barriers: the caller orders them);
// optionally the same fragments -> plane blocks in global memory (fneus_pp.h; samples beyond N as zeros)
template <int PREC, int HB, bool PLANES>
FN_DEV void w8_put_frags(unsigned char* frag, int lane, int tile, const f32x16 (&acc)[1][HB], unsigned char* const (&blk_hi)[HB],
                         unsigned char* const (&blk_lo)[HB], const PPLane& pl, const bool (&valid)[HB], bool to_lds = true) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
        unsigned char* fh = frag + hb * kW8Half;
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
            bf16x8 hi, lo;
            split_half<PREC>(acc[0][hb], sh, hi, lo);
            const int ks = 2 * tile + sh;
            if (to_lds) {
                *reinterpret_cast<bf16x8*>(fh + (ks * NPL) * kFragBytes + lane * 16) = hi;
                if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(fh + (ks * NPL + 1) * kFragBytes + lane * 16) = lo;
            }
            if constexpr (PLANES) {
                if (blk_hi[hb] != nullptr) pp_store(blk_hi[hb], ks, pl, valid[hb] ? hi : zero_bf16x8());
                if constexpr (PREC == 3) {
                    if (blk_lo[hb] != nullptr) pp_store(blk_lo[hb], ks, pl, valid[hb] ? lo : zero_bf16x8());
                }
            }
        }
    }
}

// a wave without a tile of its own in this layer (layer 3 has 7): it moves the parked copy of the encoding (k-steps 16..18)
// to the skip input of layer 4 (k-steps 14..16, fields.py:83-84) between the two barriers of the exchange
template <int PREC, int HB>
FN_DEV void w8_publish_skip(unsigned char* frag, int lane) {
    w8_barrier();
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
        BFrag<PREC> ex[3];
        lds_to_frags<PREC, 3>(frag + hb * kW8Half, lane, 16, ex);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        frags_to_lds<PREC, 3>(frag + hb * kW8Half, lane, 14, ex);
    }
    w8_barrier();
}


// =========================================================================================================================
// Staggered halves ("s8"): the 8 waves of a workgroup form two groups of 4 (waves 0-3 / 4-7: one wave of each group on every
// SIMD), each group carries its OWN HB tiles through the network in the tensor-parallel form (wave w of a group owns output
// tiles 2w, 2w+1), and group 1 runs ONE PHASE BEHIND group 0.  A layer is two phases, separated by ONE workgroup barrier each:
//   D (dense):  weight fragments from L2, B fragments from the group's LDS region, MFMAs only;
//   P (post):   activation, hi / lo split, fragments -> LDS (and planes -> HBM), operand loads: vector / LDS / memory work.
// While one group is in D its SIMD partners are in P: the matrix pipe sees one MFMA stream at a time (instead of two
// competing ones followed by two competing vector phases: 56 % MFMA busy measured for the lockstep form,
// tools/experiments/r03/k1_stamps.py), and the vector work hides behind the partner's MFMAs.  Inside a phase the waves of a
// group do not depend on each other (D reads what the previous P wrote before the barrier; P writes only the wave's own
// k-steps), so the phase barrier is the only synchronisation.
// =========================================================================================================================
// MFMA with the accumulator tile in the AGPR half of the register file (inline assembly: below 256 registers hipcc selects
// the VGPR form for the intrinsic).  The C / D traffic of the matrix pipe then stays off the VGPR ports that the SIMD partner's
// vector instructions need.  The compiler does not see an MFMA here: the callers keep the wait states behind the last one.
FN_DEV void mfma32_agpr(f32x16& c, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int TN, int D>
struct WRing {                       // weight-fragment ring: stage = one k-step of this wave's TN tiles, (hi, lo)
    bf16x8 ah[D + 1][TN], al[D + 1][TN];
};

// request stages 0 .. D-1 of a layer (issued at the END of the previous P phase: they land during the barrier)
template <int PREC, int TN, int D, bool WLO>
FN_DEV void ring_prime(WRing<TN, D>& rg, const unsigned char* __restrict__ blob, uint32_t off_hi, uint32_t off_lo, int ks_total,
                       int nt_total, int lane, int t0) {
    const unsigned voff = (unsigned)(lane + t0 * 64) * 16u;
    const gblob_t bhi = (gblob_t)blob + off_hi, blo = (gblob_t)blob + off_lo;
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < ks_total) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                // (uniform base + uniform fragment offset) + ONE 32-bit lane offset: see the addressing note in dense()
                const size_t f = (size_t)((s * nt_total + i) * 64) * 16;
                rg.ah[s][i] = *reinterpret_cast<const bf16x8 FN_GLOBAL*>(bhi + f + voff);
                if constexpr (PREC == 3 && WLO) rg.al[s][i] = *reinterpret_cast<const bf16x8 FN_GLOBAL*>(blo + f + voff);
            }
        }
}

// the MFMAs of a layer on a primed ring (dense_ldsb_h without its prologue; same operands, same summation order)
template <int PREC, int KS, int NT_TOTAL, int TN, int D, bool WLO, int HB, int HALF_BYTES>
FN_DEV void ring_dense(WRing<TN, D>& rg, const unsigned char* __restrict__ blob, uint32_t off_hi, uint32_t off_lo,
                       const unsigned char* frag /*LDS*/, f32x16 (&acc)[TN][HB], int lane, int t0) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
    const unsigned voff = (unsigned)(lane + t0 * 64) * 16u;
    const gblob_t bhi = (gblob_t)blob + off_hi, blo = (gblob_t)blob + off_lo;
    const unsigned char* fl = frag + lane * 16;
    bf16x8 bh[3][HB], bl[3][HB];          // see the hazard note in dense_ldsb(): the prefetch is pinned in front of the MFMAs
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
        bh[0][hb] = *reinterpret_cast<const bf16x8*>(fl + hb * HALF_BYTES);
        if constexpr (PREC == 3) bl[0][hb] = *reinterpret_cast<const bf16x8*>(fl + hb * HALF_BYTES + kFragBytes);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + D < KS) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const size_t f = (size_t)(((s + D) * NT_TOTAL + i) * 64) * 16;
                rg.ah[(s + D) % (D + 1)][i] = *reinterpret_cast<const bf16x8 FN_GLOBAL*>(bhi + f + voff);
                if constexpr (PREC == 3 && WLO) rg.al[(s + D) % (D + 1)][i] = *reinterpret_cast<const bf16x8 FN_GLOBAL*>(blo + f + voff);
            }
        }
        if (s + 1 < KS) {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                bh[(s + 1) % 3][hb] = *reinterpret_cast<const bf16x8*>(fl + hb * HALF_BYTES + ((s + 1) * NPL) * kFragBytes);
                if constexpr (PREC == 3)
                    bl[(s + 1) % 3][hb] = *reinterpret_cast<const bf16x8*>(fl + hb * HALF_BYTES + ((s + 1) * NPL + 1) * kFragBytes);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef FNEUS_S8_NO_MFMA                 // timing experiments only: operands are fetched, nothing is multiplied
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                asm volatile("" :: "v"(rg.ah[s % (D + 1)][i]), "v"(rg.al[s % (D + 1)][i]), "v"(bh[s % 3][hb]), "v"(bl[s % 3][hb]));
            }
#elif defined(FNEUS_MFMA_AGPR)
        // product-major order: consecutive MFMAs go to different accumulators (the per-accumulator order is unchanged)
        if constexpr (PREC == 3) {
            if constexpr (WLO) {
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) mfma32_agpr(acc[i][hb], rg.al[s % (D + 1)][i], bh[s % 3][hb]);
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) mfma32_agpr(acc[i][hb], rg.ah[s % (D + 1)][i], bl[s % 3][hb]);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) mfma32_agpr(acc[i][hb], rg.ah[s % (D + 1)][i], bh[s % 3][hb]);
#else
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                if constexpr (PREC == 3) {
                    if constexpr (WLO) acc[i][hb] = mfma32(rg.al[s % (D + 1)][i], bh[s % 3][hb], acc[i][hb]);
                    acc[i][hb] = mfma32(rg.ah[s % (D + 1)][i], bl[s % 3][hb], acc[i][hb]);
                }
                acc[i][hb] = mfma32(rg.ah[s % (D + 1)][i], bh[s % 3][hb], acc[i][hb]);
            }
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
}

// P phase: this wave's activated tiles t0 .. t0+TN-1 -> B fragments of the next layer in the group's LDS region (and, with
// PLANES, the same fragments -> plane blocks in HBM; samples beyond N as zeros).  No barriers.
template <int PREC, int TN, int HB, int HALF_BYTES, bool PLANES>
FN_DEV void s8_put_frags(unsigned char* frag, int lane, int t0, int tn, const f32x16 (&acc)[TN][HB], unsigned char* const (&blk_hi)[HB],
                         unsigned char* const (&blk_lo)[HB], const PPLane& pl, const bool (&valid)[HB], bool to_lds = true) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
        unsigned char* fh = frag + hb * HALF_BYTES;
#pragma unroll
        for (int i = 0; i < TN; ++i)
            if (i < tn) {
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    bf16x8 hi, lo;
                    split_half<PREC>(acc[i][hb], sh, hi, lo);
                    const int ks = 2 * (t0 + i) + sh;
                    if (to_lds) {
                        *reinterpret_cast<bf16x8*>(fh + (ks * NPL) * kFragBytes + lane * 16) = hi;
                        if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(fh + (ks * NPL + 1) * kFragBytes + lane * 16) = lo;
                    }
                    if constexpr (PLANES) {
                        if (blk_hi[hb] != nullptr) pp_store(blk_hi[hb], ks, pl, valid[hb] ? hi : zero_bf16x8());
                        if constexpr (PREC == 3) {
                            if (blk_lo[hb] != nullptr) pp_store(blk_lo[hb], ks, pl, valid[hb] ? lo : zero_bf16x8());
                        }
                    }
                }
            }
    }
}

template <int TN, int HB>
FN_DEV void s8_bias(const unsigned char* __restrict__ blob, uint32_t off, f32x16 (&acc)[TN][HB], int lane, int t0) {
    const f32x16 FN_GLOBAL* __restrict__ p = reinterpret_cast<const f32x16 FN_GLOBAL*>((gblob_t)blob + off);
    const int h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const f32x16 v = p[(t0 + i) * 2 + h];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) acc[i][hb] = v;
    }
}

}  // namespace fneus
