// Chain-kernel side of the fragment planes (fneus_pp.h): everything a chain kernel hands to a later kernel leaves the CU
// as the B fragments it holds anyway -- one coalesced 16-byte store per lane and fragment, straight from registers.
//   * GEMM operands (h_l, a_l, adj_l, zbar_l): bf16 hi plane, optional lo plane (exact-gradient mode), slot-permuted;
//   * lane-private data (sigma'(z_l) as 16-bit fixed point, the coupling terms of K3): same 1 KiB units, lane-linear.
#pragma once
#include "mlp_engine.h"
#include "tp_engine.h"
#include "fneus_pp.h"

namespace fneus {

typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
constexpr size_t kPPBlock = 16 * kFragBytes;       // one (tile, layer) block of a 256-wide plane

// sigma'(z) in [0,1] as 16-bit fixed point (abs. error 7.6e-6): written once by the forward chain, read by the reverse
// sweep of K2 (plain load: same kernel) and by both backward chains of K3 (STREAM: non-temporal)
FN_DEV void sig_put8(unsigned char* __restrict__ blk, int ks, int lane, const float (&v)[8]) {
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (unsigned short)__float2uint_rn(fminf(fmaxf(v[e], 0.0f), 1.0f) * 65535.0f);
    *reinterpret_cast<u16x8*>(blk + (size_t)ks * kFragBytes + lane * 16) = o;
}
template <bool STREAM>
FN_DEV void sig_get8(const unsigned char* __restrict__ blk, int ks, int lane, float (&v)[8]) {
    const u16x8* p = reinterpret_cast<const u16x8*>(blk + (size_t)ks * kFragBytes + lane * 16);
#ifdef FNEUS_DBG_NO_POSTLOAD            // timing experiments only
    if (STREAM) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.5f;
        return;
    }
#endif
    const u16x8 o = STREAM ? __builtin_nontemporal_load(p) : *p;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)o[e] * (1.0f / 65535.0f);
}

// softplus in place on tiles t0 .. t0+TN-1 of the layer; sigma' goes to the lane-private block
template <int TN>
FN_DEV void softplus_sig8(f32x16 (&acc)[TN], unsigned char* __restrict__ sblk, int t0, int lane) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float sv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float hh;
                softplus_sig(acc[t][8 * s + e], hh, sv[e]);
                acc[t][8 * s + e] = hh;
            }
            sig_put8(sblk, 2 * (t0 + t) + s, lane, sv);
        }
}

// g *= sigma'(z_l) from the lane-private block (a_l = s_l * g_hat(h_{l+1}))
template <int TN, bool STREAM>
FN_DEV void mul_sig8(f32x16 (&g)[TN], const unsigned char* __restrict__ sblk, int t0, int lane) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float sv[8];
            sig_get8<STREAM>(sblk, 2 * (t0 + t) + s, lane, sv);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[t][8 * s + e] *= sv[e];
        }
}

// hi / lo split of half `s` (registers 8s .. 8s+7) of an accumulator tile = B fragment 2t + s of the next layer
template <int PREC>
FN_DEV void split_half(const f32x16& a, int s, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = s ? a[8 + j] : a[j];
        if constexpr (PREC == 3) {
            __bf16 x, y;
            split_bf16(v, x, y);
            hi[j] = x;
            lo[j] = y;
        } else {
            hi[j] = (__bf16)v;
        }
    }
}

// Tensor-parallel exchange (tp_engine.h) + plane store: this wave's tiles t0 .. t0+TN-1 go to LDS as B fragments for the
// other waves (FRAGS) and, when a plane block is given, to global memory as the same fragments (hi, optional lo).
// Samples beyond N are stored as zeros (the GEMM sums over whole tiles).
template <int PREC, int TN, bool FRAGS>
FN_DEV void tp_exchange_pp(unsigned char* frag, int lane, int t0, const f32x16 (&acc)[TN], unsigned char* __restrict__ blk_hi,
                           unsigned char* __restrict__ blk_lo, const PPLane& pl, bool valid,
                           const BFrag<PREC>* extra = nullptr, int extra_slot = 14, int extra_n = 3) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // previous fragments are consumed
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
            bf16x8 hi, lo;
            split_half<PREC>(acc[i], sh, hi, lo);
            const int ks = 2 * (t0 + i) + sh;
            if constexpr (FRAGS) {
                *reinterpret_cast<bf16x8*>(frag + (ks * NPL) * kFragBytes + lane * 16) = hi;
                if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(frag + (ks * NPL + 1) * kFragBytes + lane * 16) = lo;
            }
            if (blk_hi != nullptr) pp_store(blk_hi, ks, pl, valid ? hi : zero_bf16x8());
            if constexpr (PREC == 3) {
                if (blk_lo != nullptr) pp_store(blk_lo, ks, pl, valid ? lo : zero_bf16x8());
            }
        }
    if (extra) {   // fragments that do not come from an accumulator tile: the skip input of layer 4 (k-steps 14..16,
                   // fields.py:83-84) or the sdf tile of K3's seed (k-steps 16, 17)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < extra_n) {
                *reinterpret_cast<bf16x8*>(frag + ((extra_slot + i) * NPL) * kFragBytes + lane * 16) = extra[i].hi;
                if constexpr (PREC == 3)
                    *reinterpret_cast<bf16x8*>(frag + ((extra_slot + i) * NPL + 1) * kFragBytes + lane * 16) = extra[i].lo;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // all fragments of the layer are in LDS
}

// ---- workgroups of HB sample halves (HB = 2: 64 samples share one pass over the weights; dense_ldsb_h) -----------------
// LDS: half hb keeps its B fragments at frag + hb * HALF_BYTES.  Everything else as tp_exchange_pp, per half.
template <int PREC, int TN, bool FRAGS, int HB, int HALF_BYTES>
FN_DEV void tph_exchange(unsigned char* frag, int lane, int t0, const f32x16 (&acc)[TN][HB], unsigned char* const (&blk_hi)[HB],
                         unsigned char* const (&blk_lo)[HB], const PPLane& pl, const bool (&valid)[HB],
                         const BFrag<PREC>* extra = nullptr /* [HB][3] */, int extra_slot = 14, int extra_n = 3) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // previous fragments are consumed
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
        unsigned char* fh = frag + hb * HALF_BYTES;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                bf16x8 hi, lo;
                split_half<PREC>(acc[i][hb], sh, hi, lo);
                const int ks = 2 * (t0 + i) + sh;
                if constexpr (FRAGS) {
                    *reinterpret_cast<bf16x8*>(fh + (ks * NPL) * kFragBytes + lane * 16) = hi;
                    if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(fh + (ks * NPL + 1) * kFragBytes + lane * 16) = lo;
                }
                if (blk_hi[hb] != nullptr) pp_store(blk_hi[hb], ks, pl, valid[hb] ? hi : zero_bf16x8());
                if constexpr (PREC == 3) {
                    if (blk_lo[hb] != nullptr) pp_store(blk_lo[hb], ks, pl, valid[hb] ? lo : zero_bf16x8());
                }
            }
        if (extra) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < extra_n) {
                    *reinterpret_cast<bf16x8*>(fh + ((extra_slot + i) * NPL) * kFragBytes + lane * 16) = extra[hb * 3 + i].hi;
                    if constexpr (PREC == 3)
                        *reinterpret_cast<bf16x8*>(fh + ((extra_slot + i) * NPL + 1) * kFragBytes + lane * 16) = extra[hb * 3 + i].lo;
                }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // all fragments of the layer are in LDS
}

// B fragments (already split) -> k-steps ks0 .. of the LDS region of one half (no barriers: the caller orders them)
template <int PREC, int NK>
FN_DEV void frags_to_lds(unsigned char* fh, int lane, int ks0, const BFrag<PREC>* bf) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        *reinterpret_cast<bf16x8*>(fh + ((ks0 + k) * NPL) * kFragBytes + lane * 16) = bf[k].hi;
        if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(fh + ((ks0 + k) * NPL + 1) * kFragBytes + lane * 16) = bf[k].lo;
    }
}

template <int PREC, int NK>
FN_DEV void lds_to_frags(const unsigned char* fh, int lane, int ks0, BFrag<PREC>* bf) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        bf[k].hi = *reinterpret_cast<const bf16x8*>(fh + ((ks0 + k) * NPL) * kFragBytes + lane * 16);
        if constexpr (PREC == 3) bf[k].lo = *reinterpret_cast<const bf16x8*>(fh + ((ks0 + k) * NPL + 1) * kFragBytes + lane * 16);
    }
}

// XP: as dense_ldsb_h (1 with PREC 3: bf16 B fragments without a lo plane, weights hi + lo)
template <int PREC, int KS, int NT_TOTAL, int T0, int TN, bool WLO, int HB, int HALF_BYTES, int XP = PREC>
FN_DEV void tph_dense(const unsigned char* __restrict__ blob, uint32_t off_hi, uint32_t off_lo, const unsigned char* frag,
                      f32x16 (&acc)[TN][HB], int lane, int t0_rt = 0) {
    // a stage of HB = 2 carries twice the MFMA time of a 32-sample stage: half the prefetch distance covers the same latency
#ifndef FNEUS_TPH_DEPTH
#define FNEUS_TPH_DEPTH ((FNEUS_TP_DEPTH + 1) / 2)
#endif
    dense_ldsb_h<PREC, KS, NT_TOTAL, T0, TN, (HB >= 2 ? FNEUS_TPH_DEPTH : FNEUS_TP_DEPTH), WLO, HB, HALF_BYTES, XP>(blob, off_hi, off_lo, frag, acc, lane, t0_rt);
}

// one-wave kernels: B fragments ks0 .. ks0+NK-1 (already split) -> plane block
template <int PREC, int NK>
FN_DEV void frags_to_plane(const BFrag<PREC>* bf, int ks0, unsigned char* __restrict__ blk_hi, unsigned char* __restrict__ blk_lo,
                           const PPLane& pl, bool valid) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        pp_store(blk_hi, ks0 + k, pl, valid ? bf[k].hi : zero_bf16x8());
        if constexpr (PREC == 3) {
            if (blk_lo != nullptr) pp_store(blk_lo, ks0 + k, pl, valid ? bf[k].lo : zero_bf16x8());
        }
    }
}

}  // namespace fneus
