// Fused-MLP engine, second generation: 16-sample wave tiles on v_mfma_f32_16x16x32_bf16, two workgroups per CU.
//
// Why (measured, profiles/r01_*): with 32-sample tiles a wave needs ~400 registers in parity mode, i.e. ONE wave per
// SIMD.  Every phase of such a wave is serial -- weight stream (bounded by the ~70 GB/s a CU gets from L2, every wave
// streaming its own copy), MFMAs, activation VALU work, stash traffic -- and the kernels spend >50 % of their time in
// s_waitcnt.  Halving the tile halves the per-wave state (accumulators 64-68 + operands 64-80 registers), so a CU
// holds TWO independent 4-wave workgroups (one wave of each per SIMD): while one workgroup sits in a barrier, an
// activation phase or a stash access, the other one feeds the matrix pipe.  The 4 waves of a workgroup share ONE
// weight stream through an LDS ring, which halves the L2->CU traffic per sample compared to the first engine.
//
// Layout facts (cdna_hip_programming.md section 3): for mfma_f32_16x16x32_bf16 lane l = (c = l&15, q = l>>4) holds
//   A[row c][k = 8q+j], B[k = 8q+j][col c] (j = 0..7) and C[row 4q+reg][col c] (reg = 0..3).
// The product is computed transposed (A = weights, B = activations, column = sample), so an activated accumulator
// becomes the next layer's B operand with no cross-lane movement: k-step ks takes output tiles 2ks and 2ks+1,
// k-slot (ks,q,j) holds feature  phi16 = 32ks + 16(j>>2) + 4q + (j&3); the packer permutes K accordingly.
#pragma once
#include "mlp_engine.h"
#include "fneus_layout16.h"

namespace fneus {
namespace e16 {

constexpr int kMaxKS = 10;             // widest input: colour layer 0 (256 + 64 -> 10 k-steps of 32)
constexpr int kWaves = 4;              // wavefronts per workgroup (64 samples)
constexpr int kMaxStageFrags = 20;     // half a k-step: up to 10 tiles x (hi, lo)
constexpr int kSlotBytes = kMaxStageFrags * kFragBytes;   // 20 KiB
constexpr int kRing = 3;
constexpr int kRingBytes = kRing * kSlotBytes;            // 60 KiB
// per-wave scratch: [16 samples][256 features] bf16 image, 8-byte row padding, hi and lo planes
constexpr int kScrStride = 520;
constexpr int kScrPlane = 16 * kScrStride;                // 8 320
constexpr int kWaveScr = 2 * kScrPlane;                   // 16 640
// The ring is only live inside dense() (barrier at entry and exit); in between the same bytes are the wave scratch.
constexpr int kEngineLds = kWaves * kWaveScr > kRingBytes ? kWaves * kWaveScr : kRingBytes;   // 66 560 -> 2 workgroups / CU

struct Eng {
    const unsigned char* blob;   // packed weights (global, L2 resident)
    unsigned char* lds;          // workgroup LDS region (ring overlaid on the wave scratches)
    int lane, wave;
};

FN_DEV f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
FN_DEV constexpr int phi16(int ks, int q, int j) { return 32 * ks + 16 * (j >> 2) + 4 * q + (j & 3); }

// Workgroup barrier that orders LDS traffic only (no vmcnt drain: global prefetches stay in flight across it).
FN_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Stage s = (k-step s>>1, half s&1): tiles [T0 + half*H0, +hn) of one k-step, hi plane then lo plane.
template <int PREC, int NT_TOTAL, int T0, int TN, int CH>
FN_DEV void stage_gload(gblob_t __restrict__ ghi, gblob_t __restrict__ glo, int wave, int s,
                        u32x4 (&r)[CH]) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
    constexpr int H0 = (TN + 1) / 2;
    const int half = (TN > 1) ? (s & 1) : 0, ks = (TN > 1) ? (s >> 1) : s;
    const int t0 = half ? H0 : 0, hn = half ? TN - H0 : H0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int fi = c * kWaves + wave;
        if (fi < hn * NPL) {
            const int t = t0 + fi % hn;
            gblob_t src = (fi >= hn) ? glo : ghi;
            r[c] = *reinterpret_cast<const u32x4 FN_GLOBAL*>(src + (size_t)((ks * NT_TOTAL + T0 + t) * kFragBytes));
        }
    }
}

template <int PREC, int TN, int CH>
FN_DEV void stage_swrite(unsigned char* __restrict__ sm, int wave, int s, const u32x4 (&r)[CH]) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
    constexpr int H0 = (TN + 1) / 2;
    const int half = (TN > 1) ? (s & 1) : 0;
    const int hn = half ? TN - H0 : H0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int fi = c * kWaves + wave;
        if (fi < hn * NPL) *reinterpret_cast<u32x4*>(sm + (s % kRing) * kSlotBytes + fi * kFragBytes) = r[c];
    }
}

template <int PREC, int KS, int NT_TOTAL, int T0, int TN, int KS0, int CH, int NS>
FN_DEV void dense_step(gblob_t __restrict__ ghi, gblob_t __restrict__ glo,
                       unsigned char* __restrict__ sm, int wave, int s, u32x4 (&pre)[CH],
                       const BFrag<PREC> (&b)[kMaxKS], f32x4 (&acc)[TN]) {
    constexpr int H0 = (TN + 1) / 2;
    // `pre` holds stage s+1 on entry; it is parked in LDS and refilled with stage s+3
    if (s + 1 < NS) stage_swrite<PREC, TN, CH>(sm, wave, s + 1, pre);
    if (s + 3 < NS) stage_gload<PREC, NT_TOTAL, T0, TN, CH>(ghi, glo, wave, s + 3, pre);
    const int half = (TN > 1) ? (s & 1) : 0, ks = (TN > 1) ? (s >> 1) : s;
    const int t0 = half ? H0 : 0, hn = half ? TN - H0 : H0;
    const unsigned char* slot = sm + (s % kRing) * kSlotBytes;
#pragma unroll
    for (int i = 0; i < H0; ++i) {
        if (i < hn) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(slot + i * kFragBytes);
            if constexpr (PREC == 3) {
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(slot + (hn + i) * kFragBytes);
                acc[t0 + i] = mfma16(al, b[KS0 + ks].hi, acc[t0 + i]);
                acc[t0 + i] = mfma16(ah, b[KS0 + ks].lo, acc[t0 + i]);
            }
            acc[t0 + i] = mfma16(ah, b[KS0 + ks].hi, acc[t0 + i]);
        }
    }
    lds_barrier();   // stage s+1 visible to all; everyone is done with slot s%3 (refilled at iteration s+2)
}

// acc[i] (tile T0+i) += sum_{ks<KS} A(ks, T0+i) * B(KS0+ks).  The 4 waves fetch each half k-step once from L2 (whole
// 1-KiB fragments, coalesced) into an LDS ring slot (lane-linear image: conflict-free ds_read_b128); 3 slots, two
// named register sets in flight from global, one LDS-only barrier per stage.
template <int PREC, int KS, int NT_TOTAL, int T0, int TN, int KS0 = 0>
FN_DEV void dense(const Eng& eg, uint32_t off_hi, uint32_t off_lo, const BFrag<PREC> (&b)[kMaxKS], f32x4 (&acc)[TN]) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
    constexpr int H0 = (TN + 1) / 2;
    constexpr int CH = (H0 * NPL + kWaves - 1) / kWaves;
    constexpr int NS = (TN > 1) ? 2 * KS : KS;
    static_assert(H0 * NPL <= kMaxStageFrags, "stage does not fit a ring slot");
    const int lane = eg.lane, wave = eg.wave;
    gblob_t __restrict__ ghi = (gblob_t)eg.blob + off_hi + lane * 16;
    gblob_t __restrict__ glo = (gblob_t)eg.blob + off_lo + lane * 16;
    unsigned char* __restrict__ sm = eg.lds + lane * 16;
    u32x4 preA[CH], preB[CH];

    stage_gload<PREC, NT_TOTAL, T0, TN, CH>(ghi, glo, wave, 0, preA);
    if (NS > 1) stage_gload<PREC, NT_TOTAL, T0, TN, CH>(ghi, glo, wave, 1, preB);
    lds_barrier();   // every wave is done with the LDS region (scratch users of the previous phase)
    stage_swrite<PREC, TN, CH>(sm, wave, 0, preA);
    if (NS > 2) stage_gload<PREC, NT_TOTAL, T0, TN, CH>(ghi, glo, wave, 2, preA);
    lds_barrier();
#pragma unroll
    for (int s2 = 0; s2 < NS; s2 += 2) {
        dense_step<PREC, KS, NT_TOTAL, T0, TN, KS0, CH, NS>(ghi, glo, sm, wave, s2, preB, b, acc);
        if (s2 + 1 < NS) dense_step<PREC, KS, NT_TOTAL, T0, TN, KS0, CH, NS>(ghi, glo, sm, wave, s2 + 1, preA, b, acc);
    }
}

// accumulators <- packed fp32 vector (natural order, 16 floats per tile): lane (c,q) takes floats 4q..4q+3
template <int T0, int TN>
FN_DEV void load_accvec(const Eng& eg, uint32_t off, f32x4 (&acc)[TN]) {
    const f32x4 FN_GLOBAL* __restrict__ p = reinterpret_cast<const f32x4 FN_GLOBAL*>((gblob_t)eg.blob + off);
    const int q = eg.lane >> 4;
#pragma unroll
    for (int i = 0; i < TN; ++i) acc[i] = p[(T0 + i) * 4 + q];
}

template <int TN>
FN_DEV void zero_acc(f32x4 (&acc)[TN]) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.0f;
}

// accumulator tiles 0..TN-1 -> B fragments KS0.. ((TN+1)/2 k-steps; an odd last tile is zero padded)
template <int PREC, int TN, int KS0 = 0>
FN_DEV void acc_to_bfrag(const f32x4 (&acc)[TN], BFrag<PREC> (&b)[kMaxKS]) {
#pragma unroll
    for (int ks = 0; ks < (TN + 1) / 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = 2 * ks + (j >> 2);
            const float v = (t < TN) ? acc[t < TN ? t : 0][j & 3] : 0.0f;
            if constexpr (PREC == 3) {
                __bf16 hi, lo;
                split_bf16(v, hi, lo);
                b[KS0 + ks].hi[j] = hi;
                b[KS0 + ks].lo[j] = lo;
            } else {
                b[KS0 + ks].hi[j] = (__bf16)v;
            }
        }
}

FN_DEV float sel4(const float (&c)[4], int q) {
    const float lo01 = (q & 1) ? c[1] : c[0];
    const float hi23 = (q & 1) ? c[3] : c[2];
    return (q & 2) ? hi23 : lo01;
}

// vector v[NF] (feature order) -> B fragments KS0.. (KS k-steps of 32, zero padded); lane quarter q selects
template <int PREC, int NF, int KS, int KS0>
FN_DEV void vec_to_bfrag(const float (&v)[NF], BFrag<PREC> (&b)[kMaxKS], int q) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float cand[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int f = phi16(ks, qq, j);
                cand[qq] = f < NF ? v[f < NF ? f : 0] : 0.0f;
            }
            const float val = sel4(cand, q);
            if constexpr (PREC == 3) {
                __bf16 hi, lo;
                split_bf16(val, hi, lo);
                b[KS0 + ks].hi[j] = hi;
                b[KS0 + ks].lo[j] = lo;
            } else {
                b[KS0 + ks].hi[j] = (__bf16)val;
            }
        }
}

// sum_f coef[f] * x_f over the features this lane holds (feature 16t + 4q + reg); caller reduces over q
template <int TN, int NF>
FN_DEV float acc_dot_partial(const f32x4 (&acc)[TN], const float (&coef)[NF], int q) {
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float cand[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int f = 16 * t + 4 * qq + r;
                cand[qq] = f < NF ? coef[f < NF ? f : 0] : 0.0f;
            }
            s = fmaf(sel4(cand, q), acc[t][r], s);
        }
    return s;
}

// reduce over the four lane quarters (same sample): every lane gets the sum
FN_DEV float sum_q(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// value of feature IDX of an accumulator-layout vector, valid in all four lanes of the sample
template <int TN, int IDX>
FN_DEV float acc_extract(const f32x4 (&acc)[TN], int q) {
    constexpr int t = IDX / 16, qq = (IDX % 16) / 4, r = IDX % 4;
    return sum_q(q == qq ? acc[t][r] : 0.0f);
}

// ---- row-major [N][LD] bf16 stash planes through the wave's LDS image (whole 512-byte rows leave the CU) ----------
template <int PREC, int TN>
FN_DEV void store_stash(unsigned char* __restrict__ wscr, int lane, const f32x4 (&acc)[TN], __bf16* __restrict__ hi,
                        __bf16* __restrict__ lo, int ld, long n0, long N, int ncols) {
    const int c = lane & 15, q = lane >> 4;
    constexpr int NPL = PREC == 3 ? 2 : 1;
    lds_fence();
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        bf16x4 vh, vl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (PREC == 3) {
                __bf16 a, b2;
                split_bf16(acc[t][e], a, b2);
                vh[e] = a;
                vl[e] = b2;
            } else {
                vh[e] = (__bf16)acc[t][e];
            }
        }
        unsigned char* dst = wscr + c * kScrStride + (16 * t + 4 * q) * 2;
        *reinterpret_cast<bf16x4*>(dst) = vh;
        if constexpr (PREC == 3) *reinterpret_cast<bf16x4*>(dst + kScrPlane) = vl;
    }
    lds_fence();
    const int P = ncols >> 2;            // 8-byte pieces per row
    const int total = 16 * P;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
        __bf16* __restrict__ plane = pl ? lo : hi;
        for (int idx = lane; idx < total; idx += 64) {
            const int row = idx / P, pc = idx - row * P;
            if (n0 + row < N)
                *reinterpret_cast<uint2*>(plane + (n0 + row) * ld + pc * 4) =
                    *reinterpret_cast<const uint2*>(wscr + pl * kScrPlane + row * kScrStride + pc * 8);
        }
    }
}

template <int TN>
FN_DEV void store_f32(const f32x4 (&acc)[TN], float* __restrict__ dst, int ld, long n, int q, bool valid) {
    if (!valid) return;
#pragma unroll
    for (int t = 0; t < TN; ++t) *reinterpret_cast<f32x4*>(dst + n * ld + 16 * t + 4 * q) = acc[t];
}

template <int TN>
FN_DEV void load_f32(f32x4 (&acc)[TN], const float* __restrict__ src, int ld, long n, int q) {
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = *reinterpret_cast<const f32x4*>(src + n * ld + 16 * t + 4 * q);
}

template <int TN>
FN_DEV void softplus_inplace(f32x4 (&acc)[TN]) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = softplus100(acc[t][r]);
}

// lane-private planes ([tile][layer][t][lane] x 4 values, see mlp_engine.h): block sizes for 16-row tiles
constexpr size_t kSigBlock = 16 * 64 * sizeof(u16x4);                 // 8 KiB per (tile, layer)
template <int PREC>
FN_DEV constexpr size_t priv_block() { return 16 * 64 * sizeof(typename PrivT<PREC>::v4); }

// softplus in place; sigma'(z) goes to the lane-private block of this (tile, layer)
template <int PREC, int TN>
FN_DEV void softplus_ps(f32x4 (&acc)[TN], unsigned char* __restrict__ ps, int lane) {
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        float sv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float hh;
            softplus_sig(acc[t][e], hh, sv[e]);
            acc[t][e] = hh;
        }
        sig_put(ps, t, lane, sv);
    }
}

// PE-like 39/33-vector held as B fragments -> [N][48] stash rows (columns phi16 < 48)
template <int PREC>
FN_DEV void store_side48(const BFrag<PREC> (&b)[kMaxKS], int ks0, __bf16* __restrict__ hi, __bf16* __restrict__ lo, long n,
                         int q, bool valid) {
    if (!valid) return;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int col = 32 * ks + 16 * g + 4 * q;
            if (col < 48) {
                bf16x4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vh[e] = b[ks0 + ks].hi[4 * g + e];
                    if constexpr (PREC == 3) vl[e] = b[ks0 + ks].lo[4 * g + e];
                }
                *reinterpret_cast<bf16x4*>(hi + n * 48 + col) = vh;
                if constexpr (PREC == 3) *reinterpret_cast<bf16x4*>(lo + n * 48 + col) = vl;
            }
        }
}

}  // namespace e16
}  // namespace fneus
