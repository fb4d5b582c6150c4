// Fused-MLP engine, second generation: 16-sample wave tiles on v_mfma_f32_16x16x32_bf16.
//
// Why: the 32-sample engine (mlp_engine.h) needs ~400 registers per wave in parity mode, i.e. ONE wave per SIMD, and
// the PMC counters show what that costs (profiles/r01_pmc_sq_counters_v0.txt): MFMA busy 16 %, 41 % parked on
// s_waitcnt, 43 % issue stalls -- every latency is exposed.  A 16-sample tile halves the per-wave state
// (accumulators 64 + operands 64..80 registers), so two to three waves share a SIMD and the hardware overlaps one
// wave's activation / memory phases with another's MFMAs.  The price is twice the A-operand traffic per FLOP, paid
// from LDS: the 8 waves of a workgroup (128 samples) share ONE weight stream through an LDS ring.
//
// Layout facts (cdna_hip_programming.md section 3): for mfma_f32_16x16x32_bf16 lane l = (c = l&15, q = l>>4) holds
//   A[row c][k = 8q+j], B[k = 8q+j][col c] (j = 0..7) and C[row 4q+reg][col c] (reg = 0..3).
// As in the first engine the product is computed transposed (A = weights, B = activations, column = sample), so an
// activated accumulator becomes the next layer's B operand with no cross-lane movement: k-step ks takes tiles 2ks and
// 2ks+1, k-slot (ks,q,j) holds feature  phi16 = 32ks + 16(j>>2) + 4q + (j&3); the packer permutes K accordingly.
#pragma once
#include "mlp_engine.h"
#include "fneus_layout16.h"

namespace fneus {
namespace e16 {

constexpr int kMaxKS = 10;             // widest input: colour layer 0 (256 + 64 -> 10 k-steps of 32)
constexpr int kWaves = 8;              // wavefronts per workgroup (128 samples)
constexpr int kMaxStageFrags = 40;     // 20 tiles x (hi, lo) >= colour reverse layer 0 (19 tiles)
constexpr int kSlotBytes = kMaxStageFrags * kFragBytes;   // 40 KiB
constexpr int kRing = 3;
constexpr int kEngineLds = kRing * kSlotBytes;            // 120 KiB: one workgroup per CU

struct Cx {
    const unsigned char* blob;
    unsigned char* smem;
    int lane, wave;
};

FN_DEV f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

FN_DEV constexpr int phi16(int ks, int q, int j) { return 32 * ks + 16 * (j >> 2) + 4 * q + (j & 3); }

template <int PREC, int NT_TOTAL, int T0, int TN, int CH>
FN_DEV void stage_gload(const unsigned char* __restrict__ ghi, const unsigned char* __restrict__ glo, int wave, int s,
                        u32x4 (&r)[CH]) {
    constexpr int F = TN * (PREC == 3 ? 2 : 1);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int fi = c * kWaves + wave;
        if (fi < F) {
            const int t = fi % TN;
            const unsigned char* src = (PREC == 3 && fi >= TN) ? glo : ghi;
            r[c] = *reinterpret_cast<const u32x4*>(src + (size_t)((s * NT_TOTAL + T0 + t) * kFragBytes));
        }
    }
}

template <int PREC, int TN, int CH>
FN_DEV void stage_swrite(unsigned char* __restrict__ sm, int wave, int slot, const u32x4 (&r)[CH]) {
    constexpr int F = TN * (PREC == 3 ? 2 : 1);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int fi = c * kWaves + wave;
        if (fi < F) *reinterpret_cast<u32x4*>(sm + slot * kSlotBytes + fi * kFragBytes) = r[c];
    }
}

// one pipeline iteration: `pre` holds stage s+1 on entry and is refilled with stage s+3
template <int PREC, int KS, int NT_TOTAL, int T0, int TN, int KS0, int CH>
FN_DEV void dense_step(const unsigned char* __restrict__ ghi, const unsigned char* __restrict__ glo,
                       unsigned char* __restrict__ sm, int wave, int s, u32x4 (&pre)[CH],
                       const BFrag<PREC> (&b)[kMaxKS], f32x4 (&acc)[TN]) {
    if (s + 1 < KS) stage_swrite<PREC, TN, CH>(sm, wave, (s + 1) % kRing, pre);
    if (s + 3 < KS) stage_gload<PREC, NT_TOTAL, T0, TN, CH>(ghi, glo, wave, s + 3, pre);
    const unsigned char* slot = sm + (s % kRing) * kSlotBytes;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(slot + i * kFragBytes);
        if constexpr (PREC == 3) {
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(slot + (TN + i) * kFragBytes);
            acc[i] = mfma16(al, b[KS0 + s].hi, acc[i]);
            acc[i] = mfma16(ah, b[KS0 + s].lo, acc[i]);
        }
        acc[i] = mfma16(ah, b[KS0 + s].hi, acc[i]);
    }
    __syncthreads();
}

// acc[i] (tile T0+i) += sum_{ks<KS} A(ks, T0+i) * B(KS0+ks).  One stage = all TN tiles of one k-step (hi plane, then
// lo plane); the 8 waves fetch it once from L2 (whole 1-KiB fragments, coalesced) into an LDS ring slot (lane-linear
// image, conflict-free ds_read_b128).  3 ring slots, two named register sets in flight, ONE barrier per stage.
template <int PREC, int KS, int NT_TOTAL, int T0, int TN, int KS0 = 0>
FN_DEV void dense(const Cx& cx, uint32_t off_hi, uint32_t off_lo, const BFrag<PREC> (&b)[kMaxKS], f32x4 (&acc)[TN]) {
    constexpr int F = TN * (PREC == 3 ? 2 : 1);
    constexpr int CH = (F + kWaves - 1) / kWaves;
    static_assert(F <= kMaxStageFrags, "stage does not fit a ring slot");
    const int lane = cx.lane, wave = cx.wave;
    const unsigned char* __restrict__ ghi = cx.blob + off_hi + lane * 16;
    const unsigned char* __restrict__ glo = cx.blob + off_lo + lane * 16;
    unsigned char* __restrict__ sm = cx.smem + lane * 16;
    u32x4 preA[CH], preB[CH];

    __syncthreads();   // every wave is done with the ring (previous layer)
    stage_gload<PREC, NT_TOTAL, T0, TN, CH>(ghi, glo, wave, 0, preA);
    stage_swrite<PREC, TN, CH>(sm, wave, 0, preA);
    if (KS > 1) stage_gload<PREC, NT_TOTAL, T0, TN, CH>(ghi, glo, wave, 1, preB);
    if (KS > 2) stage_gload<PREC, NT_TOTAL, T0, TN, CH>(ghi, glo, wave, 2, preA);
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < KS; s2 += 2) {
        dense_step<PREC, KS, NT_TOTAL, T0, TN, KS0, CH>(ghi, glo, sm, wave, s2, preB, b, acc);
        if (s2 + 1 < KS) dense_step<PREC, KS, NT_TOTAL, T0, TN, KS0, CH>(ghi, glo, sm, wave, s2 + 1, preA, b, acc);
    }
}

// accumulators <- packed fp32 vector (natural order, 16 floats per tile): lane (c,q) takes floats 4q..4q+3
template <int T0, int TN>
FN_DEV void load_accvec(const Cx& cx, uint32_t off, f32x4 (&acc)[TN]) {
    const f32x4* __restrict__ p = reinterpret_cast<const f32x4*>(cx.blob + off);
    const int q = cx.lane >> 4;
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
            const float lo01 = (q & 1) ? cand[1] : cand[0];
            const float hi23 = (q & 1) ? cand[3] : cand[2];
            const float val = (q & 2) ? hi23 : lo01;
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
            const float lo01 = (q & 1) ? cand[1] : cand[0];
            const float hi23 = (q & 1) ? cand[3] : cand[2];
            s = fmaf((q & 2) ? hi23 : lo01, acc[t][r], s);
        }
    return s;
}

// reduce over the four lane quarters (same sample): every lane gets the sum
FN_DEV float sum_q(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// ---- row-major [N][LD] stash planes (see mlp_engine.h): lane (c,q), tile t holds features 16t+4q..+3 -> 8 bytes ----
template <int PREC, int TN>
FN_DEV void store_stash(const f32x4 (&acc)[TN], __bf16* __restrict__ hi, __bf16* __restrict__ lo, int ld, long n, int q,
                        bool valid, int ncols) {
    if (!valid) return;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int col = 16 * t + 4 * q;
        if (col >= ncols) continue;
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
        *reinterpret_cast<bf16x4*>(hi + n * ld + col) = vh;
        if constexpr (PREC == 3) *reinterpret_cast<bf16x4*>(lo + n * ld + col) = vl;
    }
}

// fp32 value (hi + lo) of this lane's 4 features of tile t
template <int PREC>
FN_DEV f32x4 stash_get4(const __bf16* __restrict__ hi, const __bf16* __restrict__ lo, int ld, long n, int t, int q) {
    const int col = 16 * t + 4 * q;
    const bf16x4 vh = *reinterpret_cast<const bf16x4*>(hi + n * ld + col);
    f32x4 r;
    if constexpr (PREC == 3) {
        const bf16x4 vl = *reinterpret_cast<const bf16x4*>(lo + n * ld + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (float)vh[e] + (float)vl[e];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (float)vh[e];
    }
    return r;
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

}  // namespace e16
}  // namespace fneus
