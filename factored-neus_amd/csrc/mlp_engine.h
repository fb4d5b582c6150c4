// Wave-local fused-MLP building blocks (see fneus_common.h for the data-flow convention).
#pragma once
#include "fneus_common.h"
#include "fneus_layout.h"

namespace fneus {

constexpr int kMaxKS = 19;   // widest layer input on the path: colour layer 0 (289 -> 19 k-steps)

// acc[i] (tile T0+i) += sum_{ks<KS} A(ks, T0+i) * B(KS0 + ks)
// A fragments: blob + off_{hi,lo} + ((ks*NT_TOTAL + t)*64 + lane)*16 bytes, streamed from L2 through a small
// register ring: work is cut into stages of GT tiles of one k-step; stage s+D is requested before stage s is
// multiplied, and a scheduling barrier per stage stops hipcc from hoisting every load of the layer (which spills).
// DEPTH = prefetch distance in stages (0: the default of the precision mode).  The chain-only kernel K1 is best at 2
// (parity) / 4 (bf16); the kernels that interleave stash traffic with the chain (K2, K3) gain 8-12 % from 4 / 8 even
// though the deeper register ring costs them a few more spills (measured, tools/experiments/README.md).
// WLO = false (parity mode only): the lo halves of the WEIGHTS are neither streamed nor multiplied -- two MFMAs per
// product, w_hi * (x_hi + x_lo): a 2^-9 relative perturbation of every weight.  Used by the backward chains when the
// weight gradients are taken from bf16 planes anyway (gradient precision 1); never on a forward output.
template <int PREC, int KS, int NT_TOTAL, int T0, int TN, int KS0 = 0, int DEPTH = 0, bool WLO = true>
FN_DEV void dense(const unsigned char* __restrict__ blob, uint32_t off_hi, uint32_t off_lo,
                  const BFrag<PREC> (&b)[kMaxKS], f32x16 (&acc)[TN], int lane, int t0_rt = 0) {
    constexpr int GT = TN < 4 ? TN : 4;                 // tiles per stage
    constexpr int NG = (TN + GT - 1) / GT;              // stages per k-step
    constexpr int NS = KS * NG;                         // stages
#ifndef FNEUS_PREFETCH_X3
#define FNEUS_PREFETCH_X3 2
#define FNEUS_PREFETCH_X1 4
#endif
    constexpr int D = DEPTH > 0 ? DEPTH : (PREC == 3 ? FNEUS_PREFETCH_X3 : FNEUS_PREFETCH_X1);   // prefetch distance (stages)
    // Addressing: UNIFORM 64-bit base (blob + plane offset + fragment offset: scalar registers) + ONE 32-bit per-lane
    // byte offset.  Written as a per-lane 64-bit pointer plus constants, every fragment beyond the 4 KiB immediate range
    // gets its own 64-bit vector address, and hipcc computes dozens of them ahead of the layer loops (spills).
    // t0_rt: additional (run-time) first tile, for kernels whose waves own different output tiles.
    const unsigned voff = (unsigned)(lane + t0_rt * 64) * 16u;
    const gblob_t bhi = (gblob_t)blob + off_hi, blo = (gblob_t)blob + off_lo;
    auto whi = [&](int f) { return *reinterpret_cast<const bf16x8 FN_GLOBAL*>(bhi + (size_t)f * 16 + voff); };
    auto wlo = [&](int f) { return *reinterpret_cast<const bf16x8 FN_GLOBAL*>(blo + (size_t)f * 16 + voff); };
    bf16x8 ah[D + 1][GT], al[D + 1][GT];
#pragma unroll
    for (int s = 0; s < D; ++s) {
        if (s < NS) {
#pragma unroll
            for (int i = 0; i < GT; ++i) {
                const int t = (s % NG) * GT + i;
                if (t < TN) {
                    const int f = ((s / NG) * NT_TOTAL + T0 + t) * 64;
                    ah[s % (D + 1)][i] = whi(f);
                    if constexpr (PREC == 3 && WLO) al[s % (D + 1)][i] = wlo(f);
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + D < NS) {
#pragma unroll
            for (int i = 0; i < GT; ++i) {
                const int t = ((s + D) % NG) * GT + i;
                if (t < TN) {
                    const int f = (((s + D) / NG) * NT_TOTAL + T0 + t) * 64;
                    ah[(s + D) % (D + 1)][i] = whi(f);
                    if constexpr (PREC == 3 && WLO) al[(s + D) % (D + 1)][i] = wlo(f);
                }
            }
        }
        const int ks = s / NG;
#pragma unroll
        for (int i = 0; i < GT; ++i) {
            const int t = (s % NG) * GT + i;
            if (t < TN) {
                if constexpr (PREC == 3) {
                    if constexpr (WLO) acc[t] = mfma32(al[s % (D + 1)][i], b[KS0 + ks].hi, acc[t]);
                    acc[t] = mfma32(ah[s % (D + 1)][i], b[KS0 + ks].lo, acc[t]);
                }
                acc[t] = mfma32(ah[s % (D + 1)][i], b[KS0 + ks].hi, acc[t]);
            }
        }
#ifdef FNEUS_WAVE_SYNC_STAGES
        // keep the 4 waves of a workgroup within a few hundred cycles of each other so that their identical weight
        // loads coalesce in the CU's vector L1 instead of each going to L2 (raw barrier: no memory waits)
        if ((s % FNEUS_WAVE_SYNC_STAGES) == FNEUS_WAVE_SYNC_STAGES - 1) __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
}

// dense() for tensor-parallel workgroups in parity mode: the B fragments (activations of the whole tile, produced by all
// waves of the workgroup) stay in LDS -- frag + (ks * NPL + plane) * 1 KiB + lane * 16 -- and are fetched one k-step
// ahead of their MFMAs instead of being held in 8 registers per k-step (136 registers for a 17-k-step layer: with them
// a parity-mode wave does not fit the 256 registers that two workgroups per CU leave it).
template <int PREC, int KS, int NT_TOTAL, int T0, int TN, int DEPTH = 0, bool WLO = true>
FN_DEV void dense_ldsb(const unsigned char* __restrict__ blob, uint32_t off_hi, uint32_t off_lo,
                       const unsigned char* frag /*LDS, written by other helpers: no restrict*/, f32x16 (&acc)[TN], int lane,
                       int t0_rt = 0) {
    static_assert(TN <= 4, "one stage per k-step");
    constexpr int NPL = PREC == 3 ? 2 : 1;
    constexpr int D = DEPTH > 0 ? DEPTH : (PREC == 3 ? FNEUS_PREFETCH_X3 : FNEUS_PREFETCH_X1);
    const unsigned voff = (unsigned)(lane + t0_rt * 64) * 16u;
    const gblob_t bhi = (gblob_t)blob + off_hi, blo = (gblob_t)blob + off_lo;
    auto whi = [&](int f) { return *reinterpret_cast<const bf16x8 FN_GLOBAL*>(bhi + (size_t)f * 16 + voff); };
    auto wlo = [&](int f) { return *reinterpret_cast<const bf16x8 FN_GLOBAL*>(blo + (size_t)f * 16 + voff); };
    const unsigned char* fl = frag + lane * 16;
    bf16x8 ah[D + 1][TN], al[D + 1][TN];
    // HAZARD (measured, tools/dbg_race.py): an LDS load must not be issued into registers that an MFMA issued just
    // before it still has as a source operand -- neither the hardware nor hipcc guards this write-after-read, and the
    // load's data can land (one 16-lane quarter at a time) before the queued MFMA has read its B operand.  Seen as 16
    // consecutive samples with a wrong result a few times per 65 536-sample launch.  Source-level buffering alone does
    // not help: the register allocator reuses an operand's registers as soon as its last MFMA has been *issued*.  So the
    // prefetch of k-step s+1 is pinned (sched_barrier) in FRONT of the MFMAs of k-step s: its destination is then live
    // together with the current operands and therefore physically distinct from them.
    bf16x8 bh[3], bl[3];
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < KS) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int f = (s * NT_TOTAL + T0 + i) * 64;
                ah[s % (D + 1)][i] = whi(f);
                if constexpr (PREC == 3 && WLO) al[s % (D + 1)][i] = wlo(f);
            }
        }
    bh[0] = *reinterpret_cast<const bf16x8*>(fl);
    if constexpr (PREC == 3) bl[0] = *reinterpret_cast<const bf16x8*>(fl + kFragBytes);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + D < KS) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int f = ((s + D) * NT_TOTAL + T0 + i) * 64;
                ah[(s + D) % (D + 1)][i] = whi(f);
                if constexpr (PREC == 3 && WLO) al[(s + D) % (D + 1)][i] = wlo(f);
            }
        }
        if (s + 1 < KS) {
            bh[(s + 1) % 3] = *reinterpret_cast<const bf16x8*>(fl + ((s + 1) * NPL) * kFragBytes);
            if constexpr (PREC == 3) bl[(s + 1) % 3] = *reinterpret_cast<const bf16x8*>(fl + ((s + 1) * NPL + 1) * kFragBytes);
        }
        // pin the prefetch in front of this stage's MFMAs: its destination registers are then live together with the
        // current operands, i.e. physically distinct from them (see the note on the B buffers above)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            if constexpr (PREC == 3) {
                if constexpr (WLO) acc[i] = mfma32(al[s % (D + 1)][i], bh[s % 3], acc[i]);
                acc[i] = mfma32(ah[s % (D + 1)][i], bl[s % 3], acc[i]);
            }
            acc[i] = mfma32(ah[s % (D + 1)][i], bh[s % 3], acc[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // the caller is free to reuse the operand registers at once (e.g. for LDS loads): let the last MFMAs read them first
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
}

// dense_ldsb() for HB sample halves that share ONE pass over the weight fragments (HB = 2: a 64-sample workgroup; half
// the weight stream, half the barriers and twice the MFMA work per k-step and wave).  The B fragments of half hb live at
// frag + hb * HALF_BYTES in the same [k-step][plane] order.  Accumulators are [tile][half]: the first row is a valid
// one-tile view (a wave that owns a single tile of a short layer).
// XP: as r8_dense (r8_engine.h) -- 1 with PREC == 3: the regions hold bf16 B fragments (no lo plane), the weights stay hi + lo
template <int PREC, int KS, int NT_TOTAL, int T0, int TN, int DEPTH, bool WLO, int HB, int HALF_BYTES, int XP = PREC>
FN_DEV void dense_ldsb_h(const unsigned char* __restrict__ blob, uint32_t off_hi, uint32_t off_lo,
                         const unsigned char* frag /*LDS*/, f32x16 (&acc)[TN][HB], int lane, int t0_rt = 0) {
    static_assert(TN <= 4, "one stage per k-step");
    constexpr int NPL = XP == 3 ? 2 : 1;
    constexpr int D = DEPTH > 0 ? DEPTH : (PREC == 3 ? FNEUS_PREFETCH_X3 : FNEUS_PREFETCH_X1);
    const unsigned voff = (unsigned)(lane + t0_rt * 64) * 16u;
    const gblob_t bhi = (gblob_t)blob + off_hi, blo = (gblob_t)blob + off_lo;
    auto whi = [&](int f) { return *reinterpret_cast<const bf16x8 FN_GLOBAL*>(bhi + (size_t)f * 16 + voff); };
    auto wlo = [&](int f) { return *reinterpret_cast<const bf16x8 FN_GLOBAL*>(blo + (size_t)f * 16 + voff); };
#ifdef FNEUS_DBG_NO_WEIGHTS             // timing experiments only: no weight stream from L2
    bf16x8 wconst;
    for (int e = 0; e < 8; ++e) wconst[e] = (__bf16)(float)(lane + e);
    auto whi2 = [&](int f) { bf16x8 t = wconst; asm volatile("" : "+v"(t)); return t; };
#define whi whi2
#define wlo whi2
#endif
    const unsigned char* fl = frag + lane * 16;
    bf16x8 ah[D + 1][TN], al[D + 1][TN];
    bf16x8 bh[3][HB], bl[3][HB];          // see the hazard note in dense_ldsb(): the prefetch is pinned in front of the MFMAs
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < KS) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int f = (s * NT_TOTAL + T0 + i) * 64;
                ah[s % (D + 1)][i] = whi(f);
                if constexpr (PREC == 3 && WLO) al[s % (D + 1)][i] = wlo(f);
            }
        }
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
        bh[0][hb] = *reinterpret_cast<const bf16x8*>(fl + hb * HALF_BYTES);
        if constexpr (XP == 3) bl[0][hb] = *reinterpret_cast<const bf16x8*>(fl + hb * HALF_BYTES + kFragBytes);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + D < KS) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int f = ((s + D) * NT_TOTAL + T0 + i) * 64;
                ah[(s + D) % (D + 1)][i] = whi(f);
                if constexpr (PREC == 3 && WLO) al[(s + D) % (D + 1)][i] = wlo(f);
            }
        }
        if (s + 1 < KS) {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                bh[(s + 1) % 3][hb] = *reinterpret_cast<const bf16x8*>(fl + hb * HALF_BYTES + ((s + 1) * NPL) * kFragBytes);
                if constexpr (XP == 3)
                    bl[(s + 1) % 3][hb] = *reinterpret_cast<const bf16x8*>(fl + hb * HALF_BYTES + ((s + 1) * NPL + 1) * kFragBytes);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                if constexpr (PREC == 3 && WLO) acc[i][hb] = mfma32(al[s % (D + 1)][i], bh[s % 3][hb], acc[i][hb]);
                if constexpr (XP == 3) acc[i][hb] = mfma32(ah[s % (D + 1)][i], bl[s % 3][hb], acc[i][hb]);
                acc[i][hb] = mfma32(ah[s % (D + 1)][i], bh[s % 3][hb], acc[i][hb]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#ifdef FNEUS_DBG_NO_WEIGHTS
#undef whi
#undef wlo
#endif
}

// accumulators <- packed fp32 vector in accumulator layout ([t][h][16])
template <int NT_TOTAL, int T0, int TN>
FN_DEV void load_accvec(const unsigned char* __restrict__ blob, uint32_t off, f32x16 (&acc)[TN], int lane, int t0_rt = 0) {
    const f32x16 FN_GLOBAL* __restrict__ p = reinterpret_cast<const f32x16 FN_GLOBAL*>((gblob_t)blob + off);
    const int h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TN; ++i) acc[i] = p[(T0 + t0_rt + i) * 2 + h];
}

template <int TN>
FN_DEV void zero_acc(f32x16 (&acc)[TN]) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
}

// accumulator tiles 0..TN-1 (after an elementwise map already applied in place) -> B fragments KS0.. (2 per tile)
template <int PREC, int TN, int KS0 = 0>
FN_DEV void acc_to_bfrag(const f32x16 (&acc)[TN], BFrag<PREC> (&b)[kMaxKS]) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = acc[t][8 * s + j];
                if constexpr (PREC == 3) {
                    __bf16 hi, lo;
                    split_bf16(v, hi, lo);
                    b[KS0 + 2 * t + s].hi[j] = hi;
                    b[KS0 + 2 * t + s].lo[j] = lo;
                } else {
                    b[KS0 + 2 * t + s].hi[j] = (__bf16)v;
                }
            }
}

// fp32 rows that leave for (or arrive from) another kernel stream past the caches: non-temporal accesses keep them from
// displacing the packed weights in L2
template <int KIND, class T>
FN_DEV T stream_load(const T* p) { return __builtin_nontemporal_load(p); }
template <int KIND, class T>
FN_DEV void stream_store(T* p, T v) { __builtin_nontemporal_store(v, p); }

template <int TN>
FN_DEV void store_f32(const f32x16 (&acc)[TN], float* __restrict__ dst, int ld, long n, int h, bool valid) {
    if (!valid) return;
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[t][4 * g + e];
            stream_store<0>(reinterpret_cast<f32x4*>(dst + n * ld + 32 * t + 8 * g + 4 * h), v);
        }
}

template <int TN>
FN_DEV void load_f32(f32x16 (&acc)[TN], const float* __restrict__ src, int ld, long n, int h) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = stream_load<3>(reinterpret_cast<const f32x4*>(src + n * ld + 32 * t + 8 * g + 4 * h));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][4 * g + e] = v[e];
        }
}

// ---------------------------------------------------------------------------------------------------------
// Positional encoding (reference models/embedder.py:23-36): feature order [x, sin(2^0 x), cos(2^0 x), ...]
// pe[3 + 6k + c] = sin(2^k x_c), pe[3 + 6k + 3 + c] = cos(2^k x_c).  fn_sincos (fneus_common.h): 9.2e-8 abs (parity <= 1e-6).
// jc[f] = d pe[f] / d x_{f's coordinate}.
// ---------------------------------------------------------------------------------------------------------
template <int L, bool WITH_JAC>
FN_DEV void posenc(const float (&x)[3], float (&pe)[3 + 6 * L], float (&jc)[3 + 6 * L]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pe[c] = x[c];
        if constexpr (WITH_JAC) jc[c] = 1.0f;
    }
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const float f = (float)(1 << k);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s, co;
            fn_sincos(x[c] * f, s, co);
            pe[3 + 6 * k + c] = s;
            pe[3 + 6 * k + 3 + c] = co;
            if constexpr (WITH_JAC) {
                jc[3 + 6 * k + c] = f * co;
                jc[3 + 6 * k + 3 + c] = -f * s;
            }
        }
    }
}

// vector v[NF] (feature order) -> B fragments KS0.. in k-slot order (zero padded)
template <int PREC, int NF, int KS, int KS0>
FN_DEV void vec_to_bfrag(const float (&v)[NF], BFrag<PREC> (&b)[kMaxKS], int h) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            constexpr int dummy = 0;
            (void)dummy;
            const int f0 = phi(ks, 0, j), f1 = phi(ks, 1, j);
            const float a0 = f0 < NF ? v[f0 < NF ? f0 : 0] : 0.0f;
            const float a1 = f1 < NF ? v[f1 < NF ? f1 : 0] : 0.0f;
            const float val = h ? a1 : a0;
            if constexpr (PREC == 3) {
                __bf16 hi, lo;
                split_bf16(val, hi, lo);
                b[KS0 + ks].hi[j] = hi;
                b[KS0 + ks].lo[j] = lo;
            } else {
                b[KS0 + ks].hi[j] = to16<PREC>(val);
            }
        }
}

// accumulator tiles (feature-in-register layout) -> plain feature-order vector v[NF] needs both lane halves;
// instead the consumers below work on the accumulator layout directly:
// sum over features of coef[f] * acc-feature f, features 32*t + acc_row(reg,h); returns this lane's partial
// (caller adds the other half with xor32).
template <int TN, int NF>
FN_DEV float acc_dot_partial(const f32x16 (&acc)[TN], const float (&coef)[NF], int h) {
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f0 = 32 * t + acc_row(r, 0), f1 = 32 * t + acc_row(r, 1);
            const float c0 = f0 < NF ? coef[f0 < NF ? f0 : 0] : 0.0f;
            const float c1 = f1 < NF ? coef[f1 < NF ? f1 : 0] : 0.0f;
            s = fmaf(h ? c1 : c0, acc[t][r], s);
        }
    return s;
}

}  // namespace fneus
