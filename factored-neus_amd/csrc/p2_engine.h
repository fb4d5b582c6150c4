// Two-pass pipelined layers ("p2"): the chain-kernel form of round 3.
//
// What round 3 measured (tools/experiments/r03/coissue, k1_stamps.py, k1_s8_stamps.py):
//   * a wave whose stream is 1 MFMA + <= 4 vector instructions runs at the full matrix rate (33 cycles per
//     v_mfma_f32_32x32x16_bf16), alone on its SIMD or with a partner running the same mix;
//   * an MFMA-only wave beside a vector-only wave on the same SIMD slows BOTH (MFMA 75 cycles each, vector instructions 140+
//     cycles each).  Rounds 1 / 2 ran a layer as [MFMAs of all tiles] then [activation of all tiles]: with two independent
//     workgroups per CU the phases of SIMD partners drift against each other into exactly that pairing (35-46 % MFMA busy),
//     and in lockstep (8-wave workgroups) the matrix pipe idles during every activation phase (56 %).
// So the activation work has to sit INSIDE the MFMA stream of the same wave, which needs two independent halves per wave:
//
//   workgroup = 4 waves (one per SIMD, 512 registers each), 128 samples = 4 tiles in two sets S0 = {0, 1}, S1 = {2, 3};
//   wave w owns output tiles 2w, 2w+1 of every layer (tensor-parallel form, B fragments of all 128 samples in LDS);
//   layer l = pass A: MFMAs of set S0 (accA)  ||  vector work: activation of layer l-1, set S1 (accB) -> B fragments in LDS
//             barrier
//             pass B: MFMAs of set S1 (accB)  ||  vector work: activation of layer l,   set S0 (accA) -> B fragments in LDS
//             barrier
//   per k-step of a pass: 12 MFMAs (2 tiles x 2 sample tiles x 3 products) and the activation of 4 values (~36 vector
//   instructions), interleaved 1 : 3 by sched_group_barrier.  A weight fragment serves two sample tiles per pass; pass B takes
//   the same fragments again (L2; `KEEP` holds the hi parts of pass A in registers instead).
// LDS (per 32-sample tile): slots 0..15 = B fragments of the running layer's input (hi, lo), slots 16..18 = the positional
// encoding of the unit (read in place by layer 0 and by the skip input of layer 4: no copy).
#pragma once
#include <type_traits>
#include "pp_engine.h"

namespace fneus {

constexpr int kP2Half = 19 * 2 * kFragBytes;
constexpr int kP2Lds = 4 * kP2Half;
constexpr int kP2Dump = kP2Lds;                     // 2 KiB behind the fragments: where unpublished tiles are stored
constexpr int kP2LdsTotal = kP2Lds + 2 * kFragBytes;

// LDS slot of k-step s of a layer's B operand.  LMAP 0: slots 0.., 1: the encoding alone (layer 0), 2: 14 slots of h_4 then the
// encoding (layer 4, fields.py:83-84)
template <int LMAP>
FN_DEV constexpr int p2_slot(int s) { return LMAP == 1 ? 16 + s : (LMAP == 2 ? (s < 14 ? s : s + 2) : s); }

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1 (indices that must be constant expressions: register
// arrays indexed by anything else end up in scratch)
template <int I, int N, class F>
FN_DEV void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

FN_DEV void p2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// FNEUS_P2_STAMPS (timing experiments only): shader cycles of waves 0 and 4 of block 0, summed per part of a pass --
// [0] pass entry -> first k-step, [1] the k-steps, [2] behind the k-steps, [3] outside p2_pass (barriers, encode, finish)
#ifdef FNEUS_P2_STAMPS
__device__ unsigned long long g_p2_stamp[2][4];
__device__ unsigned long long g_p2_last[2];
#define P2_STAMP(k)                                                                                   \
    do {                                                                                              \
        if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) {                            \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                               \
            const int w_ = threadIdx.x >> 8;                                                          \
            g_p2_stamp[w_][k] += t_ - g_p2_last[w_];                                                  \
            g_p2_last[w_] = t_;                                                                       \
        }                                                                                             \
    } while (0)
#else
#define P2_STAMP(k) do { } while (0)
#endif

#ifndef FNEUS_P2_DEPTH
#define FNEUS_P2_DEPTH 3           // weight-prefetch distance in k-steps (a k-step of a pass = 12 MFMAs = 384 cycles)
#endif
#ifndef FNEUS_P2_BD
#define FNEUS_P2_BD 1              // prefetch distance of the B fragments (LDS) in k-steps; the ring holds BD + 2 k-steps
#endif

typedef __attribute__((ext_vector_type(4))) unsigned int p2_u32x4;


// weight fragments through BUFFER loads: uniform descriptor + ONE per-lane byte offset (VGPR) + a uniform offset in an SGPR:
// no vector instruction goes into addressing (global loads with offsets beyond 4 KiB cost two v_add per load, in a stream
// whose vector-issue slots are the scarce resource)
FN_DEV __amdgpu_buffer_rsrc_t p2_rsrc(const unsigned char* blob) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(blob), 0, 0x7fffffff, 0x00020000);
}
FN_DEV bf16x8 p2_wload(__amdgpu_buffer_rsrc_t r, unsigned voff, uint32_t soff, const unsigned char* blob) {
#ifdef FNEUS_P2_GLOBAL_LOADS            // debugging only
    return *reinterpret_cast<const bf16x8 FN_GLOBAL*>((gblob_t)blob + soff + voff);
#endif
#ifdef FNEUS_P2_NO_WEIGHTS              // timing experiments only: no weight stream from L2
    bf16x8 t;
    for (int e = 0; e < 8; ++e) t[e] = (__bf16)(0.001f * (float)(voff + e));
    asm volatile("" : "+v"(t));
    return t;
#else
    const p2_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return __builtin_bit_cast(bf16x8, v);
#endif
}

// what a pass needs before its first MFMA and cannot fetch without exposing the latency: the first D weight stages and the
// bias.  Requested by the PREVIOUS pass (in the slots of its last D k-steps), consumed by the pass itself.
struct P2Next {                         // where the next pass finds its weights (uniform values)
    uint32_t off_hi, off_lo, off_bias;
    int nt;                             // output tiles of the next pass's layer (fragment order [ks][t])
};
template <int D, int TN = 2>
struct P2Prime {
    bf16x8 ah[D][TN], al[D][TN];
    f32x16 bias[TN];
};

template <int PREC, int D, int TN>
FN_DEV void p2_prime_stage(P2Prime<D, TN>& pr, int s, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, const P2Next& nx,
                           const unsigned char* blob) {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const uint32_t f = (uint32_t)((s * nx.nt + i) * 64) * 16u;
        pr.ah[s][i] = p2_wload(rsrc, voff, nx.off_hi + f, blob);
        if constexpr (PREC == 3) pr.al[s][i] = p2_wload(rsrc, voff, nx.off_lo + f, blob);
    }
}
template <int PREC, int D, int TN>
FN_DEV void p2_prime_bias(P2Prime<D, TN>& pr, const unsigned char* __restrict__ blob, int lane, int t0, const P2Next& nx) {
    const f32x16 FN_GLOBAL* __restrict__ p = reinterpret_cast<const f32x16 FN_GLOBAL*>((gblob_t)blob + nx.off_bias);
#pragma unroll
    for (int i = 0; i < TN; ++i) pr.bias[i] = p[(t0 + i) * 2 + (lane >> 5)];
}
// everything at once (before the first pass of a launch)
template <int PREC, int D, int TN>
FN_DEV void p2_prime_all(P2Prime<D, TN>& pr, const unsigned char* __restrict__ blob, __amdgpu_buffer_rsrc_t rsrc, int lane, int t0,
                         const P2Next& nx) {
    const unsigned voff = (unsigned)(lane + t0 * 64) * 16u;
#pragma unroll
    for (int s = 0; s < D; ++s) p2_prime_stage<PREC, D, TN>(pr, s, rsrc, voff, nx, blob);
    p2_prime_bias<PREC, D, TN>(pr, blob, lane, t0, nx);
}

// ---- vector work of the forward chain of K1: softplus -> B fragments (ACT 1) or -> dot product with the sdf row (ACT 2) ----
// One k-step = NSLOT slots of [1 MFMA | one micro-step | one operand request], fenced slot by slot: the placement is written
// out.  (hipcc's sched_group_barrier solver gave up on most passes and left the vector work in one block beside the MFMAs:
// the time of a pass was then the SUM of the two; under a branch LLVM sinks whole phases out of their slots.)
//   MFMAs product-major: slots 0..3 lo.hi, 4..7 hi.lo, 8..11 hi.hi of accumulators (i, hb) = (0,0) (0,1) (1,0) (1,1):
//   consecutive MFMAs go to different accumulators, per accumulator the order is that of every other kernel of the engine.
//   Vector work phase-major (A for all values of the k-step, then B, then C): dependent instructions are slots apart, a slot
//   holds at most one transcendental:
//     A: z <- accumulator (AGPR), e = exp2(-|z| beta log2 e), m = max(z, 0)     B: L = log2(1 + e)
//     C: y = m + L ln2 / beta, hi / lo split, (every 8th value) store of the fragment half
//   Operand requests: slots 0..3 the weight fragments of k-step s + D (or the next pass's first stages), slots 4..7 the B
//   fragments of k-step s + 1.
// ACT: 0 none; 1 softplus -> B fragments of the next layer (k-steps 2 (t0 + i) + sh of tiles hbV, hbV + 1);
//      2 softplus -> partial dot product with cw (the sdf row of the linear last layer), added to dot[];
//      3 / 4: the same with ReLU instead of the softplus (the stage-2 visibility network, lvis_kernels.hip)
template <int PREC, int KS, int NT_TOTAL, int LMAP, int ACT, int TN = 2>
FN_DEV void p2_pass(const unsigned char* __restrict__ blob, __amdgpu_buffer_rsrc_t rsrc, uint32_t off_hi, uint32_t off_lo,
                    P2Prime<FNEUS_P2_DEPTH, TN>& pr, const P2Next& nx, unsigned char* lds, int lane, int t0, f32x16 (&accM)[TN][2], int hbM,
                    f32x16 (&accV)[TN][2], int hbV, int tnV, const f32x16 (&cw)[TN], float (&dot)[2]) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
    constexpr int D = FNEUS_P2_DEPTH;
    constexpr int NV = TN * 32;                          // values of accV per lane: TN tiles x 2 sample tiles x 16
    constexpr bool RELU = ACT >= 3, FRAGS = ACT == 1 || ACT == 3;
    static_assert(KS >= D, "a pass consumes its D primed stages");
    const unsigned voff = (unsigned)(lane + t0 * 64) * 16u;
    P2_STAMP(3);
#pragma unroll
    for (int i = 0; i < TN; ++i) {                       // bias = initial accumulator
        accM[i][0] = pr.bias[i];
        accM[i][1] = pr.bias[i];
    }
    bf16x8 ah[D + 1][TN], al[D + 1][TN];
#pragma unroll
    for (int s = 0; s < D; ++s)
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            ah[s][i] = pr.ah[s][i];
            if constexpr (PREC == 3) al[s][i] = pr.al[s][i];
        }
    const unsigned char* flM = lds + hbM * kP2Half + lane * 16;
    unsigned char* flV = lds + hbV * kP2Half + lane * 16;
    unsigned char* dump = lds + kP2Dump + lane * 16;
    constexpr int BD = FNEUS_P2_BD, RB = BD + 2;
    bf16x8 bh[RB][2], bl[RB][2];
#ifdef FNEUS_P2_NO_LDSB                 // timing experiments only: no B-fragment reads from LDS
    bf16x8 bconst;
    for (int e = 0; e < 8; ++e) bconst[e] = (__bf16)(0.002f * (float)(lane - e));
    auto ldb = [&](int hb, int slot, int plane) { bf16x8 t = bconst; asm volatile("" : "+v"(t)); return t; };
#else
    auto ldb = [&](int hb, int slot, int plane) { return *reinterpret_cast<const bf16x8*>(flM + hb * kP2Half + (slot * NPL + plane) * kFragBytes); };
#endif
#pragma unroll
    for (int b0 = 0; b0 < BD; ++b0)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            if (b0 < KS) {
                bh[b0][hb] = ldb(hb, p2_slot<LMAP>(b0 < KS ? b0 : 0), 0);
                if constexpr (PREC == 3) bl[b0][hb] = ldb(hb, p2_slot<LMAP>(b0 < KS ? b0 : 0), 1);
            }
        }
    p2_prime_bias<PREC, D, TN>(pr, blob, lane, t0, nx); // (the registers are free again: next pass's bias)
    typedef __attribute__((ext_vector_type(2))) __bf16 p2_bf16x2;
    uint32_t phw[4], plw[4];                             // the fragment half being assembled by the vector work (4 x 2 bf16)
#ifdef FNEUS_P2_VCOPY
    f32x16 vv[TN][2];
    if constexpr (ACT != 0) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                vv[i][hb] = accV[i][hb];
                asm volatile("" : "+v"(vv[i][hb]));
            }
    }
#else
    f32x16 (&vv)[TN][2] = accV;
#endif
    P2_STAMP(0);
    static_for<0, KS>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        constexpr int NSLOT = (PREC == 3 ? 6 : 2) * TN;
        constexpr int NP = NV / 2;                                  // value PAIRS of accV (a pair = one packed bf16 word)
        constexpr int MAXP = (NP + KS - 1) / KS + 1;
        float ve[2 * MAXP], vm[2 * MAXP], vl[2 * MAXP];
        constexpr int p0 = (s * NP + KS - 1) / KS;                  // first pair of this k-step: ceil(s NP / KS)
        constexpr int np = ((s + 1) * NP + KS - 1) / KS - p0;
        // micro-steps of the k-step: A(v) for its 2 np values, B(v) likewise, C(p) for its np pairs.  Every result is anchored in
        // its slot by an empty asm: LLVM otherwise sinks the arithmetic to its only use (the fragment store, 8 values later)
        auto micro = [&](auto J_) {
            constexpr int j = decltype(J_)::value;
            if constexpr (j < 4 * np) {
                constexpr int phase = j / (2 * np), vi = j % (2 * np);
                constexpr int v = 2 * p0 + vi;
                constexpr int g = v >> 3, e = v & 7;                // fragment half g = (i, hb, sh), element e
                constexpr int i = g >> 2, hb = (g >> 1) & 1, sh = g & 1;
                if constexpr (phase == 0) {
                    const float z = vv[i][hb][8 * sh + e];
                    if constexpr (RELU) {
                        asm volatile("v_max_f32 %0, 0, %1" : "=v"(vm[vi]) : "v"(z));
                    } else {
#ifdef FNEUS_DBG_CHEAP_ACT
                        ve[vi] = 0.0f;
#else
                        ve[vi] = fast_exp2(-fabsf(z) * (kBeta * kLog2e));
#endif
                        asm volatile("v_max_f32 %0, 0, %2" : "=v"(vm[vi]), "+v"(ve[vi]) : "v"(z));   // max(z, 0) in ONE instruction
                    }
                } else if constexpr (!RELU) {
#ifdef FNEUS_DBG_CHEAP_ACT
                    vl[vi] = ve[vi];
#else
                    vl[vi] = fast_log2(1.0f + ve[vi]);
#endif
                    asm volatile("" : "+v"(vl[vi]));
                }
            } else {
                constexpr int pi = j - 4 * np;                      // pair index within the k-step
                constexpr int v = 2 * (p0 + pi);
                constexpr int g = v >> 3, e = v & 7;
                constexpr int i = g >> 2, hb = (g >> 1) & 1, sh = g & 1;
                const float y0 = RELU ? vm[2 * pi] : fmaf(vl[2 * pi], kLn2 / kBeta, vm[2 * pi]);
                const float y1 = RELU ? vm[2 * pi + 1] : fmaf(vl[2 * pi + 1], kLn2 / kBeta, vm[2 * pi + 1]);
                if constexpr (FRAGS) {
                    p2_bf16x2 hv = {to16<PREC>(y0), to16<PREC>(y1)};
                    const uint32_t pk = __builtin_bit_cast(uint32_t, hv);       // ONE v_cvt_pk_bf16_f32; the hi parts as floats
                    phw[e >> 1] = pk;                                           // come back out of the packed word (shift / mask)
                    if constexpr (PREC == 3) {
                        const float h0f = __builtin_bit_cast(float, pk << 16), h1f = __builtin_bit_cast(float, pk & 0xffff0000u);
                        p2_bf16x2 lv = {(__bf16)(y0 - h0f), (__bf16)(y1 - h1f)};
                        plw[e >> 1] = __builtin_bit_cast(uint32_t, lv);
                        asm volatile("" : "+v"(phw[e >> 1]), "+v"(plw[e >> 1]));
                    } else {
                        asm volatile("" : "+v"(phw[e >> 1]));
                    }
                    if constexpr (e == 6) {
                        // UNCONDITIONAL store (a tile that is not published -- layer 3 has 7 -- goes to a dump area behind the
                        // fragments): under a branch LLVM sinks the vector work of the whole fragment half into the branch
                        const int ks = 2 * (t0 + i) + sh;
                        unsigned char* dst = i < tnV ? flV + hb * kP2Half + (ks * NPL) * kFragBytes : dump;
#ifdef FNEUS_P2_NO_LDSW                 // timing experiments only
                        asm volatile("" :: "v"(phw[0]), "v"(phw[3]), "v"(dst));
                        if constexpr (PREC == 3) asm volatile("" :: "v"(plw[0]), "v"(plw[3]));
#else
                        *reinterpret_cast<p2_u32x4*>(dst) = p2_u32x4{phw[0], phw[1], phw[2], phw[3]};
                        if constexpr (PREC == 3) *reinterpret_cast<p2_u32x4*>(dst + kFragBytes) = p2_u32x4{plw[0], plw[1], plw[2], plw[3]};
#endif
                    }
                } else {
                    dot[hb] = fmaf(y0, cw[i][8 * sh + e], dot[hb]);
                    dot[hb] = fmaf(y1, cw[i][8 * sh + e + 1], dot[hb]);
                    asm volatile("" : "+v"(dot[hb]));
                }
            }
        };
#ifdef FNEUS_P2_B_AT_START
        if constexpr (s + 1 < KS) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                bh[(s + 1) % 3][hb] = ldb(hb, p2_slot<LMAP>(s + 1), 0);
                if constexpr (PREC == 3) bl[(s + 1) % 3][hb] = ldb(hb, p2_slot<LMAP>(s + 1), 1);
            }
        }
        if constexpr (s >= 1) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                asm volatile("" ::"v"(bh[(s - 1) % 3][hb]));
                if constexpr (PREC == 3) asm volatile("" ::"v"(bl[(s - 1) % 3][hb]));
            }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NSLOT>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
#ifndef FNEUS_P2_NO_MFMA
            {
                constexpr int NACC = 2 * TN;                       // accumulators (i, hb) = (r >> 1, r & 1), product-major
                constexpr int r = q % NACC, prod = q / NACC;
                constexpr int i = r >> 1, hb = r & 1;
                if constexpr (PREC == 3) {
                    if constexpr (prod == 0) accM[i][hb] = mfma32(al[s % (D + 1)][i], bh[s % RB][hb], accM[i][hb]);
                    else if constexpr (prod == 1) accM[i][hb] = mfma32(ah[s % (D + 1)][i], bl[s % RB][hb], accM[i][hb]);
                    else accM[i][hb] = mfma32(ah[s % (D + 1)][i], bh[s % RB][hb], accM[i][hb]);
                } else {
                    accM[i][hb] = mfma32p<PREC>(ah[s % (D + 1)][i], bh[s % RB][hb], accM[i][hb]);
                }
            }
#else
            asm volatile("" :: "v"(ah[s % (D + 1)][0]), "v"(al[s % (D + 1)][0]), "v"(bh[s % RB][q & 1]), "v"(bl[s % RB][q & 1]));
#endif
            // ---- operand requests of this slot
            constexpr int NREQ = NSLOT >= 12 ? 4 : (NSLOT >= 4 ? 2 : 1);      // slots per request group
#ifdef FNEUS_P2_W_FIRST
            constexpr int qw = q, qb = q - NREQ;
#else
            constexpr int qw = q - NREQ, qb = q;           // B fragments first: they are needed at the next k-step, the weights D later
#endif
            if constexpr (qw >= 0 && qw < NREQ) {          // weight fragments: (tile, plane) = (qw & 1, qw >> 1) (parity mode)
                constexpr int per = (TN * NPL + NREQ - 1) / NREQ;
#pragma unroll
                for (int u = qw * per; u < (qw + 1) * per && u < TN * NPL; ++u) {
                    const int i = u % TN, plane = u / TN;
                    if constexpr (s + D < KS) {
                        const uint32_t f = (uint32_t)(((s + D) * NT_TOTAL + i) * 64) * 16u;
                        if (plane == 0) ah[(s + D) % (D + 1)][i] = p2_wload(rsrc, voff, off_hi + f, blob);
                        else al[(s + D) % (D + 1)][i] = p2_wload(rsrc, voff, off_lo + f, blob);
                    } else {                               // the next pass's stage s + D - KS
                        constexpr int sn = s + D - KS;
                        const uint32_t f = (uint32_t)((sn * nx.nt + i) * 64) * 16u;
                        if (plane == 0) pr.ah[sn][i] = p2_wload(rsrc, voff, nx.off_hi + f, blob);
                        else pr.al[sn][i] = p2_wload(rsrc, voff, nx.off_lo + f, blob);
                    }
                }
            }
#ifndef FNEUS_P2_B_AT_START
            if constexpr (qb >= 0 && qb < NREQ && s + BD < KS) {      // B fragments of k-step s + BD
                constexpr int per = (2 * NPL + NREQ - 1) / NREQ;
#pragma unroll
                for (int u = qb * per; u < (qb + 1) * per && u < 2 * NPL; ++u) {
                    const int hb = u & 1, plane = u >> 1;
                    if (plane == 0) bh[(s + BD) % RB][hb] = ldb(hb, p2_slot<LMAP>(s + BD < KS ? s + BD : 0), 0);
                    else bl[(s + BD) % RB][hb] = ldb(hb, p2_slot<LMAP>(s + BD < KS ? s + BD : 0), 1);
                }
            }
#endif
#ifndef FNEUS_P2_NO_VALU
            if constexpr (ACT != 0) {
                static_for<0, 5 * np>([&](auto J_) {           // micro-steps j with floor(j NSLOT / (5 np)) == q
                    constexpr int j = decltype(J_)::value;
                    if constexpr ((j * NSLOT) / (5 * np) == q) micro(J_);
                });
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
        // HAZARD (dense_ldsb(), tools/dbg_race.py): an LDS load must not land in registers that a still-queued MFMA reads
        // (nothing interlocks that write-after-read, and issue runs ahead of the matrix pipe).  The B operands of this k-step
        // stay live to its end, so the prefetch of k-step s + 1 (slots 4..7) cannot be given the registers of an operand whose
        // last MFMA has only just been issued (seen: the lo fragment of slot 7).
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            asm volatile("" ::"v"(bh[s % RB][hb]));
            if constexpr (PREC == 3) asm volatile("" ::"v"(bl[s % RB][hb]));
        }
    });
    P2_STAMP(1);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    P2_STAMP(2);
#ifdef FNEUS_P2_DRAIN
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_sleep 8" ::: "memory");
#endif
}

// the vector work of a pass alone (after the last MFMA pass of a launch)
template <int PREC, int ACT, int TN = 2>
FN_DEV void p2_valu_only(unsigned char* lds, int lane, int t0, f32x16 (&accV)[TN][2], int hbV, int tnV, const f32x16 (&cw)[TN],
                         float (&dot)[2]) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
    constexpr bool RELU = ACT >= 3, FRAGS = ACT == 1 || ACT == 3;
    unsigned char* flV = lds + hbV * kP2Half + lane * 16;
#pragma unroll
    for (int g = 0; g < 4 * TN; ++g) {
        const int i = g >> 2, hb = (g >> 1) & 1, sh = g & 1;
        bf16x8 ph, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float y = RELU ? fmaxf(accV[i][hb][8 * sh + e], 0.0f) : softplus100(accV[i][hb][8 * sh + e]);
            if constexpr (FRAGS) {
                if constexpr (PREC == 3) {
                    __bf16 a, b2;
                    split_bf16(y, a, b2);
                    ph[e] = a;
                    pl[e] = b2;
                } else {
                    ph[e] = to16<PREC>(y);
                }
            } else {
                dot[hb] = fmaf(y, cw[i][8 * sh + e], dot[hb]);
            }
        }
        if constexpr (FRAGS) {
            if (i < tnV) {
                const int ks = 2 * (t0 + i) + sh;
                *reinterpret_cast<bf16x8*>(flV + hb * kP2Half + (ks * NPL) * kFragBytes) = ph;
                if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(flV + hb * kP2Half + (ks * NPL + 1) * kFragBytes) = pl;
            }
        }
    }
}

}  // namespace fneus
