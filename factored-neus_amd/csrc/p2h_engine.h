// Two-pass pipelined layers on 64-SAMPLE workgroups ("p2h"): the schedule of p2_engine.h with two independent workgroups per CU.
//
// p2_engine.h runs 128 samples on one 8-wave workgroup per CU: every wave is in lockstep with the seven others, so whatever
// stalls one pass (a barrier, an operand from HBM behind the in-order vmcnt of a wave) idles the whole CU.  Here a workgroup is
// 4 waves (one per SIMD, <= 256 registers) on 64 samples = two sets of ONE 32-sample tile; wave w owns output tiles 2w, 2w+1:
//   layer l = pass A: MFMAs of tile 0 (accA[2])  ||  vector work: activation of layer l-1, tile 1 (accB[2]) -> B fragments in LDS
//             barrier
//             pass B: MFMAs of tile 1            ||  vector work: activation of layer l, tile 0
//             barrier
// A k-step of a pass is 6 MFMAs (2 output tiles x 3 products) + the activation of 2 values, the same shape as a pass of the
// 8-wave form; a weight fragment serves one sample tile per pass (twice the L2 weight stream per sample, measured to matter
// little), and the SIMD partner is a wave of the OTHER workgroup running the same kind of mixed stream at its own pace.
// LDS per workgroup: 2 tiles x 19 slots x (hi, lo) = 76 KiB + dump: two workgroups per CU.
#pragma once
#include "p2_engine.h"

namespace fneus {

constexpr int kP2hLds = 2 * kP2Half;
constexpr int kP2hDump = kP2hLds;
constexpr int kP2hLdsTotal = kP2hLds + 2 * kFragBytes;

// ACT as in p2_pass: 0 none; 1 softplus -> B fragments of the next layer (k-steps 2 (t0 + i) + sh of tile tV);
//                    2 softplus -> partial dot product with cw (the sdf row of the linear last layer), added to dot
template <int PREC, int KS, int NT_TOTAL, int LMAP, int ACT>
FN_DEV void p2h_pass(const unsigned char* __restrict__ blob, __amdgpu_buffer_rsrc_t rsrc, uint32_t off_hi, uint32_t off_lo,
                     P2Prime<FNEUS_P2_DEPTH, 2>& pr, const P2Next& nx, unsigned char* lds, int lane, int t0, f32x16 (&accM)[2],
                     int tM, f32x16 (&accV)[2], int tV, int tnV, const f32x16 (&cw)[2], float& dot) {
    constexpr int TN = 2;
    constexpr int NPL = PREC == 3 ? 2 : 1;
    constexpr int D = FNEUS_P2_DEPTH;
    constexpr int NV = TN * 16;                          // values of accV per lane
    static_assert(KS >= D, "a pass consumes its D primed stages");
    const unsigned voff = (unsigned)(lane + t0 * 64) * 16u;
#pragma unroll
    for (int i = 0; i < TN; ++i) accM[i] = pr.bias[i];   // bias = initial accumulator
    bf16x8 ah[D + 1][TN], al[D + 1][TN];
#pragma unroll
    for (int s = 0; s < D; ++s)
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            ah[s][i] = pr.ah[s][i];
            if constexpr (PREC == 3) al[s][i] = pr.al[s][i];
        }
    const unsigned char* flM = lds + tM * kP2Half + lane * 16;
    unsigned char* flV = lds + tV * kP2Half + lane * 16;
    unsigned char* dump = lds + kP2hDump + lane * 16;
    bf16x8 bh[3], bl[3];
    auto ldb = [&](int slot, int plane) { return *reinterpret_cast<const bf16x8*>(flM + (slot * NPL + plane) * kFragBytes); };
    bh[0] = ldb(p2_slot<LMAP>(0), 0);
    if constexpr (PREC == 3) bl[0] = ldb(p2_slot<LMAP>(0), 1);
    typedef __attribute__((ext_vector_type(2))) __bf16 p2_bf16x2;
    uint32_t phw[4], plw[4];
    static_for<0, KS>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        constexpr int NSLOT = (PREC == 3 ? 3 : 1) * TN;
        constexpr int NP = NV / 2;
        constexpr int MAXP = (NP + KS - 1) / KS + 1;
        float ve[2 * MAXP], vm[2 * MAXP], vl[2 * MAXP];
        constexpr int p0 = (s * NP + KS - 1) / KS;
        constexpr int np = ((s + 1) * NP + KS - 1) / KS - p0;
        auto micro = [&](auto J_) {
            constexpr int j = decltype(J_)::value;
            if constexpr (j < 4 * np) {
                constexpr int phase = j / (2 * np), vi = j % (2 * np);
                constexpr int v = 2 * p0 + vi;
                constexpr int g = v >> 3, e = v & 7;                // fragment half g = (i, sh), element e
                constexpr int i = g >> 1, sh = g & 1;
                if constexpr (phase == 0) {
                    const float z = accV[i][8 * sh + e];
                    ve[vi] = fast_exp2(-fabsf(z) * (kBeta * kLog2e));
                    asm volatile("v_max_f32 %0, 0, %2" : "=v"(vm[vi]), "+v"(ve[vi]) : "v"(z));
                } else {
                    vl[vi] = fast_log2(1.0f + ve[vi]);
                    asm volatile("" : "+v"(vl[vi]));
                }
            } else {
                constexpr int pi = j - 4 * np;
                constexpr int v = 2 * (p0 + pi);
                constexpr int g = v >> 3, e = v & 7;
                constexpr int i = g >> 1, sh = g & 1;
                const float y0 = fmaf(vl[2 * pi], kLn2 / kBeta, vm[2 * pi]);
                const float y1 = fmaf(vl[2 * pi + 1], kLn2 / kBeta, vm[2 * pi + 1]);
                if constexpr (ACT == 1) {
                    p2_bf16x2 hv = {(__bf16)y0, (__bf16)y1};
                    const uint32_t pk = __builtin_bit_cast(uint32_t, hv);
                    phw[e >> 1] = pk;
                    if constexpr (PREC == 3) {
                        const float h0f = __builtin_bit_cast(float, pk << 16), h1f = __builtin_bit_cast(float, pk & 0xffff0000u);
                        p2_bf16x2 lv = {(__bf16)(y0 - h0f), (__bf16)(y1 - h1f)};
                        plw[e >> 1] = __builtin_bit_cast(uint32_t, lv);
                        asm volatile("" : "+v"(phw[e >> 1]), "+v"(plw[e >> 1]));
                    } else {
                        asm volatile("" : "+v"(phw[e >> 1]));
                    }
                    if constexpr (e == 6) {          // unconditional store (p2_engine.h): unpublished tiles go to the dump area
                        const int ks = 2 * (t0 + i) + sh;
                        unsigned char* dst = i < tnV ? flV + (ks * NPL) * kFragBytes : dump;
                        *reinterpret_cast<p2_u32x4*>(dst) = p2_u32x4{phw[0], phw[1], phw[2], phw[3]};
                        if constexpr (PREC == 3) *reinterpret_cast<p2_u32x4*>(dst + kFragBytes) = p2_u32x4{plw[0], plw[1], plw[2], plw[3]};
                    }
                } else {
                    dot = fmaf(y0, cw[i][8 * sh + e], dot);
                    dot = fmaf(y1, cw[i][8 * sh + e + 1], dot);
                    asm volatile("" : "+v"(dot));
                }
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NSLOT>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            {
                constexpr int i = q % TN, prod = q / TN;           // product-major: consecutive MFMAs go to different accumulators
                if constexpr (PREC == 3) {
                    if constexpr (prod == 0) accM[i] = mfma32(al[s % (D + 1)][i], bh[s % 3], accM[i]);
                    else if constexpr (prod == 1) accM[i] = mfma32(ah[s % (D + 1)][i], bl[s % 3], accM[i]);
                    else accM[i] = mfma32(ah[s % (D + 1)][i], bh[s % 3], accM[i]);
                } else {
                    accM[i] = mfma32(ah[s % (D + 1)][i], bh[s % 3], accM[i]);
                }
            }
            // the next pass's bias: requested late (32 registers that would otherwise be held through the whole pass)
            if constexpr (q == 0 && s == (KS >= 3 ? KS - 2 : 0)) p2_prime_bias<PREC, D, TN>(pr, blob, lane, t0, nx);
            // ---- operand requests: slots 0 .. NPL-1 the B fragment planes of k-step s + 1, the remaining slots the TN x NPL
            //      weight fragments of k-step s + D (or the next pass's first stages)
            if constexpr (q < NPL && s + 1 < KS) {
                if constexpr (q == 0) bh[(s + 1) % 3] = ldb(p2_slot<LMAP>(s + 1), 0);
                else bl[(s + 1) % 3] = ldb(p2_slot<LMAP>(s + 1), 1);
            }
            constexpr int NWS = NSLOT - NPL > 0 ? NSLOT - NPL : 1;         // slots that carry weight requests
            constexpr int qw = NSLOT - NPL > 0 ? q - NPL : q;
            if constexpr (qw >= 0 && qw < NWS) {
                constexpr int per = (TN * NPL + NWS - 1) / NWS;
#pragma unroll
                for (int u = qw * per; u < (qw + 1) * per && u < TN * NPL; ++u) {
                    const int i = u % TN, plane = u / TN;
                    if constexpr (s + D < KS) {
                        const uint32_t f = (uint32_t)(((s + D) * NT_TOTAL + i) * 64) * 16u;
                        if (plane == 0) ah[(s + D) % (D + 1)][i] = p2_wload(rsrc, voff, off_hi + f, blob);
                        else al[(s + D) % (D + 1)][i] = p2_wload(rsrc, voff, off_lo + f, blob);
                    } else {
                        constexpr int sn = s + D - KS;
                        const uint32_t f = (uint32_t)((sn * nx.nt + i) * 64) * 16u;
                        if (plane == 0) pr.ah[sn][i] = p2_wload(rsrc, voff, nx.off_hi + f, blob);
                        else pr.al[sn][i] = p2_wload(rsrc, voff, nx.off_lo + f, blob);
                    }
                }
            }
            if constexpr (ACT != 0) {
                static_for<0, 5 * np>([&](auto J_) {
                    constexpr int j = decltype(J_)::value;
                    if constexpr ((j * NSLOT) / (5 * np) == q) micro(J_);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // HAZARD (p2_engine.h): the B operands of this k-step stay live to its end
        asm volatile("" ::"v"(bh[s % 3]));
        if constexpr (PREC == 3) asm volatile("" ::"v"(bl[s % 3]));
    });
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
}

// the vector work of a pass alone (after the last MFMA pass of a launch): softplus -> dot product with cw
template <int PREC>
FN_DEV void p2h_dot_only(f32x16 (&accV)[2], const f32x16 (&cw)[2], float& dot) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) dot = fmaf(softplus100(accV[i][e]), cw[i][e], dot);
}

}  // namespace fneus
