// Two-pass pipelined layers (p2_engine.h) for a ReLU network WITH the stash of a training step: the forward chain of the colour
// network (RenderingNetwork, reference models/fields.py:150-175).  The pass is the one of p2_train.h; the vector work is
//   ACT 7: ReLU -> B fragments in LDS, the u_l plane (hi; lo in the exact-gradient mode) and the ReLU bit mask of the tile;
//   ACT 8: ReLU -> plane + mask and the partial dot products with the 3 rows of the output layer (256 -> 3 as vector work).
// Mask format (include/fneus.h FneusColStash.mask, what color_bwd reads): per (tile, layer, lane) 128 bits = 4 words, word j
// = output tiles 2j (low half) and 2j + 1 (high half), bit e = accumulator register e.  A wave owns ONE output tile here, so it
// stores one 16-bit half.  The bits are collected with the carry: v_add_co(y, -1) sets it iff y != 0 (y = max(z, 0) >= 0),
// v_addc_co(m, m) shifts it into m -- two instructions per value, MSB first (v_bfrev + shift at the end).
#pragma once
#include "p2_train.h"

namespace fneus {

FN_DEV void p2_store16(uint32_t v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b16((short)v, r, (int)voff, soff, 0);
}

template <int PREC, int KS, int NT_TOTAL, int ACT, int MODE>
FN_DEV void p2_pass_relu(const unsigned char* __restrict__ blob, __amdgpu_buffer_rsrc_t rsrc, uint32_t off_hi, uint32_t off_lo,
                         P2Prime<FNEUS_P2_DEPTH, 1>& pr, const P2Next& nx, unsigned char* lds, int lane, int t0,
                         f32x16 (&accM)[1][2], int hbM, f32x16 (&accV)[1][2], int hbV, const f32x16 (&cw)[3], float (&dot)[2][3],
                         const P2St& so, unsigned voff_even, unsigned voff_odd, unsigned voff_mask) {
    // so.sig: the mask blocks of the tile pair (tile hb at + hb * 4 KiB); voff_mask: byte offset of this lane's 16-bit half
    constexpr int TN = 1;
    constexpr int NPL = PREC == 3 ? 2 : 1;
    constexpr int D = FNEUS_P2_DEPTH;
    constexpr int NV = TN * 32;
    constexpr bool TRAIN = MODE != 0;
    constexpr bool LO = MODE == 3 && PREC == 3;
    constexpr bool FRAGS = ACT == 7, DOT = ACT == 8;
    static_assert(KS >= D, "a pass consumes its D primed stages");
    const unsigned voff = (unsigned)(lane + t0 * 64) * 16u;
    accM[0][0] = pr.bias[0];
    accM[0][1] = pr.bias[0];
    bf16x8 ah[D + 1][TN], al[D + 1][TN];
#pragma unroll
    for (int s = 0; s < D; ++s) {
        ah[s][0] = pr.ah[s][0];
        if constexpr (PREC == 3) al[s][0] = pr.al[s][0];
    }
    const unsigned char* flM = lds + hbM * kP2Half + lane * 16;
    unsigned char* flV = lds + hbV * kP2Half + lane * 16;
    bf16x8 bh[3][2], bl[3][2];
    auto ldb = [&](int hb, int slot, int plane) { return *reinterpret_cast<const bf16x8*>(flM + hb * kP2Half + (slot * NPL + plane) * kFragBytes); };
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        bh[0][hb] = ldb(hb, 0, 0);
        if constexpr (PREC == 3) bl[0][hb] = ldb(hb, 0, 1);
    }
    p2_prime_bias<PREC, D, TN>(pr, blob, lane, t0, nx);
    typedef __attribute__((ext_vector_type(2))) __bf16 p2_bf16x2;
    uint32_t phw[4], plw[4], mbits = 0u;
    f32x16 (&vv)[TN][2] = accV;
    static_for<0, KS>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        constexpr int NSLOT = (PREC == 3 ? 6 : 2) * TN;
        constexpr int NP = NV / 2;
        constexpr int GS = 2;
        float vm[2 * GS];
        constexpr int p0 = (s * NP + KS - 1) / KS;
        constexpr int np = ((s + 1) * NP + KS - 1) / KS - p0;
        // micro-steps, per group of gs <= GS pairs: A(v) for its 2 gs values, then C(p) M(p) H(p) pair by pair: 5 gs; 5 np in all
        constexpr int NM = ACT == 0 ? 0 : 5 * np;
        auto micro = [&](auto J_) {
            constexpr int j = decltype(J_)::value;
            constexpr int gi = j / (5 * GS), jj = j % (5 * GS);
            constexpr int gp0 = gi * GS;
            constexpr int gs = np - gp0 < GS ? np - gp0 : GS;
            if constexpr (jj < 2 * gs) {                                // A: ReLU
                constexpr int v = 2 * (p0 + gp0) + jj;
                constexpr int g = v >> 3, e = v & 7;
                constexpr int i = g >> 2, hb = (g >> 1) & 1, sh = g & 1;
                asm volatile("v_max_f32 %0, 0, %1" : "=v"(vm[jj]) : "v"(vv[i][hb][8 * sh + e]));
            } else {
                constexpr int pi = (jj - 2 * gs) / 3, what = (jj - 2 * gs) % 3;
                constexpr int v = 2 * (p0 + gp0 + pi);
                constexpr int g = v >> 3, e = v & 7;
                constexpr int i = g >> 2, hb = (g >> 1) & 1, sh = g & 1;
                const float y0 = vm[2 * pi], y1 = vm[2 * pi + 1];
                if constexpr (what == 0) {                              // C: hi / lo words, LDS store, dot products
                    if constexpr (FRAGS || TRAIN) {
                        p2_bf16x2 hv = {(__bf16)y0, (__bf16)y1};
                        const uint32_t pk = __builtin_bit_cast(uint32_t, hv);
                        phw[e >> 1] = pk;
                        if constexpr (PREC == 3 && (FRAGS || LO)) {
                            const float h0f = __builtin_bit_cast(float, pk << 16), h1f = __builtin_bit_cast(float, pk & 0xffff0000u);
                            p2_bf16x2 lv = {(__bf16)(y0 - h0f), (__bf16)(y1 - h1f)};
                            plw[e >> 1] = __builtin_bit_cast(uint32_t, lv);
                            asm volatile("" : "+v"(phw[e >> 1]), "+v"(plw[e >> 1]));
                        } else {
                            asm volatile("" : "+v"(phw[e >> 1]));
                        }
                    }
                    if constexpr (DOT) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            dot[hb][c] = fmaf(y0, cw[c][8 * sh + e], dot[hb][c]);
                            dot[hb][c] = fmaf(y1, cw[c][8 * sh + e + 1], dot[hb][c]);
                            asm volatile("" : "+v"(dot[hb][c]));
                        }
                    }
                    if constexpr (FRAGS && e == 6) {
                        const int ks = 2 * (t0 + i) + sh;
                        unsigned char* dst = flV + hb * kP2Half + (ks * NPL) * kFragBytes;
                        *reinterpret_cast<p2_u32x4*>(dst) = p2_u32x4{phw[0], phw[1], phw[2], phw[3]};
                        if constexpr (PREC == 3) *reinterpret_cast<p2_u32x4*>(dst + kFragBytes) = p2_u32x4{plw[0], plw[1], plw[2], plw[3]};
                    }
                } else if constexpr (what == 1) {                       // M: two mask bits through the carry
                    if constexpr (TRAIN) {
                        uint32_t t1, t2;
                        asm volatile("v_add_co_u32 %1, vcc, -1, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                     "v_add_co_u32 %2, vcc, -1, %4\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                                     : "+v"(mbits), "=&v"(t1), "=&v"(t2)
                                     : "v"(y0), "v"(y1)
                                     : "vcc");
                    }
                } else {                                                // H: plane store; the tile's mask when it is complete
                    if constexpr (TRAIN && e == 6) {
                        const int ks = 2 * (t0 + i) + sh;
                        const unsigned vo = (ks & 1) ? voff_odd : voff_even;
                        p2_u32x4 w;
#pragma unroll
                        for (int k = 0; k < 4; ++k) w[k] = so.vmask[hb] ? phw[k] : 0u;
                        p2_store128<true>(w, so.hi, vo, hb * (int)kPPBlock + ks * kFragBytes);
                        if constexpr (LO) {
                            p2_u32x4 w2;
#pragma unroll
                            for (int k = 0; k < 4; ++k) w2[k] = so.vmask[hb] ? plw[k] : 0u;
                            p2_store128<true>(w2, so.lo, vo, hb * (int)kPPBlock + ks * kFragBytes);
                        }
                        if constexpr (sh == 1) {
                            const uint32_t m16 = __builtin_bitreverse32(mbits) >> 16;
                            p2_store16(m16, so.sig, voff_mask, hb * 4096);
                            mbits = 0u;
                            asm volatile("" : "+v"(mbits));
                        }
                    }
                }
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NSLOT>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            {
                constexpr int NACC = 2 * TN;
                constexpr int r = q % NACC, prod = q / NACC;
                constexpr int i = r >> 1, hb = r & 1;
                if constexpr (PREC == 3) {
                    if constexpr (prod == 0) accM[i][hb] = mfma32(al[s % (D + 1)][i], bh[s % 3][hb], accM[i][hb]);
                    else if constexpr (prod == 1) accM[i][hb] = mfma32(ah[s % (D + 1)][i], bl[s % 3][hb], accM[i][hb]);
                    else accM[i][hb] = mfma32(ah[s % (D + 1)][i], bh[s % 3][hb], accM[i][hb]);
                } else {
                    accM[i][hb] = mfma32(ah[s % (D + 1)][i], bh[s % 3][hb], accM[i][hb]);
                }
            }
            constexpr int NREQ = NSLOT >= 12 ? 4 : (NSLOT >= 4 ? 2 : 1);
            constexpr int qw = q - NREQ, qb = q;
            if constexpr (qw >= 0 && qw < NREQ) {
                constexpr int per = (TN * NPL + NREQ - 1) / NREQ;
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
            if constexpr (qb >= 0 && qb < NREQ && s + 1 < KS) {
                constexpr int per = (2 * NPL + NREQ - 1) / NREQ;
#pragma unroll
                for (int u = qb * per; u < (qb + 1) * per && u < 2 * NPL; ++u) {
                    const int hb = u & 1, plane = u >> 1;
                    if (plane == 0) bh[(s + 1) % 3][hb] = ldb(hb, s + 1, 0);
                    else bl[(s + 1) % 3][hb] = ldb(hb, s + 1, 1);
                }
            }
            if constexpr (NM > 0) {
                static_for<0, NM>([&](auto J_) {
                    constexpr int j = decltype(J_)::value;
                    if constexpr ((j * NSLOT) / NM == q) micro(J_);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            asm volatile("" ::"v"(bh[s % 3][hb]));
            if constexpr (PREC == 3) asm volatile("" ::"v"(bl[s % 3][hb]));
        }
    });
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
}

}  // namespace fneus
