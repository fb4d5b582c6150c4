// Two-pass pipelined layers (p2_engine.h) with the stash of the training step: the forward chain of K2.
//
// The pass is the one of p2_engine.h (slot = 1 MFMA + one micro-step of vector work + one operand request); the vector work
// additionally produces what the later kernels of the step read:
//   * sigma'(z_l) as 16-bit fixed point, lane-private blocks (pp_engine.h sig_put8 / sig_get8): read by the reverse sweep
//     (sdf_grad_rev) and by both chains of K3;
//   * the B fragments it publishes in LDS also go to the h_l plane (hi; lo in the exact-gradient mode): the weight-gradient
//     GEMM's operand, invalid samples of a ragged tile as zeros (fneus_pp.h);
//   * the linear last layer: fp32 feature rows [n][256] (the colour network's input) and the feature plane.
// Every store is a BUFFER store (uniform descriptor + lane offset + uniform offset: no address arithmetic on the vector pipe);
// a descriptor's num_records bounds what may be written: 0 for a tile beyond the launch (its stores are dropped), the valid
// rows only for the fp32 rows of the last tile -- no branch sits inside the slot stream (LLVM sinks work into branches).
#pragma once
#include "p2_engine.h"

namespace fneus {

// MODE 0: inference (sigma' + fp32 outputs only), 1: training, bf16 planes, 3: training, hi + lo planes
struct P2St {                               // uniform per pass.  The vector work runs on a PAIR of sample tiles (2 k, 2 k + 1): the
    __amdgpu_buffer_rsrc_t sig;             // launch holds both or neither (pp_tiles is even), their blocks lie a fixed stride
    __amdgpu_buffer_rsrc_t hi, lo;          // apart, so one descriptor per kind serves the pair (tile hb at + hb * stride):
    bool vmask[2];                          //   sig: sigma' blocks (ACT 4 / 5, stride 8 blocks) | fp32 feature rows (ACT 6)
};                                          //   hi, lo: plane blocks (h_l or the feature plane, stride 1 block)
                                            //   vmask: per lane, the sample of tile hb is inside the launch

#ifndef FNEUS_P2_TRAIN_GS
#define FNEUS_P2_TRAIN_GS 2
#endif
// buffer descriptor of an output region (raw buffer: stride 0, num_records = bytes that may be written)
FN_DEV __amdgpu_buffer_rsrc_t p2_out_rsrc(unsigned char* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000);
}

// 16-byte buffer store, NT: non-temporal.  SEEN ON gfx950 (tools/experiments/r03/k2_p2_dbg.py): hipcc reuses a store's data
// registers for the next store's data right away (v_cndmask x 4, buffer_store, v_cndmask x 4 into the same registers,
// buffer_store) and the first store then writes the second one's words -- the hi plane held lo words or sigma' words, not
// repeatably.  LLVM's hazard table exempts buffer stores with an SGPR offset from the wait states between a > 64-bit store and a
// vector write of its data registers.  The empty-looking asm READS the data registers behind the store: they stay intact up to
// it, and it holds the wait states.  (The store itself stays a builtin: the compiler has to see its SGPR operands -- an
// offset restored from a spill by v_readlane needs wait states of its own.)
template <bool NT>
FN_DEV void p2_store128(p2_u32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
#ifndef FNEUS_DBG_NO_P2STORE            // (timing experiments only: what the stash stores of a two-pass kernel cost)
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, soff, NT ? 2 : 0);
#endif
    asm volatile("s_nop 1" ::"v"(v) : "memory");
}

// ACT 0: none.  4: softplus + sigma' -> B fragments in LDS (+ plane), sigma' block.  5: 4 + partial dot product with cw (the sdf
// row of the linear last layer).  6: linear output -> fp32 rows (+ plane).
template <int PREC, int KS, int NT_TOTAL, int LMAP, int ACT, int MODE>
FN_DEV void p2_pass_st(const unsigned char* __restrict__ blob, __amdgpu_buffer_rsrc_t rsrc, uint32_t off_hi, uint32_t off_lo,
                       P2Prime<FNEUS_P2_DEPTH, 1>& pr, const P2Next& nx, unsigned char* lds, int lane, int t0,
                       f32x16 (&accM)[1][2], int hbM, f32x16 (&accV)[1][2], int hbV, int tnV, const f32x16 (&cw)[1],
                       float (&dot)[2], const P2St& so, unsigned voff_even, unsigned voff_odd, unsigned voff_row) {
    // voff_row: byte offset of this lane's row (and lane half) in the fp32 rows of tile 0 of the pair
    constexpr int TN = 1;
    constexpr int NPL = PREC == 3 ? 2 : 1;
    constexpr int D = FNEUS_P2_DEPTH;
    constexpr int NV = TN * 32;
    constexpr bool TRAIN = MODE != 0;
    // ACT 6 (the linear output: the feature vector): hi + lo planes in EVERY training mode of the parity arithmetic -- they are the colour
    // network's input (round 6: it reads the planes, the fp32 rows are optional); the h_l planes follow the gradient precision
    constexpr bool LO = ACT == 6 ? (TRAIN && PREC == 3) : (MODE == 3 && PREC == 3);
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
    unsigned char* dump = lds + kP2Dump + lane * 16;
    bf16x8 bh[3][2], bl[3][2];
    auto ldb = [&](int hb, int slot, int plane) { return *reinterpret_cast<const bf16x8*>(flM + hb * kP2Half + (slot * NPL + plane) * kFragBytes); };
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        bh[0][hb] = ldb(hb, p2_slot<LMAP>(0), 0);
        if constexpr (PREC == 3) bl[0][hb] = ldb(hb, p2_slot<LMAP>(0), 1);
    }
    p2_prime_bias<PREC, D, TN>(pr, blob, lane, t0, nx);
    typedef __attribute__((ext_vector_type(2))) __bf16 p2_bf16x2;
    uint32_t phw[4], plw[4], psw[4];        // fragment half being assembled: hi words, lo words, sigma' words
    f32x16 (&vv)[TN][2] = accV;
    const unsigned lane16 = (unsigned)lane * 16u;
    static_for<0, KS>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        constexpr int NSLOT = (PREC == 3 ? 6 : 2) * TN;
        constexpr int NP = NV / 2;
        constexpr int GS = FNEUS_P2_TRAIN_GS;   // pairs per group: the phases run group by group (a k-step of layer 0 has 5-6 pairs:
                                            // phase-major over all of them keeps ~70 temporaries alive)
        float ve[2 * GS], vm[2 * GS], vl[2 * GS], vr[2 * GS], vq[2 * GS];
        constexpr int p0 = (s * NP + KS - 1) / KS;
        constexpr int np = ((s + 1) * NP + KS - 1) / KS - p0;
        // micro-steps of a k-step with np value pairs.  ACT 4 / 5, per group of gs <= GS pairs: A(v) B(v) R(v) for its 2 gs
        // values, then C(p) H(p) pair by pair, then S(p): 9 gs; 9 np in all.  ACT 6: C(p) H(p) pair by pair: 2 np.
        constexpr int NM = ACT == 6 ? 2 * np : 9 * np;
        auto plane_store = [&](auto HB_, int ks) {            // H: the fragment half just assembled -> plane block(s)
            constexpr int hb = decltype(HB_)::value;
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
        };
        auto micro = [&](auto J_) {
            constexpr int j = decltype(J_)::value;
            if constexpr (ACT == 6) {
                constexpr int pi = j >> 1;
                constexpr int v = 2 * (p0 + pi);
                constexpr int g = v >> 3, e = v & 7;
                constexpr int i = g >> 2, hb = (g >> 1) & 1, sh = g & 1;
                if constexpr ((j & 1) == 0) {                       // C6: split / pack (training); fp32 row store every 4 values
                    if constexpr (TRAIN) {
                        const float y0 = vv[i][hb][8 * sh + e], y1 = vv[i][hb][8 * sh + e + 1];
                        p2_bf16x2 hv = {(__bf16)y0, (__bf16)y1};
                        const uint32_t pk = __builtin_bit_cast(uint32_t, hv);
                        phw[e >> 1] = pk;
                        if constexpr (LO) {
                            const float h0f = __builtin_bit_cast(float, pk << 16), h1f = __builtin_bit_cast(float, pk & 0xffff0000u);
                            p2_bf16x2 lv = {(__bf16)(y0 - h0f), (__bf16)(y1 - h1f)};
                            plw[e >> 1] = __builtin_bit_cast(uint32_t, lv);
                            asm volatile("" : "+v"(phw[e >> 1]), "+v"(plw[e >> 1]));
                        } else {
                            asm volatile("" : "+v"(phw[e >> 1]));
                        }
                    }
                    if constexpr ((e & 3) == 2) {                   // registers 8 sh + e - 2 .. + 1 = 4 consecutive features
                        constexpr int reg = 8 * sh + e - 2;
                        f32x4 fv = {vv[i][hb][reg], vv[i][hb][reg + 1], vv[i][hb][reg + 2], vv[i][hb][reg + 3]};
                        // feature 32 (t0 + i) + 8 (reg >> 2) + 4 h + (reg & 3); row and 4 h sit in voff_row
                        p2_store128<false>(__builtin_bit_cast(p2_u32x4, fv), so.sig, (voff_row + hb * 32768u), (32 * (t0 + i) + 8 * (reg >> 2)) * 4);
                    }
                } else {                                            // H
                    if constexpr (TRAIN && e == 6) plane_store(std::integral_constant<int, hb>{}, 2 * (t0 + i) + sh);
                }
            } else {
                constexpr int gi = j / (9 * GS), jj = j % (9 * GS);
                constexpr int gp0 = gi * GS;                                    // first pair of the group (within the k-step)
                constexpr int gs = np - gp0 < GS ? np - gp0 : GS;
                if constexpr (jj < 6 * gs) {
                    constexpr int phase = jj / (2 * gs), vi = jj % (2 * gs);
                    constexpr int v = 2 * (p0 + gp0) + vi;
                    constexpr int g = v >> 3, e = v & 7;
                    constexpr int i = g >> 2, hb = (g >> 1) & 1, sh = g & 1;
                    if constexpr (phase == 0) {
                        const float z = vv[i][hb][8 * sh + e];
                        ve[vi] = fast_exp2(-fabsf(z) * (kBeta * kLog2e));
                        asm volatile("v_max_f32 %0, 0, %2" : "=v"(vm[vi]), "+v"(ve[vi]) : "v"(z));
                    } else if constexpr (phase == 1) {
                        vr[vi] = 1.0f + ve[vi];
                        vl[vi] = fast_log2(vr[vi]);
                        asm volatile("" : "+v"(vl[vi]), "+v"(vr[vi]));
                    } else {                                            // R: 1 / (1 + e) and e / (1 + e)
                        vr[vi] = fast_rcp(vr[vi]);
                        vq[vi] = ve[vi] * vr[vi];
                        asm volatile("" : "+v"(vr[vi]), "+v"(vq[vi]));
                    }
                } else if constexpr (jj < 8 * gs) {
                    constexpr int pi = (jj - 6 * gs) >> 1;                      // pair within the group
                    constexpr int v = 2 * (p0 + gp0 + pi);
                    constexpr int g = v >> 3, e = v & 7;
                    constexpr int i = g >> 2, hb = (g >> 1) & 1, sh = g & 1;
                    if constexpr (((jj - 6 * gs) & 1) == 0) {           // C
                        const float y0 = fmaf(vl[2 * pi], kLn2 / kBeta, vm[2 * pi]);
                        const float y1 = fmaf(vl[2 * pi + 1], kLn2 / kBeta, vm[2 * pi + 1]);
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
                        if constexpr (ACT == 5) {
                            dot[hb] = fmaf(y0, cw[i][8 * sh + e], dot[hb]);
                            dot[hb] = fmaf(y1, cw[i][8 * sh + e + 1], dot[hb]);
                            asm volatile("" : "+v"(dot[hb]));
                        }
                        if constexpr (e == 6) {
                            const int ks = 2 * (t0 + i) + sh;
                            unsigned char* dst = i < tnV ? flV + hb * kP2Half + (ks * NPL) * kFragBytes : dump;
                            *reinterpret_cast<p2_u32x4*>(dst) = p2_u32x4{phw[0], phw[1], phw[2], phw[3]};
                            if constexpr (PREC == 3) *reinterpret_cast<p2_u32x4*>(dst + kFragBytes) = p2_u32x4{plw[0], plw[1], plw[2], plw[3]};
                        }
                    } else {                                            // H
                        if constexpr (TRAIN && e == 6) plane_store(std::integral_constant<int, hb>{}, 2 * (t0 + i) + sh);
                    }
                } else {                                                // S: sigma' = z > 0 ? 1 / (1 + e) : e / (1 + e), two per word
                    constexpr int pi = jj - 8 * gs;
                    constexpr int v = 2 * (p0 + gp0 + pi);
                    constexpr int g = v >> 3, e = v & 7;
                    constexpr int i = g >> 2, hb = (g >> 1) & 1, sh = g & 1;
                    const float s0 = vm[2 * pi] > 0.0f ? vr[2 * pi] : vq[2 * pi];
                    const float s1 = vm[2 * pi + 1] > 0.0f ? vr[2 * pi + 1] : vq[2 * pi + 1];
                    psw[e >> 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_u16(s0, s1));
                    asm volatile("" : "+v"(psw[e >> 1]));
                    if constexpr (e == 6)
                        p2_store128<false>(p2_u32x4{psw[0], psw[1], psw[2], psw[3]}, so.sig, lane16, hb * 8 * (int)kPPBlock + (2 * (t0 + i) + sh) * kFragBytes);
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
                    if (plane == 0) bh[(s + 1) % 3][hb] = ldb(hb, p2_slot<LMAP>(s + 1), 0);
                    else bl[(s + 1) % 3][hb] = ldb(hb, p2_slot<LMAP>(s + 1), 1);
                }
            }
            if constexpr (ACT != 0) {
                static_for<0, NM>([&](auto J_) {
                    constexpr int j = decltype(J_)::value;
                    if constexpr ((j * NSLOT) / NM == q) micro(J_);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // HAZARD (p2_engine.h): the B operands of this k-step stay live to its end
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            asm volatile("" ::"v"(bh[s % 3][hb]));
            if constexpr (PREC == 3) asm volatile("" ::"v"(bl[s % 3][hb]));
        }
    });
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
}

// the vector work of ACT 6 alone (the last unit's second set, behind the last MFMA pass of a launch)
template <int PREC, int MODE>
FN_DEV void p2_linear_out_only(int lane, int t0, f32x16 (&accV)[1][2], const P2St& so, unsigned voff_even, unsigned voff_odd,
                               unsigned voff_row) {
    constexpr bool TRAIN = MODE != 0;
    constexpr bool LO = TRAIN && PREC == 3;         // the feature planes are hi + lo in every training mode (as ACT 6 of p2_pass_st)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 fv = {accV[0][hb][4 * g], accV[0][hb][4 * g + 1], accV[0][hb][4 * g + 2], accV[0][hb][4 * g + 3]};
            p2_store128<false>(__builtin_bit_cast(p2_u32x4, fv), so.sig, (voff_row + hb * 32768u), (32 * t0 + 8 * g) * 4);
        }
        if constexpr (TRAIN) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                bf16x8 ph, pl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = so.vmask[hb] ? accV[0][hb][8 * sh + e] : 0.0f;
                    __bf16 a, b2;
                    split_bf16(y, a, b2);
                    ph[e] = a;
                    pl[e] = b2;
                }
                const int ks = 2 * t0 + sh;
                const unsigned vo = (ks & 1) ? voff_odd : voff_even;
                p2_store128<true>(__builtin_bit_cast(p2_u32x4, ph), so.hi, vo, hb * (int)kPPBlock + ks * kFragBytes);
                if constexpr (LO)
                    p2_store128<true>(__builtin_bit_cast(p2_u32x4, pl), so.lo, vo, hb * (int)kPPBlock + ks * kFragBytes);
            }
        }
    }
}

}  // namespace fneus
