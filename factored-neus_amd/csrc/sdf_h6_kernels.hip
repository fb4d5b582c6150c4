// K1 (SDFNetwork.sdf, reference models/fields.py:93-95 via renderer.py:199, 430) with "h6" products (h6_engine.h): ONE fp16 MFMA
// per 16 k for hi.hi and two block-scaled fp6 MFMAs per 64 k for the cross terms -- 1.5 MFMA-times per product instead of the 3 of
// the shipped parity arithmetic.  Round-5 prototype behind FNEUS_K1_H6=1 (fneus/ops.py); the two-pass schedule of
// sdf_p2_kernels.hip on 4 waves x 2 output tiles (a lane's fp6 block is the pair of tiles its wave owns).
//   fneus_h6_pack      : bf3 blob (fwd_hi + fwd_lo = W to 17 bits) -> h6 blob of the forward chain
//   fneus_sdf_fwd_h6   : the kernel
#include <stdlib.h>
#include "h6_engine.h"
#ifndef FNEUS_H6_GS
#define FNEUS_H6_GS 8
#endif
#include "fneus_kernels.h"

namespace fneus {

// one wave per (layer, block, tile): the block's four hi / lo fragment pairs -> fp16 hi fragments + the two fp6 operands
__global__ void __launch_bounds__(64) h6_pack_kernel(const unsigned char* __restrict__ blob, unsigned char* __restrict__ hblob) {
    const int lane = threadIdx.x;
    int u = blockIdx.x, l = 0;
    for (; l < 8; ++l) {
        const int n = h6_blocks(h6_kind(l)) * kSdfGeom[l].ntf;
        if (u < n) break;
        u -= n;
    }
    if (l >= 8) return;
    const int kind = h6_kind(l), nt = kSdfGeom[l].ntf;
    const int b = u / nt, t = u % nt;
    f32x16 w0, w1, wl0, wl1;          // element jj = 8 s + j: w0 = jj 0..15, w1 = 16..31
    float m = 0.0f, ml = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int ks = h6_ks(kind, b, s);
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) hi[j] = lo[j] = (__bf16)0.0f;
        if (ks >= 0) {
            hi = *reinterpret_cast<const bf16x8*>(blob + kSdfLayout.L[l].fwd_hi + (size_t)((ks * nt + t) * 64 + lane) * 16);
            lo = *reinterpret_cast<const bf16x8*>(blob + kSdfLayout.L[l].fwd_lo + (size_t)((ks * nt + t) * 64 + lane) * 16);
        }
        f16x8 h16;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float w = (float)hi[j] + (float)lo[j];
            const _Float16 wh = (_Float16)w;          // (subnormal fp16 hi parts are multiplied as they are: measured, r05/h6_dbg.sh)
            const float wl = w - (float)wh;
            h16[j] = wh;
            const int jj = 8 * s + j;
            if (jj < 16) {
                w0[jj] = w;
                wl0[jj] = wl;
            } else {
                w1[jj - 16] = w;
                wl1[jj - 16] = wl;
            }
            m = fmaxf(m, fabsf(w));
            ml = fmaxf(ml, fabsf(wl));
        }
        if (ks >= 0) *reinterpret_cast<f16x8*>(hblob + kH6Layout.L[l].hi + (size_t)((ks * nt + t) * 64 + lane) * 16) = h16;
    }
    // Q(W) in the order of xl6 (element 2 e + i = (tile i, register e) = jj 16 i + e): 2xpk16(a, b) emits a[e], b[e] alternately
    const uint32_t e6 = __builtin_bit_cast(uint32_t, fmaxf(m, 1.0e-30f)) >> 23, el6 = __builtin_bit_cast(uint32_t, fmaxf(ml, 1.0e-30f)) >> 23;
    const u32x6 q = h6_cvt_slow(w0, w1, __builtin_bit_cast(float, (e6 - 2u) << 23));
    // Q(Wl) in the same order (x6 is converted from the fp32 values of the two tiles like xl6)
    const u32x6 ql = h6_cvt_slow(wl0, wl1, __builtin_bit_cast(float, (el6 - 2u) << 23));
    unsigned char* rec = hblob + kH6Layout.L[l].rec + (size_t)(b * nt + t) * kH6Rec + lane * 16;
    const uint32_t sc = (e6 - 2u) | ((el6 - 2u) << 8);
    *reinterpret_cast<p2_u32x4*>(rec + kH6RecW) = p2_u32x4{q[0], q[1], q[2], q[3]};
    *reinterpret_cast<p2_u32x4*>(rec + kH6RecW + 1024) = p2_u32x4{q[4], q[5], sc, 0u};
    *reinterpret_cast<p2_u32x4*>(rec + kH6RecL) = p2_u32x4{ql[0], ql[1], ql[2], ql[3]};
    *reinterpret_cast<p2_u32x4*>(rec + kH6RecL + 1024) = p2_u32x4{ql[4], ql[5], sc, 0u};
}

// What a pass needs before its first MFMA is requested by the pass before it:
//   * the bias: loaded straight into accV[.][0] -- the NEXT pass's accumulators of its first sample tile -- once this pass's vector
//     work is through with them (half way); the next pass copies it to its second sample tile.  No registers of its own;
//   * its first three k-steps of fp16 hi fragments: the ring H6Whi runs on across passes (PH = ring position of a pass's k-step 0:
//     the passes of a unit have 3 3 16 .. 16 17 17 16 .. 16 k-steps = 232 = 0 mod 4, so PH is a compile-time constant per pass);
//   * the fp6 weights of its first block in the one fp6 buffer, requested behind this pass's last fp6 MFMA.
struct H6Next {
    uint32_t off_hi, off_rec, off_bias;
};
struct H6Whi {
    f16x8 hi[4][2];          // ring of four k-steps x the wave's two tiles: k-step g of the stream of passes sits in g % 4, requested
};                           // three k-steps ahead (also across passes: PH = position of a pass's k-step 0)
struct H6Wq {
    i32x8 w6[2], l6[2];      // Q(W), Q(Wl) of the two tiles: dwords 0..5 the codes, 6 the scale bytes
};
// every weight load: uniform descriptor + the ONE per-lane offset lane * 16 (lane * 4 for the scales) + a scalar offset that
// carries layer, block, tile and the wave's first tile
struct H6Lane {
    unsigned l16, hb64;                      // lane * 16, (lane >> 5) * 64 (the bias rows of the lane half)
    unsigned t0_frag, t0_rec, t0_bias;       // wave-uniform: t0 * 1024, t0 * kH6Rec, t0 * 128
};
template <int NT_TOTAL>
FN_DEV f16x8 h6_whi_load(__amdgpu_buffer_rsrc_t rsrc, uint32_t off_hi, int ks, int i, const H6Lane& ln) {
#ifdef FNEUS_H6_NO_WEIGHTS               // timing experiments only
    f16x8 t;
    for (int e = 0; e < 8; ++e) t[e] = (_Float16)(0.001f * (float)(ln.l16 + e));
    asm volatile("" : "+v"(t));
    return t;
#endif
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)ln.l16, (int)(off_hi + ln.t0_frag + (uint32_t)((ks * NT_TOTAL + i) * kFragBytes)), 0));
}
template <int NT_TOTAL>
FN_DEV void h6_wq_load(H6Wq& w, __amdgpu_buffer_rsrc_t rsrc, uint32_t off_rec, int b, const H6Lane& ln) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t rec = off_rec + ln.t0_rec + (uint32_t)((b * NT_TOTAL + i) * kH6Rec);
        w.w6[i] = h6_op8(__builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)ln.l16, (int)(rec + kH6RecW), 0),
                         __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)ln.l16, (int)(rec + kH6RecW + 1024), 0));
        w.l6[i] = h6_op8(__builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)ln.l16, (int)(rec + kH6RecL), 0),
                         __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)ln.l16, (int)(rec + kH6RecL + 1024), 0));
    }
}
// the bias of the wave's two tiles in accumulator layout ([t][h][16] fp32) -> acc[.][0]
FN_DEV void h6_bias_load(__amdgpu_buffer_rsrc_t brsrc, uint32_t off_bias, const H6Lane& ln, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, (int)ln.hb64, (int)(off_bias + ln.t0_bias + (uint32_t)(i * 128 + g * 16)), 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][0][4 * g + e] = v[e];
        }
}

// One pass: MFMAs of layer (KIND, NT_TOTAL) on the sample tiles hbM, hbM + 1 (accM; accM[.][0] holds the bias on entry) with,
// inside their stream, the vector work on accV = the previous pass's accumulators of tiles hbV, hbV + 1:
//   ACT 1: softplus -> fp16 hi fragments + fp6 blocks of the next layer's input in LDS (block = this wave's tile pair);
//          MASK7: the values of tiles >= tnV are zeros (layer 3 has 7 tiles);
//   ACT 2: softplus -> partial dot product with cw (the sdf row of the linear last layer); 0: none.
// Per block of 64 k: 16 slots hi.hi (4 k-steps x 2 tiles x 2 sample tiles), 8 slots of fp6 cross terms; a slot = one MFMA + its
// share of the vector work + operand requests, fenced by sched_barrier like p2_pass.
template <int KIND, int NT_TOTAL, int NKIND, int NNT, int ACT, int PH, int HBM, bool MASK7 = false>
FN_DEV void h6_pass(__amdgpu_buffer_rsrc_t brsrc, __amdgpu_buffer_rsrc_t rsrc, uint32_t off_hi, uint32_t off_rec,
                    H6Whi& wr, H6Wq& wq, const H6Next& nx, unsigned char* const (&tile16)[4], const H6Lane& ln, int wave,
                    f32x16 (&accM)[2][2], f32x16 (&accV)[2][2], int tnV, const f32x16 (&cw)[2], float (&dot)[2]) {
    constexpr int HBV = HBM ^ 2;             // tile16[k] = LDS address of sample tile k + lane * 16
    asm volatile("" : "+s"(off_hi), "+s"(off_rec));      // (offsets formed here, by scalar adds, not hoisted out of the unit loop and spilled)
    constexpr int NB = h6_blocks(KIND);
    constexpr int NSLOT = NB * 24;
#pragma unroll
    for (int i = 0; i < 2; ++i) accM[i][1] = accM[i][0];
    // activations: hi fragments one k-step ahead through a ring of three (p2_engine.h: an LDS load must not land in registers
    // that a queued MFMA still reads), the fp6 operands of a block behind its slot 8 (their last readers, the fp6 MFMAs of the
    // block before, have left the pipe by then)
    f16x8 bh[3][2];
    i32x8 bx6[2], bxl6[2];       // (dword 6: the scale bytes)
    auto ld_bh = [&](int slot, int hb) { return *reinterpret_cast<const f16x8*>(tile16[HBM + hb] + kH6Hi + slot * kFragBytes); };
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) bh[0][hb] = ld_bh(h6_slot(KIND, h6_ks(KIND, 0, 0)), hb);
    if constexpr (ACT == 0) h6_bias_load(brsrc, nx.off_bias, ln, accV);       // (no vector work: accV is free from the start)
    // ---- the vector work as a list of micro-steps (constant indices only); per sample tile hb:
    //   groups of 8 values: A exp2 / max, B log2, C y (in place);  M block maximum;  S scales;  H fp16 pairs;  L lo parts (in
    //   place);  Q the two conversions;  T stores;  behind sample tile 0: the next pass's bias into its registers
    // One micro-step = ONE vector instruction, and an instruction's result is used GS micro-steps later at the earliest: a wave alone
    // on its SIMD issues in order and stalls on every dependent instruction (tools/experiments/r05/h6_parts.sh: with the softplus as
    // three micro-steps of 2-4 dependent instructions the vector work alone took 130 of the launch's 151 us, 13 cycles per
    // instruction).  Per group of GS values the phases are  R z <- accumulator | N t = -|z| c | E e = exp2 t | X m = max(z, 0) |
    // A s = 1 + e | G L = log2 s | F y = m + L c'.
    constexpr int GS = FNEUS_H6_GS, NG = 32 / GS, SP = 7 * 32;   // softplus: 7 phases x 32 values
    constexpr int OPS_HB = ACT == 1 ? (SP + 16 + 1 + 16 + 32 + 2 + 8 + 1) : (ACT == 2 ? SP + 32 : 0);
    constexpr int NM = ACT == 0 ? 0 : 2 * OPS_HB + 1;
    float vz[GS], ve[GS], vm[GS], bm = 0.0f, bm2 = 0.0f, sx = 1.0f, sl = 1.0f;
    float yd[ACT == 2 ? 32 : 1];         // ACT 2: the values of a sample tile, multiplied with the sdf row behind the softplus
    H6Block ob;
    auto micro = [&](auto J_) {
        constexpr int j0_ = decltype(J_)::value;
        if constexpr (j0_ == OPS_HB) {
            h6_bias_load(brsrc, nx.off_bias, ln, accV);
        } else {
        constexpr int j = j0_ < OPS_HB ? j0_ : j0_ - 1;
        constexpr int hb = j / OPS_HB, k = j % OPS_HB;
        if constexpr (k < SP) {
            constexpr int g = k / (7 * GS), ph = (k % (7 * GS)) / GS, q = k % GS;
            constexpr int v = GS * g + q, i = v >> 4, e = v & 15;
            static_assert(NG * 7 * GS == SP, "32 values");
            if constexpr (ph == 0) {
                vz[q] = accV[i][hb][e];
                asm volatile("" : "+v"(vz[q]));
            } else if constexpr (ph == 1) {
                asm volatile("v_mul_f32 %0, %1, |%2|" : "=v"(ve[q]) : "v"(-kBeta * kLog2e), "v"(vz[q]));
            } else if constexpr (ph == 2) {
                ve[q] = fast_exp2(ve[q]);
                asm volatile("" : "+v"(ve[q]));
            } else if constexpr (ph == 3) {
                asm volatile("v_max_f32 %0, 0, %1" : "=v"(vm[q]) : "v"(vz[q]));
            } else if constexpr (ph == 4) {
                asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(ve[q]));
            } else if constexpr (ph == 5) {
                ve[q] = fast_log2(ve[q]);
                asm volatile("" : "+v"(ve[q]));
            } else {
                if constexpr (ACT == 1) {
                    float y;
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(ve[q]), "v"(kLn2 / kBeta), "v"(vm[q]));
                    if constexpr (MASK7) y = i < tnV ? y : 0.0f;      // (the eighth tile contributes nothing, also not to the scale)
                    accV[i][hb][e] = y;
                    asm volatile("" : "+v"(accV[i][hb][e]));
                } else {
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(yd[v]) : "v"(ve[q]), "v"(kLn2 / kBeta), "v"(vm[q]));
                }
            }
        } else if constexpr (ACT == 2) {                 // the dot product with the sdf row: two chains per sample tile would be
            constexpr int v = k - SP, i = v >> 4, e = v & 15;      // better still; 32 dependent fmas are 32 x 8 cycles per tile
            dot[hb] = fmaf(yd[v], cw[i][e], dot[hb]);
            asm volatile("" : "+v"(dot[hb]));
        } else if constexpr (ACT == 1) {
            constexpr int k2 = k - SP;
            if constexpr (k2 < 16) {                     // M (softplus >= 0: no absolute values)
                if constexpr (k2 == 0) bm = 1.0e-30f;      // (two chains: a v_max3 waits for the one before it)
                if constexpr (k2 == 1) bm2 = 1.0e-30f;
                if constexpr ((k2 & 1) == 0) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(bm) : "v"(accV[0][hb][k2]), "v"(accV[1][hb][k2]));
                else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(bm2) : "v"(accV[0][hb][k2]), "v"(accV[1][hb][k2]));
            } else if constexpr (k2 == 16) {             // S
                const uint32_t ex = __builtin_bit_cast(uint32_t, fmaxf(bm, bm2)) >> 23;       // (>= 27: both scale bytes stay positive)
                sx = __builtin_bit_cast(float, (ex - 2u) << 23);
                sl = __builtin_bit_cast(float, (ex - 14u) << 23);
                ob.sc = (ex - 2u) | ((ex - 14u) << 8);
                asm volatile("" : "+v"(sx), "+v"(sl), "+v"(ob.sc));
            } else if constexpr (k2 == 17) {             // x6 = Q(y) from the fp32 values of the two tiles (element 2 e + i, like xl6)
                ob.q = h6_cvt(accV[0][hb], accV[1][hb], sx);
            } else if constexpr (k2 < 34) {              // H: values 2 p, 2 p + 1 -> one packed register (round to nearest even)
                constexpr int pp = k2 - 18, v = 2 * pp, i = v >> 4, e = v & 15;
                uint32_t pk;
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(accV[i][hb][e]), "v"(accV[i][hb][e + 1]));
                const f16x2 h2 = __builtin_bit_cast(f16x2, pk);
                ob.hh[v] = h2[0];
                ob.hh[v + 1] = h2[1];
            } else if constexpr (k2 < 66) {              // L: lo = y - hi in ONE instruction (the fp16 operand widened inside the fma)
                constexpr int v = k2 - 34, i = v >> 4, e = v & 15;
                f16x2 h2 = {ob.hh[v & ~1], ob.hh[(v & ~1) + 1]};
                const uint32_t pk = __builtin_bit_cast(uint32_t, h2);
                if constexpr ((v & 1) == 0)
                    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(accV[i][hb][e]) : "v"(pk));
                else
                    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(accV[i][hb][e]) : "v"(pk));
            } else if constexpr (k2 == 66) {
                ob.ql = h6_cvt(accV[0][hb], accV[1][hb], sl);
            } else {                                     // T: 8 stores, release
                constexpr int st = k2 - 67;
                unsigned char* tile = tile16[HBV + hb];
                unsigned char* qb = tile + kH6Q + wave * 4096;          // (block = this wave's tile pair)
                if constexpr (st < 4) {
                    f16x8 f;
#pragma unroll
                    for (int jx = 0; jx < 8; ++jx) f[jx] = ob.hh[8 * st + jx];
                    *reinterpret_cast<f16x8*>(tile + kH6Hi + (4 * wave + st) * kFragBytes) = f;
                } else if constexpr (st == 4) {
                    *reinterpret_cast<p2_u32x4*>(qb + kH6QX) = p2_u32x4{ob.q[0], ob.q[1], ob.q[2], ob.q[3]};
                } else if constexpr (st == 5) {
                    *reinterpret_cast<p2_u32x4*>(qb + kH6QX + 1024) = p2_u32x4{ob.q[4], ob.q[5], ob.sc, 0u};
                } else if constexpr (st == 6) {
                    *reinterpret_cast<p2_u32x4*>(qb + kH6QL) = p2_u32x4{ob.ql[0], ob.ql[1], ob.ql[2], ob.ql[3]};
                } else if constexpr (st == 7) {
                    *reinterpret_cast<p2_u32x4*>(qb + kH6QL + 1024) = p2_u32x4{ob.ql[4], ob.ql[5], ob.sc, 0u};
                } else {                                 // (the conversions' sources and scales are free from here on)
                    h6_cvt_release(accV[0][hb], accV[1][hb], sx);
                    h6_cvt_release(accV[0][hb], accV[1][hb], sl);
                }
            }
        }
        }
    };
    static_for<0, NB>([&](auto B_) {
        constexpr int b = decltype(B_)::value;
        // running index of this block's first k-step among the pass's k-steps (ring position of the hi fragments)
        constexpr int kk0 = KIND == 2 ? (b < 4 ? 4 * b : 14) : 4 * b;
        constexpr int ns = KIND == 1 ? 3 : (KIND == 2 ? (b < 3 ? 4 : (b == 3 ? 2 : 3)) : 4);      // k-steps of the block
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 24>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            constexpr int slot = b * 24 + q;
            if constexpr (q < 16) {
                constexpr int s = q >> 2, r = q & 3, i = r >> 1, hb = r & 1;
                if constexpr (s < ns) {
                    accM[i][hb] = mfma32h(wr.hi[(PH + kk0 + s) % 4][i], bh[(kk0 + s) % 3][hb], accM[i][hb]);
                    if constexpr (hb == 1) {             // (the tile's last MFMA of this k-step is out:) its fragment three k-steps on
                        constexpr int kn = kk0 + s + 3, KS = KIND == 1 ? 3 : (KIND == 2 ? 17 : 16);
                        if constexpr (kn < KS) wr.hi[(PH + kn) % 4][i] = h6_whi_load<NT_TOTAL>(rsrc, off_hi, kn, i, ln);
                        else wr.hi[(PH + kn) % 4][i] = h6_whi_load<NNT>(rsrc, nx.off_hi, kn - KS, i, ln);
                    }
                    if constexpr (r < 2) {               // hi fragment of the next k-step of the pass (of the next block behind the last)
                        constexpr bool more = s + 1 < ns || b + 1 < NB;
                        if constexpr (more) {
                            constexpr int ks_n = s + 1 < ns ? h6_ks(KIND, b, s + 1) : h6_ks(KIND, b + 1, 0);
                            bh[(kk0 + s + 1) % 3][r] = ld_bh(h6_slot(KIND, ks_n), r);
                        }
                    }
                }
            } else {
#ifdef FNEUS_H6_T2_FIRST
                constexpr int r = (q - 16) & 3, term = 1 - ((q - 16) >> 2), i = r >> 1, hb = r & 1;
#else
                constexpr int r = (q - 16) & 3, term = (q - 16) >> 2, i = r >> 1, hb = r & 1;
#endif
#ifndef FNEUS_H6_NO_T1                  // (debugging: the two cross terms one by one)
                if constexpr (term == 0) accM[i][hb] = mfma_fp6<0, 1>(wq.w6[i], bxl6[hb], accM[i][hb], wq.w6[i][6], bxl6[hb][6]);
#endif
#ifndef FNEUS_H6_NO_T2
#ifndef FNEUS_H6_T2A
#define FNEUS_H6_T2A 1
#define FNEUS_H6_T2B 0
#endif
                if constexpr (term == 1) accM[i][hb] = mfma_fp6<FNEUS_H6_T2A, FNEUS_H6_T2B>(wq.l6[i], bx6[hb], accM[i][hb], wq.l6[i][6], bx6[hb][6]);
#endif
            }
            if constexpr (q == 8 || q == 9) {            // fp6 operands of this block, sample tile q - 8
                constexpr int hb = q - 8, blk = h6_lds_block(KIND, b);
                const unsigned char* qb = tile16[HBM + hb] + kH6Q + blk * 4096;
                bx6[hb] = h6_op8(*reinterpret_cast<const p2_u32x4*>(qb + kH6QX), *reinterpret_cast<const p2_u32x4*>(qb + kH6QX + 1024));
                bxl6[hb] = h6_op8(*reinterpret_cast<const p2_u32x4*>(qb + kH6QL), *reinterpret_cast<const p2_u32x4*>(qb + kH6QL + 1024));
            }
#ifndef FNEUS_H6_NO_VALU
            if constexpr (NM > 0) {          // micro-steps j with floor(j NSLOT / NM) == slot
                constexpr int j0 = (slot * NM + NSLOT - 1) / NSLOT, j1 = ((slot + 1) * NM + NSLOT - 1) / NSLOT;
                static_for<j0, j1>(micro);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
        // the fp6 operands of this block stay live to its end; then the fp6 weights of the next block (or of the next pass's first
        // block) go into the one buffer: a load from L2 takes longer than the last fp6 MFMA needs to read its operands
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) asm volatile("" ::"v"(bx6[hb]), "v"(bxl6[hb]));
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(wq.w6[i]), "v"(wq.l6[i]));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (b + 1 < NB) h6_wq_load<NT_TOTAL>(wq, rsrc, off_rec, b + 1, ln);
        else h6_wq_load<NNT>(wq, rsrc, nx.off_rec, 0, ln);
    });
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
}

// vector work of a pass alone (tail of the last unit): softplus -> dot
FN_DEV void h6_valu_dot(f32x16 (&accV)[2][2], const f32x16 (&cw)[2], float (&dot)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int e = 0; e < 16; ++e) dot[hb] = fmaf(softplus100(accV[i][hb][e]), cw[i][e], dot[hb]);
}

// Work unit = 128 samples, pass schedule = sdf_fwd_p2_kernel's (A = sample tiles {0, 1}, B = {2, 3}):
//   L0.A || tail of the previous unit (act 7 B -> dot)      L0.B || act 0 A
//   Ll.A || act l-1 B                                       Ll.B || act l A                (l = 1..7; act 7 A -> dot)
// Units to evaluate: all of them, or -- `sel.list` given -- the listed ones (the 128-sample units of the marked rays, list and length in
// device memory: sdf_p2_kernels.hip k1_unit_list_kernel); the samples of the others get sel.fill from the workgroups up front.
struct H6UnitSel {
    const int32_t* list;
    const int32_t* n_list;
    const unsigned char* ray_mask;
    float fill;
};
__global__ void __launch_bounds__(256, 1) sdf_fwd_h6_kernel(const unsigned char* blob, const unsigned char* hblob, PointSrc src, long N,
                                                            float* __restrict__ sdf_out, H6UnitSel sel) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    constexpr int NW = 4;
    float* red = reinterpret_cast<float*>(lds_ + kH6LdsTotal);               // [4 tiles][NW waves][32 samples]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int t0 = 2 * wave, r = lane & 31, h = lane >> 5;
    constexpr auto& LY = kSdfLayout;
    constexpr auto& HY = kH6Layout;
    const long all_units = (N + 127) / 128;
    const long units = sel.list ? (long)__builtin_amdgcn_readfirstlane(*sel.n_list) : all_units;          // positions in the list
    auto U = [&](long i) { return sel.list ? (long)__builtin_amdgcn_readfirstlane(sel.list[i]) : i; };    // position -> unit
    if (sel.list) {
        for (long u = blockIdx.x; u < all_units; u += gridDim.x)
            if (sel.ray_mask[(u * 128) / src.m] == 0 && threadIdx.x < 128 && u * 128 + threadIdx.x < N) sdf_out[u * 128 + threadIdx.x] = sel.fill;
    }
    const H6Lane ln{(unsigned)lane * 16u, (unsigned)h * 64u, (unsigned)t0 * 1024u, (unsigned)t0 * (unsigned)kH6Rec,
                    (unsigned)t0 * 128u};
    unsigned char* const tile16[4] = {lds_ + lane * 16, lds_ + kH6Tile + lane * 16, lds_ + 2 * kH6Tile + lane * 16,
                                      lds_ + 3 * kH6Tile + lane * 16};
    auto encode = [&](long unit) {          // wave w: encoding of tile w of the unit -> hi slots 16..18, fp6 block 4
        const long n = (unit * 4 + wave) * 32 + r;
        const long nc = n < N ? n : N - 1;
        float x[3], pe[39], jc[39];
        load_point(src, nc, x);
        posenc<6, false>(x, pe, jc);
        f32x16 v0, v1;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
            const int f0 = phi(jj >> 3, 0, jj & 7), f1 = phi(jj >> 3, 1, jj & 7);
            const float a0 = (jj < 24 && f0 < 39) ? pe[f0 < 39 ? f0 : 0] : 0.0f;
            const float a1 = (jj < 24 && f1 < 39) ? pe[f1 < 39 ? f1 : 0] : 0.0f;
            const float val = h ? a1 : a0;
            if (jj < 16) v0[jj] = val;
            else v1[jj - 16] = val;
        }
        H6Block ob;
        h6_quant(v0, v1, ob, false);
        h6_store(lds_ + wave * kH6Tile + lane * 16, 16, 3, 4, ob);
    };
    auto put_dot = [&](float (&dot)[2], int hb0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float p = dot[k] + xor32(dot[k]);
            if (lane < 32) red[((hb0 + k) * NW + wave) * 32 + lane] = p;
            dot[k] = 0.0f;
        }
    };
    auto finish = [&](long unit, int hb0) {     // waves hb0, hb0 + 1: sdf of tile `wave` = b_8[0] + the waves' partial dot products
        if ((wave >> 1) == (hb0 >> 1) && lane < 32) {
            f32x16 b8[1];
            load_accvec<9, 8, 1>(blob, LY.L[8].bias, b8, lane);
            float s = b8[0][0];
#pragma unroll
            for (int k = 0; k < NW; ++k) s += red[(wave * NW + k) * 32 + lane];
            const long n = (unit * 4 + wave) * 32 + r;
            if (n < N) sdf_out[n] = s;
        }
    };
    f32x16 accA[2][2], accB[2][2], cw[2];
    float dot[2] = {0.0f, 0.0f};
    auto load_cw = [&]() { load_accvec<8, 0, 2>(blob, LY.extra, cw, lane, t0); };
    const __amdgpu_buffer_rsrc_t rsrc = p2_rsrc(hblob), brsrc = p2_rsrc(blob);
    auto next_of = [&](int l) { return H6Next{HY.L[l].hi, HY.L[l].rec, LY.L[l].bias}; };
    H6Whi wr;
    H6Wq wq;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) wr.hi[ks][i] = h6_whi_load<8>(rsrc, HY.L[0].hi, ks, i, ln);
    h6_wq_load<8>(wq, rsrc, HY.L[0].rec, 0, ln);
    h6_bias_load(brsrc, LY.L[0].bias, ln, accA);
    if ((long)blockIdx.x < units) encode(U(blockIdx.x));
    p2_barrier();
    bool first = true;
    long prev_unit = 0;
#ifdef FNEUS_H6_DUMP                    // debugging (one unit only): hi fragment slot 0 of sample tiles 0 and 2 behind every layer
    auto dump_layer = [&](int idx) {
        if (wave < 2) {
            const uint32_t* srcw = reinterpret_cast<const uint32_t*>(lds_ + (wave * 2) * kH6Tile + kH6Hi);
            uint32_t* dst = reinterpret_cast<uint32_t*>(sdf_out) + 256 + idx * 512 + wave * 256;
            for (int k = 0; k < 4; ++k) dst[k * 64 + lane] = srcw[k * 64 + lane];
        }
    };
#define H6_DUMP_LAYER(IDX) dump_layer(IDX);
#else
#define H6_DUMP_LAYER(IDX)
#endif
    // pass A: MFMAs of sample tiles {0, 1} (accA) || vector work on accB; pass B the other way round
#define H6_A(KIND, NT, NKIND, NNT, ACT, PH, M7, L_, NX, TNV) \
    h6_pass<KIND, NT, NKIND, NNT, ACT, PH, 0, M7>(brsrc, rsrc, HY.L[L_].hi, HY.L[L_].rec, wr, wq, NX, tile16, ln, wave, accA, accB, TNV, cw, dot)
#define H6_B(KIND, NT, NKIND, NNT, ACT, PH, M7, L_, NX, TNV) \
    h6_pass<KIND, NT, NKIND, NNT, ACT, PH, 2, M7>(brsrc, rsrc, HY.L[L_].hi, HY.L[L_].rec, wr, wq, NX, tile16, ln, wave, accB, accA, TNV, cw, dot)
    for (long pos = blockIdx.x; pos < units; pos += gridDim.x) {
        const long unit = U(pos);
        asm volatile("" : "+s"(blob), "+s"(hblob));
        // ---- layer 0 (one block: the encoding)
        if (!first) load_cw();
        if (first) H6_A(1, 8, 1, 8, 0, 0, false, 0, next_of(0), 2);
        else H6_A(1, 8, 1, 8, 2, 0, false, 0, next_of(0), 2);
        if (!first) put_dot(dot, 2);
        p2_barrier();
        if (!first) finish(prev_unit, 2);
        first = false;
        H6_B(1, 8, 0, 8, 1, 3, false, 0, next_of(1), 2);
        p2_barrier();
        const int tn3 = 7 - t0 < 2 ? 7 - t0 : 2;                        // layer 3 has 7 tiles: its last wave publishes one fewer
        // layers 1..4: the ring stands at 2 (3 + 3 k-steps of layer 0, 16 per pass behind them); 4.B at 3; 5..7 at 0.
        // Every layer is written out.  (With `#pragma unroll 1` loops over the layers the sample tiles {0, 1} came out NaN from layer 6
        // on -- the pass behind the loop's back edge, and only through the fp6 term Q(Wl) Q(x) -- while the unrolled code is right:
        // tools/experiments/r05/h6_dump.py; cause not found, `#pragma unroll` alone is refused by the optimizer in some builds.)
#define H6_LAYER(L_, PH_, NK_, NN_)                                                     \
        H6_A(0, 8, 0, 8, 1, PH_, false, L_, next_of(L_), 2);                            \
        p2_barrier();                                                                   \
        H6_B(0, 8, NK_, NN_, 1, PH_, false, L_, next_of(L_ + 1), 2);                    \
        p2_barrier();                                                                   \
        H6_DUMP_LAYER(L_)
        H6_LAYER(1, 2, 0, 8)
        H6_LAYER(2, 2, 0, 7)
        H6_A(0, 7, 0, 7, 1, 2, false, 3, next_of(3), 2);
        p2_barrier();
        H6_B(0, 7, 2, 8, 1, 2, true, 3, next_of(4), tn3);
        p2_barrier();
        H6_DUMP_LAYER(3)
        H6_A(2, 8, 2, 8, 1, 2, true, 4, next_of(4), tn3);
        p2_barrier();
        H6_B(2, 8, 0, 8, 1, 3, false, 4, next_of(5), 2);
        p2_barrier();
        H6_DUMP_LAYER(4)
        H6_A(0, 8, 0, 8, 1, 0, false, 5, next_of(5), 2);
        p2_barrier();
        H6_B(0, 8, 0, 8, 1, 0, false, 5, next_of(6), 2);
        if (pos + gridDim.x < units) encode(U(pos + gridDim.x));        // the encoding's slots are free behind layer 4
        p2_barrier();
        H6_DUMP_LAYER(5)
        H6_LAYER(6, 0, 0, 8)
        H6_A(0, 8, 0, 8, 1, 0, false, 7, next_of(7), 2);
        p2_barrier();
        load_cw();
        H6_B(0, 8, 1, 8, 2, 0, false, 7, next_of(0), 2);
        put_dot(dot, 0);
        p2_barrier();
        H6_DUMP_LAYER(7)
#undef H6_LAYER
        finish(unit, 0);
        prev_unit = unit;
    }
#undef H6_A
#undef H6_B
    if (!first) {       // tail of the last unit: act 7 of set {2, 3} -> dot
        load_cw();
        h6_valu_dot(accB, cw, dot);
        put_dot(dot, 2);
        p2_barrier();
        finish(prev_unit, 2);
    }
}

}  // namespace fneus

extern "C" size_t fneus_h6_blob_bytes(void) { return fneus::kH6Layout.total; }

extern "C" int fneus_h6_pack(const void* blob, void* hblob, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (!blob || !hblob) {
        fneus::set_last_error("fneus_h6_pack: blob and hblob must be given");
        return -2;
    }
    int units = 0;
    for (int l = 0; l < 8; ++l) units += (l == 0 ? 1 : (l == 4 ? 5 : 4)) * fneus::kSdfGeom[l].ntf;
    hipLaunchKernelGGL(fneus::h6_pack_kernel, dim3(units), dim3(64), 0, stream, reinterpret_cast<const unsigned char*>(blob),
                       reinterpret_cast<unsigned char*>(hblob));
    return fneus::launch_status();
}

namespace fneus {
void k1_unit_list(const unsigned char* ray_mask, int m, long all_units, int32_t* work, hipStream_t stream);      // sdf_p2_kernels.hip
}

// ray_mask (may be NULL) [n_pts / m] + work [n_pts / 128 + 1] int32: only the samples of the marked rays are evaluated, the others
// get `fill` (the form of fneus_sdf_fwd_rays: ray mode, m a multiple of 128)
extern "C" int fneus_sdf_fwd_h6(const void* blob, const void* hblob, const float* pts, const float* rays_o, const float* rays_d,
                                const float* t, int m, long n_pts, const unsigned char* ray_mask, float fill, int32_t* work,
                                float* sdf_out, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    if (!blob || !hblob || !sdf_out || (!pts && (!rays_o || !rays_d || !t))) {
        fneus::set_last_error("fneus_sdf_fwd_h6: blob, hblob, sdf_out and either pts or (rays_o, rays_d, t) must be given");
        return -2;
    }
    if (ray_mask && (pts || !work || m <= 0 || m % 128 != 0)) {
        fneus::set_last_error("fneus_sdf_fwd_h6: a ray mask needs the ray form with m = k x 128 samples per ray and the work buffer");
        return -2;
    }
    static bool done = false;
    if (!done) {
        fneus::allow_big_lds(fneus::sdf_fwd_h6_kernel);
        done = true;
    }
    const fneus::PointSrc src{pts, rays_o, rays_d, t, m > 0 ? m : 1};
    const long units = (n_pts + 127) / 128;
    fneus::H6UnitSel sel{nullptr, nullptr, nullptr, 0.0f};
    if (ray_mask) {
        fneus::k1_unit_list(ray_mask, m, units, work, stream);
        sel = fneus::H6UnitSel{work + 1, work, ray_mask, fill};
    }
    hipLaunchKernelGGL(fneus::sdf_fwd_h6_kernel, dim3((unsigned)(units < 256 ? units : 256)), dim3(256),
                       fneus::kH6LdsTotal + 4 * 4 * 32 * 4, stream, reinterpret_cast<const unsigned char*>(blob),
                       reinterpret_cast<const unsigned char*>(hblob), src, n_pts, sdf_out, sel);
    return fneus::launch_status();
}
