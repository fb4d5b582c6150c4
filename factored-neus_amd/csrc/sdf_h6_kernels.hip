// K1 (SDFNetwork.sdf, reference models/fields.py:93-95 via renderer.py:199, 430) with "h6" products (h6_engine.h): ONE fp16 MFMA
// per 16 k for hi.hi and two block-scaled fp6 MFMAs per 64 k for the cross terms -- 1.5 MFMA-times per product instead of the 3 of
// the shipped parity arithmetic.  Round-5 prototype behind FNEUS_K1_H6=1 (fneus/ops.py); the two-pass schedule of
// sdf_p2_kernels.hip on 4 waves x 2 output tiles (a lane's fp6 block is the pair of tiles its wave owns).
//   fneus_h6_pack      : bf3 blob (fwd_hi + fwd_lo = W to 17 bits) -> h6 blob of the forward chain
//   fneus_sdf_fwd_h6   : the kernel
#include <stdlib.h>
#include "h6_engine.h"
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
            const _Float16 wh = (_Float16)w;
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
    const u32x6 q = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(w0, w1, __builtin_bit_cast(float, (e6 - 2u) << 23));
    // Q(Wl) in linear order: a[e] = jj 2 e, b[e] = jj 2 e + 1
    f32x16 ea, eb;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        ea[e] = 2 * e < 16 ? wl0[(2 * e) & 15] : wl1[(2 * e) & 15];
        eb[e] = 2 * e + 1 < 16 ? wl0[(2 * e + 1) & 15] : wl1[(2 * e + 1) & 15];
    }
    const u32x6 ql = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(ea, eb, __builtin_bit_cast(float, (el6 - 2u) << 23));
    unsigned char* rec = hblob + kH6Layout.L[l].rec + (size_t)(b * nt + t) * kH6Rec;
    *reinterpret_cast<p2_u32x4*>(rec + kH6RecWA + lane * 16) = p2_u32x4{q[0], q[1], q[2], q[3]};
    *reinterpret_cast<h6_u32x2*>(rec + kH6RecWB + lane * 8) = h6_u32x2{q[4], q[5]};
    *reinterpret_cast<p2_u32x4*>(rec + kH6RecLA + lane * 16) = p2_u32x4{ql[0], ql[1], ql[2], ql[3]};
    *reinterpret_cast<h6_u32x2*>(rec + kH6RecLB + lane * 8) = h6_u32x2{ql[4], ql[5]};
    *reinterpret_cast<uint32_t*>(rec + kH6RecSc + lane * 4) = (e6 - 2u) | ((el6 - 2u) << 8);
}

// what a pass needs before its first MFMA and the previous pass requests: the bias here, the first block's weights in the weight
// buffer the previous pass's last block did not use (START: which one -- the passes of a unit have 1 1 4 4 4 4 4 4 5 5 4 .. blocks)
struct H6Prime {
    f32x16 bias[2];
};
struct H6Next {
    uint32_t off_hi, off_rec, off_bias;
};

// One pass: MFMAs of layer (KIND, NT_TOTAL) on the sample tiles hbM, hbM + 1 (accM) with, inside their stream, the vector work on
// accV = the previous pass's accumulators of tiles hbV, hbV + 1:
//   ACT 1: softplus -> fp16 hi fragments + fp6 blocks of the next layer's input in LDS (block = this wave's tile pair);
//   ACT 2: softplus -> partial dot product with cw (the sdf row of the linear last layer); 0: none.
// Per block of 64 k: 16 slots hi.hi (4 k-steps x 2 tiles x 2 sample tiles), 8 slots of fp6 cross terms; a slot = one MFMA + its
// share of the vector work + operand requests, fenced by sched_barrier like p2_pass.  Operands are double buffered by block: the
// weights of block b + 1 are requested in the first slots of block b, its activations behind slot 12 (an LDS load must not land
// in registers that a queued MFMA still reads, p2_engine.h: by then the MFMAs that read that buffer have left the pipe).
template <int KIND, int NT_TOTAL, int NKIND, int NNT, int ACT, int START>
FN_DEV void h6_pass(const unsigned char* __restrict__ blob, __amdgpu_buffer_rsrc_t rsrc, uint32_t off_hi, uint32_t off_rec, H6Prime& pr,
                    H6W (&wb)[2], const H6Next& nx, unsigned char* lds, int lane, int t0, f32x16 (&accM)[2][2], int hbM, f32x16 (&accV)[2][2],
                    int hbV, int tnV, const f32x16 (&cw)[2], float (&dot)[2]) {
    constexpr int NB = h6_blocks(KIND);
    constexpr int NSLOT = NB * 24;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        accM[i][0] = pr.bias[i];
        accM[i][1] = pr.bias[i];
    }
    H6B bb[2];
    const unsigned char* setM = lds + hbM * kH6Tile;
    unsigned char* setV = lds + hbV * kH6Tile;
    h6_bload<KIND>(bb[START], setM, 0, lane);
    {   // the registers are free again: next pass's bias
        const f32x16 FN_GLOBAL* __restrict__ p = reinterpret_cast<const f32x16 FN_GLOBAL*>((gblob_t)blob + nx.off_bias);
#pragma unroll
        for (int i = 0; i < 2; ++i) pr.bias[i] = p[(t0 + i) * 2 + (lane >> 5)];
    }
    // ---- the vector work as a list of micro-steps (constant indices only); per sample tile hb:
    //   groups of 8 values: A exp2 / max, B log2, C y (in place);  M block maximum;  S scales;  H fp16 pairs;  L lo parts (in
    //   place);  Q the two conversions;  T stores
    constexpr int OPS_HB = ACT == 1 ? (4 * 24 + 16 + 1 + 16 + 32 + 2 + 9) : (ACT == 2 ? 4 * 24 : 0);
    constexpr int NM = 2 * OPS_HB;
    float ve[8], vm[8], vl[8], bm = 0.0f, sx = 1.0f, sl = 1.0f;
    H6Block ob;
    auto micro = [&](auto J_) {
        constexpr int j = decltype(J_)::value;
        constexpr int hb = j / OPS_HB, k = j % OPS_HB;
        if constexpr (k < 96) {
            constexpr int g = k / 24, ph = (k % 24) / 8, q = k % 8;
            constexpr int v = 8 * g + q, i = v >> 4, e = v & 15;
            if constexpr (ph == 0) {
                const float z = accV[i][hb][e];
                ve[q] = fast_exp2(-fabsf(z) * (kBeta * kLog2e));
                asm volatile("v_max_f32 %0, 0, %2" : "=v"(vm[q]), "+v"(ve[q]) : "v"(z));
            } else if constexpr (ph == 1) {
                vl[q] = fast_log2(1.0f + ve[q]);
                asm volatile("" : "+v"(vl[q]));
            } else {
                float y = fmaf(vl[q], kLn2 / kBeta, vm[q]);
                if constexpr (ACT == 1) {
                    y = i < tnV ? y : 0.0f;              // (layer 3 has 7 tiles: the eighth contributes nothing, also not to the scale)
                    accV[i][hb][e] = y;
                    asm volatile("" : "+v"(accV[i][hb][e]));
                } else {
                    dot[hb] = fmaf(y, cw[i][e], dot[hb]);
                    asm volatile("" : "+v"(dot[hb]));
                }
            }
        } else if constexpr (ACT == 1) {
            constexpr int k2 = k - 96;
            if constexpr (k2 < 16) {                     // M (softplus >= 0: no absolute values)
                if constexpr (k2 == 0) bm = 0.0f;
                bm = fmaxf(bm, fmaxf(accV[0][hb][k2], accV[1][hb][k2]));
                asm volatile("" : "+v"(bm));
            } else if constexpr (k2 == 16) {             // S
                h6_scales(bm, sx, sl, ob.sc);
                asm volatile("" : "+v"(sx), "+v"(sl), "+v"(ob.sc));
            } else if constexpr (k2 < 33) {              // H: values 2 p, 2 p + 1 -> one packed register
                constexpr int p = k2 - 17, v = 2 * p, i = v >> 4, e = v & 15;
                ob.hh[v] = (_Float16)accV[i][hb][e];
                ob.hh[v + 1] = (_Float16)accV[i][hb][e + 1];
            } else if constexpr (k2 < 65) {              // L
                constexpr int v = k2 - 33, i = v >> 4, e = v & 15;
                accV[i][hb][e] -= (float)ob.hh[v];
                asm volatile("" : "+v"(accV[i][hb][e]));
            } else if constexpr (k2 == 65) {
                ob.q = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(ob.hh, sx);
            } else if constexpr (k2 == 66) {
                ob.ql = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(accV[0][hb], accV[1][hb], sl);
            } else {                                     // T: 9 stores
                constexpr int st = k2 - 67;
                unsigned char* tile = setV + hb * kH6Tile;
                const int blk = t0 >> 1;
                if constexpr (st < 4) {
                    f16x8 f;
#pragma unroll
                    for (int jx = 0; jx < 8; ++jx) f[jx] = ob.hh[8 * st + jx];
                    *reinterpret_cast<f16x8*>(tile + kH6Hi + (2 * t0 + st) * kFragBytes + lane * 16) = f;
                } else if constexpr (st == 4) {
                    *reinterpret_cast<p2_u32x4*>(tile + kH6QA + blk * 1024 + lane * 16) = p2_u32x4{ob.q[0], ob.q[1], ob.q[2], ob.q[3]};
                } else if constexpr (st == 5) {
                    *reinterpret_cast<h6_u32x2*>(tile + kH6QB + blk * 512 + lane * 8) = h6_u32x2{ob.q[4], ob.q[5]};
                } else if constexpr (st == 6) {
                    *reinterpret_cast<p2_u32x4*>(tile + kH6LA + blk * 1024 + lane * 16) = p2_u32x4{ob.ql[0], ob.ql[1], ob.ql[2], ob.ql[3]};
                } else if constexpr (st == 7) {
                    *reinterpret_cast<h6_u32x2*>(tile + kH6LB + blk * 512 + lane * 8) = h6_u32x2{ob.ql[4], ob.ql[5]};
                } else {
                    *reinterpret_cast<uint32_t*>(tile + kH6SC + blk * 256 + lane * 4) = ob.sc;
                }
            }
        }
    };
    static_for<0, NB>([&](auto B_) {
        constexpr int b = decltype(B_)::value;
        constexpr int cur = (b + START) & 1, nxt = cur ^ 1;
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 24>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            constexpr int slot = b * 24 + q;
            if constexpr (q < 16) {
                constexpr int s = q >> 2, r = q & 3, i = r >> 1, hb = r & 1;
                if constexpr (h6_ks(KIND, b, s) >= 0) accM[i][hb] = mfma32h(wb[cur].hi[s][i], bb[cur].hi[s][hb], accM[i][hb]);
            } else {
                constexpr int r = (q - 16) & 3, term = (q - 16) >> 2, i = r >> 1, hb = r & 1;
                if constexpr (term == 0) accM[i][hb] = mfma_fp6<0, 1>(h6_op(wb[cur].w6[i]), h6_op(bb[cur].xl6[hb]), accM[i][hb], wb[cur].sc[i], bb[cur].sc[hb]);
                else accM[i][hb] = mfma_fp6<1, 0>(h6_op(wb[cur].l6[i]), h6_op(bb[cur].x6[hb]), accM[i][hb], wb[cur].sc[i], bb[cur].sc[hb]);
            }
            if constexpr (q == 0) {                      // weights of the next block (or of the next pass's first block)
                if constexpr (b + 1 < NB) h6_wload<KIND, NT_TOTAL>(wb[nxt], rsrc, off_hi, off_rec, b + 1, lane, t0);
                else h6_wload<NKIND, NNT>(wb[nxt], rsrc, nx.off_hi, nx.off_rec, 0, lane, t0);
            }
            if constexpr (q == 12 && b + 1 < NB) h6_bload<KIND>(bb[nxt], setM, b + 1, lane);
            if constexpr (NM > 0) {          // micro-steps j with floor(j NSLOT / NM) == slot
                constexpr int j0 = (slot * NM + NSLOT - 1) / NSLOT, j1 = ((slot + 1) * NM + NSLOT - 1) / NSLOT;
                static_for<j0, j1>(micro);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // the operands of this block stay live to its end (no prefetch may be given their registers early)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) asm volatile("" ::"v"(bb[cur].x6[hb]), "v"(bb[cur].xl6[hb]), "v"(bb[cur].hi[3][hb]));
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
__global__ void __launch_bounds__(256, 1) sdf_fwd_h6_kernel(const unsigned char* blob, const unsigned char* hblob, PointSrc src, long N,
                                                            float* __restrict__ sdf_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    constexpr int NW = 4;
    float* red = reinterpret_cast<float*>(lds_ + kH6LdsTotal);               // [4 tiles][NW waves][32 samples]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int t0 = 2 * wave, r = lane & 31, h = lane >> 5;
    constexpr auto& LY = kSdfLayout;
    constexpr auto& HY = kH6Layout;
    const long units = (N + 127) / 128;
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
        h6_store(lds_ + wave * kH6Tile, lane, 16, 3, 4, ob);
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
    const __amdgpu_buffer_rsrc_t rsrc = p2_rsrc(hblob);
    auto next_of = [&](int l) { return H6Next{HY.L[l].hi, HY.L[l].rec, LY.L[l].bias}; };
    H6Prime pr;
    H6W wb[2];
    h6_wload<1, 8>(wb[0], rsrc, HY.L[0].hi, HY.L[0].rec, 0, lane, t0);
    {
        const f32x16 FN_GLOBAL* __restrict__ p = reinterpret_cast<const f32x16 FN_GLOBAL*>((gblob_t)blob + LY.L[0].bias);
#pragma unroll
        for (int i = 0; i < 2; ++i) pr.bias[i] = p[(t0 + i) * 2 + h];
    }
    if ((long)blockIdx.x < units) encode(blockIdx.x);
    p2_barrier();
    bool first = true;
    long prev_unit = 0;
#define H6_PASS(KIND, NT, NKIND, NNT, ACT, START, L_, NX, ACCM, HBM, ACCV, HBV, TNV) \
    h6_pass<KIND, NT, NKIND, NNT, ACT, START>(blob, rsrc, HY.L[L_].hi, HY.L[L_].rec, pr, wb, NX, lds_, lane, t0, ACCM, HBM, ACCV, HBV, TNV, cw, dot)
    for (long unit = blockIdx.x; unit < units; unit += gridDim.x) {
        asm volatile("" : "+s"(blob), "+s"(hblob));
        // ---- layer 0 (one block: the encoding)
        if (!first) load_cw();
        if (first) H6_PASS(1, 8, 1, 8, 0, 0, 0, next_of(0), accA, 0, accB, 2, 2);
        else H6_PASS(1, 8, 1, 8, 2, 0, 0, next_of(0), accA, 0, accB, 2, 2);
        if (!first) put_dot(dot, 2);
        p2_barrier();
        if (!first) finish(prev_unit, 2);
        first = false;
        H6_PASS(1, 8, 0, 8, 1, 1, 0, next_of(1), accB, 2, accA, 0, 2);
        p2_barrier();
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            asm volatile("" : "+s"(blob), "+s"(hblob));
            const int tn3 = 7 - t0 < 2 ? 7 - t0 : 2;                    // layer 3 has 7 tiles: its last wave publishes one fewer
            const int tn_prev = l - 1 == 3 ? tn3 : 2;
            const int tn_this = l == 3 ? tn3 : 2;
            const H6Next same = next_of(l), following = next_of(l == 7 ? 0 : l + 1);
            // pass A: MFMAs of set {0, 1} || activation of layer l-1, set {2, 3}
            if (l == 3) H6_PASS(0, 7, 0, 7, 1, 0, 3, same, accA, 0, accB, 2, tn_prev);
            else if (l == 4) H6_PASS(2, 8, 2, 8, 1, 0, 4, same, accA, 0, accB, 2, tn_prev);
            else H6_PASS(0, 8, 0, 8, 1, 0, l, same, accA, 0, accB, 2, tn_prev);
            p2_barrier();
            // pass B: MFMAs of set {2, 3} || activation of layer l, set {0, 1} (layer 7: -> dot product)
            if (l == 2) H6_PASS(0, 8, 0, 7, 1, 0, 2, following, accB, 2, accA, 0, tn_this);
            else if (l == 3) H6_PASS(0, 7, 2, 8, 1, 0, 3, following, accB, 2, accA, 0, tn_this);
            else if (l == 4) H6_PASS(2, 8, 0, 8, 1, 1, 4, following, accB, 2, accA, 0, tn_this);
            else if (l == 7) {
                load_cw();
                H6_PASS(0, 8, 1, 8, 2, 0, 7, following, accB, 2, accA, 0, tn_this);
            } else H6_PASS(0, 8, 0, 8, 1, 0, l, following, accB, 2, accA, 0, tn_this);
            if (l == 7) put_dot(dot, 0);
            if (l == 5 && unit + gridDim.x < units) encode(unit + gridDim.x);     // the encoding's slots are free behind layer 4
            p2_barrier();
        }
        finish(unit, 0);
        prev_unit = unit;
    }
#undef H6_PASS
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
    int units = 0;
    for (int l = 0; l < 8; ++l) units += fneus::h6_blocks(fneus::h6_kind(l)) * fneus::kSdfGeom[l].ntf;
    hipLaunchKernelGGL(fneus::h6_pack_kernel, dim3(units), dim3(64), 0, stream, reinterpret_cast<const unsigned char*>(blob),
                       reinterpret_cast<unsigned char*>(hblob));
    return fneus::launch_status();
}

extern "C" int fneus_sdf_fwd_h6(const void* blob, const void* hblob, const float* pts, const float* rays_o, const float* rays_d,
                                const float* t, int m, long n_pts, float* sdf_out, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    static bool done = false;
    if (!done) {
        fneus::allow_big_lds(fneus::sdf_fwd_h6_kernel);
        done = true;
    }
    const fneus::PointSrc src{pts, rays_o, rays_d, t, m};
    const long units = (n_pts + 127) / 128;
    hipLaunchKernelGGL(fneus::sdf_fwd_h6_kernel, dim3((unsigned)(units < 256 ? units : 256)), dim3(256),
                       fneus::kH6LdsTotal + 4 * 4 * 32 * 4, stream, reinterpret_cast<const unsigned char*>(blob),
                       reinterpret_cast<const unsigned char*>(hblob), src, n_pts, sdf_out);
    return fneus::launch_status();
}
