// "h6" products (round 5, DESIGN.md section 4.4 plan (ii)): a parity-grade product in 1.5 MFMA-times instead of 3.
//
//   bf3 (shipped):  W x = Wh xh + Wh xl + Wl xh        three v_mfma_f32_32x32x16_bf16 per 16 k (bf16 hi / lo on both sides)
//   h6  (this file): W x = Wh xh                         ONE   v_mfma_f32_32x32x16_f16  per 16 k (fp16 hi: 11 significant bits)
//                        + Q(W) Q(xl) + Q(Wl) Q(x)       TWO   v_mfma_scale_f32_32x32x64_f8f6f4 per 64 k, fp6 e2m3 operands with one
//                                                        power-of-two scale per 32 consecutive k (MX block scaling): 4x the bf16 rate
// The cross terms are 2^-11 of the product and need 4-5 significant bits: exactly what an fp6 mantissa + a shared block
// exponent hold.  tests/checkers/num_schemes.py (fp64 emulation, three networks): sdf <= 2.4e-5, normal <= 6.2e-5 against
// 1.1e-5 / 2.6e-5 of bf3 -- inside north_star's 1e-4 with a 4x / 1.6x margin.  tools/experiments/r05/mx_probe pins the hardware
// semantics this relies on (operand lane map, bit layout, scale bytes, element order of the two conversions).
//
// Operand geometry.  v_mfma_scale_f32_32x32x64_f8f6f4: lane l holds 32 consecutive k of row / column l & 31, k-block l >> 5, element
// jj at bits [6 jj, 6 jj + 6) of six registers; the scale of that lane's block is one E8M0 byte of a register of the SAME lane
// (op_sel picks the byte).  The C/D layout is that of every 32x32 MFMA, so the chain's trick carries over: the activated
// accumulators of output tiles 2w, 2w + 1 held by a lane (2 x 16 values) ARE that lane's 32-element block of k-block
// h = lane >> 5 of the next layer -- element jj = 16 i + e (tile i of the pair, accumulator register e), which is k-step
// 4 b + (jj >> 3), slot jj & 7 of the 16-deep fragments (fneus_common.h phi).  No lane ever needs another lane's values, and the
// block maximum that fixes the scale is a maximum over registers.
//   x6  = Q(x):  v_cvt_scalef32_pk32_fp6_f16 on the 16 registers of packed fp16 hi parts: element order jj (linear);
//   xl6 = Q(xl): v_cvt_scalef32_2xpk16_fp6_f32 on (lo of tile 0, lo of tile 1): element 2 e + i (interleaved) -- the weights
//                of that term are packed in the same order (h6_pack_kernel), so neither side moves a register.
// A wave therefore has to own BOTH tiles of a pair: 4 waves x 2 output tiles, 512 registers, one workgroup per CU.
#pragma once
#include <type_traits>
#include "p2_engine.h"

namespace fneus {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(32))) _Float16 f16x32;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(6))) unsigned int u32x6;
typedef __attribute__((ext_vector_type(2))) unsigned int h6_u32x2;

// ---- weights: the h6 blob of the SDF network's forward chain (layers 0..7; the sdf row of layer 8 and the biases stay in the
// bf3 blob).  Per layer: fp16 hi fragments [ks][t] (1 KiB each, the order of fwd_hi), then one record per (block b, tile t):
constexpr int kH6RecWA = 0;          // Q(W), interleaved order: registers 0..3 [64 lanes][16 B]
constexpr int kH6RecWB = 1024;       //                          registers 4..5 [64][8 B]
constexpr int kH6RecLA = 1536;       // Q(Wl), linear order
constexpr int kH6RecLB = 2560;
constexpr int kH6RecSc = 3072;       // [64] dwords: byte 0 = scale of Q(W), byte 1 = scale of Q(Wl)
constexpr int kH6Rec = 3328;

// k-steps of block b of a layer's INPUT (-1: no such k-step: zeros).  kind 0: 256 inputs; 1: layer 0 (the encoding, 3 k-steps);
// 2: layer 4 ([h_3: 14 k-steps | encoding: 3], fields.py:83-84)
FN_DEV constexpr int h6_blocks(int kind) { return kind == 1 ? 1 : (kind == 2 ? 5 : 4); }
FN_DEV constexpr int h6_ks(int kind, int b, int s) {
    if (kind == 1) return s < 3 ? s : -1;
    if (kind == 2) return b < 3 ? 4 * b + s : (b == 3 ? (s < 2 ? 12 + s : -1) : (s < 3 ? 14 + s : -1));
    return 4 * b + s;
}
FN_DEV constexpr int h6_kind(int l) { return l == 0 ? 1 : (l == 4 ? 2 : 0); }
struct H6LayerOff {
    uint32_t hi, rec;
};
struct H6Layout {
    H6LayerOff L[8];
    uint32_t total;
};
constexpr H6Layout make_h6_layout() {
    H6Layout r{};
    uint32_t off = 0;
    for (int l = 0; l < 8; ++l) {
        r.L[l].hi = off;
        off += kSdfGeom[l].ksf * kSdfGeom[l].ntf * kFragBytes;
        r.L[l].rec = off;
        off += h6_blocks(h6_kind(l)) * kSdfGeom[l].ntf * kH6Rec;
    }
    r.total = off + 4096;           // (slack: the absent tile 7 of layer 3 is read like any other)
    return r;
}
constexpr H6Layout kH6Layout = make_h6_layout();

// ---- activations in LDS, per 32-sample tile
constexpr int kH6Hi = 0;                          // fp16 hi fragments: slots 0..15 of the running layer's input, 16..18 the encoding
constexpr int kH6QA = 19 * kFragBytes;            // x6  registers 0..3: [5 blocks][64][16 B]   (block 4 = the encoding)
constexpr int kH6QB = kH6QA + 5 * 1024;           // x6  registers 4..5: [5][64][8 B]
constexpr int kH6LA = kH6QB + 5 * 512;            // xl6
constexpr int kH6LB = kH6LA + 5 * 1024;
constexpr int kH6SC = kH6LB + 5 * 512;            // [5][64] dwords: byte 0 = scale of x6, byte 1 = scale of xl6
constexpr int kH6Tile = kH6SC + 5 * 256;          // 36 096 B
constexpr int kH6LdsTotal = 4 * kH6Tile;

// LDS slot of k-step s / LDS block of block b of a layer's input
FN_DEV constexpr int h6_slot(int kind, int ks) { return kind == 1 ? 16 + ks : (kind == 2 ? (ks < 14 ? ks : ks + 2) : ks); }
FN_DEV constexpr int h6_lds_block(int kind, int b) { return kind == 1 ? 4 : b; }

FN_DEV f32x16 mfma32h(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// fp6 x fp6 with the scale bytes OPA of sa / OPB of sb (0 or 1)
template <int OPA, int OPB>
FN_DEV f32x16 mfma_fp6(const i32x8& a, const i32x8& b, f32x16 c, int sa, int sb) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, OPA, sa, OPB, sb);
}
typedef __attribute__((ext_vector_type(6))) unsigned int u32x6_fwd;

// the 32 values of a lane's block -> fp16 hi parts, x6, xl6 and the scale dword.  v is overwritten by the lo parts.
// Scales: E = exponent of the block maximum; x6 = Q(x / 2^(E-2)) (block maximum in [4, 8): the top quarter binade saturates at 7.5,
// which costs what the grid spacing there costs anyway); |xl| <= half an ulp of its fp16 hi part <= 2^(E-11): xl6 = Q(xl / 2^(E-14)).
struct H6Block {
    f16x32 hh;
    u32x6 q, ql;
    uint32_t sc;
};
FN_DEV void h6_scales(float m, float& scale_x, float& scale_l, uint32_t& sc) {
    const uint32_t e = __builtin_bit_cast(uint32_t, fmaxf(m, 1.0e-30f)) >> 23;      // (>= 27: both bytes stay positive)
    scale_x = __builtin_bit_cast(float, (e - 2u) << 23);
    scale_l = __builtin_bit_cast(float, (e - 14u) << 23);
    sc = (e - 2u) | ((e - 14u) << 8);
}
FN_DEV void h6_quant(f32x16& v0, f32x16& v1, H6Block& o, bool nonneg) {
    float m = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) m = nonneg ? fmaxf(m, fmaxf(v0[e], v1[e])) : fmaxf(m, fmaxf(fabsf(v0[e]), fabsf(v1[e])));
    float sx, sl;
    h6_scales(m, sx, sl, o.sc);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const _Float16 a = (_Float16)v0[e], b = (_Float16)v1[e];
        o.hh[e] = a;
        o.hh[16 + e] = b;
        v0[e] -= (float)a;
        v1[e] -= (float)b;
    }
    o.q = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(o.hh, sx);
    o.ql = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(v0, v1, sl);
}
// ... to LDS: the block's four hi fragments at slots slot0 .. slot0 + 3 (those with s < ns), the fp6 operands at block blk
FN_DEV void h6_store(unsigned char* tile, int lane, int slot0, int ns, int blk, const H6Block& o) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s < ns) {
            f16x8 f;
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = o.hh[8 * s + j];
            *reinterpret_cast<f16x8*>(tile + kH6Hi + (slot0 + s) * kFragBytes + lane * 16) = f;
        }
    *reinterpret_cast<p2_u32x4*>(tile + kH6QA + blk * 1024 + lane * 16) = p2_u32x4{o.q[0], o.q[1], o.q[2], o.q[3]};
    *reinterpret_cast<h6_u32x2*>(tile + kH6QB + blk * 512 + lane * 8) = h6_u32x2{o.q[4], o.q[5]};
    *reinterpret_cast<p2_u32x4*>(tile + kH6LA + blk * 1024 + lane * 16) = p2_u32x4{o.ql[0], o.ql[1], o.ql[2], o.ql[3]};
    *reinterpret_cast<h6_u32x2*>(tile + kH6LB + blk * 512 + lane * 8) = h6_u32x2{o.ql[4], o.ql[5]};
    *reinterpret_cast<uint32_t*>(tile + kH6SC + blk * 256 + lane * 4) = o.sc;
}

// ---- operand sets of one block: weights of the wave's two tiles (from L2), activations of the two sample tiles of a set (LDS)
struct H6W {
    f16x8 hi[4][2];          // [k-step of the block][tile]
    u32x6 w6[2], l6[2];      // Q(W) (interleaved), Q(Wl) (linear)
    int sc[2];
};
struct H6B {
    f16x8 hi[4][2];          // [k-step][sample tile]
    u32x6 x6[2], xl6[2];
    int sc[2];
};
FN_DEV i32x8 h6_op(const u32x6& v) { return i32x8{(int)v[0], (int)v[1], (int)v[2], (int)v[3], (int)v[4], (int)v[5], 0, 0}; }

template <int KIND, int NT_TOTAL>
FN_DEV void h6_wload(H6W& w, __amdgpu_buffer_rsrc_t rsrc, uint32_t off_hi, uint32_t off_rec, int b, int lane, int t0) {
    const unsigned v16 = (unsigned)(lane + t0 * 64) * 16u;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int ks = h6_ks(KIND, b, s);
        if (ks >= 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                w.hi[s][i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)v16, (int)(off_hi + (uint32_t)((ks * NT_TOTAL + i) * 64) * 16u), 0));
        }
    }
    const unsigned l16 = (unsigned)lane * 16u, l8 = (unsigned)lane * 8u, l4 = (unsigned)lane * 4u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t rec = off_rec + (uint32_t)((b * NT_TOTAL + i) * kH6Rec) + (uint32_t)t0 * kH6Rec;
        const p2_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)l16, (int)(rec + kH6RecWA), 0);
        const h6_u32x2 a2 = __builtin_bit_cast(h6_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)l8, (int)(rec + kH6RecWB), 0));
        const p2_u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)l16, (int)(rec + kH6RecLA), 0);
        const h6_u32x2 c2 = __builtin_bit_cast(h6_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)l8, (int)(rec + kH6RecLB), 0));
        w.w6[i] = u32x6{a[0], a[1], a[2], a[3], a2[0], a2[1]};
        w.l6[i] = u32x6{c[0], c[1], c[2], c[3], c2[0], c2[1]};
        w.sc[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)l4, (int)(rec + kH6RecSc), 0);
    }
}

template <int KIND>
FN_DEV void h6_bload(H6B& o, const unsigned char* set /* first tile of the set */, int b, int lane) {
    constexpr int dummy = 0;
    (void)dummy;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const unsigned char* tile = set + hb * kH6Tile;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ks = h6_ks(KIND, b, s);
            if (ks >= 0) o.hi[s][hb] = *reinterpret_cast<const f16x8*>(tile + kH6Hi + h6_slot(KIND, ks) * kFragBytes + lane * 16);
        }
        const int blk = h6_lds_block(KIND, b);
        const p2_u32x4 a = *reinterpret_cast<const p2_u32x4*>(tile + kH6QA + blk * 1024 + lane * 16);
        const h6_u32x2 a2 = *reinterpret_cast<const h6_u32x2*>(tile + kH6QB + blk * 512 + lane * 8);
        const p2_u32x4 c = *reinterpret_cast<const p2_u32x4*>(tile + kH6LA + blk * 1024 + lane * 16);
        const h6_u32x2 c2 = *reinterpret_cast<const h6_u32x2*>(tile + kH6LB + blk * 512 + lane * 8);
        o.x6[hb] = u32x6{a[0], a[1], a[2], a[3], a2[0], a2[1]};
        o.xl6[hb] = u32x6{c[0], c[1], c[2], c[3], c2[0], c2[1]};
        o.sc[hb] = *reinterpret_cast<const int*>(tile + kH6SC + blk * 256 + lane * 4);
    }
}

}  // namespace fneus
