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
//   x6 = Q(x), xl6 = Q(xl): v_cvt_scalef32_2xpk16_fp6_f32 on (values of tile 0, values of tile 1): it emits element 2 e + i
//   (interleaved); the fp6 weights are packed in the same order (h6_pack_kernel), so neither side moves a register.
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
// An fp6 operand is stored as EIGHT dwords per lane, two planes [64 lanes][16 B] (ONE per-lane offset, lane * 16, addresses
// everything): dwords 0..5 the 32 codes, dword 6 the scales of the lane's blocks (byte 0: of Q(W) / x6, byte 1: of Q(Wl) / xl6),
// dword 7 unused.  Two 16-byte loads then land in one 8-register tuple whose first six registers ARE the MFMA operand -- no
// register is moved (assembling six registers from a 16-byte and an 8-byte piece cost a `s_waitcnt vmcnt(0)` + moves per load).
constexpr int kH6RecW = 0;           // Q(W)  dwords 0..3 | 1024: dwords 4..7
constexpr int kH6RecL = 2048;        // Q(Wl) dwords 0..3 | 3072: dwords 4..7
constexpr int kH6Rec = 4096;
// (both fp6 operands in the element order of v_cvt_scalef32_2xpk16_fp6_f32: element 2 e + i = register e of tile i of the pair)

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

// ---- activations in LDS, per 32-sample tile: every access is lane * 16 + a constant
constexpr int kH6Hi = 0;                          // fp16 hi fragments: slots 0..15 of the running layer's input, 16..18 the encoding
constexpr int kH6Q = 19 * kFragBytes;             // per block (5: block 4 = the encoding) four planes [64][16 B]:
constexpr int kH6QX = 0;                          //   x6  dwords 0..3 | 1024: {x6 4, x6 5, scales, -}
constexpr int kH6QL = 2048;                       //   xl6 dwords 0..3 | 3072: {xl6 4, xl6 5, scales, -}
constexpr int kH6Tile = kH6Q + 5 * 4096;          // 39 936 B
constexpr int kH6LdsTotal = 4 * kH6Tile;

// LDS slot of k-step s / LDS block of block b of a layer's input
FN_DEV constexpr int h6_slot(int kind, int ks) { return kind == 1 ? 16 + ks : (kind == 2 ? (ks < 14 ? ks : ks + 2) : ks); }
FN_DEV constexpr int h6_lds_block(int kind, int b) { return kind == 1 ? 4 : b; }

FN_DEV f32x16 mfma32h(f16x8 a, f16x8 b, f32x16 c) {
#ifdef FNEUS_H6_NO_MFMA                  // timing experiments only
    asm volatile("" : "+v"(c) : "v"(a), "v"(b));
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}
// fp6 x fp6 with the scale bytes OPA of sa / OPB of sb (0 or 1)
template <int OPA, int OPB>
FN_DEV f32x16 mfma_fp6(const i32x8& a, const i32x8& b, f32x16 c, int sa, int sb) {
#ifdef FNEUS_H6_NO_MFMA
    asm volatile("" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    return c;
#else
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, OPA, sa, OPB, sb);
#endif
}

// v_cvt_scalef32_2xpk16_fp6_f32 and its operands.  The instruction converts one element pair per pass and reads its 32 source
// registers AND the scale pass by pass; neither the hardware nor hipcc (ROCm 7.2) keeps a later instruction from writing a
// register it has not read yet.  Seen, each one on the GPU (tools/experiments/r05: h6_pack_check.py, h6_dbg.sh, the ISA):
//   * results in the register of the scale (`... v[66:71], v[2:17], v[18:33], v66`): pair 0 right, the rest divided by garbage;
//   * results inside a source tuple (`... v[70:75], v[70:85], ...`): a few elements wrong (sdf 1.4e-4 instead of 5e-5);
//   * `v_accvgpr_read_b32 v38, a178` right behind `... v[188:193], v[38:53], v[54:69], v196`: NaN.
// So: h6_cvt keeps scale and sources out of the result registers (an empty asm reads them behind the conversion), and whoever
// calls it keeps them alive until h6_cvt_release -- placed a few dozen issue cycles later -- or uses h6_cvt_slow.
FN_DEV u32x6 h6_cvt(const f32x16& a, const f32x16& b, float scale) {
    u32x6 q = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    asm volatile("" : "+v"(q) : "v"(scale), "v"(a), "v"(b));
    return q;
}
FN_DEV void h6_cvt_release(const f32x16& a, const f32x16& b, float scale) { asm volatile("" ::"v"(scale), "v"(a), "v"(b)); }
FN_DEV u32x6 h6_cvt_slow(const f32x16& a, const f32x16& b, float scale) {
    u32x6 q = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(q) : "v"(scale), "v"(a), "v"(b));
    return q;
}

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
    o.q = h6_cvt_slow(v0, v1, sx);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const _Float16 a = (_Float16)v0[e], b = (_Float16)v1[e];
        o.hh[e] = a;
        o.hh[16 + e] = b;
        v0[e] -= (float)a;
        v1[e] -= (float)b;
    }
    o.ql = h6_cvt_slow(v0, v1, sl);
}
// ... to LDS (tile16 = tile + lane * 16): the block's hi fragments at slots slot0 .. (those with s < ns), the fp6 operands at blk
FN_DEV void h6_store(unsigned char* tile16, int slot0, int ns, int blk, const H6Block& o) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s < ns) {
            f16x8 f;
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = o.hh[8 * s + j];
            *reinterpret_cast<f16x8*>(tile16 + kH6Hi + (slot0 + s) * kFragBytes) = f;
        }
    unsigned char* q = tile16 + kH6Q + blk * 4096;
    *reinterpret_cast<p2_u32x4*>(q + kH6QX) = p2_u32x4{o.q[0], o.q[1], o.q[2], o.q[3]};
    *reinterpret_cast<p2_u32x4*>(q + kH6QX + 1024) = p2_u32x4{o.q[4], o.q[5], o.sc, 0u};
    *reinterpret_cast<p2_u32x4*>(q + kH6QL) = p2_u32x4{o.ql[0], o.ql[1], o.ql[2], o.ql[3]};
    *reinterpret_cast<p2_u32x4*>(q + kH6QL + 1024) = p2_u32x4{o.ql[4], o.ql[5], o.sc, 0u};
}

// an operand from its two 16-byte pieces
FN_DEV i32x8 h6_op8(const p2_u32x4& a, const p2_u32x4& b) {
    return i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
}

}  // namespace fneus
