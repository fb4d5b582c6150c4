// Common device helpers for the fneus HIP kernels (gfx950 / CDNA4 only).
//
// Data-flow convention of the fused MLP kernels ("transposed, wave-local" formulation):
//   * one wavefront owns 32 ray-samples; sample index = lane & 31, lane half h = lane >> 5;
//   * a dense layer computes Z^T[o][n] = sum_k W[o][k] * H^T[k][n] with v_mfma_f32_32x32x16_bf16:
//       A operand = weight fragment (row = output feature, pre-packed on the device by pack.hip),
//       B operand = activations   (k = input feature in registers, column = sample on the lane),
//       C/D       = 32 output features x 32 samples, fp32;
//   * the C/D layout (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*h) lets the activated accumulator of
//     output tile t feed the next layer's B operand for k-steps 2t and 2t+1 with NO cross-lane movement:
//     k-slot (ks, h, j) holds feature phi(ks,h,j) = 16*ks + 8*(j>>2) + 4*h + (j&3); the weight packer applies
//     the same permutation to the K index of every A fragment.
//   * PREC = 1: plain bf16 operands (fast mode).  PREC = 3: every operand is split x = hi + lo (two bf16,
//     17 significant bits) and a product is hi*hi + hi*lo + lo*hi with fp32 accumulation (parity mode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define FN_DEV __device__ __forceinline__
// Weight blobs are read through explicit GLOBAL-address-space pointers: the kernels launder the blob pointer per tile
// (to stop LICM hoisting the weight stream), which also hides its address space from the compiler, and the resulting
// FLAT loads would tick lgkmcnt as well as vmcnt -- every LDS wait would then drain the weight prefetch.
#define FN_GLOBAL __attribute__((address_space(1)))
typedef const unsigned char FN_GLOBAL* gblob_t;

namespace fneus {

constexpr float kBeta = 100.0f;                 // Softplus(beta=100), reference models/fields.py:72
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// feature index held by k-slot (ks, h, j) of a B fragment / accumulator register mapping
FN_DEV constexpr int phi(int ks, int h, int j) { return 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3); }
// row (within a 32-row tile) held by accumulator register r of lane half h
FN_DEV constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

FN_DEV float bf16_to_f32(__bf16 v) { return (float)v; }

// x = hi + lo split (round-to-nearest-even both times)
FN_DEV void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

template <int PREC>
struct BFrag {
    bf16x8 hi;
    bf16x8 lo;   // unused when PREC == 1
};

FN_DEV f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

FN_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
FN_DEV float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
FN_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// softplus_beta(z) = max(z,0) + log(1 + exp(-beta|z|)) / beta      (== torch softplus incl. its threshold rule to fp32)
FN_DEV float softplus100(float z) {
#ifdef FNEUS_DBG_CHEAP_ACT     // timing experiment only: how much of K1 is activation VALU work
    return fmaxf(z, 0.0f);
#endif
    float e = fast_exp2(-fabsf(z) * (kBeta * kLog2e));
    return fmaxf(z, 0.0f) + fast_log2(1.0f + e) * (kLn2 / kBeta);
}
// h = softplus_beta(z) and s = sigmoid(beta z) = d h / d z from one exponential
FN_DEV void softplus_sig(float z, float& h, float& s) {
    const float e = fast_exp2(-fabsf(z) * (kBeta * kLog2e));
    h = fmaxf(z, 0.0f) + fast_log2(1.0f + e) * (kLn2 / kBeta);
    const float inv = fast_rcp(1.0f + e);
    s = z >= 0.0f ? inv : e * inv;
}
// s = sigmoid(beta z) recovered from h = softplus_beta(z):  s = 1 - exp(-beta h)
FN_DEV float sig_from_softplus(float h) { return 1.0f - fast_exp2(-h * (kBeta * kLog2e)); }

FN_DEV float sigmoidf_acc(float x) { return 1.0f / (1.0f + __expf(-x)); }

// sin and cos of the positional encodings' arguments (|x| <= a few hundred: 2^9 x 1.5 for PE10): three-constant Cody-Waite
// reduction by pi / 2 and the single-precision minimax polynomials on [-pi/4, pi/4] -- max abs error 9.2e-8 over |x| <= 800
// (float32 emulation against fp64, tests/test_host_cpu.py::test_fn_sincos_scheme), ~28 vector instructions.  libm's sincosf is
// 136 (it carries a large-argument path), 18 calls per encoded point: 2450 instructions in the encode phase of every chain kernel,
// run by 4 of a workgroup's 8 waves while the others wait -- 5-7 % of K1 / K2.  (Accurate: the encodings feed the 1e-4 outputs.)
FN_DEV void fn_sincos(float x, float& s, float& c) {
    const float q = __builtin_rintf(x * 0.63661977236758134308f);
    float r = fmaf(q, -1.5703125f, x);
    r = fmaf(q, -4.837512969970703125e-4f, r);
    r = fmaf(q, -7.54978995489188216e-8f, r);
    const float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    ps = fmaf(ps * z, r, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    pc = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    const int n = (int)q;
    const float a = (n & 1) ? pc : ps, b2 = (n & 1) ? ps : pc;          // quadrant: (s, c), (c, -s), (-s, -c), (-c, s)
    s = (n & 2) ? -a : a;
    c = ((n + 1) & 2) ? -b2 : b2;
    // Results anchored call by call.  Without it hipcc SLP-packs the arithmetic of neighbouring calls (v_pk_fma_f32 with scalar
    // register pairs as constants, op_sel_hi 0): the colour network's two-pass kernel then produced encodings that differed from
    // run to run in its bf16 build (tools/experiments/r04/col_repro_dbg.py: every launch, ~1000 of 65 536 samples, only in the units
    // encoded between two passes; the same source with this anchor, or with libm's sincosf, is bit-reproducible).  Cause not pinned
    // down (the packed form reads a 64-bit scalar operand whose upper half is not the constant); the unpacked form costs nothing here.
    asm volatile("" : "+v"(s), "+v"(c));
}

// exchange with the other lane half (lane ^ 32)
FN_DEV float xor32(float v) { return __shfl_xor(v, 32, 64); }

}  // namespace fneus
