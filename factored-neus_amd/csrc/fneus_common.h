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

// PREC 2 (round 5, the stage-3 visibility network only): ONE product per multiplication like PREC 1, on fp16 operands -- 11
// significant bits instead of 8.  The fragments travel in the same 8 x 16-bit containers (bf16x8): only the conversion into them
// (to16) and the matrix instruction (mfma32p) know the format.
typedef _Float16 fn_f16x8 __attribute__((ext_vector_type(8)));
template <int PREC>
FN_DEV __bf16 to16(float v) {
    if constexpr (PREC == 2) return __builtin_bit_cast(__bf16, (_Float16)v);
    else return (__bf16)v;
}
template <int PREC>
FN_DEV f32x16 mfma32p(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (PREC == 2)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fn_f16x8, a), __builtin_bit_cast(fn_f16x8, b), c, 0, 0, 0);
    else
        return mfma32(a, b, c);
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

// sin and cos of the positional encodings' arguments (|x| <= a few hundred: 2^9 x 1.5 for PE10), correctly rounded in > 98 % of
// the cases and never more than 0.52 ulp off (tests/test_host_cpu.py::test_fn_sincos_scheme restates it in numpy): the argument
// is reduced by pi / 2 (two constants) in double-precision fmas and the
// polynomials on [-pi/4, pi/4] (least squares on Chebyshev nodes, 1.4e-11 and 7.6e-10 from sin and cos) run in double --
// v_fma_f64 issues at the rate of v_fma_f32 on gfx950 -- so the only rounding of note is the final one to float.
//   libm's sincosf is 136 instructions (it carries a large-argument path) and 1-2 ulp; this is ~26 and lands on the value the
//   reference's host libm returns, which an fp32 Cody-Waite + cephes version (28 instructions, 1.5 ulp) did not: its encodings
//   moved hit points by ~1e-6, enough to put a leaky-ReLU input of stage 3's net_cs on the other side of its kink in the 24-ray
//   fixture (tools/experiments/r04/stage3_grad_dbg.py; 7 % of that layer's weight gradient).
// 18 calls per encoded point: ~2450 libm instructions in the encode phase of every chain kernel, run by 4 of a workgroup's 8 waves
// while the others wait.
FN_DEV void fn_sincos(float x, float& s, float& c) {
#ifdef FNEUS_LIBM_SINCOS                // A/B builds (tools/experiments/build_variant.sh)
    sincosf(x, &s, &c);
    return;
#endif
    const double xd = (double)x;
    const double q = __builtin_rint(xd * 0.63661977236758134308);
    double r = __builtin_fma(q, -1.57079632679489661923, xd);
    r = __builtin_fma(q, -6.123233995736766e-17, r);
    const double z = r * r;
    double ps = __builtin_fma(2.724990252733835e-06, z, -1.984008661428886e-04);
    ps = __builtin_fma(ps, z, 8.333331874648266e-03);
    ps = __builtin_fma(ps, z, -1.666666666385583e-01);
    ps = __builtin_fma(ps * z, r, r);
    double pc = __builtin_fma(2.4547940868609518e-05, z, -1.3888303106225684e-03);
    pc = __builtin_fma(pc, z, 4.166666466064577e-02);
    pc = __builtin_fma(pc * z, z, __builtin_fma(-0.5, z, 1.0));
    const int n = (int)q;
    const float fs = (float)ps, fc = (float)pc;
    const float a = (n & 1) ? fc : fs, b2 = (n & 1) ? fs : fc;          // quadrant: (s, c), (c, -s), (-s, -c), (-c, s)
    s = (n & 2) ? -a : a;
    c = ((n + 1) & 2) ? -b2 : b2;
    // Results anchored call by call: hipcc SLP-packed the fp32 arithmetic of neighbouring calls of the fp32 version (v_pk_fma_f32
    // with scalar register pairs as constants, op_sel_hi 0), and the colour network's two-pass kernel then produced encodings
    // that differed from run to run in its bf16 build (tools/experiments/r04/col_repro_dbg.py: every launch, ~1000 of 65 536
    // samples, only in the units encoded between two passes; with the anchor, or with libm's sincosf, bit-reproducible).  Cause
    // not pinned down; kept although there is no packed fp64 arithmetic to form.
#ifndef FNEUS_SINCOS_NO_ANCHOR           // A/B builds: is the anchor still needed with the double-precision version?
    asm volatile("" : "+v"(s), "+v"(c));
#endif
}

// exchange with the other lane half (lane ^ 32)
FN_DEV float xor32(float v) { return __shfl_xor(v, 32, 64); }

}  // namespace fneus
