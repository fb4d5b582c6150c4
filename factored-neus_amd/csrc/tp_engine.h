// Tensor-parallel workgroup helpers shared by the SDF and colour kernels: the 4 wavefronts of a workgroup share one
// 32-sample tile, wave w owns output tiles 2w, 2w+1 of every layer; activated tiles are exchanged through LDS as ready-made
// B fragments (k-step 2t+s of the next layer = half s of tile t); where a stash plane is due the same fragments are stored
// to global memory as they are (pp_engine.h).  See sdf_kernels.hip (K1 / K2) for the design notes.
#pragma once
#include "mlp_engine.h"

namespace fneus {

template <int PREC, int TN>
FN_DEV void tp_publish(unsigned char* frag, int lane, int t0, const f32x16 (&acc)[TN]) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // everyone has fetched the previous layer's fragments
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
            bf16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (PREC == 3) {
                    __bf16 a, b2;
                    split_bf16(acc[i][8 * sh + j], a, b2);
                    hi[j] = a;
                    lo[j] = b2;
                } else {
                    hi[j] = (__bf16)acc[i][8 * sh + j];
                }
            }
            const int ks = 2 * (t0 + i) + sh;
            *reinterpret_cast<bf16x8*>(frag + (ks * NPL) * kFragBytes + lane * 16) = hi;
            if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(frag + (ks * NPL + 1) * kFragBytes + lane * 16) = lo;
        }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // all fragments of the layer are in LDS
}

template <int PREC, int KS>
FN_DEV void tp_gather(const unsigned char* frag, int lane, BFrag<PREC> (&bf)[kMaxKS]) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        bf[ks].hi = *reinterpret_cast<const bf16x8*>(frag + (ks * NPL) * kFragBytes + lane * 16);
        if constexpr (PREC == 3) bf[ks].lo = *reinterpret_cast<const bf16x8*>(frag + (ks * NPL + 1) * kFragBytes + lane * 16);
    }
}

#ifndef FNEUS_TP_DEPTH
#define FNEUS_TP_DEPTH 4        // weight-prefetch depth of the tensor-parallel K2 (stages of 2 tiles); measured 2 / 3 / 4
#endif
#ifndef FNEUS_TP_LDSB_BF16
#define FNEUS_TP_LDSB_BF16 0
#endif
template <int PREC> constexpr bool kTpLdsB = PREC == 3 || FNEUS_TP_LDSB_BF16;

// B operand of a tensor-parallel layer: parity mode reads the fragments from LDS k-step by k-step (dense_ldsb), bf16
// mode has the registers to hold them all (tp_gather + dense)
template <int PREC, int KS, int NT_TOTAL, int T0, int TN, bool WLO = true>
FN_DEV void tp_dense(const unsigned char* __restrict__ blob, uint32_t off_hi, uint32_t off_lo,
                     const unsigned char* frag, const BFrag<PREC> (&bf)[kMaxKS], f32x16 (&acc)[TN], int lane,
                     int t0_rt = 0) {
    if constexpr (kTpLdsB<PREC>)
        dense_ldsb<PREC, KS, NT_TOTAL, T0, TN, FNEUS_TP_DEPTH, WLO>(blob, off_hi, off_lo, frag, acc, lane, t0_rt);
    else
        dense<PREC, KS, NT_TOTAL, T0, TN, 0, FNEUS_TP_DEPTH, WLO>(blob, off_hi, off_lo, bf, acc, lane, t0_rt);
}
template <int PREC, int KS>
FN_DEV void tp_operands(const unsigned char* frag, int lane, BFrag<PREC> (&bf)[kMaxKS]) {
    if constexpr (!kTpLdsB<PREC>) tp_gather<PREC, KS>(frag, lane, bf);
}

}  // namespace fneus
