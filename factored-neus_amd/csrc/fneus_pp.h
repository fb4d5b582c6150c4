// Fragment planes ("PP planes"): the on-device format of every activation that leaves a chain kernel for the
// weight-gradient GEMM (dw_gemm_pp.hip) or for a later chain kernel.
//
// A chain kernel holds the activations of a 32-sample tile as MFMA B fragments (fneus_common.h): fragment ks = 16
// features x 32 samples = 1 KiB; lane (sample r = lane & 31, half h = lane >> 5) holds the 8 features
//     phi(ks, h, j) = 16 ks + 8 (j >> 2) + 4 h + (j & 3),   j = 0..7            (16 bytes of bf16).
// A plane stores exactly these fragments:
//     plane[tile][F fragments][64 slots][8 bf16],      slot(r, h, ks) = (2 r + h) ^ (8 (ks & 1))
// i.e. one coalesced 16-byte store per lane and fragment, no LDS staging, no transposition in the producer.  F is the
// fragment count of the plane's block (16 for a 256-wide layer, 3 for the 39 -> 48 wide positional encoding).
//
// The GEMM contracts over SAMPLES, so it needs the transposed operand: feature on the lane, 8 consecutive samples in
// the registers.  ds_read_b64_tr_b16 gathers it from the block as it lies in LDS: per 16-lane group, lane 4q+p supplies
// the address of 4 consecutive elements (= the 4 features j & 3 of one (sample, h, j >> 2) unit: 8 bytes) of "row" q
// (= sample), and lane i receives element i of the 4 rows.  With the slot permutation above the 32 lanes of a half
// wave touch 32 distinct 8-byte bank pairs (conflict free); with slot = lane it would be a 4-way conflict.
//
// Invalid samples of a ragged last tile are stored as zeros, so the GEMM needs no row masks.
#pragma once
#include "fneus_common.h"
#include "fneus_layout.h"

namespace fneus {

// sample tiles a plane is allocated for: ceil(N / 32) rounded up to an even count (workgroups of two tiles may store a
// whole padding tile of zeros behind a ragged end)
FN_DEV long pp_tiles(long n) { return 2 * ((n + 63) / 64); }
// (host side: fneus/pp.py alloc_tiles)

// byte offset of this lane's 16 bytes inside fragment ks of a block
FN_DEV unsigned pp_slot_bytes(int lane, int ks) { return (unsigned)(((2 * (lane & 31) + (lane >> 5)) ^ (8 * (ks & 1))) * 16); }

// lane offsets for even / odd fragments, computed once per kernel
struct PPLane {
    unsigned even, odd;
};
FN_DEV PPLane pp_lane(int lane) {
    const unsigned s = (unsigned)(2 * (lane & 31) + (lane >> 5));
    return PPLane{s * 16u, (s ^ 8u) * 16u};
}

FN_DEV void pp_store(unsigned char* __restrict__ block, int ks, const PPLane& pl, bf16x8 v) {
#ifdef FNEUS_DBG_NO_PLANESTORE          // timing experiments only
    if (pl.even != 0xFFFFFFFFu) return;
#endif
    __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(block + (size_t)ks * kFragBytes + ((ks & 1) ? pl.odd : pl.even)));
}
FN_DEV bf16x8 pp_load(const unsigned char* __restrict__ block, int ks, const PPLane& pl) {
    return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(block + (size_t)ks * kFragBytes + ((ks & 1) ? pl.odd : pl.even)));
}

FN_DEV bf16x8 zero_bf16x8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.0f;
    return z;
}

}  // namespace fneus
