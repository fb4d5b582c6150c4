// Per-lobe light visibility of stage 3 (lvis_kernels.hip; reference models/inverRender.py:128-192 on the Lvis network of
// models/fields.py:338-369) in the two-pass pipelined form of p2_engine.h: the same tiles ((point, lobe) pair = 32 directions),
// operands and per-accumulator summation order as lvis_visibility_tph_kernel.
//
// A work item is a (point, chunk of 64 lobes) pair.  Round 4: its (lobe, direction) pairs that face the point are LISTED first
// (lobe-major; a direction on the far side of the normal is multiplied by zero, inverRender.py:183 -- about half of the 32
// directions of a lobe near the horizon, all of them beyond it) and the list is taken 32 entries = one tile at a time, so a tile
// holds directions of several lobes; the weighted sums per lobe are segmented sums over the list.  Before, every lobe with at
// least one facing direction cost a whole tile: 57 k tiles per step where the facing directions fill 33 k.
// A unit = 4 tiles in two sets A = {0, 1}, B = {2, 3}, 8 waves, wave w owns output tile w of every layer.
//   L0.A || tail of the previous unit (ReLU of layer 3, set B -> dot)      L0.B || ReLU 0 A
//   Ll.A || ReLU l-1 B                                                      Ll.B || ReLU l A       (l = 1..3; ReLU 3 -> dot)
// The last layer (256 -> 1) is a vector dot product on the accumulators + a fixed-order sum over the waves through LDS (as the
// sdf row of K1), followed by the sigmoid and the weighted average over the tile's 32 directions.
// LDS per tile: slots 0..15 the running layer's input (layer 0, round 5: slots 0, 1 the 2 k-steps PE4(direction); the 4 k-steps
// PE10(point) sit in slots 0..3 of tiles 0, 1 for the one pass per item that multiplies them).
#include <stdlib.h>
#ifndef FNEUS_P2_DEPTH
#define FNEUS_P2_DEPTH 2            // weight-prefetch distance of this file's passes (p2_engine.h: 3): layer 0's direction part is a pass of 2 k-steps
#endif
#include "p2_engine.h"
#include "fneus_kernels.h"
#include "lvis_p2.h"

namespace fneus {

#ifndef FNEUS_LVIS_P2_CHUNK
#define FNEUS_LVIS_P2_CHUNK 64
#endif
constexpr int kLvisP2Chunk = FNEUS_LVIS_P2_CHUNK;            // lobes per work item (<= 64: one ballot)
constexpr int kLvisP2Red = kP2LdsTotal;                      // float [4 tiles][8 waves][32]
constexpr int kLvisP2List = kP2LdsTotal + 4 * 8 * 32 * 4;    // float den[64], float num[64]: the chunk's lobes
constexpr int kLvisP2LdsTotal = kLvisP2List + 128 * 4;
// slots 17, 18 of a tile's region are not used by this network (16 input k-steps + the parked encoding in slot 16): 4 KiB each
constexpr int kLvisP2Pairs = 17 * 2 * kFragBytes;            // tile 0: uint16 [64 x 32] the listed pairs (lobe in chunk << 5 | direction)
constexpr int kLvisP2Counts = kP2Half + 17 * 2 * kFragBytes; // tile 1: int [8] listed pairs per wave of a listing round
static_assert(kLvisP2Chunk % 16 == 0 && kLvisP2Chunk <= 64, "512 threads list 16 lobes x 32 directions per round");

// k-step ks of PE10(point) as one B fragment (lvis_kernels.hip posenc3_frag)
template <int PREC>
FN_DEV void lvis_p2_posenc3(const float (&x)[3], int ks, int h, BFrag<PREC>& out) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int f = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
        float val = 0.0f;
        if (f < 63) {
            const int g = f - 3;
            const int c = f < 3 ? f : (g % 3);
            const float xc = c == 0 ? x[0] : (c == 1 ? x[1] : x[2]);
            if (f < 3) {
                val = xc;
            } else {
                float sn, cs;
                fn_sincos(xc * (float)(1 << (g / 6)), sn, cs);
                val = ((g % 6) >= 3) ? cs : sn;
            }
        }
        if constexpr (PREC == 3) {
            __bf16 a, b;
            split_bf16(val, a, b);
            out.hi[j] = a;
            out.lo[j] = b;
        } else {
            out.hi[j] = to16<PREC>(val);
        }
    }
}

template <int PREC>
__global__ void __launch_bounds__(512, 1) lvis_visibility_p2_kernel(const unsigned char* blob, const float* __restrict__ points,
                                                                    const float* __restrict__ normals,
                                                                    const float* __restrict__ dirs /*[M][32][3]*/,
                                                                    const float* __restrict__ weights /*[M][32]*/,
                                                                    const unsigned char* __restrict__ point_mask, int n_pts,
                                                                    int n_lobes, float* __restrict__ vis /*[M][n_pts]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    constexpr int TN = 1, NW = 8;
    constexpr int NPL = PREC == 3 ? 2 : 1;
    float* red = reinterpret_cast<float*>(lds_ + kLvisP2Red);
    float* den = reinterpret_cast<float*>(lds_ + kLvisP2List);
    float* num = den + 64;
    unsigned short* pairs = reinterpret_cast<unsigned short*>(lds_ + kLvisP2Pairs);
    int* counts = reinterpret_cast<int*>(lds_ + kLvisP2Counts);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int t0 = wave, r = lane & 31, h = lane >> 5;
    constexpr auto& LY = kLvisLayout;
    const int n_chunks = (n_lobes + kLvisP2Chunk - 1) / kLvisP2Chunk;
    const __amdgpu_buffer_rsrc_t rsrc = p2_rsrc(blob);
    // Layer 0 in two parts (round 5): its first four k-steps multiply the POINT's encoding -- the same for every direction of an item --
    // and run once per item (pass P below); a unit's layer-0 passes take the two k-steps of the direction's encoding from there.  Same
    // operands in the same order per accumulator as the one six-k-step pass: bit-identical, 8 of a unit's 108 k-steps fewer.
    constexpr uint32_t kDirPart = 4u * 8u * (uint32_t)kFragBytes;            // k-steps 4, 5 of layer 0's fragments ([ks][tile])
    auto next_of = [&](int l) {
        return l == 0 ? P2Next{LY.L[0].fwd_hi + kDirPart, LY.L[0].fwd_lo + kDirPart, LY.L[0].bias, 8}
                      : P2Next{LY.L[l].fwd_hi, LY.L[l].fwd_lo, LY.L[l].bias, 8};
    };
    const P2Next point_part{LY.L[0].fwd_hi, LY.L[0].fwd_lo, LY.L[0].bias, 8};
    P2Prime<FNEUS_P2_DEPTH, TN> pr;
    f32x16 accA[TN][2], accB[TN][2], cw[TN];
    float dot[2] = {0.0f, 0.0f};
    auto load_cw = [&]() { load_accvec<8, 0, TN>(blob, LY.extra, cw, lane, t0); };   // the row of the last layer in accumulator layout
    for (long item = blockIdx.x; item < (long)n_pts * n_chunks; item += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const int pt = (int)(item / n_chunks), lobe0 = (int)(item - (long)pt * n_chunks) * kLvisP2Chunk;
        const int lobe1 = lobe0 + kLvisP2Chunk < n_lobes ? lobe0 + kLvisP2Chunk : n_lobes;
        if (point_mask && point_mask[pt] == 0) {          // a ray without a surface hit (fixed-shape step): nothing to evaluate
            for (int lobe = lobe0 + (int)threadIdx.x; lobe < lobe1; lobe += blockDim.x) vis[(size_t)lobe * n_pts + pt] = 0.0f;
            continue;
        }
        float x[3], nrm[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[c] = points[pt * 3 + c];
            nrm[c] = normals[pt * 3 + c];
        }
        // ---- the (lobe, direction) pairs of the chunk on the normal's side (inverRender.py:169), lobe-major: 16 lobes per round
        int n_pairs = 0;
#pragma unroll 1
        for (int round = 0; round < kLvisP2Chunk / 16; ++round) {
            const int lj = 16 * round + (int)(threadIdx.x >> 5), lobe = lobe0 + lj;
            bool f = false;
            if (lobe < lobe1) {
                const float* d = dirs + ((size_t)lobe * 32 + r) * 3;
                f = (nrm[0] * d[0] + nrm[1] * d[1] + nrm[2] * d[2]) > 1e-6f;
            }
            const unsigned long long m = __ballot(f);
            if (lane == 0) counts[wave] = __builtin_popcountll(m);
            p2_barrier();
            int before = n_pairs, total = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const int c = counts[k];
                before += k < wave ? c : 0;
                total += c;
            }
            if (f) pairs[before + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (unsigned short)((lj << 5) | r);
            n_pairs += total;
            p2_barrier();                                 // counts are rewritten by the next round
        }
        n_pairs = __builtin_amdgcn_readfirstlane(n_pairs);
        // the lobes' weight sums over ALL 32 directions (the denominator of inverRender.py:188) and their running numerators
        for (int lj = wave; lj < kLvisP2Chunk; lj += NW) {
            float dsum = (lobe0 + lj < lobe1 && h == 0) ? weights[(size_t)(lobe0 + lj) * 32 + r] : 0.0f;
#pragma unroll
            for (int sft = 16; sft >= 1; sft >>= 1) dsum += __shfl_xor(dsum, sft, 64);
            if (lane == 0) {
                den[lj] = dsum;
                num[lj] = 0.0f;
            }
        }
        if (wave < 4) {                                   // k-step `wave` of the point's encoding -> slot `wave` of tiles 0 and 1 (set A)
            BFrag<PREC> one;
            lvis_p2_posenc3<PREC>(x, wave, h, one);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                unsigned char* dst = lds_ + t * kP2Half + (wave * NPL) * kFragBytes + lane * 16;
                *reinterpret_cast<bf16x8*>(dst) = one.hi;
                if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(dst + kFragBytes) = one.lo;
            }
        }
        p2_barrier();
        const int n_tiles = (n_pairs + 31) >> 5;
        const int n_units = (n_tiles + 3) >> 2;
        // slot r of tile t holds pair 32 t + r of the list (the last pair again beyond its end: computed, not counted)
        auto pair_of = [&](int t, bool& valid) {
            const int e = 32 * t + r;
            valid = e < n_pairs;
            return (int)pairs[valid ? e : (n_pairs > 0 ? n_pairs - 1 : 0)];
        };
        auto encode = [&](int unit, int ta, int tb) {     // waves ta .. tb: the 2 direction k-steps of tile `wave` of the unit (slots 0, 1)
            if (wave < ta || wave > tb) return;
            unsigned char* tile = lds_ + wave * kP2Half + lane * 16;
            bool valid;
            const int pr_ = pair_of(4 * unit + wave, valid);
            const int lobe = lobe0 + (pr_ >> 5), dir = pr_ & 31;
            float d[3], pe[27], jc[27];
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = dirs[((size_t)lobe * 32 + dir) * 3 + c];
            posenc<4, false>(d, pe, jc);
            BFrag<PREC> tmp[kMaxKS];
            vec_to_bfrag<PREC, 27, 2, 0>(pe, tmp, h);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                *reinterpret_cast<bf16x8*>(tile + (j * NPL) * kFragBytes) = tmp[j].hi;
                if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(tile + (j * NPL + 1) * kFragBytes) = tmp[j].lo;
            }
        };
        auto put_dot = [&](int hb0) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float p = dot[k] + xor32(dot[k]);
                if (lane < 32) red[((hb0 + k) * NW + wave) * 32 + lane] = p;
                dot[k] = 0.0f;
            }
        };
        auto finish = [&](int unit, int hb0) {            // waves hb0, hb0 + 1: output of tile `wave` -> sigmoid -> the lobes' numerators
            if (wave < hb0 || wave > hb0 + 1) return;
            const int t = 4 * unit + wave;
            if (t >= n_tiles) return;
            bool valid;
            const int pr_ = pair_of(t, valid);
            const int lj = pr_ >> 5, dir = pr_ & 31;
            f32x16 b4[1];
            load_accvec<1, 0, 1>(blob, LY.L[4].bias, b4, lane);
            float s = b4[0][0];
#pragma unroll
            for (int k = 0; k < NW; ++k) s += red[(wave * NW + k) * 32 + r];
            const float w = weights[(size_t)(lobe0 + lj) * 32 + dir];
            float v = (h == 0 && valid) ? w / (1.0f + expf(-s)) : 0.0f;             // fields.py:358 sigmoid (every listed pair faces)
            // segmented inclusive scan over lanes 0..31 (the list is lobe-major: a lobe's pairs are neighbours); the last lane of a
            // segment adds its sum to the lobe's numerator.  A lobe has at most 32 pairs, so it is split over at most two tiles:
            // 0 + a + b in either order is the same number, whichever tile comes first.
            const int key = (h == 0 && valid) ? lj : -1 - lane;
#pragma unroll
            for (int sft = 1; sft < 32; sft <<= 1) {
                const float ov = __shfl_up(v, sft, 64);
                const int ok = __shfl_up(key, sft, 64);
                if ((lane & 31) >= sft && ok == key) v += ov;
            }
            const int nk = __shfl_down(key, 1, 64);
            if (h == 0 && valid && (r == 31 || nk != key)) atomicAdd(&num[lj], v);
        };
        auto write_vis = [&]() {                          // inverRender.py:188: weighted mean over the lobe's 32 directions
            for (int lj = (int)threadIdx.x; lj < lobe1 - lobe0; lj += (int)blockDim.x)
                vis[(size_t)(lobe0 + lj) * n_pts + pt] = num[lj] / (den[lj] + 1e-6f);
        };
        if (n_pairs == 0) {                               // the whole chunk faces away
            write_vis();
            p2_barrier();
            continue;
        }
#define LVIS_PASS_AT(KS, ACT, OFF_HI, OFF_LO, NX, ACCM, HBM, ACCV, HBV)                                                       \
    p2_pass<PREC, KS, 8, 0, ACT, TN>(blob, rsrc, OFF_HI, OFF_LO, pr, NX, lds_, lane, t0, ACCM, HBM, ACCV, HBV, TN, cw, dot)
#define LVIS_PASS(KS, ACT, L_, NX, ACCM, HBM, ACCV, HBV) LVIS_PASS_AT(KS, ACT, LY.L[L_].fwd_hi, LY.L[L_].fwd_lo, NX, ACCM, HBM, ACCV, HBV)
// layer 0's direction part: 2 k-steps on top of the point part (the primed bias registers are the pass's initial accumulators)
#define LVIS_L0(ACT, NX, ACCM, HBM, ACCV, HBV)                                                                                \
    do {                                                                                                                      \
        pr.bias[0] = acc0;                                                                                                    \
        LVIS_PASS_AT(2, ACT, LY.L[0].fwd_hi + kDirPart, LY.L[0].fwd_lo + kDirPart, NX, ACCM, HBM, ACCV, HBV);                  \
    } while (0)
        // pass P: bias + the point part of layer 0, once per item (every column of the tile holds the same point: one accumulator
        // vector serves all directions)
        p2_prime_all<PREC, FNEUS_P2_DEPTH, TN>(pr, blob, rsrc, lane, t0, point_part);
        LVIS_PASS_AT(4, 0, point_part.off_hi, point_part.off_lo, next_of(0), accA, 0, accB, 2);
        const f32x16 acc0 = accA[0][0];
        p2_barrier();                                     // set A's slots 0..3 have been read: the units' direction k-steps go there
        encode(0, 0, 3);
        p2_barrier();
#pragma unroll 1
        for (int unit = 0; unit < n_units; ++unit) {
            asm volatile("" : "+s"(blob));
            if (unit > 0) {
                load_cw();
                LVIS_L0(4, next_of(0), accA, 0, accB, 2);
                put_dot(2);
            } else {
                LVIS_L0(0, next_of(0), accA, 0, accB, 2);
            }
            p2_barrier();
            if (unit > 0) finish(unit - 1, 2);
            LVIS_L0(3, next_of(1), accB, 2, accA, 0);
            p2_barrier();
#pragma unroll 1
            for (int l = 1; l <= 3; ++l) {
                asm volatile("" : "+s"(blob));
                const P2Next same = next_of(l), following = next_of(l == 3 ? 0 : l + 1);
                LVIS_PASS(16, 3, l, same, accA, 0, accB, 2);
                p2_barrier();
                if (l == 3) {
                    load_cw();
                    LVIS_PASS(16, 4, l, following, accB, 2, accA, 0);
                    put_dot(0);
                    if (unit + 1 < n_units) encode(unit + 1, 0, 1);      // set A's slots: last read by pass A of this layer
                } else {
                    LVIS_PASS(16, 3, l, following, accB, 2, accA, 0);
                }
                p2_barrier();
            }
            finish(unit, 0);
            if (unit + 1 < n_units) encode(unit + 1, 2, 3);             // set B's slots: last read by pass B of layer 3
        }
#undef LVIS_PASS
#undef LVIS_PASS_AT
#undef LVIS_L0
        load_cw();
        p2_valu_only<PREC, 4, TN>(lds_, lane, t0, accB, 2, TN, cw, dot);
        put_dot(2);
        p2_barrier();
        finish(n_units - 1, 2);
        p2_barrier();                                     // every numerator is complete
        write_vis();
        p2_barrier();                                     // red / the lists / the parked encoding are free for the next item
    }
}

template <int PREC>
static int launch_lvis_p2(const unsigned char* b, const float* points, const float* normals, const float* dirs, const float* weights,
                          const unsigned char* point_mask, int n_pts, int n_lobes, float* vis, hipStream_t stream) {
    static bool done = false;
    if (!done) {
        allow_big_lds(lvis_visibility_p2_kernel<PREC>);
        done = true;
    }
    const long items = (long)n_pts * ((n_lobes + kLvisP2Chunk - 1) / kLvisP2Chunk);
    hipLaunchKernelGGL((lvis_visibility_p2_kernel<PREC>), dim3((unsigned)(items < 2048 ? items : 2048)), dim3(512), kLvisP2LdsTotal,
                       stream, b, points, normals, dirs, weights, point_mask, n_pts, n_lobes, vis);
    return launch_status();
}

int lvis_visibility_p2(const unsigned char* b, const float* points, const float* normals, const float* dirs, const float* weights,
                       const unsigned char* point_mask, int n_pts, int n_lobes, float* vis, int prec, hipStream_t stream) {
    if (prec == 3) return launch_lvis_p2<3>(b, points, normals, dirs, weights, point_mask, n_pts, n_lobes, vis, stream);
    if (prec == 1) return launch_lvis_p2<1>(b, points, normals, dirs, weights, point_mask, n_pts, n_lobes, vis, stream);
    if (prec == 2) return launch_lvis_p2<2>(b, points, normals, dirs, weights, point_mask, n_pts, n_lobes, vis, stream);   // blob: fneus_lvis_h16_pack
    return -2;
}

}  // namespace fneus
