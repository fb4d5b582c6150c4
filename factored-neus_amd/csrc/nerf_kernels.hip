// K7: the background NeRF++ of the womask configurations (reference models/fields.py:178-259 NeRF.forward, called from
// renderer.py:112-149 render_core_outside) and its autograd.
//   input  : 4-D inverted-sphere point (p/|p|, 1/|p|) and the view direction of every background sample
//   network: PE10(point) 84 -> 8 x (Linear 256 + ReLU), the encoding re-concatenated in front of layer 5's input
//            -> alpha_linear 1 (raw density) | feature_linear 256 -> [feature | PE4(view) 27] -> Linear 128 + ReLU -> rgb 3
//   output : raw density [N] and raw rgb [N][3]  (softplus / sigmoid / compositing stay with the caller)
// Inputs carry no gradient (every z is produced under no_grad, renderer.py:425-449), so the backward is the descending
// chain only; it writes the dL/dz planes, the weight gradients are one fneus_dw_gemm_pp launch over the stash planes.
// Everything the weight-gradient GEMM reads leaves as FRAGMENT PLANES (fneus_pp.h): the B fragments a layer's activations
// are converted into anyway, stored straight from registers (one 16-byte store per lane and fragment; hi plane, lo plane
// only in the exact-gradient mode).
// Same wave-local MFMA engine as the SDF / colour kernels: one wavefront carries a 32-sample tile through the whole net.
#define FNEUS_PREFETCH_X3 4
#define FNEUS_PREFETCH_X1 8
#include <stdlib.h>
#include "pp_engine.h"
#include "fneus_kernels.h"

#ifndef FNEUS_NERF_OCC
#define FNEUS_NERF_OCC 2      // workgroups per CU the tensor-parallel kernels of this file are compiled for (experiments: 3)
#endif

namespace fneus {

constexpr int kNerfPE = 84;     // 4 + 2 * 10 * 4
constexpr int kViewPE = 27;     // 3 + 2 * 4 * 3

// embedder.py:23-36 with input_dims = 4: [x, sin(2^0 x), cos(2^0 x), ..., sin(2^9 x), cos(2^9 x)], blocks of 4
FN_DEV void posenc4x10(const float (&x)[4], float (&pe)[kNerfPE]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) pe[c] = x[c];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float f = (float)(1 << k);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float s, co;
            fn_sincos(x[c] * f, s, co);
            pe[4 + 8 * k + c] = s;
            pe[4 + 8 * k + 4 + c] = co;
        }
    }
}

FN_DEV u32x4 relu_mask_tiles8(f32x16 (&acc)[8]) {
    u32x4 m = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool pos = acc[t][r] > 0.0f;
            acc[t][r] = pos ? acc[t][r] : 0.0f;
            m[t >> 1] |= (pos ? 1u : 0u) << ((t & 1) * 16 + r);
        }
    return m;
}

template <int TN>
FN_DEV void apply_mask(f32x16 (&acc)[TN], const u32x4 m, bool valid) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool pos = (m[t >> 1] >> ((t & 1) * 16 + r)) & 1u;
            acc[t][r] = (pos && valid) ? acc[t][r] : 0.0f;
        }
}

// how many of the launch's Ncap samples are live: all of them, or the length of the list fneus_outside_select made (device memory,
// so a step that lists a different number of samples every time keeps its launch shapes).  The planes keep the strides of Ncap.
FN_DEV long nerf_count(long Ncap, const int32_t* __restrict__ n_dev) {
    if (n_dev == nullptr) return Ncap;
    const long v = __builtin_amdgcn_readfirstlane(*n_dev);
    return v < Ncap ? (v > 0 ? v : 0) : Ncap;
}

template <int PREC, bool TRAIN>
__global__ void __launch_bounds__(64, 1) nerf_fwd_kernel(const unsigned char* blob, const float* __restrict__ pts4,
                                                         const float* __restrict__ dirs, long Ncap, NerfStash st,
                                                         float* __restrict__ density, float* __restrict__ rgb,
        const int32_t* __restrict__ n_dev) {
    const long N = nerf_count(Ncap, n_dev);
    const int lane = threadIdx.x;
    const int r = lane & 31, h = lane >> 5;
    const PPLane pl = pp_lane(lane);
    const long tiles = pp_tiles(Ncap);
    constexpr auto& LY = kNerfLayout;
    for (long tile = blockIdx.x; tile * 32 < N; tile += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const long n = tile * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        BFrag<PREC> bpe[kMaxKS];      // the encoding's 6 k-steps: layer 0 and again layer 5
        BFrag<PREC> bf[kMaxKS];
        {
            float x[4], pe[kNerfPE];
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = pts4[nc * 4 + c];
            posenc4x10(x, pe);
            vec_to_bfrag<PREC, kNerfPE, 6, 0>(pe, bpe, h);
        }
        if constexpr (TRAIN)
            frags_to_plane<PREC, 6>(bpe, 0, st.pe_hi + (size_t)tile * 6 * kFragBytes,
                                    st.pe_lo ? st.pe_lo + (size_t)tile * 6 * kFragBytes : nullptr, pl, valid);
        f32x16 acc[9];
        f32x16(&a8)[8] = reinterpret_cast<f32x16(&)[8]>(acc);
        size_t mslot = (size_t)tile * 9 * 64 + lane;
        // pts_linears.0
        load_accvec<8, 0, 8>(blob, LY.L[0].bias, a8, lane);
        dense<PREC, 6, 8, 0, 8>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, bpe, a8, lane);
        {
            const u32x4 m = relu_mask_tiles8(a8);
            if constexpr (TRAIN) st.mask[mslot] = m;
        }
        acc_to_bfrag<PREC, 8>(a8, bf);
        if constexpr (TRAIN)
            frags_to_plane<PREC, 16>(bf, 0, st.h_hi + (size_t)tile * kPPBlock, st.h_lo ? st.h_lo + (size_t)tile * kPPBlock : nullptr,
                                     pl, valid);
        // pts_linears.1 .. 7 (pack entries 1..5, 7, 8; entry 6 = the encoding's columns of pts_linears.5)
        for (int l = 1; l <= 7; ++l) {
            asm volatile("" : "+s"(blob));
            const int e = l < 6 ? l : l + 1;
            load_accvec<8, 0, 8>(blob, LY.L[e].bias, a8, lane);
            dense<PREC, 16, 8, 0, 8>(blob, LY.L[e].fwd_hi, LY.L[e].fwd_lo, bf, a8, lane);
            if (l == 5) dense<PREC, 6, 8, 0, 8>(blob, LY.L[6].fwd_hi, LY.L[6].fwd_lo, bpe, a8, lane);
            const u32x4 m = relu_mask_tiles8(a8);
            if constexpr (TRAIN) st.mask[mslot + (size_t)l * 64] = m;
            acc_to_bfrag<PREC, 8>(a8, bf);
            if constexpr (TRAIN) {
                const size_t off = ((size_t)l * tiles + tile) * kPPBlock;
                frags_to_plane<PREC, 16>(bf, 0, st.h_hi + off, st.h_lo ? st.h_lo + off : nullptr, pl, valid);
            }
        }
        // feature_linear (tiles 0..7) and alpha_linear (tile 8, row 0): no activation (fields.py:248-249)
        load_accvec<9, 0, 9>(blob, LY.L[9].bias, acc, lane);
        dense<PREC, 16, 9, 0, 9>(blob, LY.L[9].fwd_hi, LY.L[9].fwd_lo, bf, acc, lane);
        if (valid && lane < 32) density[n] = acc[8][0];
        acc_to_bfrag<PREC, 8>(a8, bf);
        if constexpr (TRAIN)
            frags_to_plane<PREC, 16>(bf, 0, st.feat_hi + (size_t)tile * kPPBlock,
                                     st.feat_lo ? st.feat_lo + (size_t)tile * kPPBlock : nullptr, pl, valid);
        {
            float d[3], pe[kViewPE], jc[kViewPE];
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = dirs[nc * 3 + c];
            posenc<4, false>(d, pe, jc);
            vec_to_bfrag<PREC, kViewPE, 2, 16>(pe, bf, h);
        }
        if constexpr (TRAIN)
            frags_to_plane<PREC, 2>(&bf[16], 0, st.dpe_hi + (size_t)tile * 2 * kFragBytes,
                                    st.dpe_lo ? st.dpe_lo + (size_t)tile * 2 * kFragBytes : nullptr, pl, valid);
        // views_linears.0: [feature | PE4(view)] -> 128, ReLU
        f32x16 v[4];
        load_accvec<4, 0, 4>(blob, LY.L[10].bias, v, lane);
        dense<PREC, 18, 4, 0, 4>(blob, LY.L[10].fwd_hi, LY.L[10].fwd_lo, bf, v, lane);
        {
            u32x4 m = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const bool pos = v[t][rr] > 0.0f;
                    v[t][rr] = pos ? v[t][rr] : 0.0f;
                    m[t >> 1] |= (pos ? 1u : 0u) << ((t & 1) * 16 + rr);
                }
            if constexpr (TRAIN) st.mask[mslot + (size_t)8 * 64] = m;
        }
        acc_to_bfrag<PREC, 4>(v, bf);
        if constexpr (TRAIN)
            frags_to_plane<PREC, 8>(bf, 0, st.hv_hi + (size_t)tile * 8 * kFragBytes,
                                    st.hv_lo ? st.hv_lo + (size_t)tile * 8 * kFragBytes : nullptr, pl, valid);
        f32x16 o[1];
        load_accvec<1, 0, 1>(blob, LY.L[11].bias, o, lane);
        dense<PREC, 8, 1, 0, 1>(blob, LY.L[11].fwd_hi, LY.L[11].fwd_lo, bf, o, lane);
        if (valid && lane < 32) {
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[n * 3 + c] = o[0][c];
        }
    }
}

template <int PREC>
__global__ void __launch_bounds__(64, 1) nerf_bwd_kernel(const unsigned char* blob, long Ncap, const float* __restrict__ d_density,
                                                         const float* __restrict__ d_rgb, NerfStash st,
        const int32_t* __restrict__ n_dev) {
    const long N = nerf_count(Ncap, n_dev);
    const int lane = threadIdx.x;
    const int r = lane & 31, h = lane >> 5;
    const PPLane pl = pp_lane(lane);
    const long tiles = pp_tiles(Ncap);
    constexpr auto& LY = kNerfLayout;
    for (long tile = blockIdx.x; tile * 32 < N; tile += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const long n = tile * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        const size_t mslot = (size_t)tile * 9 * 64 + lane;
        BFrag<PREC> bf[kMaxKS];
        // the two narrow cotangent tiles: rows 0..2 of tile 0 = d rgb, row 0 of tile 1 = d density
        f32x16 zo[2];
        zero_acc(zo);
        if (h == 0 && valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) zo[0][c] = d_rgb[nc * 3 + c];
            zo[1][0] = d_density[nc];
        }
        unsigned char* zo_hi = st.zout_hi + (size_t)tile * 4 * kFragBytes;      // zout block: fragments 0, 1 = the rgb tile,
        unsigned char* zo_lo = st.zout_lo ? st.zout_lo + (size_t)tile * 4 * kFragBytes : nullptr;    // 2, 3 = the density tile
        acc_to_bfrag<PREC, 1>(reinterpret_cast<f32x16(&)[1]>(zo[0]), bf);
        frags_to_plane<PREC, 2>(bf, 0, zo_hi, zo_lo, pl, valid);
        // rgb_linear reverse: 2 k-steps -> 4 row tiles (the 128 view-branch features), then ReLU' of views_linears.0
        f32x16 v[4];
        zero_acc(v);
        dense<PREC, 2, 4, 0, 4>(blob, LY.L[11].rev_hi, LY.L[11].rev_lo, bf, v, lane);
        apply_mask<4>(v, st.mask[mslot + (size_t)8 * 64], valid);
        acc_to_bfrag<PREC, 4>(v, bf);
        frags_to_plane<PREC, 8>(bf, 0, st.zhv_hi + (size_t)tile * 8 * kFragBytes,
                                st.zhv_lo ? st.zhv_lo + (size_t)tile * 8 * kFragBytes : nullptr, pl, valid);
        // views_linears.0 reverse onto its 256 feature inputs = dL/d feature (feature_linear has no activation)
        f32x16 a8[8];
        zero_acc(a8);
        dense<PREC, 8, 8, 0, 8>(blob, LY.L[10].rev_hi, LY.L[10].rev_lo, bf, a8, lane);
        acc_to_bfrag<PREC, 8>(a8, bf);
        frags_to_plane<PREC, 16>(bf, 0, st.zfeat_hi + (size_t)tile * kPPBlock,
                                 st.zfeat_lo ? st.zfeat_lo + (size_t)tile * kPPBlock : nullptr, pl, valid);
        acc_to_bfrag<PREC, 1, 16>(reinterpret_cast<f32x16(&)[1]>(zo[1]), bf);      // k-steps 16, 17: the density row
        frags_to_plane<PREC, 2>(&bf[16], 2, zo_hi, zo_lo, pl, valid);
        // feature_linear^T dfeature + alpha_linear^T ddensity -> dL/d h_7
        zero_acc(a8);
        dense<PREC, 18, 8, 0, 8>(blob, LY.L[9].rev_hi, LY.L[9].rev_lo, bf, a8, lane);
        for (int l = 7; l >= 0; --l) {
            asm volatile("" : "+s"(blob));
            apply_mask<8>(a8, st.mask[mslot + (size_t)l * 64], valid);
            acc_to_bfrag<PREC, 8>(a8, bf);
            {
                const size_t off = ((size_t)l * tiles + tile) * kPPBlock;
                frags_to_plane<PREC, 16>(bf, 0, st.zbar_hi + off, st.zbar_lo ? st.zbar_lo + off : nullptr, pl, valid);
            }
            if (l > 0) {
                const int e = l < 6 ? l : l + 1;       // pack entry of pts_linears.l (its h columns for l = 5)
                zero_acc(a8);
                dense<PREC, 16, 8, 0, 8>(blob, LY.L[e].rev_hi, LY.L[e].rev_lo, bf, a8, lane);
            }
        }
    }
}


// ---- tensor-parallel workgroups (tp_engine.h / pp_engine.h) --------------------------------------------------------------
// The 4 wavefronts of a workgroup share one 32-sample tile, wave w owns output tiles 2w, 2w+1 of every 256-wide layer (tile w
// of the 128-wide view layer); two workgroups per CU.  At the womask shape (2560 tiles) the one-wave kernels above run
// 2.5 rounds of 1024 resident waves -- three rounds of a ~130 us chain --, the tensor-parallel form five even rounds of a
// chain four times shorter.  LDS: 16 k-steps of fragments (hi, lo) + 6 parked k-steps for the point's encoding, which
// layer 0 and layer 5 read (k-steps 16..21; the view layer's 2 encoding k-steps reuse 16, 17 once layer 5 is done).
constexpr int kNerfTpLds = 22 * 2 * kFragBytes;
constexpr int kNerfPark = 16;                      // first parked k-step

template <int PREC>
FN_DEV unsigned char* frag_at(unsigned char* frag, int ks) {
    return frag + (size_t)ks * (PREC == 3 ? 2 : 1) * kFragBytes;
}

template <int PREC, int KS>
FN_DEV void nerf_write_frags(unsigned char* frag, int lane, int ks0, const BFrag<PREC>* b) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        *reinterpret_cast<bf16x8*>(frag + ((ks0 + i) * NPL) * kFragBytes + lane * 16) = b[i].hi;
        if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(frag + ((ks0 + i) * NPL + 1) * kFragBytes + lane * 16) = b[i].lo;
    }
}

// The 16 features of k-step `ks` of the point's encoding as one B fragment, each lane computing only the 8 features of its
// own slots (feature phi(ks, h, j) = 16 ks + 8 (j >> 2) + 4 h + (j & 3); embedder.py:23-36 with input_dims = 4: feature f < 4 is
// x_f, then octave k = (f - 4) >> 3 holds sin(2^k x_c) for c = 0..3 and cos(2^k x_c) for c = 0..3).  Spreads the 80 sincos of a
// tile's encoding over the 4 waves of the workgroup (k-steps w and w + 4) instead of leaving them to one wave.
template <int PREC>
FN_DEV void posenc4_frag(const float (&x)[4], int ks, int h, BFrag<PREC>& out) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int f = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
        float val = 0.0f;
        if (f < kNerfPE) {
            const int g = f - 4, c = f < 4 ? f : (g & 3);
            const float xc = c == 0 ? x[0] : (c == 1 ? x[1] : (c == 2 ? x[2] : x[3]));
            if (f < 4) {
                val = xc;
            } else {
                float sn, cs;
                fn_sincos(xc * (float)(1 << (g >> 3)), sn, cs);
                val = (g & 4) ? cs : sn;
            }
        }
        if constexpr (PREC == 3) {
            __bf16 a, b;
            split_bf16(val, a, b);
            out.hi[j] = a;
            out.lo[j] = b;
        } else {
            out.hi[j] = (__bf16)val;
        }
    }
}

template <int TN, int HB>
FN_DEV void bias_nh(const unsigned char* __restrict__ blob, uint32_t off, f32x16 (&acc)[TN][HB], int lane, int t0_rt) {
    const f32x16 FN_GLOBAL* __restrict__ p = reinterpret_cast<const f32x16 FN_GLOBAL*>((gblob_t)blob + off);
    const int hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const f32x16 v = p[(t0_rt + i) * 2 + hh];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) acc[i][hb] = v;
    }
}

FN_DEV void nerf_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int TN>
FN_DEV uint32_t relu_bits(f32x16 (&acc)[TN]) {      // ReLU in place; bit t * 16 + r = sign of register r of tile t
    uint32_t m = 0u;
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool pos = acc[t][r] > 0.0f;
            acc[t][r] = pos ? acc[t][r] : 0.0f;
            m |= (pos ? 1u : 0u) << (t * 16 + r);
        }
    return m;
}

template <int TN>
FN_DEV void apply_bits(f32x16 (&acc)[TN], uint32_t m, bool valid) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = (((m >> (t * 16 + r)) & 1u) && valid) ? acc[t][r] : 0.0f;
}

template <int PREC, bool TRAIN>
__global__ void __launch_bounds__(256, FNEUS_NERF_OCC) nerf_fwd_tp_kernel(const unsigned char* blob, const float* __restrict__ pts4,
                                                             const float* __restrict__ dirs, long Ncap, NerfStash st,
                                                             float* __restrict__ density, float* __restrict__ rgb,
        const int32_t* __restrict__ n_dev) {
    const long N = nerf_count(Ncap, n_dev);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    unsigned char* park = frag_at<PREC>(frag, kNerfPark);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    const long tiles = pp_tiles(Ncap);
    const bool lo_planes = TRAIN && PREC == 3 && st.h_lo != nullptr;
    constexpr auto& LY = kNerfLayout;
    for (long tile = blockIdx.x; tile * 32 < N; tile += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const long n = tile * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        BFrag<PREC> bf[kMaxKS], bx[kMaxKS];
        uint32_t* mrow = reinterpret_cast<uint32_t*>(st.mask + (size_t)tile * 9 * 64 + lane);       // [9 slots][64 lanes] x 4 words
        nerf_barrier();                                   // the previous tile's fragments are consumed
        {                                                 // the point's encoding: 6 k-steps, parked for layers 0 and 5;
            float x[4];                                   // wave w builds k-steps w and w + 4
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = pts4[nc * 4 + c];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ks = wave + 4 * i;
                if (ks < 6) {
                    BFrag<PREC> one[1];
                    posenc4_frag<PREC>(x, ks, h, one[0]);
                    nerf_write_frags<PREC, 1>(frag, lane, kNerfPark + ks, one);
                    if constexpr (TRAIN)
                        frags_to_plane<PREC, 1>(one, ks, st.pe_hi + (size_t)tile * 6 * kFragBytes,
                                                lo_planes ? st.pe_lo + (size_t)tile * 6 * kFragBytes : nullptr, pl, valid);
                }
            }
        }
        nerf_barrier();
        f32x16 acc[2];
        // pts_linears.0
        tp_operands<PREC, 6>(park, lane, bx);
        load_accvec<8, 0, 2>(blob, LY.L[0].bias, acc, lane, t0);
        tp_dense<PREC, 6, 8, 0, 2>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, park, bx, acc, lane, t0);
#pragma unroll 1
        for (int l = 0; l <= 7; ++l) {
            if (l > 0) {
                const int e = l < 6 ? l : l + 1;          // pack entry of pts_linears.l (its h columns for l = 5)
                load_accvec<8, 0, 2>(blob, LY.L[e].bias, acc, lane, t0);
                tp_dense<PREC, 16, 8, 0, 2>(blob, LY.L[e].fwd_hi, LY.L[e].fwd_lo, frag, bf, acc, lane, t0);
                if (l == 5) tp_dense<PREC, 6, 8, 0, 2>(blob, LY.L[6].fwd_hi, LY.L[6].fwd_lo, park, bx, acc, lane, t0);
            }
            const uint32_t m = relu_bits<2>(acc);
            if constexpr (TRAIN) mrow[(size_t)l * 64 * 4 + wave] = m;
            const size_t off = ((size_t)l * tiles + tile) * kPPBlock;
            tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, TRAIN ? st.h_hi + off : nullptr, lo_planes ? st.h_lo + off : nullptr,
                                          pl, valid);
            tp_operands<PREC, 16>(frag, lane, bf);
        }
        // feature_linear (tiles 0..7, two per wave) and alpha_linear (tile 8, row 0: wave 0); no activation (fields.py:248-249)
        load_accvec<9, 0, 2>(blob, LY.L[9].bias, acc, lane, t0);
        tp_dense<PREC, 16, 9, 0, 2>(blob, LY.L[9].fwd_hi, LY.L[9].fwd_lo, frag, bf, acc, lane, t0);
        if (wave == 0) {
            f32x16 a1[1];
            load_accvec<9, 8, 1>(blob, LY.L[9].bias, a1, lane);
            tp_dense<PREC, 16, 9, 8, 1>(blob, LY.L[9].fwd_hi, LY.L[9].fwd_lo, frag, bf, a1, lane);
            if (valid && lane < 32) density[n] = a1[0][0];
        }
        BFrag<PREC> bd[2];
        if (wave == 1) {                                  // PE4 of the view direction: k-steps 16, 17 of the view layer
            float d[3], pe[kViewPE], jc[kViewPE];
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = dirs[nc * 3 + c];
            posenc<4, false>(d, pe, jc);
            BFrag<PREC> tmp[kMaxKS];
            vec_to_bfrag<PREC, kViewPE, 2, 0>(pe, tmp, h);
            bd[0] = tmp[0];
            bd[1] = tmp[1];
            if constexpr (TRAIN)
                frags_to_plane<PREC, 2>(bd, 0, st.dpe_hi + (size_t)tile * 2 * kFragBytes,
                                        lo_planes ? st.dpe_lo + (size_t)tile * 2 * kFragBytes : nullptr, pl, valid);
        }
        tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, TRAIN ? st.feat_hi + (size_t)tile * kPPBlock : nullptr,
                                      lo_planes ? st.feat_lo + (size_t)tile * kPPBlock : nullptr, pl, valid,
                                      wave == 1 ? bd : nullptr, kNerfPark, 2);
        tp_operands<PREC, 18>(frag, lane, bf);
        // views_linears.0: [feature | PE4(view)] -> 128 + ReLU; wave w owns tile w
        f32x16 v[1];
        load_accvec<4, 0, 1>(blob, LY.L[10].bias, v, lane, wave);
        tp_dense<PREC, 18, 4, 0, 1>(blob, LY.L[10].fwd_hi, LY.L[10].fwd_lo, frag, bf, v, lane, wave);
        {
            const uint32_t m = relu_bits<1>(v);
            if constexpr (TRAIN) reinterpret_cast<uint16_t*>(mrow + (size_t)8 * 64 * 4)[wave] = (uint16_t)m;
        }
        tp_exchange_pp<PREC, 1, true>(frag, lane, wave, v, TRAIN ? st.hv_hi + (size_t)tile * 8 * kFragBytes : nullptr,
                                      lo_planes ? st.hv_lo + (size_t)tile * 8 * kFragBytes : nullptr, pl, valid);
        tp_operands<PREC, 8>(frag, lane, bf);
        if (wave == 0) {
            f32x16 o[1];
            load_accvec<1, 0, 1>(blob, LY.L[11].bias, o, lane);
            tp_dense<PREC, 8, 1, 0, 1>(blob, LY.L[11].fwd_hi, LY.L[11].fwd_lo, frag, bf, o, lane);
            if (valid && lane < 32) {
#pragma unroll
                for (int c = 0; c < 3; ++c) rgb[n * 3 + c] = o[0][c];
            }
        }
    }
}

template <int PREC>
__global__ void __launch_bounds__(256, FNEUS_NERF_OCC) nerf_bwd_tp_kernel(const unsigned char* blob, long Ncap, const float* __restrict__ d_density,
                                                             const float* __restrict__ d_rgb, NerfStash st,
        const int32_t* __restrict__ n_dev) {
    const long N = nerf_count(Ncap, n_dev);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    const long tiles = pp_tiles(Ncap);
    const bool lo_planes = PREC == 3 && st.zbar_lo != nullptr;
    constexpr auto& LY = kNerfLayout;
    for (long tile = blockIdx.x; tile * 32 < N; tile += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const long n = tile * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        const uint32_t* mrow = reinterpret_cast<const uint32_t*>(st.mask + (size_t)tile * 9 * 64 + lane);
        BFrag<PREC> bf[kMaxKS];
        unsigned char* zo_hi = st.zout_hi + (size_t)tile * 4 * kFragBytes;      // zout block: fragments 0, 1 = the rgb tile,
        unsigned char* zo_lo = lo_planes ? st.zout_lo + (size_t)tile * 4 * kFragBytes : nullptr;       // 2, 3 = the density tile
        BFrag<PREC> bden[2];                              // the density tile's fragments (wave 0), published with dL/d feature
        if (wave == 0) {
            f32x16 zo[2];
            zero_acc(zo);
            if (h == 0 && valid) {
#pragma unroll
                for (int c = 0; c < 3; ++c) zo[0][c] = d_rgb[nc * 3 + c];
                zo[1][0] = d_density[nc];
            }
            BFrag<PREC> tmp[kMaxKS];
            acc_to_bfrag<PREC, 1>(reinterpret_cast<f32x16(&)[1]>(zo[1]), tmp);
            bden[0] = tmp[0];
            bden[1] = tmp[1];
            frags_to_plane<PREC, 2>(bden, 2, zo_hi, zo_lo, pl, valid);
            tp_exchange_pp<PREC, 1, true>(frag, lane, 0, reinterpret_cast<f32x16(&)[1]>(zo[0]), zo_hi, zo_lo, pl, valid);
        } else {
            nerf_barrier();
            nerf_barrier();
        }
        tp_operands<PREC, 2>(frag, lane, bf);
        // rgb_linear reverse: 2 k-steps -> the 128 view-branch features (wave w: tile w), then ReLU' of views_linears.0
        f32x16 v[1];
        zero_acc(v);
        tp_dense<PREC, 2, 4, 0, 1>(blob, LY.L[11].rev_hi, LY.L[11].rev_lo, frag, bf, v, lane, wave);
        apply_bits<1>(v, reinterpret_cast<const uint16_t*>(mrow + (size_t)8 * 64 * 4)[wave], valid);
        tp_exchange_pp<PREC, 1, true>(frag, lane, wave, v, st.zhv_hi + (size_t)tile * 8 * kFragBytes,
                                      lo_planes ? st.zhv_lo + (size_t)tile * 8 * kFragBytes : nullptr, pl, valid);
        tp_operands<PREC, 8>(frag, lane, bf);
        // views_linears.0 reverse onto its 256 feature inputs = dL/d feature (feature_linear has no activation)
        f32x16 acc[2];
        zero_acc(acc);
        tp_dense<PREC, 8, 8, 0, 2>(blob, LY.L[10].rev_hi, LY.L[10].rev_lo, frag, bf, acc, lane, t0);
        tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, st.zfeat_hi + (size_t)tile * kPPBlock,
                                      lo_planes ? st.zfeat_lo + (size_t)tile * kPPBlock : nullptr, pl, valid,
                                      wave == 0 ? bden : nullptr, 16, 2);          // k-steps 16, 17: the density row
        tp_operands<PREC, 18>(frag, lane, bf);
        // feature_linear^T dfeature + alpha_linear^T ddensity -> dL/d h_7
        zero_acc(acc);
        tp_dense<PREC, 18, 8, 0, 2>(blob, LY.L[9].rev_hi, LY.L[9].rev_lo, frag, bf, acc, lane, t0);
#pragma unroll 1
        for (int l = 7; l >= 0; --l) {
            apply_bits<2>(acc, mrow[(size_t)l * 64 * 4 + wave], valid);
            const size_t off = ((size_t)l * tiles + tile) * kPPBlock;
            tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, st.zbar_hi + off, lo_planes ? st.zbar_lo + off : nullptr, pl, valid);
            if (l > 0) {
                tp_operands<PREC, 16>(frag, lane, bf);
                const int e = l < 6 ? l : l + 1;
                zero_acc(acc);
                tp_dense<PREC, 16, 8, 0, 2>(blob, LY.L[e].rev_hi, LY.L[e].rev_lo, frag, bf, acc, lane, t0);
            }
        }
    }
}


// ---- 64-sample workgroups (HB = 2: two tiles share one pass over the weight fragments, dense_ldsb_h) --------------------------
// The tensor-parallel kernels are bound by the L2 weight stream; as for K3, K1 and the colour network
// two tiles per workgroup halve it.  LDS: 18 k-steps per half (the view layer's 16 + 2), two workgroups per CU; the point's
// encoding lives in REGISTERS of the wave that built it (k-steps w and w + 4 of both halves) and is written into k-steps 0..5
// for layer 0 and again, behind layer 5's first product, for its second one.
constexpr int kNerfHalf = 18 * 2 * kFragBytes;

template <int PREC, bool TRAIN>
__global__ void __launch_bounds__(256, 2) nerf_fwd_tph_kernel(const unsigned char* blob, const float* __restrict__ pts4,
                                                              const float* __restrict__ dirs, long Ncap, NerfStash st,
                                                              float* __restrict__ density, float* __restrict__ rgb,
        const int32_t* __restrict__ n_dev) {
    const long N = nerf_count(Ncap, n_dev);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    constexpr int HB = 2, HALF = kNerfHalf;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    const long tiles = pp_tiles(Ncap);
    const long groups = (N + 32 * HB - 1) / (32 * HB);
    const bool lo_planes = TRAIN && PREC == 3 && st.h_lo != nullptr;
    constexpr auto& LY = kNerfLayout;
    for (long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        asm volatile("" : "+s"(blob));
        long tile[HB], n[HB], nc[HB];
        bool valid[HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            tile[hb] = grp * HB + hb;
            n[hb] = tile[hb] * 32 + r;
            valid[hb] = n[hb] < N;
            nc[hb] = valid[hb] ? n[hb] : N - 1;
        }
        uint32_t* mrow[HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) mrow[hb] = reinterpret_cast<uint32_t*>(st.mask + (size_t)tile[hb] * 9 * 64 + lane);
        // the points' encodings: wave w builds k-steps w and w + 4 (< 6) of both halves and keeps them
        BFrag<PREC> pef[HB][2];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            float x[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = pts4[nc[hb] * 4 + c];
            posenc4_frag<PREC>(x, wave, h, pef[hb][0]);
            if (wave < 2) posenc4_frag<PREC>(x, wave + 4, h, pef[hb][1]);
            if constexpr (TRAIN) {
                unsigned char* ph = st.pe_hi + (size_t)tile[hb] * 6 * kFragBytes;
                unsigned char* plo = lo_planes ? st.pe_lo + (size_t)tile[hb] * 6 * kFragBytes : nullptr;
                frags_to_plane<PREC, 1>(&pef[hb][0], wave, ph, plo, pl, valid[hb]);
                if (wave < 2) frags_to_plane<PREC, 1>(&pef[hb][1], wave + 4, ph, plo, pl, valid[hb]);
            }
        }
        auto write_pe = [&]() {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                nerf_write_frags<PREC, 1>(frag + hb * HALF, lane, wave, &pef[hb][0]);
                if (wave < 2) nerf_write_frags<PREC, 1>(frag + hb * HALF, lane, wave + 4, &pef[hb][1]);
            }
        };
        nerf_barrier();                                   // the previous group's fragments are consumed
        write_pe();
        nerf_barrier();
        f32x16 acc[2][HB];
        bias_nh<2, HB>(blob, LY.L[0].bias, acc, lane, t0);
        tph_dense<PREC, 6, 8, 0, 2, true, HB, HALF>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, frag, acc, lane, t0);
#pragma unroll 1
        for (int l = 0; l <= 7; ++l) {
            if (l > 0) {
                const int e = l < 6 ? l : l + 1;
                bias_nh<2, HB>(blob, LY.L[e].bias, acc, lane, t0);
                tph_dense<PREC, 16, 8, 0, 2, true, HB, HALF>(blob, LY.L[e].fwd_hi, LY.L[e].fwd_lo, frag, acc, lane, t0);
                if (l == 5) {                             // the re-concatenated encoding: a second product into the same tiles
                    nerf_barrier();
                    write_pe();
                    nerf_barrier();
                    tph_dense<PREC, 6, 8, 0, 2, true, HB, HALF>(blob, LY.L[6].fwd_hi, LY.L[6].fwd_lo, frag, acc, lane, t0);
                }
            }
            unsigned char *ph[HB], *plo[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                f32x16 two[2] = {acc[0][hb], acc[1][hb]};
                const uint32_t m = relu_bits<2>(two);
                acc[0][hb] = two[0];
                acc[1][hb] = two[1];
                if constexpr (TRAIN) mrow[hb][(size_t)l * 64 * 4 + wave] = m;
                const size_t off = ((size_t)l * tiles + tile[hb]) * kPPBlock;
                ph[hb] = TRAIN ? st.h_hi + off : nullptr;
                plo[hb] = lo_planes ? st.h_lo + off : nullptr;
            }
            tph_exchange<PREC, 2, true, HB, HALF>(frag, lane, t0, acc, ph, plo, pl, valid);
        }
        // feature_linear (tiles 0..7) and alpha_linear (tile 8, row 0: wave 0); no activation
        bias_nh<2, HB>(blob, LY.L[9].bias, acc, lane, t0);
        tph_dense<PREC, 16, 9, 0, 2, true, HB, HALF>(blob, LY.L[9].fwd_hi, LY.L[9].fwd_lo, frag, acc, lane, t0);
        if (wave == 0) {
            f32x16 a1[1][HB];
            bias_nh<1, HB>(blob, LY.L[9].bias, a1, lane, 8);
            tph_dense<PREC, 16, 9, 8, 1, true, HB, HALF>(blob, LY.L[9].fwd_hi, LY.L[9].fwd_lo, frag, a1, lane);
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
                if (valid[hb] && lane < 32) density[n[hb]] = a1[0][hb][0];
        }
        BFrag<PREC> bd[HB * 3];
        if (wave == 1) {                                  // PE4 of the view directions: k-steps 16, 17 of the view layer
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                float d[3], pe[kViewPE], jc[kViewPE];
#pragma unroll
                for (int c = 0; c < 3; ++c) d[c] = dirs[nc[hb] * 3 + c];
                posenc<4, false>(d, pe, jc);
                BFrag<PREC> tmp[kMaxKS];
                vec_to_bfrag<PREC, kViewPE, 2, 0>(pe, tmp, h);
                bd[hb * 3] = tmp[0];
                bd[hb * 3 + 1] = tmp[1];
                if constexpr (TRAIN)
                    frags_to_plane<PREC, 2>(&bd[hb * 3], 0, st.dpe_hi + (size_t)tile[hb] * 2 * kFragBytes,
                                            lo_planes ? st.dpe_lo + (size_t)tile[hb] * 2 * kFragBytes : nullptr, pl, valid[hb]);
            }
        }
        {
            unsigned char *ph[HB], *plo[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                ph[hb] = TRAIN ? st.feat_hi + (size_t)tile[hb] * kPPBlock : nullptr;
                plo[hb] = lo_planes ? st.feat_lo + (size_t)tile[hb] * kPPBlock : nullptr;
            }
            tph_exchange<PREC, 2, true, HB, HALF>(frag, lane, t0, acc, ph, plo, pl, valid, wave == 1 ? bd : nullptr, 16, 2);
        }
        // views_linears.0: [feature | PE4(view)] -> 128 + ReLU; wave w owns tile w
        f32x16 v[1][HB];
        bias_nh<1, HB>(blob, LY.L[10].bias, v, lane, wave);
        tph_dense<PREC, 18, 4, 0, 1, true, HB, HALF>(blob, LY.L[10].fwd_hi, LY.L[10].fwd_lo, frag, v, lane, wave);
        {
            unsigned char *ph[HB], *plo[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                f32x16 one[1] = {v[0][hb]};
                const uint32_t m = relu_bits<1>(one);
                v[0][hb] = one[0];
                if constexpr (TRAIN) reinterpret_cast<uint16_t*>(mrow[hb] + (size_t)8 * 64 * 4)[wave] = (uint16_t)m;
                ph[hb] = TRAIN ? st.hv_hi + (size_t)tile[hb] * 8 * kFragBytes : nullptr;
                plo[hb] = lo_planes ? st.hv_lo + (size_t)tile[hb] * 8 * kFragBytes : nullptr;
            }
            tph_exchange<PREC, 1, true, HB, HALF>(frag, lane, wave, v, ph, plo, pl, valid);
        }
        if (wave == 0) {
            f32x16 o[1][HB];
            bias_nh<1, HB>(blob, LY.L[11].bias, o, lane, 0);
            tph_dense<PREC, 8, 1, 0, 1, true, HB, HALF>(blob, LY.L[11].fwd_hi, LY.L[11].fwd_lo, frag, o, lane);
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
                if (valid[hb] && lane < 32) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) rgb[n[hb] * 3 + c] = o[0][hb][c];
                }
        }
    }
}

// XP 1 (with PREC 3, bf16 zbar planes): the chain runs on the bf16 cotangents those planes hold -- W hi + lo against one bf16 fragment, two
// MFMAs per product (FNEUS_NERF_XHI=0: hi + lo cotangents inside the chain)
template <int PREC, int XP>
__global__ void __launch_bounds__(256, 2) nerf_bwd_tph_kernel(const unsigned char* blob, long Ncap, const float* __restrict__ d_density,
                                                              const float* __restrict__ d_rgb, NerfStash st,
        const int32_t* __restrict__ n_dev) {
    const long N = nerf_count(Ncap, n_dev);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    constexpr int HB = 2, HALF = kNerfHalf;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    const long tiles = pp_tiles(Ncap);
    const long groups = (N + 32 * HB - 1) / (32 * HB);
    const bool lo_planes = XP == 3 && st.zbar_lo != nullptr;
    constexpr auto& LY = kNerfLayout;
    for (long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        asm volatile("" : "+s"(blob));
        long tile[HB], nc[HB];
        bool valid[HB];
        const uint32_t* mrow[HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            tile[hb] = grp * HB + hb;
            const long n = tile[hb] * 32 + r;
            valid[hb] = n < N;
            nc[hb] = valid[hb] ? n : N - 1;
            mrow[hb] = reinterpret_cast<const uint32_t*>(st.mask + (size_t)tile[hb] * 9 * 64 + lane);
        }
        BFrag<XP> bden[HB * 3];                         // the density tiles' fragments (wave 0), published with dL/d feature
        if (wave == 0) {
            f32x16 z0[1][HB];
            unsigned char *zh[HB], *zl[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                f32x16 zo[2];
                zero_acc(zo);
                if (h == 0 && valid[hb]) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) zo[0][c] = d_rgb[nc[hb] * 3 + c];
                    zo[1][0] = d_density[nc[hb]];
                }
                z0[0][hb] = zo[0];
                zh[hb] = st.zout_hi + (size_t)tile[hb] * 4 * kFragBytes;
                zl[hb] = lo_planes ? st.zout_lo + (size_t)tile[hb] * 4 * kFragBytes : nullptr;
                BFrag<XP> tmp[kMaxKS];
                acc_to_bfrag<XP, 1>(reinterpret_cast<f32x16(&)[1]>(zo[1]), tmp);
                bden[hb * 3] = tmp[0];
                bden[hb * 3 + 1] = tmp[1];
                frags_to_plane<XP, 2>(&bden[hb * 3], 2, zh[hb], zl[hb], pl, valid[hb]);
            }
            tph_exchange<XP, 1, true, HB, HALF>(frag, lane, 0, z0, zh, zl, pl, valid);
        } else {
            nerf_barrier();
            nerf_barrier();
        }
        // rgb_linear reverse -> the 128 view-branch features (wave w: tile w), ReLU' of views_linears.0
        f32x16 v[1][HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
            for (int e = 0; e < 16; ++e) v[0][hb][e] = 0.0f;
        tph_dense<PREC, 2, 4, 0, 1, true, HB, HALF, XP>(blob, LY.L[11].rev_hi, LY.L[11].rev_lo, frag, v, lane, wave);
        {
            unsigned char *ph[HB], *plo[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                f32x16 one[1] = {v[0][hb]};
                apply_bits<1>(one, reinterpret_cast<const uint16_t*>(mrow[hb] + (size_t)8 * 64 * 4)[wave], valid[hb]);
                v[0][hb] = one[0];
                ph[hb] = st.zhv_hi + (size_t)tile[hb] * 8 * kFragBytes;
                plo[hb] = lo_planes ? st.zhv_lo + (size_t)tile[hb] * 8 * kFragBytes : nullptr;
            }
            tph_exchange<XP, 1, true, HB, HALF>(frag, lane, wave, v, ph, plo, pl, valid);
        }
        f32x16 acc[2][HB];
        auto zero2 = [&]() {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t][hb][e] = 0.0f;
        };
        // views_linears.0 reverse onto its 256 feature inputs = dL/d feature
        zero2();
        tph_dense<PREC, 8, 8, 0, 2, true, HB, HALF, XP>(blob, LY.L[10].rev_hi, LY.L[10].rev_lo, frag, acc, lane, t0);
        {
            unsigned char *ph[HB], *plo[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                ph[hb] = st.zfeat_hi + (size_t)tile[hb] * kPPBlock;
                plo[hb] = lo_planes ? st.zfeat_lo + (size_t)tile[hb] * kPPBlock : nullptr;
            }
            tph_exchange<XP, 2, true, HB, HALF>(frag, lane, t0, acc, ph, plo, pl, valid, wave == 0 ? bden : nullptr, 16, 2);
        }
        // feature_linear^T dfeature + alpha_linear^T ddensity -> dL/d h_7
        zero2();
        tph_dense<PREC, 18, 8, 0, 2, true, HB, HALF, XP>(blob, LY.L[9].rev_hi, LY.L[9].rev_lo, frag, acc, lane, t0);
#pragma unroll 1
        for (int l = 7; l >= 0; --l) {
            unsigned char *ph[HB], *plo[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                f32x16 two[2] = {acc[0][hb], acc[1][hb]};
                apply_bits<2>(two, mrow[hb][(size_t)l * 64 * 4 + wave], valid[hb]);
                acc[0][hb] = two[0];
                acc[1][hb] = two[1];
                const size_t off = ((size_t)l * tiles + tile[hb]) * kPPBlock;
                ph[hb] = st.zbar_hi + off;
                plo[hb] = lo_planes ? st.zbar_lo + off : nullptr;
            }
            tph_exchange<XP, 2, true, HB, HALF>(frag, lane, t0, acc, ph, plo, pl, valid);
            if (l > 0) {
                const int e = l < 6 ? l : l + 1;
                zero2();
                tph_dense<PREC, 16, 8, 0, 2, true, HB, HALF, XP>(blob, LY.L[e].rev_hi, LY.L[e].rev_lo, frag, acc, lane, t0);
            }
        }
    }
}

}  // namespace fneus

using namespace fneus;

// FNEUS_K7_TP=0 selects the one-wave-per-tile kernels (comparison runs); read at every call
static inline bool nerf_use_tp() {
    const char* e = getenv("FNEUS_K7_TP");
    return !(e && e[0] == '0');
}

// 64-sample workgroups for chip-filling launches (>= 1024 tiles); FNEUS_K7_HB=1 keeps 32 samples per workgroup
static inline bool nerf_use_hb2(long n_tiles) {
    const char* e = getenv("FNEUS_K7_HB");
    if (e && e[0] == '1') return false;
    return nerf_use_tp() && n_tiles >= 1024;
}

static inline int nerf_grid(long n_tiles) {
    long g = n_tiles;
    const long cap = 256 * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int fneus_nerf_bg_fwd(const void* blob, const float* pts4, const float* dirs, long n_pts,
                                 const FneusNerfStash* stash, float* density, float* rgb, int prec, int train,
                                 const int32_t* n_dev, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    if (!blob || !pts4 || !dirs || !density || !rgb) return -2;
    if (train && !stash) return -2;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(blob);
    NerfStash st = stash ? NerfStash(*stash) : NerfStash();
    if (train && (!st.pe_hi || !st.h_hi || !st.feat_hi || !st.dpe_hi || !st.hv_hi || !st.mask)) return -2;
    if (nerf_use_hb2((n_pts + 31) / 32)) {
        const long ng = (n_pts + 63) / 64;
        dim3 g2((unsigned)(ng < 2048 ? ng : 2048)), b2(256);
#define FNEUS_NERF_TPH(KERNEL, ...)                                                                      \
        do {                                                                                             \
            static bool done = false;                                                                    \
            if (!done) { fneus::allow_big_lds(KERNEL); done = true; }                                    \
            hipLaunchKernelGGL((KERNEL), g2, b2, 2 * fneus::kNerfHalf, stream, __VA_ARGS__);              \
        } while (0)
        if (prec == 3 && train) FNEUS_NERF_TPH((nerf_fwd_tph_kernel<3, true>), b, pts4, dirs, n_pts, st, density, rgb, n_dev);
        else if (prec == 3) FNEUS_NERF_TPH((nerf_fwd_tph_kernel<3, false>), b, pts4, dirs, n_pts, st, density, rgb, n_dev);
        else if (prec == 1 && train) FNEUS_NERF_TPH((nerf_fwd_tph_kernel<1, true>), b, pts4, dirs, n_pts, st, density, rgb, n_dev);
        else if (prec == 1) FNEUS_NERF_TPH((nerf_fwd_tph_kernel<1, false>), b, pts4, dirs, n_pts, st, density, rgb, n_dev);
        else return -2;
        return fneus::launch_status();
    }
    if (nerf_use_tp()) {
        const long nt = (n_pts + 31) / 32;
        dim3 g2((unsigned)(nt < 2048 ? nt : 2048)), b2(256);
#define FNEUS_NERF_TP(KERNEL, ...)                                                                       \
        do {                                                                                             \
            static bool done = false;                                                                    \
            if (!done) { fneus::allow_big_lds(KERNEL); done = true; }                                    \
            hipLaunchKernelGGL((KERNEL), g2, b2, fneus::kNerfTpLds, stream, __VA_ARGS__);                 \
        } while (0)
        if (prec == 3 && train) FNEUS_NERF_TP((nerf_fwd_tp_kernel<3, true>), b, pts4, dirs, n_pts, st, density, rgb, n_dev);
        else if (prec == 3) FNEUS_NERF_TP((nerf_fwd_tp_kernel<3, false>), b, pts4, dirs, n_pts, st, density, rgb, n_dev);
        else if (prec == 1 && train) FNEUS_NERF_TP((nerf_fwd_tp_kernel<1, true>), b, pts4, dirs, n_pts, st, density, rgb, n_dev);
        else if (prec == 1) FNEUS_NERF_TP((nerf_fwd_tp_kernel<1, false>), b, pts4, dirs, n_pts, st, density, rgb, n_dev);
        else return -2;
        return fneus::launch_status();
    }
    dim3 grid(nerf_grid((n_pts + 31) / 32)), blk(64);
    if (prec == 3 && train)
        hipLaunchKernelGGL((nerf_fwd_kernel<3, true>), grid, blk, 0, stream, b, pts4, dirs, n_pts, st, density, rgb, n_dev);
    else if (prec == 3)
        hipLaunchKernelGGL((nerf_fwd_kernel<3, false>), grid, blk, 0, stream, b, pts4, dirs, n_pts, st, density, rgb, n_dev);
    else if (prec == 1 && train)
        hipLaunchKernelGGL((nerf_fwd_kernel<1, true>), grid, blk, 0, stream, b, pts4, dirs, n_pts, st, density, rgb, n_dev);
    else if (prec == 1)
        hipLaunchKernelGGL((nerf_fwd_kernel<1, false>), grid, blk, 0, stream, b, pts4, dirs, n_pts, st, density, rgb, n_dev);
    else
        return -2;
    return fneus::launch_status();
}

extern "C" int fneus_nerf_bg_bwd(const void* blob, long n_pts, const float* d_density, const float* d_rgb,
                                 const FneusNerfStash* stash, int prec, const int32_t* n_dev, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    if (!blob || !d_density || !d_rgb || !stash) return -2;
    NerfStash st(*stash);
    if (!st.mask || !st.zbar_hi || !st.zfeat_hi || !st.zhv_hi || !st.zout_hi) return -2;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(blob);
    if (nerf_use_hb2((n_pts + 31) / 32)) {
        const long ng = (n_pts + 63) / 64;
        dim3 g2((unsigned)(ng < 2048 ? ng : 2048)), b2(256);
        const char* xe = getenv("FNEUS_NERF_XHI");
        const bool xhi = (xe ? atoi(xe) != 0 : true) && st.zbar_lo == nullptr;       // bf16 planes: the chain on the values they hold
        if (prec == 3 && xhi) FNEUS_NERF_TPH((nerf_bwd_tph_kernel<3, 1>), b, n_pts, d_density, d_rgb, st, n_dev);
        else if (prec == 3) FNEUS_NERF_TPH((nerf_bwd_tph_kernel<3, 3>), b, n_pts, d_density, d_rgb, st, n_dev);
        else if (prec == 1) FNEUS_NERF_TPH((nerf_bwd_tph_kernel<1, 1>), b, n_pts, d_density, d_rgb, st, n_dev);
        else return -2;
        return fneus::launch_status();
    }
    if (nerf_use_tp()) {
        const long nt = (n_pts + 31) / 32;
        dim3 g2((unsigned)(nt < 2048 ? nt : 2048)), b2(256);
        if (prec == 3) FNEUS_NERF_TP(nerf_bwd_tp_kernel<3>, b, n_pts, d_density, d_rgb, st, n_dev);
        else if (prec == 1) FNEUS_NERF_TP(nerf_bwd_tp_kernel<1>, b, n_pts, d_density, d_rgb, st, n_dev);
        else return -2;
        return fneus::launch_status();
    }
    dim3 grid(nerf_grid((n_pts + 31) / 32)), blk(64);
    if (prec == 3)
        hipLaunchKernelGGL(nerf_bwd_kernel<3>, grid, blk, 0, stream, b, n_pts, d_density, d_rgb, st, n_dev);
    else if (prec == 1)
        hipLaunchKernelGGL(nerf_bwd_kernel<1>, grid, blk, 0, stream, b, n_pts, d_density, d_rgb, st, n_dev);
    else
        return -2;
    return fneus::launch_status();
}
