// Forward chain of K2 (SDFNetwork.forward on the samples of a training step, reference models/fields.py:74-111) in the
// two-pass pipelined form (p2_engine.h, p2_train.h): the maths, operands and per-accumulator summation order of
// sdf_fwd_p2_kernel / sdf_fwd_grad_tp_kernel, with everything the rest of the step reads written on the way:
//   sdf [n], feature rows [n][256] (fp32), sigma'(z_l) blocks, and in training the PE / h_l / feature planes (FneusSdfStash).
// The reverse sweep (normal = d sdf / d x, the a_l planes) is a launch of its own (sdf_kernels.hip, sdf_grad_rev): it needs
// nothing from this kernel but the sigma' blocks, which went through memory inside the fused kernel as well.
#include <stdlib.h>
#ifndef FNEUS_P2_DEPTH
#define FNEUS_P2_DEPTH 2            // weight-prefetch distance of this file's passes (p2_engine.h: 3).  48 spilled registers at depth 3 (scratch reloads inside the passes), 23 at depth 2: K2's forward launch 10-22 us faster by box
#endif                              // (tools/runs/r04_ab.sh k2d2 / k2d1 / cold2)
#include "p2_train.h"
#include "fneus_kernels.h"
#include "sdf_w8.h"

namespace fneus {

// One work unit = 128 samples (4 tiles); sets A = {0, 1} (accA), B = {2, 3} (accB).  Pass schedule of a unit ("|| x" = the
// vector work inside the pass; act l = softplus + sigma' of layer l -> B fragments, planes):
//   L0.A || linear output of the previous unit, set B        L0.B || act 0 A
//   Ll.A || act l-1 B                                        Ll.B || act l A            (l = 1..7; act 7 -> also the sdf dot)
//   L8.A || act 7 B                                          L8.B || linear output A
template <int PREC, int MODE>
__global__ void __launch_bounds__(512, 1) sdf_fwd_stash_p2_kernel(const unsigned char* blob, PointSrc src, long N, SdfStash st,
                                                                   float* __restrict__ sdf_out, float* __restrict__ feat_out,
                                                                   long unit_begin, long unit_end) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    constexpr int TN = 1, NW = 8;
    constexpr bool TRAIN = MODE != 0;
    constexpr bool LO = MODE == 3 && PREC == 3;
    float* red = reinterpret_cast<float*>(lds_ + kP2LdsTotal);               // [4 tiles][NW waves][32 samples]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int t0 = wave, r = lane & 31, h = lane >> 5;
    constexpr auto& LY = kSdfLayout;
    const long units = unit_end;                 // this launch: units unit_begin .. unit_end - 1 of the (N + 127) / 128
    const long tiles = pp_tiles(N);
    const PPLane pl = pp_lane(lane);
    const unsigned voff_row = (unsigned)(r * 256 + 4 * h) * 4u;
    auto encode = [&](long unit) {          // wave w < 4: encoding of tile w of the unit -> slots 16..18 (and the PE plane)
        if (wave >= 4) return;
        const long tile = unit * 4 + wave;
        const long n = tile * 32 + r;
        const long nc = n < N ? n : N - 1;
        float x[3], pe[39], jc[39];
        load_point(src, nc, x);
        posenc<6, false>(x, pe, jc);
        BFrag<PREC> pf[kMaxKS];
        vec_to_bfrag<PREC, 39, 3, 0>(pe, pf, h);
        frags_to_lds<PREC, 3>(lds_ + wave * kP2Half, lane, 16, pf);
        if constexpr (TRAIN) {
            if (tile < tiles)
                frags_to_plane<PREC, 3>(pf, 0, st.pe_hi + (size_t)tile * 4 * kFragBytes,
                                        LO ? st.pe_lo + (size_t)tile * 4 * kFragBytes : nullptr, pl, n < N);
        }
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
        if (wave < 4 && (wave >> 1) == (hb0 >> 1) && lane < 32) {
            f32x16 b8[1];
            load_accvec<9, 8, 1>(blob, LY.L[8].bias, b8, lane);
            float s = b8[0][0];
#pragma unroll
            for (int k = 0; k < NW; ++k) s += red[(wave * NW + k) * 32 + lane];
            const long n = (unit * 4 + wave) * 32 + r;
            if (n < N) sdf_out[n] = s;
        }
    };
    // where the vector work of a pass stores: activation of layer lV (lV = 8: the linear output) of set hbV of `unit`
    auto outputs = [&](long unit, int lV, int hbV) {
        P2St so;
        const long tile = unit * 4 + hbV;                                     // first tile of the pair
        const bool ok = tile < tiles && !(lV == 3 && wave == 7);              // layer 3 has 7 tiles
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) so.vmask[hb] = (tile + hb) * 32 + r < N;
        if (lV == 8) {
            long rows = N - tile * 32;
            rows = rows < 0 ? 0 : (rows > 64 ? 64 : rows);
            // (feat_out NULL: no fp32 rows -- an empty descriptor drops the stores; the feature planes are hi + lo whenever the stash
            //  has a lo plane for them, whatever the gradient precision of the other planes: the colour network reads them)
            so.sig = p2_out_rsrc(reinterpret_cast<unsigned char*>(feat_out ? feat_out + tile * 32 * 256 : nullptr), feat_out ? (unsigned)rows * 1024u : 0u);
            if constexpr (TRAIN) {
                so.hi = p2_out_rsrc(st.feat_hi + (size_t)tile * kPPBlock, ok ? 2u * (unsigned)kPPBlock : 0u);
                if constexpr (PREC == 3)
                    so.lo = p2_out_rsrc(st.feat_lo ? st.feat_lo + (size_t)tile * kPPBlock : nullptr, (ok && st.feat_lo) ? 2u * (unsigned)kPPBlock : 0u);
            }
        } else {
            so.sig = p2_out_rsrc(st.ps + ((size_t)tile * 8 + lV) * kPPBlock, ok ? 9u * (unsigned)kPPBlock : 0u);
            if constexpr (TRAIN) {
                so.hi = p2_out_rsrc(st.h_hi + ((size_t)lV * tiles + tile) * kPPBlock, ok ? 2u * (unsigned)kPPBlock : 0u);
                if constexpr (LO) so.lo = p2_out_rsrc(st.h_lo + ((size_t)lV * tiles + tile) * kPPBlock, ok ? 2u * (unsigned)kPPBlock : 0u);
            }
        }
        return so;
    };
    f32x16 accA[TN][2], accB[TN][2], cw[TN];
    float dot[2] = {0.0f, 0.0f};
    auto load_cw = [&]() { load_accvec<8, 0, TN>(blob, LY.extra, cw, lane, t0); };   // row 0 of W_8 in accumulator layout
    const __amdgpu_buffer_rsrc_t rsrc = p2_rsrc(blob);
    auto next_of = [&](int l) { return P2Next{LY.L[l].fwd_hi, LY.L[l].fwd_lo, LY.L[l].bias, l == 3 ? 7 : (l == 8 ? 9 : 8)}; };
    P2Prime<FNEUS_P2_DEPTH, TN> pr;
    p2_prime_all<PREC, FNEUS_P2_DEPTH, TN>(pr, blob, rsrc, lane, t0, next_of(0));
    if (unit_begin + (long)blockIdx.x < units) encode(unit_begin + blockIdx.x);
    p2_barrier();
    bool first = true;
#define FNEUS_PASS(KS, NT, LMAP, ACT, L_, NX, ACCM, HBM, ACCV, HBV, TNV, SO)                                                   \
    p2_pass_st<PREC, KS, NT, LMAP, ACT, MODE>(blob, rsrc, LY.L[L_].fwd_hi, LY.L[L_].fwd_lo, pr, NX, lds_, lane, t0, ACCM, HBM, \
                                              ACCV, HBV, TNV, cw, dot, SO, pl.even, pl.odd, voff_row)
    for (long unit = unit_begin + blockIdx.x; unit < units; unit += gridDim.x) {
        asm volatile("" : "+s"(blob));
        // ---- layer 0 (3 k-steps on the encoding)
        if (first) {
            const P2St none = outputs(unit, 0, 0);
            FNEUS_PASS(3, 8, 1, 0, 0, next_of(0), accA, 0, accB, 2, TN, none);
        } else {
            const P2St so = outputs(unit - gridDim.x, 8, 2);
            FNEUS_PASS(3, 8, 1, 6, 0, next_of(0), accA, 0, accB, 2, TN, so);
        }
        p2_barrier();
        first = false;
        {
            const P2St so = outputs(unit, 0, 0);
            FNEUS_PASS(3, 8, 1, 4, 0, next_of(1), accB, 2, accA, 0, TN, so);
        }
        p2_barrier();
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            asm volatile("" : "+s"(blob));
            const int tn3 = 7 - t0 < TN ? 7 - t0 : TN;                   // layer 3 has 7 tiles: its last wave publishes none
            const int tn_prev = l - 1 == 3 ? tn3 : TN;
            const int tn_this = l == 3 ? tn3 : TN;
            const P2Next same = next_of(l), following = next_of(l + 1);
            {   // pass A: MFMAs of set {0, 1} || activation of layer l-1, set {2, 3}
                const P2St so = outputs(unit, l - 1, 2);
                if (l == 3) FNEUS_PASS(16, 7, 0, 4, 3, same, accA, 0, accB, 2, tn_prev, so);
                else if (l == 4) FNEUS_PASS(17, 8, 2, 4, 4, same, accA, 0, accB, 2, tn_prev, so);
                else FNEUS_PASS(16, 8, 0, 4, l, same, accA, 0, accB, 2, tn_prev, so);
            }
            p2_barrier();
            {   // pass B: MFMAs of set {2, 3} || activation of layer l, set {0, 1}
                const P2St so = outputs(unit, l, 0);
                if (l == 3) FNEUS_PASS(16, 7, 0, 4, 3, following, accB, 2, accA, 0, tn_this, so);
                else if (l == 4) FNEUS_PASS(17, 8, 2, 4, 4, following, accB, 2, accA, 0, tn_this, so);
                else if (l == 7) {
                    load_cw();
                    FNEUS_PASS(16, 8, 0, 5, 7, following, accB, 2, accA, 0, tn_this, so);
                } else FNEUS_PASS(16, 8, 0, 4, l, following, accB, 2, accA, 0, tn_this, so);
            }
            if (l == 7) put_dot(dot, 0);
            if (l == 5 && unit + gridDim.x < units) encode(unit + gridDim.x);     // slots 16..18 are free behind layer 4
            p2_barrier();
        }
        finish(unit, 0);
        // ---- layer 8 (linear): the 8 feature tiles by MFMA, the sdf row as the dot product above
        {
            const P2St so = outputs(unit, 7, 2);
            load_cw();
            FNEUS_PASS(16, 9, 0, 5, 8, next_of(8), accA, 0, accB, 2, TN, so);
        }
        put_dot(dot, 2);
        p2_barrier();
        finish(unit, 2);
        {
            const P2St so = outputs(unit, 8, 0);
            FNEUS_PASS(16, 9, 0, 6, 8, next_of(0), accB, 2, accA, 0, TN, so);
        }
        p2_barrier();
    }
#undef FNEUS_PASS
    if (!first) {       // tail of the last unit: the linear output of set {2, 3}
        long last = unit_begin + blockIdx.x;
        while (last + gridDim.x < units) last += gridDim.x;
        const P2St so = outputs(last, 8, 2);
        p2_linear_out_only<PREC, MODE>(lane, t0, accB, so, pl.even, pl.odd, voff_row);
    }
}

template <int PREC, int MODE>
static int launch_k2f_p2(const unsigned char* b, const PointSrc& src, long n_pts, const SdfStash& st, float* sdf_out,
                         float* feat_out, long unit_begin, long unit_end, hipStream_t stream) {
    static bool done = false;
    if (!done) {
        allow_big_lds(sdf_fwd_stash_p2_kernel<PREC, MODE>);
        done = true;
    }
    const long units = unit_end - unit_begin;
    if (units <= 0) return 0;
    hipLaunchKernelGGL((sdf_fwd_stash_p2_kernel<PREC, MODE>), dim3((unsigned)(units < 256 ? units : 256)), dim3(512),
                       kP2LdsTotal + 4 * 8 * 32 * 4, stream, b, src, n_pts, st, sdf_out, feat_out, unit_begin, unit_end);
    return launch_status();
}

// mode 0: inference (sigma' blocks + sdf + features), 1 / 3: training with bf16 / hi + lo planes
int sdf_fwd_stash_p2(const unsigned char* b, const PointSrc& src, long n_pts, const SdfStash& st, float* sdf_out, float* feat_out,
                     int prec, int mode, long unit_begin, long unit_end, hipStream_t stream) {
    if (prec == 3 && mode == 0) return launch_k2f_p2<3, 0>(b, src, n_pts, st, sdf_out, feat_out, unit_begin, unit_end, stream);
    if (prec == 3 && mode == 1) return launch_k2f_p2<3, 1>(b, src, n_pts, st, sdf_out, feat_out, unit_begin, unit_end, stream);
    if (prec == 3 && mode == 3) return launch_k2f_p2<3, 3>(b, src, n_pts, st, sdf_out, feat_out, unit_begin, unit_end, stream);
    if (prec == 1 && mode == 0) return launch_k2f_p2<1, 0>(b, src, n_pts, st, sdf_out, feat_out, unit_begin, unit_end, stream);
    if (prec == 1 && mode == 1) return launch_k2f_p2<1, 1>(b, src, n_pts, st, sdf_out, feat_out, unit_begin, unit_end, stream);
    return -2;
}

}  // namespace fneus
