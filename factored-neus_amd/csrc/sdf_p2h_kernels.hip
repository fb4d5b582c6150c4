// K1 (SDFNetwork.sdf, reference models/fields.py:93-95) in the two-pass pipelined form on 64-sample workgroups, two per CU
// (p2h_engine.h): the maths, operands and per-accumulator summation order of sdf_fwd_p2_kernel.
#include <stdlib.h>
#ifndef FNEUS_P2H_DEPTH
#define FNEUS_P2H_DEPTH 2           // weight-prefetch distance in k-steps: 3 does not fit 256 registers with two output tiles per wave
#endif
#define FNEUS_P2_DEPTH FNEUS_P2H_DEPTH
#include "p2h_engine.h"
#include "fneus_kernels.h"
#include "sdf_w8.h"

namespace fneus {

// One work unit = 64 samples (2 tiles); workgroup b takes units b, b + gridDim, ...  Pass schedule of a unit (A = tile 0,
// B = tile 1; "|| x" = the vector work inside the pass):
//   L0.A || tail of the previous unit (act 7 B -> dot)      L0.B || act 0 A
//   Ll.A || act l-1 B                                       Ll.B || act l A                (l = 1..7; act 7 A -> dot)
// The encoding of the NEXT unit is written to slots 16..18 behind layer 4 (their last reader in this unit).
template <int PREC>
__global__ void __launch_bounds__(256, 2) sdf_fwd_p2h_kernel(const unsigned char* blob, PointSrc src, long N,
                                                              float* __restrict__ sdf_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    constexpr int TN = 2, NW = 4;
    float* red = reinterpret_cast<float*>(lds_ + kP2hLdsTotal);               // [2 tiles][NW waves][32 samples]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int t0 = TN * wave, r = lane & 31, h = lane >> 5;
    constexpr auto& LY = kSdfLayout;
    const long units = (N + 63) / 64;
    auto encode = [&](long unit) {          // wave w < 2: encoding of tile w of the unit -> slots 16..18
        if (wave >= 2) return;
        const long n = (unit * 2 + wave) * 32 + r;
        const long nc = n < N ? n : N - 1;
        float x[3], pe[39], jc[39];
        load_point(src, nc, x);
        posenc<6, false>(x, pe, jc);
        BFrag<PREC> pf[kMaxKS];
        vec_to_bfrag<PREC, 39, 3, 0>(pe, pf, h);
        frags_to_lds<PREC, 3>(lds_ + wave * kP2Half, lane, 16, pf);
    };
    auto put_dot = [&](float& dot, int tile) {
        const float p = dot + xor32(dot);
        if (lane < 32) red[(tile * NW + wave) * 32 + lane] = p;
        dot = 0.0f;
    };
    auto finish = [&](long unit, int tile) {     // wave `tile`: sdf = b_8[0] + the waves' partial dot products
        if (wave == tile && lane < 32) {
            f32x16 b8[1];
            load_accvec<9, 8, 1>(blob, LY.L[8].bias, b8, lane);
            float s = b8[0][0];
#pragma unroll
            for (int k = 0; k < NW; ++k) s += red[(tile * NW + k) * 32 + lane];
            const long n = (unit * 2 + tile) * 32 + r;
            if (n < N) sdf_out[n] = s;
        }
    };
    f32x16 accA[TN], accB[TN], cw[TN];
    float dot = 0.0f;
    auto load_cw = [&]() { load_accvec<8, 0, TN>(blob, LY.extra, cw, lane, t0); };   // row 0 of W_8 in accumulator layout
    const __amdgpu_buffer_rsrc_t rsrc = p2_rsrc(blob);
    auto next_of = [&](int l) { return P2Next{LY.L[l].fwd_hi, LY.L[l].fwd_lo, LY.L[l].bias, l == 3 ? 7 : 8}; };
    P2Prime<FNEUS_P2_DEPTH, TN> pr;
    p2_prime_all<PREC, FNEUS_P2_DEPTH, TN>(pr, blob, rsrc, lane, t0, next_of(0));
    if ((long)blockIdx.x < units) encode(blockIdx.x);
    p2_barrier();
    bool first = true;
    for (long unit = blockIdx.x; unit < units; unit += gridDim.x) {
        asm volatile("" : "+s"(blob));
        if (!first) load_cw();
        if (first)
            p2h_pass<PREC, 3, 8, 1, 0>(blob, rsrc, LY.L[0].fwd_hi, LY.L[0].fwd_lo, pr, next_of(0), lds_, lane, t0, accA, 0, accB, 1, TN, cw, dot);
        else
            p2h_pass<PREC, 3, 8, 1, 2>(blob, rsrc, LY.L[0].fwd_hi, LY.L[0].fwd_lo, pr, next_of(0), lds_, lane, t0, accA, 0, accB, 1, TN, cw, dot);
        if (!first) put_dot(dot, 1);
        p2_barrier();
        if (!first) finish(unit - gridDim.x, 1);
        first = false;
        p2h_pass<PREC, 3, 8, 1, 1>(blob, rsrc, LY.L[0].fwd_hi, LY.L[0].fwd_lo, pr, next_of(1), lds_, lane, t0, accB, 1, accA, 0, TN, cw, dot);
        p2_barrier();
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            asm volatile("" : "+s"(blob));
            const int tn3 = 7 - t0 < TN ? 7 - t0 : TN;                   // layer 3 has 7 tiles: its last wave publishes one fewer
            const int tn_prev = l - 1 == 3 ? tn3 : TN;
            const int tn_this = l == 3 ? tn3 : TN;
            const P2Next same = next_of(l), following = next_of(l == 7 ? 0 : l + 1);
            if (l == 3)
                p2h_pass<PREC, 16, 7, 0, 1>(blob, rsrc, LY.L[3].fwd_hi, LY.L[3].fwd_lo, pr, same, lds_, lane, t0, accA, 0, accB, 1, tn_prev, cw, dot);
            else if (l == 4)
                p2h_pass<PREC, 17, 8, 2, 1>(blob, rsrc, LY.L[4].fwd_hi, LY.L[4].fwd_lo, pr, same, lds_, lane, t0, accA, 0, accB, 1, tn_prev, cw, dot);
            else
                p2h_pass<PREC, 16, 8, 0, 1>(blob, rsrc, LY.L[l].fwd_hi, LY.L[l].fwd_lo, pr, same, lds_, lane, t0, accA, 0, accB, 1, tn_prev, cw, dot);
            p2_barrier();
            if (l == 3)
                p2h_pass<PREC, 16, 7, 0, 1>(blob, rsrc, LY.L[3].fwd_hi, LY.L[3].fwd_lo, pr, following, lds_, lane, t0, accB, 1, accA, 0, tn_this, cw, dot);
            else if (l == 4)
                p2h_pass<PREC, 17, 8, 2, 1>(blob, rsrc, LY.L[4].fwd_hi, LY.L[4].fwd_lo, pr, following, lds_, lane, t0, accB, 1, accA, 0, tn_this, cw, dot);
            else if (l == 7) {
                load_cw();
                p2h_pass<PREC, 16, 8, 0, 2>(blob, rsrc, LY.L[7].fwd_hi, LY.L[7].fwd_lo, pr, following, lds_, lane, t0, accB, 1, accA, 0, tn_this, cw, dot);
            } else
                p2h_pass<PREC, 16, 8, 0, 1>(blob, rsrc, LY.L[l].fwd_hi, LY.L[l].fwd_lo, pr, following, lds_, lane, t0, accB, 1, accA, 0, tn_this, cw, dot);
            if (l == 7) put_dot(dot, 0);
            if (l == 5 && unit + gridDim.x < units) encode(unit + gridDim.x);     // slots 16..18 are free behind layer 4
            p2_barrier();
        }
        finish(unit, 0);
    }
    if (!first) {       // tail of the last unit: act 7 of tile 1 -> dot
        load_cw();
        p2h_dot_only<PREC>(accB, cw, dot);
        put_dot(dot, 1);
        p2_barrier();
        long last = blockIdx.x;
        while (last + gridDim.x < units) last += gridDim.x;
        finish(last, 1);
    }
}

template <int PREC>
static int launch_k1_p2h(const unsigned char* b, const PointSrc& src, long n_pts, float* sdf_out, hipStream_t stream) {
    static bool done = false;
    if (!done) {
        allow_big_lds(sdf_fwd_p2h_kernel<PREC>);
        done = true;
    }
    const long units = (n_pts + 63) / 64;
    hipLaunchKernelGGL((sdf_fwd_p2h_kernel<PREC>), dim3((unsigned)(units < 512 ? units : 512)), dim3(256),
                       kP2hLdsTotal + 2 * 4 * 32 * 4, stream, b, src, n_pts, sdf_out);
    return launch_status();
}

int sdf_fwd_p2h(const unsigned char* b, const PointSrc& src, long n_pts, float* sdf_out, int prec, hipStream_t stream) {
    if (prec == 3) return launch_k1_p2h<3>(b, src, n_pts, sdf_out, stream);
    if (prec == 1) return launch_k1_p2h<1>(b, src, n_pts, sdf_out, stream);
    return -2;
}

}  // namespace fneus
