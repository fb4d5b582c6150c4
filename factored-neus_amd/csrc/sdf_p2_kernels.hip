// SDF-network kernels in the two-pass pipelined form (p2_engine.h): the same maths, operands and per-accumulator summation
// order as sdf_kernels.hip (reference models/fields.py:74-111); the activation of one half of a workgroup's samples runs
// inside the MFMA stream of the other half.
//   K1 sdf_fwd_p2_kernel<PREC>: PE -> 8 x (dense + Softplus(beta = 100)) -> sdf row of the linear last layer (fields.py:93-95)
#include <stdlib.h>
#include "p2_engine.h"
#include "fneus_kernels.h"
#include "sdf_w8.h"

namespace fneus {

// One work unit = 128 samples (4 tiles); workgroup b takes units b, b + gridDim, ...  Pass schedule of a unit (A = set {0, 1},
// B = set {2, 3}; "|| x" = the vector work inside the pass):
//   L0.A || tail of the previous unit (act 7 B -> dot)      L0.B || act 0 A
//   Ll.A || act l-1 B                                       Ll.B || act l A                (l = 1..7; act 7 A -> dot)
// The encoding of the NEXT unit is written to slots 16..18 behind layer 4 (their last reader in this unit).
// Units to evaluate: all of them, or -- `sel` given -- the listed ones (fneus_sdf_fwd_rays: the 128-sample units of the rays that
// are marked; the list and its length lie in device memory, the other units' samples get `fill` from the workgroups up front).
struct UnitSel {
    const int32_t* list;     // nullptr: unit i is unit i
    const int32_t* n_list;
    const unsigned char* ray_mask;
    float fill;
};

template <int PREC, int TN>
__global__ void __launch_bounds__(512 / TN, 2 / TN) sdf_fwd_p2_kernel(const unsigned char* blob, PointSrc src, long N,
                                                            float* __restrict__ sdf_out, UnitSel sel) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    constexpr int NW = 8 / TN;                                                // waves: wave w owns output tiles TN w .. TN w + TN - 1
    float* red = reinterpret_cast<float*>(lds_ + kP2LdsTotal);               // [4 tiles][NW waves][32 samples]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int t0 = TN * wave, r = lane & 31, h = lane >> 5;
    constexpr auto& LY = kSdfLayout;
    const long all_units = (N + 127) / 128;
    const long units = sel.list ? (long)__builtin_amdgcn_readfirstlane(*sel.n_list) : all_units;      // positions in the list
    auto U = [&](long i) { return sel.list ? (long)__builtin_amdgcn_readfirstlane(sel.list[i]) : i; };    // position -> unit
    if (sel.list) {                         // the samples of the units that are not listed
        for (long u = blockIdx.x; u < all_units; u += gridDim.x)
            if (sel.ray_mask[(u * 128) / src.m] == 0 && threadIdx.x < 128 && u * 128 + threadIdx.x < N)
                sdf_out[u * 128 + threadIdx.x] = sel.fill;
    }
    auto encode = [&](long unit) {          // wave w < 4: encoding of tile w of the unit -> slots 16..18
        if (wave >= 4) return;
        const long n = (unit * 4 + wave) * 32 + r;
        const long nc = n < N ? n : N - 1;
        float x[3], pe[39], jc[39];
        load_point(src, nc, x);
        posenc<6, false>(x, pe, jc);
        BFrag<PREC> pf[kMaxKS];
        vec_to_bfrag<PREC, 39, 3, 0>(pe, pf, h);
        frags_to_lds<PREC, 3>(lds_ + wave * kP2Half, lane, 16, pf);
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
#ifdef FNEUS_P2_STAMPS
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) g_p2_last[threadIdx.x >> 8] = __builtin_amdgcn_s_memtime();
#endif
#ifdef FNEUS_P2_CLOCK                   // timing experiments only: shader cycles and 100 MHz ticks of every wave behind the outputs
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
    f32x16 accA[TN][2], accB[TN][2], cw[TN];
    float dot[2] = {0.0f, 0.0f};
    auto load_cw = [&]() { load_accvec<8, 0, TN>(blob, LY.extra, cw, lane, t0); };   // row 0 of W_8 in accumulator layout (the sdf
                                                                                    // row): fetched where it is used (32 registers)
    const __amdgpu_buffer_rsrc_t rsrc = p2_rsrc(blob);
    auto next_of = [&](int l) { return P2Next{LY.L[l].fwd_hi, LY.L[l].fwd_lo, LY.L[l].bias, l == 3 ? 7 : 8}; };
    P2Prime<FNEUS_P2_DEPTH, TN> pr;
    p2_prime_all<PREC, FNEUS_P2_DEPTH, TN>(pr, blob, rsrc, lane, t0, next_of(0));
    if ((long)blockIdx.x < units) encode(U(blockIdx.x));
    p2_barrier();
    bool first = true;
    long prev_unit = 0;
    for (long pos = blockIdx.x; pos < units; pos += gridDim.x) {
        const long unit = U(pos);
        asm volatile("" : "+s"(blob));
        // ---- layer 0 (3 k-steps on the encoding)
        if (!first) load_cw();
        if (first)
            p2_pass<PREC, 3, 8, 1, 0, TN>(blob, rsrc, LY.L[0].fwd_hi, LY.L[0].fwd_lo, pr, next_of(0), lds_, lane, t0, accA, 0, accB, 2, TN, cw, dot);
        else
            p2_pass<PREC, 3, 8, 1, 2, TN>(blob, rsrc, LY.L[0].fwd_hi, LY.L[0].fwd_lo, pr, next_of(0), lds_, lane, t0, accA, 0, accB, 2, TN, cw, dot);
        if (!first) put_dot(dot, 2);
        p2_barrier();
        if (!first) finish(prev_unit, 2);
        first = false;
        p2_pass<PREC, 3, 8, 1, 1, TN>(blob, rsrc, LY.L[0].fwd_hi, LY.L[0].fwd_lo, pr, next_of(1), lds_, lane, t0, accB, 2, accA, 0, TN, cw, dot);
        p2_barrier();
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            asm volatile("" : "+s"(blob));
            const int tn3 = 7 - t0 < TN ? 7 - t0 : TN;                   // layer 3 has 7 tiles: its last wave publishes one fewer
            const int tn_prev = l - 1 == 3 ? tn3 : TN;
            const int tn_this = l == 3 ? tn3 : TN;
            const P2Next same = next_of(l), following = next_of(l == 7 ? 0 : l + 1);
            // pass A: MFMAs of set {0, 1} || activation of layer l-1, set {2, 3}
            if (l == 3)
                p2_pass<PREC, 16, 7, 0, 1, TN>(blob, rsrc, LY.L[3].fwd_hi, LY.L[3].fwd_lo, pr, same, lds_, lane, t0, accA, 0, accB, 2, tn_prev, cw, dot);
            else if (l == 4)
                p2_pass<PREC, 17, 8, 2, 1, TN>(blob, rsrc, LY.L[4].fwd_hi, LY.L[4].fwd_lo, pr, same, lds_, lane, t0, accA, 0, accB, 2, tn_prev, cw, dot);
            else
                p2_pass<PREC, 16, 8, 0, 1, TN>(blob, rsrc, LY.L[l].fwd_hi, LY.L[l].fwd_lo, pr, same, lds_, lane, t0, accA, 0, accB, 2, tn_prev, cw, dot);
            p2_barrier();
            // pass B: MFMAs of set {2, 3} || activation of layer l, set {0, 1} (layer 7: -> dot product)
            if (l == 3)
                p2_pass<PREC, 16, 7, 0, 1, TN>(blob, rsrc, LY.L[3].fwd_hi, LY.L[3].fwd_lo, pr, following, lds_, lane, t0, accB, 2, accA, 0, tn_this, cw, dot);
            else if (l == 4)
                p2_pass<PREC, 17, 8, 2, 1, TN>(blob, rsrc, LY.L[4].fwd_hi, LY.L[4].fwd_lo, pr, following, lds_, lane, t0, accB, 2, accA, 0, tn_this, cw, dot);
            else if (l == 7) {
                load_cw();
                p2_pass<PREC, 16, 8, 0, 2, TN>(blob, rsrc, LY.L[7].fwd_hi, LY.L[7].fwd_lo, pr, following, lds_, lane, t0, accB, 2, accA, 0, tn_this, cw, dot);
            } else
                p2_pass<PREC, 16, 8, 0, 1, TN>(blob, rsrc, LY.L[l].fwd_hi, LY.L[l].fwd_lo, pr, following, lds_, lane, t0, accB, 2, accA, 0, tn_this, cw, dot);
            if (l == 7) put_dot(dot, 0);
            if (l == 5 && pos + gridDim.x < units) encode(U(pos + gridDim.x));    // slots 16..18 are free behind layer 4
            p2_barrier();
        }
        finish(unit, 0);
        prev_unit = unit;
    }
    if (!first) {       // tail of the last unit: act 7 of set {2, 3} -> dot
        load_cw();
        p2_valu_only<PREC, 2, TN>(lds_, lane, t0, accB, 2, TN, cw, dot);
        put_dot(dot, 2);
        p2_barrier();
        finish(prev_unit, 2);
    }
#ifdef FNEUS_P2_STAMPS
    P2_STAMP(3);
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) {
        const int w_ = threadIdx.x >> 8;
        printf("p2 K1 wave %d: entry %llu  k-steps %llu  tail %llu  outside %llu cycles\n", 4 * w_, g_p2_stamp[w_][0], g_p2_stamp[w_][1], g_p2_stamp[w_][2],
               g_p2_stamp[w_][3]);
        for (int k = 0; k < 4; ++k) g_p2_stamp[w_][k] = 0;
    }
#endif
#ifdef FNEUS_P2_CLOCK
    if (lane == 0) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(sdf_out + ((N + 3) & ~3L)) + (blockIdx.x * 4 + wave) * 2;
        st[0] = __builtin_amdgcn_s_memtime() - c0;
        st[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
#endif
}

template <int PREC, int TN>
static int launch_k1_p2(const unsigned char* b, const PointSrc& src, long n_pts, float* sdf_out, hipStream_t stream,
                        const UnitSel& sel = UnitSel{nullptr, nullptr, nullptr, 0.0f}) {
    static bool done = false;
    if (!done) {
        allow_big_lds(sdf_fwd_p2_kernel<PREC, TN>);
        done = true;
    }
    const long units = (n_pts + 127) / 128;
    hipLaunchKernelGGL((sdf_fwd_p2_kernel<PREC, TN>), dim3((unsigned)(units < 256 ? units : 256)), dim3(512 / TN),
                       kP2LdsTotal + 4 * 8 * 32 * 4, stream, b, src, n_pts, sdf_out, sel);
    return launch_status();
}

// the 128-sample units of the marked rays, in order: list [n] and its length (work[0]); one workgroup, 8 units per thread and round
__global__ void __launch_bounds__(1024) k1_unit_list_kernel(const unsigned char* __restrict__ ray_mask, int m, long all_units,
                                                            int32_t* __restrict__ work) {
    __shared__ int wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (long u0 = 0; u0 < all_units; u0 += 1024) {
        const long u = u0 + threadIdx.x;
        const bool alive = u < all_units && ray_mask[(u * 128) / m] != 0;
        const unsigned long long mk = __ballot(alive);
        if (lane == 0) wsum[wave] = __popcll(mk);
        __syncthreads();
        int before = base, total = 0;
        for (int w = 0; w < 16; ++w) {
            before += w < wave ? wsum[w] : 0;
            total += wsum[w];
        }
        if (alive) work[1 + before + __popcll(mk & ((1ull << lane) - 1ull))] = (int32_t)u;
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) work[0] = base;
}

void k1_unit_list(const unsigned char* ray_mask, int m, long all_units, int32_t* work, hipStream_t stream) {
    hipLaunchKernelGGL(k1_unit_list_kernel, dim3(1), dim3(1024), 0, stream, ray_mask, m, all_units, work);
}

// K1 on the marked rays only (fneus_sdf_fwd_rays); two-pass kernel on 8 waves
int sdf_fwd_p2_rays(const unsigned char* b, const PointSrc& src, long n_pts, const unsigned char* ray_mask, float fill, int32_t* work,
                    float* sdf_out, int prec, hipStream_t stream) {
    const long all_units = (n_pts + 127) / 128;
    hipLaunchKernelGGL(k1_unit_list_kernel, dim3(1), dim3(1024), 0, stream, ray_mask, src.m, all_units, work);
    const UnitSel sel{work + 1, work, ray_mask, fill};
    if (prec == 3) return launch_k1_p2<3, 1>(b, src, n_pts, sdf_out, stream, sel);
    if (prec == 1) return launch_k1_p2<1, 1>(b, src, n_pts, sdf_out, stream, sel);
    return -2;
}

// tn = 2: 4 waves (one per SIMD, 512 registers), tn = 1: 8 waves (two per SIMD: one wave's dependent vector instructions wait
// while the other issues)
int sdf_fwd_p2(const unsigned char* b, const PointSrc& src, long n_pts, float* sdf_out, int prec, int tn, hipStream_t stream) {
    if (prec == 3) return tn == 1 ? launch_k1_p2<3, 1>(b, src, n_pts, sdf_out, stream) : launch_k1_p2<3, 2>(b, src, n_pts, sdf_out, stream);
    if (prec == 1) return tn == 1 ? launch_k1_p2<1, 1>(b, src, n_pts, sdf_out, stream) : launch_k1_p2<1, 2>(b, src, n_pts, sdf_out, stream);
    return -2;
}

}  // namespace fneus
