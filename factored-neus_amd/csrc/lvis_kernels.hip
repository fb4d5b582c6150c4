// Per-lobe light visibility of stage 3 (reference models/inverRender.py:128-192 get_diffuse_visibility) on the distilled
// Lvis network (models/fields.py:338-369: [PE10(point) | PE4(direction)] 90 -> 4 x (256 + ReLU) -> 1 -> sigmoid).
//
// For every surface point the reference evaluates Lvis at S = 32 directions around each of the M = 128 light lobes --
// 4096 evaluations per point, up to 2.1 M per training step, the hot op of stage 3 -- zeroes the directions that face away
// from the normal and averages each lobe's samples with the weights exp(lambda (d . axis - 1)).  Here one 32-sample tile
// IS one (point, lobe) pair: sample r of the tile is direction r of the lobe.  That makes
//   * the encoding of the point (60 sincos) a per-workgroup constant: computed once per point, kept in registers by wave 0;
//   * the back-face test a per-tile decision: a lobe whose 32 directions all face away costs nothing (about half of them);
//   * the weighted average a reduction over the 32 lanes of the output tile: the kernel writes vis[lobe][point] directly --
//     the 4096 network outputs per point never reach memory.
// Tensor-parallel workgroup of 4 waves per tile (tp_engine.h), two workgroups per CU, weights streamed from L2 (0.6 MB).
#define FNEUS_PREFETCH_X3 4
#define FNEUS_PREFETCH_X1 8
#include "pp_engine.h"
#include "fneus_kernels.h"

namespace fneus {

constexpr int kLvisLds = 16 * 2 * kFragBytes;      // B fragments of a 256-wide layer (hi, lo)

template <int PREC, int KS>
FN_DEV void lvis_write_frags(unsigned char* frag, int lane, int ks0, const BFrag<PREC> (&b)[kMaxKS], int src0) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        *reinterpret_cast<bf16x8*>(frag + ((ks0 + i) * NPL) * kFragBytes + lane * 16) = b[src0 + i].hi;
        if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(frag + ((ks0 + i) * NPL + 1) * kFragBytes + lane * 16) = b[src0 + i].lo;
    }
}

template <int PREC>
__global__ void __launch_bounds__(256, 2) lvis_visibility_tp_kernel(const unsigned char* blob, const float* __restrict__ points,
                                                                    const float* __restrict__ normals,
                                                                    const float* __restrict__ dirs /*[M][32][3]*/,
                                                                    const float* __restrict__ weights /*[M][32]*/,
                                                                    const unsigned char* __restrict__ point_mask, int n_pts,
                                                                    int n_lobes, float* __restrict__ vis /*[M][n_pts]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform for the compiler too
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    constexpr auto& LY = kLvisLayout;
    for (int pt = blockIdx.x; pt < n_pts; pt += gridDim.x) {
        if (point_mask && point_mask[pt] == 0) {          // a ray without a surface hit (fixed-shape step): nothing to evaluate
            for (int lobe = threadIdx.x; lobe < n_lobes; lobe += blockDim.x) vis[(size_t)lobe * n_pts + pt] = 0.0f;
            continue;
        }
        float x[3], nrm[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[c] = points[pt * 3 + c];
            nrm[c] = normals[pt * 3 + c];
        }
        BFrag<PREC> bpe[kMaxKS];          // k-steps 0..3: PE10 of the point (wave 0), 4..5: PE4 of the direction (wave 1)
        if (wave == 0) {
            float pe[63], jc[63];
            posenc<10, false>(x, pe, jc);
            vec_to_bfrag<PREC, 63, 4, 0>(pe, bpe, h);
        }
        for (int lobe = 0; lobe < n_lobes; ++lobe) {
            asm volatile("" : "+s"(blob));
            float d[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = dirs[((size_t)lobe * 32 + r) * 3 + c];
            const bool front = (nrm[0] * d[0] + nrm[1] * d[1] + nrm[2] * d[2]) > 1e-6f;      // inverRender.py:169
            // (readfirstlane: the compiler must see a wave-uniform branch, the tile loop launders a scalar pointer)
            const int any_front = __builtin_amdgcn_readfirstlane((int)(__ballot(front) != 0ull));
            if (!any_front) {             // the whole lobe faces away from this point: visibility 0 (the same in all 4 waves)
                if (threadIdx.x == 0) vis[(size_t)lobe * n_pts + pt] = 0.0f;
                continue;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the previous tile's fragments are consumed
            if (wave == 0) lvis_write_frags<PREC, 4>(frag, lane, 0, bpe, 0);
            if (wave == 1) {
                float pe[27], jc[27];
                posenc<4, false>(d, pe, jc);
                vec_to_bfrag<PREC, 27, 2, 4>(pe, bpe, h);
                lvis_write_frags<PREC, 2>(frag, lane, 4, bpe, 4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the 6 input k-steps are in LDS
            BFrag<PREC> bf[kMaxKS];
            tp_operands<PREC, 6>(frag, lane, bf);
            f32x16 acc[2];
            load_accvec<8, 0, 2>(blob, LY.L[0].bias, acc, lane, t0);
            tp_dense<PREC, 6, 8, 0, 2>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, frag, bf, acc, lane, t0);
#pragma unroll 1
            for (int l = 1; l <= 3; ++l) {      // (no per-layer laundering of `blob` here: together with the skip path above it
                                                // makes the backend place the pointer in a VGPR and fail)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t][e] = fmaxf(acc[t][e], 0.0f);
                tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, nullptr, nullptr, pl, true);
                tp_operands<PREC, 16>(frag, lane, bf);
                load_accvec<8, 0, 2>(blob, LY.L[l].bias, acc, lane, t0);
                tp_dense<PREC, 16, 8, 0, 2>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, frag, bf, acc, lane, t0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = fmaxf(acc[t][e], 0.0f);
            tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, nullptr, nullptr, pl, true);
            tp_operands<PREC, 16>(frag, lane, bf);
            if (wave == 0) {              // output layer: one tile, row 0 = register 0 of lane half 0
                f32x16 o[1];
                load_accvec<1, 0, 1>(blob, LY.L[4].bias, o, lane);
                tp_dense<PREC, 16, 1, 0, 1>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, frag, bf, o, lane);
                const float w = weights[(size_t)lobe * 32 + r];
                float num = (h == 0 && front) ? w / (1.0f + expf(-o[0][0])) : 0.0f;      // fields.py:358 sigmoid; :183 zero when back-facing
                float den = h == 0 ? w : 0.0f;
#pragma unroll
                for (int s = 16; s >= 1; s >>= 1) {
                    num += __shfl_xor(num, s, 64);
                    den += __shfl_xor(den, s, 64);
                }
                if (lane == 0) vis[(size_t)lobe * n_pts + pt] = num / (den + 1e-6f);    // inverRender.py:188
            }
        }
    }
}

}  // namespace fneus

using namespace fneus;

extern "C" int fneus_lvis_visibility(const void* lvis_blob, const float* points, const float* normals, const float* dirs,
                                     const float* weights, const unsigned char* point_mask, int n_pts, int n_lobes, int n_dirs,
                                     float* vis, int prec, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_pts <= 0 || n_lobes <= 0) return 0;
    if (!lvis_blob || !points || !normals || !dirs || !weights || !vis) {
        set_last_error("fneus_lvis_visibility: null argument");
        return -2;
    }
    if (n_dirs != 32) {
        set_last_error("fneus_lvis_visibility: 32 directions per lobe (one MFMA tile per (point, lobe) pair)");
        return -2;
    }
    const unsigned char* b = reinterpret_cast<const unsigned char*>(lvis_blob);
    const unsigned grid = (unsigned)(n_pts < 2048 ? n_pts : 2048);
    if (prec == 3) {
        static bool done = false;
        if (!done) { fneus::allow_big_lds(lvis_visibility_tp_kernel<3>); done = true; }
        hipLaunchKernelGGL(lvis_visibility_tp_kernel<3>, dim3(grid), dim3(256), fneus::kLvisLds, stream, b, points, normals, dirs,
                           weights, point_mask, n_pts, n_lobes, vis);
    } else if (prec == 1) {
        static bool done = false;
        if (!done) { fneus::allow_big_lds(lvis_visibility_tp_kernel<1>); done = true; }
        hipLaunchKernelGGL(lvis_visibility_tp_kernel<1>, dim3(grid), dim3(256), fneus::kLvisLds, stream, b, points, normals, dirs,
                           weights, point_mask, n_pts, n_lobes, vis);
    } else {
        return -2;
    }
    return fneus::launch_status();
}
