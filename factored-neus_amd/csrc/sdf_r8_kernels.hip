// SDF-network kernels on resident-weight 8-wave workgroups (r8_engine.h): the same maths, operands, buffers and per-accumulator
// summation order as the 4-wave kernels of sdf_kernels.hip (reference models/fields.py:74-111 and its autograd double backward).
//   sdf_grad_rev_r8_kernel : the reverse sweep of K2 -- normal = d sdf / d x and the a_l planes -- on the sigma' blocks the forward
//                            launch (sdf_p2_train_kernels.hip) has written ("Reverse chain")
#include <stdlib.h>
#include "r8_engine.h"
#include "fneus_kernels.h"
#include "sdf_r8.h"

namespace fneus {

// ---- K2, reverse sweep ---------------------------------------------------------------------------------------------------------
// a_8 = e_0;  for l = 7..0:  a_l = s_l * g_hat(h_{l+1}),  g_hat(u_l) = W_l^T a_l;  normal = J^T (g_hat(u_0) + q_skip).
// Step l (7..1) of a group:  D_h: g_hat(h_l) = rev L[l] . a_l (region h)   P_h: a_{l-1} = s_{l-1} * g_hat(h_l) -> region h, plane.
// Reverse packs: L[4] has 9 row tiles (0..6: g_hat(h_4), 217 rows; 7, 8: q_skip, the PE part of the skip input) -- wave 7 computes
// both q_skip tiles (tile 7 on the resident path, tile 8 by a streamed pass over the same fragments) and parks them in the
// lane-private scratch st.qs; a_3 has 7 tiles, so wave 7 has no post phase there.  The 2 row tiles of L[0] (the 39 PE inputs) and
// the normal are wave hb's for half hb, as in sdf_fwd_grad_tph_kernel.
template <int PREC, bool TRAIN, int GP>
__global__ void __launch_bounds__(512, 1) sdf_grad_rev_r8_kernel(const unsigned char* blob, PointSrc src, long N, SdfStash st,
                                                                  float* __restrict__ normal_out, long grp_begin, long grp_end) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    constexpr int HALF = kR8Half;
    constexpr bool LO = TRAIN && PREC == 3 && GP == 3;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const PPLane pl = pp_lane(lane);
    constexpr auto& LY = kSdfLayout;
    const long tiles = pp_tiles(N);
    const long groups = grp_end >= 0 ? grp_end : (N + 63) / 64;
    const __amdgpu_buffer_rsrc_t rsrc = p2_rsrc(blob);
    const unsigned voff = (unsigned)(lane + w * 64) * 16u;
    auto rev_of = [&](int l) { return R8Layer{LY.L[l].rev_hi, LY.L[l].rev_lo, l == 4 ? 9 : 8}; };
    R8W W;
    for (long grp = grp_begin + blockIdx.x; grp < groups; grp += gridDim.x) {
        asm volatile("" : "+s"(blob));
        long tile[2], n[2], nc[2];
        bool valid[2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            tile[hb] = grp * 2 + hb;
            n[hb] = tile[hb] * 32 + r;
            valid[hb] = n[hb] < N;
            nc[hb] = valid[hb] ? n[hb] : N - 1;
        }
        auto a_blk = [&](unsigned char* base, int l, int hb) { return TRAIN && base ? base + ((size_t)l * tiles + tile[hb]) * kPPBlock : nullptr; };
        // sigma'(z_l) of this wave's tile: fragments 2 w, 2 w + 1 of the lane-private block of (tile, l)
        auto sig_load = [&](u16x8 (&sg)[2], int l, int hb) {
            const unsigned char* p = st.ps + ((size_t)tile[hb] * 8 + l) * kPPBlock + (size_t)(2 * w) * kFragBytes + lane * 16;
            sg[0] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(p));
            sg[1] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(p + kFragBytes));
        };
        auto apply = [&](f32x16& acc, const u16x8 (&sg)[2]) {       // a_l = s_l * g_hat(h_{l+1})
#pragma unroll
            for (int sh = 0; sh < 2; ++sh)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[8 * sh + e] *= (float)sg[sh][e] * (1.0f / 65535.0f);
        };
        u16x8 sg[2][2];
        sig_load(sg[0], 7, 0);
        sig_load(sg[1], 7, 1);
        r8_wload_all<PREC, 16>(W, rsrc, voff, rev_of(7), blob);
        f32x16 acc;
        {   // a_7 = s_7 * g_hat(h_8), g_hat(h_8) = row 0 of W_8 (the same for every sample)
            f32x16 g8[1];
            load_accvec<8, 0, 1>(blob, LY.extra, g8, lane, w);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                acc = g8[0];
                apply(acc, sg[hb]);
                r8_publish<PREC>(acc, lds_ + hb * HALF, lane, w, a_blk(st.a_hi, 7, hb), LO ? a_blk(st.a_lo, 7, hb) : nullptr, pl, valid[hb]);
                sig_load(sg[hb], 6, hb);
            }
        }
        p2_barrier();
        // one step: L = the layer whose reverse pack is multiplied, KS its k-steps; the post phase forms a_{L-1};
        // KSN = k-steps of the next step's pack (requested during D1), 0 = none
        auto step = [&](auto L_, auto KS_, auto KSN_) {
            constexpr int L = decltype(L_)::value, KS = decltype(KS_)::value, KSN = decltype(KSN_)::value;
            asm volatile("" : "+s"(blob));
            const R8Layer nx = rev_of(L > 1 ? L - 1 : 1);
            const bool has_post = !(L == 4 && w == 7);              // a_3 has 7 tiles
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                r8_zero(acc);
                if (hb == 0) r8_dense<PREC, KS, 0>(W, lds_ + lane * 16, acc, rsrc, voff, nx, blob);
                else r8_dense<PREC, KS, KSN>(W, lds_ + HALF + lane * 16, acc, rsrc, voff, nx, blob);
                if constexpr (L == 4) {
                    if (w == 7) {           // q_skip: tile 7 (resident path, above) and tile 8 (streamed) of this half -> scratch
                        f32x16 q8[1][1];
                        zero_acc(q8[0]);
                        dense_ldsb_h<PREC, 16, 9, 8, 1, 2, true, 1, HALF>(blob, kSdfLayout.L[4].rev_hi, kSdfLayout.L[4].rev_lo, lds_ + hb * HALF, q8, lane);
                        f32x16* q = st.qs + ((size_t)tile[hb] * 2) * 64 + lane;
                        q[0] = acc;
                        q[64] = q8[0][0];
                    }
                }
                p2_barrier();                                       // every wave has read region hb
                if (has_post) {
                    apply(acc, sg[hb]);
                    r8_publish<PREC>(acc, lds_ + hb * HALF, lane, w, a_blk(st.a_hi, L - 1, hb), LO ? a_blk(st.a_lo, L - 1, hb) : nullptr, pl,
                                     valid[hb]);
                }
                if (L >= 2 && !(L == 5 && w == 7)) sig_load(sg[hb], L - 2, hb);          // operands of the next step's post phase
            }
        };
        using std::integral_constant;
        step(integral_constant<int, 7>{}, integral_constant<int, 16>{}, integral_constant<int, 16>{});
        step(integral_constant<int, 6>{}, integral_constant<int, 16>{}, integral_constant<int, 16>{});
        step(integral_constant<int, 5>{}, integral_constant<int, 16>{}, integral_constant<int, 16>{});     // next: L[4], 9 tiles
        step(integral_constant<int, 4>{}, integral_constant<int, 16>{}, integral_constant<int, 14>{});     // next: L[3], 14 k-steps
        step(integral_constant<int, 3>{}, integral_constant<int, 14>{}, integral_constant<int, 16>{});
        step(integral_constant<int, 2>{}, integral_constant<int, 16>{}, integral_constant<int, 16>{});
        step(integral_constant<int, 1>{}, integral_constant<int, 16>{}, integral_constant<int, 0>{});
        p2_barrier();                                               // a_0 of both halves is in LDS
        // the 2 row tiles of the 39 PE inputs and normal = J^T q: wave hb for half hb
        if (w < 2) {
            const int hb = w;
            f32x16 qq[2][1];
            zero_acc(qq[0]);
            zero_acc(qq[1]);
            dense_ldsb_h<PREC, 16, 2, 0, 2, 2, true, 1, HALF>(blob, LY.L[0].rev_hi, LY.L[0].rev_lo, lds_ + hb * HALF, qq, lane);
            const f32x16* qsk = st.qs + ((size_t)tile[hb] * 2) * 64 + lane;
            f32x16 q[2];
            q[0] = qq[0][0] + __builtin_nontemporal_load(qsk);          // (written by wave 7: served from L2, not this CU's L1)
            q[1] = qq[1][0] + __builtin_nontemporal_load(qsk + 64);
            float x[3], pe[39], jc[39];       // Jacobian coefficients of the encoding, recomputed
            load_point(src, nc[hb], x);
            posenc<6, true>(x, pe, jc);
            float nrm[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float coef[39];
#pragma unroll
                for (int f = 0; f < 39; ++f) coef[f] = ((f % 3) == c) ? jc[f] : 0.0f;
                const float part = acc_dot_partial<2, 39>(q, coef, h);
                nrm[c] = part + xor32(part);
            }
            if (valid[hb] && lane < 32) {
#pragma unroll
                for (int c = 0; c < 3; ++c) normal_out[n[hb] * 3 + c] = nrm[c];
            }
        }
        p2_barrier();                                               // the group's fragments are consumed
    }
}

template <int PREC, bool TRAIN, int GP>
static int launch_rev_r8(const unsigned char* b, const PointSrc& src, long n_pts, const SdfStash& st, float* normal_out, long g_begin,
                         long g_end, hipStream_t stream) {
    static bool done = false;
    if (!done) {
        allow_big_lds(sdf_grad_rev_r8_kernel<PREC, TRAIN, GP>);
        done = true;
    }
    const long ng = g_end - g_begin;
    hipLaunchKernelGGL((sdf_grad_rev_r8_kernel<PREC, TRAIN, GP>), dim3((unsigned)(ng < 256 ? ng : 256)), dim3(512), kR8Lds, stream, b, src,
                       n_pts, st, normal_out, g_begin, g_end);
    return launch_status();
}

int sdf_grad_rev_r8(const unsigned char* b, const PointSrc& src, long n_pts, const SdfStash& st, float* normal_out, int prec, int train,
                    int gp, long g_begin, long g_end, hipStream_t stream) {
    if (g_end <= g_begin) return 0;
    if (prec == 3 && train && gp == 3) return launch_rev_r8<3, true, 3>(b, src, n_pts, st, normal_out, g_begin, g_end, stream);
    if (prec == 3 && train) return launch_rev_r8<3, true, 1>(b, src, n_pts, st, normal_out, g_begin, g_end, stream);
    if (prec == 3) return launch_rev_r8<3, false, 1>(b, src, n_pts, st, normal_out, g_begin, g_end, stream);
    if (prec == 1 && train) return launch_rev_r8<1, true, 1>(b, src, n_pts, st, normal_out, g_begin, g_end, stream);
    if (prec == 1) return launch_rev_r8<1, false, 1>(b, src, n_pts, st, normal_out, g_begin, g_end, stream);
    return -2;
}

}  // namespace fneus
