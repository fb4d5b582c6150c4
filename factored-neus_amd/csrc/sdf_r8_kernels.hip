// SDF-network kernels on resident-weight 8-wave workgroups (r8_engine.h): the same maths, operands, buffers and per-accumulator
// summation order as the 4-wave kernels of sdf_kernels.hip (reference models/fields.py:74-111 and its autograd double backward).
//   sdf_grad_rev_r8_kernel : the reverse sweep of K2 -- normal = d sdf / d x and the a_l planes -- on the sigma' blocks the forward
//                            launch (sdf_p2_train_kernels.hip) has written ("Reverse chain")
//   sdf_bwd_r8_kernel      : K3, both backward chains (Appendix A, "Backward")
// NH = 32-sample halves per workgroup (2 or 4): a layer's resident fragments serve all of them, so the L2 -> CU weight stream per
// sample -- the bound of these kernels at NH = 2 (FNEUS_R8_STAMPS: a CU takes in ~20 B / clk, a 64-sample step wants 256 KiB) --
// halves at NH = 4.  LDS: NH regions of 38 KiB (152 KiB at NH = 4).
// Every global access of the steps is a BUFFER access: a descriptor of the whole array with its exact size (scalar registers), ONE
// constant lane offset, the block / fragment offset as a scalar -- no 64-bit vector addresses (hipcc computes those ahead of the
// steps, spills them, and reloads each behind an `s_waitcnt vmcnt(0)` that drains the weight and operand requests), and a tile
// beyond the allocation (the last group of a ragged launch) reads zeros and stores nothing.  Offsets are 32-bit: the launchers
// send launches whose arrays exceed 2 GiB to the 4-wave kernels.
#include <stdlib.h>
#include "r8_engine.h"
#include "p2_train.h"
#include "fneus_kernels.h"
#include "sdf_r8.h"

namespace fneus {

// Without a barrier behind the post phase the older wave of a SIMD (w < 4) finishes its post phase first and starts the next
// dense phase beside its partner's vector work: the pairing that slows both -- measured with FNEUS_R8_STAMPS:
// post phase 730 cycles for waves 0-3, 2450 for waves 4-7, and waves 0-3 wait 2950 cycles per half step at the next barrier.
#ifndef FNEUS_R8_SYNC
#define FNEUS_R8_SYNC 1
#endif
#if FNEUS_R8_SYNC
#define R8_PHASE_SYNC() p2_barrier()
#else
#define R8_PHASE_SYNC() do { } while (0)
#endif
// FNEUS_R8_STAMPS (timing experiments only): cycles per phase kind, summed over a launch, printed by waves 0, 4, 7 of block 0
#ifdef FNEUS_R8_STAMPS
#define R8_STAMP(k)                                                              \
    do {                                                                         \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();              \
        stamp_sum[k] += t_ - stamp_last;                                         \
        stamp_last = t_;                                                         \
    } while (0)
#else
#define R8_STAMP(k) do { } while (0)
#endif

FN_DEV __amdgpu_buffer_rsrc_t r8_array(unsigned char* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, p ? (int)bytes : 0, 0x00020000);
}
FN_DEV p2_u32x4 r8_ldb(__amdgpu_buffer_rsrc_t rs, unsigned vo, uint32_t so) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, (int)so, 2);        // (nt: read once)
}

// this wave's tile half `sh` (8 values per lane) -> B fragment 2 w + sh of a half's LDS region (to_lds) and of a plane block in
// HBM (hi, lo planes; samples beyond N as zeros; a NULL plane has a zero-sized descriptor: nothing is stored)
template <int PREC, bool LO>
FN_DEV void r8_put_half(const float (&y)[8], int sh, int w, int lane, unsigned char* region, bool to_lds, __amdgpu_buffer_rsrc_t p_hi,
                        __amdgpu_buffer_rsrc_t p_lo, uint32_t p_off, const PPLane& pl, bool valid) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if constexpr (PREC == 3) {
            __bf16 a, b2;
            split_bf16(y[e], a, b2);
            hi[e] = a;
            lo[e] = b2;
        } else {
            hi[e] = (__bf16)y[e];
        }
    }
    const int ks = 2 * w + sh;
    if (to_lds) {
        *reinterpret_cast<bf16x8*>(region + (ks * NPL) * kFragBytes + lane * 16) = hi;
        if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(region + (ks * NPL + 1) * kFragBytes + lane * 16) = lo;
    }
    const unsigned vo = sh ? pl.odd : pl.even;                  // (ks & 1 == sh)
    p2_store128<true>(__builtin_bit_cast(p2_u32x4, valid ? hi : zero_bf16x8()), p_hi, vo, (int)(p_off + (uint32_t)ks * kFragBytes));
    if constexpr (LO) p2_store128<true>(__builtin_bit_cast(p2_u32x4, valid ? lo : zero_bf16x8()), p_lo, vo, (int)(p_off + (uint32_t)ks * kFragBytes));
}

// ---- K2, reverse sweep ---------------------------------------------------------------------------------------------------------
// a_8 = e_0;  for l = 7..0:  a_l = s_l * g_hat(h_{l+1}),  g_hat(u_l) = W_l^T a_l;  normal = J^T (g_hat(u_0) + q_skip).
// Step l (7..1) of a group:  D_h: g_hat(h_l) = rev L[l] . a_l (region h)   P_h: a_{l-1} = s_{l-1} * g_hat(h_l) -> region h, plane.
// Reverse packs: L[4] has 9 row tiles (0..6: g_hat(h_4), 217 rows; 7, 8: q_skip, the PE part of the skip input) -- wave 7 computes
// both q_skip tiles (tile 7 on the resident path, tile 8 by a streamed pass over the same fragments) and parks them in the
// lane-private scratch st.qs; a_3 has 7 tiles, so wave 7 has no post phase there.  The 2 row tiles of L[0] (the 39 PE inputs) and
// the normal are wave hb's for half hb, as in sdf_fwd_grad_tph_kernel.
// sigma' operands: two register sets; the set freed by the post phase of (step, half) is re-requested for the phase two further
// on -- the same half of the next step at NH = 2, half + 2 (or half - 2 of the next step) at NH = 4.
template <int PREC, bool TRAIN, int GP, int NH>
__global__ void __launch_bounds__(512, 1) sdf_grad_rev_r8_kernel(const unsigned char* blob, PointSrc src, long N, SdfStash st,
                                                                  float* __restrict__ normal_out, long grp_begin, long grp_end) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    constexpr int HALF = kR8Half;
    constexpr bool LO = TRAIN && PREC == 3 && GP == 3;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const PPLane pl = pp_lane(lane);
    const unsigned lane16 = (unsigned)lane * 16u;
    constexpr auto& LY = kSdfLayout;
    const long tiles = pp_tiles(N);
    const __amdgpu_buffer_rsrc_t rsrc = p2_rsrc(blob);
    const unsigned voff = (unsigned)(lane + w * 64) * 16u;
    const __amdgpu_buffer_rsrc_t rs_ps = r8_array(st.ps, (size_t)tiles * 8 * kPPBlock);
    const __amdgpu_buffer_rsrc_t rs_a_hi = r8_array(TRAIN ? st.a_hi : nullptr, (size_t)tiles * 8 * kPPBlock);
    const __amdgpu_buffer_rsrc_t rs_a_lo = r8_array(LO ? st.a_lo : nullptr, (size_t)tiles * 8 * kPPBlock);
    auto rev_of = [&](int l) { return R8Layer{kSdfLayout.L[l].rev_hi, kSdfLayout.L[l].rev_lo, l == 4 ? 9 : 8}; };
    R8W W;
    u16x8 sg[2][2];                     // two operand sets: phase p (= step * NH + half) uses set p & 1
#ifdef FNEUS_R8_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime(), stamp_d_prev = 0;     // [0]: outside the steps too; [4 + hb]: dense of half hb
    const unsigned long long stamp_t0 = stamp_last;
#endif
    for (long grp = grp_begin + blockIdx.x; grp < grp_end; grp += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const long tile0 = grp * NH;
        auto valid_of = [&](int hb) { return (tile0 + hb) * 32 + r < N; };
        // [slot][tile] planes: a tile beyond the allocation would alias tile 0.. of the next slot -- its offset is put beyond every array
        auto blk_off = [&](int slot, int hb) { return tile0 + hb < tiles ? (uint32_t)(((size_t)slot * tiles + tile0 + hb) * kPPBlock) : 0x7ff00000u; };
        // sigma'(z_l) of this wave's tile: fragments 2 w, 2 w + 1 of the lane-private block of (tile, l), [tile][8] blocks
        auto sig_load_at = [&](u16x8 (&sgs)[2], int l, long tile) {
            const uint32_t so = (uint32_t)(((size_t)tile * 8 + l) * kPPBlock) + (uint32_t)(2 * w) * kFragBytes;
            sgs[0] = __builtin_bit_cast(u16x8, r8_ldb(rs_ps, lane16, so));
            sgs[1] = __builtin_bit_cast(u16x8, r8_ldb(rs_ps, lane16, so + kFragBytes));
        };
        auto sig_load = [&](u16x8 (&sgs)[2], int l, int hb) { sig_load_at(sgs, l, tile0 + hb); };
        // a_l = s_l * g_hat(h_{l+1}) -> region hb, plane slot l
        auto post = [&](const f32x16& acc, const u16x8 (&sg)[2], int l, int hb) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = acc[8 * sh + e] * ((float)sg[sh][e] * (1.0f / 65535.0f));
                r8_put_half<PREC, LO>(y, sh, w, lane, lds_ + hb * HALF, true, rs_a_hi, rs_a_lo, blk_off(l, hb), pl, valid_of(hb));
            }
        };
        // (NOT requested behind the last phases of the group before, as K3 does: the streamed passes of this kernel's epilogue --
        // the PE tiles of L[0] -- would wait for their fragments behind those requests; measured 219 against 210 us)
        sig_load(sg[0], 7, 0);
        sig_load(sg[1], 7, 1);
        r8_wload_all<PREC, 16>(W, rsrc, voff, rev_of(7), blob);
        f32x16 acc;
        {   // a_7 = s_7 * g_hat(h_8), g_hat(h_8) = row 0 of W_8 (the same for every sample)
            f32x16 g8[1];
            load_accvec<8, 0, 1>(blob, LY.extra, g8, lane, w);
            static_for<0, NH>([&](auto HB_) {
                constexpr int hb = decltype(HB_)::value;
                post(g8[0], sg[hb & 1], 7, hb);
                if constexpr (hb + 2 < NH) sig_load(sg[hb & 1], 7, hb + 2);
                else sig_load(sg[hb & 1], 6, hb + 2 - NH);
            });
        }
        p2_barrier();
        // one step: L = the layer whose reverse pack is multiplied, KS its k-steps; the post phase forms a_{L-1};
        // KSN = k-steps of the next step's pack (requested during the last half's dense phase), 0 = none
        auto step = [&](auto L_, auto KS_, auto KSN_) {
            constexpr int L = decltype(L_)::value, KS = decltype(KS_)::value, KSN = decltype(KSN_)::value;
            asm volatile("" : "+s"(blob));
            const R8Layer nx = rev_of(L > 1 ? L - 1 : 1);
            const bool has_post = !(L == 4 && w == 7);              // a_3 has 7 tiles
            static_for<0, NH>([&](auto HB_) {
                constexpr int hb = decltype(HB_)::value;
                R8_STAMP(0);
                r8_zero(acc);
                r8_dense<PREC, KS, (hb == NH - 1 ? KSN : 0)>(W, lds_ + hb * HALF + lane * 16, acc, rsrc, voff, nx, blob);
                if constexpr (L == 4) {
                    if (w == 7) {           // q_skip: tile 7 (resident path, above) and tile 8 (streamed) of this half -> scratch
                        f32x16 q8[1][1];
                        zero_acc(q8[0]);
                        dense_ldsb_h<PREC, 16, 9, 8, 1, 2, true, 1, HALF>(blob, kSdfLayout.L[4].rev_hi, kSdfLayout.L[4].rev_lo, lds_ + hb * HALF, q8, lane);
                        if (tile0 + hb < tiles) {
                            f32x16* q = st.qs + ((size_t)(tile0 + hb) * 2) * 64 + lane;
                            q[0] = acc;
                            q[64] = q8[0][0];
                        }
                    }
                }
                R8_STAMP(1);
#ifdef FNEUS_R8_STAMPS
                stamp_sum[4 + hb] += stamp_sum[1] - stamp_d_prev;
                stamp_d_prev = stamp_sum[1];
#endif
                p2_barrier();                                       // every wave has read region hb
                R8_STAMP(2);
                if constexpr (hb == NH - 1) r8_request_rest<PREC, KSN>(W, rsrc, voff, nx, blob);
                if (has_post) post(acc, sg[hb & 1], L - 1, hb);
                // the freed operand set: sigma' of the phase two further on (wave 7 has no tile in layer 3: blocks allocated, unused)
                if constexpr (hb + 2 < NH) sig_load(sg[hb & 1], L - 1, hb + 2);
                else if constexpr (L >= 2) sig_load(sg[hb & 1], L - 2, hb + 2 - NH);
                R8_PHASE_SYNC();
                R8_STAMP(3);
            });
        };
        using std::integral_constant;
#define IC(v) integral_constant<int, v>{}
        step(IC(7), IC(16), IC(16));
        step(IC(6), IC(16), IC(16));
        step(IC(5), IC(16), IC(16));     // next: L[4], 9 tiles
        step(IC(4), IC(16), IC(14));     // next: L[3], 14 k-steps
        step(IC(3), IC(14), IC(16));
        step(IC(2), IC(16), IC(16));
        step(IC(1), IC(16), IC(0));
#undef IC
        if (!FNEUS_R8_SYNC) p2_barrier();                           // a_0 of every half is in LDS
        // the 2 row tiles of the 39 PE inputs and normal = J^T q: wave hb for half hb
        if (w < NH) {
            const int hb = w;
            const long n = (tile0 + hb) * 32 + r;
            const bool valid = n < N;
            const long nc = valid ? n : N - 1;
            f32x16 qq[2][1];
            zero_acc(qq[0]);
            zero_acc(qq[1]);
            dense_ldsb_h<PREC, 16, 2, 0, 2, 2, true, 1, HALF>(blob, LY.L[0].rev_hi, LY.L[0].rev_lo, lds_ + hb * HALF, qq, lane);
            if (tile0 + hb < tiles) {
                const f32x16* qsk = st.qs + ((size_t)(tile0 + hb) * 2) * 64 + lane;
                f32x16 q[2];
                q[0] = qq[0][0] + __builtin_nontemporal_load(qsk);          // (written by wave 7: served from L2, not this CU's L1)
                q[1] = qq[1][0] + __builtin_nontemporal_load(qsk + 64);
                float x[3], pe[39], jc[39];       // Jacobian coefficients of the encoding, recomputed
                load_point(src, nc, x);
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
                if (valid && lane < 32) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) normal_out[n * 3 + c] = nrm[c];
                }
            }
        }
        p2_barrier();                                               // the group's fragments are consumed
    }
#ifdef FNEUS_R8_STAMPS
    if (blockIdx.x == 0 && lane == 0 && (w == 0 || w == 4 || w == 7))
        printf("rev_r8 NH %d wave %d: total %llu cycles; dense %llu (by half: %llu %llu %llu %llu), barrier wait %llu, post %llu, rest (post -> next dense, group edges) %llu\n", NH, w,
               __builtin_amdgcn_s_memtime() - stamp_t0, stamp_sum[1], stamp_sum[4], stamp_sum[5], stamp_sum[6], stamp_sum[7], stamp_sum[2], stamp_sum[3], stamp_sum[0]);
#endif
}

template <int PREC, bool TRAIN, int GP, int NH>
static int launch_rev_r8(const unsigned char* b, const PointSrc& src, long n_pts, const SdfStash& st, float* normal_out, long g_begin,
                         long g_end, hipStream_t stream) {
    static bool done = false;
    if (!done) {
        allow_big_lds(sdf_grad_rev_r8_kernel<PREC, TRAIN, GP, NH>);
        done = true;
    }
    const long ng = g_end - g_begin;
    hipLaunchKernelGGL((sdf_grad_rev_r8_kernel<PREC, TRAIN, GP, NH>), dim3((unsigned)(ng < 256 ? ng : 256)), dim3(512), NH * kR8Half, stream, b,
                       src, n_pts, st, normal_out, g_begin, g_end);
    return launch_status();
}

// halves per workgroup: 4 (128 samples share a pass over the weights) once that still fills the chip; FNEUS_R8_NH=2 | 4 forces
// (read at every call so that tests can switch it)
static int r8_halves(long groups64) {
    const char* e = getenv("FNEUS_R8_NH");
    const int f = e ? atoi(e) : 0;
    if (f == 2 || f == 4) return f;
    return groups64 >= 2 * 256 ? 4 : 2;
}

int sdf_grad_rev_r8(const unsigned char* b, const PointSrc& src, long n_pts, const SdfStash& st, float* normal_out, int prec, int train,
                    int gp, long g_begin, long g_end, hipStream_t stream) {
    if (g_end <= g_begin) return 0;
    // (g_begin, g_end: 64-sample groups; 128-sample workgroups take pairs of them)
    const bool nh4 = r8_halves(g_end - g_begin) == 4 && (g_begin & 1) == 0;
    const long gb = nh4 ? g_begin / 2 : g_begin, ge = nh4 ? (g_end + 1) / 2 : g_end;
#define FNEUS_REV_R8(P, T, G)                                                                                         \
    return nh4 ? launch_rev_r8<P, T, G, 4>(b, src, n_pts, st, normal_out, gb, ge, stream)                            \
               : launch_rev_r8<P, T, G, 2>(b, src, n_pts, st, normal_out, gb, ge, stream)
    if (prec == 3 && train && gp == 3) FNEUS_REV_R8(3, true, 3);
    if (prec == 3 && train) FNEUS_REV_R8(3, true, 1);
    if (prec == 3) FNEUS_REV_R8(3, false, 1);
    if (prec == 1 && train) FNEUS_REV_R8(1, true, 1);
    if (prec == 1) FNEUS_REV_R8(1, false, 1);
#undef FNEUS_REV_R8
    return -2;
}

// ---- K3 ------------------------------------------------------------------------------------------------------------------------
// Backward of (sdf, feature, normal) w.r.t. the SDF-network weights (sdf_bwd_tph_kernel in sdf_kernels.hip):
//   ascending  (tangent of the reverse sweep): adj_0 = J nbar;  abar_l = W_l adj_l;  adj_{l+1} = s_l * abar_l;
//               coupling c_l = beta (1 - s_l) a_l abar_l
//   descending (ordinary backprop):            zbar_8 = [fbar ; sbar];  ubar_l = W_l^T zbar_l;  zbar_{l-1} = s_{l-1} * ubar_l + c_{l-1}
// Sixteen layer steps per group on the resident-weight form: F0..F7 (forward packs), then R8..R1 (reverse packs); the post phase of
// a step takes sigma'(z_l) and a second lane-private operand (a_l on the way up, c_l on the way down) from one of two register
// sets; the set a post phase frees is re-requested for the phase two further on.  At NH = 2 the coupling term of the top layer
// never leaves the wave: the post phases of F7 leave s_7 and c_7 in the two sets for the post phases of R8.
// Tiles: F3 has 7 output tiles and R4 is restricted to the 7 tiles of h_4 (the 39 skip inputs have no trainable ancestor): wave 7
// multiplies a spare tile there (one code path for all waves -- a branch around the dense phase makes hipcc spill the resident
// fragments at the join) and skips the post phase.  F4 reads the parked copy of qbar in slots 16..18 (LMAP 2).  The sdf tile of
// zbar_8 (k-steps 16, 17 of R8) is two streamed k-steps in front of the 16 resident ones (18 resident stages do not fit 256
// registers).
struct R8Ops {                  // operands of one post phase: fragments 2 w, 2 w + 1 of this wave's tile
    u16x8 sg[2];
    bf16x8 hi[2], lo[2];
};

template <int PREC, int GP, int NH, int XP>
__global__ void __launch_bounds__(512, 1) sdf_bwd_r8_kernel(const unsigned char* blob, PointSrc src, long N, SdfStash st, SdfBwdBufs bb,
                                                             const float* __restrict__ d_sdf, const float* __restrict__ d_feat,
                                                             const float* __restrict__ d_normal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    constexpr int HALF = XP == 3 ? kR8Half : kR8Half / 2;         // bf16 fragments: [k-step] x 1 KiB
    constexpr bool LO = PREC == 3 && GP == 3;          // lo planes exist (exact-gradient mode)
    static_assert(XP == PREC || (PREC == 3 && XP == 1 && !LO), "XP 1: the chains' activations are the bf16 values of their planes");
    constexpr bool KEEP = NH == 2;                     // c_7 stays in the operand registers across the turn
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const PPLane pl = pp_lane(lane);
    const long tiles = pp_tiles(N);
    const long groups = (N + 32 * NH - 1) / (32 * NH);
    const __amdgpu_buffer_rsrc_t rsrc = p2_rsrc(blob);
    const unsigned voff = (unsigned)(lane + w * 64) * 16u;
    const unsigned voff7 = (unsigned)(lane + (w < 7 ? w : 6) * 64) * 16u;        // packs with 7 tiles: wave 7 re-reads tile 6 (unused)
    const unsigned lane16 = (unsigned)lane * 16u;
    auto fwd_of = [&](int l) { return R8Layer{kSdfLayout.L[l].fwd_hi, kSdfLayout.L[l].fwd_lo, l == 3 ? 7 : 8}; };
    auto rev_of = [&](int l) { return R8Layer{kSdfLayout.L[l].rev_hi, kSdfLayout.L[l].rev_lo, l == 4 ? 9 : 8}; };
    const size_t sz8 = (size_t)tiles * 8 * kPPBlock, sz9 = (size_t)tiles * 9 * kPPBlock;
    const __amdgpu_buffer_rsrc_t rs_ps = r8_array(st.ps, sz8), rs_a_hi = r8_array(st.a_hi, sz8), rs_a_lo = r8_array(LO ? st.a_lo : nullptr, sz8);
    const __amdgpu_buffer_rsrc_t rs_c_hi = r8_array(bb.c_hi, sz8), rs_c_lo = r8_array(LO ? bb.c_lo : nullptr, sz8);
    const __amdgpu_buffer_rsrc_t rs_adj_hi = r8_array(bb.adj_hi, sz8), rs_adj_lo = r8_array(LO ? bb.adj_lo : nullptr, sz8);
    const __amdgpu_buffer_rsrc_t rs_zbar_hi = r8_array(bb.zbar_hi, sz9), rs_zbar_lo = r8_array(LO ? bb.zbar_lo : nullptr, sz9);
    R8W W;
    R8Ops ops[2];                       // phase p (= step * NH + half) uses set p & 1
    bool first = true;
#ifdef FNEUS_R8_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();     // [0]: outside the steps; [4..7]: descending
    const unsigned long long stamp_t0 = stamp_last;
#endif
    for (long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const long tile0 = grp * NH;
        auto valid_of = [&](int hb) { return (tile0 + hb) * 32 + r < N; };
        // [slot][tile] planes: a tile beyond the allocation would alias tile 0.. of the next slot -- its offset is put beyond every array
        auto blk_off = [&](int slot, int hb) { return tile0 + hb < tiles ? (uint32_t)(((size_t)slot * tiles + tile0 + hb) * kPPBlock) : 0x7ff00000u; };
        auto priv_off = [&](int l, int hb) { return (uint32_t)(((size_t)(tile0 + hb) * 8 + l) * kPPBlock); };            // [tile][8] lane-private
        // ---- operand requests: sigma'(z_l) and a_l (plane, slot-permuted) / c_l (lane-linear) of this wave's tile
        auto load_sig = [&](R8Ops& o, int l, int hb) {
            const uint32_t so = priv_off(l, hb) + (uint32_t)(2 * w) * kFragBytes;
            o.sg[0] = __builtin_bit_cast(u16x8, r8_ldb(rs_ps, lane16, so));
            o.sg[1] = __builtin_bit_cast(u16x8, r8_ldb(rs_ps, lane16, so + kFragBytes));
        };
        auto load_asc_at = [&](R8Ops& o, int l, long tile) {      // (any tile: the first operands of the NEXT group too)
            const uint32_t sp = (uint32_t)(((size_t)tile * 8 + l) * kPPBlock) + (uint32_t)(2 * w) * kFragBytes;
            o.sg[0] = __builtin_bit_cast(u16x8, r8_ldb(rs_ps, lane16, sp));
            o.sg[1] = __builtin_bit_cast(u16x8, r8_ldb(rs_ps, lane16, sp + kFragBytes));
            const uint32_t so = (tile < tiles ? (uint32_t)(((size_t)l * tiles + tile) * kPPBlock) : 0x7ff00000u) + (uint32_t)(2 * w) * kFragBytes;
            o.hi[0] = __builtin_bit_cast(bf16x8, r8_ldb(rs_a_hi, pl.even, so));
            o.hi[1] = __builtin_bit_cast(bf16x8, r8_ldb(rs_a_hi, pl.odd, so + kFragBytes));
            if constexpr (LO) {
                o.lo[0] = __builtin_bit_cast(bf16x8, r8_ldb(rs_a_lo, pl.even, so));
                o.lo[1] = __builtin_bit_cast(bf16x8, r8_ldb(rs_a_lo, pl.odd, so + kFragBytes));
            }
        };
        auto load_asc = [&](R8Ops& o, int l, int hb) {
            load_sig(o, l, hb);
            const uint32_t so = blk_off(l, hb) + (uint32_t)(2 * w) * kFragBytes;
            o.hi[0] = __builtin_bit_cast(bf16x8, r8_ldb(rs_a_hi, pl.even, so));
            o.hi[1] = __builtin_bit_cast(bf16x8, r8_ldb(rs_a_hi, pl.odd, so + kFragBytes));
            if constexpr (LO) {
                o.lo[0] = __builtin_bit_cast(bf16x8, r8_ldb(rs_a_lo, pl.even, so));
                o.lo[1] = __builtin_bit_cast(bf16x8, r8_ldb(rs_a_lo, pl.odd, so + kFragBytes));
            }
        };
        auto load_desc = [&](R8Ops& o, int l, int hb) {
            load_sig(o, l, hb);
            const uint32_t so = priv_off(l, hb) + (uint32_t)(2 * w) * kFragBytes;
            o.hi[0] = __builtin_bit_cast(bf16x8, r8_ldb(rs_c_hi, lane16, so));
            o.hi[1] = __builtin_bit_cast(bf16x8, r8_ldb(rs_c_hi, lane16, so + kFragBytes));
            if constexpr (LO) {
                o.lo[0] = __builtin_bit_cast(bf16x8, r8_ldb(rs_c_lo, lane16, so));
                o.lo[1] = __builtin_bit_cast(bf16x8, r8_ldb(rs_c_lo, lane16, so + kFragBytes));
            }
        };
        // One fragment half (sh) at a time, fenced: the post phase runs with the resident weights (136 registers) and both operand
        // sets live -- converting all 16 values of a tile at once does not fit beside them.
        // ascending: adj_{l+1} = s_l * abar_l -> region, plane;  c_l = beta (1 - s_l) a_l abar_l -> the scratch (keep: -> o.hi / o.lo)
        auto asc_post = [&](const f32x16& acc, R8Ops& o, int l, int hb, bool keep, bool to_lds) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                bf16x8 chi, clo;
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sv = (float)o.sg[sh][e] * (1.0f / 65535.0f);
                    float av = (float)o.hi[sh][e];
                    if constexpr (LO) av += (float)o.lo[sh][e];
                    const float abar = acc[8 * sh + e];
                    const float c = kBeta * (1.0f - sv) * av * abar;          // softplus'' * g_hat * abar  (a = s * g_hat)
                    y[e] = sv * abar;
                    if constexpr (XP == 3) {
                        __bf16 x, yy;
                        split_bf16(c, x, yy);
                        chi[e] = x;
                        clo[e] = yy;
                    } else {
                        chi[e] = (__bf16)c;
                    }
                }
                if (keep) {
                    o.hi[sh] = chi;
                    if constexpr (LO) o.lo[sh] = clo;
                } else {
                    const uint32_t so = priv_off(l, hb) + (uint32_t)(2 * w + sh) * kFragBytes;
                    p2_store128<true>(__builtin_bit_cast(p2_u32x4, chi), rs_c_hi, lane16, (int)so);
                    if constexpr (LO) p2_store128<true>(__builtin_bit_cast(p2_u32x4, clo), rs_c_lo, lane16, (int)so);
                }
                r8_put_half<XP, LO>(y, sh, w, lane, lds_ + hb * HALF, to_lds, rs_adj_hi, rs_adj_lo, blk_off(l, hb), pl, valid_of(hb));
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // descending: zbar_l = s_l * hbar_{l+1} + c_l -> region, plane (slot l)
        auto desc_post = [&](const f32x16& acc, const R8Ops& o, int l, int hb, bool to_lds) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sv = (float)o.sg[sh][e] * (1.0f / 65535.0f);
                    float c = (float)o.hi[sh][e];
                    if constexpr (LO) c += (float)o.lo[sh][e];
                    y[e] = sv * acc[8 * sh + e] + c;
                }
                r8_put_half<XP, LO>(y, sh, w, lane, lds_ + hb * HALF, to_lds, rs_zbar_hi, rs_zbar_lo, blk_off(l, hb), pl, valid_of(hb));
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (first) {                    // (later groups: requested at the end of the group before)
            load_asc(ops[0], 0, 0);
            load_asc(ops[1], 0, 1);
            r8_wload_all<PREC, 3>(W, rsrc, voff, fwd_of(0), blob);
            first = false;
        }
        // ---- qbar = J nbar (wave hb for half hb): k-steps 0..2 of F0, the qbar plane, and a copy parked in slots 16..18 (F4)
        if (w < NH) {
            const int hb = w;
            const long n = (tile0 + hb) * 32 + r;
            const bool valid = n < N;
            const long nc = valid ? n : N - 1;
            float x[3], nb[3], pe[39], jc[39], qb[39];
            load_point(src, nc, x);
#pragma unroll
            for (int c = 0; c < 3; ++c) nb[c] = valid ? d_normal[nc * 3 + c] : 0.0f;
            posenc<6, true>(x, pe, jc);
#pragma unroll
            for (int f = 0; f < 39; ++f) qb[f] = jc[f] * nb[f % 3];
            BFrag<XP> qf[kMaxKS];
            vec_to_bfrag<XP, 39, 3, 0>(qb, qf, h);
            frags_to_lds<XP, 3>(lds_ + hb * HALF, lane, 0, qf);
            frags_to_lds<XP, 3>(lds_ + hb * HALF, lane, 16, qf);
            if (tile0 + hb < tiles)
                frags_to_plane<XP, 3>(qf, 0, bb.qbar_hi + (size_t)(tile0 + hb) * 4 * kFragBytes,
                                        LO ? bb.qbar_lo + (size_t)(tile0 + hb) * 4 * kFragBytes : nullptr, pl, valid);
        }
        p2_barrier();
        f32x16 acc;
        // ---- ascending steps: L = layer (forward pack), KS its k-steps, KSN the next pack's (requested during the last half's dense)
        auto asc_step = [&](auto L_, auto KS_, auto KSN_, auto LMAP_) {
            constexpr int L = decltype(L_)::value, KS = decltype(KS_)::value, KSN = decltype(KSN_)::value, LMAP = decltype(LMAP_)::value;
            asm volatile("" : "+s"(blob));
            const R8Layer nx = L < 7 ? fwd_of(L + 1) : rev_of(8);
            const unsigned vo_next = (L + 1 == 3) ? voff7 : voff;
            const bool active = !(L == 3 && w == 7);                // F3 has 7 output tiles
            static_for<0, NH>([&](auto HB_) {
                constexpr int hb = decltype(HB_)::value;
                R8_STAMP(0);
                r8_zero(acc);
                r8_dense<PREC, KS, (hb == NH - 1 ? KSN : 0), LMAP, XP>(W, lds_ + hb * HALF + lane * 16, acc, rsrc, vo_next, nx, blob);
                R8_STAMP(1);
                p2_barrier();                                       // every wave has read region hb
                R8_STAMP(2);
                if constexpr (hb == NH - 1) r8_request_rest<PREC, KSN>(W, rsrc, vo_next, nx, blob);
                constexpr bool keep = KEEP && L == 7;
                if (active) asc_post(acc, ops[hb & 1], L, hb, keep, L < 7);     // adj_8: plane only
                // the freed operand set -> the phase two further on (at the turn: s_7, c_7 of the first descending phases)
                if constexpr (hb + 2 < NH) load_asc(ops[hb & 1], L, hb + 2);
                else if constexpr (L < 7) load_asc(ops[hb & 1], L + 1, hb + 2 - NH);
                else if constexpr (!keep) load_desc(ops[hb & 1], 7, hb + 2 - NH);
                R8_PHASE_SYNC();
                R8_STAMP(3);
            });
        };
        using std::integral_constant;
#define IC(v) integral_constant<int, v>{}
        asc_step(IC(0), IC(3), IC(16), IC(0));
        asc_step(IC(1), IC(16), IC(16), IC(0));
        asc_step(IC(2), IC(16), IC(16), IC(0));          // next: F3, 7 tiles
        asc_step(IC(3), IC(16), IC(17), IC(0));          // next: F4, 17 k-steps
        asc_step(IC(4), IC(17), IC(16), IC(2));
        asc_step(IC(5), IC(16), IC(16), IC(0));
        asc_step(IC(6), IC(16), IC(16), IC(0));
        asc_step(IC(7), IC(16), IC(16), IC(0));          // next: R8 (its first 16 k-steps)
        // ---- seed of the descending chain: zbar_8 = [fbar (tiles 0..7) ; sbar (row 0 of tile 8: k-steps 16, 17)]
        if (XP == 1 && d_feat == nullptr) {
            // the feature rows of the seed are in the plane already (slot 8 of zbar: the colour backward's bf16 fragments, heads added
            // by fneus_surface_scatter_plane): this wave's fragments 2 w, 2 w + 1 of every half -> its region, nothing to store
            static_for<0, NH>([&](auto HB_) {
                constexpr int hb = decltype(HB_)::value;
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    const int ks = 2 * w + sh;
                    const p2_u32x4 v = r8_ldb(rs_zbar_hi, sh ? pl.odd : pl.even, blk_off(8, hb) + (uint32_t)ks * kFragBytes);
                    *reinterpret_cast<p2_u32x4*>(lds_ + hb * HALF + ks * kFragBytes + lane * 16) = v;
                }
            });
        } else
        static_for<0, NH>([&](auto HB_) {
            constexpr int hb = decltype(HB_)::value;
            const long n = (tile0 + hb) * 32 + r;
            const bool valid = n < N;
            const long nc = valid ? n : N - 1;
            f32x16 t2[1];
            load_f32<1>(t2, d_feat + 32 * w, 256, nc, h);
            if (!valid) zero_acc(t2);
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = t2[0][8 * sh + e];
                r8_put_half<XP, LO>(y, sh, w, lane, lds_ + hb * HALF, true, rs_zbar_hi, rs_zbar_lo, blk_off(8, hb), pl, valid);
            }
        });
        if (w < NH) {
            const int hb = w;
            const long n = (tile0 + hb) * 32 + r;
            const bool valid = n < N;
            BFrag<XP> sf[kMaxKS];
            const float sv = (h == 0 && valid) ? d_sdf[n] : 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                sf[i].hi = zero_bf16x8();
                if constexpr (XP == 3) sf[i].lo = zero_bf16x8();
            }
            if constexpr (XP == 3) {
                __bf16 shi, slo;
                split_bf16(sv, shi, slo);
                sf[0].hi[0] = shi;
                sf[0].lo[0] = slo;
            } else {
                sf[0].hi[0] = (__bf16)sv;
            }
            frags_to_lds<XP, 2>(lds_ + hb * HALF, lane, 16, sf);
            if (tile0 + hb < tiles)
                frags_to_plane<XP, 2>(sf, 0, bb.zsdf_hi + (size_t)(tile0 + hb) * 2 * kFragBytes,
                                        LO ? bb.zsdf_lo + (size_t)(tile0 + hb) * 2 * kFragBytes : nullptr, pl, valid);
        }
        p2_barrier();
        // ---- descending steps: L = layer (reverse pack); the post phase forms zbar_{L-1}
        auto desc_step = [&](auto L_, auto KS_, auto KSN_) {
            constexpr int L = decltype(L_)::value, KS = decltype(KS_)::value, KSN = decltype(KSN_)::value;
            asm volatile("" : "+s"(blob));
            const R8Layer nx = L > 1 ? rev_of(L - 1) : fwd_of(0);
            const bool active = !(L == 4 && w == 7);                // R4: the 7 tiles of h_4 (wave 7: a q_skip tile, unused); zbar_3 has 7
            static_for<0, NH>([&](auto HB_) {
                constexpr int hb = decltype(HB_)::value;
                R8_STAMP(0);
                r8_zero(acc);
                if constexpr (L == 8 && XP == PREC) {     // the sdf tile of zbar_8: k-steps 16, 17 of R8, streamed
                    f32x16(&a1)[1][1] = reinterpret_cast<f32x16(&)[1][1]>(acc);
                    dense_ldsb_h<PREC, 2, 8, 0, 1, 2, true, 1, HALF>(blob, kSdfLayout.L[8].rev_hi + 16 * 8 * kFragBytes,
                                                                     kSdfLayout.L[8].rev_lo + 16 * 8 * kFragBytes,
                                                                     lds_ + hb * HALF + 16 * (PREC == 3 ? 2 : 1) * kFragBytes, a1, lane, w);
                }
                if constexpr (L == 8 && XP != PREC) {     // the same two k-steps on bf16 B fragments (slots 16, 17: hi only), W hi + lo
                    const unsigned char* fl16 = lds_ + hb * HALF + lane * 16 + 16 * kFragBytes;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const uint32_t f = (uint32_t)((16 + s) * 8 * 64) * 16u;
                        const bf16x8 whi = p2_wload(rsrc, voff, kSdfLayout.L[8].rev_hi + f, blob);
                        const bf16x8 wlo = p2_wload(rsrc, voff, kSdfLayout.L[8].rev_lo + f, blob);
                        const bf16x8 b = *reinterpret_cast<const bf16x8*>(fl16 + s * kFragBytes);
                        acc = mfma32(wlo, b, acc);
                        acc = mfma32(whi, b, acc);
                    }
                    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // (r8_dense: B registers may be reloaded at once)
                }
                r8_dense<PREC, KS, (hb == NH - 1 ? KSN : 0), 0, XP>(W, lds_ + hb * HALF + lane * 16, acc, rsrc, voff, nx, blob);
                R8_STAMP(5);
                p2_barrier();
                R8_STAMP(6);
                if constexpr (hb == NH - 1) r8_request_rest<PREC, KSN>(W, rsrc, voff, nx, blob);
                if (active) desc_post(acc, ops[hb & 1], L - 1, hb, L > 1);          // zbar_0: plane only
                if constexpr (hb + 2 < NH) load_desc(ops[hb & 1], L - 1, hb + 2);
                else if constexpr (L >= 2) load_desc(ops[hb & 1], L - 2, hb + 2 - NH);
                else load_asc_at(ops[hb & 1], 0, (grp + gridDim.x) * NH + hb + 2 - NH);      // the next group's first phases (beyond the
                R8_PHASE_SYNC();                                                             // last group: zeros, unused)
                R8_STAMP(7);
            });
        };
        desc_step(IC(8), IC(16), IC(16));                 // (+ the sdf tile's 2 k-steps, streamed)
        desc_step(IC(7), IC(16), IC(16));
        desc_step(IC(6), IC(16), IC(16));
        desc_step(IC(5), IC(16), IC(16));                 // next: R4, 9 tiles (0..6 used)
        desc_step(IC(4), IC(16), IC(14));                 // next: R3, 14 k-steps
        desc_step(IC(3), IC(14), IC(16));
        desc_step(IC(2), IC(16), IC(16));
        desc_step(IC(1), IC(16), IC(3));                  // next: F0 of the following group
#undef IC
        if (!FNEUS_R8_SYNC) p2_barrier();                           // the group's fragments are consumed
    }
#ifdef FNEUS_R8_STAMPS
    if (blockIdx.x == 0 && lane == 0 && (w == 0 || w == 4 || w == 7))
        printf("bwd_r8 NH %d wave %d: total %llu cycles; ascending: dense %llu, barrier wait %llu, post %llu; descending: dense %llu, barrier wait %llu, "
               "post %llu; rest (post -> next dense, qbar, seed, group edges) %llu\n", NH, w, __builtin_amdgcn_s_memtime() - stamp_t0, stamp_sum[1],
               stamp_sum[2], stamp_sum[3], stamp_sum[5], stamp_sum[6], stamp_sum[7], stamp_sum[0]);
#endif
}

template <int PREC, int GP, int NH, int XP>
static int launch_bwd_r8(const unsigned char* b, const PointSrc& src, long n_pts, const SdfStash& st, const SdfBwdBufs& bb, const float* d_sdf,
                         const float* d_feat, const float* d_normal, hipStream_t stream) {
    static bool done = false;
    if (!done) {
        allow_big_lds(sdf_bwd_r8_kernel<PREC, GP, NH, XP>);
        done = true;
    }
    const long groups = (n_pts + 32 * NH - 1) / (32 * NH);
    hipLaunchKernelGGL((sdf_bwd_r8_kernel<PREC, GP, NH, XP>), dim3((unsigned)(groups < 256 ? groups : 256)), dim3(512), NH * (XP == 3 ? kR8Half : kR8Half / 2), stream, b, src,
                       n_pts, st, bb, d_sdf, d_feat, d_normal);
    return launch_status();
}

int sdf_bwd_r8(const unsigned char* b, const PointSrc& src, long n_pts, const SdfStash& st, const SdfBwdBufs& bb, const float* d_sdf,
               const float* d_feat, const float* d_normal, int prec, int gp, hipStream_t stream) {
    const bool nh4 = r8_halves((n_pts + 63) / 64) == 4;
    // 256 samples per workgroup (NH = 8) where the bf16 regions (19 KiB per half) allow it and 128-sample groups would need a second
    // round on some CUs anyway: 65 536 points = one group per CU, every wave busy in the group's prologue (FNEUS_R8_NH=2 | 4 | 8 forces)
    const char* nhe = getenv("FNEUS_R8_NH");
    const int nhf = nhe ? atoi(nhe) : 0;
    const bool nh8 = nhf == 8 || (nhf != 2 && nhf != 4 && (n_pts + 127) / 128 > 256);
    // bf16 gradient planes (gradient precision 1 / 2): the chains' activations are the bf16 values of those planes -- two MFMAs per
    // product (W hi + lo), hi-only B fragments in LDS.  FNEUS_BWD_XHI=0: hi + lo activations inside the chains as before round 6.
    const char* xe = getenv("FNEUS_BWD_XHI");
    const bool xhi = xe ? atoi(xe) != 0 : true;
    if (d_feat == nullptr && !(prec == 3 && gp != 3 && xhi)) return -2;          // the seed from the plane: the bf16-cotangent form only
#define FNEUS_BWD_R8(P, G, X)                                                                                         \
    return nh4 ? launch_bwd_r8<P, G, 4, X>(b, src, n_pts, st, bb, d_sdf, d_feat, d_normal, stream)                   \
               : launch_bwd_r8<P, G, 2, X>(b, src, n_pts, st, bb, d_sdf, d_feat, d_normal, stream)
    if (prec == 3 && gp == 3) FNEUS_BWD_R8(3, 3, 3);
    if (prec == 3 && xhi && nh8) return launch_bwd_r8<3, 1, 8, 1>(b, src, n_pts, st, bb, d_sdf, d_feat, d_normal, stream);
    if (prec == 3 && xhi) FNEUS_BWD_R8(3, 1, 1);
    if (prec == 3) FNEUS_BWD_R8(3, 1, 3);
    if (prec == 1) FNEUS_BWD_R8(1, 1, 1);
#undef FNEUS_BWD_R8
    return -2;
}

}  // namespace fneus
