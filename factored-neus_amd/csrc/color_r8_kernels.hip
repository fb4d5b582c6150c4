// Backward of the colour network (autograd of RenderingNetwork.forward, reference models/fields.py:150-175) on resident-weight
// 8-wave workgroups (r8_engine.h): the maths, operands, planes and per-accumulator summation order of color_bwd_tph_kernel
// (color_kernels.hip), which stays for small launches and for arrays beyond the 32-bit buffer offsets.
//   zbar_4 = d rgb * rgb (1 - rgb)                       (3 rows of one tile: k-steps 0, 1 of the half's region; the zout planes)
//   R4: ubar_3 = W_4^T zbar_4  (2 k-steps, 8 tiles)      zbar_3 = relu mask_3 * ubar_3  -> region, zbar plane slot 3
//   R3 .. R1: ubar_{l-1} = W_l^T zbar_l (16 k-steps)     zbar_{l-1} = mask_{l-1} * ubar_{l-1} -> region, plane slot l - 1
//   R0: 10 row tiles -- wave w's tile w of the 8 feature tiles -> d_feat rows (fp32); the 2 side tiles by wave hb for half hb
//       (streamed) -> d_normal (columns 30..32 of the side inputs: pts and PE4(view) carry no gradient)
// A group = NH 32-sample halves; wave w owns output tile w of every step, its fragments of a step stay in registers for all halves
// and the next step's are requested during the last half's MFMAs (r8_dense).  The ReLU masks (16 bits per lane, tile and layer: the
// word the forward kernels wrote) are requested two phases ahead in two registers.
#include <stdlib.h>
#include "r8_engine.h"
#include "p2_train.h"
#include "fneus_kernels.h"
#include "color_r8.h"

namespace fneus {

#ifdef FNEUS_C8_STAMPS              // timing experiments only: cycles per phase kind of the pipelined form, block 0
#define C8_STAMP(k)                                                              \
    do {                                                                         \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();              \
        stamp_sum[k] += t_ - stamp_last;                                         \
        stamp_last = t_;                                                         \
    } while (0)
#else
#define C8_STAMP(k) do { } while (0)
#endif
#ifndef FNEUS_COLB_PIPE
#define FNEUS_COLB_PIPE 1           // bf16 cotangents: the post phase of a half inside the dense phase of the next one
#endif

// value of feature IDX (compile-time) of a two-tile accumulator-layout vector, valid in every lane of the sample pair
template <int IDX>
FN_DEV float acc_extract2(const f32x16 (&acc)[2], int h) {
    constexpr int t = IDX / 32, row = IDX % 32;
    constexpr int hh = (row >> 2) & 1;
    constexpr int reg = (row & 3) + 4 * (row >> 3);
    static_assert(acc_row(reg, hh) == row, "accumulator row mapping");
    const float v = (h == hh) ? acc[t][reg] : 0.0f;
    return v + xor32(v);
}

FN_DEV __amdgpu_buffer_rsrc_t c8_array(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}

// this wave's tile half `sh` (8 values per lane) -> B fragment 2 w + sh of a half's LDS region and of a plane block (zeros for
// samples beyond N; a NULL plane has a zero-sized descriptor: nothing is stored)
// XP: the region's fragments (3: hi + lo, 1: bf16 values); LO: the plane has a lo part (whatever the region holds)
template <int XP, bool LO>
FN_DEV void c8_put_half(const float (&y)[8], int sh, int w, int lane, unsigned char* region, bool to_lds, __amdgpu_buffer_rsrc_t p_hi,
                        __amdgpu_buffer_rsrc_t p_lo, uint32_t p_off, const PPLane& pl, bool valid) {
    constexpr int NPL = XP == 3 ? 2 : 1;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if constexpr (XP == 3 || LO) {
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
        if constexpr (XP == 3) *reinterpret_cast<bf16x8*>(region + (ks * NPL + 1) * kFragBytes + lane * 16) = lo;
    }
    const unsigned vo = sh ? pl.odd : pl.even;                  // (ks & 1 == sh)
    p2_store128<true>(__builtin_bit_cast(p2_u32x4, valid ? hi : zero_bf16x8()), p_hi, vo, (int)(p_off + (uint32_t)ks * kFragBytes));
    if constexpr (LO) p2_store128<true>(__builtin_bit_cast(p2_u32x4, valid ? lo : zero_bf16x8()), p_lo, vo, (int)(p_off + (uint32_t)ks * kFragBytes));
}

// GP 3: hi + lo planes everywhere (exact gradients); GP 1: hi planes (and zout_lo where the stash has one: gradient precision 2)
// XP 1 (with PREC 3, GP 1): the chain's activations are the bf16 values of the zbar planes (r8_dense) -- two MFMAs per product
template <int PREC, int GP, int NH, int XP>
__global__ void __launch_bounds__(512, 1) color_bwd_r8_kernel(const unsigned char* blob, long N, const float* __restrict__ d_rgb,
                                                               const float* __restrict__ rgb, ColStash st, float* __restrict__ d_feat,
                                                               float* __restrict__ d_normal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    constexpr int HALF = XP == 3 ? kR8Half : kR8Half / 2;         // bf16 fragments: [k-step] x 1 KiB
    constexpr bool LO = PREC == 3 && GP == 3;
    static_assert(XP == PREC || (PREC == 3 && XP == 1 && !LO), "XP 1: bf16 activations with bf16 planes only");
    constexpr bool PIPE = XP == 1 && PREC == 3 && FNEUS_COLB_PIPE;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const PPLane pl = pp_lane(lane);
    constexpr auto& LY = kColLayout;
    const long tiles = pp_tiles(N);
    const long groups = (N + 32 * NH - 1) / (32 * NH);
    const __amdgpu_buffer_rsrc_t rsrc = p2_rsrc(blob);
    const unsigned voff = (unsigned)(lane + w * 64) * 16u;
    const size_t sz4 = (size_t)tiles * 4 * kPPBlock;
    const __amdgpu_buffer_rsrc_t rs_z_hi = c8_array(st.zbar_hi, sz4), rs_z_lo = c8_array(LO ? st.zbar_lo : nullptr, sz4);
    const __amdgpu_buffer_rsrc_t rs_o_hi = c8_array(st.zout_hi, (size_t)tiles * 2 * kFragBytes);
    const __amdgpu_buffer_rsrc_t rs_o_lo = c8_array(PREC == 3 ? st.zout_lo : nullptr, (size_t)tiles * 2 * kFragBytes);
    const __amdgpu_buffer_rsrc_t rs_mask = c8_array(st.mask, (size_t)tiles * 4 * 1024);
    const __amdgpu_buffer_rsrc_t rs_df = c8_array(XP == 1 ? st.dfeat_hi : nullptr, (size_t)tiles * kPPBlock);      // d_feat as bf16 fragments
    const bool df_plane = XP == 1 && st.dfeat_hi != nullptr;
    const unsigned voff_mask = (unsigned)lane * 16u + (unsigned)(w >> 1) * 4u;
    const int mshift = (w & 1) * 16;
    auto rev_of = [&](int l) { return R8Layer{LY.L[l].rev_hi, LY.L[l].rev_lo, l == 0 ? 10 : 8}; };
    R8W W;
    uint32_t mk[2];                     // two operand sets: phase p (= step * NH + half) uses set p & 1
#ifdef FNEUS_C8_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
    const unsigned long long stamp_t0 = stamp_last;
#endif
    for (long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        asm volatile("" : "+s"(blob));
        const long tile0 = grp * NH;
        auto valid_of = [&](int hb) { return (tile0 + hb) * 32 + r < N; };
        // [slot][tile] planes: a tile beyond the allocation would alias tile 0.. of the next slot -- its offset is put beyond every array
        auto blk_off = [&](int slot, int hb) { return tile0 + hb < tiles ? (uint32_t)(((size_t)slot * tiles + tile0 + hb) * kPPBlock) : 0x7ff00000u; };
        auto mask_load = [&](uint32_t& m, int l, int hb) {          // (a tile beyond the allocation reads zeros)
            m = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_mask, (int)voff_mask, (int)(uint32_t)(((size_t)(tile0 + hb) * 4 + l) * 1024), 0);
        };
        // zbar_{l} = mask_l * ubar_l -> region hb, plane slot l
        auto post = [&](const f32x16& acc, uint32_t m, int l, int hb) {
            const uint32_t mm = m >> mshift;
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = ((mm >> (8 * sh + e)) & 1u) ? acc[8 * sh + e] : 0.0f;
                c8_put_half<XP, LO>(y, sh, w, lane, lds_ + hb * HALF, true, rs_z_hi, rs_z_lo, blk_off(l, hb), pl, valid_of(hb));
            }
        };
        mask_load(mk[0], 3, 0);
        mask_load(mk[1], 3, 1);
        r8_wload_all<PREC, 2>(W, rsrc, voff, rev_of(4), blob);
        // ---- zbar_4 = d rgb * sigmoid' (rows 0..2 of one tile): wave hb publishes k-steps 0, 1 of half hb and stores the zout planes
        if (w < NH) {
            const int hb = w;
            const long n = (tile0 + hb) * 32 + r;
            const bool valid = n < N;
            const long nc = valid ? n : N - 1;
            float y0[8], y1[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y0[e] = y1[e] = 0.0f;
            if (h == 0 && valid) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float y = rgb[nc * 3 + c];
                    y0[c] = d_rgb[nc * 3 + c] * y * (1.0f - y);
                }
            }
            const uint32_t zo = tile0 + hb < tiles ? (uint32_t)((size_t)(tile0 + hb) * 2 * kFragBytes) : 0x7ff00000u;
            c8_put_half<XP, PREC == 3>(y0, 0, 0, lane, lds_ + hb * HALF, true, rs_o_hi, rs_o_lo, zo, pl, valid);
            c8_put_half<XP, PREC == 3>(y1, 1, 0, lane, lds_ + hb * HALF, true, rs_o_hi, rs_o_lo, zo, pl, valid);
        }
        p2_barrier();
        f32x16 acc;
        // R0's output of half hb: this wave's feature tile -> d_feat rows, or -> fragments 2 w, 2 w + 1 of the tile's block in the
        // plane K3 reads as the seed zbar_8 (st.dfeat_hi)
        auto r0_out = [&](const f32x16& a16, int hb) {
            const long n = (tile0 + hb) * 32 + r;
            if (df_plane) {
                const uint32_t bo = tile0 + hb < tiles ? (uint32_t)((size_t)(tile0 + hb) * kPPBlock) : 0x7ff00000u;
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    bf16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (__bf16)(n < N ? a16[8 * sh + e] : 0.0f);
                    p2_store128<true>(__builtin_bit_cast(p2_u32x4, v), rs_df, sh ? pl.odd : pl.even, (int)(bo + (uint32_t)(2 * w + sh) * kFragBytes));
                }
            } else {
                const f32x16 one[1] = {a16};
                store_f32<1>(one, d_feat + 32 * w, 256, n, h, n < N);
            }
        };
        if constexpr (PIPE) {
            // ---- pipelined form (bf16 cotangents): the post phase of a half runs INSIDE the dense phase of the next one, on the other
            // of two accumulators -- four slices (select + convert of a fragment; its LDS and plane store; twice) in front of the first
            // k-steps' MFMAs.  Phase P = step * NH + half (steps R4, R3, R2, R1); post(P) forms zbar_{3 - P / NH} of half P % NH with
            // the mask in mk[P & 1]; ONE barrier per phase: behind it every wave has read region P % NH (dense P) and written region
            // (P - 1) % NH (post P - 1), which the next layer reads NH - 1 phases later at the earliest.
            f32x16 acc2[2];
            bf16x8 frag;
            auto post_slice = [&](auto PP_, auto M_) {
                constexpr int PP = decltype(PP_)::value, M = decltype(M_)::value;
                constexpr int l = 3 - PP / NH, hb = PP % NH, sh = M >> 1;
                if constexpr ((M & 1) == 0) {
                    const uint32_t mm = mk[PP & 1] >> mshift;
#pragma unroll
                    for (int e = 0; e < 8; ++e) frag[e] = (__bf16)(((mm >> (8 * sh + e)) & 1u) ? acc2[PP & 1][8 * sh + e] : 0.0f);
                    asm volatile("" : "+v"(frag));
                } else {
                    const int ks = 2 * w + sh;
                    *reinterpret_cast<bf16x8*>(lds_ + hb * HALF + ks * kFragBytes + lane * 16) = frag;
                    p2_store128<true>(__builtin_bit_cast(p2_u32x4, valid_of(hb) ? frag : zero_bf16x8()), rs_z_hi, sh ? pl.odd : pl.even,
                                      (int)(blk_off(l, hb) + (uint32_t)ks * kFragBytes));
                }
            };
            // the slices of post(PP) inside a dense phase of KS k-steps: one per k-step from the second on (two per k-step where KS < 5)
            auto side_of = [&](auto PP_, auto KS_) {
                return [&](auto S_) {
                    constexpr int PP = decltype(PP_)::value, KS = decltype(KS_)::value, sidx = decltype(S_)::value;
                    if constexpr (PP >= 0) {
                        static_for<0, 4>([&](auto M_) {
                            constexpr int M = decltype(M_)::value;
                            constexpr int slot = KS >= 5 ? M + 1 : (M * KS) / 4;
                            if constexpr (slot == sidx) post_slice(PP_, M_);
                        });
                    }
                };
            };
            auto pstep = [&](auto IDX_, auto KS_, auto KSN_) {
                constexpr int IDX = decltype(IDX_)::value, KS = decltype(KS_)::value, KSN = decltype(KSN_)::value;
                constexpr int L = 4 - IDX;
                asm volatile("" : "+s"(blob));
                const R8Layer nx = rev_of(L - 1);
                static_for<0, NH>([&](auto HB_) {
                    constexpr int hb = decltype(HB_)::value, P = IDX * NH + hb;
                    C8_STAMP(0);
                    r8_zero(acc2[P & 1]);
                    r8_dense<PREC, KS, (hb == NH - 1 ? KSN : 0), 0, XP>(W, lds_ + hb * HALF + lane * 16, acc2[P & 1], rsrc, voff, nx, blob,
                                                                           side_of(std::integral_constant<int, P - 1>{}, KS_));
                    C8_STAMP(KS == 16 ? (hb == NH - 1 ? 2 : 1) : 3);
                    if constexpr (hb == NH - 1) r8_request_rest<PREC, KSN>(W, rsrc, voff, nx, blob);
                    // the mask set post(P - 1) has consumed: the mask of phase P + 1
                    if constexpr (P >= 1 && P + 1 < 4 * NH) mask_load(mk[(P + 1) & 1], 3 - (P + 1) / NH, (P + 1) % NH);
                    p2_barrier();
                    C8_STAMP(4);
                });
            };
            using std::integral_constant;
#define IC(v) integral_constant<int, v>{}
            pstep(IC(0), IC(2), IC(16));
            pstep(IC(1), IC(16), IC(16));
            pstep(IC(2), IC(16), IC(16));
            pstep(IC(3), IC(16), IC(16));     // next: R0, 10 tiles
            // ---- R0: the 8 feature tiles -> d_feat; half 0 carries the last post phase (zbar_0 of half NH - 1), a barrier publishes it
            {
                asm volatile("" : "+s"(blob));
                const R8Layer nx = rev_of(4);
                static_for<0, NH>([&](auto HB_) {
                    constexpr int hb = decltype(HB_)::value;
                    r8_zero(acc);
                    if constexpr (hb == 0) {
                        r8_dense<PREC, 16, 0, 0, XP>(W, lds_ + lane * 16, acc, rsrc, voff, nx, blob, side_of(IC(4 * NH - 1), IC(16)));
                    } else {
                        r8_dense<PREC, 16, (hb == NH - 1 ? 2 : 0), 0, XP>(W, lds_ + hb * HALF + lane * 16, acc, rsrc, voff, nx, blob);
                    }
                    C8_STAMP(5);
                    r0_out(acc, hb);
                    if constexpr (hb == 0) p2_barrier();
                    C8_STAMP(6);
                });
            }
#undef IC
        } else {
        // one step: L = the layer whose reverse pack is multiplied, KS its k-steps; the post phase forms zbar_{L-1};
        // KSN = k-steps of the next step's pack (requested during the last half's dense phase)
        auto step = [&](auto L_, auto KS_, auto KSN_) {
            constexpr int L = decltype(L_)::value, KS = decltype(KS_)::value, KSN = decltype(KSN_)::value;
            asm volatile("" : "+s"(blob));
            const R8Layer nx = rev_of(L - 1);
            static_for<0, NH>([&](auto HB_) {
                constexpr int hb = decltype(HB_)::value;
                r8_zero(acc);
                r8_dense<PREC, KS, (hb == NH - 1 ? KSN : 0), 0, XP>(W, lds_ + hb * HALF + lane * 16, acc, rsrc, voff, nx, blob);
                p2_barrier();                                       // every wave has read region hb
                if constexpr (hb == NH - 1) r8_request_rest<PREC, KSN>(W, rsrc, voff, nx, blob);
                post(acc, mk[hb & 1], L - 1, hb);
                // the freed operand set: the mask of the phase two further on
                if constexpr (hb + 2 < NH) mask_load(mk[hb & 1], L - 1, hb + 2);
                else if constexpr (L >= 2) mask_load(mk[hb & 1], L - 2, hb + 2 - NH);
                p2_barrier();
            });
        };
        using std::integral_constant;
#define IC(v) integral_constant<int, v>{}
        step(IC(4), IC(2), IC(16));
        step(IC(3), IC(16), IC(16));
        step(IC(2), IC(16), IC(16));
        step(IC(1), IC(16), IC(16));      // next: R0, 10 tiles
#undef IC
        // ---- R0: the 8 feature tiles (wave w: tile w) -> d_feat rows; no post phase, the regions stay for the side tiles below
        {
            asm volatile("" : "+s"(blob));
            const R8Layer nx = rev_of(4);                           // the next group's first step
            static_for<0, NH>([&](auto HB_) {
                constexpr int hb = decltype(HB_)::value;
                r8_zero(acc);
                r8_dense<PREC, 16, (hb == NH - 1 ? 2 : 0), 0, XP>(W, lds_ + hb * HALF + lane * 16, acc, rsrc, voff, nx, blob);
                r0_out(acc, hb);
            });
        }
        }   // (!PIPE)
        // ---- the 2 side tiles of half hb (streamed, wave hb) -> d normal
        if (w < NH) {
            const int hb = w;
            const long n = (tile0 + hb) * 32 + r;
            const bool valid = n < N;
            f32x16 s2[2][1];
            zero_acc(s2[0]);
            zero_acc(s2[1]);
            dense_ldsb_h<PREC, 16, 10, 8, 2, 2, true, 1, HALF, XP>(blob, LY.L[0].rev_hi, LY.L[0].rev_lo, lds_ + hb * HALF, s2, lane);
            const f32x16 both[2] = {s2[0][0], s2[1][0]};
            const float g0 = acc_extract2<30>(both, h), g1 = acc_extract2<31>(both, h), g2 = acc_extract2<32>(both, h);
            if (valid && lane < 32) {
                if (st.dnormal_add) {           // the compositing backward's gradient of the same normals is in the buffer: autograd's sum
                    d_normal[n * 3 + 0] += g0;
                    d_normal[n * 3 + 1] += g1;
                    d_normal[n * 3 + 2] += g2;
                } else {
                    d_normal[n * 3 + 0] = g0;
                    d_normal[n * 3 + 1] = g1;
                    d_normal[n * 3 + 2] = g2;
                }
            }
        }
        C8_STAMP(7);
        p2_barrier();                                               // the group's fragments are consumed
    }
#ifdef FNEUS_C8_STAMPS
    if (blockIdx.x == 0 && lane == 0 && (w == 0 || w == 4 || w == 7))
        printf("col_bwd_r8 NH %d wave %d: total %llu cycles; 16-k-step dense %llu (the halves that request the next pack: %llu), R4 dense %llu, "
               "after dense -> behind the barrier %llu, R0 dense %llu, R0 output %llu, side tiles %llu, the rest (seed, group edges) %llu\n", NH, w,
               __builtin_amdgcn_s_memtime() - stamp_t0, stamp_sum[1], stamp_sum[2], stamp_sum[3], stamp_sum[4], stamp_sum[5], stamp_sum[6],
               stamp_sum[7], stamp_sum[0]);
#endif
}

template <int PREC, int GP, int NH, int XP>
static int launch_col_bwd_r8(const unsigned char* b, long n_pts, const float* d_rgb, const float* rgb, const ColStash& st, float* d_feat,
                             float* d_normal, hipStream_t stream) {
    static bool done = false;
    if (!done) {
        allow_big_lds(color_bwd_r8_kernel<PREC, GP, NH, XP>);
        done = true;
    }
    const long groups = (n_pts + 32 * NH - 1) / (32 * NH);
    hipLaunchKernelGGL((color_bwd_r8_kernel<PREC, GP, NH, XP>), dim3((unsigned)(groups < 256 ? groups : 256)), dim3(512), NH * (XP == 3 ? kR8Half : kR8Half / 2), stream, b,
                       n_pts, d_rgb, rgb, st, d_feat, d_normal);
    return launch_status();
}

int color_bwd_r8(const unsigned char* b, long n_pts, const float* d_rgb, const float* rgb, const ColStash& st, float* d_feat, float* d_normal,
                 int prec, hipStream_t stream) {
    const char* e = getenv("FNEUS_R8_NH");
    const int f = e ? atoi(e) : 0;
    const long groups64 = (n_pts + 63) / 64;
    const bool nh4 = (f == 2 || f == 4) ? f == 4 : groups64 >= 2 * 256;
    const bool exact = st.zbar_lo != nullptr;
    // bf16 zbar planes (gradient precision 1 / 2): the chain runs on those bf16 values (FNEUS_COLB_XHI=0: hi + lo inside the chain)
    const char* xe = getenv("FNEUS_COLB_XHI");
    const bool xhi = xe ? atoi(xe) != 0 : true;
    if (d_feat == nullptr && !(prec == 3 && !exact && xhi && st.dfeat_hi != nullptr)) return -2;       // no rows: the bf16-cotangent form only
#define FNEUS_COL_R8(P, G, X)                                                                                  \
    return nh4 ? launch_col_bwd_r8<P, G, 4, X>(b, n_pts, d_rgb, rgb, st, d_feat, d_normal, stream)            \
               : launch_col_bwd_r8<P, G, 2, X>(b, n_pts, d_rgb, rgb, st, d_feat, d_normal, stream)
    if (prec == 3 && exact) FNEUS_COL_R8(3, 3, 3);
    if (prec == 3 && xhi && (f == 8 || (f != 2 && f != 4 && (n_pts + 127) / 128 > 256))) return launch_col_bwd_r8<3, 1, 8, 1>(b, n_pts, d_rgb, rgb, st, d_feat, d_normal, stream);
    if (prec == 3 && xhi) FNEUS_COL_R8(3, 1, 1);
    if (prec == 3) FNEUS_COL_R8(3, 1, 3);
    if (prec == 1) FNEUS_COL_R8(1, 1, 1);
#undef FNEUS_COL_R8
    return -2;
}

}  // namespace fneus
