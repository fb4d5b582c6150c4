// SDF-network kernels on the 16-sample-tile engine (mlp_engine16.h).  Same maths, buffers and C ABI as
// sdf_kernels.hip (reference models/fields.py:74-111 SDFNetwork.forward/.sdf/.gradient and their autograd); one
// workgroup = 4 wavefronts = 64 samples sharing the weight stream, two workgroups per CU.
#include "mlp_engine16.h"
#include "fneus_kernels.h"

namespace fneus {
namespace e16 {

// ---- forward chain shared by K1/K2 -----------------------------------------------------------------------
// On return acc[0..15] = feature tiles (natural order), acc[16] row 0 (reg 0 of lane quarter 0) = sdf.
template <int PREC, bool SDF_ONLY, bool STASH>
FN_DEV void sdf_forward_chain(const Eng& eg, const float (&pe)[39], BFrag<PREC> (&bf)[kMaxKS], f32x4 (&acc)[17],
                              const SdfStash& st, long N, long n, bool valid, long tile) {
    const int lane = eg.lane, q = lane >> 4;
    const long n0 = tile * 16;
    unsigned char* wscr = eg.lds + eg.wave * kWaveScr;
    unsigned char* psb = STASH ? st.ps + (size_t)tile * 8 * kSigBlock : nullptr;
    constexpr auto& LY = kSdfLayout16;
    const size_t LS = (size_t)N * 256;
    BFrag<PREC> pef[2];
    vec_to_bfrag<PREC, 39, 2, 0>(pe, bf, q);
    pef[0] = bf[0];
    pef[1] = bf[1];
    if constexpr (STASH) store_side48<PREC>(bf, 0, st.pe_hi, st.pe_lo, n, q, valid);
    f32x4(&a16)[16] = reinterpret_cast<f32x4(&)[16]>(acc);
    f32x4(&a14)[14] = reinterpret_cast<f32x4(&)[14]>(acc);
#pragma unroll 1
    for (int l = 0; l <= 7; ++l) {
        if (l == 0) {
            load_accvec<0, 16>(eg, LY.L[0].bias, a16);
            dense<PREC, 2, 16, 0, 16>(eg, LY.L[0].fwd_hi, LY.L[0].fwd_lo, bf, a16);
        } else if (l == 3) {   // 256 -> 217 (14 tiles); its output + PE feeds layer 4 (skip connection, fields.py:83-84)
            load_accvec<0, 14>(eg, LY.L[3].bias, a14);
            dense<PREC, 8, 14, 0, 14>(eg, LY.L[3].fwd_hi, LY.L[3].fwd_lo, bf, a14);
        } else if (l == 4) {   // 9 k-steps; 1/sqrt2 folded into the pack
            load_accvec<0, 16>(eg, LY.L[4].bias, a16);
            dense<PREC, 9, 16, 0, 16>(eg, LY.L[4].fwd_hi, LY.L[4].fwd_lo, bf, a16);
        } else {
            load_accvec<0, 16>(eg, LY.L[l].bias, a16);
            dense<PREC, 8, 16, 0, 16>(eg, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, a16);
        }
        if (l == 3) {
            if constexpr (STASH) {
                softplus_ps<PREC, 14>(a14, psb + (size_t)3 * kSigBlock, lane);
                store_stash<PREC, 14>(wscr, lane, a14, st.h_hi + 3 * LS, st.h_lo + 3 * LS, 256, n0, N, 224);
            } else {
                softplus_inplace(a14);
            }
            acc_to_bfrag<PREC, 14>(a14, bf);
            bf[7] = pef[0];
            bf[8] = pef[1];
        } else {
            if constexpr (STASH) {
                softplus_ps<PREC, 16>(a16, psb + (size_t)l * kSigBlock, lane);
                store_stash<PREC, 16>(wscr, lane, a16, st.h_hi + l * LS, st.h_lo + l * LS, 256, n0, N, 256);
            } else {
                softplus_inplace(a16);
            }
            acc_to_bfrag<PREC, 16>(a16, bf);
        }
    }
    // layer 8 (linear)
    if constexpr (SDF_ONLY) {
        f32x4(&a1)[1] = reinterpret_cast<f32x4(&)[1]>(acc[16]);
        load_accvec<16, 1>(eg, LY.L[8].bias, a1);
        dense<PREC, 8, 17, 16, 1>(eg, LY.L[8].fwd_hi, LY.L[8].fwd_lo, bf, a1);
    } else {
        load_accvec<0, 17>(eg, LY.L[8].bias, acc);
        dense<PREC, 8, 17, 0, 17>(eg, LY.L[8].fwd_hi, LY.L[8].fwd_lo, bf, acc);
    }
}

#define FNEUS16_PROLOGUE()                                                              \
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];                 \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                          \
    const int c = lane & 15, q = lane >> 4;                                              \
    (void)c; (void)q

// ---- K1 ----------------------------------------------------------------------------------------------------
template <int PREC>
__global__ void __launch_bounds__(256, 2) sdf_fwd16_kernel(const unsigned char* blob, PointSrc src, long N,
                                                           float* __restrict__ sdf_out) {
    FNEUS16_PROLOGUE();
    SdfStash st{};
    for (long tile0 = (long)blockIdx.x * kWaves; tile0 * 16 < N; tile0 += (long)gridDim.x * kWaves) {
        // launder the blob pointer: otherwise LICM hoists statically addressed weight loads out of the tile loop
        asm volatile("" : "+s"(blob));
        const Eng eg{blob, lds_, lane, wave};
        const long tile = tile0 + wave;
        const long n = tile * 16 + c;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        float x[3], pe[39], jc[39];
        load_point(src, nc, x);
        posenc<6, false>(x, pe, jc);
        BFrag<PREC> bf[kMaxKS];
        f32x4 acc[17];
        sdf_forward_chain<PREC, true, false>(eg, pe, bf, acc, st, N, nc, valid, tile);
        if (valid && lane < 16) sdf_out[n] = acc[16][0];
    }
}

// ---- K2 ----------------------------------------------------------------------------------------------------
// g[t] *= sigma'(z_l) from the lane-private stash block; with TRAIN the product a_l also goes to its private block
template <int PREC, int TN, bool TRAIN>
FN_DEV void mul_sig_priv(f32x4 (&g)[TN], const unsigned char* __restrict__ ps, unsigned char* __restrict__ pa, int lane) {
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        float sv[4], av[4];
        sig_get(ps, t, lane, sv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            av[e] = g[t][e] * sv[e];
            g[t][e] = av[e];
        }
        if constexpr (TRAIN) priv_put<PREC>(pa, t, lane, av);
    }
}

template <int PREC, bool TRAIN>
__global__ void __launch_bounds__(256, 2) sdf_fwd_grad16_kernel(const unsigned char* blob, PointSrc src, long N,
                                                                SdfStash st, float* __restrict__ sdf_out,
                                                                float* __restrict__ feat_out,
                                                                float* __restrict__ normal_out) {
    FNEUS16_PROLOGUE();
    constexpr auto& LY = kSdfLayout16;
    constexpr size_t PB = priv_block<PREC>();
    const size_t LS = (size_t)N * 256;
    unsigned char* wscr = lds_ + wave * kWaveScr;
    for (long tile0 = (long)blockIdx.x * kWaves; tile0 * 16 < N; tile0 += (long)gridDim.x * kWaves) {
        asm volatile("" : "+s"(blob));
        const Eng eg{blob, lds_, lane, wave};
        const long tile = tile0 + wave;
        const long n = tile * 16 + c;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        const long n0 = tile * 16;
        float x[3], pe[39], jc[39];
        load_point(src, nc, x);
        posenc<6, true>(x, pe, jc);
        BFrag<PREC> bf[kMaxKS];
        f32x4 acc[17];
        sdf_forward_chain<PREC, false, true>(eg, pe, bf, acc, st, N, nc, valid, tile);
        const unsigned char* psb = st.ps + (size_t)tile * 8 * kSigBlock;
        unsigned char* pab = TRAIN ? st.pa + (size_t)tile * 8 * PB : nullptr;
        if (valid && lane < 16) sdf_out[n] = acc[16][0];
        f32x4(&g16)[16] = reinterpret_cast<f32x4(&)[16]>(acc);
        f32x4(&g14)[14] = reinterpret_cast<f32x4(&)[14]>(acc);
        store_f32<16>(g16, feat_out, 256, nc, q, valid);
        if constexpr (TRAIN) store_stash<PREC, 16>(wscr, lane, g16, st.feat_hi, st.feat_lo, 256, n0, N, 256);
        // ---- reverse sweep: g = d sdf / d u_l   ----
        load_accvec<0, 16>(eg, LY.extra, g16);                      // g_hat(h_8) = row 0 of W_8
        f32x4 qskip[3];
#pragma unroll 1
        for (int l = 7; l >= 1; --l) {
            if (l == 3) {
                mul_sig_priv<PREC, 14, TRAIN>(g14, psb + (size_t)3 * kSigBlock, pab + (size_t)3 * PB, lane);
                if constexpr (TRAIN)
                    store_stash<PREC, 14>(wscr, lane, g14, st.a_hi + 3 * LS, st.a_lo + 3 * LS, 256, n0, N, 224);
                acc_to_bfrag<PREC, 14>(g14, bf);
                zero_acc(g16);
                dense<PREC, 7, 16, 0, 16>(eg, LY.L[3].rev_hi, LY.L[3].rev_lo, bf, g16);
            } else {
                mul_sig_priv<PREC, 16, TRAIN>(g16, psb + (size_t)l * kSigBlock, pab + (size_t)l * PB, lane);   // a_l
                if constexpr (TRAIN)
                    store_stash<PREC, 16>(wscr, lane, g16, st.a_hi + l * LS, st.a_lo + l * LS, 256, n0, N, 256);
                acc_to_bfrag<PREC, 16>(g16, bf);
                if (l == 4) {   // 17 row tiles: 0..13 -> g_hat(h_4), 14..16 -> q_skip (PE part of the skip input)
                    zero_acc(acc);
                    dense<PREC, 8, 17, 0, 17>(eg, LY.L[4].rev_hi, LY.L[4].rev_lo, bf, acc);
                    qskip[0] = acc[14];
                    qskip[1] = acc[15];
                    qskip[2] = acc[16];
                } else {
                    zero_acc(g16);
                    dense<PREC, 8, 16, 0, 16>(eg, LY.L[l].rev_hi, LY.L[l].rev_lo, bf, g16);
                }
            }
        }
        // layer 0: 3 row tiles (39 PE inputs)
        f32x4 qv[3];
        mul_sig_priv<PREC, 16, TRAIN>(g16, psb, pab, lane);
        if constexpr (TRAIN) store_stash<PREC, 16>(wscr, lane, g16, st.a_hi, st.a_lo, 256, n0, N, 256);
        acc_to_bfrag<PREC, 16>(g16, bf);
        zero_acc(qv);
        dense<PREC, 8, 3, 0, 3>(eg, LY.L[0].rev_hi, LY.L[0].rev_lo, bf, qv);
#pragma unroll
        for (int t = 0; t < 3; ++t) qv[t] += qskip[t];
        // normal = J^T q
        float nrm[3];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            float coef[39];
#pragma unroll
            for (int f = 0; f < 39; ++f) coef[f] = ((f % 3) == cc) ? jc[f] : 0.0f;
            nrm[cc] = sum_q(acc_dot_partial<3, 39>(qv, coef, q));
        }
        if (valid && lane < 16) {
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) normal_out[n * 3 + cc] = nrm[cc];
        }
    }
}

// ---- K3 ----------------------------------------------------------------------------------------------------
// Backward of (sdf, feature, normal) w.r.t. the SDF-network weights: the two chains of SURVEY.md Appendix A
// (see sdf_kernels.hip for the derivation).
template <int PREC, int TN>
FN_DEV void asc_post(f32x4 (&acc)[TN], const unsigned char* __restrict__ ps, const unsigned char* __restrict__ pa,
                     f32x4* __restrict__ cs, int lane) {
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        float sv[4], av[4];
        sig_get(ps, t, lane, sv);
        priv_get<PREC>(pa, t, lane, av);
        f32x4 cv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float abar = acc[t][e];
            cv[e] = kBeta * (1.0f - sv[e]) * av[e] * abar;      // softplus'' * g_hat * abar  (a = s * g_hat)
            acc[t][e] = sv[e] * abar;
        }
        cs[t * 64 + lane] = cv;
    }
}

template <int PREC, int TN>
FN_DEV void desc_post(f32x4 (&acc)[TN], const unsigned char* __restrict__ ps, const f32x4* __restrict__ cs, int lane) {
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        float sv[4];
        sig_get(ps, t, lane, sv);
        const f32x4 cv = cs[t * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = sv[e] * acc[t][e] + cv[e];
    }
}

template <int PREC>
__global__ void __launch_bounds__(256, 2) sdf_bwd16_kernel(const unsigned char* blob, PointSrc src, long N, SdfStash st,
                                                           SdfBwdBufs bb, const float* __restrict__ d_sdf,
                                                           const float* __restrict__ d_feat,
                                                           const float* __restrict__ d_normal) {
    FNEUS16_PROLOGUE();
    constexpr auto& LY = kSdfLayout16;
    constexpr size_t PB = priv_block<PREC>();
    constexpr int CB = 16 * 64;          // f32x4 entries per (tile, layer) block of the coupling scratch
    const size_t LS = (size_t)N * 256;   // layer stride of the [L][N][256] planes
    unsigned char* wscr = lds_ + wave * kWaveScr;
    for (long tile0 = (long)blockIdx.x * kWaves; tile0 * 16 < N; tile0 += (long)gridDim.x * kWaves) {
        asm volatile("" : "+s"(blob));
        const Eng eg{blob, lds_, lane, wave};
        const long tile = tile0 + wave;
        const long n = tile * 16 + c;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        const long n0 = tile * 16;
        f32x4* cs = bb.cscratch + (size_t)tile * 8 * CB;
        const unsigned char* psb = st.ps + (size_t)tile * 8 * kSigBlock;
        const unsigned char* pab = st.pa + (size_t)tile * 8 * PB;
        BFrag<PREC> bf[kMaxKS];
        BFrag<PREC> qf[2];
        f32x4 acc[17];
        f32x4(&a16)[16] = reinterpret_cast<f32x4(&)[16]>(acc);
        f32x4(&a14)[14] = reinterpret_cast<f32x4(&)[14]>(acc);
        // ---- qbar = J nbar ----
        {
            float x[3], pe[39], jc[39], qb[39];
            load_point(src, nc, x);
            posenc<6, true>(x, pe, jc);
            float nb[3];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) nb[cc] = valid ? d_normal[nc * 3 + cc] : 0.0f;
#pragma unroll
            for (int f = 0; f < 39; ++f) qb[f] = jc[f] * nb[f % 3];
            vec_to_bfrag<PREC, 39, 2, 0>(qb, bf, q);
            qf[0] = bf[0];
            qf[1] = bf[1];
            store_side48<PREC>(bf, 0, bb.qbar_hi, bb.qbar_lo, nc, q, valid);
        }
        // ---- ascending chain ----
#pragma unroll 1
        for (int l = 0; l <= 7; ++l) {
            if (l == 3) {
                zero_acc(a14);
                dense<PREC, 8, 14, 0, 14>(eg, LY.L[3].fwd_hi, LY.L[3].fwd_lo, bf, a14);
                asc_post<PREC, 14>(a14, psb + (size_t)3 * kSigBlock, pab + (size_t)3 * PB, cs + 3 * CB, lane);
                store_stash<PREC, 14>(wscr, lane, a14, bb.adj_hi + 3 * LS, bb.adj_lo + 3 * LS, 256, n0, N, 224);
                acc_to_bfrag<PREC, 14>(a14, bf);
                bf[7] = qf[0];
                bf[8] = qf[1];
            } else {
                zero_acc(a16);
                if (l == 0)
                    dense<PREC, 2, 16, 0, 16>(eg, LY.L[0].fwd_hi, LY.L[0].fwd_lo, bf, a16);
                else if (l == 4)
                    dense<PREC, 9, 16, 0, 16>(eg, LY.L[4].fwd_hi, LY.L[4].fwd_lo, bf, a16);
                else
                    dense<PREC, 8, 16, 0, 16>(eg, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, a16);
                asc_post<PREC, 16>(a16, psb + (size_t)l * kSigBlock, pab + (size_t)l * PB, cs + l * CB, lane);
                store_stash<PREC, 16>(wscr, lane, a16, bb.adj_hi + l * LS, bb.adj_lo + l * LS, 256, n0, N, 256);
                acc_to_bfrag<PREC, 16>(a16, bf);
            }
        }
        // ---- descending chain ----
        load_f32<16>(a16, d_feat, 256, nc, q);
        if (!valid) zero_acc(a16);
        {
            f32x4(&z1)[1] = reinterpret_cast<f32x4(&)[1]>(acc[16]);
            zero_acc(z1);
            if (q == 0 && valid) acc[16][0] = d_sdf[nc];
            store_stash<PREC, 16>(wscr, lane, a16, bb.zbar_hi + 8 * LS, bb.zbar_lo + 8 * LS, 256, n0, N, 256);
            store_stash<PREC, 1>(wscr, lane, z1, bb.zsdf_hi, bb.zsdf_lo, 32, n0, N, 16);
        }
        acc_to_bfrag<PREC, 17>(acc, bf);
        zero_acc(a16);
        dense<PREC, 9, 16, 0, 16>(eg, LY.L[8].rev_hi, LY.L[8].rev_lo, bf, a16);
#pragma unroll 1
        for (int l = 7; l >= 1; --l) {
            // here acc = ubar_{l+1} = hbar_{l+1};  zbar_l = s_l * hbar_{l+1} + c_l
            if (l == 3) {
                desc_post<PREC, 14>(a14, psb + (size_t)3 * kSigBlock, cs + 3 * CB, lane);
                store_stash<PREC, 14>(wscr, lane, a14, bb.zbar_hi + 3 * LS, bb.zbar_lo + 3 * LS, 256, n0, N, 224);
                acc_to_bfrag<PREC, 14>(a14, bf);
                zero_acc(a16);
                dense<PREC, 7, 16, 0, 16>(eg, LY.L[3].rev_hi, LY.L[3].rev_lo, bf, a16);
            } else {
                desc_post<PREC, 16>(a16, psb + (size_t)l * kSigBlock, cs + l * CB, lane);
                store_stash<PREC, 16>(wscr, lane, a16, bb.zbar_hi + l * LS, bb.zbar_lo + l * LS, 256, n0, N, 256);
                acc_to_bfrag<PREC, 16>(a16, bf);
                if (l == 4) {   // ubar_4 restricted to the h_4 rows (14 tiles of the 17-tile reverse pack)
                    zero_acc(a14);
                    dense<PREC, 8, 17, 0, 14>(eg, LY.L[4].rev_hi, LY.L[4].rev_lo, bf, a14);
                } else {
                    zero_acc(a16);
                    dense<PREC, 8, 16, 0, 16>(eg, LY.L[l].rev_hi, LY.L[l].rev_lo, bf, a16);
                }
            }
        }
        desc_post<PREC, 16>(a16, psb, cs, lane);
        store_stash<PREC, 16>(wscr, lane, a16, bb.zbar_hi, bb.zbar_lo, 256, n0, N, 256);
    }
}

static inline int grid16(long n_pts) {
    const long wg = (n_pts + 16 * kWaves - 1) / (16 * kWaves);
    const long cap = 256 * 2 * 4;
    return (int)(wg < 1 ? 1 : (wg > cap ? cap : wg));
}

template <class K>
static void big_lds_once(K k) {
    static bool done = false;
    if (!done) {
        allow_big_lds(k);
        done = true;
    }
}

#define FNEUS16_LAUNCH(KERNEL, ...)                                                                               \
    do {                                                                                                          \
        big_lds_once(KERNEL);                                                                                     \
        hipLaunchKernelGGL(KERNEL, dim3(grid16(n_pts)), dim3(64 * kWaves), kEngineLds, stream, __VA_ARGS__);      \
    } while (0)

int launch_sdf_fwd(const unsigned char* b, PointSrc src, long n_pts, float* sdf_out, int prec, hipStream_t stream) {
    if (prec == 3)
        FNEUS16_LAUNCH(sdf_fwd16_kernel<3>, b, src, n_pts, sdf_out);
    else if (prec == 1)
        FNEUS16_LAUNCH(sdf_fwd16_kernel<1>, b, src, n_pts, sdf_out);
    else
        return -2;
    return launch_status();
}

int launch_sdf_fwd_grad(const unsigned char* b, PointSrc src, long n_pts, SdfStash st, float* sdf_out, float* feat_out,
                        float* normal_out, int prec, int train, hipStream_t stream) {
    if (prec == 3 && train)
        FNEUS16_LAUNCH((sdf_fwd_grad16_kernel<3, true>), b, src, n_pts, st, sdf_out, feat_out, normal_out);
    else if (prec == 3)
        FNEUS16_LAUNCH((sdf_fwd_grad16_kernel<3, false>), b, src, n_pts, st, sdf_out, feat_out, normal_out);
    else if (prec == 1 && train)
        FNEUS16_LAUNCH((sdf_fwd_grad16_kernel<1, true>), b, src, n_pts, st, sdf_out, feat_out, normal_out);
    else if (prec == 1)
        FNEUS16_LAUNCH((sdf_fwd_grad16_kernel<1, false>), b, src, n_pts, st, sdf_out, feat_out, normal_out);
    else
        return -2;
    return launch_status();
}

int launch_sdf_bwd(const unsigned char* b, PointSrc src, long n_pts, SdfStash st, SdfBwdBufs bb, const float* d_sdf,
                   const float* d_feat, const float* d_normal, int prec, hipStream_t stream) {
    if (prec == 3)
        FNEUS16_LAUNCH(sdf_bwd16_kernel<3>, b, src, n_pts, st, bb, d_sdf, d_feat, d_normal);
    else if (prec == 1)
        FNEUS16_LAUNCH(sdf_bwd16_kernel<1>, b, src, n_pts, st, bb, d_sdf, d_feat, d_normal);
    else
        return -2;
    return launch_status();
}

}  // namespace e16
}  // namespace fneus
