// SDF-network kernels on the 16-sample-tile engine (mlp_engine16.h).  Same maths and C ABI semantics as
// sdf_kernels.hip (reference models/fields.py:74-111); 8 wavefronts per workgroup share the weight stream.
#include "mlp_engine16.h"
#include "fneus_kernels.h"

namespace fneus {
namespace e16 {

// forward chain; on return acc[0..15] = feature tiles, acc[16] row 0 (reg 0 of quarter 0) = sdf.
template <int PREC, bool SDF_ONLY, bool STASH>
FN_DEV void sdf_forward_chain(const Cx& cx, const float (&pe)[39], BFrag<PREC> (&bf)[kMaxKS], f32x4 (&acc)[17],
                              const SdfStash& st, long N, long n, bool valid) {
    const int q = cx.lane >> 4;
    constexpr auto& LY = kSdfLayout16;
    BFrag<PREC> pef[2];
    vec_to_bfrag<PREC, 39, 2, 0>(pe, bf, q);
    pef[0] = bf[0];
    pef[1] = bf[1];
    if constexpr (STASH) {
        if (valid) {   // PE rows [N][48]: columns phi16(ks, q, j) < 48
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int col = 32 * ks + 16 * g + 4 * q;
                    if (col < 48) {
                        bf16x4 vh, vl;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            vh[e] = bf[ks].hi[4 * g + e];
                            if constexpr (PREC == 3) vl[e] = bf[ks].lo[4 * g + e];
                        }
                        *reinterpret_cast<bf16x4*>(st.pe_hi + n * 48 + col) = vh;
                        if constexpr (PREC == 3) *reinterpret_cast<bf16x4*>(st.pe_lo + n * 48 + col) = vl;
                    }
                }
        }
    }
    f32x4(&a16)[16] = reinterpret_cast<f32x4(&)[16]>(acc);
    const size_t LS = (size_t)N * 256;
    load_accvec<0, 16>(cx, LY.L[0].bias, a16);
    dense<PREC, 2, 16, 0, 16>(cx, LY.L[0].fwd_hi, LY.L[0].fwd_lo, bf, a16);
    softplus_inplace(a16);
    if constexpr (STASH) store_stash<PREC, 16>(a16, st.h_hi, st.h_lo, 256, n, q, valid, 256);
    acc_to_bfrag<PREC, 16>(a16, bf);
    for (int l = 1; l <= 2; ++l) {
        load_accvec<0, 16>(cx, LY.L[l].bias, a16);
        dense<PREC, 8, 16, 0, 16>(cx, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, a16);
        softplus_inplace(a16);
        if constexpr (STASH) store_stash<PREC, 16>(a16, st.h_hi + l * LS, st.h_lo + l * LS, 256, n, q, valid, 256);
        acc_to_bfrag<PREC, 16>(a16, bf);
    }
    {
        f32x4(&a14)[14] = reinterpret_cast<f32x4(&)[14]>(acc);
        load_accvec<0, 14>(cx, LY.L[3].bias, a14);
        dense<PREC, 8, 14, 0, 14>(cx, LY.L[3].fwd_hi, LY.L[3].fwd_lo, bf, a14);
        softplus_inplace(a14);
        if constexpr (STASH) store_stash<PREC, 14>(a14, st.h_hi + 3 * LS, st.h_lo + 3 * LS, 256, n, q, valid, 224);
        acc_to_bfrag<PREC, 14>(a14, bf);
        bf[7] = pef[0];
        bf[8] = pef[1];
    }
    load_accvec<0, 16>(cx, LY.L[4].bias, a16);
    dense<PREC, 9, 16, 0, 16>(cx, LY.L[4].fwd_hi, LY.L[4].fwd_lo, bf, a16);
    softplus_inplace(a16);
    if constexpr (STASH) store_stash<PREC, 16>(a16, st.h_hi + 4 * LS, st.h_lo + 4 * LS, 256, n, q, valid, 256);
    acc_to_bfrag<PREC, 16>(a16, bf);
    for (int l = 5; l <= 7; ++l) {
        load_accvec<0, 16>(cx, LY.L[l].bias, a16);
        dense<PREC, 8, 16, 0, 16>(cx, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, a16);
        softplus_inplace(a16);
        if constexpr (STASH) store_stash<PREC, 16>(a16, st.h_hi + l * LS, st.h_lo + l * LS, 256, n, q, valid, 256);
        acc_to_bfrag<PREC, 16>(a16, bf);
    }
    if constexpr (SDF_ONLY) {
        f32x4(&a1)[1] = reinterpret_cast<f32x4(&)[1]>(acc[16]);
        load_accvec<16, 1>(cx, LY.L[8].bias, a1);
        dense<PREC, 8, 17, 16, 1>(cx, LY.L[8].fwd_hi, LY.L[8].fwd_lo, bf, a1);
    } else {
        load_accvec<0, 17>(cx, LY.L[8].bias, acc);
        dense<PREC, 8, 17, 0, 17>(cx, LY.L[8].fwd_hi, LY.L[8].fwd_lo, bf, acc);
    }
}

template <int PREC>
__global__ void __launch_bounds__(512, 2) sdf_fwd16_kernel(const unsigned char* blob, PointSrc src, long N,
                                                           float* __restrict__ sdf_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15;
    SdfStash st{};
    for (long tile0 = (long)blockIdx.x * kWaves; tile0 * 16 < N; tile0 += (long)gridDim.x * kWaves) {
        asm volatile("" : "+s"(blob));
        const Cx cx{blob, ring, lane, wave};
        const long n = (tile0 + wave) * 16 + c;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        float x[3], pe[39], jc[39];
        load_point(src, nc, x);
        posenc<6, false>(x, pe, jc);
        BFrag<PREC> bf[kMaxKS];
        f32x4 acc[17];
        sdf_forward_chain<PREC, true, false>(cx, pe, bf, acc, st, N, nc, valid);
        if (valid && lane < 16) sdf_out[n] = acc[16][0];
    }
}

}  // namespace e16
}  // namespace fneus

using namespace fneus;

static inline int grid16(long n_pts) {
    const long wg = (n_pts + 16 * e16::kWaves - 1) / (16 * e16::kWaves);
    return (int)(wg < 1 ? 1 : (wg > 1024 ? 1024 : wg));
}

static void init16() {
    static bool done = false;
    if (done) return;
    allow_big_lds(e16::sdf_fwd16_kernel<3>);
    allow_big_lds(e16::sdf_fwd16_kernel<1>);
    done = true;
}

extern "C" int fneus16_sdf_fwd(const void* blob, const float* pts, const float* rays_o, const float* rays_d,
                               const float* t, int m, long n_pts, float* sdf_out, int prec, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    init16();
    PointSrc src{pts, rays_o, rays_d, t, m > 0 ? m : 1};
    const unsigned char* b = reinterpret_cast<const unsigned char*>(blob);
    if (prec == 3)
        hipLaunchKernelGGL(e16::sdf_fwd16_kernel<3>, dim3(grid16(n_pts)), dim3(512), e16::kEngineLds, stream, b, src, n_pts, sdf_out);
    else if (prec == 1)
        hipLaunchKernelGGL(e16::sdf_fwd16_kernel<1>, dim3(grid16(n_pts)), dim3(512), e16::kEngineLds, stream, b, src, n_pts, sdf_out);
    else
        return -2;
    return fneus::launch_status();
}
