// experiment: K1 with 4-wave workgroups, waves kept in step by raw barriers (L1 coalescing of the weight stream)
#ifndef SYNC_STAGES
#define SYNC_STAGES 0
#endif
#if SYNC_STAGES > 0
#define FNEUS_WAVE_SYNC_STAGES SYNC_STAGES
#endif
#include "mlp_engine.h"
#include "fneus_kernels.h"
namespace fneus {
template <int TN>
FN_DEV void softplus_inplace(f32x16 (&acc)[TN]) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = softplus100(acc[t][r]);
}
template <int PREC>
__global__ void __launch_bounds__(256, 1) k1sync(const unsigned char* blob, const float* pts, long N, float* __restrict__ sdf_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    constexpr auto& LY = kSdfLayout;
    for (long tile0 = (long)blockIdx.x * 4; tile0 * 32 < N; tile0 += (long)gridDim.x * 4) {
        asm volatile("" : "+s"(blob));
        const long n = (tile0 + wave) * 32 + r;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        float x[3], pe[39], jc[39];
        for (int c = 0; c < 3; ++c) x[c] = pts[nc * 3 + c];
        posenc<6, false>(x, pe, jc);
        BFrag<PREC> bf[kMaxKS];
        f32x16 acc[9];
        BFrag<PREC> pef[3];
        vec_to_bfrag<PREC, 39, 3, 0>(pe, bf, h);
        for (int i = 0; i < 3; ++i) pef[i] = bf[i];
        f32x16(&a8)[8] = reinterpret_cast<f32x16(&)[8]>(acc);
        load_accvec<8, 0, 8>(blob, LY.L[0].bias, a8, lane);
        dense<PREC, 3, 8, 0, 8>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, bf, a8, lane);
        softplus_inplace(a8); acc_to_bfrag<PREC, 8>(a8, bf);
        for (int l = 1; l <= 2; ++l) {
            load_accvec<8, 0, 8>(blob, LY.L[l].bias, a8, lane);
            dense<PREC, 16, 8, 0, 8>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, a8, lane);
            softplus_inplace(a8); acc_to_bfrag<PREC, 8>(a8, bf);
        }
        {
            f32x16(&a7)[7] = reinterpret_cast<f32x16(&)[7]>(acc);
            load_accvec<7, 0, 7>(blob, LY.L[3].bias, a7, lane);
            dense<PREC, 16, 7, 0, 7>(blob, LY.L[3].fwd_hi, LY.L[3].fwd_lo, bf, a7, lane);
            softplus_inplace(a7); acc_to_bfrag<PREC, 7>(a7, bf);
            for (int i = 0; i < 3; ++i) bf[14 + i] = pef[i];
        }
        load_accvec<8, 0, 8>(blob, LY.L[4].bias, a8, lane);
        dense<PREC, 17, 8, 0, 8>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, bf, a8, lane);
        softplus_inplace(a8); acc_to_bfrag<PREC, 8>(a8, bf);
        for (int l = 5; l <= 7; ++l) {
            load_accvec<8, 0, 8>(blob, LY.L[l].bias, a8, lane);
            dense<PREC, 16, 8, 0, 8>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, a8, lane);
            softplus_inplace(a8); acc_to_bfrag<PREC, 8>(a8, bf);
        }
        f32x16(&a1)[1] = reinterpret_cast<f32x16(&)[1]>(acc[8]);
        load_accvec<9, 8, 1>(blob, LY.L[8].bias, a1, lane);
        dense<PREC, 16, 9, 8, 1>(blob, LY.L[8].fwd_hi, LY.L[8].fwd_lo, bf, a1, lane);
        if (valid && lane < 32) sdf_out[n] = acc[8][0];
    }
}
}
extern "C" int k1sync_run(const void* blob, const float* pts, long n, float* out, int prec, void* stream) {
    long wgs = ((n + 31) / 32 + 3) / 4; if (wgs > 1024) wgs = 1024;
    if (prec == 3) hipLaunchKernelGGL(fneus::k1sync<3>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)blob, pts, n, out);
    else hipLaunchKernelGGL(fneus::k1sync<1>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)blob, pts, n, out);
    return 0;
}
