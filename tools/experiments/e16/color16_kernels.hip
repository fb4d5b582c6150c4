// Colour-network kernels on the 16-sample-tile engine (mlp_engine16.h).  Same maths, buffers and C ABI as
// color_kernels.hip (reference models/fields.py:150-175 RenderingNetwork.forward, mode 'idr', and its autograd).
// K-slot order of layer 0: 256 feature slots (8 k-steps), then the 33 "side" inputs (2 k-steps) in the reference's
// column order (pts, PE(view), normal).
#include "mlp_engine16.h"
#include "fneus_kernels.h"

namespace fneus {
namespace e16 {

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ReLU in place; returns this lane's 64 sign bits (bit 4*t + reg) for the backward pass
FN_DEV u32x2 relu_mask16(f32x4 (&acc)[16]) {
    u32x2 m = {0u, 0u};
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool pos = acc[t][r] > 0.0f;
            acc[t][r] = pos ? acc[t][r] : 0.0f;
            m[t >> 3] |= (pos ? 1u : 0u) << ((t & 7) * 4 + r);
        }
    return m;
}

template <int TN>
FN_DEV void relu_inplace(f32x4 (&acc)[TN]) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = fmaxf(acc[t][r], 0.0f);
}

template <int PREC, bool TRAIN>
__global__ void __launch_bounds__(256, 2) color_fwd16_kernel(const unsigned char* blob, PointSrc src, long N,
                                                             const float* __restrict__ dirs,      // [N][3] or nullptr (ray mode)
                                                             const float* __restrict__ normal,    // [N][3]
                                                             const float* __restrict__ feat,      // [N][256]
                                                             ColStash st, float* __restrict__ rgb_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    constexpr auto& LY = kColLayout16;
    const size_t LS = (size_t)N * 256;
    unsigned char* wscr = lds_ + wave * kWaveScr;
    u32x2* masks = reinterpret_cast<u32x2*>(st.mask);
    for (long tile0 = (long)blockIdx.x * kWaves; tile0 * 16 < N; tile0 += (long)gridDim.x * kWaves) {
        asm volatile("" : "+s"(blob));
        const Eng eg{blob, lds_, lane, wave};
        const long tile = tile0 + wave;
        const long n = tile * 16 + c;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        const long n0 = tile * 16;
        float side[33];
        {
            float x[3], d[3], pe[27], jc[27];
            load_point(src, nc, x);
            if (dirs) {
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) d[cc] = dirs[nc * 3 + cc];
            } else {
                const long ray = nc / src.m;
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) d[cc] = src.rays_d[ray * 3 + cc];
            }
            posenc<4, false>(d, pe, jc);
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) side[cc] = x[cc];
#pragma unroll
            for (int f = 0; f < 27; ++f) side[3 + f] = pe[f];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) side[30 + cc] = normal[nc * 3 + cc];
        }
        BFrag<PREC> bf[kMaxKS];
        f32x4 acc[16];
        load_f32<16>(acc, feat, 256, nc, q);
        acc_to_bfrag<PREC, 16>(acc, bf);
        vec_to_bfrag<PREC, 33, 2, 8>(side, bf, q);
        if constexpr (TRAIN) store_side48<PREC>(bf, 8, st.side_hi, st.side_lo, nc, q, valid);
#pragma unroll 1
        for (int l = 0; l <= 3; ++l) {
            load_accvec<0, 16>(eg, LY.L[l].bias, acc);
            if (l == 0)
                dense<PREC, 10, 16, 0, 16>(eg, LY.L[0].fwd_hi, LY.L[0].fwd_lo, bf, acc);
            else
                dense<PREC, 8, 16, 0, 16>(eg, LY.L[l].fwd_hi, LY.L[l].fwd_lo, bf, acc);
            if constexpr (TRAIN) {
                masks[((size_t)tile * 4 + l) * 64 + lane] = relu_mask16(acc);
                store_stash<PREC, 16>(wscr, lane, acc, st.u_hi + l * LS, st.u_lo + l * LS, 256, n0, N, 256);
            } else {
                relu_inplace(acc);
            }
            acc_to_bfrag<PREC, 16>(acc, bf);
        }
        f32x4 o[1];
        load_accvec<0, 1>(eg, LY.L[4].bias, o);
        dense<PREC, 8, 1, 0, 1>(eg, LY.L[4].fwd_hi, LY.L[4].fwd_lo, bf, o);
        // rows 0..2 of the single output tile live in registers 0..2 of lane quarter 0
        if (valid && lane < 16) {
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) rgb_out[n * 3 + cc] = 1.0f / (1.0f + expf(-o[0][cc]));   // fields.py:173-174
        }
    }
}

template <int PREC>
__global__ void __launch_bounds__(256, 2) color_bwd16_kernel(const unsigned char* blob, long N,
                                                             const float* __restrict__ d_rgb,   // [N][3]
                                                             const float* __restrict__ rgb,     // [N][3] forward output
                                                             ColStash st, float* __restrict__ d_feat /*[N][256]*/,
                                                             float* __restrict__ d_normal /*[N][3]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    constexpr auto& LY = kColLayout16;
    const size_t LS = (size_t)N * 256;
    unsigned char* wscr = lds_ + wave * kWaveScr;
    const u32x2* masks = reinterpret_cast<const u32x2*>(st.mask);
    for (long tile0 = (long)blockIdx.x * kWaves; tile0 * 16 < N; tile0 += (long)gridDim.x * kWaves) {
        asm volatile("" : "+s"(blob));
        const Eng eg{blob, lds_, lane, wave};
        const long tile = tile0 + wave;
        const long n = tile * 16 + c;
        const bool valid = n < N;
        const long nc = valid ? n : N - 1;
        const long n0 = tile * 16;
        BFrag<PREC> bf[kMaxKS];
        f32x4 acc[19];
        // zbar_4 = d rgb * sigmoid'  (3 rows of one tile)
        {
            f32x4 z[1];
            zero_acc(z);
            if (q == 0) {
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    const float y = rgb[nc * 3 + cc];
                    z[0][cc] = valid ? d_rgb[nc * 3 + cc] * y * (1.0f - y) : 0.0f;
                }
            }
            store_stash<PREC, 1>(wscr, lane, z, st.zbar_hi + 4 * LS, st.zbar_lo + 4 * LS, 32, n0, N, 16);
            acc_to_bfrag<PREC, 1>(z, bf);
        }
        f32x4(&a16)[16] = reinterpret_cast<f32x4(&)[16]>(acc);
        // layer 4 reverse: 1 k-step -> 16 row tiles
        zero_acc(a16);
        dense<PREC, 1, 16, 0, 16>(eg, LY.L[4].rev_hi, LY.L[4].rev_lo, bf, a16);
#pragma unroll 1
        for (int l = 3; l >= 0; --l) {
            // zbar_l = relu'(z_l) * ubar_{l+1}: sign bits from the forward pass (lane-private, one 8-byte load)
            {
                const u32x2 m = masks[((size_t)tile * 4 + l) * 64 + lane];
#pragma unroll
                for (int t = 0; t < 16; ++t)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const bool pos = (m[t >> 3] >> ((t & 7) * 4 + rr)) & 1u;
                        a16[t][rr] = (pos && valid) ? a16[t][rr] : 0.0f;
                    }
            }
            store_stash<PREC, 16>(wscr, lane, a16, st.zbar_hi + l * LS, st.zbar_lo + l * LS, 256, n0, N, 256);
            acc_to_bfrag<PREC, 16>(a16, bf);
            if (l > 0) {
                zero_acc(a16);
                dense<PREC, 8, 16, 0, 16>(eg, LY.L[l].rev_hi, LY.L[l].rev_lo, bf, a16);
            }
        }
        // layer 0 reverse: 19 row tiles (16 feature tiles + 3 side tiles)
        zero_acc(acc);
        dense<PREC, 8, 19, 0, 19>(eg, LY.L[0].rev_hi, LY.L[0].rev_lo, bf, acc);
        store_f32<16>(a16, d_feat, 256, nc, q, valid);
        {
            f32x4(&s3)[3] = reinterpret_cast<f32x4(&)[3]>(acc[16]);
            const float g0 = acc_extract<3, 30>(s3, q), g1 = acc_extract<3, 31>(s3, q), g2 = acc_extract<3, 32>(s3, q);
            if (valid && lane < 16) {
                d_normal[n * 3 + 0] = g0;
                d_normal[n * 3 + 1] = g1;
                d_normal[n * 3 + 2] = g2;
            }
        }
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

int launch_color_fwd(const unsigned char* b, PointSrc src, long n_pts, const float* dirs, const float* normal,
                     const float* feat, ColStash st, float* rgb_out, int prec, int train, hipStream_t stream) {
    if (prec == 3 && train)
        FNEUS16_LAUNCH((color_fwd16_kernel<3, true>), b, src, n_pts, dirs, normal, feat, st, rgb_out);
    else if (prec == 3)
        FNEUS16_LAUNCH((color_fwd16_kernel<3, false>), b, src, n_pts, dirs, normal, feat, st, rgb_out);
    else if (prec == 1 && train)
        FNEUS16_LAUNCH((color_fwd16_kernel<1, true>), b, src, n_pts, dirs, normal, feat, st, rgb_out);
    else if (prec == 1)
        FNEUS16_LAUNCH((color_fwd16_kernel<1, false>), b, src, n_pts, dirs, normal, feat, st, rgb_out);
    else
        return -2;
    return launch_status();
}

int launch_color_bwd(const unsigned char* b, long n_pts, const float* d_rgb, const float* rgb, ColStash st, float* d_feat,
                     float* d_normal, int prec, hipStream_t stream) {
    if (prec == 3)
        FNEUS16_LAUNCH(color_bwd16_kernel<3>, b, n_pts, d_rgb, rgb, st, d_feat, d_normal);
    else if (prec == 1)
        FNEUS16_LAUNCH(color_bwd16_kernel<1>, b, n_pts, d_rgb, rgb, st, d_feat, d_normal);
    else
        return -2;
    return launch_status();
}

}  // namespace e16
}  // namespace fneus
