// Weight gradient of the colour network's OUTPUT layer (lin4: 256 -> 3, reference models/fields.py:170-174 through autograd's addmm
// backward) with exact operands: gradient precision 2 (fneus/ops.py _gprec).
//   dW4[c][k] = sum_n zout[n][c] u3[n][k],   db4[c] = sum_n zout[n][c],   zout[n][c] = d_rgb[n][c] y (1 - y),  y = rgb[n][c]
// It is the one product of a step whose bf16 operand rounding exceeds the exact mode's gradient bounds (3 x 256 outputs, every one a
// sum over all samples of two rounded factors: tools/experiments/r05/gprec_tensors.py).  Round 5 ran it as a launch of the
// fragment-plane GEMM with three MFMAs per product (36 us: a one-workgroup-per-CU launch pays its ramp, its drain and 65 536-atomic
// epilogues whatever it streams).  The product is 50 MFLOP: here it is a streaming pass over the hi + lo planes of u_3 (1 KiB per
// sample, the kernel's only traffic of note) with fp32 FMAs -- zout is formed in fp32 from d_rgb and rgb themselves, not read back
// from a rounded plane.  What the forms of this kernel taught (tools/experiments/r06/cod_time.py, 65 536 samples, cold caches):
//   * every workgroup adding its 771 sums to the same 771 addresses: 38 us (85 with two-lane atomic instructions) for ~10 us of
//     streaming -- same-address float atomics are serialised at the memory side (512 adders per address);
//   * a workgroup per (fragment, slice of the tiles), <= 32 adders per address: 42 us whatever the slice count -- reading 1 KiB of
//     every 16 KiB block streams at 2.2 TB/s (no memory traffic 12.6 us, without the planes 20.5, without the atomics 44);
//   * a branch around a request (`tile < tiles ? load : 0`) makes hipcc wait for every request in turn: 63 us.
// So: a workgroup reads WHOLE tiles (wave w fragments 4 w .. 4 w + 3: 4 KiB contiguous per plane), every request of an iteration is
// unconditional and issued before the first use, the 32 sample lanes of a lane half are summed by a reduce-scatter (93 exchanges, the
// lanes end with 3 different sums each: full-wave atomic instructions), and the sums go to one of 16 REPLICAS of the 771 outputs
// (16 adders per address); a one-workgroup launch behind it folds the replicas into the gradient and clears them.
#include <stdlib.h>
#include "pp_engine.h"
#include "fneus_kernels.h"

namespace fneus {

constexpr int kCodReplicas = 16, kCodStride = 772;      // floats per replica: dW [3][256], db [3], one of padding

__global__ void __launch_bounds__(256) color_out_dw_kernel(const unsigned char* __restrict__ u_hi, const unsigned char* __restrict__ u_lo,
                                                           const float* __restrict__ d_rgb, const float* __restrict__ rgb, long N,
                                                           float* __restrict__ rep) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const PPLane pl = pp_lane(lane);
    const long tiles = (N + 31) / 32;
    float* out = rep + (blockIdx.x % kCodReplicas) * kCodStride;
    float acc[96];                      // value (f, j, c) at index (8 f + j) 3 + c
    float zs[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 96; ++i) acc[i] = 0.0f;
    const float lo_on = u_lo != nullptr ? 1.0f : 0.0f;
    const unsigned char* u_lo_ = u_lo != nullptr ? u_lo : u_hi;
    // two tiles per iteration: 16 + 4 requests in flight per lane
    for (long t0 = (long)blockIdx.x * 2; t0 < tiles; t0 += (long)gridDim.x * 2) {
        bf16x8 vh[2][4], vl[2][4];
        float y[2][3], g[2][3], z[2][3];
        bool valid[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool live = t0 + k < tiles;
            const long tile = live ? t0 + k : tiles - 1;                // clamped: what does not count is multiplied by a zero zout
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                vh[k][f] = pp_load(u_hi + (size_t)tile * kPPBlock, 4 * w + f, pl);
                vl[k][f] = pp_load(u_lo_ + (size_t)tile * kPPBlock, 4 * w + f, pl);
            }
            const long n = tile * 32 + r;
            valid[k] = live && n < N;
            const long nc = n < N ? n : N - 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                y[k][c] = __builtin_nontemporal_load(rgb + nc * 3 + c);
                g[k][c] = __builtin_nontemporal_load(d_rgb + nc * 3 + c);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                z[k][c] = valid[k] ? g[k][c] * y[k][c] * (1.0f - y[k][c]) : 0.0f;
                zs[c] += z[k][c];
            }
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float u = fmaf(lo_on, (float)vl[k][f][j], (float)vh[k][f][j]);
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[(8 * f + j) * 3 + c] = fmaf(z[k][c], u, acc[(8 * f + j) * 3 + c]);
                }
        }
    }
    // reduce-scatter over the 32 samples of a lane half (lanes with equal h hold the same features): at every step a lane keeps one
    // half of its values and adds its partner's copy of that half
    int base = 0;
#define FNEUS_RS_STEP(HALF, M)                                                      \
    {                                                                               \
        const bool up = (r & M) != 0;                                               \
        _Pragma("unroll") for (int i = 0; i < HALF; ++i) {                          \
            const float keep = up ? acc[i + HALF] : acc[i];                         \
            const float send = up ? acc[i] : acc[i + HALF];                         \
            acc[i] = keep + __shfl_xor(send, M, 64);                                \
        }                                                                           \
        base += up ? HALF : 0;                                                      \
    }
    FNEUS_RS_STEP(48, 16)
    FNEUS_RS_STEP(24, 8)
    FNEUS_RS_STEP(12, 4)
    FNEUS_RS_STEP(6, 2)
    FNEUS_RS_STEP(3, 1)
#undef FNEUS_RS_STEP
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int idx = base + i;
        const int f = idx / 24, j = (idx / 3) & 7, c = idx % 3;
        atomicAdd(out + c * 256 + phi(4 * w + f, h, j), acc[i]);
    }
    if (w == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = zs[c];
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) atomicAdd(out + 768 + c, v);
        }
    }
}

// the replicas -> the gradient (accumulated), and cleared for the next launch
__global__ void __launch_bounds__(256) color_out_fold_kernel(float* __restrict__ rep, float* __restrict__ dW, float* __restrict__ db,
                                                             const float* __restrict__ fold_src, int fold_n, float* __restrict__ fold_dst) {
    if (fold_src != nullptr) {          // a rider: sum(fold_src) added to *fold_dst (fixed order: lane-strided, butterfly, waves 0..3)
        __shared__ float wsum[4];
        float v = 0.0f;
        for (int i = threadIdx.x; i < fold_n; i += 256) v += fold_src[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) *fold_dst += (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    }
    for (int i = threadIdx.x; i < 771; i += 256) {
        float v[kCodReplicas];
#pragma unroll
        for (int k = 0; k < kCodReplicas; ++k) v[k] = rep[k * kCodStride + i];
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < kCodReplicas; ++k) {
            s += v[k];
            rep[k * kCodStride + i] = 0.0f;
        }
        if (i < 768) atomicAdd(dW + i, s);
        else if (db != nullptr) atomicAdd(db + (i - 768), s);
    }
}

}  // namespace fneus

using namespace fneus;

extern "C" int fneus_color_out_dw_scratch_floats(void) { return kCodReplicas * kCodStride; }

extern "C" int fneus_color_out_dw(const void* u3_hi, const void* u3_lo, const float* d_rgb, const float* rgb, long n_pts, float* dW, float* db,
                                  float* scratch, const float* fold_src, int fold_n, float* fold_dst, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    if (u3_hi == nullptr || d_rgb == nullptr || rgb == nullptr || dW == nullptr || scratch == nullptr) return -2;
    const long tiles = (n_pts + 31) / 32;
    const long pairs = (tiles + 1) / 2;
    hipLaunchKernelGGL(color_out_dw_kernel, dim3((unsigned)(pairs < 512 ? pairs : 512)), dim3(256), 0, stream,
                       static_cast<const unsigned char*>(u3_hi), static_cast<const unsigned char*>(u3_lo), d_rgb, rgb, n_pts, scratch);
    hipLaunchKernelGGL(color_out_fold_kernel, dim3(1), dim3(256), 0, stream, scratch, dW, db, fold_dst ? fold_src : nullptr, fold_n, fold_dst);
    return fneus::launch_status();
}
