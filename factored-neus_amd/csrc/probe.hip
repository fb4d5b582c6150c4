// Box calibration for bench.py (round 6): two ~100 us probes that tell a slow lease from a regression -- the same code measured 5-8 %
// apart on different MI355X boxes, and the chip holds a lower clock under matrix load on random operands than the
// 2.4 GHz its peak figures assume (MI355X_MICROARCH.md, DVFS give-back).  Not part of the reference's path: no reference line to cite.
//   fneus_probe_mfma: back-to-back v_mfma_f32_32x32x16_bf16 on random operands in registers, every SIMD busy (4 waves per SIMD);
//                     also returns shader cycles and 100 MHz ticks of one wave: the clock the chip held.
//   fneus_probe_copy: a float4 grid-stride copy.
#include "fneus_common.h"
#include "fneus_kernels.h"

namespace fneus {

__global__ void __launch_bounds__(256) probe_mfma_kernel(const bf16x8* __restrict__ src, int iters, unsigned long long* __restrict__ ticks,
                                                         float* __restrict__ sink) {
    bf16x8 a[4], b[4];
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = src[(threadIdx.x + 256 * i) & 2047];
        b[i] = src[(threadIdx.x + 256 * i + 1024) & 2047];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j & 3] = mfma32(a[j & 3], b[(j + (j >> 2)) & 3], acc[j & 3]);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7] + acc[i][15];
    if (s == 123.456f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ticks[0] = t1 - t0;
        ticks[1] = r1 - r0;
    }
}

__global__ void __launch_bounds__(256) probe_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace fneus

using namespace fneus;

extern "C" long fneus_probe_mfma_flops(int iters) { return 1024L * 4 * iters * 16 * 32768; }

extern "C" int fneus_probe_mfma(const void* operands_32kib, int iters, unsigned long long* ticks /*[2]*/, float* sink, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (operands_32kib == nullptr || ticks == nullptr || sink == nullptr || iters <= 0) return -2;
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(1024), dim3(256), 0, stream, static_cast<const bf16x8*>(operands_32kib), iters, ticks, sink);
    return fneus::launch_status();
}

extern "C" int fneus_probe_copy(const void* src, void* dst, long bytes, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (bytes <= 0) return 0;
    if (src == nullptr || dst == nullptr || (bytes & 15)) return -2;
    hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 16), dim3(256), 0, stream, static_cast<const f32x4*>(src), static_cast<f32x4*>(dst), bytes / 16);
    return fneus::launch_status();
}
