// Two waves per SIMD (512-thread workgroup), each issuing `v_mfma_f32_32x32x16_bf16` streams: cycles per MFMA for
//   mode 0: ONE accumulator (every MFMA depends on the previous one)      mode 1: three accumulators round-robin
//   mode 2: one accumulator, 2 ds_read_b128 per 3 MFMAs (the r8 k-step)   mode 3: three accumulators + the ds_reads
// printed per wave of block 0 (older half w < 4, younger half w >= 4).  hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>
__global__ void __launch_bounds__(512, 1) k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(0.001f * (lane + e + i));
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)(0.002f * (lane - e + i));
    for (int i = threadIdx.x; i < 16 * 1024; i += 512) ((float*)lds)[i] = 0.001f * i;
    __syncthreads();
    f32x16 acc[3];
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const unsigned char* fl = lds + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (MODE >= 2) {
                b[0] = *reinterpret_cast<const bf16x8*>(fl + (2 * s) * 1024);
                b[1] = *reinterpret_cast<const bf16x8*>(fl + (2 * s + 1) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0 || MODE == 2) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s & 3], b[0], acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 1) & 3], b[1], acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 1) & 3], b[0], acc[0], 0, 0, 0);
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s & 3], b[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 1) & 3], b[1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 1) & 3], b[0], acc[2], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0) cyc[w] = t1 - t0;
}
template <int MODE> void run(float* out, unsigned long long* cyc, int iters) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d: cycles per MFMA, waves 0..7:", MODE);
    for (int w = 0; w < 8; ++w) printf(" %.1f", (double)h[w] / (iters * 48.0));
    printf("\n");
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    const int iters = 2000;
    run<0>(out, cyc, iters); run<1>(out, cyc, iters); run<2>(out, cyc, iters); run<3>(out, cyc, iters);
    return 0;
}
