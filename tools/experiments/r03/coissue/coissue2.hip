// Round-3 microbenchmark 2: the REAL activation sequence (softplus beta = 100, hi / lo bf16 split) interleaved with MFMAs
// by sched_group_barrier, as p2_engine.h does it -- without LDS / global traffic.  One wave per SIMD (256 threads, 512 registers).
// Output: cycles per k-step of 12 MFMAs (384 = the matrix pipe's rate) for V activated values per k-step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float softplus100(float z) {
#if TRANS
    float e = __builtin_amdgcn_exp2f(-fabsf(z) * (100.0f * 1.4426950408889634f));
    return fmaxf(z, 0.0f) + __builtin_amdgcn_logf(1.0f + e) * (0.6931471805599453f / 100.0f);
#else
    return fmaxf(z, 0.0f) * 1.01f;
#endif
}

template <int V, int VPM, bool MFMA>
__global__ void __launch_bounds__(256, 1) k(unsigned long long* out, float* sink, const float* in, int iters) {
    f32x16 accM[4], accV[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) { accM[i][r] = 0.f; accV[i][r] = in[(i * 16 + r) * 256 + threadIdx.x]; }
    bf16x8 a[2], al[2], b[2], bl[2];
    for (int e = 0; e < 8; ++e)
        for (int q = 0; q < 2; ++q) {
            a[q][e] = (__bf16)in[threadIdx.x + e + q]; al[q][e] = (__bf16)in[threadIdx.x + 9 + e + q];
            b[q][e] = (__bf16)in[threadIdx.x + 20 + e + q]; bl[q][e] = (__bf16)in[threadIdx.x + 31 + e + q];
        }
    float chk = 0.f;
    bf16x8 ph, pl;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            if (MFMA) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) accM[i * 2 + hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], b[hb], accM[i * 2 + hb], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) accM[i * 2 + hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bl[hb], accM[i * 2 + hb], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) accM[i * 2 + hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[hb], accM[i * 2 + hb], 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int idx = (s * V + v) & 63;
                const int e = (s * V + v) & 7;
                float y = softplus100(accV[idx >> 4][idx & 15]);
                __bf16 hi = (__bf16)y, lo = (__bf16)(y - (float)hi);
                ph[e] = hi;
                pl[e] = lo;
                if (e == 7) {
                    // consume the fragment halves (stand-in for the two ds_write_b128)
                    asm volatile("" :: "v"(ph), "v"(pl));
                }
                accV[idx >> 4][idx & 15] = y + 0.25f;     // keep the values changing
            }
            if (MFMA) {
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (V > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 4; ++i) chk += accM[i][0] + accM[i][15] + accV[i][3];
    if (chk == 123.456f) sink[0] = chk;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V, int VPM, bool MFMA>
void run(unsigned long long* d_out, float* d_sink, float* d_in) {
    const int iters = 100, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<V, VPM, MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<V, VPM, MFMA>), dim3(blocks), dim3(256), 100 * 1024, 0, d_out, d_sink, d_in, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto x : h) s += x;
    printf("TRANS=%d V=%d values per k-step, %d vector instr per MFMA in the pattern, MFMA %s: %7.1f cycles per k-step (12 MFMAs = 384)\n", TRANS, V, VPM,
           MFMA ? "on " : "off", s / h.size() / (16.0 * iters));
}

int main() {
    unsigned long long* d_out;
    float *d_sink, *d_in;
    (void)hipMalloc(&d_out, 256 * 4 * 8);
    (void)hipMalloc(&d_sink, 4);
    (void)hipMalloc(&d_in, 64 * 256 * 4 + 4096);
    std::vector<float> h(64 * 256 + 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)((i * 7919) % 200) - 1.0f;
    (void)hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 3, true>(d_out, d_sink, d_in);
    run<2, 2, true>(d_out, d_sink, d_in);
    run<4, 3, true>(d_out, d_sink, d_in);
    run<4, 4, true>(d_out, d_sink, d_in);
    run<6, 5, true>(d_out, d_sink, d_in);
    run<2, 2, false>(d_out, d_sink, d_in);
    run<4, 3, false>(d_out, d_sink, d_in);
    run<6, 5, false>(d_out, d_sink, d_in);
    return 0;
}
