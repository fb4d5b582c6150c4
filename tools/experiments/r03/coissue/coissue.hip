// Round-3 microbenchmark: how much vector work hides beside v_mfma_f32_32x32x16_bf16 on one SIMD of gfx950?
//   intra: every wave runs { 1 MFMA, F vector fillers } repeated; 1 or 2 waves per SIMD
//   inter: waves 0-3 of a 512-thread workgroup run MFMAs only, waves 4-7 (their SIMD partners) run fillers only
// Accumulators in VGPRs ("v") or AGPRs ("a").  Fillers: v_fma_f32 (FILL = 0), v_exp_f32 every 4th (FILL = 1).
// Build: hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue ; run on the GPU box.  Output: cycles per MFMA (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MFMA_V(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define MFMA_A(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define FMA(x, m, d) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(d))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

template <int F, int FILL>
__device__ __forceinline__ void fillers(float (&v)[8], float m, float d, int& k) {
#pragma unroll
    for (int f = 0; f < F; ++f) {
        if (FILL == 1 && (f & 3) == 3) EXP(v[k & 7]);
        else FMA(v[k & 7], m, d);
        ++k;
    }
}

// MODE 0: intra-wave interleave; MODE 1: inter-wave (waves >= 4: fillers only, waves < 4: MFMAs only); MODE 2: fillers only in
// every wave (the vector work alone); ACC 0: VGPR accumulators, 1: AGPR
template <int F, int FILL, int MODE, int ACC>
__global__ void __launch_bounds__(512, 2) k(unsigned long long* out, float* sink, int iters) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 0.5f + 0.01f * e + 0.001f * threadIdx.x;
    const float m = 0.999f, d = 0.0003f;
    int kk = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const bool do_mfma = MODE == 0 || (MODE == 1 && wave < 4);
    const bool do_fill = MODE == 0 || MODE == 2 || (MODE == 1 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (do_mfma) {
                if (ACC) MFMA_A(acc[j & 3], a, b);
                else MFMA_V(acc[j & 3], a, b);
            }
            if (do_fill) fillers<F, FILL>(v, m, d, kk);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    for (int e = 0; e < 8; ++e) s += v[e];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int F, int FILL, int MODE, int ACC>
void run(const char* name, int threads, unsigned long long* d_out, float* d_sink) {
    const int iters = 200, blocks = 256;
    hipLaunchKernelGGL((k<F, FILL, MODE, ACC>), dim3(blocks), dim3(threads), 100 * 1024, 0, d_out, d_sink, iters);
    hipLaunchKernelGGL((k<F, FILL, MODE, ACC>), dim3(blocks), dim3(threads), 100 * 1024, 0, d_out, d_sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    double lo = 0, hi = 0;
    int nlo = 0, nhi = 0;
    for (int bI = 0; bI < blocks; ++bI)
        for (int w = 0; w < nw; ++w) {
            if (w < 4) { lo += h[bI * 8 + w]; ++nlo; } else { hi += h[bI * 8 + w]; ++nhi; }
        }
    const double n_mfma = 16.0 * iters;
    printf("%-44s F=%d fill=%d acc=%s threads=%d : waves 0-3 %7.1f cyc per 16-MFMA-slot/16", name, F, FILL, ACC ? "agpr" : "vgpr", threads, lo / nlo / n_mfma);
    if (nhi) printf("   waves 4-7 %7.1f", hi / nhi / n_mfma);
    printf("\n");
}

int main() {
    unsigned long long* d_out;
    float* d_sink;
    hipMalloc(&d_out, 256 * 8 * 8);
    hipMalloc(&d_sink, 4);
    for (auto* kern : {(const void*)0}) (void)kern;
#define ALLOW(...) hipFuncSetAttribute(reinterpret_cast<const void*>(k<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
#define RUN(name, F, FILL, MODE, ACC, T) ALLOW(F, FILL, MODE, ACC); run<F, FILL, MODE, ACC>(name, T, d_out, d_sink)
    printf("cycles per MFMA slot (the loop issues 1 MFMA + F fillers per slot); 32.0 = the matrix pipe's rate for one stream\n");
    RUN("MFMA only, 1 wave/SIMD", 0, 0, 0, 0, 256);
    RUN("MFMA only, 2 waves/SIMD (both stream MFMAs)", 0, 0, 0, 0, 512);
    RUN("fillers only (4/slot), 1 wave/SIMD", 4, 0, 2, 0, 256);
    RUN("fillers only (4/slot), 2 waves/SIMD", 4, 0, 2, 0, 512);
    RUN("fillers only (4/slot, exp), 1 wave/SIMD", 4, 1, 2, 0, 256);
    RUN("fillers only (4/slot, exp), 2 waves/SIMD", 4, 1, 2, 0, 512);
    RUN("intra 1 wave/SIMD", 2, 0, 0, 0, 256);
    RUN("intra 1 wave/SIMD", 4, 0, 0, 0, 256);
    RUN("intra 1 wave/SIMD", 6, 0, 0, 0, 256);
    RUN("intra 1 wave/SIMD", 8, 0, 0, 0, 256);
    RUN("intra 1 wave/SIMD", 4, 1, 0, 0, 256);
    RUN("intra 1 wave/SIMD", 4, 0, 0, 1, 256);
    RUN("intra 1 wave/SIMD", 6, 0, 0, 1, 256);
    RUN("intra 2 waves/SIMD", 2, 0, 0, 0, 512);
    RUN("intra 2 waves/SIMD", 4, 0, 0, 0, 512);
    RUN("intra 2 waves/SIMD", 6, 0, 0, 0, 512);
    RUN("intra 2 waves/SIMD", 8, 0, 0, 0, 512);
    RUN("intra 2 waves/SIMD", 4, 1, 0, 0, 512);
    RUN("intra 2 waves/SIMD", 4, 0, 0, 1, 512);
    RUN("intra 2 waves/SIMD", 8, 0, 0, 1, 512);
    RUN("inter: waves 0-3 MFMA, 4-7 fillers", 2, 0, 1, 0, 512);
    RUN("inter: waves 0-3 MFMA, 4-7 fillers", 4, 0, 1, 0, 512);
    RUN("inter: waves 0-3 MFMA, 4-7 fillers", 8, 0, 1, 0, 512);
    RUN("inter: waves 0-3 MFMA, 4-7 fillers", 4, 1, 1, 0, 512);
    RUN("inter: waves 0-3 MFMA, 4-7 fillers", 4, 0, 1, 1, 512);
    RUN("inter: waves 0-3 MFMA, 4-7 fillers", 8, 0, 1, 1, 512);
    return 0;
}
