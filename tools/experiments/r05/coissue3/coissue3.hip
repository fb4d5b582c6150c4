// Round-5 microbenchmark: WHICH ingredient of the two-pass stream keeps the vector work from hiding behind the MFMAs?
// (tools/experiments/r05/k1_parts.sh: in the shipped K1 the matrix pipe's 76 us and the vector work's 83 us add up to the launch's
// 183 us, although r03/coissue shows 4 independent v_fma per MFMA slot running at the matrix rate.)
// One slot = 1 v_mfma_f32_32x32x16_bf16 + a pattern of vector / memory instructions; cycles per slot by s_memtime.
//   P0 3 independent v_fma                     P1 v_add -> v_log (dependent) + v_fma          P2 v_mul -> v_exp (dependent) + v_max
//   P3 v_exp + v_log + v_fma, independent      P4 P0 + one ds_read_b128 per slot (+ counted lgkmcnt wait)
//   P5 P0 + one 16-byte buffer-style global load per 3 slots (+ counted vmcnt wait)
//   P6 the whole K1 slot mix: 3 vector instructions (every 3rd slot holds the dependent pairs) + ds_read + load
//   P7 P1 with the dependent pair split over two slots (producer in slot s, consumer in slot s + 1)
// Build: hipcc --offload-arch=gfx950 -O3 coissue3.hip -o coissue3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MFMA_V(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define FMA(x, m, d) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(d))
#define ADD1(y, x) asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(y) : "v"(x))
#define LOG(y, x) asm volatile("v_log_f32 %0, %1" : "=v"(y) : "v"(x))
#define EXP(y, x) asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(x))
#define MUL(y, x, m) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y) : "v"(x), "v"(m))
#define MAX0(y, x) asm volatile("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x))

template <int P, bool MFMA>
__global__ void __launch_bounds__(512, 1) k(unsigned long long* out, float* sink, const float* gsrc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    float v[8], w[8];
    for (int e = 0; e < 8; ++e) { v[e] = 0.5f + 0.01f * e + 0.001f * threadIdx.x; w[e] = 0.25f; }
    const float m = 0.999f, d = 0.0003f;
    f32x4 ld[4], gl[4];
    for (int i = 0; i < 4; ++i) { ld[i] = f32x4{0, 0, 0, 0}; gl[i] = f32x4{0, 0, 0, 0}; }
    const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + lane;
    const f32x4* gp = reinterpret_cast<const f32x4*>(gsrc) + (blockIdx.x * 512 + threadIdx.x) % 4096;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            if (MFMA) MFMA_V(acc[j & 3], a, b);
            const int q = j & 7;
            if (P == 0) { FMA(v[q], m, d); FMA(v[(q + 3) & 7], m, d); FMA(v[(q + 5) & 7], m, d); }
            if (P == 1) { ADD1(w[q], v[q]); LOG(v[q], w[q]); FMA(v[(q + 3) & 7], m, d); }
            if (P == 2) { MUL(w[q], v[q], m); EXP(v[q], w[q]); MAX0(w[(q + 3) & 7], v[(q + 3) & 7]); }
            if (P == 3) { EXP(w[q], v[q]); LOG(w[(q + 3) & 7], v[(q + 3) & 7]); FMA(v[(q + 5) & 7], m, d); }
            if (P == 4) {
                FMA(v[q], m, d); FMA(v[(q + 3) & 7], m, d); FMA(v[(q + 5) & 7], m, d);
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[j & 3]) : "v"((unsigned)(size_t)lp & 0xffff), "n"(0));
                asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                asm volatile("" ::"v"(ld[(j + 1) & 3]));
            }
            if (P == 5) {
                FMA(v[q], m, d); FMA(v[(q + 3) & 7], m, d); FMA(v[(q + 5) & 7], m, d);
                if (j % 3 == 0) {
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gl[(j / 3) & 3]) : "v"(gp));
                    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    asm volatile("" ::"v"(gl[(j / 3 + 1) & 3]));
                }
            }
            if (P == 6) {
                if (j % 3 == 0) { MUL(w[q], v[q], m); EXP(v[q], w[q]); MAX0(w[(q + 3) & 7], v[(q + 3) & 7]); }
                if (j % 3 == 1) { ADD1(w[q], v[q]); LOG(v[q], w[q]); FMA(v[(q + 3) & 7], m, d); }
                if (j % 3 == 2) { FMA(v[q], m, d); FMA(v[(q + 3) & 7], m, d); FMA(v[(q + 5) & 7], m, d); }
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[j & 3]) : "v"((unsigned)(size_t)lp & 0xffff), "n"(0));
                if (j % 3 == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gl[(j / 3) & 3]) : "v"(gp));
                asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                if (j % 3 == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                asm volatile("" ::"v"(ld[(j + 1) & 3]), "v"(gl[(j / 3 + 1) & 3]));
            }
            if (P == 7) { LOG(v[(q + 7) & 7], w[(q + 7) & 7]); ADD1(w[q], v[(q + 2) & 7]); FMA(v[(q + 4) & 7], m, d); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15] + ld[i][0] + gl[i][1];
    for (int e = 0; e < 8; ++e) s += v[e] + w[e];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int P, bool MFMA>
void run(const char* name, int threads, unsigned long long* d_out, float* d_sink, float* d_src) {
    const int iters = 200, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<P, MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<P, MFMA>), dim3(blocks), dim3(threads), 100 * 1024, 0, d_out, d_sink, d_src, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    double sum = 0;
    for (int bI = 0; bI < blocks; ++bI)
        for (int w = 0; w < nw; ++w) sum += h[bI * 8 + w];
    printf("P%d %-52s %s  %d waves/SIMD: %6.1f cycles per slot\n", P, name, MFMA ? "MFMA + work" : "work alone ", nw / 4, sum / (blocks * nw) / (24.0 * iters));
}

int main() {
    unsigned long long* d_out; float *d_sink, *d_src;
    hipMalloc(&d_out, 256 * 8 * 8); hipMalloc(&d_sink, 4); hipMalloc(&d_src, 1 << 20);
    hipMemset(d_src, 0, 1 << 20);
#define BOTH(P, name) run<P, true>(name, 256, d_out, d_sink, d_src); run<P, false>(name, 256, d_out, d_sink, d_src); \
                      run<P, true>(name, 512, d_out, d_sink, d_src); run<P, false>(name, 512, d_out, d_sink, d_src)
    BOTH(0, "3 independent v_fma");
    BOTH(1, "v_add -> v_log (dependent) + v_fma");
    BOTH(2, "v_mul -> v_exp (dependent) + v_max");
    BOTH(3, "v_exp + v_log + v_fma, independent");
    BOTH(7, "v_log | v_add | v_fma, consumer one slot behind");
    BOTH(4, "3 v_fma + ds_read_b128 + lgkmcnt(3)");
    BOTH(5, "3 v_fma + global_load_dwordx4 every 3rd slot");
    BOTH(6, "the K1 mix: 3 vector + ds_read + load / 3");
    return 0;
}
