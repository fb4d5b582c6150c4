// Round-6 microbenchmark: the two-pass K1 WITHOUT any memory traffic runs its MFMAs and its vector work one after the other
// (tools/experiments/r06/k1_overlap.sh: MFMAs alone 1081 us, vector work alone 1122, both 2018 at 1 M points), although one MFMA +
// three vector instructions per slot run at the matrix rate in r05/coissue3.  What is different in the real stream?
//   NACC  accumulators the MFMAs rotate over (coissue3: 4 independent ones; K1 with TN = 1: 2, each MFMA depends on the one two back)
//   PAT   0: three independent v_fma per slot;  1: the softplus / split micro-steps of p2_engine.h (A A B B C -), ~3.3 per slot
//   AGPR  accumulators in AGPRs ("a") or VGPRs ("v")
// Cycles per slot by s_memtime, one wave per SIMD (256 threads) and two (512).  Build: hipcc --offload-arch=gfx950 -O3 coissue4.hip -o coissue4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MFMA_V(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define MFMA_A(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))

template <int NACC, int PAT, bool AGPR, bool MFMA, int REP = 1>
__global__ void __launch_bounds__(512, 1) k(unsigned long long* out, float* sink, const float* src, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    float z[32];                                   // stands for the other set's accumulators (read-only here)
    for (int e = 0; e < 32; ++e) z[e] = src[(threadIdx.x * 32 + e) & 4095] - 0.5f;
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 0.5f + 0.01f * e + 0.001f * threadIdx.x;
    const float m = 0.999f, d = 0.0003f;
    float ve[2], vm[2], vl[2], chk = 0.f;
    unsigned ph = 0, pl = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += REP) {
#pragma unroll
        for (int s = 0; s < 16 * REP; ++s) {       // REP x 16 k-steps of 6 slots, written out (code size: the I-cache experiment)
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                __builtin_amdgcn_sched_barrier(0);
                if (MFMA) {
                    if (AGPR) MFMA_A(acc[(6 * s + q) % NACC], a, b);
                    else MFMA_V(acc[(6 * s + q) % NACC], a, b);
                }
                if (PAT == 0) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(m), "v"(d));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(q + 3) & 7]) : "v"(m), "v"(d));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(q + 5) & 7]) : "v"(m), "v"(d));
                } else {
                    const float z0 = z[(2 * s) & 31], z1 = z[(2 * s + 1) & 31];
                    if (q == 0 || q == 1) {        // A: e = exp2(-|z| c), m = max(z, 0)
                        const float zz = q ? z1 : z0;
                        float t;
                        asm volatile("v_mul_f32_e64 %0, |%1|, %2" : "=v"(t) : "v"(zz), "s"(-144.26950408889634f));
                        asm volatile("v_exp_f32 %0, %1" : "=v"(ve[q]) : "v"(t));
                        asm volatile("v_max_f32 %0, 0, %1" : "=v"(vm[q]) : "v"(zz));
                    } else if (q == 2 || q == 3) { // B: l = log2(1 + e)
                        float t;
                        asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(t) : "v"(ve[q - 2]));
                        asm volatile("v_log_f32 %0, %1" : "=v"(vl[q - 2]) : "v"(t));
                    } else if (q == 4) {           // C: y = m + l k; hi / lo split of the pair
                        float y0, y1, h0, h1, l0, l1;
                        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y0) : "v"(vl[0]), "s"(0.0069314718f), "v"(vm[0]));
                        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y1) : "v"(vl[1]), "s"(0.0069314718f), "v"(vm[1]));
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(ph) : "v"(y0), "v"(y1));
                        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(h0) : "v"(ph));
                        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(h1) : "v"(ph));
                        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(l0) : "v"(y0), "v"(h0));
                        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(l1) : "v"(y1), "v"(h1));
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pl) : "v"(l0), "v"(l1));
                        asm volatile("" ::"v"(ph), "v"(pl));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < NACC; ++i) chk += acc[i][0] + acc[i][15];
    for (int e = 0; e < 8; ++e) chk += v[e];
    chk += __uint_as_float(ph) + __uint_as_float(pl);
    if (chk == 123.456f) sink[0] = chk;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NACC, int PAT, bool AGPR, bool MFMA, int REP = 1>
void run(const char* name, int threads, unsigned long long* d_out, float* d_sink, float* d_src) {
    const int iters = 128, blocks = 256;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NACC, PAT, AGPR, MFMA, REP>), dim3(blocks), dim3(threads), 0, 0, d_out, d_sink, d_src, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    double lo = 0, hi = 0;
    for (int bI = 0; bI < blocks; ++bI)
        for (int w = 0; w < nw; ++w) (w < 4 ? lo : hi) += h[bI * 8 + w];
    printf("%-30s REP=%-2d NACC=%d PAT=%d acc=%s %s %d wave/SIMD: waves 0-3 %6.1f", name, REP, NACC, PAT, AGPR ? "agpr" : "vgpr", MFMA ? "MFMA+work" : "work only",
           nw / 4, lo / (blocks * 4) / (96.0 * iters));
    if (nw > 4) printf("  waves 4-7 %6.1f", hi / (blocks * 4) / (96.0 * iters));
    printf(" cycles per slot\n");
}

int main() {
    unsigned long long* d_out; float *d_sink, *d_src;
    (void)hipMalloc(&d_out, 256 * 8 * 8); (void)hipMalloc(&d_sink, 4); (void)hipMalloc(&d_src, 4096 * 4);
    std::vector<float> hs(4096);
    for (int i = 0; i < 4096; ++i) hs[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f;
    (void)hipMemcpy(d_src, hs.data(), 4096 * 4, hipMemcpyHostToDevice);
#define ALL(NACC, PAT, AGPR, name)                                      \
    run<NACC, PAT, AGPR, true>(name, 256, d_out, d_sink, d_src);        \
    run<NACC, PAT, AGPR, true>(name, 512, d_out, d_sink, d_src)
    run<4, 1, false, false>("softplus micro-steps alone", 256, d_out, d_sink, d_src);
    run<4, 1, false, false>("softplus micro-steps alone", 512, d_out, d_sink, d_src);
    ALL(4, 0, false, "3 independent v_fma");
    ALL(2, 0, false, "3 independent v_fma");
    ALL(1, 0, false, "3 independent v_fma");
    ALL(4, 1, false, "softplus micro-steps");
    ALL(2, 1, false, "softplus micro-steps");
    ALL(1, 1, false, "softplus micro-steps");
    ALL(2, 1, true, "softplus micro-steps");
    ALL(2, 0, true, "3 independent v_fma");
#define REPS(R)                                                                          \
    run<2, 1, false, true, R>("softplus, body written out", 256, d_out, d_sink, d_src);  \
    run<2, 1, false, true, R>("softplus, body written out", 512, d_out, d_sink, d_src);  \
    run<2, 1, false, false, R>("softplus alone, written out", 512, d_out, d_sink, d_src)
    REPS(2); REPS(4); REPS(8); REPS(16); REPS(32); REPS(64);
    return 0;
}
