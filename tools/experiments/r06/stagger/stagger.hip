// Round-6 microbenchmark: do the SIMD partners of an 8-wave "r8" workgroup (csrc/r8_engine.h) have to be in the same phase?
// The shipped kernels hold waves w and w + 4 in lockstep: D (48 dependent MFMAs of one output tile, B fragments from LDS) . barrier .
// P (post phase: activation, hi / lo split, fragments -> LDS, planes -> HBM) . barrier -- the matrix pipe idles during P.
// Here: the same two phases, either in lockstep (MODE 0) or staggered (MODE 1..3: waves 0-3 run D while waves 4-7 run P and vice
// versa; MODE 2: s_setprio 1 inside D; MODE 3: s_setprio 1 inside P).  PW = weight of the post phase (1: ~K2 reverse, 2: ~K3).
// Output: cycles per (D + P) of a wave and the launch's wall time.  Ideal: 3072 cycles of MFMAs per SIMD and iteration.
// Build: hipcc --offload-arch=gfx950 -O3 stagger.hip -o stagger
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kFrag = 1024;
constexpr int kRegion = 16 * 2 * kFrag;          // 16 k-steps x (hi, lo)

__device__ __forceinline__ bf16x8 lds_read(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <int MODE>
__device__ __forceinline__ void dense(const bf16x8 (&Ah)[16], const bf16x8 (&Al)[16], unsigned fl, f32x16& acc) {
    bf16x8 bh[4], bl[4];
    if (MODE == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s) { bh[s] = lds_read(fl + (2 * s) * kFrag); bl[s] = lds_read(fl + (2 * s + 1) * kFrag); }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (s + 2 < 16) { bh[(s + 2) & 3] = lds_read(fl + (2 * (s + 2)) * kFrag); bl[(s + 2) & 3] = lds_read(fl + (2 * (s + 2) + 1) * kFrag); }
        if (s + 2 < 16) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if (s + 1 < 16) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(Al[s]), "v"(bh[s & 3]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(Ah[s]), "v"(bl[s & 3]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(Ah[s]), "v"(bh[s & 3]));
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    if (MODE == 2) __builtin_amdgcn_s_setprio(0);
}

// post phase: softplus(beta = 100) + sigma' + hi / lo split of 16 values; 4 LDS fragment stores, 2 + 2 plane stores, 2 operand loads
template <int MODE, int PW, int MEM>
__device__ __forceinline__ void post(f32x16& acc, unsigned out_lds, float* __restrict__ gst, const float* __restrict__ gld, int w, f32x4& g0, f32x4& g1) {
    if (MODE == 3) __builtin_amdgcn_s_setprio(1);
    if (MEM) asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0), "+v"(g1)::"memory");
    bf16x8 hi[2], lo[2];
    unsigned short sg[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float z = acc[r];
#pragma unroll
        for (int rep = 0; rep < PW; ++rep) {
            float e = __builtin_amdgcn_exp2f(-fabsf(z) * 144.26950408889634f);
            float y = fmaxf(z, 0.0f) + __builtin_amdgcn_logf(1.0f + e) * 0.006931471805599453f;
            float sgm = __builtin_amdgcn_rcpf(1.0f + e);
            sgm = z > 0.f ? sgm : 1.0f - sgm;
            z = y * (rep ? g0[r & 3] : 1.0f) + (rep ? sgm : 0.0f);
            if (rep == PW - 1) sg[r] = (unsigned short)(sgm * 65535.0f);
        }
        __bf16 h = (__bf16)z;
        hi[r >> 3][r & 7] = h;
        lo[r >> 3][r & 7] = (__bf16)(z - (float)h);
    }
    f32x4 s0, s1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s0[i] = __uint_as_float((unsigned)sg[2 * i] | ((unsigned)sg[2 * i + 1] << 16)) + g0[i] * 1e-30f;
        s1[i] = __uint_as_float((unsigned)sg[8 + 2 * i] | ((unsigned)sg[8 + 2 * i + 1] << 16)) + g1[i] * 1e-30f;
    }
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
        const unsigned a = out_lds + ((2 * w + sh) * 2) * kFrag;
        asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(hi[sh]) : "memory");
        asm volatile("ds_write_b128 %0, %1" ::"v"(a + kFrag), "v"(lo[sh]) : "memory");
        if (MEM) __builtin_nontemporal_store(hi[sh], reinterpret_cast<bf16x8*>(gst) + sh * 64);
    }
    if (MEM) {
        __builtin_nontemporal_store(s0, reinterpret_cast<f32x4*>(gst) + 128);
        __builtin_nontemporal_store(s1, reinterpret_cast<f32x4*>(gst) + 192);
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(g0) : "v"(gld));
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024 nt" : "=v"(g1) : "v"(gld));
    } else {
        asm volatile("" ::"v"(s0), "v"(s1));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if (MODE == 3) __builtin_amdgcn_s_setprio(0);
}

template <int MODE, int PW, int MEM>
__global__ void __launch_bounds__(512, 1) k(unsigned long long* out, float* gbuf, const bf16x8* wsrc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    bf16x8 Ah[16], Al[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) { Ah[s] = wsrc[(s * 8 + wave) * 64 + lane]; Al[s] = wsrc[((16 + s) * 8 + wave) * 64 + lane]; }
    for (int i = threadIdx.x; i < 2 * kRegion / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.0f;
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.01f * (r + lane);
    const unsigned base = (unsigned)(size_t)lds & 0xffff;
    float* gst = gbuf + ((size_t)blockIdx.x * 8 + wave) * 1024 + lane * 4;              // 4 KiB per wave
    const float* gld = gbuf + (size_t)(256 * 8 + blockIdx.x * 8 + wave) * 1024 + lane * 4;
    f32x4 g0 = f32x4{1, 1, 1, 1}, g1 = g0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned rin = base + (it & 1) * kRegion + lane * 16, rout = base + ((it + 1) & 1) * kRegion + lane * 16;
        if (MODE == 0 || wave < 4) {
            dense<MODE>(Ah, Al, rin, acc);
            __builtin_amdgcn_s_barrier();
            post<MODE, PW, MEM>(acc, rout, gst, gld, wave, g0, g1);
            __builtin_amdgcn_s_barrier();
        } else {
            post<MODE, PW, MEM>(acc, rin, gst, gld, wave, g0, g1);
            __builtin_amdgcn_s_barrier();
            dense<MODE>(Ah, Al, rout, acc);
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc[3] == 123.456f) gbuf[0] = acc[3];
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int PW, int MEM>
void run(const char* name, unsigned long long* d_out, float* d_g, bf16x8* d_w) {
    const int iters = 400, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, PW, MEM>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kRegion);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, PW, MEM>), dim3(blocks), dim3(512), 2 * kRegion, 0, d_out, d_g, d_w, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, PW, MEM>), dim3(blocks), dim3(512), 2 * kRegion, 0, d_out, d_g, d_w, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += h[b * 8 + w];
    printf("%-46s PW=%d MEM=%d : waves 0-3 %7.0f  waves 4-7 %7.0f cycles per (D + P);  MFMA floor 3072;  wall %.3f ms (%.0f ns per iteration)\n", name, PW, MEM,
           lo / (blocks * 4) / iters, hi / (blocks * 4) / iters, ms, ms * 1e6 / iters);
}

int main() {
    unsigned long long* d_out; float* d_g; bf16x8* d_w;
    hipMalloc(&d_out, 256 * 8 * 8); hipMalloc(&d_g, 2 * 256 * 8 * 4096); hipMalloc(&d_w, 32 * 8 * 1024);
    hipMemset(d_g, 0, 2 * 256 * 8 * 4096);
    std::vector<unsigned short> hw(32 * 8 * 512);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 24);      // bf16 near 0.01
    hipMemcpy(d_w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    run<0, 1, 1>("lockstep  D . bar . P . bar", d_out, d_g, d_w);
    run<1, 1, 1>("staggered (0-3: D|P, 4-7: P|D)", d_out, d_g, d_w);
    run<2, 1, 1>("staggered, s_setprio 1 in D", d_out, d_g, d_w);
    run<3, 1, 1>("staggered, s_setprio 1 in P", d_out, d_g, d_w);
    run<0, 2, 1>("lockstep  D . bar . P . bar", d_out, d_g, d_w);
    run<1, 2, 1>("staggered (0-3: D|P, 4-7: P|D)", d_out, d_g, d_w);
    run<2, 2, 1>("staggered, s_setprio 1 in D", d_out, d_g, d_w);
    run<3, 2, 1>("staggered, s_setprio 1 in P", d_out, d_g, d_w);
    run<0, 1, 0>("lockstep  D . bar . P . bar", d_out, d_g, d_w);
    run<1, 1, 0>("staggered (0-3: D|P, 4-7: P|D)", d_out, d_g, d_w);
    run<2, 1, 0>("staggered, s_setprio 1 in D", d_out, d_g, d_w);
    run<3, 1, 0>("staggered, s_setprio 1 in P", d_out, d_g, d_w);
    run<0, 2, 0>("lockstep  D . bar . P . bar", d_out, d_g, d_w);
    run<1, 2, 0>("staggered (0-3: D|P, 4-7: P|D)", d_out, d_g, d_w);
    run<2, 2, 0>("staggered, s_setprio 1 in D", d_out, d_g, d_w);
    run<3, 2, 0>("staggered, s_setprio 1 in P", d_out, d_g, d_w);
    return 0;
}
