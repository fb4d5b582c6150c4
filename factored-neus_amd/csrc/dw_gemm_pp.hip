// Weight-gradient GEMM over FRAGMENT PLANES:  C[o][i] += scale * sum_samples ( A[s][o] * B[s][i]  (+ A2[s][o] * B2[s][i]) )
//
// Replaces the dW = X^T Y products of torch autograd's addmm backward for the Linear layers of the SDF network
// (reference models/fields.py:86; second term = the double backward of SDFNetwork.gradient, fields.py:104-110:
// dW_l = zbar_l^T u_l + a_l^T adj_l).
//
// Operand format ("PP plane", fneus_pp.h): the chain kernels keep a 32-sample tile as MFMA B fragments -- 16 features x
// 32 samples = 1 KiB, lane (sample r, half h) holds 8 features -- and store exactly those fragments, hi part only (or hi
// and lo planes in the exact-gradient mode), one coalesced 16-byte store per lane.  A block = the fragments of one
// sample tile of one layer.  Here a block travels global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging
// registers, no VALU) into a 4-stage ring and is read back with ds_read_b64_tr_b16, which delivers the transposed
// operand (feature on the lane, 8 consecutive samples in the registers) straight in MFMA layout; the slot permutation
// of the plane makes those reads bank-conflict free.  The kernel is HBM-bound by construction (64 KiB of operands per
// 256 MFMAs): what matters is bytes in flight, not issue slots.
//
// One workgroup = 4 waves (one per SIMD, 2 x 2), output tile 256 x 256 (wave: 128 x 128 = 4 x 4 tiles of 32 x 32),
// split-K over sample tiles, fp32 atomics into the zero-initialised gradient.
#include <stdlib.h>
#include "fneus_common.h"
#include "fneus_kernels.h"
#include "fneus_pp.h"

namespace fneus {

struct GemmPPJob {
    const unsigned char *a_hi, *a_lo, *b_hi, *b_lo;
    const unsigned char *a2_hi, *a2_lo, *b2_hi, *b2_lo;
    uint32_t a_blk, b_blk, a2_blk, b2_blk;      // bytes between consecutive sample tiles (0: one constant block)
    uint16_t a_f0, b_f0, a2_f0, b2_f0;          // first fragment of the operand inside a block
    int32_t mt, nt;                             // 32-row / 32-column tiles of the product (<= 8 each)
    float* c;
    float* bias;
    int32_t ldc, m, n;
    float scale;
    int32_t wg_base, splits;
    int32_t n_tiles, pad_;                      // sample tiles of THIS product's planes; 0: the launch's
    const int32_t* n_dev;                       // or NULL: device-side count of the samples its planes hold this step
};
static_assert(sizeof(GemmPPJob) == sizeof(FneusGemmPPJob), "GemmPPJob must mirror FneusGemmPPJob");

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;
typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
typedef __attribute__((address_space(1))) const unsigned char glb_u8_t;

FN_DEV bf16x4 tr_read_pp(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(p));
}

// GP = 1: hi planes only (one MFMA per product);  GP = 3: hi + lo planes, hi*hi + hi*lo + lo*hi
template <int GP>
struct PPCfg {
    static constexpr int kParts = GP == 3 ? 4 : 2;                // A_hi, B_hi (, A_lo, B_lo)
    static constexpr int kStageBytes = kParts * 16 * 1024;
#ifndef FNEUS_GPP_READ_ALL_FIRST
#define FNEUS_GPP_READ_ALL_FIRST 1
#endif
#ifndef FNEUS_GPP_STAGES
#define FNEUS_GPP_STAGES 4
#endif
    static constexpr int kStages = GP == 3 ? 2 : FNEUS_GPP_STAGES;
    static constexpr int kDpw = kParts * 16 / 4;                  // LDS-DMA instructions per wave per stage
    static constexpr int kLds = kStages * kStageBytes;            // 128 KiB either way (default stage count)
};

#ifdef FNEUS_GPP_NT
#define FNEUS_GPP_NT_STR " nt"
#else
#define FNEUS_GPP_NT_STR ""
#endif

template <int N>
FN_DEV void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else static_assert(N == 0, "add the count");
}

// DET: the workgroup's partial tile (and bias partial) goes to `scratch` [workgroup][256 x 256 + 256] with plain stores instead
// of fp32 atomics into C; dw_gemm_pp_reduce_kernel then adds the partials of a product in split order: bit-reproducible
// gradients (FNEUS_DETERMINISTIC=1).  The atomics of the default mode land in arrival order.
constexpr int kDetTile = 256 * 256 + 256;

// sample tiles the products run over: all the planes hold, or -- n_samples_dev given -- those of the first *n_samples_dev
// samples (the planes of a launch that read its sample count from device memory, fneus_nerf_bg_fwd: tiles beyond it hold
// whatever an earlier step left there)
FN_DEV int live_tiles(int n_tiles, const int32_t* __restrict__ n_samples_dev) {
    if (n_samples_dev == nullptr) return n_tiles;
    const int t = (__builtin_amdgcn_readfirstlane(*n_samples_dev) + 31) >> 5;
    return t < n_tiles ? (t > 0 ? t : 0) : n_tiles;
}

template <int GP, bool DET = false>
__global__ void __launch_bounds__(256, 1) dw_gemm_pp_kernel(const GemmPPJob* __restrict__ jobs, int n_jobs, int n_tiles,
                                                            float* __restrict__ scratch = nullptr,
                                                            const int32_t* __restrict__ n_samples_dev = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    n_tiles = live_tiles(n_tiles, n_samples_dev);
    using Cfg = PPCfg<GP>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // job lookup and everything derived from it is wave-uniform: keep it in scalar registers (no struct copy: hipcc puts a
    // dynamically indexed copy into scratch, and scratch loads tick vmcnt -- every reload would drain the DMA ring)
    int ji = 0;
    for (int i = 1; i < n_jobs; ++i)
        if (jobs[i].wg_base <= (int)blockIdx.x) ji = i;
    const GemmPPJob* __restrict__ jp = jobs + ji;
    const int mt = jp->mt, nt = jp->nt;
    const int split = blockIdx.x - jp->wg_base;
    if (jp->n_tiles > 0) n_tiles = jp->n_tiles;                   // (products over other planes in the same launch: RefColor's 2 B rows,
    n_tiles = live_tiles(n_tiles, jp->n_dev);                     //  the background network's listed samples)
    const int per = (n_tiles + jp->splits - 1) / jp->splits;      // sample tiles [tb, te) of this workgroup
    const int tb = split * per;
    const int te = tb + per < n_tiles ? tb + per : n_tiles;
    if (tb >= te) return;
    const int nterm = jp->a2_hi != nullptr ? 2 : 1;
    const int nst = (te - tb) * nterm;

    // ---- LDS-DMA issue: slot i of a stage = fragment (i & 15) of part (i >> 4) (0 A_hi, 1 B_hi, 2 A_lo, 3 B_lo); wave w
    // issues slots w*kDpw .. w*kDpw + kDpw-1, i.e. always fragments of ONE part: its descriptor is fixed per term
    const int part = (wave * Cfg::kDpw) >> 4, k0 = (wave * Cfg::kDpw) & 15;
    const bool isb = part & 1, islo = part & 2;
    const unsigned char* base0 = isb ? (islo ? jp->b_lo : jp->b_hi) : (islo ? jp->a_lo : jp->a_hi);
    const unsigned char* base1 = isb ? (islo ? jp->b2_lo : jp->b2_hi) : (islo ? jp->a2_lo : jp->a2_hi);
    const uint32_t blk0 = isb ? jp->b_blk : jp->a_blk, blk1 = isb ? jp->b2_blk : jp->a2_blk;
    const int f00 = isb ? jp->b_f0 : jp->a_f0, f01 = isb ? jp->b2_f0 : jp->a2_f0;
    const int nf = 2 * (isb ? nt : mt);
    const unsigned lane16 = lane * 16;
    auto issue = [&](int s) {
        const int term = nterm == 2 ? (s & 1) : 0, tile = tb + (nterm == 2 ? (s >> 1) : s);
        const unsigned char* blockp = (term ? base1 : base0) + (size_t)tile * (term ? blk1 : blk0) +
                                      (size_t)(term ? f01 : f00) * kFragBytes;
        unsigned char* stage = smem + (s % Cfg::kStages) * Cfg::kStageBytes + (wave * Cfg::kDpw) * kFragBytes;
        const unsigned lds0 = (unsigned)(uintptr_t)(lds_u8_t*)stage;
#pragma unroll
        for (int d = 0; d < Cfg::kDpw; ++d) {
            const int k = k0 + d;
            const int kc = k < nf ? k : nf - 1;                      // unused slots re-read the last fragment (never consumed)
            const unsigned char* src = blockp + (size_t)kc * kFragBytes;         // wave-uniform; + lane * 16 in the instruction
            // Inline asm on purpose: hipcc counts a builtin LDS-DMA as a pending LDS write and puts s_waitcnt vmcnt(0) in
            // front of the next ds_read of ANY address -- that would drain the whole ring every stage.  The waits are
            // counted by hand below (wait_vm + s_barrier).  M0 = LDS destination, saved and restored for the compiler.
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" FNEUS_GPP_NT_STR "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(lane16), "s"(src), "s"(lds0 + d * kFragBytes)
                         : "memory");
        }
    };
    float* const cptr = jp->c;
    float* const bptr = jp->bias;
    const int ldc = jp->ldc, m_rows = jp->m, n_cols = jp->n;
    const float scale = jp->scale;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool do_bias = bptr != nullptr && wc == 0;

    // transposed-read addressing (fneus_pp.h): lane -> (fhalf, k-half hh, row q, chunk p)
    const int fhalf = (lane >> 4) & 1, hh = lane >> 5, q = (lane & 15) >> 2, p = lane & 3;
    const unsigned lane_off = fhalf * 1024 + (16 * hh + 2 * q + (p & 1)) * 16 + (p >> 1) * 8;
    const unsigned om0 = lane_off + (fhalf ? 128 : 0), om1 = lane_off + (fhalf ? 0 : 128);
    auto frag = [&](const unsigned char* part, int T, int kk) {
#ifdef FNEUS_GPP_NO_LDSREAD        // timing experiment: MFMAs on register constants, no LDS traffic
        bf16x8 c;
#pragma unroll
        for (int e = 0; e < 8; ++e) c[e] = (__bf16)(float)(T + kk + e + lane);
        return c;
#endif
        const bf16x4 v0 = tr_read_pp(part + om0 + T * 2048 + kk * 512);
        const bf16x4 v1 = tr_read_pp(part + om1 + T * 2048 + kk * 512);
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r[e] = v0[e];
            r[4 + e] = v1[e];
        }
        return r;
    };

    constexpr int P = Cfg::kStages - 1;          // prefetch distance in stages
#pragma unroll 1
    for (int s = 0; s < P && s < nst; ++s) issue(s);
    const bool row_active = 4 * wr < mt, col_active = 4 * wc < nt;
#pragma unroll 1
    for (int s = 0; s < nst; ++s) {
        // stage s has landed (own DMAs counted, everybody else's by the barrier); buffer (s-1) % kStages is free again
        if (nst - 1 - s >= P - 1) wait_vm<(P - 1) * Cfg::kDpw>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
#ifndef FNEUS_GPP_ISSUE_LATE
        if (s + P < nst) issue(s + P);
#endif
#ifndef FNEUS_GPP_NO_MFMA
        {   // (waves beyond mt / nt multiply never-consumed LDS contents: no branch around the MFMAs, hipcc otherwise
            //  shuttles the 256 accumulator registers between the two register files at every tile)
            const unsigned char* stage = smem + (s % Cfg::kStages) * Cfg::kStageBytes;
            const unsigned char* sA = stage;
            const unsigned char* sB = stage + 16 * 1024;
            const unsigned char* sAl = stage + 32 * 1024;
            const unsigned char* sBl = stage + 48 * 1024;
            const float bias_on = (do_bias && (nterm == 2 ? (s & 1) == 0 : true)) ? 1.0f : 0.0f;
#if FNEUS_GPP_READ_ALL_FIRST
            // every fragment of the stage is requested before the first product: one exposed LDS latency per stage, then 32
            // MFMAs back to back (64 fragment registers; the accumulators sit in the other half of the register file)
            bf16x8 fa[2][4], fb[2][4], fal[2][4], fbl[2][4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fb[kk][j] = frag(sB, 4 * wc + j, kk);
                    fa[kk][j] = frag(sA, 4 * wr + j, kk);
                    if constexpr (GP == 3) {
                        fbl[kk][j] = frag(sBl, 4 * wc + j, kk);
                        fal[kk][j] = frag(sAl, 4 * wr + j, kk);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = 0.0f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        t += (float)fa[kk][i][e];
                        if constexpr (GP == 3) t += (float)fal[kk][i][e];
                    }
                    bsum[i] = fmaf(bias_on, t, bsum[i]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (GP == 3) {
                            acc[i][j] = mfma32(fal[kk][i], fb[kk][j], acc[i][j]);
                            acc[i][j] = mfma32(fa[kk][i], fbl[kk][j], acc[i][j]);
                        }
                        acc[i][j] = mfma32(fa[kk][i], fb[kk][j], acc[i][j]);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 fb[4], fbl[4], fa[2], fal[2];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fb[j] = frag(sB, 4 * wc + j, kk);
                    if constexpr (GP == 3) fbl[j] = frag(sBl, 4 * wc + j, kk);
                }
                fa[0] = frag(sA, 4 * wr, kk);
                if constexpr (GP == 3) fal[0] = frag(sAl, 4 * wr, kk);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // next A tile is read ahead of this tile's MFMAs (distinct registers: LDS loads must not land in
                    // the operands of MFMAs that are still queued -- mlp_engine.h dense_ldsb)
                    if (i + 1 < 4) {
                        fa[(i + 1) & 1] = frag(sA, 4 * wr + i + 1, kk);
                        if constexpr (GP == 3) fal[(i + 1) & 1] = frag(sAl, 4 * wr + i + 1, kk);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    {   // column sums of A (first term) for the bias gradient, from the fragments already in registers
                        float t = 0.0f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            t += (float)fa[i & 1][e];
                            if constexpr (GP == 3) t += (float)fal[i & 1][e];
                        }
                        bsum[i] = fmaf(bias_on, t, bsum[i]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#ifdef FNEUS_GPP_READ_ONLY         // timing experiment: the LDS reads without the products
                        asm volatile("" :: "v"(fa[i & 1]), "v"(fb[j]));
                        continue;
#endif
                        if constexpr (GP == 3) {
                            acc[i][j] = mfma32(fal[i & 1], fb[j], acc[i][j]);
                            acc[i][j] = mfma32(fa[i & 1], fbl[j], acc[i][j]);
                        }
                        acc[i][j] = mfma32(fa[i & 1], fb[j], acc[i][j]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#endif
#endif
#ifdef FNEUS_GPP_ISSUE_LATE        // timing experiment: the next stage is requested after this stage's products
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + P < nst) issue(s + P);
#endif
    }
#ifdef FNEUS_GPP_NO_ATOMIC
    if (n_jobs > 0) return;
#endif
    if constexpr (DET) {
        float* part = scratch + (size_t)blockIdx.x * kDetTile;
        if (row_active && col_active) {
            const int cc = lane & 31;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = 32 * (4 * wc + j) + cc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[(32 * (4 * wr + i) + acc_row(r, hh)) * 256 + col] = acc[i][j][r];
                }
        }
        if (do_bias && row_active) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float t = bsum[i] + xor32(bsum[i]);
                if (lane < 32) part[256 * 256 + 32 * (4 * wr + i) + (lane & 31)] = t;
            }
        }
        return;
    }
    // ---- epilogue: fp32 atomics, two 128-byte row segments per wave instruction
    if (row_active && col_active) {
        const int cc = lane & 31;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = 32 * (4 * wc + j) + cc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * (4 * wr + i) + acc_row(r, hh);
                    if (row < m_rows && col < n_cols) atomicAdd(cptr + (size_t)row * ldc + col, scale * acc[i][j][r]);
                }
            }
    }
    if (do_bias && row_active) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = bsum[i] + xor32(bsum[i]);
            const int row = 32 * (4 * wr + i) + (lane & 31);
            if (lane < 32 && row < m_rows) atomicAdd(bptr + row, t);
        }
    }
}

// C[row][col] += scale * (partials of the product's workgroups in split order); bias[row] += the same of the bias partials
__global__ void __launch_bounds__(256) dw_gemm_pp_reduce_kernel(const GemmPPJob* __restrict__ jobs, int n_tiles,
                                                                const float* __restrict__ scratch,
                                                                const int32_t* __restrict__ n_samples_dev) {
    n_tiles = live_tiles(n_tiles, n_samples_dev);
    if (n_tiles <= 0) return;
    const GemmPPJob* __restrict__ jp = jobs + blockIdx.y;
    if (jp->n_tiles > 0) n_tiles = jp->n_tiles;
    n_tiles = live_tiles(n_tiles, jp->n_dev);
    if (n_tiles <= 0) return;
    const int m = jp->m, n = jp->n, splits = jp->splits, base = jp->wg_base;
    const int per = (n_tiles + splits - 1) / splits;
    const int live = (n_tiles + per - 1) / per;          // workgroups of the product that had sample tiles (the others wrote nothing)
    const int total = m * n + (jp->bias != nullptr ? m : 0);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const bool is_bias = idx >= m * n;
        const int row = is_bias ? idx - m * n : idx / n, col = is_bias ? 0 : idx % n;
        const size_t off = is_bias ? (size_t)256 * 256 + row : (size_t)row * 256 + col;
        float sum = 0.0f;
        for (int s = 0; s < live; ++s) sum += scratch[(size_t)(base + s) * kDetTile + off];
        if (is_bias) atomicAdd(jp->bias + row, sum);
        else atomicAdd(jp->c + (size_t)row * jp->ldc + col, jp->scale * sum);     // (one addend per element and launch: order-free)
    }
}

}  // namespace fneus

using namespace fneus;

extern "C" int fneus_dw_gemm_pp_det(const void* jobs_dev, int n_jobs, int n_wgs, long n_sample_tiles, const int32_t* n_samples_dev,
                                    int gprec, float* scratch, long scratch_floats, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_jobs <= 0 || n_wgs <= 0 || n_sample_tiles <= 0) return 0;
    if (scratch == nullptr || scratch_floats < (long)n_wgs * kDetTile) {
        fneus::set_last_error("fneus_dw_gemm_pp_det: scratch must hold n_wgs x (256 x 256 + 256) floats");
        return -2;
    }
    const GemmPPJob* jobs = reinterpret_cast<const GemmPPJob*>(jobs_dev);
    static bool attr_set = false;
    if (!attr_set) {
        allow_big_lds(dw_gemm_pp_kernel<1, true>);
        allow_big_lds(dw_gemm_pp_kernel<3, true>);
        attr_set = true;
    }
    if (gprec == 1)
        hipLaunchKernelGGL((dw_gemm_pp_kernel<1, true>), dim3(n_wgs), dim3(256), PPCfg<1>::kLds, stream, jobs, n_jobs, (int)n_sample_tiles, scratch, n_samples_dev);
    else if (gprec == 3)
        hipLaunchKernelGGL((dw_gemm_pp_kernel<3, true>), dim3(n_wgs), dim3(256), PPCfg<3>::kLds, stream, jobs, n_jobs, (int)n_sample_tiles, scratch, n_samples_dev);
    else
        return -2;
    hipLaunchKernelGGL(dw_gemm_pp_reduce_kernel, dim3(64, n_jobs), dim3(256), 0, stream, jobs, (int)n_sample_tiles, scratch, n_samples_dev);
    return fneus::launch_status();
}

extern "C" int fneus_dw_gemm_pp(const void* jobs_dev, int n_jobs, int n_wgs, long n_sample_tiles, const int32_t* n_samples_dev,
                                int gprec, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_jobs <= 0 || n_wgs <= 0 || n_sample_tiles <= 0) return 0;
    const GemmPPJob* jobs = reinterpret_cast<const GemmPPJob*>(jobs_dev);
    static bool attr_set = false;
    if (!attr_set) {
        allow_big_lds(dw_gemm_pp_kernel<1, false>);
        allow_big_lds(dw_gemm_pp_kernel<3, false>);
        attr_set = true;
    }
    if (gprec == 1)
        hipLaunchKernelGGL((dw_gemm_pp_kernel<1, false>), dim3(n_wgs), dim3(256), PPCfg<1>::kLds, stream, jobs, n_jobs, (int)n_sample_tiles, (float*)nullptr, n_samples_dev);
    else if (gprec == 3)
        hipLaunchKernelGGL((dw_gemm_pp_kernel<3, false>), dim3(n_wgs), dim3(256), PPCfg<3>::kLds, stream, jobs, n_jobs, (int)n_sample_tiles, (float*)nullptr, n_samples_dev);
    else
        return -2;
    return fneus::launch_status();
}
