// Weight-gradient GEMM:  C[m][n] += scale * sum_samples ( A[s][m] * B[s][n]  (+ A2[s][m] * B2[s][n]) )
//
// Replaces the dW = X^T Y products of torch autograd's addmm/mm backward for every Linear on the path (reference
// models/fields.py:86, 168; the second segment carries the double-backward term of SURVEY.md Appendix A:
// dW_l = zbar_l^T u_l + a_l^T adj_l).  Operands are the row-major bf16 stash planes written by the chain kernels, so
// the contraction runs over the ROW index of both operands ("TN" GEMM): tiles are staged in LDS and read back with
// ds_read_b64_tr_b16 (hardware transpose) straight into MFMA operand layout.  Split-K over samples with fp32 atomics.
//
// One workgroup = 512 threads = 2x4 waves, output tile 256x256 (wave: 128x64 = 4x2 MFMA 32x32x16 tiles), 32-sample
// LDS blocks double buffered (147 KiB in parity mode).
#include <stdlib.h>
#include "fneus_common.h"
#include "fneus_kernels.h"

namespace fneus {

struct GemmJob {
    const __bf16 *a_hi, *a_lo, *b_hi, *b_lo;
    const __bf16 *a2_hi, *a2_lo, *b2_hi, *b2_lo;
    float* c;
    float* bias;
    int lda, ldb, lda2, ldb2, ldc;
    int m, n;
    int a_w;       // readable width (elements) of A / A2 rows starting at the given pointer (multiple of 8)
    int a2_mode;   // 0: A2 from memory, 1: A2 is the implicit matrix with column 0 == 1 (dW row of the sdf output)
    float scale;
    int tile_base;
    int b_w;       // readable width of B / B2 rows
};
static_assert(sizeof(GemmJob) == sizeof(FneusGemmJob), "GemmJob must mirror FneusGemmJob");

constexpr int KB = 32;               // samples per LDS block
constexpr int TM = 256, TN = 256;    // output tile per workgroup
constexpr int ROWB = 576;            // LDS row stride in bytes: 512 data + 64 pad -> conflict-free transposed reads
constexpr int PLANEB = KB * ROWB;    // 18 432 bytes per staged plane tile
constexpr int NTHREADS = 512;        // 8 waves: 2 (M) x 4 (N), wave tile 128 x 64
constexpr int CHUNKS = (KB * 512 / 16) / NTHREADS;   // 16-byte chunks per thread per plane = 2

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

FN_DEV bf16x4 tr_read(const unsigned char* lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds_addr));
}

// global -> registers: this thread's chunks of a [KB][256] bf16 plane tile (zero beyond s_end / beyond w columns)
FN_DEV void load_plane(u32x4 (&r)[CHUNKS], const __bf16* __restrict__ g, int ld, int w, long s0, long s_end, int c0, int tid) {
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const int idx = tid + NTHREADS * i;
        const int row = idx >> 5, cc = idx & 31;
        const long s = s0 + row;
        const int col = c0 + cc * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (g != nullptr && s < s_end && col < w) v = *reinterpret_cast<const u32x4*>(g + s * ld + col);
        r[i] = v;
    }
}

FN_DEV void load_e0(u32x4 (&r)[CHUNKS], long s0, long s_end, int c0, int tid) {
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const int idx = tid + NTHREADS * i;
        const int row = idx >> 5, cc = idx & 31;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (cc == 0 && c0 == 0 && (s0 + row) < s_end) v[0] = 0x3F80u;   // bf16(1.0) in column 0
        r[i] = v;
    }
}

FN_DEV void store_plane(unsigned char* lds, const u32x4 (&r)[CHUNKS], int tid) {
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const int idx = tid + NTHREADS * i;
        const int row = idx >> 5, cc = idx & 31;
        *reinterpret_cast<u32x4*>(lds + row * ROWB + cc * 16) = r[i];
    }
}

// MFMA operand fragment (32 "rows" starting at column f0 of the tile, k-step kk) from an LDS plane tile
FN_DEV bf16x8 frag_from_lds(const unsigned char* tile, int f0, int kk, int lane) {
    const int hh = lane >> 5, fhalf = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    const unsigned char* a = tile + (kk * 16 + 8 * hh + q) * ROWB + (f0 + 16 * fhalf + 4 * p) * 2;
    const bf16x4 v0 = tr_read(a);
    const bf16x4 v1 = tr_read(a + 4 * ROWB);
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = v0[e];
        r[4 + e] = v1[e];
    }
    return r;
}

// One workgroup: 256x256 output tile, one K chunk.  Blocks of 32 samples are software pipelined:
//   global loads of block b+1 (into registers) | MFMAs of block b from LDS buffer b&1 | registers -> buffer (b+1)&1 | barrier
template <int PREC>
__global__ void __launch_bounds__(NTHREADS) dw_gemm_kernel(const GemmJob* __restrict__ jobs, int n_jobs, long N, int kchunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NPL = PREC == 3 ? 4 : 2;              // staged planes per buffer: A_hi, B_hi (, A_lo, B_lo)
    constexpr int BUFB = NPL * PLANEB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    int ji = 0;
    for (int i = 1; i < n_jobs; ++i)
        if (jobs[i].tile_base <= (int)blockIdx.x) ji = i;
    const GemmJob jb = jobs[ji];
    const int tiles_n = (jb.n + TN - 1) / TN;
    const int tl = blockIdx.x - jb.tile_base;
    const int tm = tl / tiles_n, tn = tl % tiles_n;
    const long s_begin = (long)blockIdx.y * kchunk;
    const long s_end = (s_begin + kchunk < N) ? s_begin + kchunk : N;
    if (s_begin >= s_end) return;
    const int nb1 = (int)((s_end - s_begin + KB - 1) / KB);          // blocks per segment
    const bool has2 = jb.b2_hi != nullptr;
    const int nb = has2 ? 2 * nb1 : nb1;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float bias_acc = 0.0f;
    const bool do_bias = (jb.bias != nullptr) && (tn == 0) && (tid < TM);

    struct Stage {
        u32x4 ah[CHUNKS], bh[CHUNKS], al[CHUNKS], bl[CHUNKS];
    };
    Stage S0;
    auto fetch = [&](int b, Stage& S) {
        const int seg = (b >= nb1) ? 1 : 0;
        const long s0 = s_begin + (long)(b - seg * nb1) * KB;
        const __bf16* ah = seg ? jb.a2_hi : jb.a_hi;
        const __bf16* al = seg ? jb.a2_lo : jb.a_lo;
        const __bf16* bh = seg ? jb.b2_hi : jb.b_hi;
        const __bf16* bl = seg ? jb.b2_lo : jb.b_lo;
        const int lda = seg ? jb.lda2 : jb.lda, ldb = seg ? jb.ldb2 : jb.ldb;
        const bool e0 = seg && jb.a2_mode == 1;
        if (e0) load_e0(S.ah, s0, s_end, tm * TM, tid);
        else load_plane(S.ah, ah, lda, jb.a_w, s0, s_end, tm * TM, tid);
        load_plane(S.bh, bh, ldb, jb.b_w, s0, s_end, tn * TN, tid);
        if constexpr (PREC == 3) {
            if (e0) load_plane(S.al, nullptr, 0, 0, s0, s_end, 0, tid);
            else load_plane(S.al, al, lda, jb.a_w, s0, s_end, tm * TM, tid);
            load_plane(S.bl, bl, ldb, jb.b_w, s0, s_end, tn * TN, tid);
        }
    };
    auto commit = [&](int buf, const Stage& S) {
        unsigned char* base = smem + buf * BUFB;
        store_plane(base, S.ah, tid);
        store_plane(base + PLANEB, S.bh, tid);
        if constexpr (PREC == 3) {
            store_plane(base + 2 * PLANEB, S.al, tid);
            store_plane(base + 3 * PLANEB, S.bl, tid);
        }
    };

    // Operand pipeline, ONE register set: block b+1 is written to LDS right AFTER the barrier that opens iteration b and
    // block b+2 is requested at once, so a block's loads have a whole iteration (products + barrier) to land instead of
    // the products alone.  (History, N = 65 536 SDF launch, parity mode: written before the barrier 706 us; this order
    // 648 us; with a second register set, i.e. two blocks in flight, 732 us -- it spills 50 registers -- and 265 vs 279 us
    // in bf16 mode where it does not; stream alone 476 us, products alone 391 us: tools/experiments/README.md.)
    auto step = [&](int b, Stage& S) {
        __syncthreads();             // buffer b&1 is complete; every wave is done with the products of block b-1
#ifndef FNEUS_DBG_GEMM_NO_LOAD      // timing experiments only (tools/experiments): the compute loop without its operand stream
        if (b + 1 < nb) {
#ifndef FNEUS_DBG_GEMM_NO_COMMIT
            commit((b + 1) & 1, S);
#endif
#ifndef FNEUS_DBG_GEMM_NO_FETCH
            if (b + 2 < nb) fetch(b + 2, S);
#endif
        }
#endif
        const unsigned char* sA_hi = smem + (b & 1) * BUFB;
        const unsigned char* sB_hi = sA_hi + PLANEB;
        const unsigned char* sA_lo = sA_hi + 2 * PLANEB;
        const unsigned char* sB_lo = sA_hi + 3 * PLANEB;
#ifdef FNEUS_DBG_GEMM_NO_BIAS
        if (false) {
#else
        if (do_bias && b < nb1) {
#endif
            float s = 0.0f;
            for (int row = 0; row < KB; ++row) {
                s += (float)*reinterpret_cast<const __bf16*>(sA_hi + row * ROWB + tid * 2);
                if constexpr (PREC == 3) s += (float)*reinterpret_cast<const __bf16*>(sA_lo + row * ROWB + tid * 2);
            }
            bias_acc += s;
        }
#ifndef FNEUS_DBG_GEMM_NO_MFMA      // timing experiments only: the operand stream (global -> registers -> LDS) without the products
#pragma unroll
        for (int kk = 0; kk < KB / 16; ++kk) {
            // fragment reads run one A tile ahead of the products (hipcc otherwise batches "all reads, wait, all MFMAs",
            // and with both waves of a SIMD released by the same barrier their read phases coincide); the scheduling
            // barriers keep that order.  HAZARD as in mlp_engine.h dense_ldsb(): an LDS read must not land in the
            // registers of an MFMA still queued -- the prefetch targets the OTHER fragment set, last used one stage ago.
            bf16x8 fb_h[2], fb_l[2], fa_h[2], fa_l[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fb_h[j] = frag_from_lds(sB_hi, wc * 64 + j * 32, kk, lane);
                if constexpr (PREC == 3) fb_l[j] = frag_from_lds(sB_lo, wc * 64 + j * 32, kk, lane);
            }
            fa_h[0] = frag_from_lds(sA_hi, wr * 128, kk, lane);
            if constexpr (PREC == 3) fa_l[0] = frag_from_lds(sA_lo, wr * 128, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i + 1 < 4) {
                    fa_h[(i + 1) & 1] = frag_from_lds(sA_hi, wr * 128 + (i + 1) * 32, kk, lane);
                    if constexpr (PREC == 3) fa_l[(i + 1) & 1] = frag_from_lds(sA_lo, wr * 128 + (i + 1) * 32, kk, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (PREC == 3) {
                        acc[i][j] = mfma32(fa_l[i & 1], fb_h[j], acc[i][j]);
                        acc[i][j] = mfma32(fa_h[i & 1], fb_l[j], acc[i][j]);
                    }
                    acc[i][j] = mfma32(fa_h[i & 1], fb_h[j], acc[i][j]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
    };
    fetch(0, S0);
    commit(0, S0);
    if (nb > 1) fetch(1, S0);
    for (int b = 0; b < nb; ++b) step(b, S0);
    // epilogue: fp32 atomics (two 128-byte row segments per wave instruction)
    const int hh = lane >> 5, cc = lane & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = tn * TN + wc * 64 + j * 32 + cc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * TM + wr * 128 + i * 32 + acc_row(r, hh);
                if (row < jb.m && col < jb.n) atomicAdd(jb.c + (size_t)row * jb.ldc + col, jb.scale * acc[i][j][r]);
            }
        }
    if (do_bias) {
        const int row = tm * TM + tid;
        if (row < jb.m) atomicAdd(jb.bias + row, bias_acc);
    }
}

}  // namespace fneus

using namespace fneus;

extern "C" int fneus_dw_gemm(const void* jobs_dev, int n_jobs, int n_tiles, long n_samples, int prec,
                             fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_tiles <= 0 || n_samples <= 0) return 0;
    // split-K so that the grid holds ~256 workgroups (one 147 KiB workgroup per CU in parity mode)
    int split = 256 / n_tiles;
    // ... but a workgroup should own at least FNEUS_GEMM_MIN_BLOCKS blocks: its epilogue is 65 536 atomics whatever it summed
    // (the RefColor launch has 1024 samples; default measured in tools/experiments/README.md)
    static const int min_blocks = getenv("FNEUS_GEMM_MIN_BLOCKS") ? atoi(getenv("FNEUS_GEMM_MIN_BLOCKS")) : 4;   // RefColor launch: 50 us at 1 / 2, 38 us at 4, 41 us at 8, 60 us at 16
    const long max_split = n_samples / ((long)KB * min_blocks);
    if (split > max_split) split = (int)max_split;
    if (split < 1) split = 1;
    long kchunk = (n_samples + split - 1) / split;
    kchunk = ((kchunk + KB - 1) / KB) * KB;
    split = (int)((n_samples + kchunk - 1) / kchunk);
    dim3 grid(n_tiles, split), blk(NTHREADS);
    const GemmJob* jobs = reinterpret_cast<const GemmJob*>(jobs_dev);
    static bool attr_set = false;
    if (!attr_set) {
        allow_big_lds(dw_gemm_kernel<3>);
        allow_big_lds(dw_gemm_kernel<1>);
        attr_set = true;
    }
    if (prec == 3)
        hipLaunchKernelGGL(dw_gemm_kernel<3>, grid, blk, 2 * 4 * PLANEB, stream, jobs, n_jobs, n_samples, (int)kchunk);
    else if (prec == 1)
        hipLaunchKernelGGL(dw_gemm_kernel<1>, grid, blk, 2 * 2 * PLANEB, stream, jobs, n_jobs, n_samples, (int)kchunk);
    else
        return -2;
    return fneus::launch_status();
}
