// Weight-gradient GEMM:  C[m][n] += scale * sum_samples ( A[s][m] * B[s][n]  (+ A2[s][m] * B2[s][n]) )
//
// Replaces the dW = X^T Y products of torch autograd's addmm/mm backward for every Linear on the path (reference
// models/fields.py:86, 168; the second segment carries the double-backward term of SURVEY.md Appendix A:
// dW_l = zbar_l^T u_l + a_l^T adj_l).  Operands are the row-major bf16 stash planes written by the chain kernels, so
// the contraction runs over the ROW index of both operands ("TN" GEMM): tiles are staged in LDS and read back with
// ds_read_b64_tr_b16 (hardware transpose) straight into MFMA operand layout.  Split-K over samples with fp32 atomics.
//
// One workgroup = 256 threads = 2x2 waves, output tile 128x128 (wave: 64x64 = 2x2 MFMA 32x32x16 tiles).
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

constexpr int KB = 64;               // samples per LDS block
constexpr int ROWB = 320;            // LDS row stride in bytes: 256 data + 64 pad -> conflict-free transposed reads
constexpr int TILEB = KB * ROWB;     // 20480

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

FN_DEV bf16x4 tr_read(const unsigned char* lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds_addr));
}

// stage a [KB][128] bf16 tile: rows s0..s0+63 (zero beyond s_end), cols c0..c0+127 (zero beyond w)
FN_DEV void stage_tile(unsigned char* lds, const __bf16* __restrict__ g, int ld, int w, long s0, long s_end, int c0, int tid) {
    const int cchunk = tid & 15;            // 16-byte chunk within the 256-byte row
    const int rbase = tid >> 4;             // 16 rows per pass
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = rbase + 16 * p;
        const long s = s0 + row;
        const int col = c0 + cchunk * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g != nullptr && s < s_end && col < w) v = *reinterpret_cast<const uint4*>(g + s * ld + col);
        *reinterpret_cast<uint4*>(lds + row * ROWB + cchunk * 16) = v;
    }
}

FN_DEV void stage_e0(unsigned char* lds, long s0, long s_end, int c0, int tid) {
    const int cchunk = tid & 15;
    const int rbase = tid >> 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = rbase + 16 * p;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (cchunk == 0 && c0 == 0 && (s0 + row) < s_end) v.x = 0x3F80u;   // bf16(1.0) in element 0
        *reinterpret_cast<uint4*>(lds + row * ROWB + cchunk * 16) = v;
    }
}

// MFMA operand fragment (32 "rows" starting at column f0 of the tile, k-step kk) from an LDS tile
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

template <int PREC>
__global__ void __launch_bounds__(256) dw_gemm_kernel(const GemmJob* __restrict__ jobs, int n_jobs, long N, int kchunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA_hi = smem;
    unsigned char* sB_hi = smem + TILEB;
    unsigned char* sA_lo = smem + 2 * TILEB;
    unsigned char* sB_lo = smem + 3 * TILEB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int ji = 0;
    for (int i = 1; i < n_jobs; ++i)
        if (jobs[i].tile_base <= (int)blockIdx.x) ji = i;
    const GemmJob jb = jobs[ji];
    const int tiles_n = (jb.n + 127) / 128;
    const int tl = blockIdx.x - jb.tile_base;
    const int tm = tl / tiles_n, tn = tl % tiles_n;
    const long s_begin = (long)blockIdx.y * kchunk;
    const long s_end = (s_begin + kchunk < N) ? s_begin + kchunk : N;
    if (s_begin >= s_end) return;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float bias_acc = 0.0f;
    const bool do_bias = (jb.bias != nullptr) && (tn == 0) && (tid < 128);

    for (int seg = 0; seg < 2; ++seg) {
        const __bf16* ah = seg ? jb.a2_hi : jb.a_hi;
        const __bf16* al = seg ? jb.a2_lo : jb.a_lo;
        const __bf16* bh = seg ? jb.b2_hi : jb.b_hi;
        const __bf16* bl = seg ? jb.b2_lo : jb.b_lo;
        const int lda = seg ? jb.lda2 : jb.lda, ldb = seg ? jb.ldb2 : jb.ldb;
        const bool e0 = seg && jb.a2_mode == 1;
        if (bh == nullptr) continue;
        for (long s0 = s_begin; s0 < s_end; s0 += KB) {
            __syncthreads();
            if (e0) stage_e0(sA_hi, s0, s_end, tm * 128, tid);
            else stage_tile(sA_hi, ah, lda, jb.a_w, s0, s_end, tm * 128, tid);
            stage_tile(sB_hi, bh, ldb, jb.b_w, s0, s_end, tn * 128, tid);
            if constexpr (PREC == 3) {
                if (e0) stage_tile(sA_lo, nullptr, 0, 0, s0, s_end, 0, tid);
                else stage_tile(sA_lo, al, lda, jb.a_w, s0, s_end, tm * 128, tid);
                stage_tile(sB_lo, bl, ldb, jb.b_w, s0, s_end, tn * 128, tid);
            }
            __syncthreads();
            if (do_bias && seg == 0) {
                float s = 0.0f;
                for (int row = 0; row < KB; ++row) {
                    s += (float)*reinterpret_cast<const __bf16*>(sA_hi + row * ROWB + tid * 2);
                    if constexpr (PREC == 3) s += (float)*reinterpret_cast<const __bf16*>(sA_lo + row * ROWB + tid * 2);
                }
                bias_acc += s;
            }
#pragma unroll
            for (int kk = 0; kk < KB / 16; ++kk) {
                bf16x8 fa_h[2], fb_h[2], fa_l[2], fb_l[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa_h[i] = frag_from_lds(sA_hi, wr * 64 + i * 32, kk, lane);
                    fb_h[i] = frag_from_lds(sB_hi, wc * 64 + i * 32, kk, lane);
                    if constexpr (PREC == 3) {
                        fa_l[i] = frag_from_lds(sA_lo, wr * 64 + i * 32, kk, lane);
                        fb_l[i] = frag_from_lds(sB_lo, wc * 64 + i * 32, kk, lane);
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (PREC == 3) {
                            acc[i][j] = mfma32(fa_l[i], fb_h[j], acc[i][j]);
                            acc[i][j] = mfma32(fa_h[i], fb_l[j], acc[i][j]);
                        }
                        acc[i][j] = mfma32(fa_h[i], fb_h[j], acc[i][j]);
                    }
            }
        }
    }
    // epilogue: fp32 atomics (two 128-byte row segments per wave instruction)
    const int hh = lane >> 5, cc = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = tn * 128 + wc * 64 + j * 32 + cc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * 128 + wr * 64 + i * 32 + acc_row(r, hh);
                if (row < jb.m && col < jb.n) atomicAdd(jb.c + (size_t)row * jb.ldc + col, jb.scale * acc[i][j][r]);
            }
        }
    if (do_bias) {
        const int row = tm * 128 + tid;
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
    // split-K so that the grid holds ~1024 workgroups (2 per CU resident at 80 KiB LDS each in parity mode)
    int split = 1024 / n_tiles;
    if (split < 1) split = 1;
    long kchunk = (n_samples + split - 1) / split;
    kchunk = ((kchunk + KB - 1) / KB) * KB;
    split = (int)((n_samples + kchunk - 1) / kchunk);
    dim3 grid(n_tiles, split), blk(256);
    const GemmJob* jobs = reinterpret_cast<const GemmJob*>(jobs_dev);
    if (prec == 3) {
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(dw_gemm_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILEB);
            attr_set = true;
        }
        hipLaunchKernelGGL(dw_gemm_kernel<3>, grid, blk, 4 * TILEB, stream, jobs, n_jobs, n_samples, (int)kchunk);
    } else if (prec == 1) {
        hipLaunchKernelGGL(dw_gemm_kernel<1>, grid, blk, 2 * TILEB, stream, jobs, n_jobs, n_samples, (int)kchunk);
    } else {
        return -2;
    }
    return fneus::launch_status();
}
