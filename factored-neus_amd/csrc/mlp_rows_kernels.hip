// The trained MLPs of stages 2 and 3 on a few hundred rows: Lvis and IndirectLight (reference models/fields.py:338-413), the BRDF
// auto-encoder and net_cs of EnvmapMaterialNetwork (models/inverRender.py:451-598).  Plain nn.Linear layers, fp32, 512 .. 2048 rows
// per step: every product is a few hundred MFLOP, so what a step pays is launches -- through torch one GEMM per Linear and
// direction, an element-wise launch per activation and direction, a column reduction per bias gradient (36 + 22 + 12 launches per
// stage-3 step).  Here a LAYER is one launch and a network's weight + bias gradients are ONE launch:
//   forward          Y_l = act_l(Y_{l-1} W_l^T + b_l)                               bias and activation in the epilogue
//   backward, input  delta_{l-1} = (delta_l W_l) * act'_{l-1}(Y_{l-1})              the activation's derivative from its OUTPUT
//   backward, params dW_l = delta_l^T Y_{l-1},  db_l = column sums of delta_l        every layer of the network as one group
// (delta_l = gradient of the pre-activation; the top layer's delta = dY * act'(Y) is applied while the operand is loaded).
// Arithmetic: fp32 MFMA (v_mfma_f32_32x32x2_f32, the unit rocBLAS's fp32 kernels use: products and sums in fp32).  A workgroup owns a
// 32 x 32 tile of the result, its 8 waves split the contraction and their partial tiles are added in wave order (LDS): the result
// does not depend on the launch geometry and is bit-reproducible.  [512 x 512] x 512 = 256 workgroups of 32 MFMAs per wave.
// Independent jobs (the layers of one group) share a launch: blockIdx -> (job, tile) through the prefix table in the arguments.
#include "fneus_common.h"
#include "fneus_kernels.h"

namespace fneus {

constexpr int kRowsMaxJobs = 16;
constexpr int kRowsBatch = 4;

struct RowsJob {
    const float* a;        // D = A B: a(i, k), i < m
    const float* b;        // b(k, j), j < n
    float* d;              // d(i, j) at d[i * ldd + j]
    const float* bias;     // forward: bias[j] or null
    const float* aux;      // backward, input: the previous layer's output [m][ldaux] (act'), null for act 0
    const float* a_aux;    // the top layer's output in A's layout: a(i, k) *= act'(a_aux(i, k)); null = A is delta already
    float* db;             // backward, params: db[i] = sum_k a(i, k); null = no bias
    int m, n, k;
    int lda, ldb, ldd, ldaux;
    int act, a_act;        // 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 sigmoid
    int first_tile, tiles_n;
};

struct RowsJobs {
    RowsJob j[kRowsMaxJobs];
    int n;
};

enum { EPI_FWD = 0, EPI_DX = 1, EPI_DW = 2 };

__device__ __forceinline__ float rows_act(float v, int act) {
    switch (act) {
        case 1: return v > 0.0f ? v : 0.0f;
        case 2: return v > 0.0f ? v : 0.2f * v;
        case 3: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

// derivative of the activation, from its output y (ReLU / LeakyReLU keep the sign; torch: the slope for x <= 0).  Branch-free: the
// activation code is turned into (slope for y <= 0, sigmoid?) once per workgroup
struct RowsDAct {
    float neg;
    bool sig;
};
__device__ __forceinline__ RowsDAct rows_dact_of(int act) { return RowsDAct{act == 1 ? 0.0f : (act == 2 ? 0.2f : 1.0f), act == 3}; }
__device__ __forceinline__ float rows_dact(float y, RowsDAct d) {
    const float lin = y > 0.0f ? 1.0f : d.neg;
    return d.sig ? y * (1.0f - y) : lin;
}

// An operand as a bounds-checked buffer (a load beyond `bytes` returns 0: no branch around any load -- with branches the compiler
// waits for every load before it issues the next, eight memory latencies per round instead of one per batch).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const float* p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}

// 8 values of an operand for one lane: element (i, k0 + s), s = 0..7, UNMASKED (rows_mask8 zeroes what lies beyond the matrix).
// KMAJOR: stored at p[k * ld + i] (the lanes of a half read 32 consecutive floats), else p[i * ld + k] (a lane reads 8 consecutive
// floats: two 16-byte loads when VEC: ld a multiple of 4 and the base 16-byte aligned).
template <bool KMAJOR, bool VEC>
__device__ __forceinline__ void rows_load8(__amdgpu_buffer_rsrc_t rs, int ld, int i, int k0, float out[8]) {
    if constexpr (KMAJOR) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
            out[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ((k0 + s) * ld + i) * 4, 0, 0));
    } else if constexpr (VEC) {
        const f32x4 u = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (i * ld + k0) * 4, 0, 0));
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (i * ld + k0 + 4) * 4, 0, 0));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            out[s] = u[s];
            out[4 + s] = v[s];
        }
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s)
            out[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (i * ld + k0 + s) * 4, 0, 0));
    }
}

// (the loaded registers are pinned first: left to itself the compiler moves each load INTO the branch of its mask and waits for it
// there -- one memory latency per load instead of one per batch)
__device__ __forceinline__ void rows_mask8(float v[8], int i, int ni, int k0, int kend) {
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(v[s]));
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = (i < ni && k0 + s < kend) ? v[s] : 0.0f;
}

// One wave's share [kb, ke) of the contraction, kRowsBatch rounds of 16 k (8 MFMAs) at a time: every load of a batch is in flight
// before its first MFMA, which waits for its own round's operands only (a [512 x 512] x 512 layer is one batch per wave; its 64
// fp32 MFMAs per SIMD are 1.7 us -- the chip's fp32 matrix rate -- and its operands take about as long to arrive).  MEASURED
// (tools/experiments/r05/rows_parts.sh, us per launch inside a graph): an empty launch of this shape 1.6, without the MFMAs 6.6,
// without the loads 5.7, all of it 7.9; a second register set with the next batch's loads behind this batch's MFMAs made the
// compiler wait where the loads are issued (copies into the loop-carried set) and was dropped.
template <bool A_KMAJOR, bool B_KMAJOR, bool VEC, bool AUX>
struct RowsBatch {
    float a[kRowsBatch][8], b[kRowsBatch][8], y[AUX ? kRowsBatch : 1][8];
    __device__ __forceinline__ void load(const RowsJob& J, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, __amdgpu_buffer_rsrc_t ry,
                                         int i, int j, int h, int kk) {
#pragma unroll
        for (int q = 0; q < kRowsBatch; ++q) {
            const int k0 = kk + 16 * q + 8 * h;
#ifdef FNEUS_ROWS_NO_LOADS            // timing experiments only
            for (int s = 0; s < 8; ++s) a[q][s] = 0.001f * (float)(i + s), b[q][s] = 0.002f * (float)(j + k0);
#else
            rows_load8<A_KMAJOR, VEC>(ra, J.lda, i, k0, a[q]);
            rows_load8<B_KMAJOR, VEC>(rb, J.ldb, j, k0, b[q]);
#endif
            if constexpr (AUX) rows_load8<A_KMAJOR, VEC>(ry, J.lda, i, k0, y[q]);
        }
        __builtin_amdgcn_sched_barrier(0);      // (every load of the batch is issued before the first mask is taken)
    }
    template <bool SUM>
    __device__ __forceinline__ void mfmas(const RowsJob& J, RowsDAct da, int i, int j, int h, int kk, int ke, f32x16& acc, float& asum) {
#pragma unroll
        for (int q = 0; q < kRowsBatch; ++q) {
            const int k0 = kk + 16 * q + 8 * h;
            rows_mask8(a[q], i, J.m, k0, ke);
            rows_mask8(b[q], j, J.n, k0, ke);
            if constexpr (AUX) {
                rows_mask8(y[q], i, J.m, k0, ke);
#pragma unroll
                for (int s = 0; s < 8; ++s) a[q][s] *= rows_dact(y[q][s], da);
            }
            if (kk + 16 * q < ke) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    if constexpr (SUM) asum += a[q][s];
#ifdef FNEUS_ROWS_NO_MFMA             // timing experiments only
                    acc[s] = fmaf(a[q][s], b[q][s], acc[s]);
#else
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][s], b[q][s], acc, 0, 0, 0);
#endif
                }
            }
        }
    }
};

template <bool A_KMAJOR, bool B_KMAJOR, bool VEC, bool AUX, bool SUM>
__device__ __forceinline__ void rows_contract(const RowsJob& J, int i, int j, int h, int kb, int ke, f32x16& acc, float& asum) {
    const long a_bytes = (A_KMAJOR ? (long)J.k * J.lda : (long)J.m * J.lda) * 4;
    const long b_bytes = (B_KMAJOR ? (long)J.k * J.ldb : (long)J.n * J.ldb) * 4;
    const __amdgpu_buffer_rsrc_t ra = rows_rsrc(J.a, a_bytes), rb = rows_rsrc(J.b, b_bytes), ry = rows_rsrc(J.a_aux, a_bytes);
    const RowsDAct da = rows_dact_of(J.a_act);
    RowsBatch<A_KMAJOR, B_KMAJOR, VEC, AUX> set;
    for (int kk = kb; kk < ke; kk += 16 * kRowsBatch) {
        set.load(J, ra, rb, ry, i, j, h, kk);
        set.template mfmas<SUM>(J, da, i, j, h, kk, ke, acc, asum);
    }
}

template <bool A_KMAJOR, bool B_KMAJOR, int EPI>
__global__ void __launch_bounds__(512) rows_gemm_kernel(RowsJobs js) {
    __shared__ float red[8][16][64];
    __shared__ float red_db[8][32];
#ifdef FNEUS_ROWS_EMPTY                   // timing experiments only: what a launch of this shape costs
    return;
#endif
    int ji = 0;
    while (ji + 1 < js.n && js.j[ji + 1].first_tile <= (int)blockIdx.x) ++ji;
    const RowsJob& J = js.j[ji];
    const int tile = (int)blockIdx.x - J.first_tile;
    const int i0 = 32 * (tile / J.tiles_n), j0 = 32 * (tile % J.tiles_n);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    // the contraction in 8 contiguous parts, each a multiple of 16 (one round: 8 MFMAs)
    const int part = ((J.k + 7) / 8 + 15) / 16 * 16;
    const int kb = wave * part;
    const int ke = kb + part < J.k ? kb + part : J.k;
    // 16-byte loads for the operands a lane reads along k: every one of them aligned
    const bool vec = (A_KMAJOR || ((J.lda & 3) == 0 && (reinterpret_cast<size_t>(J.a) & 15) == 0 &&
                                   (J.a_aux == nullptr || (reinterpret_cast<size_t>(J.a_aux) & 15) == 0))) &&
                     (B_KMAJOR || ((J.ldb & 3) == 0 && (reinterpret_cast<size_t>(J.b) & 15) == 0));
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    float asum = 0.0f;
    constexpr bool SUM = EPI == EPI_DW;
    if (J.a_aux) {
        if (vec) rows_contract<A_KMAJOR, B_KMAJOR, true, true, SUM>(J, i0 + r, j0 + r, h, kb, ke, acc, asum);
        else rows_contract<A_KMAJOR, B_KMAJOR, false, true, SUM>(J, i0 + r, j0 + r, h, kb, ke, acc, asum);
    } else {
        if (vec) rows_contract<A_KMAJOR, B_KMAJOR, true, false, SUM>(J, i0 + r, j0 + r, h, kb, ke, acc, asum);
        else rows_contract<A_KMAJOR, B_KMAJOR, false, false, SUM>(J, i0 + r, j0 + r, h, kb, ke, acc, asum);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][e][lane] = acc[e];
    if constexpr (EPI == EPI_DW) {
        asum += xor32(asum);
        if (lane < 32) red_db[wave][lane] = asum;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int o = (int)threadIdx.x + 512 * q;
        const int e = o >> 6, ln = o & 63;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[w][e][ln];
        const int gi = i0 + (e >> 2) * 8 + (ln >> 5) * 4 + (e & 3), gj = j0 + (ln & 31);
        if (gi < J.m && gj < J.n) {
            if constexpr (EPI == EPI_FWD) {
                if (J.bias) v += J.bias[gj];
                v = rows_act(v, J.act);
            } else if constexpr (EPI == EPI_DX) {
                if (J.aux) v *= rows_dact(J.aux[(size_t)gi * J.ldaux + gj], rows_dact_of(J.act));
            }
            J.d[(size_t)gi * J.ldd + gj] = v;
        }
    }
    if constexpr (EPI == EPI_DW) {
        if (J.db && j0 == 0 && threadIdx.x < 32 && i0 + (int)threadIdx.x < J.m) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += red_db[w][threadIdx.x];
            J.db[i0 + threadIdx.x] = s;
        }
    }
}

static int rows_launch(const FneusMlpJob* jobs, int n_jobs, int kind, hipStream_t stream) {
    for (int first = 0; first < n_jobs; first += kRowsMaxJobs) {
        RowsJobs js;
        js.n = n_jobs - first < kRowsMaxJobs ? n_jobs - first : kRowsMaxJobs;
        int tiles = 0;
        for (int q = 0; q < js.n; ++q) {
            const FneusMlpJob& f = jobs[first + q];
            RowsJob& J = js.j[q];
            if (f.rows <= 0 || f.n_in <= 0 || f.n_out <= 0 || (kind != EPI_DW && !f.weight)) {
                set_last_error("fneus_mlp: a layer needs rows, n_in, n_out > 0 (and its weight, but for the parameter gradients)");
                return -2;
            }
            J.bias = nullptr, J.aux = nullptr, J.a_aux = nullptr, J.db = nullptr;
            J.act = 0, J.a_act = 0, J.ldaux = 0;
            if (kind == EPI_FWD) {               // y[rows][n_out] = act(x[rows][n_in] W[n_out][n_in]^T + b)
                if (!f.x || !f.y) { set_last_error("fneus_mlp_forward: x and y must be given"); return -2; }
                J.a = f.x, J.lda = f.n_in, J.b = f.weight, J.ldb = f.n_in, J.d = f.y, J.ldd = f.n_out;
                J.m = f.rows, J.n = f.n_out, J.k = f.n_in;
                J.bias = f.bias, J.act = f.act;
            } else if (kind == EPI_DX) {         // dx[rows][n_in] = (delta[rows][n_out] W[n_out][n_in]) * act_in'(x)
                if (!f.dy || !f.dx || (f.act_in != 0 && !f.x) || (f.act != 0 && !f.y)) {
                    set_last_error("fneus_mlp_backward_input: dy, dx and the outputs the activations' derivatives are taken from must be given");
                    return -2;
                }
                J.a = f.dy, J.lda = f.n_out, J.b = f.weight, J.ldb = f.n_in, J.d = f.dx, J.ldd = f.n_in;
                J.m = f.rows, J.n = f.n_in, J.k = f.n_out;
                if (f.act_in != 0) J.aux = f.x, J.ldaux = f.n_in, J.act = f.act_in;
                if (f.act != 0) J.a_aux = f.y, J.a_act = f.act;
            } else {                             // dW[n_out][n_in] = delta^T x, db[n_out] = column sums of delta
                if (!f.dy || !f.x || !f.d_weight || (f.act != 0 && !f.y)) {
                    set_last_error("fneus_mlp_backward_params: dy, x, d_weight (and y for an activation) must be given");
                    return -2;
                }
                J.a = f.dy, J.lda = f.n_out, J.b = f.x, J.ldb = f.n_in, J.d = f.d_weight, J.ldd = f.n_in;
                J.m = f.n_out, J.n = f.n_in, J.k = f.rows;
                J.db = f.d_bias;
                if (f.act != 0) J.a_aux = f.y, J.a_act = f.act;
            }
            // (operands are addressed through 32-bit buffer offsets: an array of 2 GiB or more is another caller's problem -- the stage-2 / 3
            //  networks see a few thousand rows)
            const long widest = f.n_in > f.n_out ? f.n_in : f.n_out;
            if ((long)f.rows * widest * 4 >= (1L << 31) || (long)f.n_in * f.n_out * 4 >= (1L << 31)) {
                set_last_error("fneus_mlp: an operand of 2 GiB or more (rows x width x 4 bytes): split the rows");
                return -2;
            }
            J.tiles_n = (J.n + 31) / 32;
            J.first_tile = tiles;
            tiles += ((J.m + 31) / 32) * J.tiles_n;
        }
        if (kind == EPI_FWD) hipLaunchKernelGGL((rows_gemm_kernel<false, false, EPI_FWD>), dim3((unsigned)tiles), dim3(512), 0, stream, js);
        else if (kind == EPI_DX) hipLaunchKernelGGL((rows_gemm_kernel<false, true, EPI_DX>), dim3((unsigned)tiles), dim3(512), 0, stream, js);
        else hipLaunchKernelGGL((rows_gemm_kernel<true, true, EPI_DW>), dim3((unsigned)tiles), dim3(512), 0, stream, js);
    }
    return launch_status();
}

}  // namespace fneus

using namespace fneus;

extern "C" int fneus_mlp_forward(const FneusMlpJob* layers, int n_layers, fneus_stream_t stream) {
    fneus::clear_status();
    if (n_layers <= 0) return 0;
    if (!layers) { fneus::set_last_error("fneus_mlp_forward: layers must be given"); return -2; }
    return rows_launch(layers, n_layers, EPI_FWD, (hipStream_t)stream);
}

extern "C" int fneus_mlp_backward_input(const FneusMlpJob* layers, int n_layers, fneus_stream_t stream) {
    fneus::clear_status();
    if (n_layers <= 0) return 0;
    if (!layers) { fneus::set_last_error("fneus_mlp_backward_input: layers must be given"); return -2; }
    return rows_launch(layers, n_layers, EPI_DX, (hipStream_t)stream);
}

extern "C" int fneus_mlp_backward_params(const FneusMlpJob* layers, int n_layers, fneus_stream_t stream) {
    fneus::clear_status();
    if (n_layers <= 0) return 0;
    if (!layers) { fneus::set_last_error("fneus_mlp_backward_params: layers must be given"); return -2; }
    return rows_launch(layers, n_layers, EPI_DW, (hipStream_t)stream);
}
