// Weight packing: natural fp32 parameters -> MFMA A-operand fragments (bf16 hi/lo planes) + accumulator-layout
// fp32 vectors.  One launch packs every layer of a network from a job table built once by the host
// (fneus/netdesc.py).  Runs every optimiser step (weights change), so it is a single small kernel.
#include "fneus_common.h"
#include "fneus_layout.h"
#include "fneus_pack.h"
#include "fneus_kernels.h"

namespace fneus {

// rowscale[r] = g/||v||, invnorm[r] = 1/||v|| for every weight-normalised row (nn.utils.weight_norm, dim=0:
// reference models/fields.py:67-68, 139-140).  One wavefront per row.
FN_DEV void rowscale_row(const RowInfo* __restrict__ rows, int n_rows, const float* __restrict__ raw,
                         float* __restrict__ rowscale, float* __restrict__ invnorm, int row, int lane) {
    if (row >= n_rows) return;
    const RowInfo ri = rows[row];
    float s = 0.0f;
    for (int i = lane; i < ri.n_in; i += 64) {
        const float v = raw[ri.off_v + i];
        s += v * v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) {
        if (ri.off_g == 0xFFFFFFFFu) {
            rowscale[row] = 1.0f;
            invnorm[row] = 0.0f;
        } else {
            const float inv = 1.0f / sqrtf(s);
            rowscale[row] = raw[ri.off_g] * inv;
            invnorm[row] = inv;
        }
    }
}
__global__ void __launch_bounds__(64) rowscale_kernel(const RowInfo* __restrict__ rows, int n_rows,
                                                      const float* __restrict__ raw, float* __restrict__ rowscale,
                                                      float* __restrict__ invnorm) {
    rowscale_row(rows, n_rows, raw, rowscale, invnorm, blockIdx.x, threadIdx.x);
}

// Backward of the fold W = g v/||v|| (+ bias pass-through): d_eff (effective layout: W then b per layer) -> raw grads.
//   dg = <dW, v>/||v|| ;  dv = (g/||v||) (dW - v <dW, v>/||v||^2)
// d_eff is CONSUMED: every value is cleared once it has been read, so the weight-gradient GEMM of the next step
// accumulates (fp32 atomics) into a zeroed buffer without a memset.
FN_DEV void wn_backward_block(const RowInfo* __restrict__ rows, int n_rows, const float* __restrict__ raw,
                              const float* __restrict__ rowscale, const float* __restrict__ invnorm, float* __restrict__ d_eff,
                              float* __restrict__ d_raw, const int4* __restrict__ segs, int n_segs, int row, int lane) {
    if (row >= n_rows) {
        // workgroups behind the rows: bias gradients pass straight through,
        // segs[i] = (src_off, dst_off, count, _): d_raw[dst_off + j] += d_eff[src_off + j]
        const int seg = row - n_rows;
        if (seg < n_segs) {
            const int4 sg = segs[seg];
            for (int j = lane; j < sg.z; j += 64) {
                d_raw[sg.y + j] += d_eff[sg.x + j];
                d_eff[sg.x + j] = 0.0f;
            }
        }
        return;
    }
    const RowInfo ri = rows[row];
    if (ri.off_g == 0xFFFFFFFFu) {
        for (int i = lane; i < ri.n_in; i += 64) {
            d_raw[ri.off_v + i] += d_eff[ri.off_w_eff + i];
            d_eff[ri.off_w_eff + i] = 0.0f;
        }
        return;
    }
    float dot = 0.0f;
    for (int i = lane; i < ri.n_in; i += 64) dot += d_eff[ri.off_w_eff + i] * raw[ri.off_v + i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) dot += __shfl_xor(dot, d, 64);
    const float inv = invnorm[row], rs = rowscale[row];
    const float c = dot * inv * inv;
    for (int i = lane; i < ri.n_in; i += 64) {
        d_raw[ri.off_v + i] += rs * (d_eff[ri.off_w_eff + i] - raw[ri.off_v + i] * c);
        d_eff[ri.off_w_eff + i] = 0.0f;
    }
    if (lane == 0) d_raw[ri.off_g] += dot * inv;
}
__global__ void __launch_bounds__(64) wn_backward_kernel(const RowInfo* __restrict__ rows, int n_rows,
                                                         const float* __restrict__ raw, const float* __restrict__ rowscale,
                                                         const float* __restrict__ invnorm, float* __restrict__ d_eff,
                                                         float* __restrict__ d_raw, const int4* __restrict__ segs,
                                                         int n_segs) {
    wn_backward_block(rows, n_rows, raw, rowscale, invnorm, d_eff, d_raw, segs, n_segs, blockIdx.x, threadIdx.x);
}
// several networks in one launch (fneus_wn_backward_multi), like the multi-network refresh below
constexpr int kMaxWnTasks = 8;
struct WnTasks {
    int n;
    int first[kMaxWnTasks + 1];
    const RowInfo* rows[kMaxWnTasks];
    int n_rows[kMaxWnTasks], n_segs[kMaxWnTasks];
    const int4* segs[kMaxWnTasks];
    const float* raw[kMaxWnTasks];
    const float* rowscale[kMaxWnTasks];
    const float* invnorm[kMaxWnTasks];
    float* d_eff[kMaxWnTasks];
    float* d_raw[kMaxWnTasks];
};
__global__ void __launch_bounds__(64) wn_backward_multi_kernel(WnTasks T) {
    int k = 0;
    for (int i = 1; i < T.n; ++i)
        if (T.first[i] <= (int)blockIdx.x) k = i;
    wn_backward_block(T.rows[k], T.n_rows[k], T.raw[k], T.rowscale[k], T.invnorm[k], T.d_eff[k], T.d_raw[k], T.segs[k], T.n_segs[k],
                      (int)blockIdx.x - T.first[k], threadIdx.x);
}

FN_DEV void pack_unit(const PackJob* __restrict__ jobs, int n_jobs, const int* __restrict__ maps,
                      const float* __restrict__ params, const float* __restrict__ rowscale,
                      unsigned char* __restrict__ blob, int unit, int lane) {
    // locate the job: unit_base is ascending -- a bisection (6 dependent scalar loads for the SDF network's 40 jobs instead of one per
    // job: 14.9 -> 13.1 us per launch; four waves per workgroup instead of one changed nothing: the launch is a chain of dependent
    // gathers -- job, index maps, parameters, row scales -- not dispatch)
    int ji = 0, hi_j = n_jobs - 1;
    while (ji < hi_j) {
        const int mid = (ji + hi_j + 1) >> 1;
        if (jobs[mid].unit_base <= unit) ji = mid;
        else hi_j = mid - 1;
    }
    const PackJob jb = jobs[ji];
    const int u = unit - jb.unit_base;
    const int r = lane & 31, h = lane >> 5;
    if (jb.kind == PACK_FRAG) {
        const int ks = u / jb.nt, t = u % jb.nt;
        // geometry 0: lane = (row r = lane&31, half h): k-slot 16ks + 8h + j ; geometry 1: (row c = lane&15, quarter q): 32ks + 8q + j
        const int rr = jb.geom ? (lane & 15) : r;
        const int row = maps[jb.rowmap + (jb.geom ? 16 : 32) * t + rr];
        const int kbase = jb.geom ? 32 * ks + 8 * (lane >> 4) : 16 * ks + 8 * h;
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = maps[jb.kmap + kbase + j];
            float v = 0.0f;
            if (row >= 0 && k >= 0) {
                v = jb.scale * (jb.transposed ? params[jb.src + (size_t)k * jb.ld + row]
                                              : params[jb.src + (size_t)row * jb.ld + k]);
                if (jb.rs_base >= 0) v *= rowscale[jb.rs_base + (jb.rs_mode == 1 ? k : row)];
            }
            __bf16 a, b;
            split_bf16(v, a, b);
            hi[j] = a;
            lo[j] = b;
        }
        *reinterpret_cast<bf16x8*>(blob + jb.dst_hi + (size_t)u * kFragBytes + lane * 16) = hi;
        *reinterpret_cast<bf16x8*>(blob + jb.dst_lo + (size_t)u * kFragBytes + lane * 16) = lo;
    } else if (jb.geom == 0) {  // PACK_ACCVEC: fp32 vector in accumulator layout [t][h][16]
        const int t = u;
        if (lane < 32) {
            const int hh = lane >> 4, reg = lane & 15;
            const int idx = maps[jb.rowmap + 32 * t + acc_row(reg, hh)];
            float v = idx >= 0 ? jb.scale * params[jb.src + (size_t)idx * jb.ld] : 0.0f;
            if (jb.rs_base >= 0 && idx >= 0) v *= rowscale[jb.rs_mode == 2 ? jb.rs_base : jb.rs_base + idx];
            reinterpret_cast<float*>(blob + jb.dst_hi)[(t * 2 + hh) * 16 + reg] = v;
        }
    } else {                    // 16-row tiles: natural order, 16 floats per tile
        const int t = u;
        if (lane < 16) {
            const int idx = maps[jb.rowmap + 16 * t + lane];
            float v = idx >= 0 ? jb.scale * params[jb.src + (size_t)idx * jb.ld] : 0.0f;
            if (jb.rs_base >= 0 && idx >= 0) v *= rowscale[jb.rs_mode == 2 ? jb.rs_base : jb.rs_base + idx];
            reinterpret_cast<float*>(blob + jb.dst_hi)[t * 16 + lane] = v;
        }
    }
}
__global__ void __launch_bounds__(64) pack_kernel(const PackJob* __restrict__ jobs, int n_jobs,
                                                  const int* __restrict__ maps,
                                                  const float* __restrict__ params,
                                                  const float* __restrict__ rowscale,
                                                  unsigned char* __restrict__ blob) {
    pack_unit(jobs, n_jobs, maps, params, rowscale, blob, blockIdx.x, threadIdx.x);
}

// The same two kernels over SEVERAL networks in one launch each (fneus_refresh_multi): a training step re-packs four or five
// networks at its start, each launch a few microseconds of work behind ~8 us of launch latency.  The task table travels by
// value in the kernel arguments; block b belongs to the task whose [first, first + count) holds it.
constexpr int kMaxPackTasks = 8;
struct PackTasks {
    int n;
    int first_unit[kMaxPackTasks + 1], first_row[kMaxPackTasks + 1];
    const PackJob* jobs[kMaxPackTasks];
    int n_jobs[kMaxPackTasks];
    const int* maps[kMaxPackTasks];
    const float* params[kMaxPackTasks];
    float* rowscale[kMaxPackTasks];
    float* invnorm[kMaxPackTasks];
    unsigned char* blob[kMaxPackTasks];
    const RowInfo* rows[kMaxPackTasks];
};
__global__ void __launch_bounds__(64) rowscale_multi_kernel(PackTasks T) {
    int k = 0;
    for (int i = 1; i < T.n; ++i)
        if (T.first_row[i] <= (int)blockIdx.x) k = i;
    rowscale_row(T.rows[k], T.first_row[k + 1] - T.first_row[k], T.params[k], T.rowscale[k], T.invnorm[k],
                 (int)blockIdx.x - T.first_row[k], threadIdx.x);
}
__global__ void __launch_bounds__(64) pack_multi_kernel(PackTasks T) {
    int k = 0;
    for (int i = 1; i < T.n; ++i)
        if (T.first_unit[i] <= (int)blockIdx.x) k = i;
    pack_unit(T.jobs[k], T.n_jobs[k], T.maps[k], T.params[k], T.rowscale[k], T.blob[k], (int)blockIdx.x - T.first_unit[k],
              threadIdx.x);
}

}  // namespace fneus

extern "C" int fneus_refresh_multi(const FneusPackTask* tasks, int n_tasks, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_tasks <= 0) return 0;
    if (n_tasks > fneus::kMaxPackTasks || !tasks) return -2;
    fneus::PackTasks T;
    T.n = n_tasks;
    int units = 0, rows = 0;
    for (int i = 0; i < n_tasks; ++i) {
        T.first_unit[i] = units;
        T.first_row[i] = rows;
        units += tasks[i].n_units > 0 ? tasks[i].n_units : 0;
        rows += tasks[i].n_rows > 0 ? tasks[i].n_rows : 0;
        T.jobs[i] = reinterpret_cast<const fneus::PackJob*>(tasks[i].jobs);
        T.n_jobs[i] = tasks[i].n_jobs;
        T.maps[i] = tasks[i].maps;
        T.params[i] = tasks[i].params;
        T.rowscale[i] = tasks[i].rowscale;
        T.invnorm[i] = tasks[i].invnorm;
        T.blob[i] = reinterpret_cast<unsigned char*>(tasks[i].blob);
        T.rows[i] = reinterpret_cast<const fneus::RowInfo*>(tasks[i].rows);
    }
    T.first_unit[n_tasks] = units;
    T.first_row[n_tasks] = rows;
    for (int i = n_tasks + 1; i <= fneus::kMaxPackTasks; ++i) T.first_unit[i] = T.first_row[i] = 0x7fffffff;
    if (rows > 0) hipLaunchKernelGGL(fneus::rowscale_multi_kernel, dim3(rows), dim3(64), 0, stream, T);
    if (units > 0) hipLaunchKernelGGL(fneus::pack_multi_kernel, dim3(units), dim3(64), 0, stream, T);
    return fneus::launch_status();
}

extern "C" int fneus_pack(const void* jobs, int n_jobs, int n_units, const int* maps, const float* params,
                          const float* rowscale, void* blob, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_units <= 0) return 0;
    hipLaunchKernelGGL(fneus::pack_kernel, dim3(n_units), dim3(64), 0, stream,
                       reinterpret_cast<const fneus::PackJob*>(jobs), n_jobs, maps, params, rowscale,
                       reinterpret_cast<unsigned char*>(blob));
    return fneus::launch_status();
}

extern "C" int fneus_rowscale(const void* rows, int n_rows, const float* raw, float* rowscale, float* invnorm,
                              fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(fneus::rowscale_kernel, dim3(n_rows), dim3(64), 0, stream,
                       reinterpret_cast<const fneus::RowInfo*>(rows), n_rows, raw, rowscale, invnorm);
    return fneus::launch_status();
}

extern "C" int fneus_wn_backward(const void* rows, int n_rows, const void* bias_segs, int n_segs, const float* raw,
                                 const float* rowscale, const float* invnorm, float* d_eff, float* d_raw,
                                 fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rows + n_segs > 0)
        hipLaunchKernelGGL(fneus::wn_backward_kernel, dim3(n_rows + n_segs), dim3(64), 0, stream,
                           reinterpret_cast<const fneus::RowInfo*>(rows), n_rows, raw, rowscale, invnorm, d_eff, d_raw,
                           reinterpret_cast<const int4*>(bias_segs), n_segs);
    return fneus::launch_status();
}

extern "C" int fneus_wn_backward_multi(const FneusWnTask* tasks, int n_tasks, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_tasks <= 0) return 0;
    if (n_tasks > fneus::kMaxWnTasks || !tasks) return -2;
    fneus::WnTasks T;
    T.n = n_tasks;
    int blocks = 0;
    for (int i = 0; i < n_tasks; ++i) {
        T.first[i] = blocks;
        blocks += tasks[i].n_rows + tasks[i].n_segs;
        T.rows[i] = reinterpret_cast<const fneus::RowInfo*>(tasks[i].rows);
        T.n_rows[i] = tasks[i].n_rows;
        T.n_segs[i] = tasks[i].n_segs;
        T.segs[i] = reinterpret_cast<const int4*>(tasks[i].bias_segs);
        T.raw[i] = tasks[i].raw;
        T.rowscale[i] = tasks[i].rowscale;
        T.invnorm[i] = tasks[i].invnorm;
        T.d_eff[i] = tasks[i].d_eff;
        T.d_raw[i] = tasks[i].d_raw;
    }
    for (int i = n_tasks; i <= fneus::kMaxWnTasks; ++i) T.first[i] = i == n_tasks ? blocks : 0x7fffffff;
    if (blocks > 0) hipLaunchKernelGGL(fneus::wn_backward_multi_kernel, dim3(blocks), dim3(64), 0, stream, T);
    return fneus::launch_status();
}
