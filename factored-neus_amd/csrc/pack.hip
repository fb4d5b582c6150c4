// Weight packing: natural fp32 parameters -> MFMA A-operand fragments (bf16 hi/lo planes) + accumulator-layout
// fp32 vectors.  One launch packs every layer of a network from a job table built once by the host
// (fneus/netdesc.py).  Runs every optimiser step (weights change), so it is a single small kernel.
#include "fneus_common.h"
#include "fneus_layout.h"
#include "fneus_pack.h"
#include "fneus_kernels.h"

namespace fneus {

__global__ void __launch_bounds__(64) pack_kernel(const PackJob* __restrict__ jobs, int n_jobs,
                                                  const int* __restrict__ maps,
                                                  const float* __restrict__ params,
                                                  unsigned char* __restrict__ blob) {
    const int unit = blockIdx.x;
    const int lane = threadIdx.x;
    // locate the job (n_jobs is a few dozen; unit_base is ascending)
    int ji = 0;
    for (int i = 1; i < n_jobs; ++i)
        if (jobs[i].unit_base <= unit) ji = i;
    const PackJob jb = jobs[ji];
    const int u = unit - jb.unit_base;
    const int r = lane & 31, h = lane >> 5;
    if (jb.kind == PACK_FRAG) {
        const int ks = u / jb.nt, t = u % jb.nt;
        const int row = maps[jb.rowmap + 32 * t + r];
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = maps[jb.kmap + 16 * ks + 8 * h + j];
            float v = 0.0f;
            if (row >= 0 && k >= 0)
                v = jb.scale * (jb.transposed ? params[jb.src + (size_t)k * jb.ld + row]
                                              : params[jb.src + (size_t)row * jb.ld + k]);
            __bf16 a, b;
            split_bf16(v, a, b);
            hi[j] = a;
            lo[j] = b;
        }
        *reinterpret_cast<bf16x8*>(blob + jb.dst_hi + (size_t)u * kFragBytes + lane * 16) = hi;
        *reinterpret_cast<bf16x8*>(blob + jb.dst_lo + (size_t)u * kFragBytes + lane * 16) = lo;
    } else {  // PACK_ACCVEC: fp32 vector in accumulator layout [t][h][16]
        const int t = u;
        if (lane < 32) {
            const int hh = lane >> 4, reg = lane & 15;
            const int idx = maps[jb.rowmap + 32 * t + acc_row(reg, hh)];
            const float v = idx >= 0 ? jb.scale * params[jb.src + (size_t)idx * jb.ld] : 0.0f;
            reinterpret_cast<float*>(blob + jb.dst_hi)[(t * 2 + hh) * 16 + reg] = v;
        }
    }
}

}  // namespace fneus

extern "C" int fneus_pack(const void* jobs, int n_jobs, int n_units, const int* maps, const float* params,
                          void* blob, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_units <= 0) return 0;
    hipLaunchKernelGGL(fneus::pack_kernel, dim3(n_units), dim3(64), 0, stream,
                       reinterpret_cast<const fneus::PackJob*>(jobs), n_jobs, maps, params,
                       reinterpret_cast<unsigned char*>(blob));
    return fneus::launch_status();
}
