// Adam step over every parameter of the stage-1 model in one launch (reference: torch.optim.Adam, exp_runner.py:108,
// 179-181: lr schedule applied by the caller, betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).
// The parameters of the fused MLPs live in flat buffers, so the ~60 tensors of the model are a handful of contiguous
// segments; PyTorch's multi-tensor Adam took 0.14 ms for these 1.6 M values (28 bytes each: 3 us of HBM traffic).
// The gradient is cleared in the same pass (the next backward accumulates into zeroed buffers: no memsets).
#include <stdlib.h>
#include "fneus_common.h"
#include "fneus_kernels.h"

namespace fneus {

constexpr int kAdamMaxSegs = 32;         // per launch (kernel-argument space); longer tables take several launches
constexpr int kAdamChunk = 1024;          // values per workgroup iteration: 4 per thread, their loads issued together

struct AdamSegs {
    float* p[kAdamMaxSegs];
    float* g[kAdamMaxSegs];
    float* m[kAdamMaxSegs];
    float* v[kAdamMaxSegs];
    long first_chunk[kAdamMaxSegs + 1];   // prefix sum of ceil(count / kAdamChunk)
    long count[kAdamMaxSegs];
    int n;
};

// `step` (float[2]): [0] the number of steps taken so far, [1] an arrival counter (bits of an unsigned, zero between launches).
// The step count of THIS update is step[0] + 1; the last workgroup to finish writes it back (round 5: the increment was a
// launch of its own).  TICK = 0: a further launch of the same update (tables of more than 32 segments): the count is current.
template <int TICK>
__global__ void __launch_bounds__(256) adam_kernel(AdamSegs s, const float* __restrict__ lr_ptr,
                                                   float* __restrict__ step_ptr, double beta1d, double beta2d,
                                                   float eps, int zero_grad) {
    // hyper-parameters are doubles on the host side of torch.optim.Adam: 1 - beta is rounded once, from the double
    const float beta1 = (float)beta1d, beta2 = (float)beta2d;
    const float omb1 = (float)(1.0 - beta1d), omb2 = (float)(1.0 - beta2d);
    __shared__ float bc[2];
    if (threadIdx.x == 0) {          // the two double-precision pow() once per workgroup, not per thread
        const double t = (double)*step_ptr + (TICK ? 1.0 : 0.0);
#ifdef FNEUS_ADAM_NO_POW                 // timing experiment only
        bc[0] = 0.5f + (float)t * 1e-9f;
        bc[1] = 0.25f;
#else
        bc[0] = (float)(1.0 - pow(beta1d, t));
        bc[1] = (float)(1.0 - pow(beta2d, t));
#endif
    }
    __syncthreads();
    const float lr = *lr_ptr;
    const float step_size = lr / bc[0], inv_sqrt_bc2 = 1.0f / sqrtf(bc[1]);
    const long n_chunks = s.first_chunk[s.n];
    for (long chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        int seg = 0;
        while (seg + 1 < s.n && s.first_chunk[seg + 1] <= chunk) ++seg;
        const long base = (chunk - s.first_chunk[seg]) * kAdamChunk;
        float* __restrict__ p = s.p[seg] + base;
        float* __restrict__ g = s.g[seg] + base;
        float* __restrict__ m = s.m[seg] + base;
        float* __restrict__ v = s.v[seg] + base;
        const long left = s.count[seg] - base;
        const int cnt = left < kAdamChunk ? (int)left : kAdamChunk;
        auto update = [&](float gk, float& mk, float& vk, float& pk) {
            mk = beta1 * mk + omb1 * gk;
            vk = beta2 * vk + omb2 * gk * gk;
            pk = pk - step_size * mk / (sqrtf(vk) * inv_sqrt_bc2 + eps);
        };
        // A whole chunk whose four arrays are 16-byte aligned (every chunk but a segment's last, of every segment the trainers build):
        // one 16-byte load per array and thread, no predicate anywhere (the predicated form puts every load and store into a branch of
        // its own: 19.1 -> 18.1 us at 1.06 M values; the workgroup count below was the larger part).
        const bool whole = cnt == kAdamChunk && ((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) |
                                                  reinterpret_cast<size_t>(v)) & 15) == 0;
        if (whole) {
            const int i = threadIdx.x * 4;
            f32x4 gq = *reinterpret_cast<const f32x4*>(g + i), mq = *reinterpret_cast<const f32x4*>(m + i);
            f32x4 vq = *reinterpret_cast<const f32x4*>(v + i), pq = *reinterpret_cast<const f32x4*>(p + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float mk = mq[k], vk = vq[k], pk = pq[k];
                update(gq[k], mk, vk, pk);
                mq[k] = mk, vq[k] = vk, pq[k] = pk;
            }
            *reinterpret_cast<f32x4*>(m + i) = mq;
            *reinterpret_cast<f32x4*>(v + i) = vq;
            *reinterpret_cast<f32x4*>(p + i) = pq;
            if (zero_grad) *reinterpret_cast<f32x4*>(g + i) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            continue;
        }
        // (the 16 loads of a thread are independent: one memory latency per chunk instead of four)
        float gi[4], mi[4], vi[4], pi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k;
            const bool ok = i < cnt;
            gi[k] = ok ? g[i] : 0.0f;
            mi[k] = ok ? m[i] : 0.0f;
            vi[k] = ok ? v[i] : 0.0f;
            pi[k] = ok ? p[i] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < cnt) {
                update(gi[k], mi[k], vi[k], pi[k]);
                m[i] = mi[k];
                v[i] = vi[k];
                p[i] = pi[k];
                if (zero_grad) g[i] = 0.0f;
            }
        }
    }
    if constexpr (TICK != 0) {           // every workgroup has read step[0] (above) before it arrives here
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned* cnt = reinterpret_cast<unsigned*>(step_ptr + 1);
            if (atomicAdd(cnt, 1u) == gridDim.x - 1) {
                *cnt = 0u;
                *step_ptr += 1.0f;
            }
        }
    }
}

}  // namespace fneus

using namespace fneus;

extern "C" int fneus_adam(const FneusAdamSegment* segs /*host array*/, int n_segs, const float* lr, float* step, double beta1,
                          double beta2, double eps, int zero_grad, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_segs <= 0) return 0;
    bool ticked = false;
    for (int first = 0; first < n_segs; first += kAdamMaxSegs) {
        AdamSegs s;
        s.n = n_segs - first < kAdamMaxSegs ? n_segs - first : kAdamMaxSegs;
        long chunks = 0;
        for (int i = 0; i < s.n; ++i) {
            const FneusAdamSegment& sg = segs[first + i];
            s.p[i] = sg.param;
            s.g[i] = sg.grad;
            s.m[i] = sg.exp_avg;
            s.v[i] = sg.exp_avg_sq;
            s.count[i] = sg.count;
            s.first_chunk[i] = chunks;
            chunks += (sg.count + kAdamChunk - 1) / kAdamChunk;
        }
        s.first_chunk[s.n] = chunks;
        if (chunks == 0) continue;
        // One workgroup per CU, each taking every 256th chunk: MEASURED (1.06 M values, us per launch) 4096 / 1024 workgroups 18.0, 512:
        // 12.6, 256: 10.8, 128: 12.1 -- a workgroup per chunk paid for a thousand arrivals on ONE counter (the step count's tick
        // below) and a thousand double-precision pow() pairs; FNEUS_ADAM_GRID overrides.
        static const long max_grid = getenv("FNEUS_ADAM_GRID") ? atol(getenv("FNEUS_ADAM_GRID")) : 256;
        const long grid = chunks < max_grid ? chunks : max_grid;
        if (!ticked) hipLaunchKernelGGL(adam_kernel<1>, dim3((unsigned)grid), dim3(256), 0, stream, s, lr, step, beta1, beta2, (float)eps, zero_grad);
        else hipLaunchKernelGGL(adam_kernel<0>, dim3((unsigned)grid), dim3(256), 0, stream, s, lr, step, beta1, beta2, (float)eps, zero_grad);
        ticked = true;
    }
    return fneus::launch_status();
}
