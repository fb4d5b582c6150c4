// Per-ray tail of the stage-1 step, fused:
//   surface_gather : the two samples bracketing the first SDF sign change of every ray (renderer.py:290-293, 316-327):
//                    their index, depth, feature row and normal, packed as [2B] rows for the RefColor heads
//   stage1_loss    : RefColor shading (linear -> sRGB, clip; fields.py:262-268, 331-335), the two-sample blend
//                    (renderer.py:336-343), the four loss terms of the training loop (exp_runner.py:141-177) AND their
//                    gradients with respect to every differentiable input, in one launch.
// In PyTorch this tail was ~230 element-wise / reduction kernels on [512]-ray tensors (0.7 ms of a 5.2 ms step).
// The loss needs batch-wide normalisers before any per-ray gradient can be formed, so it runs as ONE workgroup that
// sweeps the rays twice (B is a few hundred to a few thousand); sums are reduced in a fixed order (deterministic).
#include "pp_engine.h"
#include "fneus_kernels.h"

namespace fneus {

// ---- L2 warm-up riding on launches that leave the chip idle ------------------------------------------------------------------
// The two RefColor launches of a step are 64 workgroups each and as long as ONE tile's chain: 35 us in the step against 21 us
// with their 1.1 MB of weight fragments per head in L2 (every XCD has its own).  surface_gather (512 single-wave workgroups)
// runs right in front of the forward launch and the loss kernel (ONE workgroup) right in front of the backward one: extra
// workgroups of these launches read the fragments the next launch is about to stream -- workgroup w serves XCD w % 8
// (round-robin dispatch), the workgroups of an XCD share the lines of every range.  The caller names the ranges per launch (FneusWarmRanges).
constexpr int kWarmRanges = FNEUS_MAX_WARM_RANGES;
struct WarmList {
    int n;
    const unsigned char* p[kWarmRanges];
    unsigned lines[kWarmRanges];            // 128-byte lines
};
FN_DEV void warm_l2(const WarmList& W, int wb, int n_wb, int tid, int n_threads) {
    const int j = wb >> 3, J = n_wb >> 3;
    unsigned acc = 0;
    for (int r = 0; r < W.n; ++r)
        for (unsigned line = (unsigned)(j * n_threads + tid); line < W.lines[r]; line += (unsigned)(J * n_threads))
            acc += *reinterpret_cast<const unsigned*>(W.p[r] + (size_t)line * 128);
    asm volatile("" ::"v"(acc));
}
// The ranges are an ARGUMENT of the launch that warms them (FneusWarmRanges, passed by the caller that owns the buffers): the
// library keeps no pointer beyond a call, and a captured graph holds exactly the ranges its own step named.
static WarmList warm_list(const FneusWarmRanges* w) {
    WarmList W;
    W.n = 0;
    if (w != nullptr && w->n > 0) {
        W.n = w->n < kWarmRanges ? w->n : kWarmRanges;
        for (int i = 0; i < W.n; ++i) {
            W.p[i] = reinterpret_cast<const unsigned char*>(w->ptr[i]);
            W.lines[i] = (unsigned)(w->bytes[i] / 128);
        }
    }
    return W;
}

// one wavefront per ray: sel[2b], sel[2b+1] = b*n + hi - 1, b*n + hi  with hi = sdf_mask ? min_idx : 1
__global__ void __launch_bounds__(64) surface_gather_kernel(const int32_t* __restrict__ min_idx,
                                                            const unsigned char* __restrict__ sdf_mask,
                                                            const float* __restrict__ mid_z,    // [B][n]
                                                            const float* __restrict__ feat,     // [B*n][256], or NULL: the planes
                                                            const unsigned char* __restrict__ feat_hi,   // fragment planes of the
                                                            const unsigned char* __restrict__ feat_lo,   // features (lo may be NULL)
                                                            const float* __restrict__ normal,   // [B*n][3]
                                                            int n, int32_t* __restrict__ sel, float* __restrict__ t_sel,
                                                            float* __restrict__ feat_sel, float* __restrict__ normal_sel,
                                                            int n_rays, WarmList W) {
    if ((int)blockIdx.x >= n_rays) {        // extra workgroups: L2 warm-up (see above)
        warm_l2(W, (int)blockIdx.x - n_rays, (int)gridDim.x - n_rays, threadIdx.x, 64);
        return;
    }
    const int b = blockIdx.x, lane = threadIdx.x;
    const int hi = sdf_mask[b] ? min_idx[b] : 1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const long src = (long)b * n + hi - 1 + k;
        const long dst = 2L * b + k;
        if (feat != nullptr) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(feat + src * 256 + lane * 4);
            *reinterpret_cast<f32x4*>(feat_sel + dst * 256 + lane * 4) = v;
        } else if (lane < 32) {
            // the row from the feature planes (fneus_pp.h): lane = (k-step, lane half h) holds the 8 features phi(ks, h, 0..7) =
            // 16 ks + 4 h + {0..3} and 16 ks + 8 + 4 h + {0..3} of sample r = src & 31 as hi + lo bf16: two float4 of the row
            const int ks = lane >> 1, hh = lane & 1, rr = (int)(src & 31);
            const size_t off = (size_t)(src >> 5) * kPPBlock + (size_t)ks * kFragBytes + pp_slot_bytes(rr + 32 * hh, ks);
            const bf16x8 vh = *reinterpret_cast<const bf16x8*>(feat_hi + off);
            f32x4 a, b2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = (float)vh[j];
                b2[j] = (float)vh[4 + j];
            }
            if (feat_lo != nullptr) {
                const bf16x8 vl = *reinterpret_cast<const bf16x8*>(feat_lo + off);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] += (float)vl[j];
                    b2[j] += (float)vl[4 + j];
                }
            }
            *reinterpret_cast<f32x4*>(feat_sel + dst * 256 + 16 * ks + 4 * hh) = a;
            *reinterpret_cast<f32x4*>(feat_sel + dst * 256 + 16 * ks + 8 + 4 * hh) = b2;
        }
        if (lane < 3) normal_sel[dst * 3 + lane] = normal[src * 3 + lane];
        if (lane == 3) t_sel[dst] = mid_z[src];
        if (lane == 4) sel[dst] = (int32_t)src;
    }
}

struct LossArgs {
    // inputs
    const float* color;      // [B][3]  composited colour
    const float* true_rgb;   // [B][3]
    const float* mask_in;    // [B]
    const float* wsum;       // [B]     weight sum
    const float* eik_num;    // [B]     eikonal numerator / denominator per ray (renderer.py:370-372)
    const float* eik_den;    // [B]
    const float* diffuse;    // [2B][3] net_cd output (after sigmoid)
    const float* spec;       // [2B][3] column 0 = net_cs output (after sigmoid)
    const float* wpair;      // [B][2]  compositing weights of the two bracketing samples
    const unsigned char* sdf_mask;   // [B]
    const float* norms;      // optional [4]: sum mask, sum mask*sdf_mask, sum eik_den, ray count over the GLOBAL batch
    int B;
    float igr_weight, mask_weight, surface_weight;
    // outputs
    float* losses;           // [8] loss, color, surface, eikonal, mask, psnr, mask_sum, mask_sdf_sum
    float* surface_color;    // [B][3]
    float* specular_color;   // [B][3]
    float* diffuse_color;    // [B][3]
    float* d_color;          // [B][3]
    float* d_wsum;           // [B]
    float* d_eiknum;         // [B]
    float* d_wpair;          // [B][2]
    float* d_diffuse;        // [2B][3]
    float* d_spec;           // [2B][3] (column 0; 1, 2 zero)
};

// linear -> sRGB (fields.py:262-268) and its derivative on the selected branch
FN_DEV float srgb(float x) {
    const float eps = 1.1920928955078125e-07f;
    return x <= 0.0031308f ? (323.0f / 25.0f) * x : (211.0f * powf(fmaxf(x, eps), 5.0f / 12.0f) - 11.0f) / 200.0f;
}
FN_DEV float dsrgb(float x) {
    return x <= 0.0031308f ? (323.0f / 25.0f) : (211.0f / 200.0f) * (5.0f / 12.0f) * powf(x, -7.0f / 12.0f);
}
FN_DEV float clip01(float y) { return fminf(fmaxf(y, 0.0f), 1.0f); }
FN_DEV float sgn(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

constexpr int kLossThreads = 1024;

// block-wide sum of NV values per thread, fixed reduction order; result valid in every thread
template <int NV>
FN_DEV void block_sum(float (&v)[NV], float* sm /*[NV][16]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[i] += __shfl_xor(v[i], d, 64);
        if (lane == 0) sm[i * 16 + wave] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = 0.0f;
        for (int w = 0; w < kLossThreads / 64; ++w) s += sm[i * 16 + w];
        v[i] = s;
    }
    __syncthreads();
}

// shading of one ray: blend of the two bracketing samples.  TWO threads per ray (neighbouring lanes 2 b, 2 b + 1 of a wave), one
// per bracketing sample: the 14 sRGB curves of a ray (powf each) were the kernel -- one 1024-thread workgroup, half of it idle with
// 512 rays, every curve evaluated again in the gradient sweep: 25 us per step.  Each thread now evaluates the 7 curves of its
// sample once (the values stay in registers for the gradient sweep when the batch fits one pass) and the pair exchanges its
// terms by lane shuffle.
struct HalfShade {
    float brdf[3], y[3];          // specular + diffuse of this sample per channel, its sRGB value (unclipped)
    float ys, yd[3];              // clip(sRGB(specular)), clip(sRGB(diffuse))
    float w;                      // this sample's blend weight + 1e-5
};
FN_DEV void shade_half(const LossArgs& a, int b, int k, HalfShade& h) {
    h.w = a.wpair[b * 2 + k] + 1e-5f;
    const float sp = a.spec[(2 * b + k) * 3];
    h.ys = clip01(srgb(sp));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float df = a.diffuse[(2 * b + k) * 3 + c];
        h.brdf[c] = sp + df;
        h.y[c] = srgb(h.brdf[c]);
        h.yd[c] = clip01(srgb(df));
    }
}
// both samples' terms of a ray in blend order (sample 0 first: the sums of the one-thread version, bit for bit)
FN_DEV float pair_blend(float mine, float w_mine, int k, float W) {
    const float t = mine * w_mine, o = __shfl_xor(t, 1, 64);
    return ((k == 0 ? t : o) + (k == 0 ? o : t)) / W;
}

__global__ void __launch_bounds__(kLossThreads) stage1_loss_kernel(LossArgs a, WarmList W) {
    __shared__ float red[8 * 16];
    const int tid = threadIdx.x;
    if (blockIdx.x > 0) {                   // extra workgroups: L2 warm-up (see above)
        warm_l2(W, (int)blockIdx.x - 1, (int)gridDim.x - 1, tid, kLossThreads);
        return;
    }
    const bool use_mask = a.mask_weight > 0.0f;
    const bool single = 2 * a.B <= kLossThreads;          // one pass: the shading of sweep 1 is still in registers in sweep 2
    HalfShade hs;
    // ---- sweep 1: batch sums ----
    // 0 mask_sum, 1 mask*sdf_mask, 2 |colour error|, 3 |surface error| (unweighted by 1/sum), 4 eik num, 5 eik den,
    // 6 BCE sum, 7 squared colour error
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = tid; p < 2 * a.B; p += kLossThreads) {
        const int b = p >> 1, k = p & 1;
        const float m = use_mask ? (a.mask_in[b] > 0.5f ? 1.0f : 0.0f) : 1.0f;
        const bool sm = a.sdf_mask[b] != 0;
        shade_half(a, b, k, hs);
        const float Wsum = hs.w + __shfl_xor(hs.w, 1, 64);
        float surf[3], spec_c, diff_c[3];
        spec_c = pair_blend(hs.ys, hs.w, k, Wsum);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            surf[c] = pair_blend(clip01(hs.y[c]), hs.w, k, Wsum);
            diff_c[c] = pair_blend(hs.yd[c], hs.w, k, Wsum);
        }
        if (k != 0) continue;                // the ray's sums and outputs: its first thread
        acc[0] += m;
        acc[1] += sm ? m : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float sc = sm ? surf[c] : 1.0f;
            const float ce = (a.color[b * 3 + c] - a.true_rgb[b * 3 + c]) * m;
            acc[2] += fabsf(ce);
            acc[7] += ce * (a.color[b * 3 + c] - a.true_rgb[b * 3 + c]);
            if (sm) acc[3] += fabsf(a.surface_weight * (sc - a.true_rgb[b * 3 + c]) * m);
            a.surface_color[b * 3 + c] = sc;
            a.specular_color[b * 3 + c] = sm ? spec_c : 1.0f;
            a.diffuse_color[b * 3 + c] = sm ? diff_c[c] : 1.0f;
        }
        acc[4] += a.eik_num[b];
        acc[5] += a.eik_den[b];
        const float w = fminf(fmaxf(a.wsum[b], 1e-3f), 1.0f - 1e-3f);
        acc[6] += -(m * logf(w) + (1.0f - m) * logf(1.0f - w));      // F.binary_cross_entropy (its -100 log clamp is moot here)
    }
    block_sum<8>(acc, red);
    // Normalisers: of this batch, or -- data parallel -- of the global batch (summed over the ranks by the caller), so
    // that R ranks x B rays give exactly the loss and gradient of one R*B-ray batch once the ranks' results are summed.
    const float mask_sum = (a.norms ? a.norms[0] : acc[0]) + 1e-5f;
    const float mask_sdf_sum = (a.norms ? a.norms[1] : acc[1]) + 1e-5f;
    const float eik_den = (a.norms ? a.norms[2] : acc[5]) + 1e-5f;
    const float n_rays = a.norms ? a.norms[3] : (float)a.B;
    const float color_loss = acc[2] / mask_sum;
    const float surface_loss = acc[3] / mask_sdf_sum;
    const float eik_loss = acc[4] / eik_den;
    const float mask_loss = acc[6] / n_rays;
    if (tid == 0) {
        a.losses[0] = color_loss + surface_loss + eik_loss * a.igr_weight + mask_loss * a.mask_weight;
        a.losses[1] = color_loss;
        a.losses[2] = surface_loss;
        a.losses[3] = eik_loss;
        a.losses[4] = mask_loss;
        a.losses[5] = 20.0f * log10f(1.0f / sqrtf(acc[7] / ((acc[0] + 1e-5f) * 3.0f)));     // psnr of THIS batch
        a.losses[6] = mask_sum;
        a.losses[7] = mask_sdf_sum;
        a.losses[8] = a.losses[0];            // the total once more: callers take this slot as the loss tensor of its own (no copy launch)
    }
    // ---- sweep 2: gradients of the total loss (thread (b, k): the gradients of sample k; k = 0 also the ray's) ----
    for (int p = tid; p < 2 * a.B; p += kLossThreads) {
        const int b = p >> 1, k = p & 1;
        const float m = use_mask ? (a.mask_in[b] > 0.5f ? 1.0f : 0.0f) : 1.0f;
        const bool sm = a.sdf_mask[b] != 0;
        if (!single) shade_half(a, b, k, hs);
        const float Wsum = hs.w + __shfl_xor(hs.w, 1, 64);
        float dwk = 0.0f, dspk = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float t = a.true_rgb[b * 3 + c];
            const float surf = sm ? pair_blend(clip01(hs.y[c]), hs.w, k, Wsum) : 1.0f;
            if (k == 0) a.d_color[b * 3 + c] = sgn((a.color[b * 3 + c] - t) * m) * m / mask_sum;
            // surface term: |surface_weight (surf - t) m| / mask_sdf_sum over rays with a sign change
            const float dsurf = sm ? sgn(a.surface_weight * (surf - t) * m) * a.surface_weight * m / mask_sdf_sum : 0.0f;
            const float yk = hs.y[c];
            const float inside = (yk >= 0.0f && yk <= 1.0f) ? 1.0f : 0.0f;          // torch.clip passes the gradient on [0, 1]
            const float dy = dsurf * hs.w / Wsum * inside;
            const float dx = dy != 0.0f ? dy * dsrgb(hs.brdf[c]) : 0.0f;
            a.d_diffuse[(2 * b + k) * 3 + c] = dx;
            dspk += dx;
            dwk += dsurf * (clip01(yk) - surf) / Wsum;
        }
        a.d_wpair[b * 2 + k] = dwk;
        a.d_spec[(2 * b + k) * 3 + 0] = dspk;
        a.d_spec[(2 * b + k) * 3 + 1] = 0.0f;
        a.d_spec[(2 * b + k) * 3 + 2] = 0.0f;
        if (k != 0) continue;
        a.d_eiknum[b] = a.igr_weight / eik_den;
        const float wr = a.wsum[b];
        const bool in = wr >= 1e-3f && wr <= 1.0f - 1e-3f;
        const float w = fminf(fmaxf(wr, 1e-3f), 1.0f - 1e-3f);
        a.d_wsum[b] = in ? a.mask_weight / n_rays * (-m / w + (1.0f - m) / (1.0f - w)) : 0.0f;
    }
}

// the batch sums that normalise the loss terms: [sum mask, sum mask*sdf_mask, sum eik_den, ray count]
__global__ void __launch_bounds__(kLossThreads) stage1_norms_kernel(const float* __restrict__ mask_in,
                                                                    const unsigned char* __restrict__ sdf_mask,
                                                                    const float* __restrict__ eik_den, int B,
                                                                    float mask_weight, float* __restrict__ norms) {
    __shared__ float red[3 * 16];
    float acc[3] = {0, 0, 0};
    for (int b = threadIdx.x; b < B; b += kLossThreads) {
        const float m = mask_weight > 0.0f ? (mask_in[b] > 0.5f ? 1.0f : 0.0f) : 1.0f;
        acc[0] += m;
        acc[1] += sdf_mask[b] ? m : 0.0f;
        acc[2] += eik_den[b];
    }
    block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        norms[0] = acc[0];
        norms[1] = acc[1];
        norms[2] = acc[2];
        norms[3] = (float)B;
    }
}

// the way back of surface_gather: d_feat[sel[i]] += sum over the heads of d_feat_heads[h][i], the same for the normals.  One
// wavefront per gathered row; the 2 B selected rows are distinct (two consecutive samples of each ray), so plain read-modify-write.
__global__ void __launch_bounds__(64) surface_scatter_kernel(const int32_t* __restrict__ sel, const float* __restrict__ dfh,
                                                             const float* __restrict__ dnh, int n_heads, long n_rows,
                                                             float* __restrict__ d_feat, float* __restrict__ d_normal) {
    const long i = blockIdx.x;
    const int lane = threadIdx.x;
    const long dst = sel[i];
    if (dfh != nullptr) {
        f32x4 s = *reinterpret_cast<const f32x4*>(d_feat + dst * 256 + lane * 4);
        for (int hd = 0; hd < n_heads; ++hd) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(dfh + ((long)hd * n_rows + i) * 256 + lane * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += v[e];
        }
        *reinterpret_cast<f32x4*>(d_feat + dst * 256 + lane * 4) = s;
    }
    if (dnh != nullptr && lane < 3) {
        float s = d_normal[dst * 3 + lane];
        for (int hd = 0; hd < n_heads; ++hd) s += dnh[((long)hd * n_rows + i) * 3 + lane];
        d_normal[dst * 3 + lane] = s;
    }
}

// the same on the feature cotangent's bf16 fragments (fneus_pp.h): lane l holds features 4 l .. 4 l + 3 of the row = 8 contiguous
// bytes of fragment ks = l >> 2 (h = l & 1, j0 = 4 ((l >> 1) & 1)) in slot (2 r + h) ^ 8 (ks & 1) of the row's tile
__global__ void __launch_bounds__(64) surface_scatter_plane_kernel(const int32_t* __restrict__ sel, const float* __restrict__ dfh,
                                                                   const float* __restrict__ dnh, int n_heads, long n_rows,
                                                                   unsigned char* __restrict__ plane, float* __restrict__ d_normal) {
    const long i = blockIdx.x;
    const int lane = threadIdx.x;
    const long dst = sel[i];
    if (dfh != nullptr) {
        const int ks = lane >> 2, hh = lane & 1, j0 = 4 * ((lane >> 1) & 1), r = (int)(dst & 31);
        const unsigned slot = (unsigned)((2 * r + hh) ^ (8 * (ks & 1)));
        __bf16* p = reinterpret_cast<__bf16*>(plane + (size_t)(dst >> 5) * kPPBlock + (size_t)ks * kFragBytes + slot * 16u) + j0;
        float s[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = (float)p[e];
        for (int hd = 0; hd < n_heads; ++hd) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(dfh + ((long)hd * n_rows + i) * 256 + lane * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = (__bf16)s[e];
    }
    if (dnh != nullptr && lane < 3) {
        float s = d_normal[dst * 3 + lane];
        for (int hd = 0; hd < n_heads; ++hd) s += dnh[((long)hd * n_rows + i) * 3 + lane];
        d_normal[dst * 3 + lane] = s;
    }
}

}  // namespace fneus

using namespace fneus;

extern "C" int fneus_surface_scatter_plane(const int32_t* sel, const float* d_feat_heads, const float* d_normal_heads, int n_heads,
                                           long n_rows, void* dfeat_hi, long n_pts, float* d_normal, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rows <= 0 || n_heads <= 0) return 0;
    if (sel == nullptr || n_pts <= 0 || (d_feat_heads != nullptr && dfeat_hi == nullptr) || (d_normal_heads != nullptr && d_normal == nullptr))
        return -2;
    hipLaunchKernelGGL(surface_scatter_plane_kernel, dim3((unsigned)n_rows), dim3(64), 0, stream, sel, d_feat_heads, d_normal_heads, n_heads,
                       n_rows, reinterpret_cast<unsigned char*>(dfeat_hi), d_normal);
    return fneus::launch_status();
}

extern "C" int fneus_surface_scatter(const int32_t* sel, const float* d_feat_heads, const float* d_normal_heads, int n_heads, long n_rows,
                                     float* d_feat, float* d_normal, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rows <= 0 || n_heads <= 0) return 0;
    if (sel == nullptr || (d_feat_heads != nullptr && d_feat == nullptr) || (d_normal_heads != nullptr && d_normal == nullptr)) return -2;
    hipLaunchKernelGGL(surface_scatter_kernel, dim3((unsigned)n_rows), dim3(64), 0, stream, sel, d_feat_heads, d_normal_heads, n_heads, n_rows,
                       d_feat, d_normal);
    return fneus::launch_status();
}

extern "C" int fneus_surface_gather(const int32_t* min_idx, const unsigned char* sdf_mask, const float* mid_z,
                                    const float* feat, const void* feat_hi, const void* feat_lo, const float* normal, int n_rays, int n,
                                    int32_t* sel, float* t_sel, float* feat_sel, float* normal_sel, const FneusWarmRanges* warm,
                                    fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0) return 0;
    if (n < 2) return -2;
    const WarmList W = warm_list(warm);
    if (feat == nullptr && feat_hi == nullptr) return -2;
    hipLaunchKernelGGL(surface_gather_kernel, dim3(n_rays + (W.n > 0 ? 1024 : 0)), dim3(64), 0, stream, min_idx, sdf_mask, mid_z, feat,
                       static_cast<const unsigned char*>(feat_hi), static_cast<const unsigned char*>(feat_lo), normal, n, sel, t_sel, feat_sel,
                       normal_sel, n_rays, W);
    return fneus::launch_status();
}

extern "C" int fneus_stage1_loss(const float* color, const float* true_rgb, const float* mask_in, const float* wsum,
                                 const float* eik_num, const float* eik_den, const float* diffuse, const float* spec, const float* wpair,
                                 const unsigned char* sdf_mask, const float* norms, int n_rays, float igr_weight,
                                 float mask_weight, float surface_weight, float* losses, float* surface_color, float* specular_color,
                                 float* diffuse_color, float* d_color, float* d_wsum, float* d_eiknum, float* d_wpair,
                                 float* d_diffuse, float* d_spec, const FneusWarmRanges* warm, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0) return -2;
    LossArgs a{color, true_rgb, mask_in, wsum, eik_num, eik_den, diffuse, spec, wpair, sdf_mask, norms, n_rays, igr_weight, mask_weight,
               surface_weight, losses, surface_color, specular_color, diffuse_color, d_color, d_wsum, d_eiknum, d_wpair,
               d_diffuse, d_spec};
    const WarmList W = warm_list(warm);
    hipLaunchKernelGGL(stage1_loss_kernel, dim3(1 + (W.n > 0 ? 16 : 0)), dim3(kLossThreads), 0, stream, a, W);
    return fneus::launch_status();
}

extern "C" int fneus_stage1_norms(const float* mask_in, const unsigned char* sdf_mask, const float* eik_den, int n_rays,
                                  float mask_weight, float* norms, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0) return -2;
    hipLaunchKernelGGL(stage1_norms_kernel, dim3(1), dim3(kLossThreads), 0, stream, mask_in, sdf_mask, eik_den, n_rays,
                       mask_weight, norms);
    return fneus::launch_status();
}
