// Stage 2: predicted indirect radiance of a surface point along its sampled directions, fused from the raw output of the
// IndirectLight MLP: the network's output transform (reference models/fields.py:395-413: angles -> lobe axis, sharpness,
// amplitude) and query_indir_illum (models/calLvis.py:323-336: normalise the axes, sum of spherical Gaussians) were ~25
// element-wise PyTorch launches forward and ~45 backward on [512, 4, 24, 3] tensors.  One wavefront per point, lane l = lobe l:
//   theta = 2 pi sigmoid(o0), phi = 2 pi sigmoid(o1), axis = (cos theta sin phi, sin theta sin phi, cos phi) / |.|
//   lambda = 30 sigmoid(o2) + 0.1, mu_c = relu(o_{3+c})
//   radiance[s][c] = sum_l mu_c[l] exp(lambda_l (axis_l . d_s - 1))
// The backward is the hand-derived adjoint (checked against autograd of the element-wise formulation, tests/test_hip_stage2.py).
#include "fneus_common.h"
#include "fneus_kernels.h"

namespace fneus {

constexpr int kIndirMaxS = 8;
constexpr float kTwoPi = 6.283185307179586f;

struct IndirLobe {
    float sg0, sg1, sg2;            // sigmoids of o0, o1, o2
    float st, ct, sp, cp;           // sin / cos of theta, phi
    float ax[3], inv_norm;          // normalised axis, 1 / |raw axis|
    float lam, mu[3];
    bool pos[3];
};

FN_DEV IndirLobe indir_lobe(const float* __restrict__ o) {
    IndirLobe L;
    L.sg0 = 1.0f / (1.0f + expf(-o[0]));
    L.sg1 = 1.0f / (1.0f + expf(-o[1]));
    L.sg2 = 1.0f / (1.0f + expf(-o[2]));
    const float theta = L.sg0 * kTwoPi, phi = L.sg1 * kTwoPi;
    sincosf(theta, &L.st, &L.ct);
    sincosf(phi, &L.sp, &L.cp);
    const float a0 = L.ct * L.sp, a1 = L.st * L.sp, a2 = L.cp;
    L.inv_norm = 1.0f / sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    L.ax[0] = a0 * L.inv_norm;
    L.ax[1] = a1 * L.inv_norm;
    L.ax[2] = a2 * L.inv_norm;
    L.lam = L.sg2 * 30.0f + 0.1f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        L.pos[c] = o[3 + c] > 0.0f;
        L.mu[c] = L.pos[c] ? o[3 + c] : 0.0f;
    }
    return L;
}

__global__ void __launch_bounds__(64) indir_illum_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ dirs, int n, int nl,
                                                             int ns, float* __restrict__ radiance) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const bool live = lane < nl;
    float o[6] = {0, 0, 0, 0, 0, 0};
    if (live) {
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = raw[((size_t)i * nl + lane) * 6 + k];
    }
    const IndirLobe L = indir_lobe(o);
    for (int s = 0; s < ns; ++s) {
        const float* d = dirs + ((size_t)i * ns + s) * 3;
        const float cosv = L.ax[0] * d[0] + L.ax[1] * d[1] + L.ax[2] * d[2];
        const float w = live ? expf(L.lam * (cosv - 1.0f)) : 0.0f;
        float v[3] = {L.mu[0] * w, L.mu[1] * w, L.mu[2] * w};
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] += __shfl_xor(v[c], sh, 64);
        }
        if (lane < 3) radiance[((size_t)i * ns + s) * 3 + lane] = lane == 0 ? v[0] : (lane == 1 ? v[1] : v[2]);
    }
}

__global__ void __launch_bounds__(64) indir_illum_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ dirs,
                                                             const float* __restrict__ d_rad, int n, int nl, int ns,
                                                             float* __restrict__ d_raw) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n || lane >= nl) return;
    float o[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = raw[((size_t)i * nl + lane) * 6 + k];
    const IndirLobe L = indir_lobe(o);
    float dmu[3] = {0, 0, 0}, dlam = 0.0f, du[3] = {0, 0, 0};
    for (int s = 0; s < ns; ++s) {
        const float* d = dirs + ((size_t)i * ns + s) * 3;
        const float* g = d_rad + ((size_t)i * ns + s) * 3;
        const float cosv = L.ax[0] * d[0] + L.ax[1] * d[1] + L.ax[2] * d[2];
        const float w = expf(L.lam * (cosv - 1.0f));
        const float a = g[0] * L.mu[0] + g[1] * L.mu[1] + g[2] * L.mu[2];      // dL/dw
#pragma unroll
        for (int c = 0; c < 3; ++c) dmu[c] += g[c] * w;
        dlam += a * w * (cosv - 1.0f);
        const float dcos = a * w * L.lam;
#pragma unroll
        for (int c = 0; c < 3; ++c) du[c] += dcos * d[c];
    }
    // through axis = raw / |raw|
    const float udu = L.ax[0] * du[0] + L.ax[1] * du[1] + L.ax[2] * du[2];
    float dr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) dr[c] = (du[c] - L.ax[c] * udu) * L.inv_norm;
    const float dtheta = dr[0] * (-L.st * L.sp) + dr[1] * (L.ct * L.sp);
    const float dphi = dr[0] * (L.ct * L.cp) + dr[1] * (L.st * L.cp) - dr[2] * L.sp;
    float* out = d_raw + ((size_t)i * nl + lane) * 6;
    out[0] = dtheta * kTwoPi * L.sg0 * (1.0f - L.sg0);
    out[1] = dphi * kTwoPi * L.sg1 * (1.0f - L.sg1);
    out[2] = dlam * 30.0f * L.sg2 * (1.0f - L.sg2);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[3 + c] = L.pos[c] ? dmu[c] : 0.0f;
}

}  // namespace fneus

using namespace fneus;

extern "C" int fneus_indir_illum_fwd(const float* raw, const float* dirs, int n, int n_lobes, int n_dirs, float* radiance,
                                     fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    clear_status();
    if (n <= 0) return 0;
    if (!raw || !dirs || !radiance || n_lobes <= 0 || n_lobes > 64 || n_dirs <= 0 || n_dirs > kIndirMaxS) return -2;
    hipLaunchKernelGGL(indir_illum_fwd_kernel, dim3(n), dim3(64), 0, stream, raw, dirs, n, n_lobes, n_dirs, radiance);
    return launch_status();
}

extern "C" int fneus_indir_illum_bwd(const float* raw, const float* dirs, const float* d_radiance, int n, int n_lobes, int n_dirs,
                                     float* d_raw, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    clear_status();
    if (n <= 0) return 0;
    if (!raw || !dirs || !d_radiance || !d_raw || n_lobes <= 0 || n_lobes > 64 || n_dirs <= 0 || n_dirs > kIndirMaxS) return -2;
    hipLaunchKernelGGL(indir_illum_bwd_kernel, dim3(n), dim3(64), 0, stream, raw, dirs, d_radiance, n, n_lobes, n_dirs, d_raw);
    return launch_status();
}

// ---- sRGB transfer curves as ONE element-wise launch (and one for the adjoint) -------------------------------------------------
// linear_to_srgb / srgb_to_linear (reference models/math_utils.py:138-152, used by RefColor, fields.py:329-335, and by the stage-3
// tone mapping, inverRender.py:13-18) are where(x <= t, a x, pow-branch): 7-8 element-wise PyTorch launches each on [512, 3]
// tensors, as many again in the backward, seven calls per stage-3 step.  mode bit 0: 0 = linear -> sRGB, 1 = sRGB -> linear;
// bit 1: clip the result to [0, 1] (torch.clip behind the curve: zero gradient outside).
namespace fneus {

FN_DEV float srgb_curve(float x, int mode, float& slope) {
    const float eps = 1.1920929e-07f;       // torch.finfo(float32).eps: the clamp in front of the power
    float y;
    if ((mode & 1) == 0) {
        if (x <= 0.0031308f) {
            y = (323.0f / 25.0f) * x;
            slope = 323.0f / 25.0f;
        } else {
            const float c = fmaxf(x, eps);
            const float p = powf(c, 5.0f / 12.0f);
            y = (211.0f * p - 11.0f) / 200.0f;
            slope = x >= eps ? (211.0f / 200.0f) * (5.0f / 12.0f) * p / c : 0.0f;
        }
    } else {
        if (x <= 0.04045f) {
            y = (25.0f / 323.0f) * x;
            slope = 25.0f / 323.0f;
        } else {
            const float t = (200.0f * x + 11.0f) / 211.0f;
            const float c = fmaxf(t, eps);
            const float p = powf(c, 12.0f / 5.0f);
            y = p;
            slope = t >= eps ? (12.0f / 5.0f) * p / c * (200.0f / 211.0f) : 0.0f;
        }
    }
    if (mode & 2) {
        if (y < 0.0f || y > 1.0f) slope = 0.0f;
        y = fminf(fmaxf(y, 0.0f), 1.0f);
    }
    // NaN in, NaN out (value and slope), as torch.where over torch.clamp(...) ** p does (math_utils.py:138-152): fmaxf / fminf
    // return the other operand for a NaN, which would turn a diverged run's colour into a finite number and hide the fault
    if (x != x) {
        y = x;
        slope = x;
    }
    return y;
}

__global__ void __launch_bounds__(256) srgb_fwd_kernel(const float* __restrict__ x, long n, int mode, float* __restrict__ y) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s;
    y[i] = srgb_curve(x[i], mode, s);
}
__global__ void __launch_bounds__(256) srgb_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, long n, int mode,
                                                       float* __restrict__ dx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s;
    (void)srgb_curve(x[i], mode, s);
    dx[i] = dy[i] * s;
}

// The tail of render_with_all_sg (inverRender.py:277, 440, 306-309) for the colour a training step reads: the four lobe sums of
// fneus_sg_render_fwd [n][4][3] (direct specular / diffuse, indirect specular / diffuse) -> clamp each, env = clamp(direct
// specular + diffuse), indir = clamp(indirect ...), rgb = clip(sRGB(env + indir)).  The same operations in the same order as the
// element-wise formulation (3 clamps, 3 adds, the curve: 8 launches forward, ~25 in autograd's backward).
// The latent-sparsity term of stage 3 (inverRender.py:609-612; mateIllu.py:163): rho_hat_j = mean over the marked points of
// sigmoid(latent[i][j]), kl = mean_j (rho log(rho / rho_hat_j) + (1 - rho) log((1 - rho) / (1 - rho_hat_j))); 0 without a marked
// point.  ~22 element-wise launches on [32]-element tensors forward and ~20 backward, as one launch each: ONE workgroup, the
// column sums over the points in a fixed order (thread t sums rows t, t + 32, ... of column t & 31; then a tree over the 32 row
// groups).  stats [34]: rho_hat [32], the number of marked points, kl -- saved for the backward.
constexpr int kKlDim = 32;
// activated != 0: `latent` holds sigmoid(latent) already (the encoder's last layer applied it: fneus_mlp_forward, act 3)
__global__ void __launch_bounds__(1024) latent_kl_fwd_kernel(const float* __restrict__ latent, const unsigned char* __restrict__ mask,
                                                             int n, float rho, int activated, float* __restrict__ stats) {
    __shared__ float part[32][kKlDim + 1];
    __shared__ float cnts[32];
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    float s = 0.0f, c = 0.0f;
    for (int i = grp; i < n; i += 32) {
        const float w = mask ? (mask[i] ? 1.0f : 0.0f) : 1.0f;
        const float x = latent[(long)i * kKlDim + col];
        s += activated ? w * x : w / (1.0f + expf(-x));
        c += w;
    }
    part[grp][col] = s;
    if (col == 0) cnts[grp] = c;
    __syncthreads();
    if (grp == 0) {
        float tot = 0.0f, cnt = 0.0f;
        for (int g = 0; g < 32; ++g) {
            tot += part[g][col];
            cnt += cnts[g];
        }
        const float rh = cnt > 0.0f ? tot / fmaxf(cnt, 1.0f) : rho;
        float term = rho * logf(rho / rh) + (1.0f - rho) * logf((1.0f - rho) / (1.0f - rh));
        for (int d = 16; d >= 1; d >>= 1) term += __shfl_xor(term, d, 64);
        stats[col] = rh;
        if (col == 0) {
            stats[32] = cnt;
            stats[33] = cnt > 0.0f ? term / (float)kKlDim : 0.0f;
        }
    }
}
// d latent[i][j] = d_kl / 32 * (-rho / rho_hat_j + (1 - rho) / (1 - rho_hat_j)) * w_i / cnt * act (1 - act)
__global__ void __launch_bounds__(256) latent_kl_bwd_kernel(const float* __restrict__ latent, const unsigned char* __restrict__ mask,
                                                            int n, float rho, int activated, const float* __restrict__ stats,
                                                            const float* __restrict__ d_kl, float* __restrict__ d_latent) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * kKlDim) return;
    const int col = (int)(idx & (kKlDim - 1));
    const long i = idx >> 5;
    const float cnt = stats[32], rh = stats[col];
    const float w = mask ? (mask[i] ? 1.0f : 0.0f) : 1.0f;
    float g = 0.0f;
    if (cnt > 0.0f && w != 0.0f) {
        const float a = 1.0f / (1.0f + expf(-latent[idx]));
        g = d_kl[0] * (1.0f / (float)kKlDim) * (-rho / rh + (1.0f - rho) / (1.0f - rh)) / fmaxf(cnt, 1.0f) * (activated ? 1.0f : a * (1.0f - a));
    }
    d_latent[idx] = g;
}

// The two L1 terms of a stage-2 step (lvis.py:164-170) over the primary rays with a hit: lvis_loss = sum |gt - pre| / (4 n_hit +
// 1e-6) on [B][4], radiance_loss = sum |gt - pre| / (12 n_hit + 1e-6) on [B][4][3], and their gradients with respect to the
// predictions (rows of rays without a hit: zero).  ONE workgroup, fixed summation order; ~35 element-wise launches before.
__global__ void __launch_bounds__(1024) stage2_loss_kernel(const float* __restrict__ gt_lvis, const float* __restrict__ pre_lvis,
                                                           const float* __restrict__ gt_rad, const float* __restrict__ pre_rad,
                                                           const unsigned char* __restrict__ hit, int n, float* __restrict__ out,
                                                           float* __restrict__ d_lvis, float* __restrict__ d_rad) {
    __shared__ float red[3][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a[3] = {0.0f, 0.0f, 0.0f};          // n_hit, sum |lvis error|, sum |radiance error|
    for (int i = threadIdx.x; i < n; i += 1024) {
        if (!hit[i]) continue;
        a[0] += 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) a[1] += fabsf(gt_lvis[i * 4 + k] - pre_lvis[i * 4 + k]);
#pragma unroll
        for (int k = 0; k < 12; ++k) a[2] += fabsf(gt_rad[i * 12 + k] - pre_rad[i * 12 + k]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int d = 32; d >= 1; d >>= 1) a[k] += __shfl_xor(a[k], d, 64);
        if (lane == 0) red[k][wave] = a[k];
    }
    __syncthreads();
    float t[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int w = 0; w < 16; ++w) t[k] += red[k][w];
    const float dl = t[0] * 4.0f + 1e-6f, dr = t[0] * 12.0f + 1e-6f;
    if (threadIdx.x == 0) {
        out[0] = t[1] / dl;
        out[1] = t[2] / dr;
        out[2] = t[0];
    }
    auto sg = [](float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); };
    for (int i = threadIdx.x; i < n; i += 1024) {
        const bool h = hit[i] != 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) d_lvis[i * 4 + k] = h ? -sg(gt_lvis[i * 4 + k] - pre_lvis[i * 4 + k]) / dl : 0.0f;
#pragma unroll
        for (int k = 0; k < 12; ++k) d_rad[i * 12 + k] = h ? -sg(gt_rad[i * 12 + k] - pre_rad[i * 12 + k]) / dr : 0.0f;
    }
}

// The image terms of the stage-3 step (mateIllu.py:152-172): w = mask x hit, rgb_loss = sum |diff w| / (sum w + 1e-5),
// psnr = 20 log10(1 / sqrt(sum diff^2 w / ((sum w + 1e-5) 3))), and d rgb_loss / d rgb.  ONE workgroup, fixed summation order
// (thread t takes rays t, t + 1024, ...; wave tree; waves in order) -- ~22 element-wise launches forward and ~8 backward before.
__global__ void __launch_bounds__(1024) stage3_loss_kernel(const float* __restrict__ rgb, const float* __restrict__ true_rgb,
                                                           const float* __restrict__ mask, const unsigned char* __restrict__ hit,
                                                           int n, float* __restrict__ out /*[3]: rgb_loss, psnr, sum w*/,
                                                           float* __restrict__ d_rgb) {
    __shared__ float red[3][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a[3] = {0.0f, 0.0f, 0.0f};          // sum w, sum |diff w|, sum diff^2 w
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float w = mask[i] * (hit[i] ? 1.0f : 0.0f);
        a[0] += w;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = rgb[i * 3 + c] - true_rgb[i * 3 + c];
            a[1] += fabsf(d * w);
            a[2] += d * d * w;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int d = 32; d >= 1; d >>= 1) a[k] += __shfl_xor(a[k], d, 64);
        if (lane == 0) red[k][wave] = a[k];
    }
    __syncthreads();
    float t[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int w = 0; w < 16; ++w) t[k] += red[k][w];
    const float denom = t[0] + 1e-5f;
    if (threadIdx.x == 0) {
        out[0] = t[1] / denom;
        out[1] = 20.0f * log10f(1.0f / sqrtf(t[2] / (denom * 3.0f)));
        out[2] = t[0];
    }
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float w = mask[i] * (hit[i] ? 1.0f : 0.0f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = (rgb[i * 3 + c] - true_rgb[i * 3 + c]) * w;
            d_rgb[i * 3 + c] = (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * w / denom;
        }
    }
}

FN_DEV float clamp01_nan(float v) { return v != v ? v : fminf(fmaxf(v, 0.0f), 1.0f); }      // torch.clamp: NaN in, NaN out

__global__ void __launch_bounds__(256) sg_combine_fwd_kernel(const float* __restrict__ sums, long n, int has_indir, float* __restrict__ rgb) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 3) return;
    const long p = i / 3;
    const int c = (int)(i - p * 3);
    const float* q = sums + p * 12 + c;
    const float env = clamp01_nan(clamp01_nan(q[0]) + clamp01_nan(q[3]));
    float ind = 0.0f;
    if (has_indir) ind = clamp01_nan(clamp01_nan(q[6]) + clamp01_nan(q[9]));
    float slope;
    rgb[i] = srgb_curve(env + ind, 2, slope);
}
__global__ void __launch_bounds__(256) sg_combine_bwd_kernel(const float* __restrict__ sums, const float* __restrict__ d_rgb, long n,
                                                             int has_indir, float* __restrict__ d_sums) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 3) return;
    const long p = i / 3;
    const int c = (int)(i - p * 3);
    const float* q = sums + p * 12 + c;
    float* dq = d_sums + p * 12 + c;
    auto in01 = [](float v) { return v >= 0.0f && v <= 1.0f; };        // torch.clamp passes the gradient on [min, max]
    const float e = clamp01_nan(q[0]) + clamp01_nan(q[3]), env = clamp01_nan(e);
    float f = 0.0f, ind = 0.0f;
    if (has_indir) {
        f = clamp01_nan(q[6]) + clamp01_nan(q[9]);
        ind = clamp01_nan(f);
    }
    float slope;
    (void)srgb_curve(env + ind, 2, slope);
    const float g = d_rgb[i] * slope;
    const float ge = in01(e) ? g : 0.0f, gi = (has_indir && in01(f)) ? g : 0.0f;
    dq[0] = in01(q[0]) ? ge : 0.0f;
    dq[3] = in01(q[3]) ? ge : 0.0f;
    dq[6] = (has_indir && in01(q[6])) ? gi : 0.0f;
    dq[9] = (has_indir && in01(q[9])) ? gi : 0.0f;
}

}  // namespace fneus

extern "C" int fneus_stage2_loss(const float* gt_lvis, const float* pre_lvis, const float* gt_rad, const float* pre_rad,
                                 const unsigned char* hit, int n, float* out, float* d_pre_lvis, float* d_pre_rad,
                                 fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (!gt_lvis || !pre_lvis || !gt_rad || !pre_rad || !hit || !out || !d_pre_lvis || !d_pre_rad || n <= 0) return -2;
    hipLaunchKernelGGL(fneus::stage2_loss_kernel, dim3(1), dim3(1024), 0, stream, gt_lvis, pre_lvis, gt_rad, pre_rad, hit, n, out,
                       d_pre_lvis, d_pre_rad);
    return fneus::launch_status();
}

extern "C" int fneus_stage3_loss(const float* rgb, const float* true_rgb, const float* mask, const unsigned char* hit, int n, float* out,
                                 float* d_rgb, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (!rgb || !true_rgb || !mask || !hit || !out || !d_rgb || n <= 0) return -2;
    hipLaunchKernelGGL(fneus::stage3_loss_kernel, dim3(1), dim3(1024), 0, stream, rgb, true_rgb, mask, hit, n, out, d_rgb);
    return fneus::launch_status();
}

extern "C" int fneus_latent_kl_fwd(const float* latent, const unsigned char* point_mask, int n, float rho, int activated, float* stats,
                                   fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (!latent || !stats || n <= 0 || !(rho > 0.0f && rho < 1.0f)) return -2;
    hipLaunchKernelGGL(fneus::latent_kl_fwd_kernel, dim3(1), dim3(1024), 0, stream, latent, point_mask, n, rho, activated, stats);
    return fneus::launch_status();
}
extern "C" int fneus_latent_kl_bwd(const float* latent, const unsigned char* point_mask, int n, float rho, int activated, const float* stats,
                                   const float* d_kl, float* d_latent, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (!latent || !stats || !d_kl || !d_latent || n <= 0) return -2;
    hipLaunchKernelGGL(fneus::latent_kl_bwd_kernel, dim3((unsigned)(((long)n * 32 + 255) / 256)), dim3(256), 0, stream, latent, point_mask,
                       n, rho, activated, stats, d_kl, d_latent);
    return fneus::launch_status();
}

extern "C" int fneus_sg_combine_fwd(const float* sums, long n, int has_indir, float* rgb, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n <= 0) return 0;
    if (!sums || !rgb) return -2;
    hipLaunchKernelGGL(fneus::sg_combine_fwd_kernel, dim3((unsigned)((n * 3 + 255) / 256)), dim3(256), 0, stream, sums, n, has_indir, rgb);
    return fneus::launch_status();
}
extern "C" int fneus_sg_combine_bwd(const float* sums, const float* d_rgb, long n, int has_indir, float* d_sums, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n <= 0) return 0;
    if (!sums || !d_rgb || !d_sums) return -2;
    hipLaunchKernelGGL(fneus::sg_combine_bwd_kernel, dim3((unsigned)((n * 3 + 255) / 256)), dim3(256), 0, stream, sums, d_rgb, n, has_indir,
                       d_sums);
    return fneus::launch_status();
}

extern "C" int fneus_srgb_fwd(const float* x, long n, int mode, float* y, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n <= 0) return 0;
    if (!x || !y || mode < 0 || mode > 3) return -2;
    hipLaunchKernelGGL(fneus::srgb_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, n, mode, y);
    return fneus::launch_status();
}
extern "C" int fneus_srgb_bwd(const float* x, const float* dy, long n, int mode, float* dx, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n <= 0) return 0;
    if (!x || !dy || !dx || mode < 0 || mode > 3) return -2;
    hipLaunchKernelGGL(fneus::srgb_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, dy, n, mode, dx);
    return fneus::launch_status();
}

// ---- the direction set of get_diffuse_visibility (reference models/inverRender.py:133-161) in one launch ---------------------
// nsamp directions around every light lobe inside a cone whose opening follows the lobe's sharpness:
//   axis = lobe / (|lobe| + 1e-6), U = norm(z x axis), V = norm(axis x U), phi_range = acos(-1.95 min_m(lambda) / lambda + 1),
//   theta = 2 pi u_theta, phi = phi_range u_phi, dir = U cos(theta) sin(phi) + V sin(theta) sin(phi) + axis cos(phi),
//   weight = exp(lambda (dir . axis - 1)).
// ~45 element-wise PyTorch launches on [128, 32, 3] tensors before.  One workgroup: the minimum over the lobes is a block reduction.
namespace fneus {

// SGS: `lobes` is the light-SG table lgtSGs [M][7] itself and `lambdas` is unused -- the lobe axis sg[0..2] / (|sg[0..2]| + 1e-6) and the
// sharpness |sg[3]| of render_with_all_sg (inverRender.py:420-421) are taken here instead of four element-wise launches before this one
template <bool SGS>
__global__ void __launch_bounds__(256) vis_sample_dirs_kernel(const float* __restrict__ lobes, const float* __restrict__ lambdas,
                                                              const float* __restrict__ u_theta, const float* __restrict__ u_phi, int M,
                                                              int S, float* __restrict__ dirs, float* __restrict__ w) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float mn = 3.4e38f;
    for (int m = tid; m < M; m += 256) mn = fminf(mn, SGS ? fabsf(lobes[m * 7 + 3]) : lambdas[m]);
    red[tid] = mn;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) red[tid] = fminf(red[tid], red[tid + s]);
        __syncthreads();
    }
    const float lam_min = red[0];
    const float tiny = 1e-6f;
    for (int i = tid; i < M * S; i += 256) {
        const int m = i / S;
        float ax[3];
        if constexpr (SGS) {
#pragma unroll
            for (int c = 0; c < 3; ++c) ax[c] = lobes[m * 7 + c];
            const float n0 = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + tiny;
#pragma unroll
            for (int c = 0; c < 3; ++c) ax[c] /= n0;
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) ax[c] = lobes[m * 3 + c];
        }
        const float na = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + tiny;
#pragma unroll
        for (int c = 0; c < 3; ++c) ax[c] /= na;
        // U = norm(z x axis) = norm((-ax1, ax0, 0)),  V = norm(axis x U)
        float U[3] = {-ax[1], ax[0], 0.0f};
        const float nu = sqrtf(U[0] * U[0] + U[1] * U[1]) + tiny;
        U[0] /= nu;
        U[1] /= nu;
        float V[3] = {ax[1] * U[2] - ax[2] * U[1], ax[2] * U[0] - ax[0] * U[2], ax[0] * U[1] - ax[1] * U[0]};
        const float nv = sqrtf(V[0] * V[0] + V[1] * V[1] + V[2] * V[2]) + tiny;
#pragma unroll
        for (int c = 0; c < 3; ++c) V[c] /= nv;
        const float lam = SGS ? fabsf(lobes[m * 7 + 3]) : lambdas[m];
        const float phi_range = acosf((-1.95f * lam_min) / lam + 1.0f);
        const float th = u_theta[i] * 2.0f * 3.14159265358979323846f, ph = u_phi[i] * phi_range;
        float st, ct, sp, cp;
        sincosf(th, &st, &ct);
        sincosf(ph, &sp, &cp);
        float d[3], dot = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d[c] = U[c] * ct * sp + V[c] * st * sp + ax[c] * cp;
            dot += d[c] * ax[c];
            dirs[(size_t)i * 3 + c] = d[c];
        }
        w[i] = expf(lam * (dot - 1.0f));
    }
}

}  // namespace fneus

extern "C" int fneus_vis_sample_dirs(const float* lobes, const float* lambdas, const float* u_theta, const float* u_phi, int n_lobes,
                                     int n_samp, float* dirs, float* weights, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_lobes <= 0 || n_samp <= 0) return 0;
    if (!lobes || !lambdas || !u_theta || !u_phi || !dirs || !weights) return -2;
    hipLaunchKernelGGL(fneus::vis_sample_dirs_kernel<false>, dim3(1), dim3(256), 0, stream, lobes, lambdas, u_theta, u_phi, n_lobes, n_samp,
                       dirs, weights);
    return fneus::launch_status();
}
extern "C" int fneus_vis_sample_dirs_sgs(const float* lgt_sgs, const float* u_theta, const float* u_phi, int n_lobes, int n_samp, float* dirs,
                                         float* weights, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_lobes <= 0 || n_samp <= 0) return 0;
    if (!lgt_sgs || !u_theta || !u_phi || !dirs || !weights) return -2;
    hipLaunchKernelGGL(fneus::vis_sample_dirs_kernel<true>, dim3(1), dim3(256), 0, stream, lgt_sgs, (const float*)nullptr, u_theta, u_phi, n_lobes,
                       n_samp, dirs, weights);
    return fneus::launch_status();
}

// ---- the inputs of EnvmapMaterialNetwork's MLPs (reference models/inverRender.py:530-545) in one launch: unit normal and view direction,
// the reflected direction, the encodings.  n = normal / (|normal| + 1e-6), v = -ray_dir / (|ray_dir| + 1e-6), r = 2 (v . n) n - v;
// enc_pts [rows][63] = embed(point, 10) (the BRDF encoder's input), x_cs [rows][90] = [embed(point, 10) | embed(r, 4)] (net_cs's input).
// Ten element-wise launches, three fneus_embed launches and a concatenation before.  Thread (row, j): j < 3 a point coordinate, j >= 3 a
// coordinate of the directions.
namespace fneus {
__global__ void __launch_bounds__(256) material_inputs_kernel(const float* __restrict__ points, const float* __restrict__ ray_dirs,
                                                              const float* __restrict__ normals, int n, float* __restrict__ n_unit,
                                                              float* __restrict__ view_dirs, float* __restrict__ enc_pts,
                                                              float* __restrict__ x_cs) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * 6) return;
    const int row = idx / 6, j = idx - row * 6;
    if (j < 3) {
        const float v = points[row * 3 + j];
        float* a = enc_pts + (size_t)row * 63;
        float* b = x_cs + (size_t)row * 90;
        a[j] = v;
        b[j] = v;
        for (int k = 0; k < 10; ++k) {
            float s, c;
            sincosf(__fmul_rn(v, (float)(1 << k)), &s, &c);
            a[3 * (1 + 2 * k) + j] = s;
            a[3 * (2 + 2 * k) + j] = c;
            b[3 * (1 + 2 * k) + j] = s;
            b[3 * (2 + 2 * k) + j] = c;
        }
        return;
    }
    const int c = j - 3;
    const float tiny = 1e-6f;
    float nr[3], rd[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        nr[q] = normals[row * 3 + q];
        rd[q] = ray_dirs[row * 3 + q];
    }
    const float nn = sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]) + tiny;
    const float nd = sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]) + tiny;
    float vd[3], dot = 0.0f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        nr[q] = nr[q] / nn;
        vd[q] = -(rd[q] / nd);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) dot += vd[q] * nr[q];
    const float ref = 2.0f * dot * nr[c] - vd[c];
    n_unit[row * 3 + c] = nr[c];
    view_dirs[row * 3 + c] = vd[c];
    float* b = x_cs + (size_t)row * 90 + 63;
    b[c] = ref;
    for (int k = 0; k < 4; ++k) {
        float s, co;
        sincosf(__fmul_rn(ref, (float)(1 << k)), &s, &co);
        b[3 * (1 + 2 * k) + c] = s;
        b[3 * (2 + 2 * k) + c] = co;
    }
}
}  // namespace fneus

extern "C" int fneus_material_inputs(const float* points, const float* ray_dirs, const float* normals, int n, float* n_unit,
                                     float* view_dirs, float* enc_pts, float* x_cs, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n <= 0) return 0;
    if (!points || !ray_dirs || !normals || !n_unit || !view_dirs || !enc_pts || !x_cs) {
        fneus::set_last_error("fneus_material_inputs: every array must be given");
        return -2;
    }
    hipLaunchKernelGGL(fneus::material_inputs_kernel, dim3((unsigned)((n * 6 + 255) / 256)), dim3(256), 0, stream, points, ray_dirs, normals, n,
                       n_unit, view_dirs, enc_pts, x_cs);
    return fneus::launch_status();
}

// ---- IndirectLight's output transform alone (models/fields.py:395-413), forward only: raw [n][L][6] -> lgtSGs [n][L][7] = (axis,
// lambda, mu).  Stage 3 evaluates the frozen network once per step: ~15 element-wise launches before. ---------------------------
namespace fneus {
__global__ void __launch_bounds__(256) indir_sgs_kernel(const float* __restrict__ raw, long n_lobes_total, float* __restrict__ sgs) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lobes_total) return;
    float o[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = raw[i * 6 + k];
    const IndirLobe L = indir_lobe(o);
    float* out = sgs + i * 7;
    out[0] = L.ct * L.sp;
    out[1] = L.st * L.sp;
    out[2] = L.cp;
    out[3] = L.lam;
    out[4] = L.mu[0];
    out[5] = L.mu[1];
    out[6] = L.mu[2];
}
}  // namespace fneus

extern "C" int fneus_indir_sgs(const float* raw, long n_lobes_total, float* sgs, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_lobes_total <= 0) return 0;
    if (!raw || !sgs) return -2;
    hipLaunchKernelGGL(fneus::indir_sgs_kernel, dim3((unsigned)((n_lobes_total + 255) / 256)), dim3(256), 0, stream, raw, n_lobes_total, sgs);
    return fneus::launch_status();
}
