// Per-ray kernels: hierarchical sampler and NeuS alpha / compositing (reference models/renderer.py).
//   K6a upsample      : NeuSRenderer.up_sample + sample_pdf(det=True)            renderer.py:152-189, 43-77
//   K6b merge         : cat_z_vals sort-merge of z (and sdf)                     renderer.py:191-205
//       sections      : dists / mid_z of render_core                              renderer.py:223-226
//   K5  composite_fwd : SDF -> alpha -> weights -> colour, eikonal sums, first sign change, inside-sphere weights
//                                                                                renderer.py:245-274, 290-293, 328-332, 360-372
//   K5  composite_bwd : hand-written adjoint of the above (SURVEY.md Appendix A reminders)
// One wavefront (64 lanes) per ray; each lane owns a contiguous chunk of <= 4 samples (n <= 256); scans are
// chunk-serial + wave-level.  All arithmetic fp32; these kernels are latency/HBM bound (< 1 % of the step).
#include "fneus_common.h"
#include "fneus_kernels.h"
#include "ray_sampler.h"

namespace fneus {

// ---------------------------------------------------------------------------------------------------------------
// K6a: new z by inverse-CDF sampling of the NeuS weights at a fixed inv_s
// ---------------------------------------------------------------------------------------------------------------
template <int PER>
__global__ void __launch_bounds__(64) upsample_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                      const float* __restrict__ z_in, const float* __restrict__ sdf_in,
                                                      int m, int k, float inv_s, float* __restrict__ z_new) {
    constexpr int MAXN = 64 * PER;
    __shared__ float zs[MAXN], ss[MAXN], cdf[MAXN + 1];
    const int ray = blockIdx.x, lane = threadIdx.x;
    float o[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = rays_o[ray * 3 + c];
        d[c] = rays_d[ray * 3 + c];
    }
    for (int i = lane; i < m; i += 64) {
        zs[i] = z_in[(size_t)ray * m + i];
        ss[i] = sdf_in[(size_t)ray * m + i];
    }
    __syncthreads();
    upsample_ray<PER>(o, d, zs, ss, cdf, m, k, inv_s, lane, z_new + (size_t)ray * k, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------
// K6b: stable rank merge of (z_old | z_new); sdf rides along when given
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) merge_kernel(const float* __restrict__ z_old, const float* __restrict__ s_old, int m,
                                                   const float* __restrict__ z_new, const float* __restrict__ s_new, int k,
                                                   float* __restrict__ z_out, float* __restrict__ s_out) {
    __shared__ float zs[MAXN];
    const int ray = blockIdx.x, lane = threadIdx.x;
    const int n = m + k;
    for (int i = lane; i < n; i += 64) zs[i] = (i < m) ? z_old[(size_t)ray * m + i] : z_new[(size_t)ray * k + (i - m)];
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
        const float z = zs[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float zj = zs[j];
            rank += (zj < z) || (zj == z && j < i);
        }
        z_out[(size_t)ray * n + rank] = z;
        if (s_out) s_out[(size_t)ray * n + rank] = (i < m) ? s_old[(size_t)ray * m + i] : s_new[(size_t)ray * k + (i - m)];
    }
}

// cat_z_vals of one up-sampling step FUSED with the up_sample of the next one (the steps of a ray depend on nothing but the
// ray: renderer.py:433-446 is a per-ray recurrence with one SDF evaluation in the middle):
//   (z_old | z_new, s_old | s_new)  -- stable rank merge -->  z_out, s_out [m + k]           (= merge_kernel)
//   up_sample(z_out, s_out, inv_s)  ----------------------->  z_next [k_next]                 (= upsample_kernel)
//   last step (z_final != null): z_final [m + k + k_next] = merge(z_out | z_next)             (= merge_kernel without sdf)
// Same arithmetic as the separate kernels (results are bit-identical, tests/test_hip_rays.py); 4 launches fewer per render.
__global__ void __launch_bounds__(64) merge_upsample_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                            const float* __restrict__ z_old, const float* __restrict__ s_old, int m,
                                                            const float* __restrict__ z_new, const float* __restrict__ s_new, int k,
                                                            float inv_s, int k_next, float* __restrict__ z_out,
                                                            float* __restrict__ s_out, float* __restrict__ z_next,
                                                            float* __restrict__ z_final, float sample_dist,
                                                            float* __restrict__ dists, float* __restrict__ mid_z) {
    __shared__ float zin[MAXN], sin_[MAXN], zs[MAXN], ss[MAXN], cdf[MAXN + 1], znx[MAXN];
    merge_upsample_ray<true>(blockIdx.x, threadIdx.x, rays_o, rays_d, z_old, s_old, m, z_new, s_new, false, k, inv_s, k_next, z_out, s_out,
                             z_next, z_final, sample_dist, dists, mid_z, zin, sin_, zs, ss, cdf, znx);
}

__global__ void __launch_bounds__(256) sections_kernel(const float* __restrict__ z, int n_rays, int n, float sample_dist,
                                                       float* __restrict__ dists, float* __restrict__ mid_z) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n_rays * n) return;
    const int i = (int)(idx % n);
    const float z0 = z[idx];
    const float dd = (i + 1 < n) ? z[idx + 1] - z0 : sample_dist;
    dists[idx] = dd;
    mid_z[idx] = z0 + dd * 0.5f;
}

// ---------------------------------------------------------------------------------------------------------------
// K5 forward
// ---------------------------------------------------------------------------------------------------------------
struct SecVals {   // per-sample quantities shared by forward and backward
    float sdf, tc, dist, pc, nc, raw, alpha, inside, relax, gn;
};

FN_DEV SecVals section_values(const float* o, const float* d, float mz, float dist, float sdf, const float* g,
                              float inv_s, float car) {
    SecVals v;
    v.sdf = sdf;
    v.dist = dist;
    v.tc = d[0] * g[0] + d[1] * g[1] + d[2] * g[2];                       // renderer.py:248
    const float ic = -(fmaxf(-v.tc * 0.5f + 0.5f, 0.0f) * (1.0f - car) + fmaxf(-v.tc, 0.0f) * car);   // :250-251
    const float en = sdf + ic * dist * 0.5f, ep = sdf - ic * dist * 0.5f;  // :255-256
    v.pc = sigmoid_acc(ep * inv_s);
    v.nc = sigmoid_acc(en * inv_s);
    v.raw = (v.pc - v.nc + 1e-5f) / (v.pc + 1e-5f);                         // :268
    v.alpha = fminf(fmaxf(v.raw, 0.0f), 1.0f);
    const float pn = pt_norm(o, d, mz);
    v.inside = pn < 1.0f ? 1.0f : 0.0f;                                     // :270-272
    v.relax = pn < 1.2f ? 1.0f : 0.0f;
    v.gn = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    return v;
}

// inv_s_mode 0: *p is inv_s itself.  inv_s_mode 1: *p is the `variance` parameter of SingleVarianceNetwork and
// inv_s = clip(exp(10 variance), 1e-6, 1e6) (fields.py:262-268, renderer.py:245); the backward kernel then returns the
// per-ray gradient with respect to `variance` (clip passes the gradient on [1e-6, 1e6]).
FN_DEV float load_inv_s(const float* p, int mode) {
    return mode ? fminf(fmaxf(expf(10.0f * *p), 1e-6f), 1e6f) : *p;
}
FN_DEV float inv_s_chain(const float* p, int mode) {
    if (!mode) return 1.0f;
    const float e = expf(10.0f * *p);
    return (e >= 1e-6f && e <= 1e6f) ? 10.0f * e : 0.0f;
}

// Background model (womask, renderer.py:350-356): when bg_alpha / bg_color [B][n + n_out] are given, inside the unit
// sphere the SDF branch is used, outside the NeRF++ background, and n_out extra background samples are appended:
//   alpha_i = alpha_i*inside_i + bg_alpha_i*(1 - inside_i)  (i < n),   alpha_i = bg_alpha_i  (n <= i < n + n_out)
__global__ void __launch_bounds__(64) composite_fwd_kernel(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ mid_z,
    const float* __restrict__ dists, const float* __restrict__ sdf, const float* __restrict__ normal,
    const float* __restrict__ rgb, const float* __restrict__ inv_s_ptr, int inv_s_mode, int n, float car_host, const float* __restrict__ car_dev,
    const float* __restrict__ bg_alpha, const float* __restrict__ bg_color, int n_out,
    float* __restrict__ weights, float* __restrict__ color, float* __restrict__ wsum, float* __restrict__ wmax,
    float* __restrict__ cdf_out, float* __restrict__ inside_out, float* __restrict__ eik /*[2][B]*/,
    int* __restrict__ min_idx_out, unsigned char* __restrict__ mask_out, float* __restrict__ wpair /*[B][2]*/,
    const float* __restrict__ back_rgb /*[back_rows][3] or NULL*/, int back_rows) {
    const float car = car_dev ? *car_dev : car_host;     // device scalar: the value can change between replays of a captured step
    const int ray = blockIdx.x, lane = threadIdx.x;
    const float inv_s = load_inv_s(inv_s_ptr, inv_s_mode);
    const bool bg = bg_alpha != nullptr;
    const int nt = n + (bg ? n_out : 0);
    float o[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = rays_o[ray * 3 + c];
        d[c] = rays_d[ray * 3 + c];
    }
    const size_t base = (size_t)ray * n, baset = (size_t)ray * nt;
    float fac[PER], fin[PER], T[PER], Tin[PER], al[PER], alpre[PER], ins[PER];
    float csum[3] = {0, 0, 0}, ws = 0.0f, wm = 0.0f, en = 0.0f, ed = 0.0f, insum = 0.0f;
    int firstneg = 1 << 30;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = lane * PER + j;
        fac[j] = 1.0f; fin[j] = 1.0f; al[j] = 0.0f; alpre[j] = 0.0f; ins[j] = 0.0f;
        if (i < n) {
            const float g[3] = {normal[(base + i) * 3], normal[(base + i) * 3 + 1], normal[(base + i) * 3 + 2]};
            const SecVals v = section_values(o, d, mid_z[base + i], dists[base + i], sdf[base + i], g, inv_s, car);
            alpre[j] = v.alpha;
            ins[j] = v.inside;
            al[j] = bg ? v.alpha * v.inside + bg_alpha[baset + i] * (1.0f - v.inside) : v.alpha;
            fin[j] = 1.0f - v.alpha * v.inside + 1e-7f;
            cdf_out[base + i] = v.pc;
            inside_out[base + i] = v.inside;
            en += v.relax * (v.gn - 1.0f) * (v.gn - 1.0f);   // :370-372
            ed += v.relax;
            insum += v.inside;
            if (v.sdf < 0.0f) firstneg = min(firstneg, i);     // first index with sign(sdf) = -1   (:290-291)
        } else if (i < nt) {
            al[j] = bg_alpha[baset + i];
        }
        if (i < nt) fac[j] = 1.0f - al[j] + 1e-7f;
    }
    excl_cumprod(fac, T, lane);
    excl_cumprod(fin, Tin, lane);
    const int idx = wave_min_i(firstneg);
    const float in_total = wave_sum(insum);
    const bool mask = (idx < n) && (idx >= 1) && (in_total > 0.0f);    // :292
    float wlo = 0.0f, whi = 0.0f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = lane * PER + j;
        if (i < nt) {
            const float w = al[j] * T[j];
            weights[baset + i] = w;
            ws += w;
            wm = fmaxf(wm, w);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float cf;
                if (i < n) cf = bg ? rgb[(base + i) * 3 + c] * ins[j] + bg_color[(baset + i) * 3 + c] * (1.0f - ins[j])
                                   : rgb[(base + i) * 3 + c];
                else cf = bg_color[(baset + i) * 3 + c];
                csum[c] += w * cf;
            }
            if (mask && i < n) {
                const float win = alpre[j] * ins[j] * Tin[j];
                if (i == idx - 1) wlo = win;
                if (i == idx) whi = win;
            }
        }
    }
    ws = wave_sum(ws);
    wm = wave_max(wm);
    en = wave_sum(en);
    ed = wave_sum(ed);
    wlo = wave_sum(wlo);
    whi = wave_sum(whi);
#pragma unroll
    for (int c = 0; c < 3; ++c) csum[c] = wave_sum(csum[c]);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // renderer.py:367-368: color + background_rgb * (1 - weights_sum), the product and the sum rounded separately as the
            // element-wise formulation does
            const float back = back_rgb ? __fmul_rn(back_rgb[(back_rows > 1 ? ray : 0) * 3 + c], __fadd_rn(1.0f, -ws)) : 0.0f;
            color[ray * 3 + c] = back_rgb ? __fadd_rn(csum[c], back) : csum[c];
        }
        wsum[ray] = ws;
        wmax[ray] = wm;
        eik[ray] = en;                       // planar: both rows are contiguous [B] vectors for the loss kernel
        eik[gridDim.x + ray] = ed;
        min_idx_out[ray] = (idx < n) ? idx : 0;
        mask_out[ray] = mask ? 1 : 0;
        wpair[ray * 2] = wlo;
        wpair[ray * 2 + 1] = whi;
    }
}

// exclusive suffix sum over the ray of per-sample values (chunk layout)
FN_DEV void excl_suffix_sum(const float (&v)[PER], float (&S)[PER], int lane) {
    // A true scan from the END of the ray: the terms decay with the transmittance, and "total - prefix" would leave every
    // late sample with the rounding error of the (much larger) early partial sums -- which the division by 1 - alpha of a
    // nearly opaque sample then amplifies (seen as 1e-8 noise on d alpha of the far background samples, where the true
    // gradient is 1e-13 ... 1e-21: 2 % of the background density bias' gradient, a sum that cancels 77 : 1).
    float loc = 0.0f;
#pragma unroll
    for (int j = PER - 1; j >= 0; --j) loc += v[j];
    float inc = loc;           // -> sum over the chunks of lanes >= lane
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_down(inc, d, 64);
        if (lane + d < 64) inc += o;
    }
    const float nxt = __shfl_down(inc, 1, 64);
    float run = lane < 63 ? nxt : 0.0f;   // sum of all chunks after this lane
#pragma unroll
    for (int j = PER - 1; j >= 0; --j) {
        S[j] = run;
        run += v[j];
    }
}

__global__ void __launch_bounds__(64) composite_bwd_kernel(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ mid_z,
    const float* __restrict__ dists, const float* __restrict__ sdf, const float* __restrict__ normal,
    const float* __restrict__ rgb, const float* __restrict__ inv_s_ptr, int inv_s_mode, int n, float car_host, const float* __restrict__ car_dev,
    const float* __restrict__ bg_alpha, const float* __restrict__ bg_color, int n_out,
    const int* __restrict__ min_idx, const unsigned char* __restrict__ mask_in,
    const float* __restrict__ d_color /*[B][3]*/, const float* __restrict__ d_wsum /*[B]*/,
    const float* __restrict__ d_weights /*[B][nt] or null*/, const float* __restrict__ d_wpair /*[B][2]*/,
    const float* __restrict__ d_eiknum /*[B]*/, float* __restrict__ d_sdf, float* __restrict__ d_normal,
    float* __restrict__ d_rgb, float* __restrict__ d_inv_s /*[B]*/, float* __restrict__ d_bg_alpha /*[B][nt]*/,
    float* __restrict__ d_bg_color /*[B][nt][3]*/,
    const float* __restrict__ back_rgb /*[back_rows][3] or NULL*/, int back_rows) {
    const float car = car_dev ? *car_dev : car_host;     // device scalar: the value can change between replays of a captured step
    const int ray = blockIdx.x, lane = threadIdx.x;
    const float inv_s = load_inv_s(inv_s_ptr, inv_s_mode);
    const bool bg = bg_alpha != nullptr;
    const int nt = n + (bg ? n_out : 0);
    float o[3], d[3], dc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = rays_o[ray * 3 + c];
        d[c] = rays_d[ray * 3 + c];
        dc[c] = d_color[ray * 3 + c];
    }
    float dws = d_wsum[ray];
    if (back_rgb) {                          // color = ... + background_rgb (1 - wsum): wsum's cotangent gets -(d_color . background_rgb)
        const float* bk = back_rgb + (back_rows > 1 ? ray : 0) * 3;
        dws -= __fadd_rn(__fadd_rn(__fmul_rn(dc[0], bk[0]), __fmul_rn(dc[1], bk[1])), __fmul_rn(dc[2], bk[2]));
    }
    const float deik = d_eiknum[ray];
    const bool mask = mask_in[ray] != 0;
    const int idx = min_idx[ray];
    const float dlo = mask ? d_wpair[ray * 2] : 0.0f, dhi = mask ? d_wpair[ray * 2 + 1] : 0.0f;
    const size_t base = (size_t)ray * n, baset = (size_t)ray * nt;
    SecVals sv[PER];
    float fac[PER], fin[PER], T[PER], Tin[PER], ww[PER], wwin[PER], S[PER], Sin[PER], wbar[PER], wbin[PER], al[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = lane * PER + j;
        fac[j] = 1.0f; fin[j] = 1.0f; al[j] = 0.0f;
        sv[j] = SecVals{};
        if (i < n) {
            const float g[3] = {normal[(base + i) * 3], normal[(base + i) * 3 + 1], normal[(base + i) * 3 + 2]};
            sv[j] = section_values(o, d, mid_z[base + i], dists[base + i], sdf[base + i], g, inv_s, car);
            al[j] = bg ? sv[j].alpha * sv[j].inside + bg_alpha[baset + i] * (1.0f - sv[j].inside) : sv[j].alpha;
            fin[j] = 1.0f - sv[j].alpha * sv[j].inside + 1e-7f;
        } else if (i < nt) {
            al[j] = bg_alpha[baset + i];
        }
        if (i < nt) fac[j] = 1.0f - al[j] + 1e-7f;
    }
    excl_cumprod(fac, T, lane);
    excl_cumprod(fin, Tin, lane);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = lane * PER + j;
        wbar[j] = 0.0f; wbin[j] = 0.0f; ww[j] = 0.0f; wwin[j] = 0.0f;
        if (i < nt) {
            const float w = al[j] * T[j];
            const float ins = (i < n) ? sv[j].inside : 0.0f;
            float cf[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (i < n) cf[c] = bg ? rgb[(base + i) * 3 + c] * ins + bg_color[(baset + i) * 3 + c] * (1.0f - ins)
                                      : rgb[(base + i) * 3 + c];
                else cf[c] = bg_color[(baset + i) * 3 + c];
            }
            float wb = dws + dc[0] * cf[0] + dc[1] * cf[1] + dc[2] * cf[2];
            if (d_weights) wb += d_weights[baset + i];
            wbar[j] = wb;
            ww[j] = wb * w;
            if (i < n) {
                const float win = sv[j].alpha * ins * Tin[j];
                const float wbi = (i == idx - 1) ? dlo : ((i == idx) ? dhi : 0.0f);
                wbin[j] = wbi;
                wwin[j] = wbi * win;
                const float kr = bg ? ins : 1.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c) d_rgb[(base + i) * 3 + c] = w * dc[c] * kr;
            }
            if (bg) {
                const float kb = (i < n) ? (1.0f - ins) : 1.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c) d_bg_color[(baset + i) * 3 + c] = w * dc[c] * kb;
            }
        }
    }
    excl_suffix_sum(ww, S, lane);
    excl_suffix_sum(wwin, Sin, lane);
    float dinv = 0.0f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = lane * PER + j;
        if (i < nt) {
            const float afin_bar = T[j] * wbar[j] - S[j] / fac[j];
            if (bg) d_bg_alpha[baset + i] = afin_bar * ((i < n) ? (1.0f - sv[j].inside) : 1.0f);
            if (i < n) {
                const SecVals& v = sv[j];
                float abar = afin_bar * (bg ? v.inside : 1.0f);
                const float abar_in = Tin[j] * wbin[j] - Sin[j] / fin[j];
                abar += v.inside * abar_in;
                const float rbar = (v.raw >= 0.0f && v.raw <= 1.0f) ? abar : 0.0f;     // clip(0,1) backward
                const float cden = v.pc + 1e-5f;
                const float pcbar = rbar * (1.0f / cden - (v.pc - v.nc + 1e-5f) / (cden * cden));
                const float ncbar = -rbar / cden;
                const float dpc = v.pc * (1.0f - v.pc), dnc = v.nc * (1.0f - v.nc);
                const float ic = -(fmaxf(-v.tc * 0.5f + 0.5f, 0.0f) * (1.0f - car) + fmaxf(-v.tc, 0.0f) * car);
                const float ep = v.sdf - ic * v.dist * 0.5f, en = v.sdf + ic * v.dist * 0.5f;
                const float epbar = pcbar * dpc * inv_s, enbar = ncbar * dnc * inv_s;
                dinv += pcbar * dpc * ep + ncbar * dnc * en;
                d_sdf[base + i] = epbar + enbar;
                const float icbar = (enbar - epbar) * v.dist * 0.5f;
                const float dic_dtc = 0.5f * (1.0f - car) * ((-v.tc * 0.5f + 0.5f) > 0.0f ? 1.0f : 0.0f) +
                                      car * ((-v.tc) > 0.0f ? 1.0f : 0.0f);
                const float tcbar = icbar * dic_dtc;
                const float g[3] = {normal[(base + i) * 3], normal[(base + i) * 3 + 1], normal[(base + i) * 3 + 2]};
                const float ek = (v.gn > 0.0f) ? deik * v.relax * 2.0f * (v.gn - 1.0f) / v.gn : 0.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c) d_normal[(base + i) * 3 + c] = tcbar * d[c] + ek * g[c];
            }
        }
    }
    dinv = wave_sum(dinv);
    if (lane == 0) d_inv_s[ray] = dinv * inv_s_chain(inv_s_ptr, inv_s_mode);
}

// Coarse depths of a batch of rays: near / far of the unit sphere (dataset.py:186-192, near_far_from_sphere), the
// n_samples uniform depths of renderer.py:393-395 and the per-ray jitter of renderer.py:405-409 in one launch
// (the PyTorch formulation is 17 element-wise kernels on [B] / [B,64] tensors).  torch.linspace is reproduced
// exactly: step * i from the start for the first half, end - step * (n - 1 - i) for the second.
__global__ void __launch_bounds__(256) ray_setup_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                        const float* __restrict__ near_in, const float* __restrict__ far_in,
                                                        const float* __restrict__ t_rand, int n_rays, int n,
                                                        float* __restrict__ z_vals) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n_rays * n) return;
    const int ray = (int)(idx / n), i = (int)(idx - (long)ray * n);
    float near, far;
    if (near_in) {
        near = near_in[ray];
        far = far_in[ray];
    } else {
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float o = rays_o[ray * 3 + c], d = rays_d[ray * 3 + c];
            a = __fadd_rn(a, __fmul_rn(d, d));
            b = __fadd_rn(b, __fmul_rn(o, d));
        }
        const float mid = __fdiv_rn(__fmul_rn(0.5f, -__fmul_rn(2.0f, b)), a);
        near = __fadd_rn(mid, -1.0f);
        far = __fadd_rn(mid, 1.0f);
    }
    const float step = __fdiv_rn(1.0f, (float)(n - 1));
    const float u = i < n / 2 ? __fmul_rn(step, (float)i) : __fadd_rn(1.0f, -__fmul_rn(step, (float)(n - 1 - i)));
    float z = __fadd_rn(near, __fmul_rn(__fadd_rn(far, -near), u));
    if (t_rand) z = __fadd_rn(z, __fdiv_rn(__fmul_rn(__fadd_rn(t_rand[ray], -0.5f), 2.0f), (float)n));
    z_vals[idx] = z;
}


// ---------------------------------------------------------------------------------------------------------------
// Stage 2 (lvis.py): first surface hit of a ray and the occlusion of a secondary ray
// ---------------------------------------------------------------------------------------------------------------
// One wavefront per ray, n <= 256 samples.
//   first hit   renderer.py:586-604 = calLvis.py:178-196: idx = first sample with sign(sdf) = -1;
//               mask = (idx exists) & (idx >= 1) & (some sample lies inside the unit sphere);
//               z_surf = (s_lo z_hi - s_hi z_lo) / (s_lo - s_hi + 1e-10) between samples idx-1 and idx; p = o + d z_surf.
//               Rays without a hit get z_surf = 0 (p = o): the callers evaluate the networks at fixed shape.
//   occlusion   calLvis.py:93-150 (compute_weight, cos_anneal_ratio = 0): sum of the NeuS weights of the samples inside
//               the unit sphere; only when normal / dists are given.
__global__ void __launch_bounds__(64) ray_hit_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                     const float* __restrict__ mid_z, const float* __restrict__ sdf,
                                                     const float* __restrict__ dists, const float* __restrict__ normal,
                                                     const unsigned char* __restrict__ inside_mask, int n, float inv_s,
                                                     unsigned char* __restrict__ mask_out, float* __restrict__ z_surf,
                                                     float* __restrict__ pts_surf, float* __restrict__ occlusion,
                                                     float* __restrict__ weights) {
    const int ray = blockIdx.x, lane = threadIdx.x;
    float o[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = rays_o[ray * 3 + c];
        d[c] = rays_d[ray * 3 + c];
    }
    const size_t base = (size_t)ray * n;
    const bool occ = normal != nullptr;
    float fac[PER], T[PER], al[PER], ins[PER];
    float insum = 0.0f;
    int firstneg = 1 << 30;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = lane * PER + j;
        fac[j] = 1.0f; al[j] = 0.0f; ins[j] = 0.0f;
        if (i < n) {
            const float s = sdf[base + i], mz = mid_z[base + i];
            if (occ) {
                const float g[3] = {normal[(base + i) * 3], normal[(base + i) * 3 + 1], normal[(base + i) * 3 + 2]};
                const SecVals v = section_values(o, d, mz, dists[base + i], s, g, inv_s, 0.0f);
                al[j] = v.alpha;
                ins[j] = v.inside;
                fac[j] = 1.0f - v.alpha + 1e-7f;
            } else {
                ins[j] = pt_norm(o, d, mz) < 1.0f ? 1.0f : 0.0f;
            }
            insum += ins[j];
            if (s < 0.0f) firstneg = min(firstneg, i);
        }
    }
    const int idx = wave_min_i(firstneg);
    const float in_total = wave_sum(insum);
    const bool inside_any = inside_mask ? inside_mask[ray] != 0 : in_total > 0.0f;
    const bool mask = (idx < n) && (idx >= 1) && inside_any;
    if (occ) {
        excl_cumprod(fac, T, lane);
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = lane * PER + j;
            if (i < n) {
                const float w = al[j] * T[j];
                acc += w * ins[j];
                if (weights) weights[base + i] = w;
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) occlusion[ray] = acc;
    }
    if (lane == 0) {
        float z = 0.0f;
        if (mask) {
            const float z_lo = mid_z[base + idx - 1], z_hi = mid_z[base + idx];
            const float s_lo = sdf[base + idx - 1], s_hi = sdf[base + idx];
            z = (s_lo * z_hi - s_hi * z_lo) / (s_lo - s_hi + 1e-10f);
        }
        mask_out[ray] = mask ? 1 : 0;
        z_surf[ray] = z;
#pragma unroll
        for (int c = 0; c < 3; ++c) pts_surf[ray * 3 + c] = o[c] + d[c] * z;
    }
}

// calLvis.py:302-320 (sample_dirs) with the draws of :351-355: S directions per surface point at polar angle
// phi = asin(0.95 u_z) from the normal and azimuth theta = 2 pi u_theta in the tangent frame built from the x axis;
// also writes the origin of every secondary ray (the surface point, repeated S times: calLvis.py:357).
__global__ void __launch_bounds__(256) sample_dirs_kernel(const float* __restrict__ surf, const float* __restrict__ normal,
                                                          const float* __restrict__ u_theta, const float* __restrict__ u_z,
                                                          long total, int S, float* __restrict__ origins,
                                                          float* __restrict__ dirs) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long pt = idx / S;
    const float tiny = 1e-6f;
    float n[3] = {normal[pt * 3], normal[pt * 3 + 1], normal[pt * 3 + 2]};
    const float nn = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) + tiny;
#pragma unroll
    for (int c = 0; c < 3; ++c) n[c] = n[c] / nn;
    // U = unit(x_axis x n), V = unit(n x U)
    float U[3] = {0.0f, -n[2], n[1]};
    const float un = sqrtf(U[1] * U[1] + U[2] * U[2]) + tiny;
    U[1] = U[1] / un;
    U[2] = U[2] / un;
    float V[3] = {n[1] * U[2] - n[2] * U[1], n[2] * U[0] - n[0] * U[2], n[0] * U[1] - n[1] * U[0]};
    const float vn = sqrtf(V[0] * V[0] + V[1] * V[1] + V[2] * V[2]) + tiny;
    const float theta = u_theta[idx] * 2.0f * 3.14159265358979323846f;
    const float phi = asinf(u_z[idx] * 0.95f);
    const float ct = cosf(theta), st = sinf(theta), cp = cosf(phi), sp = sinf(phi);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dirs[idx * 3 + c] = U[c] * ct * sp + (V[c] / vn) * st * sp + n[c] * cp;
        origins[idx * 3 + c] = surf[pt * 3 + c];
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Ray generation (models/dataset.py:115-151): pixel (x, y) -> p = K^-1[:3,:3] (x, y, 1) -> v = p / |p| -> d = R v,
// o = pose[:3, 3].  Images, masks and cameras are device resident: a training batch costs one launch and no host copy.
// ---------------------------------------------------------------------------------------------------------------
FN_DEV void pixel_ray(const float* __restrict__ Kinv /*[4][4]*/, const float* __restrict__ pose /*[4][4]*/, float x, float y,
                      float* o, float* d) {
    float p[3], v[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)     // torch.matmul of a [3,3] with a [3,1] column: products added left to right
        p[r] = __fadd_rn(__fadd_rn(__fmul_rn(Kinv[r * 4 + 0], x), __fmul_rn(Kinv[r * 4 + 1], y)), Kinv[r * 4 + 2]);
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(p[0], p[0]), __fmul_rn(p[1], p[1])), __fmul_rn(p[2], p[2])));
#pragma unroll
    for (int r = 0; r < 3; ++r) v[r] = __fdiv_rn(p[r], nrm);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        d[r] = __fadd_rn(__fadd_rn(__fmul_rn(pose[r * 4 + 0], v[0]), __fmul_rn(pose[r * 4 + 1], v[1])), __fmul_rn(pose[r * 4 + 2], v[2]));
        o[r] = pose[r * 4 + 3];
    }
}

// gen_random_rays_at (dataset.py:133-151): out [n][10] = rays_o, rays_d, rgb, mask[..., :1] of the given integer pixels
__global__ void __launch_bounds__(256) gen_random_rays_kernel(const float* __restrict__ Kinv, const float* __restrict__ pose,
                                                              const float* __restrict__ image, const float* __restrict__ mask,
                                                              int H, int W, const long long* __restrict__ px,
                                                              const long long* __restrict__ py, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long x = px[i], y = py[i];
    float o[3], d[3];
    pixel_ray(Kinv, pose, (float)x, (float)y, o, d);
    float* row = out + (size_t)i * 10;
    const size_t pix = ((size_t)y * W + (size_t)x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        row[c] = o[c];
        row[3 + c] = d[c];
        row[6 + c] = image[pix + c];
    }
    row[9] = mask[pix];
}

// gen_rays_at (dataset.py:115-131): all rays of one camera at the pixel positions tx [nx] x ty [ny] (torch.linspace of the
// caller) -> rays_o, rays_v [ny][nx][3] (the reference's transposed layout: image row major)
__global__ void __launch_bounds__(256) gen_rays_grid_kernel(const float* __restrict__ Kinv, const float* __restrict__ pose,
                                                            const float* __restrict__ tx, const float* __restrict__ ty, int nx,
                                                            int ny, float* __restrict__ rays_o, float* __restrict__ rays_v) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)nx * ny) return;
    const int iy = (int)(i / nx), ix = (int)(i - (long)iy * nx);
    float o[3], d[3];
    pixel_ray(Kinv, pose, tx[ix], ty[iy], o, d);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rays_o[i * 3 + c] = o[c];
        rays_v[i * 3 + c] = d[c];
    }
}


// ---------------------------------------------------------------------------------------------------------------
// womask background branch, render_core_outside (renderer.py:112-149): the element-wise work around the NeRF++ kernels (K7)
// ---------------------------------------------------------------------------------------------------------------
// sections of the merged depths, inverted-sphere points (p / |p|, 1 / |p|) with |p| clipped to [1, 1e10], view directions
FN_DEV float softplus1(float x) { return x > 20.0f ? x : log1pf(expf(x)); }              // F.softplus defaults (beta 1, threshold 20)

// sample i of a ray's merged depths z [n_rays][nt] -> its section length, the inverted-sphere point (p / |p|, 1 / |p|) of the
// section's mid point with |p| clipped to [1, 1e10], and the ray direction
FN_DEV void outside_point(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ z, int ray,
                          int i, int nt, float sample_dist, float* __restrict__ p4, float* __restrict__ dir3,
                          float* __restrict__ dist_out) {
    const long idx = (long)ray * nt + i;
    const float z0 = z[idx];
    const float dist = i + 1 < nt ? __fadd_rn(z[idx + 1], -z0) : sample_dist;          // :121-122
    const float mid = __fadd_rn(z0, __fmul_rn(dist, 0.5f));                              // :123
    float p[3], s = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        p[c] = __fadd_rn(rays_o[ray * 3 + c], __fmul_rn(rays_d[ray * 3 + c], mid));      // :126
        s = __fadd_rn(s, __fmul_rn(p[c], p[c]));
        dir3[c] = rays_d[ray * 3 + c];
    }
    const float dis = fminf(fmaxf(sqrtf(s), 1.0f), 1e10f);                               // :128
#pragma unroll
    for (int c = 0; c < 3; ++c) p4[c] = __fdiv_rn(p[c], dis);                            // :129
    p4[3] = __fdiv_rn(1.0f, dis);
    *dist_out = dist;
}

__global__ void __launch_bounds__(256) outside_points_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                             const float* __restrict__ z, int n_rays, int nt, float sample_dist,
                                                             float* __restrict__ pts4, float* __restrict__ dirs,
                                                             float* __restrict__ dists) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n_rays * nt) return;
    const int ray = (int)(idx / nt), i = (int)(idx - (long)ray * nt);
    outside_point(rays_o, rays_d, z, ray, i, nt, sample_dist, pts4 + idx * 4, dirs + idx * 3, dists + idx);
}

// ---- the background network only where its output is used --------------------------------------------------------------------
// render() evaluates the NeRF at all n + n_out merged depths of a ray (renderer.py:452-458) and render_core blends sample i < n
// as alpha_i inside_i + bg_alpha_i (1 - inside_i), colour likewise (renderer.py:350-356): with inside_i = 1 the background value
// is multiplied by exactly 0, forward and backward.  The depths between near and far lie inside the unit sphere for most of a
// ray (the importance samples nearly always), so ~3/4 of the evaluations are of that kind.  fneus_outside_select lists the
// samples whose background value IS used -- i >= n, or |o + d mid_i| >= 1 for the core's own section mid point (the expression
// of section_values, pt_norm: round-to-nearest operations only, so both kernels see the same number; kSelectMargin keeps a
// band below 1 in the list all the same) -- in ray-major order, and the NeRF kernels, the alpha kernels and the weight-gradient
// GEMM run over that list, its length read from device memory (no host synchronisation, fixed launch shapes).
constexpr float kSelectMargin = 1e-5f;

// pass 1, one workgroup of 256 threads per ray (nt <= 256): rank[ray][i] = position of sample i among the ray's listed samples
// or -1, cnt[ray] = how many; the unlisted samples' alpha / colour (what the compositing reads for them) are written as zeros
__global__ void __launch_bounds__(256) outside_flags_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                            const float* __restrict__ z_core, int n, int nt, float sample_dist,
                                                            int32_t* __restrict__ cnt, int32_t* __restrict__ rank,
                                                            float* __restrict__ alpha_full, float* __restrict__ rgb_full) {
    __shared__ int wsum[4];
    const int ray = blockIdx.x, i = threadIdx.x, lane = i & 63, wave = i >> 6;
    bool used = false;
    if (i < nt) {
        used = true;
        if (i < n) {
            const float z0 = z_core[(long)ray * n + i];
            const float dd = (i + 1 < n) ? z_core[(long)ray * n + i + 1] - z0 : sample_dist;      // sections_kernel
            used = !(pt_norm(rays_o + ray * 3, rays_d + ray * 3, z0 + dd * 0.5f) < 1.0f - kSelectMargin);
        }
    }
    const unsigned long long m = __ballot(used);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    if (i < nt) {
        const long idx = (long)ray * nt + i;
        rank[idx] = used ? base + before : -1;
        if (!used) {
            alpha_full[idx] = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb_full[idx * 3 + c] = 0.0f;
        }
    }
    if (i == 0) cnt[ray] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// pass 2, one workgroup per ray: the ray's first list position = sum of the counts of the rays before it; the listed samples'
// inverted-sphere points, directions, section lengths and sample indices go to their list positions
__global__ void __launch_bounds__(256) outside_compact_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                              const float* __restrict__ z_feed, int n_rays, int nt,
                                                              float sample_dist, const int32_t* __restrict__ cnt,
                                                              const int32_t* __restrict__ rank, float* __restrict__ pts4,
                                                              float* __restrict__ dirs, float* __restrict__ dists,
                                                              int32_t* __restrict__ sel, int32_t* __restrict__ count) {
    __shared__ int wsum[4];
    const int ray = blockIdx.x, i = threadIdx.x, lane = i & 63, wave = i >> 6;
    int part = 0;
    for (int b = i; b < ray; b += 256) part += cnt[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    const int first = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (ray == n_rays - 1 && i == 0) *count = first + cnt[ray];
    if (i >= nt) return;
    const long idx = (long)ray * nt + i;
    const int rk = rank[idx];
    if (rk < 0) return;
    const long k = (long)first + rk;
    sel[k] = (int32_t)idx;
    outside_point(rays_o, rays_d, z_feed, ray, i, nt, sample_dist, pts4 + k * 4, dirs + k * 3, dists + k);
}

// fneus_outside_alpha_fwd / _bwd over the list: entry k < *count is sample sel[k] of the [B][nt] arrays
__global__ void __launch_bounds__(256) outside_alpha_sel_fwd_kernel(const float* __restrict__ density, const float* __restrict__ raw,
                                                                    const float* __restrict__ dists, const int32_t* __restrict__ sel,
                                                                    const int32_t* __restrict__ count, float* __restrict__ alpha_full,
                                                                    float* __restrict__ rgb_full) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *count) return;
    const long idx = sel[k];
    alpha_full[idx] = 1.0f - expf(-softplus1(density[k]) * dists[k]);
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb_full[idx * 3 + c] = 1.0f / (1.0f + expf(-raw[k * 3 + c]));
}

__global__ void __launch_bounds__(256) outside_alpha_sel_bwd_kernel(const float* __restrict__ density, const float* __restrict__ rgb_full,
                                                                    const float* __restrict__ dists, const int32_t* __restrict__ sel,
                                                                    const int32_t* __restrict__ count,
                                                                    const float* __restrict__ d_alpha_full,
                                                                    const float* __restrict__ d_rgb_full, long cap,
                                                                    float* __restrict__ d_density, float* __restrict__ d_raw) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cap) return;
    if (k >= *count) {                                   // (never read by the backward kernels; defined all the same)
        d_density[k] = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) d_raw[k * 3 + c] = 0.0f;
        return;
    }
    const long idx = sel[k];
    const float x = density[k], dist = dists[k];
    const float dsp = x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x));                       // softplus'
    d_density[k] = d_alpha_full ? d_alpha_full[idx] * dist * expf(-softplus1(x) * dist) * dsp : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float y = rgb_full[idx * 3 + c];
        d_raw[k * 3 + c] = d_rgb_full ? d_rgb_full[idx * 3 + c] * y * (1.0f - y) : 0.0f;
    }
}


// alpha = 1 - exp(-softplus(density) * dist), colour = sigmoid(raw)                      (renderer.py:137-138)
__global__ void __launch_bounds__(256) outside_alpha_fwd_kernel(const float* __restrict__ density, const float* __restrict__ raw,
                                                                const float* __restrict__ dists, long n, float* __restrict__ alpha,
                                                                float* __restrict__ rgb) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    alpha[idx] = 1.0f - expf(-softplus1(density[idx]) * dists[idx]);
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[idx * 3 + c] = 1.0f / (1.0f + expf(-raw[idx * 3 + c]));
}

__global__ void __launch_bounds__(256) outside_alpha_bwd_kernel(const float* __restrict__ density, const float* __restrict__ rgb,
                                                                const float* __restrict__ dists, const float* __restrict__ d_alpha,
                                                                const float* __restrict__ d_rgb, long n,
                                                                float* __restrict__ d_density, float* __restrict__ d_raw) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float x = density[idx], dist = dists[idx];
    const float dsp = x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x));                       // softplus'
    d_density[idx] = d_alpha ? d_alpha[idx] * dist * expf(-softplus1(x) * dist) * dsp : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float y = rgb[idx * 3 + c];
        d_raw[idx * 3 + c] = d_rgb ? d_rgb[idx * 3 + c] * y * (1.0f - y) : 0.0f;
    }
}


// Embedder.embed (embedder.py:23-36): [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] per row, d <= 4 columns
__global__ void __launch_bounds__(256) embed_kernel(const float* __restrict__ x, long n, int d, int n_freqs, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * d) return;
    const long row = idx / d;
    const int c = (int)(idx - row * d);
    const int w = d * (1 + 2 * n_freqs);
    const float v = x[idx];
    float* o = out + row * w;
    o[c] = v;
    for (int k = 0; k < n_freqs; ++k) {
        float s, co;
        sincosf(__fmul_rn(v, (float)(1 << k)), &s, &co);
        o[d * (1 + 2 * k) + c] = s;
        o[d * (2 + 2 * k) + c] = co;
    }
}

}  // namespace fneus

using namespace fneus;

#define FN_CHECK_N(n) if ((n) > MAXN || (n) < 2) { set_last_error("samples per ray must be in [2, 256]"); return -2; }

extern "C" int fneus_upsample(const float* rays_o, const float* rays_d, const float* z, const float* sdf, int n_rays, int m,
                              int k, float inv_s, float* z_new, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0) return 0;
    if (m > 512 || m < 2) { set_last_error("fneus_upsample: samples per ray must be in [2, 512]"); return -2; }
    if (m <= 256)
        hipLaunchKernelGGL(upsample_kernel<4>, dim3(n_rays), dim3(64), 0, stream, rays_o, rays_d, z, sdf, m, k, inv_s, z_new);
    else
        hipLaunchKernelGGL(upsample_kernel<8>, dim3(n_rays), dim3(64), 0, stream, rays_o, rays_d, z, sdf, m, k, inv_s, z_new);
    return fneus::launch_status();
}

extern "C" int fneus_merge(const float* z_old, const float* s_old, int m, const float* z_new, const float* s_new, int k,
                           int n_rays, float* z_out, float* s_out, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0) return 0;
    FN_CHECK_N(m + k);
    hipLaunchKernelGGL(merge_kernel, dim3(n_rays), dim3(64), 0, stream, z_old, s_old, m, z_new, s_new, k, z_out, s_out);
    return fneus::launch_status();
}

extern "C" int fneus_merge_upsample(const float* rays_o, const float* rays_d, const float* z_old, const float* s_old, int m,
                                    const float* z_new, const float* s_new, int k, int n_rays, float inv_s, int k_next,
                                    float* z_out, float* s_out, float* z_next, float* z_final, float sample_dist, float* dists,
                                    float* mid_z, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0) return 0;
    FN_CHECK_N(m + k + (z_final ? k_next : 0));
    if (k_next < 1 || k_next > MAXN) { set_last_error("fneus_merge_upsample: k_next must be in [1, 256]"); return -2; }
    hipLaunchKernelGGL(merge_upsample_kernel, dim3(n_rays), dim3(64), 0, stream, rays_o, rays_d, z_old, s_old, m, z_new, s_new, k,
                       inv_s, k_next, z_out, s_out, z_next, z_final, sample_dist, (z_final && mid_z) ? dists : nullptr, mid_z);
    return fneus::launch_status();
}

extern "C" int fneus_sections(const float* z, int n_rays, int n, float sample_dist, float* dists, float* mid_z,
                              fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    const long total = (long)n_rays * n;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(sections_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, z, n_rays, n, sample_dist,
                       dists, mid_z);
    return fneus::launch_status();
}

extern "C" int fneus_composite_fwd(const float* rays_o, const float* rays_d, const float* mid_z, const float* dists,
                                   const float* sdf, const float* normal, const float* rgb, const float* inv_s,
                                   int inv_s_mode, int n_rays, int n, float cos_anneal_ratio, const float* cos_anneal_dev,
                                   const float* bg_alpha,
                                   const float* bg_color, int n_out, float* weights, float* color, float* wsum,
                                   float* wmax, float* cdf, float* inside, float* eik, int32_t* min_idx,
                                   unsigned char* sdf_mask, float* wpair, const float* background_rgb, int background_rows,
                                   fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0) return 0;
    FN_CHECK_N(n + (bg_alpha ? n_out : 0));
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(n_rays), dim3(64), 0, stream, rays_o, rays_d, mid_z, dists, sdf, normal,
                       rgb, inv_s, inv_s_mode, n, cos_anneal_ratio, cos_anneal_dev, bg_alpha, bg_color, n_out, weights, color, wsum, wmax, cdf, inside,
                       eik, min_idx, sdf_mask, wpair, background_rgb, background_rows);
    return fneus::launch_status();
}

extern "C" int fneus_composite_bwd(const float* rays_o, const float* rays_d, const float* mid_z, const float* dists,
                                   const float* sdf, const float* normal, const float* rgb, const float* inv_s,
                                   int inv_s_mode, int n_rays, int n, float cos_anneal_ratio, const float* cos_anneal_dev,
                                   const float* bg_alpha,
                                   const float* bg_color, int n_out, const int32_t* min_idx,
                                   const unsigned char* sdf_mask, const float* d_color, const float* d_wsum,
                                   const float* d_weights, const float* d_wpair, const float* d_eiknum, float* d_sdf,
                                   float* d_normal, float* d_rgb, float* d_inv_s, float* d_bg_alpha, float* d_bg_color,
                                   const float* background_rgb, int background_rows, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0) return 0;
    FN_CHECK_N(n + (bg_alpha ? n_out : 0));
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(n_rays), dim3(64), 0, stream, rays_o, rays_d, mid_z, dists, sdf, normal,
                       rgb, inv_s, inv_s_mode, n, cos_anneal_ratio, cos_anneal_dev, bg_alpha, bg_color, n_out, min_idx, sdf_mask, d_color, d_wsum,
                       d_weights, d_wpair, d_eiknum, d_sdf, d_normal, d_rgb, d_inv_s, d_bg_alpha, d_bg_color, background_rgb, background_rows);
    return fneus::launch_status();
}

// [B][10] batch rows (rays_o 3, rays_d 3, rgb 3, mask 1: dataset.py:133-151) -> four contiguous arrays, one launch
__global__ void __launch_bounds__(256) split_batch_kernel(const float* __restrict__ data, int n_rays, float* __restrict__ rays_o,
                                                          float* __restrict__ rays_d, float* __restrict__ rgb,
                                                          float* __restrict__ mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rays * 10) return;
    const int ray = i / 10, c = i - ray * 10;
    const float v = data[i];
    if (c < 3) rays_o[ray * 3 + c] = v;
    else if (c < 6) rays_d[ray * 3 + c - 3] = v;
    else if (c < 9) rgb[ray * 3 + c - 6] = v;
    else mask[ray] = v;
}

extern "C" int fneus_split_batch(const float* data, int n_rays, float* rays_o, float* rays_d, float* rgb, float* mask,
                                 fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0) return 0;
    if (!data || !rays_o || !rays_d || !rgb || !mask) return -2;
    hipLaunchKernelGGL(split_batch_kernel, dim3((unsigned)((n_rays * 10 + 255) / 256)), dim3(256), 0, stream, data, n_rays,
                       rays_o, rays_d, rgb, mask);
    return fneus::launch_status();
}

extern "C" int fneus_ray_setup(const float* rays_o, const float* rays_d, const float* near, const float* far,
                               const float* t_rand, int n_rays, int n_samples, float* z_vals, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    const long total = (long)n_rays * n_samples;
    if (total <= 0) return 0;
    if (n_samples < 2 || (near == nullptr) != (far == nullptr) || (!near && (!rays_o || !rays_d))) return -2;
    hipLaunchKernelGGL(ray_setup_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, rays_o, rays_d, near, far,
                       t_rand, n_rays, n_samples, z_vals);
    return fneus::launch_status();
}

extern "C" int fneus_ray_hit(const float* rays_o, const float* rays_d, const float* mid_z, const float* sdf, const float* dists,
                             const float* normal, const unsigned char* inside_mask, int n_rays, int n, float inv_s,
                             unsigned char* sdf_mask, float* z_surf, float* pts_surf, float* occlusion, float* weights,
                             fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0) return 0;
    FN_CHECK_N(n);
    if ((normal != nullptr) != (dists != nullptr) || (normal != nullptr && occlusion == nullptr)) {
        set_last_error("fneus_ray_hit: normal, dists and occlusion go together");
        return -2;
    }
    hipLaunchKernelGGL(ray_hit_kernel, dim3(n_rays), dim3(64), 0, stream, rays_o, rays_d, mid_z, sdf, dists, normal, inside_mask, n,
                       inv_s, sdf_mask, z_surf, pts_surf, occlusion, weights);
    return fneus::launch_status();
}

extern "C" int fneus_sample_dirs(const float* surf, const float* normal, const float* u_theta, const float* u_z, int n_pts,
                                 int n_dirs, float* origins, float* dirs, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    const long total = (long)n_pts * n_dirs;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(sample_dirs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, surf, normal, u_theta, u_z,
                       total, n_dirs, origins, dirs);
    return fneus::launch_status();
}

extern "C" int fneus_gen_random_rays(const float* intrinsics_inv, const float* pose, const float* image, const float* mask, int H,
                                     int W, const long long* pixels_x, const long long* pixels_y, int n_rays, float* out,
                                     fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0) return 0;
    if (!intrinsics_inv || !pose || !image || !mask || !pixels_x || !pixels_y || !out || H <= 0 || W <= 0) {
        set_last_error("fneus_gen_random_rays: null argument");
        return -2;
    }
    hipLaunchKernelGGL(gen_random_rays_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, stream, intrinsics_inv, pose, image, mask, H,
                       W, pixels_x, pixels_y, n_rays, out);
    return fneus::launch_status();
}

extern "C" int fneus_gen_rays_grid(const float* intrinsics_inv, const float* pose, const float* tx, const float* ty, int nx, int ny,
                                   float* rays_o, float* rays_v, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    const long total = (long)nx * ny;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(gen_rays_grid_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, intrinsics_inv, pose, tx, ty,
                       nx, ny, rays_o, rays_v);
    return fneus::launch_status();
}

// z_vals_outside of NeuSRenderer.render (renderer.py:397-400, 411-419) in one launch: the n_out depths of the inverted-sphere
// parametrisation per ray.  lin_k = torch.linspace(1e-3, 1 - 1/(n_out + 1), n_out)[k] (torch's own evaluation order: from the start
// for the lower half, from the end for the upper one), jittered inside its cell [lower_k, upper_k] by u[b][k] when u is given,
// then z[b][k] = far_b / t[n_out - 1 - k] + 1 / n_samples with far from the rays' unit-sphere bounds (dataset.py:186-192) unless
// an explicit far is passed.
__global__ void __launch_bounds__(256) outside_z_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                        const float* __restrict__ far_in, const float* __restrict__ u, int n_rays,
                                                        int n_out, int n_samples, float* __restrict__ z) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_rays * n_out) return;
    const int b = (int)(i / n_out), k = (int)(i - (long)b * n_out);
    const int src = n_out - 1 - k;                    // torch.flip
    const float start = 1e-3f, end = 1.0f - 1.0f / ((float)n_out + 1.0f);
    const float step = (end - start) / (float)(n_out - 1);
    auto lin = [&](int j) { return j < n_out / 2 ? start + step * (float)j : end - step * (float)(n_out - 1 - j); };
    float t = lin(src);
    if (u) {
        const float lo = src == 0 ? lin(0) : 0.5f * (lin(src) + lin(src - 1));
        const float hi = src == n_out - 1 ? lin(n_out - 1) : 0.5f * (lin(src + 1) + lin(src));
        t = lo + (hi - lo) * u[(long)b * n_out + src];
    }
    float far;
    if (far_in) {
        far = far_in[b];
    } else {
        const float ox = rays_o[b * 3], oy = rays_o[b * 3 + 1], oz = rays_o[b * 3 + 2];
        const float dx = rays_d[b * 3], dy = rays_d[b * 3 + 1], dz = rays_d[b * 3 + 2];
        const float a = dx * dx + dy * dy + dz * dz;
        const float bb = 2.0f * (ox * dx + oy * dy + oz * dz);
        far = 0.5f * (-bb) / a + 1.0f;
    }
    z[i] = far / t + 1.0f / (float)n_samples;
}

extern "C" int fneus_outside_z(const float* rays_o, const float* rays_d, const float* far, const float* u, int n_rays, int n_out,
                               int n_samples, float* z, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    const long total = (long)n_rays * n_out;
    if (total <= 0) return 0;
    if (n_out < 2 || n_samples <= 0 || (!far && (!rays_o || !rays_d))) return -2;
    hipLaunchKernelGGL(outside_z_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, rays_o, rays_d, far, u, n_rays,
                       n_out, n_samples, z);
    return fneus::launch_status();
}

extern "C" int fneus_outside_points(const float* rays_o, const float* rays_d, const float* z, int n_rays, int nt, float sample_dist,
                                    float* pts4, float* dirs, float* dists, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    const long total = (long)n_rays * nt;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(outside_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, rays_o, rays_d, z, n_rays, nt,
                       sample_dist, pts4, dirs, dists);
    return fneus::launch_status();
}

extern "C" int fneus_outside_select(const float* rays_o, const float* rays_d, const float* z_core, const float* z_feed, int n_rays,
                                    int n, int nt, float sample_dist, int32_t* work, float* pts4, float* dirs, float* dists,
                                    int32_t* sel, int32_t* count, float* alpha_full, float* rgb_full, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0) return 0;
    if (nt > 256 || n < 0 || n > nt || !rays_o || !rays_d || !z_core || !z_feed || !work || !pts4 || !dirs || !dists || !sel || !count ||
        !alpha_full || !rgb_full) {
        fneus::set_last_error("fneus_outside_select: needs n <= nt <= 256 and every buffer");
        return -2;
    }
    int32_t* cnt = work;
    int32_t* rank = work + n_rays;
    hipLaunchKernelGGL(outside_flags_kernel, dim3(n_rays), dim3(256), 0, stream, rays_o, rays_d, z_core, n, nt, sample_dist, cnt, rank,
                       alpha_full, rgb_full);
    hipLaunchKernelGGL(outside_compact_kernel, dim3(n_rays), dim3(256), 0, stream, rays_o, rays_d, z_feed, n_rays, nt, sample_dist, cnt,
                       rank, pts4, dirs, dists, sel, count);
    return fneus::launch_status();
}

extern "C" int fneus_outside_alpha_sel_fwd(const float* density, const float* rgb_raw, const float* dists, const int32_t* sel,
                                           const int32_t* count, long cap, float* alpha_full, float* rgb_full,
                                           fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (cap <= 0) return 0;
    hipLaunchKernelGGL(outside_alpha_sel_fwd_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, stream, density, rgb_raw, dists,
                       sel, count, alpha_full, rgb_full);
    return fneus::launch_status();
}

extern "C" int fneus_outside_alpha_sel_bwd(const float* density, const float* rgb_full, const float* dists, const int32_t* sel,
                                           const int32_t* count, long cap, const float* d_alpha_full, const float* d_rgb_full,
                                           float* d_density, float* d_rgb_raw, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (cap <= 0) return 0;
    hipLaunchKernelGGL(outside_alpha_sel_bwd_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, stream, density, rgb_full, dists,
                       sel, count, d_alpha_full, d_rgb_full, cap, d_density, d_rgb_raw);
    return fneus::launch_status();
}

extern "C" int fneus_outside_alpha_fwd(const float* density, const float* rgb_raw, const float* dists, long n, float* alpha,
                                       float* rgb, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n <= 0) return 0;
    hipLaunchKernelGGL(outside_alpha_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, density, rgb_raw, dists, n,
                       alpha, rgb);
    return fneus::launch_status();
}

extern "C" int fneus_outside_alpha_bwd(const float* density, const float* rgb, const float* dists, const float* d_alpha,
                                       const float* d_rgb, long n, float* d_density, float* d_rgb_raw, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n <= 0) return 0;
    hipLaunchKernelGGL(outside_alpha_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, density, rgb, dists, d_alpha,
                       d_rgb, n, d_density, d_rgb_raw);
    return fneus::launch_status();
}

extern "C" int fneus_embed(const float* x, long n_rows, int d, int n_freqs, float* out, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rows <= 0) return 0;
    if (!x || !out || d < 1 || n_freqs < 0 || n_freqs > 24) {
        set_last_error("fneus_embed: bad argument");
        return -2;
    }
    const long total = n_rows * d;
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, n_rows, d, n_freqs, out);
    return fneus::launch_status();
}
