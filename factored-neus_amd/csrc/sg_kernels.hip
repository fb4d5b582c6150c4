// Stage-3 spherical-Gaussian rendering (reference models/inverRender.py:314-449 render_with_sg with :83-125, :264-283
// lambda_trick / hemisphere_int / integrate_rgb) as ONE forward and ONE backward launch.
//
// Per surface point and light lobe the reference evaluates a chain of ~80 element-wise tensor ops on [points, lobes, 3]
// tensors (twice: 128 direct lobes with visibility, 24 indirect lobes without), ~600 launches forward and ~1500 with
// autograd's backward.  Here a wavefront owns a point, its lanes stride over the lobes, and the whole chain runs in
// registers.  The backward pass recomputes the chain with forward-mode dual numbers and contracts the partials with the incoming
// cotangents: no hand-derived adjoint of the chain to keep in sync with the forward code, the same templated function
// (lobe_factors) serves both.  Round 5: the colour channels' albedos and amplitudes multiply the chain's result, so their partials
// are read off its value and only the roughness and the lobe's axis / sharpness carry tangents -- one pass of 5 where there were
// two of 7 (lobe_adjoint) -- and the light table's gradient leaves as one atomic per workgroup, not per point: 155 -> 27 us per step.
//
// Outputs are the lobe SUMS before integrate_rgb's clamp (inverRender.py:277): [n][4][3] = direct specular, direct diffuse,
// indirect specular, indirect diffuse; the clamps, the tone mapping and the losses stay with the caller.
#include <math.h>
#include "fneus_common.h"
#include "fneus_kernels.h"

namespace fneus {

constexpr float kTiny = 1e-6f;      // inverRender.py:12 TINY_NUMBER
constexpr float kPi = 3.14159265358979323846f;

// ---- forward-mode dual numbers; N = 0 degenerates to plain floats -----------------------------------------------------------
template <int N>
struct Dual {
    float v;
    float d[N > 0 ? N : 1];
};
template <int N> FN_DEV Dual<N> mk(float v) {
    Dual<N> r;
    r.v = v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = 0.0f;
    return r;
}
template <int N> FN_DEV Dual<N> var(float v, int idx) {      // an independent variable: d/d(idx) = 1
    Dual<N> r = mk<N>(v);
    if (idx >= 0 && idx < N) r.d[idx] = 1.0f;
    return r;
}
#define FN_DUAL_UNARY(NAME, VAL, DER)                                      \
    template <int N> FN_DEV Dual<N> NAME(const Dual<N>& a) {               \
        Dual<N> r;                                                         \
        const float val = (VAL), der = (DER);                              \
        r.v = val;                                                         \
        _Pragma("unroll") for (int i = 0; i < N; ++i) r.d[i] = der * a.d[i]; \
        return r;                                                          \
    }
template <int N> FN_DEV Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
template <int N> FN_DEV Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
template <int N> FN_DEV Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
template <int N> FN_DEV Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r;
    const float inv = 1.0f / b.v;
    r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
template <int N> FN_DEV Dual<N> operator+(const Dual<N>& a, float b) { Dual<N> r = a; r.v += b; return r; }
template <int N> FN_DEV Dual<N> operator-(const Dual<N>& a, float b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> FN_DEV Dual<N> operator*(const Dual<N>& a, float b) {
    Dual<N> r;
    r.v = a.v * b;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b;
    return r;
}
template <int N> FN_DEV Dual<N> operator/(const Dual<N>& a, float b) { return a * (1.0f / b); }
template <int N> FN_DEV Dual<N> operator+(float a, const Dual<N>& b) { return b + a; }
template <int N> FN_DEV Dual<N> operator*(float a, const Dual<N>& b) { return b * a; }
template <int N> FN_DEV Dual<N> operator-(float a, const Dual<N>& b) { return mk<N>(a) - b; }
template <int N> FN_DEV Dual<N> operator/(float a, const Dual<N>& b) { return mk<N>(a) / b; }
FN_DUAL_UNARY(dsqrt, sqrtf(a.v), 0.5f / sqrtf(a.v))
FN_DUAL_UNARY(dexp, expf(a.v), expf(a.v))
FN_DUAL_UNARY(dabs, fabsf(a.v), (a.v > 0.0f ? 1.0f : (a.v < 0.0f ? -1.0f : 0.0f)))
// torch.clamp: the gradient passes where the value lies inside the closed range
FN_DUAL_UNARY(dclamp_min0, fmaxf(a.v, 0.0f), (a.v >= 0.0f ? 1.0f : 0.0f))
FN_DUAL_UNARY(dclamp_max0, fminf(a.v, 0.0f), (a.v <= 0.0f ? 1.0f : 0.0f))
FN_DUAL_UNARY(dclamp_min_tiny, fmaxf(a.v, kTiny), (a.v >= kTiny ? 1.0f : 0.0f))
template <int N> FN_DEV Dual<N> dmin(const Dual<N>& a, const Dual<N>& b) {      // torch.min(a, b): ties share the gradient
    if (a.v < b.v) return a;
    if (b.v < a.v) return b;
    return (a + b) * 0.5f;
}

template <int N> struct Vec3 { Dual<N> x, y, z; };
template <int N> FN_DEV Dual<N> dot3(const Vec3<N>& a, const Vec3<N>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <int N> FN_DEV Vec3<N> unit_tiny(const Vec3<N>& a) {       // norm_axis: x / (|x| + TINY)
    const Dual<N> inv = 1.0f / (dsqrt(dot3(a, a)) + kTiny);
    return Vec3<N>{a.x * inv, a.y * inv, a.z * inv};
}
template <int N> FN_DEV Vec3<N> cvec(const float (&v)[3]) { return Vec3<N>{mk<N>(v[0]), mk<N>(v[1]), mk<N>(v[2])}; }

// inverRender.py:83-103 (mus handled by the caller: final_mu = mu1 * mu2 * exp(diff))
template <int N>
FN_DEV void lambda_trick(const Vec3<N>& lobe1, const Dual<N>& lam1, const Vec3<N>& lobe2, const Dual<N>& lam2, Vec3<N>& lobes,
                         Dual<N>& lam3, Dual<N>& expdiff) {
    const Dual<N> ratio = lam1 / (lam2 + kTiny);
    const Vec3<N> l1 = unit_tiny(lobe1), l2 = unit_tiny(lobe2);
    const Dual<N> d = dot3(l1, l2);
    const Dual<N> tmp = dmin(dsqrt(ratio * ratio + 1.0f + 2.0f * ratio * d + kTiny), ratio + 1.0f);
    lam3 = lam2 * tmp;
    const Dual<N> c1 = ratio / (tmp + kTiny), c2 = 1.0f / (tmp + kTiny);
    lobes = Vec3<N>{c1 * l1.x + c2 * l2.x, c1 * l1.y + c2 * l2.y, c1 * l1.z + c2 * l2.z};
    expdiff = dexp(lam2 * (tmp - ratio - 1.0f));
}

// inverRender.py:106-125
template <int N>
FN_DEV Dual<N> hemisphere_int(const Dual<N>& lambda_val, const Dual<N>& cos_beta) {
    const Dual<N> lam = dclamp_min_tiny(lambda_val);
    const Dual<N> inv = 1.0f / (lam + kTiny);
    const Dual<N> t = dsqrt(lam + kTiny) * (1.6988f + 10.8438f * inv) / (1.0f + 6.2201f * inv + 10.2415f * inv * inv + kTiny);
    const Dual<N> inv_a = dexp(mk<N>(0.0f) - t);
    const Dual<N> inv_b = dexp(mk<N>(0.0f) - t * dclamp_min0(cos_beta));
    const Dual<N> s1 = (1.0f - inv_a * inv_b) / (1.0f - inv_a + inv_b - inv_a * inv_b + kTiny);
    const Dual<N> b = dexp(t * dclamp_max0(cos_beta));
    const Dual<N> s2 = (b - inv_a) / ((1.0f - inv_a) * (b + 1.0f) + kTiny);
    const Dual<N> s = cos_beta.v >= 0.0f ? s1 : s2;
    const Dual<N> two_pi_over = (2.0f * kPi) / lam;
    const Dual<N> e1 = dexp(mk<N>(0.0f) - lam), e2 = dexp(mk<N>(0.0f) - 2.0f * lam);
    const Dual<N> a_b = two_pi_over * (e1 - e2), a_u = two_pi_over * (1.0f - e1);
    return a_b * (1.0f - s) + a_u * s;
}

// the per-channel-independent factor of integrate_rgb (inverRender.py:264-275): rgb_c = mu_c * W
template <int N>
FN_DEV Dual<N> integrate_weight(const float (&nrm)[3], const Vec3<N>& lobes, const Dual<N>& lams) {
    const float mu_cos = 32.7080f, lambda_cos = 0.0315f, alpha_cos = 31.7003f;
    const Vec3<N> n = cvec<N>(nrm);
    Vec3<N> lobe_p;
    Dual<N> lam_p, ed;
    lambda_trick(n, mk<N>(lambda_cos), lobes, lams, lobe_p, lam_p, ed);
    const Dual<N> dot1 = dclamp_min0(dot3(lobe_p, n));
    const Dual<N> dot2 = dclamp_min0(dot3(lobes, n));
    return mu_cos * ed * hemisphere_int(lam_p, dot1) - alpha_cos * hemisphere_int(lams, dot2);
}

struct PointConst {          // per point, parameter independent
    float n[3], v[3], warp[3], v_dot_lobe, fresnel, dot1, dot2;
};
FN_DEV PointConst point_consts(const float* __restrict__ normal, const float* __restrict__ view, float f0) {
    PointConst p;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        p.n[c] = normal[c];
        p.v[c] = view[c];
    }
    p.v_dot_lobe = fmaxf(p.n[0] * p.v[0] + p.n[1] * p.v[1] + p.n[2] * p.v[2], 0.0f);          // :353-355
    float w[3], nw = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        w[c] = 2.0f * p.v_dot_lobe * p.n[c] - p.v[c];
        nw += w[c] * w[c];
    }
    nw = sqrtf(nw) + kTiny;
    float hv[3], nh = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        p.warp[c] = w[c] / nw;                                                               // :356-357
        hv[c] = p.warp[c] + p.v[c];
        nh += hv[c] * hv[c];
    }
    nh = sqrtf(nh) + kTiny;
    float vdh = 0.0f, d1 = 0.0f, d2 = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        vdh += p.v[c] * (hv[c] / nh);
        d1 += p.warp[c] * p.n[c];
        d2 += p.v[c] * p.n[c];
    }
    vdh = fmaxf(vdh, 0.0f);                                                                  // :363-365
    p.fresnel = f0 + (1.0f - f0) * exp2f(-(5.55473f * vdh + 6.8316f) * vdh);                 // :368
    p.dot1 = fmaxf(d1, 0.0f);                                                                // :370-375
    p.dot2 = fmaxf(d2, 0.0f);
    return p;
}

// one (point, lobe) pair, the factors every colour channel shares: they depend on the roughness r and on the lobe's axis and
// sharpness sg[0..3] only
template <int N>
struct LobeFactors {
    Dual<N> brdf_mu, moi, ed, w_spec, w_diff;
};
template <int N>
FN_DEV LobeFactors<N> lobe_factors(const PointConst& pc, const Dual<N>& r, const Dual<N> (&sg)[7]) {
    LobeFactors<N> f;
    const Vec3<N> raw{sg[0], sg[1], sg[2]};
    const Dual<N> inv_len = 1.0f / (dsqrt(dot3(raw, raw)) + kTiny);                          // :334
    const Vec3<N> lobe{raw.x * inv_len, raw.y * inv_len, raw.z * inv_len};
    const Dual<N> lam = dabs(sg[3]);                                                         // :335
    const Dual<N> inv_r4 = 2.0f / (r * r * r * r);                                           // :347
    const Dual<N> warp_lam = inv_r4 / (4.0f * pc.v_dot_lobe + kTiny);                        // :358
    const Dual<N> k = (r + 1.0f) * (r + 1.0f) / 8.0f;                                        // :376
    const Dual<N> g1 = pc.dot1 / (pc.dot1 * (1.0f - k) + k + kTiny);
    const Dual<N> g2 = pc.dot2 / (pc.dot2 * (1.0f - k) + k + kTiny);
    f.moi = pc.fresnel * (g1 * g2) / (4.0f * pc.dot1 * pc.dot2 + kTiny);                     // :382
    f.brdf_mu = inv_r4 / kPi;                                                                // :349
    const Vec3<N> warp = cvec<N>(pc.warp);
    Vec3<N> fl;
    Dual<N> fla;
    lambda_trick(lobe, lam, warp, warp_lam, fl, fla, f.ed);                                  // :411-412
    f.w_spec = integrate_weight(pc.n, fl, fla);                                              // :415
    f.w_diff = integrate_weight(pc.n, lobe, lam);                                            // :433
    return f;
}

// one (point, lobe) pair: spec[c], diff[c] contributions.  mat = (roughness, diffuse albedo[3], specular albedo[3]); sg = the
// lobe's 7 parameters; vis = the lobe's visibility at the point (1 for the indirect lobes).
template <int N>
FN_DEV void lobe_terms(const PointConst& pc, const Dual<N> (&mat)[7], const Dual<N> (&sg)[7], float vis, Dual<N> (&spec)[3],
                       Dual<N> (&diff)[3]) {
    const LobeFactors<N> f = lobe_factors<N>(pc, mat[0], sg);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const Dual<N> mu = dabs(sg[4 + c]) * vis;                                            // :336, :402 / :407
        spec[c] = mu * (mat[4 + c] * f.brdf_mu * f.moi) * f.ed * f.w_spec;                   // :383, :100, :275
        diff[c] = mu * (mat[1 + c] / kPi) * f.w_diff;                                        // :428-431
    }
}

// The adjoint of one pair for the cotangents cs = (spec[3], diff[3]).  The colour channels enter as products --
//   spec_c = |sg_{4+c}| vis * mat_{4+c} * S,  diff_c = |sg_{4+c}| vis * mat_{1+c} * D,  S = brdf_mu moi ed w_spec,  D = w_diff / pi --
// so the six albedos and the three amplitudes need no tangent: their partials are the other factors.  What is left are the
// roughness and the lobe's axis and sharpness: ONE pass of the shared chain with NT tangents (5 for a direct lobe, 1 for an
// indirect one, whose parameters are constants) where round 4 ran two passes of 7.
template <int NT>
FN_DEV void lobe_adjoint(const PointConst& pc, const float (&m)[7], const float (&s)[7], float vis, const float* cs, float (&gm)[7],
                         float (&gs)[7]) {
    Dual<NT> sg[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) sg[i] = var<NT>(s[i], i < 4 ? 1 + i : -1);
    const LobeFactors<NT> f = lobe_factors<NT>(pc, var<NT>(m[0], 0), sg);
    const Dual<NT> S = f.brdf_mu * f.moi * f.ed * f.w_spec, D = f.w_diff / kPi;
    float A = 0.0f, B = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float mu = fabsf(s[4 + c]) * vis;
        A += cs[c] * mu * m[4 + c];
        B += cs[3 + c] * mu * m[1 + c];
        gm[1 + c] += cs[3 + c] * mu * D.v;
        gm[4 + c] += cs[c] * mu * S.v;
        const float sgn = s[4 + c] > 0.0f ? 1.0f : (s[4 + c] < 0.0f ? -1.0f : 0.0f);
        gs[4 + c] = sgn * vis * (cs[c] * m[4 + c] * S.v + cs[3 + c] * m[1 + c] * D.v);
    }
    gm[0] += A * S.d[0] + B * D.d[0];
#pragma unroll
    for (int i = 0; i < 4; ++i) gs[i] = NT > 1 ? A * S.d[NT > 1 ? 1 + i : 0] + B * D.d[NT > 1 ? 1 + i : 0] : 0.0f;
}

// The point's material (roughness, diffuse albedo[3], specular albedo[3]).  HEADS: straight from the outputs of the two MLP heads --
// `mat` is the BRDF decoder's [n][4] = (diffuse albedo rgb, raw roughness), `cs` net_cs's [n]: roughness = 0.9 raw + 0.09
// (inverRender.py:557), the specular albedo is cs in all three channels (:560) -- instead of the [n][7] table five element-wise
// launches assembled (and five took apart again in the backward).
template <bool HEADS>
FN_DEV void load_material(const float* __restrict__ mat, const float* __restrict__ cs, int pt, float (&m)[7]) {
    if constexpr (HEADS) {
        m[0] = __fadd_rn(__fmul_rn(mat[pt * 4 + 3], 0.9f), 0.09f);        // (two roundings, like the two tensor ops: no fma)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            m[1 + c] = mat[pt * 4 + c];
            m[4 + c] = cs[pt];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 7; ++i) m[i] = mat[pt * 7 + i];
    }
}
// gradient of the material: gm [7] -> d_mat (+ d_cs); ADD: on top of what is there (written by the same lane before)
template <bool HEADS, bool ADD>
FN_DEV void store_material_grad(const float (&gm)[7], int pt, float* __restrict__ d_mat, float* __restrict__ d_cs) {
    if constexpr (HEADS) {
        const float g[4] = {gm[1], gm[2], gm[3], 0.9f * gm[0]};
#pragma unroll
        for (int c = 0; c < 4; ++c) d_mat[pt * 4 + c] = ADD ? d_mat[pt * 4 + c] + g[c] : g[c];
        const float gc = gm[4] + gm[5] + gm[6];
        d_cs[pt] = ADD ? d_cs[pt] + gc : gc;
    } else {
#pragma unroll
        for (int i = 0; i < 7; ++i) d_mat[pt * 7 + i] = ADD ? d_mat[pt * 7 + i] + gm[i] : gm[i];
    }
}

// ---- kernels: one wavefront per point, lanes stride over the direct (M) and indirect (L) lobes ------------------------------
template <bool HEADS>
__global__ void __launch_bounds__(64) sg_render_fwd_kernel(const float* __restrict__ lgt /*[M][7]*/, const float* __restrict__ ind /*[n][L][7]*/,
                                                           const float* __restrict__ vis /*[M][n]*/, const float* __restrict__ normal,
                                                           const float* __restrict__ view, const float* __restrict__ mat /*[n][7]*/,
                                                           const float* __restrict__ cs, int n, int M, int L, float f0,
                                                           float* __restrict__ out /*[n][4][3]*/) {
    const int pt = blockIdx.x, lane = threadIdx.x;
    const PointConst pc = point_consts(normal + pt * 3, view + pt * 3, f0);
    float mf[7];
    load_material<HEADS>(mat, cs, pt, mf);
    Dual<0> m[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) m[i] = mk<0>(mf[i]);
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
    for (int j = lane; j < M + L; j += 64) {
        const bool direct = j < M;
        const float* src = direct ? lgt + j * 7 : ind + ((size_t)pt * L + (j - M)) * 7;
        Dual<0> sg[7], spec[3], diff[3];
#pragma unroll
        for (int i = 0; i < 7; ++i) sg[i] = mk<0>(src[i]);
        lobe_terms<0>(pc, m, sg, direct ? vis[(size_t)j * n + pt] : 1.0f, spec, diff);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            acc[(direct ? 0 : 6) + c] += spec[c].v;
            acc[(direct ? 3 : 9) + c] += diff[c].v;
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        float v = acc[i];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
        if (lane == 0) out[pt * 12 + i] = v;
    }
}

// Backward.  A workgroup of 4 waves owns kSgBwdPts points (a wave takes every fourth of them), a lane the lobes lane, lane + 64, ...
// The light table's gradient is summed over the workgroup's points in registers, over its waves in LDS (fixed order) and leaves as
// ONE atomic per (lobe, parameter) and workgroup: with an atomic per (point, lobe, parameter) -- 458 752 of them on 896 addresses
// for 512 points -- the launch took 155 us, eleven times the forward (the dual-number arithmetic was never the cost: one pass
// of 5 tangents instead of two of 7 left it at 159).
#ifndef FNEUS_SG_BWD_PTS
#define FNEUS_SG_BWD_PTS 4          // MEASURED (512 points, 128 + 24 lobes): 16 points per workgroup 80 us, 8: 44, 4: 27
#endif
constexpr int kSgBwdPts = FNEUS_SG_BWD_PTS;
constexpr int kSgBwdSlots = 4;             // lobes per lane and sweep: 256 direct lobes per sweep (the reference has 128)
template <bool HEADS>
__global__ void __launch_bounds__(256) sg_render_bwd_kernel(const float* __restrict__ lgt, const float* __restrict__ ind,
                                                            const float* __restrict__ vis, const float* __restrict__ normal,
                                                            const float* __restrict__ view, const float* __restrict__ mat,
                                                            const float* __restrict__ cs, int n, int M, int L, float f0,
                                                            const float* __restrict__ d_out /*[n][4][3]*/, float* __restrict__ d_mat /*[n][7]*/,
                                                            float* __restrict__ d_cs, float* __restrict__ d_lgt /*[M][7], atomics*/) {
    __shared__ float red[4][kSgBwdSlots * 64 * 7];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt0 = blockIdx.x * kSgBwdPts;
    // ---- the material parameters' gradient and the indirect lobes: a point at a time, lanes stride over all lobes
    for (int q = wave; q < kSgBwdPts; q += 4) {
        const int pt = pt0 + q;
        if (pt >= n) break;
        const PointConst pc = point_consts(normal + pt * 3, view + pt * 3, f0);
        float co[12], gm[7], m[7];
#pragma unroll
        for (int i = 0; i < 12; ++i) co[i] = d_out[pt * 12 + i];
#pragma unroll
        for (int i = 0; i < 7; ++i) gm[i] = 0.0f;
        load_material<HEADS>(mat, cs, pt, m);
        for (int j = lane; j < L; j += 64) {            // constants (IndirectLight is frozen): the material parameters' share only
            float sg[7], gs[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) sg[i] = ind[((size_t)pt * L + j) * 7 + i];
            lobe_adjoint<1>(pc, m, sg, 1.0f, co + 6, gm, gs);
        }
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) gm[i] += __shfl_xor(gm[i], s, 64);
        if (lane == 0) store_material_grad<HEADS, false>(gm, pt, d_mat, d_cs);
    }
    // (d_mat of a wave's points holds the indirect share now; the same wave adds the direct lobes' share below)
    // ---- the direct lobes, kSgBwdSlots x 64 at a time
    for (int base = 0; base < M; base += kSgBwdSlots * 64) {
        float acc[kSgBwdSlots][7];
#pragma unroll
        for (int k = 0; k < kSgBwdSlots; ++k)
#pragma unroll
            for (int i = 0; i < 7; ++i) acc[k][i] = 0.0f;
        for (int q = wave; q < kSgBwdPts; q += 4) {
            const int pt = pt0 + q;
            if (pt >= n) break;
            const PointConst pc = point_consts(normal + pt * 3, view + pt * 3, f0);
            float co[6], gm[7], m[7];
#pragma unroll
            for (int i = 0; i < 6; ++i) co[i] = d_out[pt * 12 + i];
#pragma unroll
            for (int i = 0; i < 7; ++i) gm[i] = 0.0f;
            load_material<HEADS>(mat, cs, pt, m);
#pragma unroll
            for (int k = 0; k < kSgBwdSlots; ++k) {
                const int j = base + k * 64 + lane;
                if (j < M) {
                    float sg[7], gs[7];
#pragma unroll
                    for (int i = 0; i < 7; ++i) sg[i] = lgt[j * 7 + i];
                    lobe_adjoint<5>(pc, m, sg, vis[(size_t)j * n + pt], co, gm, gs);
#pragma unroll
                    for (int i = 0; i < 7; ++i) acc[k][i] += gs[i];
                }
            }
#pragma unroll
            for (int i = 0; i < 7; ++i)
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) gm[i] += __shfl_xor(gm[i], s, 64);
            if (lane == 0) store_material_grad<HEADS, true>(gm, pt, d_mat, d_cs);
        }
        __syncthreads();     // (red: the previous sweep's readers are done)
#pragma unroll
        for (int k = 0; k < kSgBwdSlots; ++k)
#pragma unroll
            for (int i = 0; i < 7; ++i) red[wave][(k * 64 + lane) * 7 + i] = acc[k][i];
        __syncthreads();
        for (int e = threadIdx.x; e < kSgBwdSlots * 64 * 7; e += 256) {
            const int j = base + e / 7;
            if (j < M) atomicAdd(d_lgt + (size_t)base * 7 + e, red[0][e] + red[1][e] + red[2][e] + red[3][e]);
        }
    }
}

}  // namespace fneus

using namespace fneus;

static int sg_fwd(const float* lgt_sgs, const float* indir_sgs, const float* vis, const float* normal, const float* view,
                  const float* material, const float* cs, bool heads, int n_pts, int n_direct, int n_indirect, float f0, float* out,
                  hipStream_t stream) {
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    if (!lgt_sgs || !vis || !normal || !view || !material || !out || (n_indirect > 0 && !indir_sgs) || (heads && !cs)) {
        set_last_error("fneus_sg_render_fwd: null argument");
        return -2;
    }
    if (heads) hipLaunchKernelGGL(sg_render_fwd_kernel<true>, dim3(n_pts), dim3(64), 0, stream, lgt_sgs, indir_sgs, vis, normal, view, material, cs,
                                  n_pts, n_direct, n_indirect, f0, out);
    else hipLaunchKernelGGL(sg_render_fwd_kernel<false>, dim3(n_pts), dim3(64), 0, stream, lgt_sgs, indir_sgs, vis, normal, view, material, cs,
                            n_pts, n_direct, n_indirect, f0, out);
    return fneus::launch_status();
}

static int sg_bwd(const float* lgt_sgs, const float* indir_sgs, const float* vis, const float* normal, const float* view,
                  const float* material, const float* cs, bool heads, int n_pts, int n_direct, int n_indirect, float f0,
                  const float* d_out, float* d_material, float* d_cs, float* d_lgt_sgs, hipStream_t stream) {
    fneus::clear_status();
    if (n_pts <= 0) return 0;
    if (!lgt_sgs || !vis || !normal || !view || !material || !d_out || !d_material || !d_lgt_sgs || (n_indirect > 0 && !indir_sgs) ||
        (heads && (!cs || !d_cs))) {
        set_last_error("fneus_sg_render_bwd: null argument");
        return -2;
    }
    const dim3 grid((n_pts + kSgBwdPts - 1) / kSgBwdPts);
    if (heads) hipLaunchKernelGGL(sg_render_bwd_kernel<true>, grid, dim3(256), 0, stream, lgt_sgs, indir_sgs, vis, normal, view, material, cs, n_pts,
                                  n_direct, n_indirect, f0, d_out, d_material, d_cs, d_lgt_sgs);
    else hipLaunchKernelGGL(sg_render_bwd_kernel<false>, grid, dim3(256), 0, stream, lgt_sgs, indir_sgs, vis, normal, view, material, cs, n_pts,
                            n_direct, n_indirect, f0, d_out, d_material, d_cs, d_lgt_sgs);
    return fneus::launch_status();
}

extern "C" int fneus_sg_render_fwd(const float* lgt_sgs, const float* indir_sgs, const float* vis, const float* normal,
                                   const float* view, const float* material, int n_pts, int n_direct, int n_indirect,
                                   float specular_reflectance, float* out, fneus_stream_t stream) {
    return sg_fwd(lgt_sgs, indir_sgs, vis, normal, view, material, nullptr, false, n_pts, n_direct, n_indirect, specular_reflectance, out,
                  (hipStream_t)stream);
}
extern "C" int fneus_sg_render_bwd(const float* lgt_sgs, const float* indir_sgs, const float* vis, const float* normal,
                                   const float* view, const float* material, int n_pts, int n_direct, int n_indirect,
                                   float specular_reflectance, const float* d_out, float* d_material, float* d_lgt_sgs,
                                   fneus_stream_t stream) {
    return sg_bwd(lgt_sgs, indir_sgs, vis, normal, view, material, nullptr, false, n_pts, n_direct, n_indirect, specular_reflectance, d_out,
                  d_material, nullptr, d_lgt_sgs, (hipStream_t)stream);
}
extern "C" int fneus_sg_render_heads_fwd(const float* lgt_sgs, const float* indir_sgs, const float* vis, const float* normal,
                                         const float* view, const float* brdf, const float* cs, int n_pts, int n_direct, int n_indirect,
                                         float specular_reflectance, float* out, fneus_stream_t stream) {
    return sg_fwd(lgt_sgs, indir_sgs, vis, normal, view, brdf, cs, true, n_pts, n_direct, n_indirect, specular_reflectance, out,
                  (hipStream_t)stream);
}
extern "C" int fneus_sg_render_heads_bwd(const float* lgt_sgs, const float* indir_sgs, const float* vis, const float* normal,
                                         const float* view, const float* brdf, const float* cs, int n_pts, int n_direct, int n_indirect,
                                         float specular_reflectance, const float* d_out, float* d_brdf, float* d_cs, float* d_lgt_sgs,
                                         fneus_stream_t stream) {
    return sg_bwd(lgt_sgs, indir_sgs, vis, normal, view, brdf, cs, true, n_pts, n_direct, n_indirect, specular_reflectance, d_out, d_brdf,
                  d_cs, d_lgt_sgs, (hipStream_t)stream);
}
