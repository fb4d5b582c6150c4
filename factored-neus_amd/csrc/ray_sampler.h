// Per-ray pieces of the hierarchical sampler (reference models/renderer.py:152-205, 43-77) shared by the per-ray kernels of
// ray_kernels.hip and by the K1 launch that carries a sampler step in its epilogue (sdf_w8_kernels.hip, round 6): one wave per ray,
// the ray's depths and sdf values in LDS arrays of that wave.
#pragma once
#include "fneus_common.h"

namespace fneus {

constexpr int MAXN = 256;
constexpr int PER = 4;

FN_DEV float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

// inclusive wave scans over 64 lanes
FN_DEV float wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(v, d, 64);
        if (lane >= d) v *= o;
    }
    return v;
}
FN_DEV float wave_incl_sum(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
FN_DEV float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
FN_DEV int wave_min_i(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
    return v;
}
FN_DEV float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

FN_DEV float pt_norm(const float* o, const float* d, float z) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float p = __fadd_rn(o[c], __fmul_rn(d[c], z));
        s = __fadd_rn(s, __fmul_rn(p, p));
    }
    return sqrtf(s);
}

// exclusive cumulative product over the ray: vals[j] for this lane's chunk [lane*PER, lane*PER+PER) -> T[j]
template <int PER>
FN_DEV void excl_cumprod(const float (&f)[PER], float (&T)[PER], int lane) {
    float loc = 1.0f;
#pragma unroll
    for (int j = 0; j < PER; ++j) loc *= f[j];
    const float inc = wave_incl_prod(loc, lane);
    float run = __shfl_up(inc, 1, 64);
    if (lane == 0) run = 1.0f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        T[j] = run;
        run *= f[j];
    }
}

// The body of merge_upsample_kernel for ONE ray, run by one wave (lane 0..63): zin / sin_ / zs / ss / cdf (MAXN + 1) / znx are the wave's
// LDS arrays.  SOLO: the wave is the whole workgroup (`__syncthreads`); otherwise (the wave is one of several of a workgroup, each with
// its own ray and arrays: sdf_w8_kernels.hip) a wave-level fence orders its LDS traffic -- the arithmetic is the same, bit for bit.
template <bool SOLO>
FN_DEV void ray_sync() {
    if constexpr (SOLO) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}


// PER samples per lane: 4 for the primary rays (m <= 256), 8 for the 512 coarse samples of a stage-2 secondary ray
// (calLvis.py:55-90 is the same algorithm as renderer.py:152-189 and returns the new depths only)
// the body of up_sample + sample_pdf for ONE ray held in LDS (zs, ss: m depths and sdf values, visible to the wave; cdf:
// scratch of m + 1 floats): the k new depths go to z_new (global) and, when given, to z_new_lds
template <int PER, bool SOLO = true>
FN_DEV void upsample_ray(const float (&o)[3], const float (&d)[3], const float* zs, const float* ss, float* cdf, int m, int k,
                         float inv_s, int lane, float* __restrict__ z_new, float* z_new_lds) {
    const int ns = m - 1;   // sections
    float alpha[PER], fac[PER], T[PER], w[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = lane * PER + j;
        float a = 0.0f;
        if (i < ns) {
            const float z0 = zs[i], z1 = zs[i + 1], s0 = ss[i], s1 = ss[i + 1];
            const bool inside = (pt_norm(o, d, z0) < 1.0f) || (pt_norm(o, d, z1) < 1.0f);
            const float cosv = (s1 - s0) / (z1 - z0 + 1e-5f);
            float prev = 0.0f;
            if (i > 0) prev = (s0 - ss[i - 1]) / (z0 - zs[i - 1] + 1e-5f);
            float cv = fminf(prev, cosv);
            cv = fminf(fmaxf(cv, -1e3f), 0.0f) * (inside ? 1.0f : 0.0f);
            const float dist = z1 - z0;
            const float mid = (s0 + s1) * 0.5f;
            const float pe = mid - cv * dist * 0.5f, ne = mid + cv * dist * 0.5f;
            const float pc = sigmoid_acc(pe * inv_s), nc = sigmoid_acc(ne * inv_s);
            a = (pc - nc + 1e-5f) / (pc + 1e-5f);
        }
        alpha[j] = a;
        fac[j] = (i < ns) ? (1.0f - a + 1e-7f) : 1.0f;
    }
    excl_cumprod(fac, T, lane);
    float loc = 0.0f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = lane * PER + j;
        w[j] = (i < ns) ? alpha[j] * T[j] + 1e-5f : 0.0f;   // sample_pdf: weights + 1e-5
        loc += w[j];
    }
    const float total = wave_sum(loc);
    // cdf = [0, cumsum(pdf)]
    float locp = 0.0f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        w[j] = w[j] / total;
        locp += w[j];
    }
    const float inc = wave_incl_sum(locp, lane);
    float run = inc - locp;
    if (lane == 0) cdf[0] = 0.0f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = lane * PER + j;
        run += w[j];
        if (i < ns) cdf[i + 1] = run;
    }
    ray_sync<SOLO>();
    for (int q = lane; q < k; q += 64) {
        // torch.linspace(0.5/k, 1-0.5/k, k)
        const float start = 0.5f / k, end = 1.0f - 0.5f / k;
        const float step = (k > 1) ? (end - start) / (float)(k - 1) : 0.0f;
        const float u = (q < k / 2) ? start + step * q : end - step * (k - 1 - q);
        // searchsorted(cdf, u, right=True) = number of entries <= u
        int lo = 0, hi = m;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = max(lo - 1, 0), above = min(lo, m - 1);
        const float cb = cdf[below], ca = cdf[above];
        float den = ca - cb;
        if (den < 1e-5f) den = 1.0f;
        const float t = (u - cb) / den;
        const float zq = zs[below] + t * (zs[above] - zs[below]);
        z_new[q] = zq;
        if (z_new_lds) z_new_lds[q] = zq;
    }
}


// Rank of element i of (a[0..m) | b[0..k)) in the STABLE sort of the concatenation (what merge_kernel finds with one scan over all
// m + k entries per element) when `a` is ascending -- the old depths of a sampler step always are: ray_setup's depths or the
// output of the previous merge.  `b` (the new depths of sample_pdf: ascending up to rounding) is scanned as it is, so no order is
// assumed within it.  An old entry has its i predecessors of `a` in front of it plus the entries of b strictly below it; a new
// entry the entries of b in front of it (stable) plus every entry of `a` that is <= it (upper bound by bisection).
FN_DEV int merged_rank(const float* a, int m, const float* b, int k, int i) {
    const bool is_new = i >= m;
    const int ib = i - m;
    const float z = is_new ? b[ib] : a[i];
    int r = 0;
    for (int j = 0; j < k; ++j) {
        const float zj = b[j];
        r += (zj < z) || (is_new && zj == z && j < ib);
    }
    if (!is_new) return r + i;
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= z) lo = mid + 1; else hi = mid;
    }
    return r + lo;
}

template <bool SOLO>
FN_DEV void merge_upsample_ray(int ray, int lane, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                               const float* __restrict__ z_old, const float* __restrict__ s_old, int m,
                               const float* __restrict__ z_new, const float* s_new /*global [B][k], or the ray's k values in LDS: s_new_ray*/,
                               bool s_new_is_ray, int k, float inv_s, int k_next, float* __restrict__ z_out, float* __restrict__ s_out,
                               float* __restrict__ z_next, float* __restrict__ z_final, float sample_dist, float* __restrict__ dists,
                               float* __restrict__ mid_z, float* zin, float* sin_, float* zs, float* ss, float* cdf, float* znx) {
    const int n = m + k;
    for (int i = lane; i < n; i += 64) {
        zin[i] = (i < m) ? z_old[(size_t)ray * m + i] : z_new[(size_t)ray * k + (i - m)];
        sin_[i] = (i < m) ? s_old[(size_t)ray * m + i] : (s_new_is_ray ? s_new[i - m] : s_new[(size_t)ray * k + (i - m)]);
    }
    ray_sync<SOLO>();
    for (int i = lane; i < n; i += 64) {
        const float z = zin[i];
        const int rank = merged_rank(zin, m, zin + m, k, i);
        zs[rank] = z;
        ss[rank] = sin_[i];
        z_out[(size_t)ray * n + rank] = z;
        s_out[(size_t)ray * n + rank] = sin_[i];
    }
    ray_sync<SOLO>();
    float o[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = rays_o[ray * 3 + c];
        d[c] = rays_d[ray * 3 + c];
    }
    upsample_ray<4, SOLO>(o, d, zs, ss, cdf, n, k_next, inv_s, lane, z_next + (size_t)ray * k_next, znx);
    if (z_final == nullptr) return;
    ray_sync<SOLO>();
    const int nf = n + k_next;
    for (int i = lane; i < nf; i += 64) {
        const float z = (i < n) ? zs[i] : znx[i - n];
        const int rank = merged_rank(zs, n, znx, k_next, i);
        z_final[(size_t)ray * nf + rank] = z;
        if (dists) zin[rank] = z;                    // (zin is free: the first merge has read it)
    }
    if (dists == nullptr) return;
    // the sections of the final depths (sections_kernel's expressions: renderer.py:223-226) -- what render_core asks for next
    ray_sync<SOLO>();
    for (int i = lane; i < nf; i += 64) {
        const float z0 = zin[i];
        const float dd = (i + 1 < nf) ? zin[i + 1] - z0 : sample_dist;
        dists[(size_t)ray * nf + i] = dd;
        mid_z[(size_t)ray * nf + i] = z0 + dd * 0.5f;
    }
}

}  // namespace fneus
