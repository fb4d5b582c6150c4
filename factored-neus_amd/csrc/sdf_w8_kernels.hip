// SDF-network kernels on 8-wave workgroups (w8_engine.h): the same maths, operands and summation order per sample as
// sdf_kernels.hip (reference models/fields.py:74-111), another distribution of the work over the waves of a CU.
//   K1 sdf_fwd_w8p_kernel<PREC>: one tile per 8-wave workgroup with primed layers, for the latency-bound launches of the hierarchical
//   sampler (renderer.py:430: 16 new depths per ray = 256 tiles).  (The lockstep and staggered chip-filling forms of round 3 were
//   removed in round 6: no path selected them; DESIGN.md 4.1b keeps their measurements.)
#include <stdlib.h>
#include "w8_engine.h"
#include "p2_engine.h"
#include "fneus_kernels.h"
#include "sdf_w8.h"
#include "ray_sampler.h"

namespace fneus {

#ifdef FNEUS_W8_STAMPS
// timing experiments only (tools/experiments/r03/k1_stamps.py): s_memtime stamps of the first group of the first 32 workgroups,
// written BEHIND the n_pts outputs (the caller allocates the room): [block][wave][layer][5]
#define W8_STAMP(i)                                                                                       \
    do {                                                                                                  \
        if (stamp_on && lane == 0) stamps[((blockIdx.x * 8 + wave) * 8 + l) * 5 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define W8_STAMP(i) do { } while (0)
#endif

template <int HB>
FN_DEV void w8_softplus(f32x16 (&acc)[1][HB]) {
#pragma unroll
    for (int hb = 0; hb < HB; ++hb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][hb][r] = softplus100(acc[0][hb][r]);
}

// ---- K1 for the latency-bound launches of the hierarchical sampler (renderer.py:430: 16 new depths per ray = 8192 points = one
// 32-sample tile per CU): what such a launch costs is the serial latency of ONE tile through nine layers, and 60 % of a layer
// was waiting for its weight fragments (phase stamps, tools/experiments/r03/k1_stamps.py: dense 5570 cycles for 48 MFMAs).
// Here the whole layer's fragments of the wave's tile are in registers before the layer starts: stage s of the NEXT layer is
// requested into the register slot that stage s of the running layer has just vacated (16 slots x (hi, lo) = 128 registers).
typedef __attribute__((ext_vector_type(4))) unsigned int w8_u32x4;
FN_DEV bf16x8 w8_bload(__amdgpu_buffer_rsrc_t r, unsigned voff, uint32_t soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// LMAP as in p2_engine.h: 1 = the encoding alone (slots 16..18), 2 = 14 slots then the encoding; NXKS = k-steps of the next layer
template <int PREC, int KS, int LMAP, int NXKS>
FN_DEV void w8p_dense(bf16x8 (&wh)[17], bf16x8 (&wl)[17], const unsigned char* frag, f32x16& acc, int lane, __amdgpu_buffer_rsrc_t rsrc,
                      unsigned voff, uint32_t nx_hi, uint32_t nx_lo, int nx_nt) {
    constexpr int nx_ks = NXKS;
    constexpr int NPL = PREC == 3 ? 2 : 1;
    auto slot = [](int s) { return LMAP == 1 ? 16 + s : (LMAP == 2 ? (s < 14 ? s : s + 2) : s); };
    const unsigned char* fl = frag + lane * 16;
    bf16x8 bh[3], bl[3];
    bh[0] = *reinterpret_cast<const bf16x8*>(fl + (slot(0) * NPL) * kFragBytes);
    if constexpr (PREC == 3) bl[0] = *reinterpret_cast<const bf16x8*>(fl + (slot(0) * NPL + 1) * kFragBytes);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
            bh[(s + 1) % 3] = *reinterpret_cast<const bf16x8*>(fl + (slot(s + 1) * NPL) * kFragBytes);
            if constexpr (PREC == 3) bl[(s + 1) % 3] = *reinterpret_cast<const bf16x8*>(fl + (slot(s + 1) * NPL + 1) * kFragBytes);
        }
        if (s >= 1) {      // (hazard note of dense_ldsb(): keep the three B buffers three register sets)
            asm volatile("" ::"v"(bh[(s - 1) % 3]));
            if constexpr (PREC == 3) asm volatile("" ::"v"(bl[(s - 1) % 3]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PREC == 3) {
            acc = mfma32(wl[s], bh[s % 3], acc);
            acc = mfma32(wh[s], bl[s % 3], acc);
        }
        acc = mfma32(wh[s], bh[s % 3], acc);
        __builtin_amdgcn_sched_barrier(0);
        if (s < nx_ks) {       // the slot is free: stage s of the next layer
            const uint32_t f = (uint32_t)(s * nx_nt * 64) * 16u;
            wh[s] = w8_bload(rsrc, voff, nx_hi + f);
            if constexpr (PREC == 3) wl[s] = w8_bload(rsrc, voff, nx_lo + f);
        }
    }
#pragma unroll
    for (int s = KS; s < 17; ++s)
        if (s < nx_ks) {
            const uint32_t f = (uint32_t)(s * nx_nt * 64) * 16u;
            wh[s] = w8_bload(rsrc, voff, nx_hi + f);
            if constexpr (PREC == 3) wl[s] = w8_bload(rsrc, voff, nx_lo + f);
        }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
}

// MERGE (round 6): the launch carries a sampler step in its epilogue.  The samples are the k new depths of each ray (ray form, k = 16 or
// 32: a tile = 32 / k whole rays); behind the tile's sdf values, wave w < 32 / k runs cat_z_vals + up_sample (+ the last step's merge and
// sections) of ray tile * (32 / k) + w -- the body of merge_upsample_kernel (ray_sampler.h), bit for bit -- with the new sdf values taken
// from LDS.  renderer.py:433-446 is a per-ray recurrence: nothing of it needs another workgroup, and a launch boundary + the merge
// kernel's own ramp per up-sampling step go (3 launches of the training step's 31).
struct W8Merge {
    const float *z_old, *s_old;     // [B][m]
    int m;
    const float* z_new;             // [B][k] (= src.t)
    int k, n_rays;
    float inv_s;
    int k_next;
    float *z_out, *s_out, *z_next, *z_final;
    float sample_dist;
    float *dists, *mid_z;
};

// Several sampler steps in ONE launch (n_steps > 1, MERGE): a workgroup's tile is the same 32 / k rays in every step -- it evaluates the
// depths its own merge of the step before has written (z_next of step j = z_new of step j + 1, z_out / s_out = z_old / s_old), so the
// recurrence of renderer.py:433-446 needs no other workgroup and no launch boundary; the stores of a step are made visible to the
// workgroup (fence + the barrier at the top of the next step) before wave 0 encodes the new points.
constexpr int kW8MaxSteps = 4;
struct W8Steps {
    W8Merge st[kW8MaxSteps];
    int n;
};

template <int PREC, bool MERGE>
__global__ void __launch_bounds__(512, 2) sdf_fwd_w8p_kernel(const unsigned char* blob, PointSrc src0, long N, float* __restrict__ sdf_out,
                                                             W8Steps steps) {
    __shared__ __attribute__((aligned(16))) unsigned char frag[kW8Half];
    __shared__ float red[8 * 32];
    __shared__ float snew[MERGE ? 32 : 1];
    __shared__ float rayb[MERGE ? 2 : 1][MERGE ? 6 * MAXN + 8 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    constexpr auto& LY = kSdfLayout;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(blob), 0, 0x7fffffff, 0x00020000);
    const long tiles = (N + 31) / 32;
    auto tile_of = [&](int l) { return (l == 3 && wave == 7) ? 6 : wave; };      // layer 3 has 7 tiles: wave 7 repeats tile 6 (not published)
    auto bias_of = [&](int l) {
        const f32x16 FN_GLOBAL* __restrict__ p = reinterpret_cast<const f32x16 FN_GLOBAL*>((gblob_t)blob + LY.L[l].bias);
        return p[tile_of(l) * 2 + h];
    };
#ifdef FNEUS_W8_STAMPS
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(sdf_out + ((N + 3) & ~3L));
#endif
    const int n_steps = MERGE ? steps.n : 1;
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x)
    for (int step = 0; step < n_steps; ++step) {
        const W8Merge& mg = steps.st[MERGE ? step : 0];
        PointSrc src = src0;
        if (MERGE) src.t = mg.z_new;
#ifdef FNEUS_W8_STAMPS
        const bool stamp_on = tile == blockIdx.x && blockIdx.x < 32 && step == 0;
#endif
        asm volatile("" : "+s"(blob));
        w8_barrier();                                   // the previous tile's fragments and sums are consumed
        bf16x8 wh[17], wl[17];                          // (declared per tile: while the point is encoded only layer 0's three stages are live)
        f32x16 bias;
        {
            const unsigned voff0 = (unsigned)(lane + wave * 64) * 16u;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                wh[s] = w8_bload(rsrc, voff0, LY.L[0].fwd_hi + (uint32_t)(s * 8 * 64) * 16u);
                if constexpr (PREC == 3) wl[s] = w8_bload(rsrc, voff0, LY.L[0].fwd_lo + (uint32_t)(s * 8 * 64) * 16u);
            }
            bias = bias_of(0);
        }
        if (wave == 0) {                                // the encoding: slots 16..18 (read in place by layers 0 and 4)
            const long n = tile * 32 + r;
            const long nc = n < N ? n : N - 1;
            float x[3], pe[39], jc[39];
            load_point(src, nc, x);
            posenc<6, false>(x, pe, jc);
            BFrag<PREC> pf[kMaxKS];
            vec_to_bfrag<PREC, 39, 3, 0>(pe, pf, h);
            frags_to_lds<PREC, 3>(frag, lane, 16, pf);
        }
        w8_barrier();
        f32x16 acc1[1][1];
        f32x16& acc = acc1[0][0];
        // the eight layers written out (compile-time layer index: the register slots of the fragments keep their identity)
        static_for<0, 8>([&](auto L_) {
            constexpr int l = decltype(L_)::value;
            constexpr int ln = l == 7 ? 0 : l + 1;                  // the layer whose fragments are requested meanwhile
            constexpr int KSl = l == 0 ? 3 : (l == 4 ? 17 : 16), LMAPl = l == 0 ? 1 : (l == 4 ? 2 : 0);
            constexpr int NXKS = l == 7 ? 0 : (ln == 4 ? 17 : 16);
            const unsigned voff = (unsigned)(lane + tile_of(ln) * 64) * 16u;
            W8_STAMP(0);
            acc = bias;
            w8p_dense<PREC, KSl, LMAPl, NXKS>(wh, wl, frag, acc, lane, rsrc, voff, kSdfLayout.L[ln].fwd_hi, kSdfLayout.L[ln].fwd_lo, ln == 3 ? 7 : 8);
            if constexpr (l < 7) bias = bias_of(ln);
            W8_STAMP(1);
            w8_softplus<1>(acc1);
#ifdef FNEUS_W8_STAMPS
            asm volatile("" :: "v"(acc1[0][0]));
#endif
            W8_STAMP(2);
            if constexpr (l < 7) {
                w8_barrier();                                       // everyone has read the previous layer's fragments
                W8_STAMP(3);
                if (!(l == 3 && wave == 7)) {
                    unsigned char* none[1] = {nullptr};
                    const bool valid[1] = {true};
                    const PPLane pl = pp_lane(lane);
                    w8_put_frags<PREC, 1, false>(frag, lane, wave, acc1, none, none, pl, valid);
                }
                w8_barrier();                                       // all fragments of the layer are in LDS
                W8_STAMP(4);
            }
        });
        {   // sdf = b_8[0] + W_8[0, :] . h_8
            f32x16 cw[1];
            load_accvec<8, 0, 1>(blob, LY.extra, cw, lane, wave);
            float p = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) p = fmaf(acc[i], cw[0][i], p);
            p += xor32(p);
            if (lane < 32) red[wave * 32 + lane] = p;
            w8_barrier();
            if (wave == 0 && lane < 32) {
                f32x16 b8[1];
                load_accvec<9, 8, 1>(blob, LY.L[8].bias, b8, lane);
                float s = b8[0][0];
#pragma unroll
                for (int w = 0; w < 8; ++w) s += red[w * 32 + lane];
                const long n = tile * 32 + r;
                if (n < N && sdf_out != nullptr) sdf_out[n] = s;
                if constexpr (MERGE) snew[lane] = s;
            }
        }
        if constexpr (MERGE) {
            w8_barrier();                               // the tile's values are in LDS
            const int R = 32 / mg.k;
            if (wave < R) {
                const long ray = tile * R + wave;
                if (ray < mg.n_rays) {
                    float* b = rayb[wave];
                    merge_upsample_ray<false>((int)ray, lane, src.rays_o, src.rays_d, mg.z_old, mg.s_old, mg.m, mg.z_new, snew + wave * mg.k, true,
                                              mg.k, mg.inv_s, mg.k_next, mg.z_out, mg.s_out, mg.z_next, mg.z_final, mg.sample_dist, mg.dists,
                                              mg.mid_z, b, b + MAXN, b + 2 * MAXN, b + 3 * MAXN, b + 4 * MAXN, b + 5 * MAXN + 4);
                }
            }
            if (step + 1 < n_steps) __threadfence_block();          // the next step of this tile reads what the merge has stored
        }
    }
}

int sdf_fwd_w8p(const unsigned char* b, const PointSrc& src, long n_pts, float* sdf_out, int prec, hipStream_t stream) {
    const long tiles = (n_pts + 31) / 32;
    const dim3 grid((unsigned)(tiles < 1024 ? tiles : 1024));
    const W8Steps none{};
    if (prec == 3) hipLaunchKernelGGL((sdf_fwd_w8p_kernel<3, false>), grid, dim3(512), 0, stream, b, src, n_pts, sdf_out, none);
    else if (prec == 1) hipLaunchKernelGGL((sdf_fwd_w8p_kernel<1, false>), grid, dim3(512), 0, stream, b, src, n_pts, sdf_out, none);
    else return -2;
    return launch_status();
}

}  // namespace fneus

using namespace fneus;

// K1 on the new depths of an up-sampling step + that step's cat_z_vals / the next step's up_sample in ONE launch (renderer.py:430-446;
// fneus_sdf_fwd followed by fneus_merge_upsample, bit for bit).  -3: a shape this form does not take (the caller runs the two launches).
extern "C" int fneus_sdf_fwd_merge_upsample(const void* blob, const float* rays_o, const float* rays_d, const float* z_old, const float* s_old,
                                            int m, const float* z_new, int k, int n_rays, float inv_s, int k_next, float* z_out, float* s_out,
                                            float* z_next, float* z_final, float sample_dist, float* dists, float* mid_z, float* s_new_out,
                                            int prec, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0) return 0;
    if (!blob || !rays_o || !rays_d || !z_old || !s_old || !z_new || !z_out || !s_out || !z_next) return -2;
    const long n_pts = (long)n_rays * k;
    const long tiles = (n_pts + 31) / 32;
    if ((k != 16 && k != 32) || tiles >= 1024 || m + k > MAXN || m + k + k_next > MAXN || k_next < 1 || (prec != 3 && prec != 1)) return -3;
    PointSrc src{nullptr, rays_o, rays_d, z_new, k};
    W8Steps one{};
    one.st[0] = W8Merge{z_old, s_old, m, z_new, k, n_rays, inv_s, k_next, z_out, s_out, z_next, z_final, sample_dist, (z_final && mid_z) ? dists : nullptr, mid_z};
    one.n = 1;
    const dim3 grid((unsigned)tiles);
    const unsigned char* b = reinterpret_cast<const unsigned char*>(blob);
    if (prec == 3) hipLaunchKernelGGL((sdf_fwd_w8p_kernel<3, true>), grid, dim3(512), 0, stream, b, src, n_pts, s_new_out, one);
    else hipLaunchKernelGGL((sdf_fwd_w8p_kernel<1, true>), grid, dim3(512), 0, stream, b, src, n_pts, s_new_out, one);
    return fneus::launch_status();
}

// The same for ALL remaining steps of a render in one launch (renderer.py:433-446 for i = 1 .. up_sample_steps - 1): step j evaluates
// z_new[j], merges it into (z_old[j], s_old[j]) -> (z_out[j], s_out[j]) and draws z_next[j]; the caller chains the buffers (z_old[j + 1] =
// z_out[j], s_old[j + 1] = s_out[j], z_new[j + 1] = z_next[j]); the last step writes z_final (+ sections).  -3: shapes it does not take.
extern "C" int fneus_sdf_fwd_merge_upsample_steps(const void* blob, const float* rays_o, const float* rays_d, int n_steps,
                                                  const FneusSamplerStep* st, int k, int n_rays, int k_next, float* z_final,
                                                  float sample_dist, float* dists, float* mid_z, int prec, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_rays <= 0 || n_steps <= 0) return 0;
    if (!blob || !rays_o || !rays_d || !st || !z_final) return -2;
    const long n_pts = (long)n_rays * k;
    const long tiles = (n_pts + 31) / 32;
    if (n_steps > fneus::kW8MaxSteps || (k != 16 && k != 32) || tiles >= 1024 || k_next < 1 || (prec != 3 && prec != 1)) return -3;
    // (one tile per workgroup: a tile's steps must stay with their workgroup, and with tiles < 1024 the grid covers them all)
    W8Steps all{};
    all.n = n_steps;
    for (int j = 0; j < n_steps; ++j) {
        const bool last = j + 1 == n_steps;
        if (!st[j].z_old || !st[j].s_old || !st[j].z_new || !st[j].z_out || !st[j].s_out || !st[j].z_next) return -2;
        if (st[j].m + k > MAXN || st[j].m + k + k_next > MAXN) return -3;
        all.st[j] = W8Merge{st[j].z_old, st[j].s_old, st[j].m, st[j].z_new, k, n_rays, st[j].inv_s, k_next, st[j].z_out, st[j].s_out, st[j].z_next,
                            last ? z_final : nullptr, sample_dist, (last && mid_z) ? dists : nullptr, last ? mid_z : nullptr};
    }
    PointSrc src{nullptr, rays_o, rays_d, st[0].z_new, k};
    const dim3 grid((unsigned)tiles);
    const unsigned char* b = reinterpret_cast<const unsigned char*>(blob);
    if (prec == 3) hipLaunchKernelGGL((sdf_fwd_w8p_kernel<3, true>), grid, dim3(512), 0, stream, b, src, n_pts, (float*)nullptr, all);
    else hipLaunchKernelGGL((sdf_fwd_w8p_kernel<1, true>), grid, dim3(512), 0, stream, b, src, n_pts, (float*)nullptr, all);
    return fneus::launch_status();
}

namespace fneus {

}  // namespace fneus
