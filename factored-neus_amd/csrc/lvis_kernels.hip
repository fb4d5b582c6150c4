// Per-lobe light visibility of stage 3 (reference models/inverRender.py:128-192 get_diffuse_visibility) on the distilled
// Lvis network (models/fields.py:338-369: [PE10(point) | PE4(direction)] 90 -> 4 x (256 + ReLU) -> 1 -> sigmoid).
//
// For every surface point the reference evaluates Lvis at S = 32 directions around each of the M = 128 light lobes --
// 4096 evaluations per point, up to 2.1 M per training step, the hot op of stage 3 -- zeroes the directions that face away
// from the normal and averages each lobe's samples with the weights exp(lambda (d . axis - 1)).  Here one 32-sample tile
// IS one (point, lobe) pair: sample r of the tile is direction r of the lobe.  That makes
//   * the encoding of the point (60 sincos) a per-workgroup constant: computed once per (point, chunk of 32 lobes), its four
//     k-steps by the four waves, and parked in LDS;
//   * the back-face test a per-tile decision: a lobe whose 32 directions all face away costs nothing (about half of them);
//   * the weighted average a reduction over the 32 lanes of the output tile: the kernel writes vis[lobe][point] directly --
//     the 4096 network outputs per point never reach memory.
// Tensor-parallel workgroup of 4 waves per tile (tp_engine.h), two workgroups per CU, weights streamed from L2 (0.6 MB).
#define FNEUS_PREFETCH_X3 4
#define FNEUS_PREFETCH_X1 8
#include <stdlib.h>
#include "pp_engine.h"
#include "fneus_kernels.h"
#include "lvis_p2.h"

#ifndef FNEUS_LVIS_P2_DEFAULT
#define FNEUS_LVIS_P2_DEFAULT 1
#endif
#ifndef FNEUS_LVIS_OCC
#define FNEUS_LVIS_OCC 2      // workgroups per CU the tensor-parallel kernels of this file are compiled for (experiments: 3)
#endif

namespace fneus {

constexpr int kLvisLds = 22 * 2 * kFragBytes;      // 16 k-steps of a 256-wide layer (hi, lo) + the 6 input k-steps, parked
constexpr int kLvisPark = 16;                      // k-steps 16..19: PE10 of the point (once per workgroup), 20, 21: PE4(direction)
constexpr int kLvisChunk = 32;                     // lobes per workgroup: the launch is (points x lobe chunks) workgroups, so that
                                                   // skipped lobes / points shorten the launch instead of idling a resident slot

template <int PREC, int KS>
FN_DEV void lvis_write_frags(unsigned char* frag, int lane, int ks0, const BFrag<PREC>* b) {
    constexpr int NPL = PREC == 3 ? 2 : 1;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        *reinterpret_cast<bf16x8*>(frag + ((ks0 + i) * NPL) * kFragBytes + lane * 16) = b[i].hi;
        if constexpr (PREC == 3) *reinterpret_cast<bf16x8*>(frag + ((ks0 + i) * NPL + 1) * kFragBytes + lane * 16) = b[i].lo;
    }
}

// the 16 features of k-step ks of PE10(point) (63 features, embedder.py:23-36 with input_dims = 3) as one B fragment; each
// lane computes only the 8 features of its own slots (feature phi(ks, h, j)), so the 60 sincos of a point are spread over
// the four waves of the workgroup
template <int PREC>
FN_DEV void posenc3_frag(const float (&x)[3], int ks, int h, BFrag<PREC>& out) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int f = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
        float val = 0.0f;
        if (f < 63) {
            const int g = f - 3;
            const int c = f < 3 ? f : (g % 3);
            const float xc = c == 0 ? x[0] : (c == 1 ? x[1] : x[2]);
            if (f < 3) {
                val = xc;
            } else {
                float sn, cs;
                fn_sincos(xc * (float)(1 << (g / 6)), sn, cs);
                val = ((g % 6) >= 3) ? cs : sn;
            }
        }
        if constexpr (PREC == 3) {
            __bf16 a, b;
            split_bf16(val, a, b);
            out.hi[j] = a;
            out.lo[j] = b;
        } else {
            out.hi[j] = (__bf16)val;
        }
    }
}

template <int PREC>
__global__ void __launch_bounds__(256, FNEUS_LVIS_OCC) lvis_visibility_tp_kernel(const unsigned char* blob, const float* __restrict__ points,
                                                                    const float* __restrict__ normals,
                                                                    const float* __restrict__ dirs /*[M][32][3]*/,
                                                                    const float* __restrict__ weights /*[M][32]*/,
                                                                    const unsigned char* __restrict__ point_mask, int n_pts,
                                                                    int n_lobes, float* __restrict__ vis /*[M][n_pts]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    unsigned char* park = frag + (size_t)kLvisPark * (PREC == 3 ? 2 : 1) * kFragBytes;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform for the compiler too
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    constexpr auto& LY = kLvisLayout;
    const int n_chunks = (n_lobes + kLvisChunk - 1) / kLvisChunk;
    for (long item = blockIdx.x; item < (long)n_pts * n_chunks; item += gridDim.x) {
        const int pt = (int)(item / n_chunks), lobe0 = (int)(item - (long)pt * n_chunks) * kLvisChunk;
        const int lobe1 = lobe0 + kLvisChunk < n_lobes ? lobe0 + kLvisChunk : n_lobes;
        if (point_mask && point_mask[pt] == 0) {          // a ray without a surface hit (fixed-shape step): nothing to evaluate
            for (int lobe = lobe0 + (int)threadIdx.x; lobe < lobe1; lobe += blockDim.x) vis[(size_t)lobe * n_pts + pt] = 0.0f;
            continue;
        }
        float x[3], nrm[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[c] = points[pt * 3 + c];
            nrm[c] = normals[pt * 3 + c];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         // the previous item's parked fragments are consumed
        {
            BFrag<PREC> one[1];
            posenc3_frag<PREC>(x, wave, h, one[0]);       // k-step `wave` of the point's encoding
            lvis_write_frags<PREC, 1>(frag, lane, kLvisPark + wave, one);
        }
        for (int lobe = lobe0; lobe < lobe1; ++lobe) {
            asm volatile("" : "+s"(blob));
            float d[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = dirs[((size_t)lobe * 32 + r) * 3 + c];
            const bool front = (nrm[0] * d[0] + nrm[1] * d[1] + nrm[2] * d[2]) > 1e-6f;      // inverRender.py:169
            // (readfirstlane: the compiler must see a wave-uniform branch, the tile loop launders a scalar pointer)
            const int any_front = __builtin_amdgcn_readfirstlane((int)(__ballot(front) != 0ull));
            if (!any_front) {             // the whole lobe faces away from this point: visibility 0 (the same in all 4 waves)
                if (threadIdx.x == 0) vis[(size_t)lobe * n_pts + pt] = 0.0f;
                continue;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the previous tile's fragments are consumed
            if (wave == 1) {
                float pe[27], jc[27];
                posenc<4, false>(d, pe, jc);
                BFrag<PREC> tmp[kMaxKS];
                vec_to_bfrag<PREC, 27, 2, 0>(pe, tmp, h);
                lvis_write_frags<PREC, 2>(frag, lane, kLvisPark + 4, tmp);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the 6 input k-steps are in LDS
            BFrag<PREC> bf[kMaxKS];
            tp_operands<PREC, 6>(park, lane, bf);
            f32x16 acc[2];
            load_accvec<8, 0, 2>(blob, LY.L[0].bias, acc, lane, t0);
            tp_dense<PREC, 6, 8, 0, 2>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, park, bf, acc, lane, t0);
#pragma unroll 1
            for (int l = 1; l <= 3; ++l) {      // (no per-layer laundering of `blob` here: together with the skip path above it
                                                // makes the backend place the pointer in a VGPR and fail)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t][e] = fmaxf(acc[t][e], 0.0f);
                tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, nullptr, nullptr, pl, true);
                tp_operands<PREC, 16>(frag, lane, bf);
                load_accvec<8, 0, 2>(blob, LY.L[l].bias, acc, lane, t0);
                tp_dense<PREC, 16, 8, 0, 2>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, frag, bf, acc, lane, t0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = fmaxf(acc[t][e], 0.0f);
            tp_exchange_pp<PREC, 2, true>(frag, lane, t0, acc, nullptr, nullptr, pl, true);
            tp_operands<PREC, 16>(frag, lane, bf);
            if (wave == 0) {              // output layer: one tile, row 0 = register 0 of lane half 0
                f32x16 o[1];
                load_accvec<1, 0, 1>(blob, LY.L[4].bias, o, lane);
                tp_dense<PREC, 16, 1, 0, 1>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, frag, bf, o, lane);
                const float w = weights[(size_t)lobe * 32 + r];
                float num = (h == 0 && front) ? w / (1.0f + expf(-o[0][0])) : 0.0f;      // fields.py:358 sigmoid; :183 zero when back-facing
                float den = h == 0 ? w : 0.0f;
#pragma unroll
                for (int s = 16; s >= 1; s >>= 1) {
                    num += __shfl_xor(num, s, 64);
                    den += __shfl_xor(den, s, 64);
                }
                if (lane == 0) vis[(size_t)lobe * n_pts + pt] = num / (den + 1e-6f);    // inverRender.py:188
            }
        }
    }
}


// ---- two lobes per pass (HB = 2) -------------------------------------------------------------------------------------------
// The tensor-parallel kernels stream every weight fragment from L2 once per 32-sample tile: at the full MFMA rate that is
// ~85 B / clk / CU, 52 TB/s chip-wide against the ~18 TB/s the L2s deliver -- which is where the 34-37 % MFMA-busy plateau of
// all of them comes from.  Here a workgroup carries TWO (point, lobe) tiles through the network at once (dense_ldsb_h: one
// pass over the weight fragments feeds both halves), halving the weight stream per evaluation.  No stash, so the register
// budget allows it: accumulators [2 tiles][2 halves].
constexpr int kLvisHalf = 16 * 2 * kFragBytes;     // B fragments of one half (the 6 input k-steps use the first 6)
constexpr int kLvisLds2 = 2 * kLvisHalf;

template <int PREC>
__global__ void __launch_bounds__(256, 2) lvis_visibility_tph_kernel(const unsigned char* blob, const float* __restrict__ points,
                                                                     const float* __restrict__ normals,
                                                                     const float* __restrict__ dirs /*[M][32][3]*/,
                                                                     const float* __restrict__ weights /*[M][32]*/,
                                                                     const unsigned char* __restrict__ point_mask, int n_pts,
                                                                     int n_lobes, float* __restrict__ vis /*[M][n_pts]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];
    unsigned char* frag = lds_;
    constexpr int HB = 2, HALF = kLvisHalf;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const int t0 = 2 * wave;
    const PPLane pl = pp_lane(lane);
    constexpr auto& LY = kLvisLayout;
    const int n_chunks = (n_lobes + kLvisChunk - 1) / kLvisChunk;
    unsigned char* const none[HB] = {nullptr, nullptr};
    const bool both[HB] = {true, true};
    for (long item = blockIdx.x; item < (long)n_pts * n_chunks; item += gridDim.x) {
        const int pt = (int)(item / n_chunks), lobe0 = (int)(item - (long)pt * n_chunks) * kLvisChunk;
        const int lobe1 = lobe0 + kLvisChunk < n_lobes ? lobe0 + kLvisChunk : n_lobes;
        if (point_mask && point_mask[pt] == 0) {
            for (int lobe = lobe0 + (int)threadIdx.x; lobe < lobe1; lobe += blockDim.x) vis[(size_t)lobe * n_pts + pt] = 0.0f;
            continue;
        }
        float x[3], nrm[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[c] = points[pt * 3 + c];
            nrm[c] = normals[pt * 3 + c];
        }
        BFrag<PREC> pe_one[1];                            // k-step `wave` of the point's encoding, kept in registers
        posenc3_frag<PREC>(x, wave, h, pe_one[0]);
        for (int la = lobe0; la < lobe1; la += HB) {
            asm volatile("" : "+s"(blob));
            float d[HB][3];
            bool front[HB];
            int any_front[HB];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const int lobe = la + hb < lobe1 ? la + hb : la;
#pragma unroll
                for (int c = 0; c < 3; ++c) d[hb][c] = dirs[((size_t)lobe * 32 + r) * 3 + c];
                front[hb] = (la + hb < lobe1) && (nrm[0] * d[hb][0] + nrm[1] * d[hb][1] + nrm[2] * d[hb][2]) > 1e-6f;
                any_front[hb] = __builtin_amdgcn_readfirstlane((int)(__ballot(front[hb]) != 0ull));
            }
            if (!any_front[0] && !any_front[1]) {
                if (threadIdx.x < HB && la + (int)threadIdx.x < lobe1) vis[(size_t)(la + threadIdx.x) * n_pts + pt] = 0.0f;
                continue;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the previous pair's fragments are consumed
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) lvis_write_frags<PREC, 1>(frag + hb * HALF, lane, wave, pe_one);
            if (wave >= 2) {                              // waves 2, 3: PE4 of the directions of half 0, 1 (k-steps 4, 5)
                const int hb = wave - 2;
                float pe[27], jc[27], dd[3] = {hb ? d[1][0] : d[0][0], hb ? d[1][1] : d[0][1], hb ? d[1][2] : d[0][2]};
                posenc<4, false>(dd, pe, jc);
                BFrag<PREC> tmp[kMaxKS];
                vec_to_bfrag<PREC, 27, 2, 0>(pe, tmp, h);
                lvis_write_frags<PREC, 2>(frag + hb * HALF, lane, 4, tmp);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            f32x16 acc[2][HB];
            {
                f32x16 b2[2];
                load_accvec<8, 0, 2>(blob, LY.L[0].bias, b2, lane, t0);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) acc[t][hb] = b2[t];
            }
            tph_dense<PREC, 6, 8, 0, 2, true, HB, HALF>(blob, LY.L[0].fwd_hi, LY.L[0].fwd_lo, frag, acc, lane, t0);
#pragma unroll 1
            for (int l = 1; l <= 4; ++l) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[t][hb][e] = fmaxf(acc[t][hb][e], 0.0f);
                tph_exchange<PREC, 2, true, HB, HALF>(frag, lane, t0, acc, none, none, pl, both);
                if (l <= 3) {
                    f32x16 b2[2];
                    load_accvec<8, 0, 2>(blob, LY.L[l].bias, b2, lane, t0);
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int hb = 0; hb < HB; ++hb) acc[t][hb] = b2[t];
                    tph_dense<PREC, 16, 8, 0, 2, true, HB, HALF>(blob, LY.L[l].fwd_hi, LY.L[l].fwd_lo, frag, acc, lane, t0);
                }
            }
            if (wave == 0) {              // output layer: one tile per half, row 0 = register 0 of lane half 0
                f32x16 o[1][HB], b1[1];
                load_accvec<1, 0, 1>(blob, LY.L[4].bias, b1, lane);
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) o[0][hb] = b1[0];
                tph_dense<PREC, 16, 1, 0, 1, true, HB, HALF>(blob, LY.L[4].fwd_hi, LY.L[4].fwd_lo, frag, o, lane);
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) {
                    if (la + hb < lobe1) {
                        const float w = weights[(size_t)(la + hb) * 32 + r];
                        float num = (h == 0 && front[hb]) ? w / (1.0f + expf(-o[0][hb][0])) : 0.0f;
                        float den = h == 0 ? w : 0.0f;
#pragma unroll
                        for (int s = 16; s >= 1; s >>= 1) {
                            num += __shfl_xor(num, s, 64);
                            den += __shfl_xor(den, s, 64);
                        }
                        if (lane == 0) vis[(size_t)(la + hb) * n_pts + pt] = num / (den + 1e-6f);
                    }
                }
            }
        }
    }
}

}  // namespace fneus

using namespace fneus;

extern "C" int fneus_lvis_visibility(const void* lvis_blob, const float* points, const float* normals, const float* dirs,
                                     const float* weights, const unsigned char* point_mask, int n_pts, int n_lobes, int n_dirs,
                                     float* vis, int prec, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (n_pts <= 0 || n_lobes <= 0) return 0;
    if (!lvis_blob || !points || !normals || !dirs || !weights || !vis) {
        set_last_error("fneus_lvis_visibility: null argument");
        return -2;
    }
    if (n_dirs != 32) {
        set_last_error("fneus_lvis_visibility: 32 directions per lobe (one MFMA tile per (point, lobe) pair)");
        return -2;
    }
    const unsigned char* b = reinterpret_cast<const unsigned char*>(lvis_blob);
    const long items = (long)n_pts * ((n_lobes + fneus::kLvisChunk - 1) / fneus::kLvisChunk);
    const unsigned grid = (unsigned)(items < 8192 ? items : 8192);
    // two-pass pipelined kernel on 8-wave workgroups (lvis_p2_kernels.hip); FNEUS_LVIS_P2=0: the 4-wave kernels of this file
    const char* p2_env = getenv("FNEUS_LVIS_P2");
    if (prec == 2 || ((p2_env ? atoi(p2_env) : FNEUS_LVIS_P2_DEFAULT) != 0 && (prec == 3 || prec == 1)))      // (prec 2 exists in this form only)
        return fneus::lvis_visibility_p2(b, points, normals, dirs, weights, point_mask, n_pts, n_lobes, vis, prec, stream);
    const char* env = getenv("FNEUS_LVIS_HB");            // 2 (default): two lobes share a pass over the weights; 1: one lobe per pass
    if (!(env && env[0] == '1')) {
        if (prec == 3) {
            static bool done = false;
            if (!done) { fneus::allow_big_lds(lvis_visibility_tph_kernel<3>); done = true; }
            hipLaunchKernelGGL(lvis_visibility_tph_kernel<3>, dim3(grid), dim3(256), fneus::kLvisLds2, stream, b, points, normals, dirs,
                               weights, point_mask, n_pts, n_lobes, vis);
        } else if (prec == 1) {
            static bool done = false;
            if (!done) { fneus::allow_big_lds(lvis_visibility_tph_kernel<1>); done = true; }
            hipLaunchKernelGGL(lvis_visibility_tph_kernel<1>, dim3(grid), dim3(256), fneus::kLvisLds2, stream, b, points, normals, dirs,
                               weights, point_mask, n_pts, n_lobes, vis);
        } else {
            return -2;
        }
        return fneus::launch_status();
    }
    if (prec == 3) {
        static bool done = false;
        if (!done) { fneus::allow_big_lds(lvis_visibility_tp_kernel<3>); done = true; }
        hipLaunchKernelGGL(lvis_visibility_tp_kernel<3>, dim3(grid), dim3(256), fneus::kLvisLds, stream, b, points, normals, dirs,
                           weights, point_mask, n_pts, n_lobes, vis);
    } else if (prec == 1) {
        static bool done = false;
        if (!done) { fneus::allow_big_lds(lvis_visibility_tp_kernel<1>); done = true; }
        hipLaunchKernelGGL(lvis_visibility_tp_kernel<1>, dim3(grid), dim3(256), fneus::kLvisLds, stream, b, points, normals, dirs,
                           weights, point_mask, n_pts, n_lobes, vis);
    } else {
        return -2;
    }
    return fneus::launch_status();
}

// ---- the Lvis blob for prec 2: every forward weight fragment as ONE fp16 value per weight (the sum of its bf16 hi + lo parts,
// rounded once), in the place of the hi fragment; biases, the last layer's fp32 row and everything else copied.  A one-off per
// parameter change (stage 3 keeps the network frozen).
namespace fneus {
__global__ void __launch_bounds__(256) lvis_h16_pack_kernel(const unsigned char* __restrict__ blob, unsigned char* __restrict__ out) {
    constexpr auto& LY = kLvisLayout;
    const uint32_t off = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (off >= LY.total) return;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(blob + off);
#pragma unroll
    for (int l = 0; l < kLvisLayers; ++l) {
        const uint32_t bytes = (uint32_t)(kLvisGeom[l].ksf * kLvisGeom[l].ntf) * kFragBytes;
        if (off >= LY.L[l].fwd_hi && off < LY.L[l].fwd_hi + bytes) {
            const bf16x8 lo = *reinterpret_cast<const bf16x8*>(blob + LY.L[l].fwd_lo + (off - LY.L[l].fwd_hi));
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = to16<2>((float)v[e] + (float)lo[e]);
        }
    }
    *reinterpret_cast<bf16x8*>(out + off) = v;
}
}  // namespace fneus

extern "C" size_t fneus_lvis_blob_bytes(void) { return fneus::kLvisLayout.total; }

extern "C" int fneus_lvis_h16_pack(const void* lvis_blob, void* out, fneus_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    fneus::clear_status();
    if (!lvis_blob || !out) {
        fneus::set_last_error("fneus_lvis_h16_pack: blob and out must be given");
        return -2;
    }
    const unsigned n16 = (fneus::kLvisLayout.total + 15) / 16;
    hipLaunchKernelGGL(fneus::lvis_h16_pack_kernel, dim3((n16 + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const unsigned char*>(lvis_blob),
                       reinterpret_cast<unsigned char*>(out));
    return fneus::launch_status();
}
