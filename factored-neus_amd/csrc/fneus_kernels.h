// Internal kernel-argument structs (device side view of the C ABI structs in include/fneus.h).
#pragma once
#include <string.h>
#include "fneus_common.h"
#include "../../include/fneus.h"

namespace fneus {

struct PointSrc {
    const float* pts;      // [N][3] or nullptr
    const float* rays_o;   // [B][3]
    const float* rays_d;   // [B][3]
    const float* t;        // [N]   (ray mode: p = o[n/m] + d[n/m]*t[n])
    int m;
};

// device-side views of FneusSdfStash / FneusSdfBwdBufs (include/fneus.h): fragment planes as byte pointers
struct SdfStash {
    unsigned char *pe_hi, *pe_lo;     // [tiles][4 KiB]
    unsigned char *h_hi, *h_lo;       // [8][tiles][16 KiB]
    unsigned char *a_hi, *a_lo;       // [8][tiles][16 KiB]
    unsigned char *feat_hi, *feat_lo; // [tiles][16 KiB]     feature vector (colour-network input)
    unsigned char* ps;                // sigma'(z_l), u16 fixed point: [tiles][8][16 KiB]
    f32x16* qs;                       // q_skip scratch of K2's reverse sweep: [tiles][2][64] accumulator tiles (lane-private)
    SdfStash() = default;
    SdfStash(const FneusSdfStash& s)
        : pe_hi((unsigned char*)s.pe_hi), pe_lo((unsigned char*)s.pe_lo), h_hi((unsigned char*)s.h_hi),
          h_lo((unsigned char*)s.h_lo), a_hi((unsigned char*)s.a_hi), a_lo((unsigned char*)s.a_lo),
          feat_hi((unsigned char*)s.feat_hi), feat_lo((unsigned char*)s.feat_lo), ps((unsigned char*)s.ps), qs((f32x16*)s.qs) {}
};

struct SdfBwdBufs {
    unsigned char *qbar_hi, *qbar_lo;   // [tiles][4 KiB]      adj_0 = J nbar
    unsigned char *adj_hi, *adj_lo;     // [8][tiles][16 KiB]  slot l = adj_{l+1}
    unsigned char *zbar_hi, *zbar_lo;   // [9][tiles][16 KiB]  slot l = zbar_l (slot 8: feature rows of the last layer)
    unsigned char *zsdf_hi, *zsdf_lo;   // [tiles][2 KiB]      feature 0 = dL/dsdf
    unsigned char *c_hi, *c_lo;         // [tiles][8][16 KiB]  coupling terms, lane-private
    SdfBwdBufs() = default;
    SdfBwdBufs(const FneusSdfBwdBufs& s)
        : qbar_hi((unsigned char*)s.qbar_hi), qbar_lo((unsigned char*)s.qbar_lo), adj_hi((unsigned char*)s.adj_hi),
          adj_lo((unsigned char*)s.adj_lo), zbar_hi((unsigned char*)s.zbar_hi), zbar_lo((unsigned char*)s.zbar_lo),
          zsdf_hi((unsigned char*)s.zsdf_hi), zsdf_lo((unsigned char*)s.zsdf_lo), c_hi((unsigned char*)s.c_hi),
          c_lo((unsigned char*)s.c_lo) {}
};

struct ColStash {                       // fragment planes (include/fneus.h FneusColStash)
    unsigned char *side_hi, *side_lo;   // [tiles][4 KiB]      pts | PE4(view) | normal (33 of 64 features)
    unsigned char *u_hi, *u_lo;         // [4][tiles][16 KiB]  slot l = relu output of layer l (= input of l+1)
    unsigned char *zbar_hi, *zbar_lo;   // [4][tiles][16 KiB]  slot l = dL/dz_l
    unsigned char *zout_hi, *zout_lo;   // [tiles][2 KiB]      dL/dz of the output layer (3 features)
    u32x4* mask;                        // lane-private ReLU masks: [tiles][4][64] x 128 bits
    unsigned char *feat_hi, *feat_lo;   // [tiles][16 KiB]     surface head only: its (gathered) input features
    unsigned char* dfeat_hi;            // [tiles][16 KiB]     color_bwd without d_feat rows: the feature cotangent as bf16 fragments
    int dnormal_add;                    // color_bwd_r8: d_normal += instead of =
    ColStash() { memset(this, 0, sizeof(*this)); }
    ColStash(const FneusColStash& s)
        : side_hi((unsigned char*)s.side_hi), side_lo((unsigned char*)s.side_lo), u_hi((unsigned char*)s.u_hi),
          u_lo((unsigned char*)s.u_lo), zbar_hi((unsigned char*)s.zbar_hi), zbar_lo((unsigned char*)s.zbar_lo),
          zout_hi((unsigned char*)s.zout_hi), zout_lo((unsigned char*)s.zout_lo), mask((u32x4*)s.mask),
          feat_hi((unsigned char*)s.feat_hi), feat_lo((unsigned char*)s.feat_lo), dfeat_hi((unsigned char*)s.dfeat_hi), dnormal_add(s.dnormal_add) {}
};

struct NerfStash {      // fragment planes (fneus_pp.h): [tiles][F fragments][64 slots][8 bf16]; *_lo NULL unless gradient precision 3
    unsigned char *pe_hi, *pe_lo;       // F = 6   PE10 of the 4-D background point (84 features used)
    unsigned char *h_hi, *h_lo;         // [8][tiles][16] slot l = relu output of pts_linears.l
    unsigned char *feat_hi, *feat_lo;   // F = 16  feature_linear output
    unsigned char *dpe_hi, *dpe_lo;     // F = 2   PE4 of the view direction (27 features used)
    unsigned char *hv_hi, *hv_lo;       // F = 8   relu output of views_linears.0
    u32x4* mask;                        // lane-private ReLU masks: [tiles][9][64] x 128 bits (slot 8: views layer)
    unsigned char *zbar_hi, *zbar_lo;   // [8][tiles][16] slot l = dL/dz of pts_linears.l     (written by the backward)
    unsigned char *zfeat_hi, *zfeat_lo; // F = 16  dL/d feature
    unsigned char *zhv_hi, *zhv_lo;     // F = 8   dL/dz of views_linears.0
    unsigned char *zout_hi, *zout_lo;   // F = 4   fragments 0, 1: rows 0..2 = dL/d rgb; fragments 2, 3: row 0 = dL/d density
    NerfStash() { memset(this, 0, sizeof(*this)); }
    NerfStash(const FneusNerfStash& s)
        : pe_hi((unsigned char*)s.pe_hi), pe_lo((unsigned char*)s.pe_lo), h_hi((unsigned char*)s.h_hi), h_lo((unsigned char*)s.h_lo),
          feat_hi((unsigned char*)s.feat_hi), feat_lo((unsigned char*)s.feat_lo), dpe_hi((unsigned char*)s.dpe_hi),
          dpe_lo((unsigned char*)s.dpe_lo), hv_hi((unsigned char*)s.hv_hi), hv_lo((unsigned char*)s.hv_lo), mask((u32x4*)s.mask),
          zbar_hi((unsigned char*)s.zbar_hi), zbar_lo((unsigned char*)s.zbar_lo), zfeat_hi((unsigned char*)s.zfeat_hi),
          zfeat_lo((unsigned char*)s.zfeat_lo), zhv_hi((unsigned char*)s.zhv_hi), zhv_lo((unsigned char*)s.zhv_lo),
          zout_hi((unsigned char*)s.zout_hi), zout_lo((unsigned char*)s.zout_lo) {}
};

FN_DEV void load_point(const PointSrc& s, long n, float (&x)[3]) {
    if (s.pts) {
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = s.pts[n * 3 + c];
    } else {
        const long ray = n / s.m;
        const float t = s.t[n];
#pragma unroll
        for (int c = 0; c < 3; ++c)   // mul then add, separately rounded, like torch (renderer.py:233, 428)
            x[c] = __fadd_rn(s.rays_o[ray * 3 + c], __fmul_rn(s.rays_d[ray * 3 + c], t));
    }
}

void set_last_error(const char* msg);

// the fused-MLP kernels use 130 KiB of dynamic LDS (> the 64 KiB default limit)
template <class K>
inline void allow_big_lds(K kernel) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// HIP keeps a sticky per-thread "last error" that other libraries (e.g. PyTorch's own runtime probing) may have set:
// clear it before a launch, then report only what this launch produced.
inline void clear_status() { (void)hipGetLastError(); }
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    set_last_error(hipGetErrorString(e));
    return -1;
}

}  // namespace fneus
