// Internal kernel-argument structs (device side view of the C ABI structs in include/fneus.h).
#pragma once
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

struct SdfStash {
    __bf16* pe_hi;   __bf16* pe_lo;
    __bf16* h_hi;    __bf16* h_lo;
    __bf16* a_hi;    __bf16* a_lo;
    __bf16* feat_hi; __bf16* feat_lo;
    SdfStash() = default;
    SdfStash(const FneusSdfStash& s)
        : pe_hi((__bf16*)s.pe_hi), pe_lo((__bf16*)s.pe_lo), h_hi((__bf16*)s.h_hi), h_lo((__bf16*)s.h_lo),
          a_hi((__bf16*)s.a_hi), a_lo((__bf16*)s.a_lo), feat_hi((__bf16*)s.feat_hi), feat_lo((__bf16*)s.feat_lo) {}
};

void set_last_error(const char* msg);

}  // namespace fneus
