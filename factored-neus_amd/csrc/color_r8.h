// launcher of the colour network's backward on resident-weight 8-wave workgroups (color_r8_kernels.hip), called by fneus_color_bwd
#pragma once
#include "fneus_kernels.h"

namespace fneus {

// chip-filling launches whose planes stay within 32-bit buffer offsets; hi + lo planes iff st.zbar_lo
int color_bwd_r8(const unsigned char* blob, long n_pts, const float* d_rgb, const float* rgb, const ColStash& st, float* d_feat,
                 float* d_normal, int prec, hipStream_t stream);

}  // namespace fneus
