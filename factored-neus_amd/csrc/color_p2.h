// launcher of the colour network's forward in the two-pass pipelined form (color_p2_kernels.hip), called by fneus_color_fwd
#pragma once
#include "fneus_kernels.h"

namespace fneus {

// mode 0: inference, 1 / 3: training with bf16 / hi + lo planes
int color_fwd_p2(const unsigned char* blob, const PointSrc& src, long n_pts, const float* dirs, const float* normal, const float* feat,
                 const ColStash& st, float* rgb_out, int prec, int mode, hipStream_t stream);

}  // namespace fneus
