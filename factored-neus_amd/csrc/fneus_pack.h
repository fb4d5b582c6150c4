// Pack-job descriptor shared by pack.hip and the host (fneus/netdesc.py mirrors it as a numpy dtype).
#pragma once
#include <stdint.h>

namespace fneus {

enum { PACK_FRAG = 0, PACK_ACCVEC = 1 };

struct PackJob {
    int32_t kind;        // PACK_FRAG / PACK_ACCVEC
    int32_t unit_base;   // first work unit (block) of this job
    uint32_t dst_hi;     // byte offset in the blob (hi plane, or the fp32 vector)
    uint32_t dst_lo;     // byte offset of the lo plane (PACK_FRAG)
    uint32_t src;        // float offset of the source matrix/vector in the flat parameter buffer
    int32_t ld;          // leading dimension of the source (PACK_FRAG: in_dim; PACK_ACCVEC: element stride)
    int32_t ks, nt;      // fragment grid (PACK_FRAG) / nt tiles (PACK_ACCVEC)
    int32_t transposed;  // 0: A[row][k] = W[row][k]   1: A[row][k] = W[k][row]
    uint32_t rowmap;     // int offset into maps: nt*32 entries, source row (or -1)
    uint32_t kmap;       // int offset into maps: ks*16 entries in k-slot order, source k (or -1)
    float scale;
    int32_t rs_base;     // first entry of this layer in the row-scale table (g/||v|| per output row), or -1
    int32_t rs_mode;     // 0: scale by the fragment row, 1: by the k index (transposed packs), 2: by rs_base itself
    int32_t geom;        // 0: 32-row tiles / 16-deep k-steps (mfma 32x32x16)   1: 16-row tiles / 32-deep (mfma 16x16x32)
    int32_t pad;
};

// One weight-normalised row: v[row][0..n_in), g[row]  ->  rowscale = g/||v||, inv_norm = 1/||v||
struct RowInfo {
    uint32_t off_v;      // float offset of the row of weight_v in the raw parameter buffer
    uint32_t off_g;      // float offset of weight_g[row]; 0xFFFFFFFF = plain Linear (scale 1)
    int32_t n_in;
    uint32_t off_w_eff;  // float offset of the same row in the effective-parameter (gradient) buffer
};

}  // namespace fneus
