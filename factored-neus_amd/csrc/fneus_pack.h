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
};

}  // namespace fneus
