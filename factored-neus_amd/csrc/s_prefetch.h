// Scalar prefetch: bring lines of a read-once operand block into L2 ahead of the vector loads that consume it.
//
// A wave's vector-memory operations return in order, so an HBM load issued ahead of time delays every weight fragment requested
// behind it.  Scalar loads are counted separately from them (lgkmcnt, out of order): 64 `s_load_dword` at a
// 128-byte stride touch 8 KiB of the block through the scalar cache into L2 without entering the vmcnt queue; all of them
// write ONE dummy SGPR, which the caller hands back to s_prefetch_done() behind a point where lgkmcnt is 0 anyway (a barrier),
// so that the compiler keeps that register reserved while the loads are in flight.  LDS waits of the compiler (lgkmcnt(N))
// only become more conservative: at least the operations it counted have returned when the counter is down to N.
// MEASURED (K2's reverse sweep, -DFNEUS_K2_SPF): slower, 307-316 us against 284 -- "more conservative" means that the first LDS
// waits of the dense phase wait for the scalar loads as well, i.e. for HBM.  Kept as a record of the experiment.
#pragma once
#include <stdint.h>
#include "fneus_common.h"

namespace fneus {

// 8 KiB starting at the (wave-uniform) address p
FN_DEV uint32_t s_prefetch_8k(const void* p) {
    uint32_t d;
    asm volatile(
                 "s_load_dword %0, %1, 0\n\t"
                 "s_load_dword %0, %1, 128\n\t"
                 "s_load_dword %0, %1, 256\n\t"
                 "s_load_dword %0, %1, 384\n\t"
                 "s_load_dword %0, %1, 512\n\t"
                 "s_load_dword %0, %1, 640\n\t"
                 "s_load_dword %0, %1, 768\n\t"
                 "s_load_dword %0, %1, 896\n\t"
                 "s_load_dword %0, %1, 1024\n\t"
                 "s_load_dword %0, %1, 1152\n\t"
                 "s_load_dword %0, %1, 1280\n\t"
                 "s_load_dword %0, %1, 1408\n\t"
                 "s_load_dword %0, %1, 1536\n\t"
                 "s_load_dword %0, %1, 1664\n\t"
                 "s_load_dword %0, %1, 1792\n\t"
                 "s_load_dword %0, %1, 1920\n\t"
                 "s_load_dword %0, %1, 2048\n\t"
                 "s_load_dword %0, %1, 2176\n\t"
                 "s_load_dword %0, %1, 2304\n\t"
                 "s_load_dword %0, %1, 2432\n\t"
                 "s_load_dword %0, %1, 2560\n\t"
                 "s_load_dword %0, %1, 2688\n\t"
                 "s_load_dword %0, %1, 2816\n\t"
                 "s_load_dword %0, %1, 2944\n\t"
                 "s_load_dword %0, %1, 3072\n\t"
                 "s_load_dword %0, %1, 3200\n\t"
                 "s_load_dword %0, %1, 3328\n\t"
                 "s_load_dword %0, %1, 3456\n\t"
                 "s_load_dword %0, %1, 3584\n\t"
                 "s_load_dword %0, %1, 3712\n\t"
                 "s_load_dword %0, %1, 3840\n\t"
                 "s_load_dword %0, %1, 3968\n\t"
                 "s_load_dword %0, %1, 4096\n\t"
                 "s_load_dword %0, %1, 4224\n\t"
                 "s_load_dword %0, %1, 4352\n\t"
                 "s_load_dword %0, %1, 4480\n\t"
                 "s_load_dword %0, %1, 4608\n\t"
                 "s_load_dword %0, %1, 4736\n\t"
                 "s_load_dword %0, %1, 4864\n\t"
                 "s_load_dword %0, %1, 4992\n\t"
                 "s_load_dword %0, %1, 5120\n\t"
                 "s_load_dword %0, %1, 5248\n\t"
                 "s_load_dword %0, %1, 5376\n\t"
                 "s_load_dword %0, %1, 5504\n\t"
                 "s_load_dword %0, %1, 5632\n\t"
                 "s_load_dword %0, %1, 5760\n\t"
                 "s_load_dword %0, %1, 5888\n\t"
                 "s_load_dword %0, %1, 6016\n\t"
                 "s_load_dword %0, %1, 6144\n\t"
                 "s_load_dword %0, %1, 6272\n\t"
                 "s_load_dword %0, %1, 6400\n\t"
                 "s_load_dword %0, %1, 6528\n\t"
                 "s_load_dword %0, %1, 6656\n\t"
                 "s_load_dword %0, %1, 6784\n\t"
                 "s_load_dword %0, %1, 6912\n\t"
                 "s_load_dword %0, %1, 7040\n\t"
                 "s_load_dword %0, %1, 7168\n\t"
                 "s_load_dword %0, %1, 7296\n\t"
                 "s_load_dword %0, %1, 7424\n\t"
                 "s_load_dword %0, %1, 7552\n\t"
                 "s_load_dword %0, %1, 7680\n\t"
                 "s_load_dword %0, %1, 7808\n\t"
                 "s_load_dword %0, %1, 7936\n\t"
                 "s_load_dword %0, %1, 8064\n\t"
                 : "=&s"(d)
                 : "s"(p)
                 : "memory");
    return d;
}
FN_DEV void s_prefetch_done(uint32_t d) { asm volatile("" ::"s"(d)); }

}  // namespace fneus
