// The BSP builders' result (rt_bsp of rt_host.h), shared by the host builder (rt_host.cpp) and the device builder
// (rt_bspgpu.hip).  Not part of the C ABI.
#ifndef RT_BSP_INTERNAL_H
#define RT_BSP_INTERNAL_H

#include "rt_host.h"

#include <cstdint>
#include <vector>

struct rt_bsp {
    float root_box[6];
    int32_t root_index = 0;
    std::vector<rt_inode> inodes;
    std::vector<uint32_t> leaf_first;   // leaf_count + 1
    std::vector<rt_tri_ref> refs;
    uint32_t depth = 0, oversized = 0;
};

#endif
