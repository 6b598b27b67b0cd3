// A scene description flattened into the arrays the kernels read (host memory; rt_scene_create uploads them).
// Shared by rt_flatten.cpp (which fills it) and rt_device.hip (which uploads it).  Not part of the C ABI.
#ifndef RT_FLATTEN_H
#define RT_FLATTEN_H

#include "rt_amd.h"

#include <cstdint>
#include <vector>

struct rt_flat_scene {
    // `packed` holds the leaf bounds records (64 bytes each, in the form leafAxes says, see rt_leaf_bounds.h), the record of
    // node slot s at byte 64 * (frontSlots - 1 - s), then the node records (32 bytes each).  The kernels' node pointer is
    // packed + 64 * frontSlots bytes; the bounds record of slot s then sits 64 * (s + 1) bytes in front of it.  frontSlots
    // equals nodeSlots.
    std::vector<float> packed;
    uint32_t nodeSlots = 0, frontSlots = 0;
    uint32_t leafAxes = 1;              // RT_LEAF_AXES_ONE (cone form) or RT_LEAF_AXES_PACKED (axes form)
    std::vector<float> leafTris;        // 12 floats per (leaf, triangle) reference, plus one spare record
    std::vector<uint32_t> leafIds;      // {model, tri, scene-wide triangle, 0} per reference
    std::vector<float> shadeRecs;       // 12 floats per reference: {model, vn1} {vn2, vn3.x} {vn3.yz, 0, 0}
    uint32_t refCount = 0, depth = 0;
    float rootLo[3] = {0, 0, 0}, rootHi[3] = {0, 0, 0};
    float boxAbsMax = 0.0f;             // largest |coordinate| of the root box
};

// Validates the description and fills `out`.  Returns RT_OK or an rt_status with rt_last_error() set.
int rt_internal_flatten(const rt_scene_desc* D, rt_flat_scene& out);
// The three instantiated vertex normals of every triangle (9 floats per scene-wide triangle index).
int rt_internal_gather_tri_normals(const rt_scene_desc* D, const char* who, std::vector<float>& out);
// One 48-byte shading record.
void rt_internal_shade_record(float* rec, uint32_t model, const float* normals9);
int rt_internal_fail(int code, const char* message);

#endif
