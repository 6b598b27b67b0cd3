/* rt_amd.h -- C ABI of the MI355X path-tracing core (librt_amd.so).
 *
 * This is the drop-in boundary for the reference renderer's hot path.  The reference has no plugin
 * ABI of its own: its hot path sits behind the inline C++ function
 *     void render(RenderData const& data, Span<glm::vec3> image)         src/render.hpp:215
 * whose only caller is src/main.cpp:230, and behind
 *     BSPTree::lineTriNearestIntersection<Surfaces>(Line const&, float)  src/bsp.hpp:50-51
 * Every entry point below names the reference interface it stands in for.  Plain pointers and
 * sizes only; no C++ or torch types cross this boundary.  All functions return 0 on success and a
 * negative rt_status otherwise; rt_last_error() gives the message for the calling thread.
 *
 * Threading: a scene handle may be used by one host thread at a time; use one handle per GPU.
 * All "device" pointers are HIP device pointers on the scene's GPU.  Every call on a handle shares the
 * handle's workspace: a call that arrives on a different hip_stream than the previous one first waits
 * for that previous stream.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RT_OK = 0,
    RT_ERR_INVALID = -1,      /* bad argument / inconsistent description */
    RT_ERR_HIP = -2,          /* a HIP runtime call failed */
    RT_ERR_NO_DEVICE = -3,    /* no usable GPU */
    RT_ERR_LIMIT = -4,        /* tree deeper than the traversal stack, size overflow */
    RT_ERR_NOMEM = -5
} rt_status;

/* One interior node exactly as the reference stores it (BSPTree::INode, src/bsp.hpp:148-159):
 * two child Nodes {box min xyz, box max xyz, index} and the division axis.  60 bytes.
 * index > 0: interior node (index-1);  index < 0: leaf (-index-1);  index == 0: empty. */
typedef struct {
    float neg_box[6]; int32_t neg_index;
    float pos_box[6]; int32_t pos_index;
    uint8_t axis; uint8_t pad_[3];
} rt_inode;

/* {model (mesh instance) index, triangle index within the model}: MeshTriIndex, src/mesh.hpp:50-53,
 * widened from 16 to 32 bits. */
typedef struct { uint32_t mesh, tri; } rt_tri_ref;

/* Scene description = BSPTree contents (src/bsp.hpp:170-172) + RayTraceData (src/render.hpp:28-35)
 * with the model->mesh and model->material permutations already applied.  The caller keeps
 * ownership; rt_scene_create copies what it needs into HBM. */
typedef struct {
    /* BSPTree::_root, _inodes, _leaves */
    float root_box[6];
    int32_t root_index;
    const rt_inode* inodes;
    uint32_t inode_count;
    const uint32_t* leaf_first_ref;      /* leaf_count+1 offsets into leaf_refs (leaf order preserved) */
    const rt_tri_ref* leaf_refs;         /* Leaf::triIndices of every leaf, concatenated */
    uint32_t leaf_count;
    /* Scene::preprocessedTris (src/mesh.hpp:141-144): 12 floats {normal, v1, v1ToV3, v1ToV2} */
    const float* pretris;
    const uint32_t* model_pretri_begin;  /* model_count+1 offsets: model m owns [begin[m], begin[m+1]) */
    uint32_t pretri_count;
    /* RayTraceData::vertexNormals / vertexRanges (instantiated meshes) */
    const float* vertex_normals;
    const uint32_t* model_vertex_begin;  /* model_count+1 offsets into vertex_normals */
    uint32_t vertex_count;
    /* RayTraceData::tris / triRanges (base meshes; indices relative to the mesh's vertex range) */
    const uint32_t* base_tris;           /* 3 per triangle */
    const uint32_t* model_tri_begin;     /* per model: first base triangle of its mesh */
    uint32_t base_tri_count;
    /* RayTraceData::materials: PreprocessedMaterial per model (src/material.hpp:23-29), 12 floats
     * {ndfAlphaSq, geometryAlphaSq, f0[3], adjustedColour[3], emission[3], 0} */
    const float* model_materials;
    uint32_t model_count;
    /* Optional alternative to pretris: the same 12 floats per entry of leaf_refs (parallel array), which
     * is what Leaf::tris holds in the reference (src/bsp.hpp:161-168); when non-null, pretris may be null
     * (model_pretri_begin and pretri_count are still required: they number the triangles). */
    const float* leaf_pretris;
} rt_scene_desc;

typedef struct rt_scene rt_scene;

/* One node of the alternative accelerator, a 4-wide bounding volume hierarchy (no counterpart in the
 * reference, whose only accelerator is BSPTree, src/bsp.hpp:35-48).  64 bytes.  The boxes of up to four
 * children are stored as 8-bit offsets from the node's own lower corner: plane = origin[a] + q * 2^(exp[a]-127),
 * rounded outwards by the builder, so a stored box always contains the child's triangles.
 * child[k] == 0xFFFFFFFF: no child; bit 31 set: a leaf, triangles [first, first+count) of the hierarchy's
 * triangle order with first = bits 0-27 and count-1 = bits 28-30; otherwise the number of an interior node,
 * always larger than the node's own number (node 0 is the root). */
typedef struct {
    float origin[3];
    uint8_t exp[3];               /* per axis: IEEE-754 exponent field of the quantisation step */
    uint8_t pad0_;
    uint8_t qlo_x[4], qlo_y[4], qlo_z[4], qhi_x[4], qhi_y[4], qhi_z[4];      /* byte k: child k */
    uint32_t pad1_[2];
    uint32_t child[4];
} rt_bvh4_node;

/* LineMeshIntersection (src/bsp.hpp:23-29).  mesh == 0xFFFFFFFF marks "no intersection"
 * (std::nullopt in the reference). */
typedef struct {
    float t, u, v;            /* t, pointCoord2, pointCoord3 */
    float point[3];
    uint32_t mesh, tri;
} rt_hit;

#define RT_FLAG_COUNT_WORK      1u   /* fill the work counters of rt_stats (slower kernels) */
#define RT_FLAG_PROFILE_KERNELS 2u   /* time every kernel launch with HIP events (adds syncs at the end) */
/* Russian roulette, an estimator extension the reference does not have (its paths end only at the bounce
 * limit, src/render.hpp:117): from the third bounce on a path survives with probability
 * p = min(1, largest throughput component), drawn from the bounce's own Philox block, and its throughput
 * is divided by p.  The converged image is unchanged; fewer long paths are traced. */
#define RT_FLAG_RUSSIAN_ROULETTE 4u
/* Answer the path loop's nearest-intersection queries (src/render.hpp:104) from the bounding volume
 * hierarchy attached with rt_scene_attach_bvh4 instead of the BSP tree.  Same triangle test, bit for bit
 * (src/geometry.hpp:99-132); the answer is the nearest accepted triangle of the whole scene whose hit point
 * lies in the root box (ties: lowest triangle index).  Of the BSP leaf's in-box rule (src/bsp.hpp:82-87) only
 * the root box is left, so the two accelerators can differ on a ray whose hit point rounds to just outside an
 * inner leaf's box and inside no other leaf that holds the triangle: none in 469 000 rays of the built-in
 * scene, a few in 100 000 on large meshes.  A second difference: the hierarchy's boxes are the triangles' corners with a
 * fixed padding, while the reference's triangle test accepts a line that grazes a triangle's plane (|normal . direction|
 * just above 1e-6) many edge lengths outside the triangle; such hits are found by the BSP path (whose leaf bounds widen
 * as the line approaches the plane) and can be missed here -- a set of measure about 1e-7 of random rays.  So this is an
 * alternative accelerator with the reference's converged image, not a bit-exact one.  The hierarchy's box test treats a
 * direction component below 2^-80 in magnitude as zero (the line as parallel to that axis): harmless for a direction
 * of ordinary length, where such a component moves the line by nothing over any finite distance, but a direction
 * that is that short as a whole is outside this accelerator's domain (the BSP path has no such limit). */
#define RT_FLAG_ACCEL_BVH 8u
/* Per-lobe sampling, an estimator extension the reference does not have (it draws every continuation direction
 * from the GGX lobe, src/render.hpp:145-167): the GGX half vector with probability pS, a cosine-weighted direction
 * otherwise, pS = sum(f0) / (sum(f0) + pi * sum(adjustedColour)) held within [0.1, 0.9] (1 for a material without a
 * diffuse part), weighted by BRDF * cos / (pS pdfS + (1 - pS) pdfD).  Same expectation, less variance on glossy
 * dielectrics.  The lobe is chosen with the fourth word of the bounce's Philox block. */
#define RT_FLAG_LOBE_SAMPLING 16u
/* Before every bounce but the first, sort the queue of live paths by the cell of the ray's origin (32 x 32 x 32 grid
 * over the root box, Morton order) and the octant of its direction, so that the lanes of a wave start in the same part
 * of the tree.  Only the order in which paths are traced changes: the image is the same bit for bit.  Costs one
 * stream synchronisation per bounce and 8 more bytes of workspace per path; off by default because on the scenes
 * measured the sort costs more than the traversal gains.  Not available in rt_render_device_static.  No counterpart
 * in the reference. */
#define RT_FLAG_SORT_RAYS 32u
/* For the front_only argument of rt_intersect / rt_intersect_device: bit 0 = front faces only
 * (Surfaces::FRONT_ONLY, src/geometry.hpp:95-97), bit 1 = use the attached bounding volume hierarchy. */
#define RT_INTERSECT_FRONT_ONLY 1
#define RT_INTERSECT_BVH 2

/* RenderData (src/render.hpp:38-44) + the compile-time constants of src/render.hpp:47-49 made
 * run-time, + a tile rectangle for multi-GPU sharding. */
typedef struct {
    uint32_t width, height;       /* full image; defines the pixel index y*width+x used by the RNG */
    uint32_t x0, y0, w, h;        /* rectangle to render; output is row-major w*h*3 floats */
    uint32_t spp;                 /* PIXEL_SAMPLE_RATE */
    uint32_t bounce_limit;        /* RAY_BOUNCE_LIMIT */
    float t_min;                  /* RAY_INTERSECTION_T_MIN */
    uint64_t seed;
    float camera_position[3];
    float pixel_to_ray[9];        /* column-major mat3 {right, down, o}, src/camera.hpp:45 */
    uint32_t flags;
    uint32_t max_batch_paths;     /* paths traced together; 0 = as many as the call has, within 2^30 paths and half
                                   * of the GPU's memory (80 bytes of workspace per path) */
    /* Interleaved row groups, for sharding one frame over several GPUs with a single call per GPU: with
     * row_stride > 1, row r of the output (0 <= r < h) is image row y0 + (r / row_group) * row_group * row_stride
     * + r % row_group, i.e. this call renders every row_stride-th group of row_group rows, starting at y0.
     * row_stride 0 or 1: the plain rectangle. */
    uint32_t row_group, row_stride;
} rt_render_params;

typedef struct {
    uint64_t paths;
    uint64_t segments;            /* closest-hit queries issued by the path loop, src/render.hpp:104 */
    uint64_t shaded_hits;
    uint64_t box_tests, tris_tested, leaf_visits, inode_visits;   /* RT_FLAG_COUNT_WORK only */
    double ms_total;              /* whole call, host clock, after the final sync */
    double ms_raygen, ms_trace, ms_shade, ms_accumulate;          /* RT_FLAG_PROFILE_KERNELS only */
    uint32_t trace_launches, batches;
    /* RT_FLAG_COUNT_WORK only: box tests the conservative slab filter left to the exact six-face test, and
     * decisions of the kernels' conservative shortcuts that disagreed with the exact evaluation (the counting
     * kernels evaluate both: the slab filter against the six-face test on every box, the leaf normal cone against
     * the facing test of every triangle; anything but 0 is a bug). */
    uint64_t box_exact_fallbacks, box_filter_errors;
} rt_stats;

/* Upload a scene to GPU `device` (replaces constructing RayTraceData, src/main.cpp:219-229). */
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);
/* A scene description as one contiguous, position-independent byte buffer, for handing a prepared scene to
 * other processes (one process per GPU: rank 0 prepares the scene and broadcasts the buffer once; the
 * reference is a single process and has no counterpart).  rt_scene_serialize with blob == NULL stores the
 * size needed in *size; otherwise *size is the capacity on entry and the bytes written on return.
 * rt_scene_desc_from_blob points a description at the arrays inside the buffer (no copy; the buffer must be
 * 8-byte aligned and outlive the description); rt_scene_create_from_blob uploads straight from it. */
int rt_scene_serialize(const rt_scene_desc* desc, void* blob, uint64_t* size);
int rt_scene_desc_from_blob(const void* blob, uint64_t size, rt_scene_desc* out_desc);
int rt_scene_create_from_blob(const void* blob, uint64_t size, int device, rt_scene** out);
/* Attach a 4-wide bounding volume hierarchy over desc->pretris (which must be non-null) to an uploaded
 * scene: tri_order[k] is the index into desc->pretris of the hierarchy's k-th triangle (a permutation of
 * 0..pretri_count-1).  The nodes are validated (child numbers ascending, leaf ranges inside the triangle
 * order, at most 64 deferred children on any descent) before anything is uploaded.
 * rt_scene_build_bvh4 builds one with rt_bvh4_build (rt_host.h) and attaches it; max_leaf_tris 1..8,
 * threads as for rt_bsp_build. */
int rt_scene_attach_bvh4(rt_scene* scene, const rt_scene_desc* desc, const rt_bvh4_node* nodes, uint32_t node_count,
                         const uint32_t* tri_order);
int rt_scene_build_bvh4(rt_scene* scene, const rt_scene_desc* desc, uint32_t max_leaf_tris, int threads);
/* Bytes of HBM held by the scene, and the depth of its tree. */
int rt_scene_info(const rt_scene* scene, uint64_t* device_bytes, uint32_t* node_count, uint32_t* leaf_tri_count,
                  uint32_t* depth);

/* BSPTree::lineTriNearestIntersection<Surfaces> (src/bsp.hpp:50-145) for n lines.
 * origins/directions: 3 floats per line.  Host-pointer and device-pointer forms. */
int rt_intersect(rt_scene* scene, const float* origins, const float* directions, uint64_t n, float t_min,
                 int front_only, rt_hit* out);
int rt_intersect_device(rt_scene* scene, const float* d_origins, const float* d_directions, uint64_t n,
                        float t_min, int front_only, rt_hit* d_out, void* hip_stream);

/* rt_intersect on the BSP tree through the kernels that count work (as RT_FLAG_COUNT_WORK does for a render): same
 * hits, and stats receives box_tests, tris_tested, leaf_visits, inode_visits, box_exact_fallbacks and
 * box_filter_errors -- the number of decisions of the traversal's conservative shortcuts (slab filter, leaf normal
 * cones, leaf triangle bounds) that disagreed with the reference's exact evaluation on these lines; anything but 0 is
 * a bug.  A test hook: the counting kernels visit every node and triangle the reference does, and are slower. */
int rt_intersect_check(rt_scene* scene, const float* origins, const float* directions, uint64_t n, float t_min,
                       int front_only, rt_hit* out, rt_stats* stats);

/* render() (src/render.hpp:215-234): mean radiance per pixel, linear HDR, row-major, 3 floats per
 * pixel.  rt_render writes to host memory, rt_render_device to device memory (no PCIe copy). */
int rt_render(rt_scene* scene, const rt_render_params* params, float* out_rgb, rt_stats* stats);
int rt_render_device(rt_scene* scene, const rt_render_params* params, float* d_out_rgb, rt_stats* stats,
                     void* hip_stream);

/* render() on several GPUs from one process (the reference's render() is one call from one thread, src/render.hpp:215-234,
 * src/main.cpp:230; it has no multi-device form).  scenes[k] are handles of the same scene, normally one per GPU (two
 * handles on one GPU are allowed).  The rectangle's rows are cut into bands dealt round-robin to the handles; every handle
 * renders all of its bands with one call on its own GPU, all GPUs at once; the bands are copied GPU-to-GPU (peer copies
 * over xGMI) into the frame on scenes[0]'s GPU.  The image is the one a single rt_render of the rectangle gives, bit for
 * bit (the sampler is keyed by the pixel's position in the full image).  rt_render_multi copies the frame to host memory,
 * rt_render_multi_device leaves it at d_out_rgb, a device pointer on scenes[0]'s GPU.  params must describe a plain
 * rectangle (row_stride 0 or 1); stats are the sums over the handles, ms_total the whole call. */
int rt_render_multi(rt_scene* const* scenes, uint32_t scene_count, const rt_render_params* params, float* out_rgb,
                    rt_stats* stats);
int rt_render_multi_device(rt_scene* const* scenes, uint32_t scene_count, const rt_render_params* params, float* d_out_rgb,
                           rt_stats* stats);

/* rt_render_device for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and same image, bit
 * for bit, but the path state lives in a fixed array inside the library's code object (about 22 MiB per
 * GPU, held until the process ends) instead of buffers grown with hipMalloc, and the call allocates
 * nothing.  Limits: the rectangle holds at most RT_STATIC_MAX_PIXELS pixels (RT_ERR_LIMIT beyond that);
 * each batch traces at most that many paths, so large sample counts take more, smaller batches;
 * RT_FLAG_PROFILE_KERNELS is rejected.  With stats == NULL the call only enqueues work on hip_stream.
 * Every call on one GPU, from any scene handle, shares that GPU's array: calls that overlap on one GPU
 * race, so run them on one stream or separate them by a synchronisation. */
#define RT_STATIC_MAX_PIXELS 262144u
int rt_render_device_static(rt_scene* scene, const rt_render_params* params, float* d_out_rgb, rt_stats* stats,
                            void* hip_stream);

/* The steps between render() and the image file (src/main.cpp:233-240): reinhardToneMap, linearToSRGB,
 * nanToRed, infToGreen (src/image.hpp:15-70), medianFilter<1> (:73-109) and floatTo8BitUInt (:48-50).
 * hdr: width*height*3 floats as written by rt_render; rgb8: width*height*3 bytes, the payload of the
 * reference's P6 file.  d_srgb (width*height*3 floats, must not alias d_hdr) receives the frame before the
 * median; d_median (optional, same size) the frame after it.  Border pixels of the median use their
 * in-image neighbours only (the reference's border values depend on stale array contents). */
int rt_postprocess_device(const float* d_hdr, uint32_t width, uint32_t height, float* d_srgb, float* d_median,
                          uint8_t* d_rgb8, void* hip_stream);
int rt_postprocess(int device, const float* hdr, uint32_t width, uint32_t height, uint8_t* rgb8);

/* Test hook: evaluates one scalar building block of the kernels on the GPU for n inputs so that its
 * bits can be pinned: op 0 a/b, 1 sqrt(a), 2 sin,cos(2*pi*a) (2 outputs), 3 Philox words (2 outputs),
 * 4 fma(a,b,a), 5 a*b+a unfused.  Not part of the rendering interface. */
int rt_selftest_math(int op, const float* a, const float* b, uint32_t n, float* out);

/* Test hooks for the leaf triangle bounds (host only, no GPU; no counterpart in the reference, which scans every leaf
 * its line enters, src/bsp.hpp:106-117).  rt_debug_flatten returns the node and leaf bounds records exactly as
 * rt_scene_create uploads them: first the bounds records, 16 floats each (the record of node slot s at float offset
 * 16 * (node_slots - 1 - s): lo[3], hi[3], cone word, leaf index, cK, cE, cN and five words describing the normals:
 * sin b, cos b, A[3] when *leaf_axes is 1, ten 16-bit halves -- sin b2 and three axes -- when it is 3), then 8 floats
 * per node record (slot s at 16 * node_slots + 8 * s: lo[3], hi[3], index, aux).  With packed == NULL only the sizes
 * are returned.
 * rt_debug_leaf_drop evaluates, on the host and with the kernels' own arithmetic, whether the traversal would leave
 * the leaf at node slot slots[i] unscanned for line i (1) or scan it (0). */
int rt_debug_flatten(const rt_scene_desc* desc, float* packed, uint64_t* packed_floats, uint32_t* node_slots,
                     uint32_t* leaf_axes, float* box_abs_max);
int rt_debug_leaf_drop(const float* packed, uint32_t node_slots, uint32_t leaf_axes, float box_abs_max, const uint32_t* slots,
                       const float* origins, const float* directions, uint64_t n, uint8_t* out_drop);
/* The same for the leaf normal cones: would the traversal skip a leaf whose node record's aux word is cone_words[i] for
 * direction i, because the cone says that every triangle of the leaf fails the facing test (Surfaces::FRONT_ONLY,
 * src/geometry.hpp:129-131)? */
int rt_debug_cone_skip(const uint32_t* cone_words, const float* directions, uint64_t n, uint8_t* out_skip);

const char* rt_last_error(void);
/* Library/ABI version, bumped on any layout change of the structs above. */
uint32_t rt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
