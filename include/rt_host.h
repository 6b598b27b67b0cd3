/* rt_host.h -- host-side (CPU) helpers of librt_amd.so: the scene preprocessing steps that the
 * reference runs before render() (src/main.cpp:191-216), plus a fast BSP builder that produces the
 * same tree as BSPTree's constructor.  None of these touch the GPU.  Index types are the
 * reference's (src/index_types.hpp:8-14) widened to 32 bits.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* preprocessMaterial (src/material.hpp:32-45).  material: {colour[3], roughness, metalness,
 * emission[3]}; out: 12 floats in the rt_scene_desc::model_materials layout. */
void rt_preprocess_material(const float material[8], float out[12]);

/* glm::quat(eulerAngles) as used by src/main.cpp:136,146-147 (libraries/glm/glm/detail/type_quat.inl:204-213).
 * quat = {w,x,y,z}. */
void rt_quat_from_euler(const float euler[3], float quat[4]);

/* Camera::forward/down/right (src/camera.hpp:18-28). */
void rt_camera_basis(const float quat[4], float forward[3], float down[3], float right[3]);

/* pixelToRayTransform (src/camera.hpp:33-46); out = column-major mat3 {right, down, o}. */
void rt_pixel_to_ray_transform(const float forward[3], const float down[3], const float right[3], float fov,
                               uint32_t width, uint32_t height, float out[9]);

/* instantiateMeshes (src/mesh.hpp:90-138).  transforms: 10 floats per model {position[3], quat
 * w,x,y,z, scale[3]} (MeshTransform, src/mesh.hpp:57-60).  Outputs sized by the sum of the models'
 * mesh vertex counts; out_model_vertex_begin has model_count+1 entries. */
void rt_instantiate_meshes(const float* positions, const float* normals, const uint32_t* mesh_vertex_ranges,
                           const float* transforms, const uint32_t* model_meshes, uint32_t model_count,
                           float* out_positions, float* out_normals, uint32_t* out_model_vertex_begin);

/* preprocessTris (src/mesh.hpp:149-188) + preprocessTri (src/geometry.hpp:77-82).  out_pretris: 12
 * floats per triangle {normal, v1, v1ToV3, v1ToV2}; out_model_pretri_begin has model_count+1 entries. */
void rt_preprocess_tris(const float* inst_positions, const uint32_t* model_vertex_begin, const uint32_t* base_tris,
                        const uint32_t* mesh_tri_ranges, const uint32_t* model_meshes, uint32_t model_count,
                        float* out_pretris, uint32_t* out_model_pretri_begin);

/* The same two steps on GPU `device`, one lane per instantiated vertex / triangle (host pointers in and out;
 * mesh_count = number of {begin,size} pairs in the range arrays).  Bit-identical outputs to the functions
 * above.  Return 0 or a negative rt_status (message in rt_last_error). */
int rt_instantiate_meshes_device(int device, const float* positions, const float* normals, const uint32_t* mesh_vertex_ranges,
                                 uint32_t mesh_count, const float* transforms, const uint32_t* model_meshes,
                                 uint32_t model_count, float* out_positions, float* out_normals,
                                 uint32_t* out_model_vertex_begin);
int rt_preprocess_tris_device(int device, const float* inst_positions, const uint32_t* model_vertex_begin,
                              const uint32_t* base_tris, const uint32_t* mesh_tri_ranges, uint32_t mesh_count,
                              const uint32_t* model_meshes, uint32_t model_count, float* out_pretris,
                              uint32_t* out_model_pretri_begin);

/* computeBoundingBox (src/geometry.hpp:85-92) followed by the inflation of src/main.cpp:207-208
 * (inflate_mode 0: min,max *= 1.001) or an additive pad of 0.001*max extent (inflate_mode 1, for
 * scenes that do not straddle the origin). */
void rt_scene_bounding_box(const float* inst_positions, uint64_t vertex_count, int inflate_mode, float out_box[6]);

/* srgbToLinear / linearToSRGB for one channel (src/image.hpp:20-41); used by the built-in scene's
 * material colours (src/main.cpp:172). */
float rt_srgb_to_linear(float srgb);
float rt_linear_to_srgb(float linear);

/* The image file of src/main.cpp:272-280: binary PPM, "P6\n<width> <height>\n255\n" followed by
 * width*height*3 bytes.  Returns 0, or -1 if the file cannot be written. */
int rt_write_ppm(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);

/* One entry of the Meshes initializer list (src/scene.cpp:8-30: positions, normals, indexed triangles
 * with one index per vertex) read from a Wavefront OBJ or a PLY file (ascii or binary_little_endian).
 * The reference has no loader (its meshes are the literals of src/main.cpp:25-121).  Polygons become
 * fans; OBJ position/normal index pairs become single vertices; missing normals are replaced by
 * area-weighted vertex normals.  Returns 0 or -1 (message in rt_mesh_last_error). */
typedef struct rt_mesh_file rt_mesh_file;
int rt_mesh_load(const char* path, rt_mesh_file** out);
void rt_mesh_counts(const rt_mesh_file* mesh, uint32_t* vertex_count, uint32_t* tri_count);
/* positions, normals: 3 floats per vertex; tris: 3 indices per triangle.  Valid until rt_mesh_destroy. */
void rt_mesh_view(const rt_mesh_file* mesh, const float** positions, const float** normals, const uint32_t** tris);
void rt_mesh_destroy(rt_mesh_file* mesh);
const char* rt_mesh_last_error(void);

/* lineIntersectsBox (src/geometry.hpp:135-214) and triIntersectsBox (src/geometry.hpp:217-428). */
int rt_line_intersects_box(const float origin[3], const float direction[3], const float box[6]);
int rt_tri_intersects_box(const float tri[9], const float box[6]);

/* BSPTree constructor (src/bsp.hpp:35-48, 174-285).  Produces the same nodes and leaves in the same
 * order as the reference (interior nodes in pre-order, leaves in creation order, leaf triangles in
 * ascending (model, triangle) order) but visits only each node's candidate triangles instead of
 * rescanning the whole scene.  Where the reference would recurse without bound (more than
 * max_leaf_tris triangles meeting in an arbitrarily small box) the builder stops at max_depth and
 * emits an oversized leaf.  threads > 1 builds subtrees in parallel and restores the sequential order
 * afterwards (same arrays, bit for bit); threads <= 0 picks OMP_NUM_THREADS or up to 16 hardware threads. */
typedef struct rt_bsp rt_bsp;
int rt_bsp_build(const float* inst_positions, const uint32_t* model_vertex_begin, const uint32_t* base_tris,
                 const uint32_t* mesh_tri_ranges, const uint32_t* model_meshes, uint32_t model_count,
                 const float root_box[6], uint32_t max_leaf_tris, uint32_t max_depth, int threads, rt_bsp** out);
/* The same tree, array for array, with the per-level work on GPU `device` (the exact triangle/box predicate on every
 * (node, candidate) pair of a level, prefix sums, order-preserving scatter into leaf lists and the next level's
 * candidates); the host decides leaf / split / empty per node and numbers the nodes depth-first afterwards.  No
 * counterpart in the reference, whose constructor (src/bsp.hpp:35-48) is sequential.  At most 2^31 candidate pairs
 * per level (RT_ERR_LIMIT beyond). */
int rt_bsp_build_device(const float* inst_positions, const uint32_t* model_vertex_begin, const uint32_t* base_tris,
                        const uint32_t* mesh_tri_ranges, const uint32_t* model_meshes, uint32_t model_count,
                        const float root_box[6], uint32_t max_leaf_tris, uint32_t max_depth, int device, rt_bsp** out);
/* For every box of boxes6 (min xyz, max xyz), the number of triangles of the WHOLE scene that pass the reference's
 * triangle/box predicate (src/geometry.hpp:217-428) on it: what the reference's constructor computes at every node
 * (src/bsp.hpp:188-211), evaluated on GPU `device` without any candidate list.  A check on the builders for scenes far
 * beyond the reference's own reach: a leaf's box must give the leaf's size, an empty child's box 0, a divided node's
 * box more than max_leaf_tris. */
int rt_bsp_exhaustive_counts_device(const float* inst_positions, const uint32_t* model_vertex_begin, const uint32_t* base_tris,
                                    const uint32_t* mesh_tri_ranges, const uint32_t* model_meshes, uint32_t model_count,
                                    const float* boxes6, uint32_t box_count, int device, uint32_t* counts);
void rt_bsp_destroy(rt_bsp* bsp);
/* Sizes: inode_count, leaf_count, ref_count, depth, oversized_leaves. */
void rt_bsp_counts(const rt_bsp* bsp, uint32_t counts[5]);
/* Views into the builder's arrays, valid until rt_bsp_destroy; ready to drop into rt_scene_desc. */
void rt_bsp_view(const rt_bsp* bsp, float root_box[6], int32_t* root_index, const rt_inode** inodes,
                 const uint32_t** leaf_first_ref, const rt_tri_ref** leaf_refs);

/* Builder of the alternative accelerator (rt_bvh4_node, rt_amd.h): binned surface-area-heuristic splits
 * down to leaves of at most max_leaf_tris (1..8) triangles, two binary levels folded into each 4-wide
 * node, child boxes padded by 2^-18 of the largest coordinate and then quantised outwards to 8 bits per plane.  pretris: 12 floats per triangle as in
 * rt_scene_desc.  The result does not depend on `threads` (<= 0: OMP_NUM_THREADS or up to 16 hardware
 * threads).  No counterpart in the reference. */
typedef struct rt_bvh4 rt_bvh4;
int rt_bvh4_build(const float* pretris, uint32_t pretri_count, uint32_t max_leaf_tris, int threads, rt_bvh4** out);
void rt_bvh4_destroy(rt_bvh4* bvh);
/* node_count, triangle_count, depth (levels of 4-wide nodes), stack_need (largest number of deferred
 * children on any descent), leaf_count, largest leaf. */
void rt_bvh4_counts(const rt_bvh4* bvh, uint32_t counts[6]);
/* Views valid until rt_bvh4_destroy; tri_order[k] = index into pretris of the hierarchy's k-th triangle. */
void rt_bvh4_view(const rt_bvh4* bvh, const rt_bvh4_node** nodes, const uint32_t** tri_order);

#ifdef __cplusplus
}
#endif
#endif
