/* Prints, as JSON, what the C compiler makes of every struct in include/rt_api.h: size, and offset and size of
 * each field.  tests/test_host_logic.py compares this with raytracer_amd/_capi.py (ctypes), with the #[repr(C)]
 * declarations in ffi/src/lib.rs and with the table in INTEGRATION.md.  Test infrastructure only. */
#include <stddef.h>
#include <stdio.h>

#include "../../include/rt_api.h"

#define S(T) printf("%s\n \"%s\": {\"size\": %zu, \"fields\": {", first_s ? "" : ",", #T, sizeof(T)), first_s = 0, first_f = 1
#define F(T, M) printf("%s\"%s\": [%zu, %zu]", first_f ? "" : ", ", #M, offsetof(T, M), sizeof(((T *)0)->M)), first_f = 0
#define E() printf("}}")

int main(void)
{
    int first_s = 1, first_f = 1;
    printf("{");
    S(rt_shape_node); F(rt_shape_node, kind); F(rt_shape_node, child_a); F(rt_shape_node, child_b); F(rt_shape_node, flags);
    F(rt_shape_node, params); F(rt_shape_node, o2w); F(rt_shape_node, w2o); E();
    S(rt_material); F(rt_material, kind); F(rt_material, texture); F(rt_material, fuzz); F(rt_material, ior); F(rt_material, intensity); E();
    S(rt_texture); F(rt_texture, kind); F(rt_texture, image); F(rt_texture, c1); F(rt_texture, c2); F(rt_texture, scale); E();
    S(rt_image); F(rt_image, width); F(rt_image, height); F(rt_image, offset); E();
    S(rt_mesh); F(rt_mesh, first_vertex); F(rt_mesh, n_vertices); F(rt_mesh, first_normal); F(rt_mesh, n_normals);
    F(rt_mesh, first_triangle); F(rt_mesh, n_triangles); E();
    S(rt_object); F(rt_object, shape); F(rt_object, material); E();
    S(rt_camera); F(rt_camera, fov_factor); F(rt_camera, camera_to_world); E();
    S(rt_scene_desc); F(rt_scene_desc, camera); F(rt_scene_desc, background);
    F(rt_scene_desc, n_objects); F(rt_scene_desc, objects); F(rt_scene_desc, n_nodes); F(rt_scene_desc, nodes);
    F(rt_scene_desc, n_child_list); F(rt_scene_desc, child_list); F(rt_scene_desc, n_materials); F(rt_scene_desc, materials);
    F(rt_scene_desc, n_textures); F(rt_scene_desc, textures); F(rt_scene_desc, n_images); F(rt_scene_desc, images);
    F(rt_scene_desc, n_image_bytes); F(rt_scene_desc, image_data); F(rt_scene_desc, n_meshes); F(rt_scene_desc, meshes);
    F(rt_scene_desc, n_vertices); F(rt_scene_desc, vertices); F(rt_scene_desc, n_normals); F(rt_scene_desc, normals);
    F(rt_scene_desc, n_triangles); F(rt_scene_desc, tri_v); F(rt_scene_desc, tri_n); E();
    S(rt_options); F(rt_options, width); F(rt_options, height); F(rt_options, samples); F(rt_options, max_depth);
    F(rt_options, bias); F(rt_options, seed); F(rt_options, tile_size); F(rt_options, shard_rank); F(rt_options, shard_count);
    F(rt_options, flags); E();
    S(rt_stats); F(rt_stats, rays); F(rt_stats, primary_rays); F(rt_stats, strata); F(rt_stats, bounce_passes); F(rt_stats, guards);
    F(rt_stats, interval_overflow); F(rt_stats, ms_device); F(rt_stats, ms_raygen); F(rt_stats, ms_intersect); F(rt_stats, ms_shade);
    F(rt_stats, ms_mesh); F(rt_stats, ms_tail); F(rt_stats, n_intersect_launches); F(rt_stats, n_mesh_launches);
    F(rt_stats, mesh_rays); F(rt_stats, intersect_rays); F(rt_stats, mesh_records); F(rt_stats, mesh_one_rays); E();
    S(rt_hit); F(rt_hit, t); F(rt_hit, n); F(rt_hit, uv); F(rt_hit, object); F(rt_hit, pad); E();
    S(rt_progress); F(rt_progress, on_stratum); F(rt_progress, on_snapshot); F(rt_progress, min_interval_ms);
    F(rt_progress, reserved); F(rt_progress, user); E();
    printf("\n}\n");
    return 0;
}
