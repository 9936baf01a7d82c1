// rt_flatten.h — host-side result of flattening an rt_scene_desc for the device.
#pragma once
#include "../../include/rt_api.h"
#include "rt_device.h"

#include <string>
#include <vector>

namespace rt_host {

struct FlatScene {
    std::vector<DevObject> objects;
    std::vector<DevLeaf> leaves;
    std::vector<int32_t> prog;
    std::vector<DevMesh> meshes;
    std::vector<uint32_t> grid_bits;
    std::vector<uint32_t> grid_dist;
    std::vector<uint32_t> cell_start;
    std::vector<double> tris;
    std::vector<double> cell_tris;
    std::vector<float> cell_pairs; // 20 per two entries of cell_tris: the f32 pre-test records of k_mesh
    std::vector<uint32_t> tri_n;
    std::vector<double> tri_nrm;
    std::vector<DevMeshJob> jobs;
    std::vector<DevKdNode> kd;
    std::vector<uint32_t> kd_index;
    std::vector<double> leaf_tris;
    std::vector<double> grp_box, grp_tris;
    std::vector<uint32_t> grp_idx;
    std::vector<double> normals;
    std::vector<DevMaterial> materials;
    std::vector<DevTexture> textures;
    std::vector<uint8_t> image_data;
    double background[3];
    double cam_fov_factor;
    double cam_c2w[16];
    bool uses_csg = false;
    int max_stack = 0, max_intervals = 0; // over all CSG programs / fog boundaries: lists alive at once, intervals in a list (bounds)
    double grid_crowd = 0.0; // worst mesh: triangles in the cell of an average (cell, triangle) reference
};

// returns RT_OK or a negative rt_status with `err` filled
int flatten_scene(const rt_scene_desc *desc, FlatScene &out, std::string &err);

} // namespace rt_host
