// rt_device.h — flattened, device-resident scene layout shared by the host
// flattener (rt_flatten.cpp) and the kernels (rt_kernels.hip).
//
// The reference walks `Box<dyn Shape>` trees and re-derives an object-space
// ray, a 4x4 inverse and a normalised normal at every level
// (system.rs:106-110,190-199).  Here every primitive ("leaf") carries the
// composed world->leaf and leaf->world affine maps, so one ray visits each
// leaf with two 3x4 products, and interval end points are kept as
// (world t, tag) pairs; normals and uv are derived once, for the nearest hit.
#pragma once
#include <stdint.h>

// affine map, row-vector convention of matrix.rs: out[c] = sum_r p[r]*m[r*3+c] + m[9+c]
struct DevXform {
    double m[12];
};

enum DevLeafKind {
    DL_NONE = 0, // carries transforms only (composite / medium frames)
    DL_SPHERE = 1,
    DL_PLANE = 2,
    DL_RECT_XY = 3,
    DL_RECT_XZ = 4,
    DL_RECT_ZY = 5,
    DL_CUBE = 6,
    DL_CYLINDER = 7,
    DL_TORUS = 8,
};

// params:
//  sphere   : c.xyz, radius^2
//  plane    : origin.xyz, normal.xyz
//  rect     : origin.xyz, normal sign (+1/-1), lo0, hi0, lo1, hi1
//  cube     : 6 faces (min_x,max_x,min_y,max_y,min_z,max_z) x {coord, lo0, hi0, lo1, hi1}, then centre.xyz
//  cylinder : radius, height
//  torus    : ring radius, tube radius
#define DEV_LEAF_PARAMS 36
struct DevLeaf {
    int32_t kind;
    int32_t identity; // both maps are exactly the identity
    int32_t pad0, pad1;
    DevXform W; // world -> leaf
    DevXform O; // leaf -> world
    double p[DEV_LEAF_PARAMS];
};

enum DevObjectKind {
    DO_LEAF = 0,   // one primitive
    DO_CSG = 1,    // postfix program over leaves
    DO_MESH = 2,   // Composite of meshes
    DO_MEDIUM = 3, // HomogenousMedium around a boundary (leaf or program)
};

enum DevRootMode {
    DM_FIRST_POSITIVE = 0, // shapes/mod.rs:45-51 (sphere, cylinder, torus, CSG)
    DM_FIRST_ENTRY = 1,    // shapes/mod.rs:53-55 (cube) and plane/rect `intersect`
};

// CSG program words: op in the top byte, leaf index below
#define DEV_OP_LEAF 0
#define DEV_OP_UNION 1
#define DEV_OP_INTERSECTION 2
#define DEV_OP_DIFFERENCE 3

struct DevObject {
    int32_t kind;
    int32_t mode;
    int32_t leaf0;    // first leaf of this object
    int32_t n_leaves;
    int32_t prog0;    // first program word (DO_CSG, DO_MEDIUM with CSG boundary)
    int32_t prog_len; // 0 -> single leaf `leaf0`
    int32_t frame;    // DO_MESH / DO_MEDIUM: leaf index (DL_NONE) holding the node's own maps
    int32_t mesh0, n_meshes;
    int32_t material;
    int32_t needs_uv; // material's texture reads uv
    int32_t pad;
    double density;
};

// 16-byte kd-tree node.  Interior: a = (right child << 2) | axis, the left
// child is the next node.  Leaf: a = (first entry in tri index list << 2) | 3,
// b = triangle count.
struct DevKdNode {
    double split;
    uint32_t a;
    uint32_t b;
};

struct DevMesh {
    uint32_t tri0, n_tris;   // into tri data (72 B records: v0, e1, e2) and tri_n
    uint32_t normal0;        // into normals
    uint32_t kd0;            // root node
    uint32_t kd_index0;      // base of this mesh's triangle index list
    uint32_t smooth;
    double qmin[3], qmax[3]; // Mesh::new bounds, which include the origin (mesh.rs:32-42)
    double bmin[3], bmax[3]; // tight bounds for the kd-tree traversal
};

struct DevMaterial {
    int32_t kind;
    int32_t texture;
    double fuzz, ior, intensity;
};

struct DevTexture {
    int32_t kind;
    int32_t width, height, pad;
    uint64_t offset;
    double c1[3], c2[3], scale;
};

struct DevScene {
    const DevObject *objects;
    const DevLeaf *leaves;
    const int32_t *prog;
    const DevMesh *meshes;
    const DevKdNode *kd;
    const uint32_t *kd_index;
    const double *tris;    // 9 per triangle
    const double *leaf_tris; // 10 per kd leaf reference, in leaf order: the 9 above + the triangle index (low word)
    const uint32_t *tri_n; // 3 per triangle (normal indices, mesh-relative)
    const double *normals; // xyz
    const DevMaterial *materials;
    const DevTexture *textures;
    const uint8_t *image_data;
    int32_t n_objects;
    int32_t pad;
    double background[3];
    double cam_fov_factor;
    double cam_c2w[16];
};

#define DEV_MAX_LEAVES_PER_OBJECT 256
#define DEV_MAX_INTERVALS 6   // per list
#define DEV_CSG_STACK 4       // lists alive at once
#define DEV_KD_STACK 20       // kd-tree depth limit = LDS stack entries per lane
