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
    int32_t skip_ord; // DL_NONE leaf of a DEV_PROG_SKIP word: 1-based ordinal of that node in its object, else 0
    int32_t pad1;
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
// A union or difference whose own transform rotates or scales: when one operand is missed the reference
// returns the other's intervals without this node's to_world (csg.rs:33-38,216-221).  The word then carries
// this flag and, in its low bits, the object-relative index of a DL_NONE leaf with W = world -> node,
// p[0..8] = linear part of world -> parent, O = (world -> parent) * (node -> world).  csg_eval scales the
// surviving end points by |d W_node| / |d W_parent| and sets tag bit 12 + skip_ord - 1; the nearest-hit code
// undoes the scale to find the surface point and takes the normal through O.
#define DEV_PROG_SKIP (1 << 23)
#define DEV_MAX_SKIPS 20

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
    int32_t has_bounds; // bit 0: wmin/wmax hold world bounds of the primitives a ray must meet for the object to yield
                        // anything (DO_CSG, DO_MEDIUM, a torus leaf): a ray that misses them skips the object without
                        // running its program; bit 1: no hit of the object lies before those bounds either
    double density;
    double wmin[3], wmax[3];
};

// 16-byte kd-tree node (the alternative to the grid, see k_mesh).  Interior: a = (right child << 2) | axis,
// the left child is the next node.  Leaf: a = (first entry in the leaf-ordered records << 2) | 3,
// b = triangle count.
struct DevKdNode {
    double split;
    uint32_t a;
    uint32_t b;
};

// One uniform grid per mesh.  The reference tests every triangle of a mesh
// (mesh.rs:54-64); the grid only decides which triangles a ray visits.
//  - occupancy words, one per 32 cells, x fastest: (bits, rank of the word's first
//    cell among the mesh's occupied cells).  The k_mesh workgroups keep them in
//    LDS, so stepping through empty space never leaves the CU;
//  - distance nibbles, one per cell: Chebyshev distance in cells to the nearest
//    occupied cell, capped at 15.  From a cell at distance k a ray may move k-1
//    cells along any axis without meeting a triangle, so open space costs a few
//    jumps instead of one step per cell.  Also in LDS;
//  - cell_start[rank] .. cell_start[rank+1]: the occupied cell's run of 80-byte
//    triangle records in `cell_tris` (v0, e1, e2, triangle index), ascending
//    index.  Also kept in LDS while all meshes' entries fit DEV_GRID_MAX_STARTS.
struct DevMesh {
    uint32_t tri0, n_tris;   // into tri data (72 B records: v0, e1, e2) and tri_n
    uint32_t normal0;        // into normals
    uint32_t smooth;
    uint32_t n[3];           // cells along x, y, z (each <= 255)
    uint32_t bits0;          // first (bits, rank) pair of this mesh in grid_bits
    uint32_t cell0;          // first entry of this mesh's cell_start run (occupied cells + 1 entries)
    uint32_t dist0;          // first word of this mesh's distance nibbles in grid_dist
    double qmin[3], qmax[3]; // Mesh::new bounds, which include the origin (mesh.rs:32-42)
    double bmin[3], bmax[3]; // grid bounds: the tight bounds of the triangles, padded
    double cell[3], inv_cell[3];
    // groups of up to 64 neighbouring triangles for the wave-per-ray kernels (k_mesh_brute, k_tail)
    uint32_t grp0, n_groups;
    uint32_t kd0;       // root node of this mesh's kd-tree (node indices inside it are mesh-relative)
    uint32_t kd_index0; // base of this mesh's leaf-ordered triangle records
    uint32_t use_kd;    // this mesh is walked through its kd-tree (k_mesh<KD = true>), not its grid
    float crowd;        // triangles in the cell of an average (cell, triangle) reference of its grid
    // Up to DEV_MESH_PARTS boxes (padded like bmin / bmax) that together hold every triangle, chosen by the flattener
    // to have the least total surface: a ray that enters the mesh's bounds but none of these meets no triangle, so
    // k_intersect does not queue it for the walk (LinkedTorus.obj: two thin rings in a box that is mostly empty -
    // 45 % of the camera rays that cross the bounds miss all four parts).  n_parts 0: no such boxes (the bounds are
    // as good).
    uint32_t n_parts, pad_parts;
    double part[4][6]; // min xyz, max xyz
    // the same boxes in world space (the bounds of their eight corners under the object's frame, padded again): what
    // k_intersect tests, with the world ray as it is
    double wbox[6];
    double wpart[4][6];
};
#define DEV_MESH_PARTS 4

// One (mesh object, mesh) pair, in the order Ray::trace / Composite::intersect would visit them:
// everything k_mesh needs to start the walk of one mesh and to finish a hit, in one record (the
// walking waves read it from LDS instead of chasing objects -> leaves -> meshes through memory).
struct DevMeshJob {
    int32_t object;   // index in the object list
    int32_t mesh;     // index within the object (Composite child order)
    int32_t identity; // the object's frame maps are the identity
    uint32_t smooth;
    uint32_t n[3];
    uint32_t bits0, cell0, dist0;
    uint32_t tri0;    // into tris / tri_nrm
    uint32_t kd0, kd_index0;
    uint32_t use_kd;  // walked by k_mesh<KD = true>; the other instantiation passes this job by
    DevXform W, O;    // the object's frame: world -> object, object -> world
    double qmin[3], qmax[3];
    double bmin[3], bmax[3];
    double inv_cell[3];
    double cell[3];
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
    const uint32_t *grid_bits;  // (bits, rank) pairs of every mesh grid, concatenated
    const uint32_t *grid_dist;  // distance nibbles of every mesh grid, 8 cells per word
    const uint32_t *cell_start; // per mesh: occupied cells + 1 offsets into cell_tris (absolute record indices)
    const double *tris;      // 9 per triangle
    const double *cell_tris; // 10 per (cell, triangle) reference, in cell order: the 9 above + the triangle index (low word);
                             // runs padded to an even length (index ~0u)
    const float *cell_pairs; // 20 per two entries of cell_tris: v0 - cell centre, e1, e2, error bound, each as (A, B) in f32
    const uint32_t *tri_n; // 3 per triangle (normal indices, mesh-relative)
    const double *tri_nrm; // 9 per triangle: its three vertex normals
    const DevKdNode *kd;     // kd-trees of all meshes
    const double *leaf_tris; // 10 per kd leaf reference, in leaf order: v0, e1, e2, triangle index (low word)
    const double *grp_box;   // 6 per group: padded bounds (min xyz, max xyz)
    const double *grp_tris;  // 64 slots per group x 9: the triangle records in group order
    const uint32_t *grp_idx; // 64 slots per group: the triangle's index in its mesh, ~0u for an empty slot
    const DevMeshJob *jobs;
    const double *normals; // xyz
    const DevMaterial *materials;
    const DevTexture *textures;
    const uint8_t *image_data;
    int32_t n_objects;
    uint32_t n_grid_words; // (bits, rank) pairs in grid_bits
    uint32_t n_cell_starts; // entries in cell_start
    uint32_t n_dist_words;  // words in grid_dist
    uint32_t n_jobs;
    uint32_t n_leaves, n_meshes, n_materials, n_textures;
    uint32_t starts_in_lds; // k_mesh keeps cell_start in LDS too (they fit beside the other grid tables)
    uint32_t lds_list_cap, lds_list_n; // interval lists in LDS (k_intersect<.., LL = true>): intervals per list, lists per ray; 0: none
    double background[3];
    double cam_fov_factor;
    double cam_c2w[16];
};

#define DEV_MAX_LEAVES_PER_OBJECT 256
#define DEV_MAX_INTERVALS 12  // per list (beyond: counted in rt_stats.interval_overflow, the object is a miss for that ray)
#define DEV_CSG_STACK 8       // lists alive at once: right-nested operations need one each, checked at rt_scene_create
#define DEV_LDS_LIST_MAX_INTERVALS 15 // limits of the interval lists kept in LDS (LdsLists, rt_dev_shapes.h: four bits per count,
#define DEV_LDS_LIST_MAX_LISTS 8      // a 32-bit word of counts); what fits is decided per scene from the flattener's bounds
#define DEV_GRID_MAX_WORDS 2048 // occupancy words of all meshes together: 16 KB of LDS with their ranks, 32 KB of distances
#define DEV_GRID_MAX_STARTS 6144 // cell_start entries kept in LDS (24 KB); beyond that they are read from memory
#define DEV_GRID_MAX_AXIS 255
#define DEV_KD_STACK 20       // kd-tree depth limit = LDS stack entries per lane
#define DEV_MESH_JOBS_LDS 16 // job records kept in LDS (beyond that they are read from memory)
