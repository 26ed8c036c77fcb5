// rt_internal.h -- shared between the host side (scene_build.cpp, api.cpp) and
// the HIP kernels (kernels.hip) of librt_hip.so.  Not part of the ABI.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include "../../include/rt_hip.h"

// ---------------------------------------------------------------------------
// Device data layout (all arrays live in HBM, uploaded once per scene)
//
// KD nodes of ALL objects are stored in one array in the order the reference's
// traversal pops them: pre-order with the RIGHT child first (hw4.cpp:831-836
// pushes left then right, so right is popped first), objects in file order
// (hw4.cpp:808).  Every record carries the index to continue at when its
// subtree is skipped, so a wave walks the array front to back without a stack.
struct DevNode {            // 48 B, fetched wave-uniformly (scalar path)
    float lo[3];            // center - extend/2.0f  (hw4.cpp:287, precomputed with the same float ops)
    float hi[3];            // center + extend/2.0f
    int32_t skip;           // next node when this subtree is not entered (== index+1 for a leaf)
    int32_t meta;           // bits 0..7 depth, bit 8 leaf, bits 12..13 split axis (inner nodes)
    float margin;           // relative slack for distance culling of this subtree (fast traversal)
    int32_t leaf_first;     // leaf: first record in leaf_tris (== reference visit rank of its first triangle)
    int32_t leaf_count;
    int32_t group_first;    // leaf: first DevLeafGroup; the leaf's records are cut into ceil(leaf_count / 8) groups
};
static_assert(sizeof(DevNode) == 48, "DevNode must be 48 bytes");

struct DevLeafRange { int32_t first; int32_t count; };

// One triangle reference of a leaf, stored in leaf order (a triangle that is
// in several leaves is stored several times), so a leaf is one contiguous run.
struct DevLeafTri {         // 48 B
    float v0[3];            // vertices[0]
    float e1[3];            // vertices[1]-vertices[0]   (hw4.cpp:177, same float op precomputed)
    float e2[3];            // vertices[2]-vertices[0]   (hw4.cpp:178)
    int32_t prim;           // scene-file index
    int32_t flags;          // bit0: material refract > 0.1 (final.cpp:999)
    int32_t rank;           // position of this reference in the reference's visit order (tie-break key)
};
static_assert(sizeof(DevLeafTri) == 48, "DevLeafTri must be 48 bytes");

// Inside a leaf the records are sorted along a Morton curve and cut into groups of <= RT_GROUP consecutive
// records; a group carries the bounding box of its triangles.  The fast traversal skips a group whose (inflated)
// box a ray misses or that lies beyond its best hit -- a culling layer under the reference's tree.
#define RT_GROUP 6
struct DevLeafGroup {       // 64 B
    float lo[3]; float hi[3];
    float gm;               // max |edge1|*|edge2| over the group's triangles
    float area2_min;        // min |edge1 x edge2| over its non-degenerate triangles
    float axis[3];          // unit mean normal; every triangle normal is within the cone (cos_t, sin_t) around +-axis
    float cos_t, sin_t;
    float pad[3];
};
static_assert(sizeof(DevLeafGroup) == 64, "DevLeafGroup must be 64 bytes");

#define RT_RAY_COUNTERS 64
#define RT_PF_SKIP_SHADOW 1   // refract > 0.1
#define RT_PF_SPHERE      2

// The build's own hierarchy over the tree triangles (bvh_build.cpp): a binary BVH in pre-order with skip
// pointers, eight layouts like the reference nodes.  A node carries what the group test of 4.1b needs to inflate its
// box soundly: the largest |e1||e2| and the smallest |e1 x e2| of its triangles, and (small subtrees) their normal cone.
struct DevBvhNode {         // 64 B
    float b[6];             // bounding box of the subtree's triangles as {near_x, far_x, near_y, far_y, near_z, far_z} in the
                            // MIRRORED frame of the node's octant layout o: an axis whose bit is set in o (rays travel towards
                            // -axis) is negated, so near = -hi, far = -lo there and near <= far on every axis.  A ray of that
                            // octant, mirrored the same way, has only positive direction components: its slab entry is the
                            // largest `near` parameter and its exit the smallest `far` one, no per-axis min / max; the pairs
                            // (near, far) feed packed f32 instructions.  Layout 0 is the unmirrored box (lo, hi interleaved).
    float slack;            // box inflation per unit of reach for a unit-length ray when nothing bounds |det| but the
                            // reference's cut-off (trace_device.h: lateral_slack with |dir| <= 1.001), + 2e-5 for the slabs
    float cone_k;           // with the cone: the inflation per unit reach is cone_k / cos(angle + cone), if smaller
    float axis[3];          // normal cone: axis, cos, sin of the half angle (rounded outwards); valid if meta & 0x200
    float cos_t, sin_t;
    int32_t skip;           // inner node: next node when the subtree is not entered; leaf (next = index + 1): first record in bvh_tris
    int32_t meta;           // depth | leaf << 8 | cone << 9 | (inner: split axis) << 12 | (leaf: triangle count) << 16
    float rho;              // |orig - root centre| + rho >= |orig - v0| + 0.44 (|e1| + |e2|) for every triangle below
};
static_assert(sizeof(DevBvhNode) == 64, "DevBvhNode must be 64 bytes");
// The same nodes (layout 0) in 32 bytes, for the per-lane walks (trace_device.h: bvh_lane_next_leaf): a lane's step is two
// 16-byte loads of half a line instead of four of a whole one.  No cone: the inflation is slack * d0 + srho with both
// factors rounded UP to 14 bits (8 exponent, 6 mantissa), srho >= slack * rho -- never less than the 64-byte node gives.
struct DevBvhNode32 {
    float b[6];             // {lo.x, hi.x, lo.y, hi.y, lo.z, hi.z}
    uint32_t link;          // inner: skip; leaf: 0x80000000 | first record in bvh_tris
    uint32_t sq;            // slack14 << 18 | srho14 << 4 | triangles of a leaf (<= 7) << 1
};
static_assert(sizeof(DevBvhNode32) == 32, "DevBvhNode32 must be 32 bytes");
struct DevRefLeaf { int32_t leaf; int32_t rank; };   // a reference leaf (index in layout 7) holding a triangle, and the
                                                     // reference's visit rank of that record
// reachable_rank's first probe for BVH triangle t in ONE record (made on the device by k_tri_reach from bvh_tri_ref,
// bvh_ref_leaf and ref_chain_box): the chain box of the first reference leaf holding t and that record's visit rank.
// lo[0] = NaN: no chain box (or no leaf): the caller takes reachable_rank itself.
struct DevTriReach { float lo[3]; float hi[3]; int32_t rank; int32_t pad; };
static_assert(sizeof(DevTriReach) == 32, "DevTriReach must be 32 bytes");

struct DevLoose {           // 48 B: a primitive tested one by one after the trees
    float a[9];             // triangle: v0,e1,e2   sphere: center(3), radius, unused
    int32_t prim;
    int32_t flags;          // RT_PF_*
    int32_t pad;
};
static_assert(sizeof(DevLoose) == 48, "DevLoose must be 48 bytes");

struct DevMaterial {        // 48 B
    float surf[3];
    float Ka, Kd, Ks, specularity, reflect, refract, Nr;
    float pad[2];
};

struct DevLight {           // 64 B
    float pos[3]; float intensity;
    float color[3]; int32_t type;
    float dir1[3]; float size1;
    float dir2[3]; float size2;
};

// Photon map in HBM: two arrays in the reference's 1-based heap order (slot 0 unused), back to back in one
// allocation.  A tree visit reads only DevPhotonPos (16 B); the power array is touched once per found photon.
// 10 M photons: 160 MB of positions (fits the 256 MB Infinity Cache) + 160 MB of powers.
struct DevPhotonPos { float pos[3]; uint32_t packed; };    // packed: plane (bits 0..1) | RT_PH_UNREACHABLE | theta<<8 | phi<<16
struct DevPhotonPow { float power[3]; uint32_t pad; };     // pad: second heap only, the photon's slot in the reference's heap
static_assert(sizeof(DevPhotonPos) == 16 && sizeof(DevPhotonPow) == 16, "photon records must be 16 bytes");
// base pointer of the allocation: [n+1] DevPhotonPos, [n+1] DevPhotonPow in the REFERENCE's heap order, then the same
// photons again in the order of rt_photon_product_order: [n+2] DevPhotonPos (one padding record: both children are
// fetched together), [n+1] DevPhotonPow, and [2 (n/2 + 2)] box corners: records 2i, 2i+1 = lo, hi of the bounding box of
// the sub-heap of node i, for the n/2 nodes that have a child
// ... and, 64-byte aligned behind the box corners, [n+1] 64-byte node records for the per-lane walk (kernels_photon_wave.hip):
// {pos[3], packed, box of the left child's sub-heap lo[3] hi[3], box of the right child's}; a child without children
// contributes its position as a degenerate box, and its RT_PH_UNREACHABLE flag moves to bit 3 (left) / 4 (right) of packed.
// Offsets in 16-byte records from the base of the allocation:
#ifdef __HIPCC__
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif
RT_HD static inline size_t rt_photon_nodes64_offset(int64_t n) { return (4 * ((size_t)n + 1) + 1 + 2 * ((size_t)n / 2 + 2) + 3) & ~(size_t)3; }
RT_HD static inline size_t rt_photon_image_records(int64_t n) { return rt_photon_nodes64_offset(n) + 4 * ((size_t)n + 1); }
typedef DevPhotonPos DevPhoton;
#define RT_PH_UNREACHABLE 4u      // second heap: the reference's LocatePhoton never reaches this photon (Q8)

struct DevSceneView {       // passed by value to kernels
    const DevNode* nodes; int32_t nnodes;      // reference order (= layout 7: right child first everywhere)
    const DevNode* nodes_oct[8];               // layout o: at a node split on axis a the child on the side the
                                               // ray comes from is first (bit a of o set = direction negative)
    const DevLeafTri* leaf_tris;
    const DevLeafGroup* leaf_groups;
    const DevLoose* loose; int32_t nloose;
    // the build's own hierarchy; nbvh = 0: none
    const DevBvhNode* bvh_oct[8]; int32_t nbvh;
    const DevBvhNode32* bvh_c32;                       // compact nodes of layout 0 (NULL: leaves of more than 7 triangles)
    float bvh_root_centre[3];           // what DevBvhNode::rho is measured from (the root box's centre, rounded)
    float bvh_root_rho;                 // rho of the root node
    unsigned long long* ray_counter;    // next ray of a k_trace_*_bvh_lanes launch.  The scene owns a ring of RT_RAY_COUNTERS such
                                        // words, 64 B apart; every launch takes the next one and zeroes it on its stream, so two
                                        // launches in flight (other streams, or back to back) never claim from the same word
    int32_t lane_waves;                 // waves of a refilling per-lane kernel (0 = the default, kernels_trace.hip: lanes_grid)
    int32_t lane_layout_mask;           // per-lane walks use layout (octant & mask): 0 = one layout for all (default), 7 = the ray's own octant
    const DevLeafTri* bvh_tris;         // every tree triangle once; .rank = its index t into bvh_tri_ref
    const int32_t* bvh_tri_ref;         // [t][2]: first, count into bvh_ref_leaf
    const DevTriReach* bvh_tri_reach;   // [t]: the first probe of reachable_rank in one record (single-step per-lane walk)
    int32_t lane_step;                  // per-lane walks one record at a time (trace_device.h: bvh_lane_step); needs bvh_c32 and bvh_tri_reach
    int32_t lane_refill_batch;          // ... idle lanes claim new rays when this many wait (or no lane walks)
    int32_t lane_chunk;                 // ... a wave claims at most this many rays with one atomic
    int32_t lane_acc_batch;             // ... pending candidates take their reachability test when this many wait (or no lane can step)
    const int32_t* bvh_up;              // single-step walk from the ORIGIN's leaf (trace_device.h: bvh_lane_start_from): [node] the sibling of
                                        // the node's parent -- where a walk that climbs goes on after the node's subtree -- or nbvh; 4 entries of padding
    const int32_t* prim_start;          // [prim][4]: the leaf holding triangle `prim`, the node after it, the leaf's sibling (or nbvh), 0; leaf -1 = none
    const DevRefLeaf* bvh_ref_leaf;     // the reference leaves holding triangle t, by rising rank
    const int32_t* ref_parent;          // parent of reference node i (layout 7), -1 = a root
    const float* ref_chain_box;         // [node][6]: lo, hi of the intersection of the boxes on node i's chain (NaN lo[0]: none)
    const float* prim_normal;   // [nprim][3] fixed-point normal (triangles)
    const int32_t* prim_material;
    const uint8_t* prim_type;
    const float* prim_geom;     // [nprim][4] sphere centre+radius (spheres only)
    const DevMaterial* materials;
    const DevLight* lights; int32_t nlights;
    // rect-light sampling of the photon renderer: GetSamplePoses (photon.h:90-111) per light, concatenated;
    // light l owns samples [light_sample_first[l], light_sample_first[l+1]); lightMeshes (final.cpp:710-731)
    const float* light_samples; const int32_t* light_sample_first; int32_t nlight_samples;
    const float* light_tris; int32_t nlight_tris;      // [nlight_tris][9]: v0, edge1, edge2
    float tol_f;                // smallest float t with (double)t > tolerance_depth  <=>  len >= tol_f
    int32_t culling;
    float shadow_bias, refract_bias;
};

// ---------------------------------------------------------------------------
// host-side flattened scene produced by scene_build.cpp
// std::vector whose resize() leaves trivially constructible elements uninitialised: the builders fill every element they
// size, and zero-filling gigabytes on one thread first costs more than the fill itself
template <class T>
struct rt_default_init_alloc : std::allocator<T> {
    template <class U> struct rebind { using other = rt_default_init_alloc<U>; };
    rt_default_init_alloc() = default;
    template <class U> rt_default_init_alloc(const rt_default_init_alloc<U>&) {}
    template <class U> void construct(U* p) { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
template <class T> using rt_uvec = std::vector<T, rt_default_init_alloc<T>>;

struct HostFlatScene {
    rt_uvec<DevNode> nodes[8];                 // nodes[7] is the reference order; the others are filled on the host only when
                                               // host_layouts is set (the device derives them: kernels_build.hip)
    std::vector<int32_t> kd_roots;             // first node of every object's tree in nodes[7]
    bool host_layouts = false;
    int device = -1;                           // the scene's device (for the device half of the build: kernels_kdbuild.hip); -1: host only
    rt_uvec<DevLeafTri> leaf_tris;
    rt_uvec<DevLeafGroup> leaf_groups;
    rt_uvec<int32_t> rank_prim;                // the triangle (scene prim) of the leaf record with visit rank r
    std::vector<DevLoose> loose;
    rt_uvec<DevBvhNode> bvh_nodes[8];          // bvh_build.cpp
    rt_uvec<DevBvhNode32> bvh_c32;
    rt_uvec<DevLeafTri> bvh_tris;
    rt_uvec<int32_t> bvh_tri_ref;
    rt_uvec<DevRefLeaf> bvh_ref_leaf;
    rt_uvec<int32_t> ref_parent;
    rt_uvec<float> ref_chain_box;
    float bvh_root_centre[3] = {0.f, 0.f, 0.f};
    rt_uvec<float> prim_normal;
    rt_uvec<float> prim_geom;
    int64_t kd_nodes_built = 0;
    int32_t kd_max_depth = 0;
    int64_t ntri = 0, nsphere = 0;
};

// kernels_kdbuild.hip: KDNode::BuildKD on the device, one tree; the result breadth-first
struct RtKdBfsNode { float center[3], extend[3]; int32_t axis, cnt, child, leaf_off; };   // child: left child's index (right follows), -1 = a leaf
int rtk_kd_build_device(int device, const rt_prim* prims, int32_t id0, int64_t nprims, const int32_t* ids, int64_t nt, float model_scale,
                        std::vector<RtKdBfsNode>* nodes, std::vector<int32_t>* leaf_ids, int32_t* max_depth, std::string* err);
// scene_build.cpp
int rt_build_flat_scene(const rt_scene_desc* d, HostFlatScene* out, std::string* err);
int rt_build_bvh(const rt_scene_desc* d, HostFlatScene* flat, std::string* err);      // bvh_build.cpp
// photon_build.cpp
void rt_photon_balance_host(rt_photon* photons, int64_t n);
void rt_photon_product_order(const rt_photon* heap, int64_t n, std::vector<int32_t>& src, std::vector<int8_t>& plane);
int rt_host_threads();     // threads of the host-side builders: RT_HOST_THREADS, else min(16, cores)

// smallest float f such that (double)f > t   (so  (double)x > t  <=>  x >= f  for every float x)
float rt_float_above(double t);

// ---------------------------------------------------------------------------
// kernel launchers (defined in the .hip files)
#include <hip/hip_runtime_api.h>
// d_order (per-lane kernels only): the order in which rays are handed to lanes (rtk_ray_order), NULL = index order
// RaySrc: where the triangle ray i STARTS ON is written, for the per-lane kernels' walks from the origin's leaf
// (trace_device.h: bvh_lane_start_from): the int32 at base + (i / div) * stride; base NULL = not known.  The children of a
// queue carry it in an array of their own (stride 4, div 1), the shadow rays of a pass read their hit's (rt_hit, div = lights).
struct RaySrc { const char* base; int32_t stride, div; };
hipError_t rtk_trace_closest(const DevSceneView& sc, const rt_ray* d_rays, const int32_t* d_active, int64_t n,
                             float eps_f, int fast, rt_hit* d_out, hipStream_t st, const int32_t* d_order = nullptr,
                             const RaySrc* src = nullptr);
hipError_t rtk_trace_any(const DevSceneView& sc, const rt_ray* d_rays, const int32_t* d_active, const float* d_tmax,
                         int64_t n, float eps_f, int fast, uint8_t* d_occ, hipStream_t st, const int32_t* d_order = nullptr,
                         const RaySrc* src = nullptr);
size_t rtk_ray_order_workspace(int64_t n);
// what rtk_ray_order sorts by: the whole key (cell of the origin, direction octant, finer Morton bits), the octant alone
// (a stable partition), or the cell alone
#define RT_ORDER_FULL 0
#define RT_ORDER_OCTANT 1
#define RT_ORDER_CELL 3
hipError_t rtk_ray_order(const rt_ray* d_rays, const int32_t* d_active, int64_t n, const float lo[3], const float hi[3],
                         void* d_work, const int32_t** order_out, hipStream_t st, int what = RT_ORDER_FULL);
hipError_t rtk_trace_work(const DevSceneView& sc, const rt_ray* d_rays, const int32_t* d_active, const float* d_tmax,
                          int64_t n, float eps_f, int fast, int any, unsigned long long* d_out4, hipStream_t st);
hipError_t rtk_generate_linear(const rt_camera& cam, int64_t first, int64_t n, float ox, float oy, rt_ray* d_rays,
                               hipStream_t st);
hipError_t rtk_generate_drt(const rt_camera& cam, const int32_t* d_tiles, int64_t ntiles, int spp, uint64_t seed,
                            float lens_radius, float focal_distance, rt_ray* d_rays, int32_t* d_pix, float* d_w,
                            uint32_t* d_code, int32_t* d_pixlist, hipStream_t st);
hipError_t rtk_generate_tiled(const rt_camera& cam, const int32_t* d_tiles, int64_t ntiles, float ox, float oy,
                              rt_ray* d_rays, int32_t* d_pix, hipStream_t st);

// kernels_trace.hip: the next word group of the scene's ring of claim counters, zeroed on `st` (one ring for every
// refilling kernel of the library, so two launches in flight never claim from the same words)
hipError_t rtk_next_ray_counter(DevSceneView& sc, hipStream_t st);
// kernels_build.hip: the other seven octant layouts of the reference trees (from layout 7) and of the BVH (from layout 0)
hipError_t rtk_kd_layouts(const DevNode* src, int32_t n, const int32_t* d_roots, int32_t nroots, DevNode* const dst[7], hipStream_t st);
hipError_t rtk_bvh_layouts(const DevBvhNode* src, int32_t n, DevBvhNode* const dst[8], hipStream_t st);
hipError_t rtk_tri_reach(const int32_t* tri_ref, const DevRefLeaf* ref_leaf, const float* chain_box, int64_t nt, DevTriReach* out, hipStream_t st);
// kernels_paths.hip: the rays of a queue with their whole subtrees of the recursion, one weighted sum per queue entry
int rtk_paths_max_lights();
size_t rtk_paths_stack_bytes(int64_t n, int depth0, int max_depth);
hipError_t rtk_paths(const DevSceneView& sc, const rt_ray* rays, const int32_t* pix, const float* w, const float* nr,
                     const uint32_t* code, int64_t n, int depth0, int max_depth, float eps_f, float ambient, const float eye[3],
                     float start_nr, uint64_t seed, float light_radius, float glossy, float* contrib, void* stack,
                     unsigned long long* counters, unsigned long long* work, hipStream_t st, const int32_t* order = nullptr,
                     const int32_t* src = nullptr);

// kernels_shade.hip
hipError_t rtk_shade_flags(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* pix, int64_t n,
                           int spawn_children, int primary_pass, uint8_t* child_flags, rt_ray* srays, int32_t* sactive,
                           int32_t* hit_ids, uint8_t* hit_any, unsigned long long* counters, unsigned int* block_pairs,
                           const uint32_t* code, uint64_t seed, float light_radius,
                           hipStream_t st);   // block_pairs: scratch, 2 x ceil(n/256) uint32
hipError_t rtk_shade_lights(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* pix,
                            const float* weight, int64_t n, const uint8_t* occluded, float ambient, const float eye[3],
                            float* contrib, unsigned long long* counters, hipStream_t st);
hipError_t rtk_accumulate(const int32_t* pix, const float* contrib, int64_t n, float* accum, hipStream_t st);
int64_t rtk_scan_blocks_needed(int64_t nflags);
hipError_t rtk_compact_children(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* pix,
                                const float* weight, const float* nr_in, const uint8_t* flags, int64_t nflags,
                                uint32_t* block_sums, uint32_t* d_total, rt_ray* orays, int32_t* opix, float* oweight,
                                float* onr, unsigned long long* counters, const uint32_t* code, uint32_t* ocode,
                                uint64_t seed, float glossy, hipStream_t st, uint8_t* ooct = nullptr, int32_t* osrc = nullptr);
// ooct (optional): one byte per child, the octant of its direction; rtk_octant_order turns them into the order in which a
// wave-synchronous pass takes the queue (a stable partition by octant)
size_t rtk_octant_order_workspace(int64_t n);
hipError_t rtk_octant_order(const uint8_t* d_oct, int64_t n, void* d_work, const int32_t** order_out, hipStream_t st);
hipError_t rtk_reduce_pairs(const unsigned int* block_pairs, int64_t nblocks, unsigned long long* counters, hipStream_t st);
// kernels_primary.hip: the primary + shadow frame in one kernel (rt_params.primary_only, fast traversal, <= 32 lights)
int rtk_primary_only_max_lights();
hipError_t rtk_primary_only(const DevSceneView& sc, const rt_camera& cam, const int32_t* d_tiles, int64_t ntiles, float eps_f,
                            float ambient, const float bg[3], uint8_t* d_rgb8, int32_t* d_hit_ids, float* d_radiance,
                            unsigned int* d_block_pairs, int bvh, int64_t packed_base, hipStream_t st);
// packed_base >= 0 (rt_params.packed_output): pixel slot i is written to rgb8[3 (packed_base + i)] instead of its place in the frame
hipError_t rtk_finalize(const int32_t* pix0, int64_t n, const float* accum, const uint8_t* hit_any, const float bg[3],
                        uint8_t* rgb8, int64_t packed_base, hipStream_t st);
hipError_t rtk_copy_radiance(const int32_t* pix0, int64_t n, const float* accum, float* out, hipStream_t st);
hipError_t rtk_zero_pixels(const int32_t* pix0, int64_t n, float* accum, uint8_t* hit_any, hipStream_t st);

hipError_t rtk_scan_flags(const uint8_t* flags, int64_t n, uint32_t* block_sums, uint32_t* d_total, hipStream_t st);

// kernels_emit.hip
hipError_t rtk_emit_photons(const DevSceneView& sc, const rt_emit_params& e, int64_t first, int64_t count, float eps_f,
                            int fast, int write, uint8_t* d_counts, const uint32_t* d_offsets, rt_photon* d_out, int64_t cap,
                            hipStream_t st);
hipError_t rtk_count_offsets(const uint8_t* d_counts, int64_t n, const uint32_t* d_block_offsets, uint32_t* d_offsets,
                             hipStream_t st);

// kernels_pm.hip (photon-mapping renderer)
hipError_t rtk_pm_generate(const DevSceneView& sc, const rt_camera& cam, const int32_t* d_tiles, int64_t tile0, int64_t nt,
                           int ntap, int use_filter, float eps_f, rt_ray* rays, int32_t* sid, float* w, uint32_t* code,
                           int32_t* pixmap, uint8_t* white, uint8_t* hit_any, int32_t* hit_ids, float* accum,
                           unsigned long long* ctr, hipStream_t st);
hipError_t rtk_pm_shade(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* sid, int64_t n,
                        int ntap, int depth0, int spawn, float start_nr, uint8_t* flags, float* nrm, float* nrmL,
                        float* qpos, uint8_t* hit_any, int32_t* hit_ids, const int32_t* pixmap, unsigned long long* ctr,
                        hipStream_t st);
hipError_t rtk_pm_shadow(const DevSceneView& sc, const rt_hit* hits, const int32_t* sid, const float* nrmL, int64_t n,
                         int S, float eps_f, int fast, uint8_t* occ, unsigned long long* ctr, hipStream_t st);
hipError_t rtk_pm_direct(const DevSceneView& sc, const rt_hit* hits, const int32_t* sid, const float* w, const float* nrmL,
                         int64_t n, int S, const uint8_t* occ, const float* caustic, const float eye[3], float* contrib,
                         hipStream_t st);
hipError_t rtk_pm_gather_spawn(const rt_hit* hits, const int32_t* sid, const float* w, const uint32_t* code,
                               const float* nrmL, const int32_t* pixmap, int64_t n, int G, int ntap, uint64_t seed,
                               rt_ray* grays, int32_t* gsid, float* gw, hipStream_t st);
hipError_t rtk_gi_shade(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* sid, int64_t n,
                        int last, float start_nr, uint8_t* flags, float* qpos, float* qnrm, float* kd,
                        unsigned long long* ctr, hipStream_t st);
hipError_t rtk_gi_contrib(const int32_t* sid, const float* w, const float* kd, const float* irr, int64_t n, float inv_g,
                          float* contrib, hipStream_t st);
hipError_t rtk_pm_spawn(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* sid, const float* w,
                        const uint32_t* code, const uint8_t* flags, int64_t nflags, uint32_t* block_sums, uint32_t* d_total,
                        float start_nr, rt_ray* orays, int32_t* osid, float* ow, uint32_t* ocode, unsigned long long* ctr,
                        hipStream_t st);
hipError_t rtk_gi_spawn(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* sid, const float* w,
                        const uint8_t* flags, int64_t nflags, uint32_t* block_sums, uint32_t* d_total, float start_nr,
                        rt_ray* orays, int32_t* osid, float* ow, hipStream_t st);
hipError_t rtk_pm_finalize(const int32_t* pixmap, int64_t nps, int ntap, int use_filter, const float* accum,
                           const uint8_t* white, const uint8_t* hit_any, const float bg[3], uint8_t* rgb8, float* radiance,
                           hipStream_t st);

// kernels_photon.hip
size_t rtk_knn_lds_bytes(int k);
hipError_t rtk_photon_build_device(rt_photon* d_photons, int64_t n, float power_scale, DevPhotonPos** d_image, float box6[6],
                                   hipStream_t st, int* status);
hipError_t rtk_photon_knn(const DevPhoton* ph, int64_t n, int64_t half, const float* d_pos3, const float* d_nrm3,
                          int64_t q, int k, float maxDist, int mode, int32_t* d_idx, float* d_d2, int32_t* d_found,
                          float* d_r2, float* d_rgb, const float* d_tables, hipStream_t st);
// cooperative kernel (mode 1 irradiance / mode 2 candidate set) + strict redo of the queries it flags; k <= 112
hipError_t rtk_photon_knn_fast(const DevPhoton* ph, int64_t n, int64_t half, const float* box6, unsigned long long* d_stats, const float* d_pos3, const float* d_nrm3,
                               int64_t q, int k, float maxDist, int mode, int32_t* d_idx, float* d_d2, int32_t* d_found,
                               float* d_r2, float* d_rgb, const float* d_tables, uint8_t* d_flagged, hipStream_t st, hipStream_t redo_st = nullptr);
hipError_t rtk_photon_irr_wave(const DevPhoton* ph, int64_t n, int64_t half, const float* box6, const float* d_pos3, const float* d_nrm3,
                               int64_t q, int k, float maxDist, float* d_rgb, const float* d_tables, int32_t* d_redo_list,
                               unsigned* d_redo_count, void* d_work, unsigned long long* d_stats, hipStream_t st);
size_t rtk_photon_irr_wave_workspace(int64_t q);
size_t rtk_knn_fast_scratch_bytes(int64_t q);
#define RT_KNN_FAST_KMAX 112
