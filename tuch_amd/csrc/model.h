// Device-resident constants of one body model (faces, packed geodesic mask,
// segment and region tables).  Created once, read-only in the hot calls.
#pragma once
#include "common.h"
#include <vector>

// Cluster tree over the faces (cluster_tree.hip): host form.  nodes[i] = {cap_off, cap_len, exact_off,
// exact_len (0 for inner nodes), skip, child0, child1, num_faces}, preorder numbering; offsets index
// the stream (vidx, sign): leaf strips first ([0, exact_len)), then the boundary caps of all nodes.
struct tuch_cluster_tree {
    int V = 0, F = 0, num_nodes = 0, exact_len = 0, stream_len = 0, num_qblocks = 0, num_heights = 0;
    std::vector<int32_t> nodes;
    std::vector<int32_t> vidx;
    std::vector<float> sign;
    std::vector<int32_t> qperm;                       // [num_qblocks * 128] vertex ids, surface-coherent
    std::vector<int32_t> height_off, height_nodes;    // nodes grouped by height (0 = leaves)
    std::vector<int32_t> frontier_off, frontier_nodes;
    // per frontier f, at frontier_off[f] * num_qblocks: (subtree index << 16 | query block), heavy first
    std::vector<int32_t> launch_order;
    std::vector<int32_t> ancestors;                   // [frontier_total][8]: ancestors of a frontier node, root first, -1 padded
    // rows [num_nodes][2]: the positions in qperm of the vertices below a node (first, count); every
    // vertex belongs to exactly one leaf (the first leaf in preorder that touches it)
    std::vector<int32_t> rows;
    std::vector<int32_t> face_leaf;                   // [F]: sequence number (preorder) of the leaf that holds the face
};

bool tuch_cluster_tree_build_impl(int V, int F, const int32_t* faces, int leaf_faces, tuch_cluster_tree& t);
void tuch_build_strips(const int32_t* faces, int F, std::vector<int32_t>& vidx, std::vector<float>& sign,
                       int* num_strips);

struct tuch_contact_model;

// Switches of the hot calls (A/B measurements, tests).  The environment variables of the same names (TUCH_ + upper
// case) are read ONCE, when the model is created; afterwards only tuch_contact_model_set_option changes them.  A hot
// call never looks at the environment: what a captured graph does cannot depend on it silently.
struct tuch_options {
    int winding_ray = 1;        // 0: never by ray crossings, 1: when only the flags are wanted, 2: also for w (crossings - fan)
    int winding_tree = 1;       // 0: flat strip walk instead of the cluster tree
    int winding_strips = 1;     // 0: per-triangle kernel
    int tree_waves = 32768;     // frontier choice of the solid-angle walk (128-query blocks)
    int ray_pair_cap = 16;      // (ray, leaf) pairs per query the pair list has room for
    int ray_waves = 32768;      // wavefronts of the crossing kernel
    int ray_fans = 0;           // the vertices' closing fans (they need the vertices only): 0 inside ray_finalize_verts_kernel, 1 by extra workgroups of the inside test's FIRST launch (ray_leaf_bounds_kernel), 2 of ray_near_kernel's launch.  Round 5, batch 64: 0.427 / 0.444 / 0.424 ms per step -- the finalize kernel drops from 28 to 6 - 13 us, but the launch that carries the fans grows by more (1: the search then starts ahead of ray_near, which crawls beside it: 117 us)
    int ray_cross = 0;          // the vertices' inside test: 0 the three launches of rounds 2 - 5 (ray_near -> ray_tiles_fill -> ray_leaf; also what arbitrary points take), 1 near lists, regrouping and crossings in ONE launch (ray_cross_kernel, round 6: one wavefront per (leaf, body), rays gathered in an LDS ring -- no lists, no pair table; 150 us alone against 51 + 24 + 100 and 6.1e7 against 6.5e7 vector instructions at batch 64, but the replayed step is 0.412 against 0.410 ms there -- the step's middle is bound by the SUM of the vector work of this test and of the search beside it, not by the chain -- and 0.214 against 0.176 ms at batch 8, where 3440 long one-wave chains are fewer than the chip's wave slots)
    int ray_cross_split = 0;    // ray_cross_kernel: wavefronts that share a leaf's query blocks (0: 4 up to batch 8, 2 up to 32, else 1)
    int v2v_tree = 1;           // 0: flat nearest-vertex search
    int v2v_waves = 0;          // frontier choice of the search (wavefronts aimed at; 0: the form's own default)
    int v2v_cap = 0;            // SMPLify-DC stage 2: 1 the search of vertices the previous iteration found outside the body starts at the loss's cap (euclthres) and the few that turn out inside are searched again behind the inside test (tuch_v2v_min_model_capped / _fix, round 6).  Exact (bit-identical fits, tests/test_gpu_properties.py); the search itself drops from 171 to 81 us inside the step at batch 64, the step does not move (0.405 against 0.415 ms, within the blocks' spread; batch 8: 0.183 against 0.177): the inside test's chain is the critical path either way, and the second pass is one more dependent launch on it.  Off.
    int v2v_flat = 2;           // search: 2 lanes over a subtree's leaves first (v2v_scan_kernel), 0 the stackless walk (v2v_tree_kernel)
    int v2v_pairs = 24;         // scan: a leaf in reach of FEWER columns of the wavefront than this is not walked row by row for all 64 lanes; its (leaf, column) pairs are queued and evaluated one per lane (round 5); 0: every leaf row by row (round 4)
    int v2v_lds = -1;           // search beside the inside test: -1 capped at 7 wavefronts per SIMD by register count (-4 / -5 / -6: at that many; round 4, lighter scan: 7 0.487, 6 0.494, 5 0.512, 4 0.530 ms per step), > 0 by an LDS allocation of that many bytes per workgroup (6400: round 2), 0 uncapped
    int seg_splits = 16;        // face splits of the solid-angle segment kernel
    int seg_assist = 1;         // 0: the segment pass counts its body-face crossings itself (read at create only)
    int seg_fused = 1;          // the segment filter behind the body test: 1 one launch (segment_one_kernel, round 4), 0 the general six-launch pass (A/B, tests)
    int canary = 0;             // 1: guard words between the regions of every workspace, see tuch_workspace_canaries
    int hd_search = 1;          // HD branch: 1 nearest admissible point on the matrix cores (hd_search.hip), 0 v2v_indexed_kernel
    int hd_search_waves = 4;    // wavefronts per block of 64 columns in that search (4, 2 or 1)
    int hd_overlap = 1;         // HD branch: the inside test of the selected points on a second stream beside their search
};
void tuch_options_from_env(tuch_options* o);

// Inside test by signed ray crossings (ray_winding.hip): exterior flags of the model's own vertices / of arbitrary
// points, identical to thresholding the winding-number sum wherever that sum is well separated from the threshold.
bool tuch_ray_available(const tuch_contact_model* m);
// (query block, leaf) entries -> the rays of every leaf, in tiles of 64 (ray_winding.hip: ray_near_kernel's lists regrouped
// by ray_tiles_fill_kernel)
struct RayEntry { int32_t leaf, node; uint32_t mask_lo, mask_hi; };   // which of the block's 64 lanes
struct RayTile { int32_t ex_off, ex_len, first, n; };                 // n <= 64 slots pairs[first ..] of one leaf
struct RayBody { int32_t tiles, overflow; };
struct TreeNode;
// computes the layout of tuch_ray_exterior_verts (Q = 0) / tuch_ray_exterior_points: for a recording tuch_ws_scope
void tuch_ray_layout_touch(const tuch_contact_model* m, int B, int Q);
size_t tuch_ray_workspace_bytes(const tuch_contact_model* m, int B, int Q);
int tuch_ray_exterior_verts(const tuch_contact_model* m, const float* verts, int B, float thresh, uint8_t* exterior,
                            float* w, void* workspace, hipStream_t s, unsigned long long* stats_host,
                            uint8_t* exterior_copy = nullptr);
// crossings of every vertex with the body faces of each of its segments, [B][tree_qblocks*128][8], left in the
// workspace of tuch_ray_exterior_verts by a model with seg_elem_mask
const int32_t* tuch_ray_segment_counts(const tuch_contact_model* m, int B, const void* workspace);
void tuch_ray_segment_prepare(const tuch_contact_model* m, const float* verts, const float* caps, int assisted, int B,
                              float* seg_entries, hipStream_t s);
bool tuch_ray_segment_fused_available(const tuch_contact_model* m);
int tuch_ray_segment_flags_one(const tuch_contact_model* m, const float* verts, const uint8_t* body_flags,
                               const int32_t* leaf_counts, int B, float thresh, uint8_t* exterior, hipStream_t s);
int tuch_ray_segment_flags(const tuch_contact_model* m, const float* verts, const float* caps, const int32_t* seg_count,
                           const int32_t* seg_list, const int32_t* leaf_counts, int B, int nsplit, float thresh, float* seg_tris,
                           int32_t* seg_partial, float* seg_w, uint8_t* seg_ext, uint8_t* exterior, hipStream_t s);
int tuch_ray_exterior_points(const tuch_contact_model* m, const float* verts, const float* points, const int32_t* counts,
                             int B, int Q, float thresh, uint8_t* exterior, float* w, void* workspace, hipStream_t s);

// nearest admissible point within ragged point sets (v2v.hip), options of the fused HD branch
int tuch_v2v_min_indexed_seeded(const float* points, const int32_t* vertex_ids, const int32_t* offsets,
                                const int32_t* counts, const float* seed_best, const int32_t* seed_arg,
                                const int32_t* all_masked_arg, const uint64_t* geomask_bits, int B, int V,
                                int max_points_per_body, float* min_d2, int32_t* argmin, void* workspace, hipStream_t s);
extern "C" size_t tuch_v2v_min_indexed_workspace_bytes(int B, int max_points_per_body);
// the same search on the matrix cores (hd_search.hip)
size_t tuch_hd_search_workspace_bytes(int B, int max_points_per_body);
int tuch_hd_search(const float* points, const int32_t* vertex_ids, const int32_t* offsets, const int32_t* counts,
                   const int32_t* all_masked_arg, const uint64_t* geomask_bits, int B, int V, int max_points_per_body,
                   float* min_d2, int32_t* argmin, void* workspace, hipStream_t s, int waves);
extern "C" size_t tuch_winding_points_workspace_bytes(const tuch_contact_model* m, int B, int Q);
extern "C" int tuch_winding_points(const tuch_contact_model* m, const float* verts, const float* points,
                                   const int32_t* counts, int B, int Q, float thresh, float* w,
                                   uint8_t* exterior, void* workspace, size_t workspace_bytes, void* stream);

struct tuch_contact_model {
    int device;
    tuch_options opt;
    int V, F;
    int32_t* faces;            // [F,3]
    int32_t* tickets;          // [8] arrival counters of "the last block adds up" kernels, zero between calls
    int32_t* canary_hits;      // [1] guard words found changed (option canary, workspace.h)
    uint64_t* mask_bits;       // [W][V] or nullptr
    // triangle strips over `faces` (built at create): stream of vertex ids with a per-element
    // sign: 0 = priming vertex (no triangle), +1/-1 = emit triangle (p-2, p-1, p) with that
    // orientation sign relative to the original face
    int strip_len, num_strips;
    int32_t* strip_vidx;       // [strip_len]
    float* strip_sign;         // [strip_len]
    // cluster tree for the hierarchical winding numbers (device copies; tree_nodes == 0: not available)
    int tree_nodes, tree_stream_len, tree_exact_len, tree_qblocks, tree_heights, tree_leaves;
    int32_t* tree_node;        // [tree_nodes][8]
    int32_t* tree_vidx;        // [tree_stream_len]
    float* tree_sign;
    int32_t* tree_sign_word;   // [like tree_sign] orientation as an integer in bits 0-23, seg_elem_mask in bits 24-31
    int32_t* tree_qperm;       // [tree_qblocks*128]
    int32_t* tree_height_off;  // [tree_heights+1]
    int32_t* tree_height_nodes;
    int32_t* tree_frontier_nodes;
    int32_t* tree_launch_order;
    int32_t* tree_ancestors;   // [frontier_total][8]
    int32_t* tree_rows;        // [tree_nodes][2]
    int32_t* tree_v2v_info;    // [tree_nodes][2]: skip pointer | leaf: first row + (rows << 20), inner node: -1; or nullptr
    // geodesic mask in the tree's vertex order: tree_mask_bits[w][j'], bit k = geomask[qperm[j']][qperm[64 w + k]],
    // w < 2 * tree_qblocks, j' < V; tree_masked[w][node] bit k: column 64 w + k has an allowed row below the node
    // (0: the mask rules the whole node out for the wavefront that owns these columns)
    uint64_t* tree_mask_bits;
    uint64_t* tree_masked;
    // the same for the leaf scan of the search (v2v.hip: v2v_scan_kernel): per frontier subtree its leaves as a range of
    // the preorder leaf sequence, and the lane table of tree_masked by leaf index instead of node
    int32_t* tree_sub_leaf;      // [frontier_total][2] = (first leaf index, number of leaves)
    uint64_t* tree_masked_leaf;  // [2 * tree_qblocks][tree_leaves]
    // packed-row form of the search (v2v.hip: v2v_scan_kernel): every leaf's rows padded to groups of four; the first
    // group of each leaf (prefix, [tree_leaves + 1]) and the mask words by padded row ([2 * tree_qblocks][4 * tree_groups],
    // 0 for a padding row)
    int32_t* tree_leaf_group;
    uint64_t* tree_mask_bits_g;
    int tree_groups;
    int tree_leaf_rows_max;    // most rows (vertices) in one leaf
    int mask_symmetric;        // geomask[i][j] == geomask[j][i] for all i, j
    int tree_num_frontiers;
    int tree_leaf_runs_tile;       // 1: the leaves' strip runs [ex_off, ex_off + ex_len) tile [0, tree_exact_len) without gaps
    int* tree_frontier_off_host;   // [tree_num_frontiers+1]
    int* tree_sub_leaf_host;       // host copy of tree_sub_leaf
    int32_t* tree_face_leaf_host;  // [F] leaf (preorder sequence number) of every face, host copy (or nullptr)
    int32_t* tree_qperm_host;      // [tree_qblocks*128] host copy of tree_qperm (or nullptr)
    // ordered one-ring of every vertex (closed manifold meshes only, else nullptr): ring_vidx[ring_off[v] + j] = r_j
    // with the faces around v being (v, r_j, r_{j+1}) in their own orientation, j cyclic (ray_winding.hip)
    int32_t* ring_off;         // [V+1]
    int32_t* ring_vidx;
    int ring_max;              // largest valence
    // segments (tuch/utils/segmentation.py): CSR over segments
    int num_segments, num_caps, seg_q_total, seg_f_total;
    int32_t* seg_q_off;        // [S+1] into seg_q_vidx
    int32_t* seg_q_vidx;       // segment_vidx lists, concatenated
    int32_t* seg_f_off;        // [S+1] into seg_faces (in faces)
    int32_t* seg_faces;        // [seg_f_total,3], cap vertex c is index V + c
    int32_t* cap_off;          // [K+1] into cap_vidx
    int32_t* cap_vidx;         // ordered boundary loops, concatenated
    // For the ray-crossing form of the segment test (ray_winding.hip), ids as in seg_faces (cap vertex c is V + c):
    // the links (x -> y) of the faces (v, x, y) around every segment vertex v within its segment, and the boundary
    // chain of every "closed" segment -- the directed edges whose reverse is missing, with their net multiplicity
    // (empty when the caps really close the segment; the reference's construction does not guarantee that).
    int32_t* seg_link_off;     // [seg_q_total+1] into seg_link (pairs)
    int32_t* seg_link;         // [.][2]
    // the segment's faces followed by its boundary edges as entries (x, y, -multiplicity): what the crossing kernel walks
    int32_t* seg_ray_off;      // [S+1] into seg_ray_ent (triples)
    int32_t* seg_ray_ent;      // [seg_ray_total][3]
    int seg_ray_total;
    // Leaf-assisted form: the crossings of a segment vertex with the BODY faces of its segment are counted by the body's
    // own inside test (ray_leaf_kernel walks those faces for that vertex anyway), the segment pass keeps the cap faces
    // and the boundary edges.  Available when the model has a cluster tree, every body face of a segment is a face of
    // the model in the same orientation, and there are at most eight segments; else nullptr.
    int32_t* seg_elem_mask;    // [tree_exact_len] per strip element: bit s = its triangle is a face of segment s
    int32_t* seg_vmask;        // [tree_qblocks*128] by tree position: bit s = the vertex is a vertex of segment s
    int32_t* seg_vpos;         // [V] tree position of every vertex
    int32_t* seg_cap_off;      // [S+1] into seg_cap_ent: as seg_ray_* without the body faces
    int32_t* seg_cap_range;    // [S+1] the caps of segment s are cap_range[s] .. cap_range[s+1] (fused segment pass), or nullptr
    int32_t* seg_cap_ent;
    int seg_cap_total;
    int num_seg_blocks;        // 64-query blocks over all segments
    int32_t* seg_blocks;       // [num_seg_blocks][2] = (segment, first query within the segment)
    int32_t* seg_of_q;         // [seg_q_total] segment of every entry of seg_q_vidx
    int* seg_q_off_host;       // host copies for grid sizing
    int* seg_f_off_host;
    int seg_q_max;
    // contact regions (ContactSigSMPL / classes): CSR over regions
    int num_regions, num_pairs, region_max;
    int32_t* region_off;       // [R+1]
    int32_t* region_vidx;
    int32_t* pairs;            // [P,2]
    // geodesic mask restricted to every region pair: rows = vertices of the first region,
    // bit k of a row = geomask[row vertex][k-th vertex of the second region]
    uint32_t* pair_mask;       // concatenated [n1][ceil(n2/32)] blocks, or nullptr
    int64_t* pair_mask_off;    // [P+1] word offsets
    int* region_off_host;
};
