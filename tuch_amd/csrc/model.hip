// tuch_contact_model: uploads the per-model constants once (faces, bit-packed
// geodesic mask, segment tables, region tables).  The only place the library
// allocates device memory; the hot calls never do.
#include "model.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include <unordered_map>

extern "C" int tuch_geomask_words(int V);

namespace {

template <typename T>
int upload(T** dst, const T* src, size_t count)
{
    return tuch_table_upload((void**)dst, src, count * sizeof(T));
}

int* host_copy(const int32_t* src, size_t n)
{
    int* p = (int*)malloc(sizeof(int) * (n ? n : 1));
    if (n) memcpy(p, src, sizeof(int) * n);
    return p;
}

}  // namespace

extern "C" void tuch_contact_model_destroy(tuch_contact_model* m)
{
    if (!m) return;
    void* dev[] = {m->ring_off, m->ring_vidx, m->faces, m->mask_bits, m->strip_vidx, m->strip_sign, m->tree_node, m->tree_vidx, m->tree_sign, m->tree_sign_word, m->tree_qperm,
                   m->tree_height_off, m->tree_height_nodes, m->tree_frontier_nodes, m->tree_launch_order, m->tree_ancestors, m->tree_rows, m->tree_v2v_info, m->tree_mask_bits, m->tree_masked, m->tree_sub_leaf, m->tree_masked_leaf, m->tree_leaf_group, m->tree_mask_bits_g, m->seg_blocks, m->seg_of_q, m->seg_q_off, m->seg_q_vidx, m->seg_f_off, m->seg_faces, m->seg_link_off, m->seg_link, m->seg_ray_off, m->seg_ray_ent, m->seg_elem_mask, m->seg_vmask, m->seg_vpos, m->seg_cap_off, m->seg_cap_ent, m->seg_cap_range,
                   m->cap_off, m->cap_vidx, m->region_off, m->region_vidx, m->pairs, m->pair_mask, m->pair_mask_off, m->tickets, m->canary_hits};
    for (void* p : dev) tuch_table_free(p);
    free(m->tree_frontier_off_host);
    free(m->tree_sub_leaf_host);
    free(m->tree_face_leaf_host);
    free(m->tree_qperm_host);
    free(m->seg_q_off_host);
    free(m->seg_f_off_host);
    free(m->region_off_host);
    free(m);
}

extern "C" int tuch_contact_model_create(
    tuch_contact_model** out, int V, int F, const int32_t* faces,
    const uint8_t* geomask,   // host [V,V] bytes (geod > geothres) or NULL
    int num_segments, const int32_t* seg_q_off, const int32_t* seg_q_vidx,
    const int32_t* seg_f_off, const int32_t* seg_faces,
    int num_caps, const int32_t* cap_off, const int32_t* cap_vidx,
    int num_regions, const int32_t* region_off, const int32_t* region_vidx,
    int num_pairs, const int32_t* pairs)
{
    TUCH_REQUIRE(out && faces && V > 0 && F > 0, "tuch_contact_model_create: bad mesh arguments");
    TUCH_REQUIRE(num_segments >= 0 && num_caps >= 0 && num_regions >= 0 && num_pairs >= 0,
                 "tuch_contact_model_create: negative table size");
    TUCH_REQUIRE(num_segments == 0 || (seg_q_off && seg_q_vidx && seg_f_off && seg_faces),
                 "tuch_contact_model_create: segment tables missing");
    TUCH_REQUIRE(num_caps == 0 || (cap_off && cap_vidx), "tuch_contact_model_create: cap tables missing");
    TUCH_REQUIRE(num_pairs == 0 || (num_regions > 0 && region_off && region_vidx && pairs),
                 "tuch_contact_model_create: region tables missing");
    for (int i = 0; i < F * 3; ++i)
        TUCH_REQUIRE(faces[i] >= 0 && faces[i] < V, "tuch_contact_model_create: face index %d out of range",
                     faces[i]);
    tuch_contact_model* m = (tuch_contact_model*)calloc(1, sizeof(tuch_contact_model));
    m->V = V;
    m->F = F;
    m->opt = tuch_options();
    tuch_options_from_env(&m->opt);          // the only place the hot calls' switches are read from the environment
    (void)hipGetDevice(&m->device);
    int rc = upload(&m->faces, faces, (size_t)F * 3);
    if (rc == TUCH_OK) {
        const int32_t zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        rc = upload(&m->tickets, zeros, 8);
        if (rc == TUCH_OK) rc = upload(&m->canary_hits, zeros, 1);
    }
    if (rc == TUCH_OK) {
        std::vector<int32_t> sv;
        std::vector<float> ss;
        tuch_build_strips(faces, F, sv, ss, &m->num_strips);
        m->strip_len = (int)sv.size();
        rc = upload(&m->strip_vidx, sv.data(), sv.size());
        if (rc == TUCH_OK) rc = upload(&m->strip_sign, ss.data(), ss.size());
    }
    std::vector<int32_t> tree_vidx_host, tree_qperm_host;     // for the segment tables further down
    std::vector<float> tree_sign_host;
    std::vector<int32_t> elem_mask_host;                       // per strip element: the segments that list its triangle
    int tree_exact_host = 0;
    if (rc == TUCH_OK) {
        // cluster tree for the hierarchical winding numbers; a mesh that is not a closed manifold (or
        // has too many clusters for the LDS-resident boxes) simply keeps the flat strip path
        tuch_cluster_tree t;
        const char* e = getenv("TUCH_TREE_LEAF_FACES");
        // at most 1800 nodes: their slabs (80 B) and child indices (8 B) are staged in the CU's 160 KB of LDS by
        // tree_inner_bounds_kernel
        // 32 faces per leaf (SMPL: 430 leaves): measured best for the step at batch 64 -- 16: 104.5, 24: 109.2, 32: 110.3,
        // 40: 109.4, 48: 108.4, 64: 104.7, 96: 98.6 k body iterations/s (tighter slabs and boxes against more leaves to
        // test and emptier tiles); larger meshes get larger leaves so that the tree stays under the node limit below
        const int leaf_faces = e ? atoi(e) : (F / 850 > 32 ? F / 850 : 32);
        if (tuch_cluster_tree_build_impl(V, F, faces, leaf_faces, t) && t.num_nodes <= 1800) {
            m->tree_nodes = t.num_nodes;
            m->tree_stream_len = t.stream_len;
            m->tree_exact_len = t.exact_len;
            m->tree_qblocks = t.num_qblocks;
            m->tree_heights = t.num_heights;
            m->tree_leaves = t.height_off[1];
            {   // do the leaves' strip runs tile the exact part of the stream?  (ray_winding.hip poses the strip leaf by leaf)
                std::vector<std::pair<int, int>> runs;
                for (int i = 0; i < t.height_off[1]; ++i) {
                    const int nd = t.height_nodes[i];
                    runs.emplace_back(t.nodes[(size_t)nd * 8 + 2], t.nodes[(size_t)nd * 8 + 3]);
                }
                std::sort(runs.begin(), runs.end());
                int at = 0;
                bool tile = true;
                for (const auto& r : runs) { tile = tile && r.first == at; at = r.first + r.second; }
                m->tree_leaf_runs_tile = tile && at == t.exact_len ? 1 : 0;
            }
            m->tree_num_frontiers = (int)t.frontier_off.size() - 1;
            m->tree_frontier_off_host = host_copy(t.frontier_off.data(), t.frontier_off.size());
            m->tree_face_leaf_host = host_copy(t.face_leaf.data(), t.face_leaf.size());
            m->tree_qperm_host = host_copy(t.qperm.data(), t.qperm.size());
            tree_vidx_host = t.vidx; tree_sign_host = t.sign; tree_qperm_host = t.qperm; tree_exact_host = t.exact_len;
            rc = upload(&m->tree_node, t.nodes.data(), t.nodes.size());
            if (rc == TUCH_OK) rc = upload(&m->tree_vidx, t.vidx.data(), t.vidx.size());
            if (rc == TUCH_OK) rc = upload(&m->tree_sign, t.sign.data(), t.sign.size());
            if (rc == TUCH_OK) rc = upload(&m->tree_qperm, t.qperm.data(), t.qperm.size());
            if (rc == TUCH_OK) rc = upload(&m->tree_height_off, t.height_off.data(), t.height_off.size());
            if (rc == TUCH_OK) rc = upload(&m->tree_height_nodes, t.height_nodes.data(), t.height_nodes.size());
            if (rc == TUCH_OK) rc = upload(&m->tree_frontier_nodes, t.frontier_nodes.data(), t.frontier_nodes.size());
            if (rc == TUCH_OK) rc = upload(&m->tree_launch_order, t.launch_order.data(), t.launch_order.size());
            if (rc == TUCH_OK) rc = upload(&m->tree_ancestors, t.ancestors.data(), t.ancestors.size());
            if (rc == TUCH_OK) rc = upload(&m->tree_rows, t.rows.data(), t.rows.size());
            if (rc == TUCH_OK) {
                // what the nearest-vertex walk needs of a node besides its box (rides in the box's padding, v2v.hip)
                std::vector<int32_t> info((size_t)t.num_nodes * 2);
                bool fits = V < (1 << 20);
                for (int i = 0; i < t.num_nodes; ++i) {
                    const bool leaf = t.nodes[(size_t)i * 8 + 5] < 0;
                    const int off = t.rows[(size_t)i * 2], n = t.rows[(size_t)i * 2 + 1];
                    fits = fits && (!leaf || n < (1 << 11));
                    info[2 * i] = t.nodes[(size_t)i * 8 + 4];
                    info[2 * i + 1] = leaf ? (off | (n << 20)) : -1;
                }
                if (fits) rc = upload(&m->tree_v2v_info, info.data(), info.size());
            }
            if (rc == TUCH_OK && geomask) {
                // the mask in the tree's vertex order, and which (query block, node) pairs it rules out entirely
                const int Wp = 2 * t.num_qblocks, N = t.num_nodes;
                std::vector<uint64_t> bits((size_t)Wp * V, 0);
                for (int jp = 0; jp < V; ++jp) {
                    const uint8_t* row = geomask + (size_t)t.qperm[jp] * V;
                    for (int ip = 0; ip < V; ++ip)
                        if (row[t.qperm[ip]]) bits[(size_t)(ip >> 6) * V + jp] |= (uint64_t)1 << (ip & 63);
                }
                // per 64-column block (one wavefront's columns) and node: the columns with ANY allowed row below the node
                std::vector<uint64_t> lanes((size_t)Wp * N, 0);
                for (int qb = 0; qb < Wp; ++qb) {
                    const uint64_t* w0 = bits.data() + (size_t)qb * V;
                    for (int i = N - 1; i >= 0; --i) {
                        const int c0 = t.nodes[(size_t)i * 8 + 5], c1 = t.nodes[(size_t)i * 8 + 6];
                        uint64_t any = 0;
                        if (c0 >= 0) {
                            any = lanes[(size_t)qb * N + c0] | lanes[(size_t)qb * N + c1];
                        } else {
                            const int lo = t.rows[(size_t)i * 2], n = t.rows[(size_t)i * 2 + 1];
                            for (int j = lo; j < lo + n; ++j) any |= w0[j];
                        }
                        lanes[(size_t)qb * N + i] = any;
                    }
                }
                rc = upload(&m->tree_mask_bits, bits.data(), bits.size());
                if (rc == TUCH_OK) rc = upload(&m->tree_masked, lanes.data(), lanes.size());
                // flat form: leaves by their preorder sequence (= their position among the height-0 nodes, which are
                // listed in ascending node order); a subtree's leaves are a contiguous range of it
                const int L = t.height_off[1];
                std::vector<int32_t> leaf_index(N, -1);
                bool ascending = true;
                for (int i = 0; i < L; ++i) {
                    leaf_index[t.height_nodes[i]] = i;
                    ascending = ascending && (i == 0 || t.height_nodes[i] > t.height_nodes[i - 1]);
                }
                if (rc == TUCH_OK && ascending) {
                    std::vector<int32_t> sub((size_t)t.frontier_nodes.size() * 2);
                    for (size_t k = 0; k < t.frontier_nodes.size(); ++k) {
                        const int lo = t.frontier_nodes[k], hi = t.nodes[(size_t)lo * 8 + 4];
                        int first = -1, count = 0;
                        for (int i = lo; i < hi; ++i)
                            if (leaf_index[i] >= 0) { if (first < 0) first = leaf_index[i]; ++count; }
                        sub[2 * k] = first < 0 ? 0 : first;
                        sub[2 * k + 1] = count;
                    }
                    std::vector<uint64_t> by_leaf((size_t)Wp * L + 8, 0);        // + padding
                    for (int qb = 0; qb < Wp; ++qb)
                        for (int i = 0; i < L; ++i) by_leaf[(size_t)qb * L + i] = lanes[(size_t)qb * N + t.height_nodes[i]];
                    m->tree_sub_leaf_host = host_copy(sub.data(), sub.size());
                    rc = upload(&m->tree_sub_leaf, sub.data(), sub.size());
                    if (rc == TUCH_OK) rc = upload(&m->tree_masked_leaf, by_leaf.data(), by_leaf.size());
                    // packed-row form: the leaves' rows in groups of four
                    std::vector<int32_t> group(L + 1, 0);
                    for (int i = 0; i < L; ++i) group[i + 1] = group[i] + (t.rows[(size_t)t.height_nodes[i] * 2 + 1] + 3) / 4;
                    // for the leaf-major search (v2v.hip: v2v_tiles_kernel): a leaf's rows fit one 64-bit window of a column's
                    // mask row, and that row stands for the column's own admissible rows only if the mask is symmetric
                    m->tree_leaf_rows_max = 0;
                    for (int i = 0; i < L; ++i) m->tree_leaf_rows_max = std::max(m->tree_leaf_rows_max, (int)t.rows[(size_t)t.height_nodes[i] * 2 + 1]);
                    m->mask_symmetric = 1;
                    for (int a = 0; a < V && m->mask_symmetric; ++a)
                        for (int c2 = a + 1; c2 < V; ++c2)
                            if ((geomask[(size_t)a * V + c2] != 0) != (geomask[(size_t)c2 * V + a] != 0)) { m->mask_symmetric = 0; break; }
                    const int G = group[L];
                    std::vector<uint64_t> bits_g((size_t)Wp * G * 4 + 8, 0);
                    for (int qb = 0; qb < Wp; ++qb)
                        for (int i = 0; i < L; ++i) {
                            const int lo = t.rows[(size_t)t.height_nodes[i] * 2], n = t.rows[(size_t)t.height_nodes[i] * 2 + 1];
                            for (int k = 0; k < n; ++k)
                                bits_g[(size_t)qb * G * 4 + (size_t)group[i] * 4 + k] = bits[(size_t)qb * V + lo + k];
                        }
                    m->tree_groups = G;
                    if (rc == TUCH_OK) rc = upload(&m->tree_leaf_group, group.data(), group.size());
                    if (rc == TUCH_OK) rc = upload(&m->tree_mask_bits_g, bits_g.data(), bits_g.size());
                }
            }
        }
    }
    if (rc == TUCH_OK && m->tree_nodes > 0) {
        // ordered one-rings (the tree exists, so the mesh is a closed manifold): face (v, a, b) in its cyclic order
        // contributes the directed link edge a -> b; the link of v is the single cycle through them
        std::vector<int32_t> deg(V + 1, 0);
        for (int i = 0; i < F * 3; ++i) ++deg[faces[i] + 1];
        for (int v = 0; v < V; ++v) deg[v + 1] += deg[v];
        std::vector<int32_t> from((size_t)F * 3), to((size_t)F * 3), fill(deg.begin(), deg.end() - 1);
        for (int f = 0; f < F; ++f)
            for (int k = 0; k < 3; ++k) {
                const int v = faces[3 * f + k], slot = fill[v]++;
                from[slot] = faces[3 * f + (k + 1) % 3];
                to[slot] = faces[3 * f + (k + 2) % 3];
            }
        std::vector<int32_t> ring((size_t)F * 3);
        bool ok = true;
        int ring_max = 0;
        for (int v = 0; v < V && ok; ++v) {
            const int lo = deg[v], n = deg[v + 1] - lo;
            if (n < 3) { ok = false; break; }
            if (n > ring_max) ring_max = n;
            int cur = from[lo];
            for (int j = 0; j < n && ok; ++j) {
                ring[lo + j] = cur;
                int nxt = -1;
                for (int e = lo; e < lo + n; ++e)
                    if (from[e] == cur) { nxt = to[e]; break; }
                if (nxt < 0) ok = false;
                cur = nxt;
            }
            if (ok && cur != from[lo]) ok = false;          // the cycle must close after exactly n steps
            for (int j = 0; j < n && ok; ++j)               // ... and visit n distinct vertices
                for (int i = 0; i < j; ++i)
                    if (ring[lo + i] == ring[lo + j]) ok = false;
        }
        if (ok) {
            m->ring_max = ring_max;
            rc = upload(&m->ring_off, deg.data(), deg.size());
            if (rc == TUCH_OK) rc = upload(&m->ring_vidx, ring.data(), ring.size());
        }
    }
    if (rc == TUCH_OK && geomask) {
        // bit-pack on the host: bits[w][j], bit k = geomask[j][64 w + k]
        const int W = tuch_geomask_words(V);
        std::vector<uint64_t> bits((size_t)W * V, 0);
        for (int j = 0; j < V; ++j) {
            const uint8_t* row = geomask + (size_t)j * V;
            for (int i = 0; i < V; ++i)
                if (row[i]) bits[(size_t)(i >> 6) * V + j] |= (uint64_t)1 << (i & 63);
        }
        rc = upload(&m->mask_bits, bits.data(), bits.size());
    }
    if (rc == TUCH_OK && num_segments > 0) {
        m->num_segments = num_segments;
        m->num_caps = num_caps;
        m->seg_q_total = seg_q_off[num_segments];
        m->seg_f_total = seg_f_off[num_segments];
        m->seg_q_off_host = host_copy(seg_q_off, num_segments + 1);
        m->seg_f_off_host = host_copy(seg_f_off, num_segments + 1);
        for (int s = 0; s < num_segments; ++s) {
            const int n = seg_q_off[s + 1] - seg_q_off[s];
            if (n > m->seg_q_max) m->seg_q_max = n;
        }
        for (int i = 0; i < m->seg_f_total * 3 && rc == TUCH_OK; ++i)
            if (seg_faces[i] < 0 || seg_faces[i] >= V + num_caps) {
                tuch_set_error("tuch_contact_model_create: segment face index %d out of range", seg_faces[i]);
                rc = TUCH_ERR_ARG;
            }
        std::vector<int32_t> blocks;
        for (int s = 0; s < num_segments; ++s)
            for (int q = 0; q < seg_q_off[s + 1] - seg_q_off[s]; q += 64) { blocks.push_back(s); blocks.push_back(q); }
        m->num_seg_blocks = (int)blocks.size() / 2;
        std::vector<int32_t> seg_of_q((size_t)m->seg_q_total);
        for (int s = 0; s < num_segments; ++s)
            for (int q = seg_q_off[s]; q < seg_q_off[s + 1]; ++q) seg_of_q[q] = s;
        if (rc == TUCH_OK) rc = upload(&m->seg_of_q, seg_of_q.data(), seg_of_q.size());
        if (rc == TUCH_OK) rc = upload(&m->seg_blocks, blocks.data(), blocks.size());
        if (rc == TUCH_OK) rc = upload(&m->seg_q_off, seg_q_off, (size_t)num_segments + 1);
        if (rc == TUCH_OK) rc = upload(&m->seg_q_vidx, seg_q_vidx, (size_t)m->seg_q_total);
        if (rc == TUCH_OK) rc = upload(&m->seg_f_off, seg_f_off, (size_t)num_segments + 1);
        if (rc == TUCH_OK) rc = upload(&m->seg_faces, seg_faces, (size_t)m->seg_f_total * 3);
        if (rc == TUCH_OK && num_caps > 0) {
            rc = upload(&m->cap_off, cap_off, (size_t)num_caps + 1);
            if (rc == TUCH_OK) rc = upload(&m->cap_vidx, cap_vidx, (size_t)cap_off[num_caps]);
        }
        // Tables for the ray-crossing form of the segment test: the star of every segment vertex as links, and the
        // boundary chain of every segment mesh (model.h).
        if (rc == TUCH_OK) {
            std::vector<int32_t> loff(1, 0), links, eoff(1, 0), ent;
            for (int sg = 0; sg < num_segments; ++sg) {
                const int32_t* fs = seg_faces + 3 * (size_t)seg_f_off[sg];
                const int nf = seg_f_off[sg + 1] - seg_f_off[sg];
                auto key = [](int a, int b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; };
                std::unordered_map<uint64_t, int> net;                // directed edge -> occurrences - occurrences reversed
                std::unordered_map<int, std::vector<int32_t>> star;   // vertex -> (x, y) of its faces (v, x, y)
                for (int f = 0; f < nf; ++f)
                    for (int k = 0; k < 3; ++k) {
                        const int v = fs[3 * f + k], x = fs[3 * f + (k + 1) % 3], y = fs[3 * f + (k + 2) % 3];
                        if (v < x) ++net[key(v, x)]; else --net[key(x, v)];
                        star[v].push_back(x);
                        star[v].push_back(y);
                    }
                ent.insert(ent.end(), fs, fs + 3 * (size_t)nf);
                std::vector<uint64_t> open_edges;
                for (const auto& e : net)
                    if (e.second != 0) open_edges.push_back(e.first);
                std::sort(open_edges.begin(), open_edges.end());      // a fixed order of the sums
                for (uint64_t e : open_edges) {
                    const int a = (int)(e >> 32), b2 = (int)(uint32_t)e, mlt = net[e];
                    // closing chain = links - boundary: the boundary edge x -> y enters with -multiplicity
                    if (mlt > 0) { ent.push_back(a); ent.push_back(b2); ent.push_back(-mlt); }
                    else { ent.push_back(b2); ent.push_back(a); ent.push_back(mlt); }
                }
                eoff.push_back((int32_t)(ent.size() / 3));
                for (int q = seg_q_off[sg]; q < seg_q_off[sg + 1]; ++q) {
                    const auto it = star.find(seg_q_vidx[q]);
                    if (it != star.end()) links.insert(links.end(), it->second.begin(), it->second.end());
                    loff.push_back((int32_t)(links.size() / 2));
                }
            }
            if (links.empty()) links.assign(2, 0);
            m->seg_ray_total = (int)(ent.size() / 3);
            rc = upload(&m->seg_link_off, loff.data(), loff.size());
            if (rc == TUCH_OK) rc = upload(&m->seg_link, links.data(), links.size());
            if (rc == TUCH_OK) rc = upload(&m->seg_ray_off, eoff.data(), eoff.size());
            if (rc == TUCH_OK) rc = upload(&m->seg_ray_ent, ent.data(), ent.size());
            // leaf-assisted form (model.h)
            // option seg_assist = 0: keep the segment pass self-contained (A/B, tests)
            bool assist = rc == TUCH_OK && m->tree_nodes > 0 && tree_exact_host > 0 && num_segments <= 8 && m->opt.seg_assist != 0;
            if (assist) {
                auto key3 = [](int a, int b, int c) {          // rotation with the smallest id first: orientation kept
                    if (b < a && b < c) { const int t2 = a; a = b; b = c; c = t2; }
                    else if (c < a && c < b) { const int t2 = c; c = b; b = a; a = t2; }
                    return ((uint64_t)(uint32_t)a << 42) ^ ((uint64_t)(uint32_t)b << 21) ^ (uint64_t)(uint32_t)c;
                };
                assist = V < (1 << 21);
                std::unordered_map<uint64_t, int32_t> face_mask;      // body face -> segments that list it
                for (int f = 0; f < F && assist; ++f) face_mask[key3(faces[3 * f], faces[3 * f + 1], faces[3 * f + 2])] = 0;
                std::vector<int32_t> coff(1, 0), cent;
                for (int sg = 0; sg < num_segments && assist; ++sg) {
                    for (int e = eoff[sg]; e < eoff[sg + 1]; ++e) {
                        const int32_t* t3 = &ent[3 * (size_t)e];
                        if (t3[2] >= 0 && t3[0] < V && t3[1] < V && t3[2] < V) {       // a body face of the segment
                            const auto it = face_mask.find(key3(t3[0], t3[1], t3[2]));
                            if (it == face_mask.end() || (it->second >> sg) & 1) { if (getenv("TUCH_DEBUG")) fprintf(stderr, "assist: seg %d face (%d %d %d) %s\n", sg, t3[0], t3[1], t3[2], it == face_mask.end() ? "not a body face" : "listed twice"); assist = false; break; }
                            it->second |= 1 << sg;
                        } else {                                                        // cap face or boundary edge
                            cent.insert(cent.end(), t3, t3 + 3);
                        }
                    }
                    coff.push_back((int32_t)(cent.size() / 3));
                }
                std::vector<int32_t> vseg((size_t)V, 0);
                for (int sg = 0; sg < num_segments && assist; ++sg)
                    for (int q = seg_q_off[sg]; q < seg_q_off[sg + 1]; ++q) vseg[seg_q_vidx[q]] |= 1 << sg;
                if (assist) {
                    std::vector<int32_t> emask((size_t)(tree_exact_host + 2) / 3 * 3 + 6, 0),    // padded like the posed stream
                                          vmask(tree_qperm_host.size(), 0), vpos(V, 0);
                    for (int p2 = 2; p2 < tree_exact_host; ++p2) {
                        if (tree_sign_host[p2] == 0.0f) continue;
                        int a = tree_vidx_host[p2 - 2], b2 = tree_vidx_host[p2 - 1], c = tree_vidx_host[p2];
                        if (tree_sign_host[p2] < 0.0f) { const int t2 = a; a = b2; b2 = t2; }     // odd permutation of the face
                        const auto it = face_mask.find(key3(a, b2, c));
                        if (it == face_mask.end()) { if (getenv("TUCH_DEBUG")) fprintf(stderr, "assist: stream element %d not a face\n", p2); assist = false; break; }
                        emask[p2] = it->second;
                    }
                    for (size_t i = 0; i < tree_qperm_host.size(); ++i) vmask[i] = vseg[tree_qperm_host[i]];
                    for (int i = V - 1; i >= 0; --i) vpos[tree_qperm_host[i]] = i;
                    if (cent.empty()) cent.assign(3, 0);
                    if (assist) {
                        m->seg_cap_total = coff.back();
                        elem_mask_host = emask;
                        rc = upload(&m->seg_elem_mask, emask.data(), emask.size());
                        if (rc == TUCH_OK) rc = upload(&m->seg_vmask, vmask.data(), vmask.size());
                        if (rc == TUCH_OK) rc = upload(&m->seg_vpos, vpos.data(), vpos.size());
                        if (rc == TUCH_OK) rc = upload(&m->seg_cap_off, coff.data(), coff.size());
                        if (rc == TUCH_OK) rc = upload(&m->seg_cap_ent, cent.data(), cent.size());
                        // the caps a segment's entries and links refer to: consecutive cap ids (the fused segment pass
                        // keeps a segment's centroids in LDS), segments in order; anything else keeps the six-launch pass
                        std::vector<int32_t> crange(1, 0);
                        bool consecutive = true;
                        for (int sg = 0; sg < num_segments && consecutive; ++sg) {
                            int lo = num_caps, hi = -1;
                            auto see = [&](int id) { if (id >= V) { lo = std::min(lo, id - V); hi = std::max(hi, id - V); } };
                            for (int e = coff[sg]; e < coff[sg + 1]; ++e)
                                for (int k = 0; k < 3; ++k) see(cent[3 * (size_t)e + k]);
                            for (int q = seg_q_off[sg]; q < seg_q_off[sg + 1]; ++q)
                                for (int e = loff[q]; e < loff[q + 1]; ++e) { see(links[2 * (size_t)e]); see(links[2 * (size_t)e + 1]); }
                            if (hi < 0) { crange.push_back(crange.back()); continue; }
                            consecutive = lo == crange.back() && hi - lo + 1 <= 8;
                            crange.push_back(hi + 1);
                        }
                        if (rc == TUCH_OK && consecutive) rc = upload(&m->seg_cap_range, crange.data(), crange.size());
                    }
                }
            }
        }
    }
    if (rc == TUCH_OK && num_regions > 0) {
        m->num_regions = num_regions;
        m->num_pairs = num_pairs;
        m->region_off_host = host_copy(region_off, num_regions + 1);
        for (int r = 0; r < num_regions; ++r) {
            const int n = region_off[r + 1] - region_off[r];
            if (n > m->region_max) m->region_max = n;
        }
        rc = upload(&m->region_off, region_off, (size_t)num_regions + 1);
        if (rc == TUCH_OK) rc = upload(&m->region_vidx, region_vidx, (size_t)region_off[num_regions]);
        if (rc == TUCH_OK && num_pairs > 0) rc = upload(&m->pairs, pairs, (size_t)num_pairs * 2);
        if (rc == TUCH_OK && num_pairs > 0 && geomask) {
            std::vector<int64_t> off(num_pairs + 1, 0);
            for (int p = 0; p < num_pairs; ++p) {
                const int n1 = region_off[pairs[2 * p] + 1] - region_off[pairs[2 * p]];
                const int n2 = region_off[pairs[2 * p + 1] + 1] - region_off[pairs[2 * p + 1]];
                off[p + 1] = off[p] + (int64_t)n1 * ((n2 + 31) / 32);
            }
            std::vector<uint32_t> words((size_t)off[num_pairs], 0u);
            for (int p = 0; p < num_pairs; ++p) {
                const int32_t* r1 = region_vidx + region_off[pairs[2 * p]];
                const int32_t* r2 = region_vidx + region_off[pairs[2 * p + 1]];
                const int n1 = region_off[pairs[2 * p] + 1] - region_off[pairs[2 * p]];
                const int n2 = region_off[pairs[2 * p + 1] + 1] - region_off[pairs[2 * p + 1]];
                const int wpr = (n2 + 31) / 32;
                for (int a = 0; a < n1; ++a)
                    for (int k = 0; k < n2; ++k)
                        if (geomask[(size_t)r1[a] * V + r2[k]])
                            words[(size_t)off[p] + (size_t)a * wpr + (k >> 5)] |= 1u << (k & 31);
            }
            rc = upload(&m->pair_mask, words.data(), words.size());
            if (rc == TUCH_OK) rc = upload(&m->pair_mask_off, off.data(), off.size());
        }
    }
    if (rc == TUCH_OK && !tree_sign_host.empty()) {
        // the fourth word of a posed strip element (ray_winding.hip: RayElem): the orientation as the INTEGER +1 / -1 / 0 in
        // bits 0-23 and the element's segments (seg_elem_mask) in bits 24-31 -- both wave-uniform in the crossing kernel,
        // read and tested on the scalar unit
        std::vector<int32_t> word(tree_sign_host.size(), 0);
        for (size_t p = 0; p < word.size(); ++p) {
            const int32_t sg = tree_sign_host[p] > 0.0f ? 1 : tree_sign_host[p] < 0.0f ? -1 : 0;
            const int32_t em = p < elem_mask_host.size() ? elem_mask_host[p] : 0;
            word[p] = (int32_t)(((uint32_t)em << 24) | ((uint32_t)sg & 0xffffffu));
        }
        rc = upload(&m->tree_sign_word, word.data(), word.size());
    }
    if (rc != TUCH_OK) {
        tuch_contact_model_destroy(m);
        *out = nullptr;
        return rc;
    }
    *out = m;
    return TUCH_OK;
}

namespace {
struct OptionName { const char* name; int tuch_options::*field; };
const OptionName kOptions[] = {
    {"winding_ray", &tuch_options::winding_ray}, {"winding_tree", &tuch_options::winding_tree},
    {"winding_strips", &tuch_options::winding_strips}, {"tree_waves", &tuch_options::tree_waves},
    {"ray_pair_cap", &tuch_options::ray_pair_cap}, {"ray_waves", &tuch_options::ray_waves}, {"ray_fans", &tuch_options::ray_fans}, {"ray_cross", &tuch_options::ray_cross}, {"v2v_cap", &tuch_options::v2v_cap}, {"ray_cross_split", &tuch_options::ray_cross_split},
    {"v2v_tree", &tuch_options::v2v_tree}, {"v2v_flat", &tuch_options::v2v_flat}, {"v2v_pairs", &tuch_options::v2v_pairs}, {"v2v_waves", &tuch_options::v2v_waves}, {"v2v_lds", &tuch_options::v2v_lds},
    {"seg_splits", &tuch_options::seg_splits}, {"seg_assist", &tuch_options::seg_assist}, {"seg_fused", &tuch_options::seg_fused},
    {"canary", &tuch_options::canary}, {"hd_search", &tuch_options::hd_search}, {"hd_search_waves", &tuch_options::hd_search_waves}, {"hd_overlap", &tuch_options::hd_overlap},
};
}  // namespace

void tuch_options_from_env(tuch_options* o)
{
    for (const OptionName& k : kOptions) {
        char env[64] = "TUCH_";
        size_t n = strlen(env);
        for (const char* c = k.name; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)toupper((unsigned char)*c);
        env[n] = 0;
        const char* e = getenv(env);
        if (e && *e) o->*(k.field) = atoi(e);
    }
}

extern "C" int tuch_contact_model_set_option(tuch_contact_model* m, const char* name, int value)
{
    TUCH_REQUIRE(m && name, "tuch_contact_model_set_option: null argument");
    for (const OptionName& k : kOptions)
        if (!strcmp(k.name, name)) {
            TUCH_REQUIRE(strcmp(name, "seg_assist") != 0, "tuch_contact_model_set_option: seg_assist is fixed when the model "
                         "is created (TUCH_SEG_ASSIST)");
            m->opt.*(k.field) = value;
            return TUCH_OK;
        }
    tuch_set_error("tuch_contact_model_set_option: unknown option '%s'", name);
    return TUCH_ERR_ARG;
}

extern "C" int tuch_contact_model_canary_hits(const tuch_contact_model* m, int* hits_host, int reset)
{
    TUCH_REQUIRE(m && hits_host, "tuch_contact_model_canary_hits: null argument");
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(hits_host, m->canary_hits, sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess ||
        (reset && hipMemset(m->canary_hits, 0, sizeof(int32_t)) != hipSuccess)) {
        tuch_set_error("tuch_contact_model_canary_hits: %s", hipGetErrorString(hipGetLastError()));
        return TUCH_ERR_HIP;
    }
    return TUCH_OK;
}

extern "C" int tuch_contact_model_get_option(const tuch_contact_model* m, const char* name, int* value)
{
    TUCH_REQUIRE(m && name && value, "tuch_contact_model_get_option: null argument");
    if (!strcmp(name, "seg_fused_active")) {        // read-only: does the segment filter run as the one fused launch?
        *value = tuch_ray_segment_fused_available(m) ? 1 : 0;
        return TUCH_OK;
    }
    for (const OptionName& k : kOptions)
        if (!strcmp(k.name, name)) {
            *value = m->opt.*(k.field);
            return TUCH_OK;
        }
    tuch_set_error("tuch_contact_model_get_option: unknown option '%s'", name);
    return TUCH_ERR_ARG;
}

extern "C" const uint64_t* tuch_contact_model_mask_bits(const tuch_contact_model* m)
{
    return m ? m->mask_bits : nullptr;
}

extern "C" const uint64_t* tuch_contact_model_tree_mask_bits(const tuch_contact_model* m)
{
    return m ? m->tree_mask_bits : nullptr;
}

extern "C" int32_t* tuch_contact_model_tickets(const tuch_contact_model* m)
{
    return m ? m->tickets : nullptr;
}

extern "C" const int32_t* tuch_contact_model_faces(const tuch_contact_model* m)
{
    return m ? m->faces : nullptr;
}

extern "C" int tuch_contact_model_tree_order(const tuch_contact_model* m, int32_t* qperm_host, int32_t* face_leaf_host)
{
    TUCH_REQUIRE(m, "tuch_contact_model_tree_order: null model");
    TUCH_REQUIRE(m->tree_nodes > 0 && m->tree_qperm_host && m->tree_face_leaf_host,
                 "tuch_contact_model_tree_order: the model has no cluster tree");
    if (qperm_host) memcpy(qperm_host, m->tree_qperm_host, sizeof(int32_t) * m->V);
    if (face_leaf_host) memcpy(face_leaf_host, m->tree_face_leaf_host, sizeof(int32_t) * m->F);
    return TUCH_OK;
}

extern "C" int tuch_contact_model_strips(const tuch_contact_model* m, int* stream_len, int* num_strips,
                                         int32_t* vidx_host, float* sign_host)
{
    TUCH_REQUIRE(m, "tuch_contact_model_strips: null model");
    if (stream_len) *stream_len = m->strip_len;
    if (num_strips) *num_strips = m->num_strips;
    if (vidx_host && tuch_table_download(vidx_host, m->strip_vidx, sizeof(int32_t) * m->strip_len) != TUCH_OK)
        return TUCH_ERR_HIP;
    if (sign_host && tuch_table_download(sign_host, m->strip_sign, sizeof(float) * m->strip_len) != TUCH_OK)
        return TUCH_ERR_HIP;
    return TUCH_OK;
}

extern "C" int tuch_contact_model_info(const tuch_contact_model* m, int* V, int* F, int* num_segments,
                                       int* seg_q_total, int* num_pairs)
{
    TUCH_REQUIRE(m, "tuch_contact_model_info: null model");
    if (V) *V = m->V;
    if (F) *F = m->F;
    if (num_segments) *num_segments = m->num_segments;
    if (seg_q_total) *seg_q_total = m->seg_q_total;
    if (num_pairs) *num_pairs = m->num_pairs;
    return TUCH_OK;
}
