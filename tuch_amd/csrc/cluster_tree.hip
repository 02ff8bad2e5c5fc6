// Cluster tree over the faces of a closed triangle mesh (host code, built once per model).
//
// Why: the winding number of tuch/utils/contact.py:112-147 sums the solid angle of EVERY face
// for every query.  The solid angle of a surface patch depends only on its boundary curve as
// long as the query is outside the volume between two surfaces with that boundary: if P is a set
// of faces and T any triangulation of P's boundary loops (same directed edges), P - T is a
// closed surface and sum_P omega - sum_T omega = 4 pi * (winding of P - T around q) = 0 for every
// q outside it.  P - T lies in the convex hull of P's vertices, so "q outside the posed
// bounding box of P" is a sufficient test.  A patch of K faces has O(sqrt K) boundary edges.
// (This is the hierarchical evaluation of generalized winding numbers, Jacobson et al. 2013,
// restated for a deforming mesh: the tree and the cap triangulations are topological and are
// built once; only the boxes depend on the pose.)  The substitution is exact in real
// arithmetic; in float32 it changes rounding only.
//
// Construction (topology only, no geometry, deterministic):
//   1. leaves  = Voronoi cells of farthest-point-sampled seed faces under hop distance on the
//                face adjacency graph (~leaf_faces faces each), ears removed;
//   2. tree    = repeated pairwise merging of adjacent clusters (longest shared border first);
//   3. per node: the boundary loops, each triangulated as ONE zig-zag strip
//                r0 r1 r(n-1) r2 r(n-2) ... (n stream elements for n-2 triangles);
//      per leaf: its faces as greedy triangle strips;
//   4. nodes are numbered in preorder (first child = node + 1, `skip` = first node after the
//      subtree) so that the device walks the tree without a stack;
//   5. query order: vertices sorted by the leaf they touch, cut into blocks of 128, so that one
//      wavefront's queries are neighbours on the surface.
#include "model.h"
#include <algorithm>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <unordered_map>
#include <vector>

namespace {

inline uint64_t key2(int a, int b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; }

// adj[3 f + k] = face across the edge (faces[f][k], faces[f][k+1]); false unless every directed
// edge occurs once and has its reverse (closed, consistently oriented 2-manifold)
bool face_adjacency(const int32_t* faces, int V, int F, std::vector<int>& adj)
{
    std::unordered_map<uint64_t, int> dir;
    dir.reserve((size_t)F * 3 * 2);
    for (int f = 0; f < F; ++f)
        for (int k = 0; k < 3; ++k) {
            const int a = faces[3 * f + k], b = faces[3 * f + (k + 1) % 3];
            if (a == b || a < 0 || b < 0 || a >= V || b >= V) return false;
            if (!dir.emplace(key2(a, b), f).second) return false;
        }
    adj.assign((size_t)F * 3, -1);
    for (int f = 0; f < F; ++f)
        for (int k = 0; k < 3; ++k) {
            const int a = faces[3 * f + k], b = faces[3 * f + (k + 1) % 3];
            auto it = dir.find(key2(b, a));
            if (it == dir.end() || it->second == f) return false;
            adj[3 * f + k] = it->second;
        }
    return true;
}

constexpr int kFar = 1 << 29;

// multi-source breadth-first search over the face graph
void bfs(const std::vector<int>& adj, int F, const std::vector<int>& sources, std::vector<int>& dist,
         std::vector<int>* owner)
{
    dist.assign(F, kFar);
    if (owner) owner->assign(F, -1);
    std::vector<int> queue;
    queue.reserve(F);
    for (size_t i = 0; i < sources.size(); ++i) {
        if (dist[sources[i]] == 0) continue;
        dist[sources[i]] = 0;
        if (owner) (*owner)[sources[i]] = (int)i;
        queue.push_back(sources[i]);
    }
    for (size_t h = 0; h < queue.size(); ++h) {
        const int f = queue[h];
        for (int k = 0; k < 3; ++k) {
            const int g = adj[3 * f + k];
            if (dist[g] == kFar) {
                dist[g] = dist[f] + 1;
                if (owner) (*owner)[g] = (*owner)[f];
                queue.push_back(g);
            }
        }
    }
}

int argmax(const std::vector<int>& v)
{
    int best = 0;
    for (size_t i = 1; i < v.size(); ++i)
        if (v[i] > v[best]) best = (int)i;
    return best;
}

// leaves: label[f] in [0, count)
int voronoi_leaves(const std::vector<int>& adj, int F, int k, std::vector<int>& label)
{
    std::vector<int> dist, dmin, seeds;
    bfs(adj, F, {0}, dist, nullptr);
    for (int& d : dist)
        if (d == kFar) d = -1;                         // start inside face 0's component
    seeds.push_back(argmax(dist));
    bfs(adj, F, seeds, dmin, nullptr);
    while ((int)seeds.size() < k || dmin[argmax(dmin)] == kFar) {
        const int s = argmax(dmin);
        if (dmin[s] == 0) break;
        seeds.push_back(s);
        bfs(adj, F, {s}, dist, nullptr);
        for (int f = 0; f < F; ++f) dmin[f] = std::min(dmin[f], dist[f]);
    }
    bfs(adj, F, seeds, dist, &label);
    // ear removal: a face with two neighbours in the same other cluster moves there
    for (int it = 0; it < 32; ++it) {
        int moved = 0;
        for (int f = 0; f < F; ++f) {
            const int a = label[adj[3 * f]], b = label[adj[3 * f + 1]], c = label[adj[3 * f + 2]];
            const int own = label[f];
            int to = -1;
            if (a != own && (a == b || a == c)) to = a;
            else if (b != own && b == c) to = b;
            if (to >= 0) { label[f] = to; ++moved; }
        }
        if (!moved) break;
    }
    std::vector<int> remap(seeds.size(), -1);
    int count = 0;
    for (int f = 0; f < F; ++f) {
        if (remap[label[f]] < 0) remap[label[f]] = count++;
        label[f] = remap[label[f]];
    }
    return count;
}

struct Group {
    std::vector<int> faces;
    int c0 = -1, c1 = -1;
};

// boundary loops of a face set as simple cycles of vertices, directed as in the owning faces
bool boundary_loops(const int32_t* faces, const std::vector<int>& adj, const std::vector<int>& fs,
                    std::vector<char>& inset, std::vector<std::vector<int>>& loops)
{
    loops.clear();
    for (int f : fs) inset[f] = 1;
    std::vector<std::pair<int, int>> edges;
    for (int f : fs)
        for (int k = 0; k < 3; ++k)
            if (!inset[adj[3 * f + k]]) edges.emplace_back(faces[3 * f + k], faces[3 * f + (k + 1) % 3]);
    for (int f : fs) inset[f] = 0;
    std::unordered_map<int, std::vector<int>> out;        // vertex -> unused outgoing edges
    for (int e = (int)edges.size() - 1; e >= 0; --e) out[edges[e].first].push_back(e);
    std::vector<char> used(edges.size(), 0);
    std::unordered_map<int, int> pos;
    std::vector<int> path;
    for (size_t e0 = 0; e0 < edges.size(); ++e0) {
        if (used[e0]) continue;
        path.clear();
        pos.clear();
        path.push_back(edges[e0].first);
        pos[edges[e0].first] = 0;
        // consume e0 first
        auto& o0 = out[edges[e0].first];
        o0.erase(std::find(o0.begin(), o0.end(), (int)e0));
        used[e0] = 1;
        int v = edges[e0].second;
        for (;;) {
            auto it = pos.find(v);
            if (it != pos.end()) {
                const int j = it->second;
                loops.emplace_back(path.begin() + j, path.end());
                for (size_t i = j + 1; i < path.size(); ++i) pos.erase(path[i]);
                path.resize(j + 1);
            } else {
                pos[v] = (int)path.size();
                path.push_back(v);
            }
            auto& o = out[path.back()];
            if (o.empty()) break;
            const int e = o.back();
            o.pop_back();
            used[e] = 1;
            v = edges[e].second;
        }
        if (path.size() != 1) return false;                 // unbalanced boundary: not a closed manifold
    }
    for (auto& l : loops)
        if (l.size() < 3) return false;
    return true;
}

void pad3(std::vector<int32_t>& vidx, std::vector<float>& sign, size_t begin)
{
    while ((vidx.size() - begin) % 3 != 0) {
        vidx.push_back(vidx.back());
        sign.push_back(0.0f);
    }
}

}  // namespace

// Greedy triangle strips.  Every face appears exactly once as an emitted element; the sign
// says whether (stream[p-2], stream[p-1], stream[p]) is an even permutation of the face.
// Seeds are the faces with the fewest unused neighbours (strip ends first); from a seed the strip
// is grown in both directions, best of the three rotations.
void tuch_build_strips(const int32_t* faces, int F, std::vector<int32_t>& vidx, std::vector<float>& sign,
                       int* num_strips)
{
    std::unordered_map<uint64_t, int> edge_face;      // directed edge (a -> b) -> face
    edge_face.reserve((size_t)F * 3 * 2);
    for (int f = 0; f < F; ++f)
        for (int k = 0; k < 3; ++k) edge_face[key2(faces[3 * f + k], faces[3 * f + (k + 1) % 3])] = f;
    std::vector<char> used(F, 0), mark(F, 0);
    auto third = [&](int f, int a, int b) {            // vertex of face f that is neither a nor b
        for (int k = 0; k < 3; ++k) {
            const int v = faces[3 * f + k];
            if (v != a && v != b) return v;
        }
        return -1;
    };
    auto across = [&](int f, int y, int z) {           // the other face on the edge {y, z}
        auto it = edge_face.find(key2(y, z));
        if (it != edge_face.end() && it->second != f) return it->second;
        it = edge_face.find(key2(z, y));
        if (it != edge_face.end() && it->second != f) return it->second;
        return -1;
    };
    // grow from face f entered with the vertex order (a, b, c); faces taken are marked
    auto walk = [&](int f, int a, int b, int c, std::vector<int>& seq, std::vector<int>& fseq) {
        seq.assign({a, b, c});
        fseq.assign(1, f);
        mark[f] = 1;
        int cur = f;
        for (;;) {
            const int n = (int)seq.size();
            const int y = seq[n - 2], z = seq[n - 1];
            const int nf = across(cur, y, z);
            if (nf < 0 || used[nf] || mark[nf]) break;
            const int d = third(nf, y, z);
            if (d < 0) break;
            seq.push_back(d);
            fseq.push_back(nf);
            mark[nf] = 1;
            cur = nf;
        }
    };
    auto free_neighbours = [&](int f) {
        int n = 0;
        for (int k = 0; k < 3; ++k) {
            const int g = across(f, faces[3 * f + k], faces[3 * f + (k + 1) % 3]);
            n += (g >= 0 && !used[g]);
        }
        return n;
    };
    std::vector<int> fwd, ffwd, bwd, fbwd, seq, fseq, best_seq, best_f;
    if (num_strips) *num_strips = 0;
    int remaining = F;
    while (remaining > 0) {
        int seed = -1, seed_free = 4;
        for (int f = 0; f < F && seed_free > 0; ++f) {
            if (used[f]) continue;
            const int n = free_neighbours(f);
            if (n < seed_free) { seed = f; seed_free = n; }
        }
        best_seq.clear();
        for (int r = 0; r < 3; ++r) {
            const int a = faces[3 * seed + r], b = faces[3 * seed + (r + 1) % 3], c = faces[3 * seed + (r + 2) % 3];
            walk(seed, a, b, c, fwd, ffwd);
            walk(seed, c, b, a, bwd, fbwd);            // the other direction, around the faces just taken
            seq.assign(bwd.rbegin(), bwd.rend() - 3);
            seq.insert(seq.end(), fwd.begin(), fwd.end());
            fseq.assign(fbwd.rbegin(), fbwd.rend() - 1);
            fseq.insert(fseq.end(), ffwd.begin(), ffwd.end());
            for (int ff : ffwd) mark[ff] = 0;
            for (int ff : fbwd) mark[ff] = 0;
            if (seq.size() > best_seq.size()) { best_seq = seq; best_f = fseq; }
        }
        for (size_t i = 0; i < best_seq.size(); ++i) {
            vidx.push_back(best_seq[i]);
            if (i < 2) { sign.push_back(0.0f); continue; }
            const int ff = best_f[i - 2];
            used[ff] = 1;
            --remaining;
            // parity of (s[i-2], s[i-1], s[i]) relative to the face's own order
            const int a = best_seq[i - 2], b = best_seq[i - 1];
            int ia = -1, ib = -1;
            for (int k = 0; k < 3; ++k) {
                if (faces[3 * ff + k] == a) ia = k;
                if (faces[3 * ff + k] == b) ib = k;
            }
            sign.push_back(((ia + 1) % 3 == ib) ? 1.0f : -1.0f);
        }
        if (num_strips) ++*num_strips;
    }
}

bool tuch_cluster_tree_build_impl(int V, int F, const int32_t* faces, int leaf_faces, tuch_cluster_tree& t)
{
    t = tuch_cluster_tree();
    t.V = V;
    t.F = F;
    if (leaf_faces < 8) leaf_faces = 8;
    std::vector<int> adj;
    if (F < 4 || !face_adjacency(faces, V, F, adj)) return false;

    // ---- leaves
    std::vector<int> label;
    const int want = std::max(1, (int)lround((double)F / leaf_faces));
    const int nleaf = voronoi_leaves(adj, F, want, label);
    std::vector<Group> groups(nleaf);
    for (int f = 0; f < F; ++f) groups[label[f]].faces.push_back(f);

    // ---- merge tree
    std::vector<int> cur(nleaf), lab(label);
    for (int i = 0; i < nleaf; ++i) cur[i] = i;
    while (cur.size() > 1) {
        std::unordered_map<uint64_t, int> w;
        for (int f = 0; f < F; ++f)
            for (int k = 0; k < 3; ++k) {
                const int g = adj[3 * f + k];
                if (lab[f] != lab[g]) ++w[key2(lab[f], lab[g])];
            }
        std::unordered_map<int, std::vector<std::pair<int, int>>> nbr;
        for (auto& kv : w) nbr[(int)(kv.first >> 32)].emplace_back((int)(uint32_t)kv.first, kv.second);
        std::vector<int> order(cur);
        std::sort(order.begin(), order.end(), [&](int a, int b) {
            const size_t sa = groups[a].faces.size(), sb = groups[b].faces.size();
            return sa != sb ? sa < sb : a < b;
        });
        std::unordered_map<int, int> match;
        for (int c : order) {
            if (match.count(c)) continue;
            int best = -1;
            double best_score = -1.0;
            auto it = nbr.find(c);
            if (it != nbr.end())
                for (auto& nw : it->second) {
                    if (match.count(nw.first)) continue;
                    const double score = nw.second / sqrt((double)groups[nw.first].faces.size());
                    if (score > best_score || (score == best_score && nw.first < best)) {
                        best_score = score;
                        best = nw.first;
                    }
                }
            if (best >= 0) { match[c] = best; match[best] = c; }
        }
        if (match.empty()) {                           // disconnected components: join in order
            for (size_t i = 0; i + 1 < cur.size(); i += 2) { match[cur[i]] = cur[i + 1]; match[cur[i + 1]] = cur[i]; }
        }
        std::vector<int> next;
        std::unordered_map<int, char> done;
        for (int c : cur) {
            if (done.count(c)) continue;
            auto it = match.find(c);
            if (it == match.end()) { next.push_back(c); continue; }
            const int d = it->second;
            done[c] = 1; done[d] = 1;
            Group g;
            g.c0 = c; g.c1 = d;
            g.faces = groups[c].faces;
            g.faces.insert(g.faces.end(), groups[d].faces.begin(), groups[d].faces.end());
            const int id = (int)groups.size();
            for (int f : g.faces) lab[f] = id;
            groups.push_back(std::move(g));
            next.push_back(id);
        }
        cur.swap(next);
    }
    const int root = cur[0];
    const int N = (int)groups.size();

    // ---- preorder numbering
    std::vector<int> pre(N, -1), order;           // order[preorder index] = group id
    std::vector<int> skip(N, 0), height(N, 0);
    {
        std::vector<std::pair<int, int>> stack;   // (group, state)
        stack.emplace_back(root, 0);
        while (!stack.empty()) {
            auto& top = stack.back();
            const int g = top.first;
            if (top.second == 0) {
                pre[g] = (int)order.size();
                order.push_back(g);
                top.second = 1;
                if (groups[g].c0 >= 0) stack.emplace_back(groups[g].c0, 0);
            } else if (top.second == 1) {
                top.second = 2;
                if (groups[g].c1 >= 0) stack.emplace_back(groups[g].c1, 0);
            } else {
                skip[g] = (int)order.size();
                if (groups[g].c0 >= 0) height[g] = 1 + std::max(height[groups[g].c0], height[groups[g].c1]);
                stack.pop_back();
            }
        }
    }

    // ---- exact streams of the leaves (in preorder), then the cap streams of all nodes
    t.num_nodes = N;
    t.face_leaf.assign(F, 0);
    t.nodes.assign((size_t)N * 8, 0);
    std::vector<int32_t> sub;
    std::vector<int> leaf_seq(N, -1);
    int nleaves_seen = 0;
    for (int i = 0; i < N; ++i) {
        const Group& g = groups[order[i]];
        int32_t* nd = &t.nodes[(size_t)i * 8];
        nd[4] = skip[order[i]];
        nd[5] = g.c0 >= 0 ? pre[g.c0] : -1;
        nd[6] = g.c1 >= 0 ? pre[g.c1] : -1;
        nd[7] = (int)g.faces.size();
        if (g.c0 >= 0) continue;
        leaf_seq[i] = nleaves_seen++;
        for (int f : g.faces) t.face_leaf[f] = leaf_seq[i];
        sub.clear();
        for (int f : g.faces) { sub.push_back(faces[3 * f]); sub.push_back(faces[3 * f + 1]); sub.push_back(faces[3 * f + 2]); }
        const size_t begin = t.vidx.size();
        tuch_build_strips(sub.data(), (int)g.faces.size(), t.vidx, t.sign, nullptr);
        pad3(t.vidx, t.sign, begin);
        nd[2] = (int)begin;
        nd[3] = (int)(t.vidx.size() - begin);
    }
    t.exact_len = (int)t.vidx.size();
    std::vector<char> inset(F, 0);
    std::vector<std::vector<int>> loops;
    for (int i = 0; i < N; ++i) {
        const Group& g = groups[order[i]];
        if (!boundary_loops(faces, adj, g.faces, inset, loops)) return false;
        const size_t begin = t.vidx.size();
        for (auto& r : loops) {
            const int n = (int)r.size();
            int lo = 1, hi = n - 1;
            for (int k = 0; k < n; ++k) {
                const int v = k == 0 ? r[0] : ((k & 1) ? r[lo++] : r[hi--]);
                t.vidx.push_back(v);
                t.sign.push_back(k < 2 ? 0.0f : ((k & 1) ? -1.0f : 1.0f));
            }
        }
        if (t.vidx.size() > begin) pad3(t.vidx, t.sign, begin);
        t.nodes[(size_t)i * 8 + 0] = (int)begin;
        t.nodes[(size_t)i * 8 + 1] = (int)(t.vidx.size() - begin);
    }
    t.stream_len = (int)t.vidx.size();

    // ---- nodes by height (bottom-up bounding boxes)
    int H = 0;
    for (int g = 0; g < N; ++g) H = std::max(H, height[g]);
    t.num_heights = H + 1;
    t.height_off.assign(H + 2, 0);
    for (int i = 0; i < N; ++i) ++t.height_off[height[order[i]] + 1];
    for (int h = 0; h <= H; ++h) t.height_off[h + 1] += t.height_off[h];
    t.height_nodes.resize(N);
    {
        std::vector<int> fill(t.height_off.begin(), t.height_off.end() - 1);
        for (int i = 0; i < N; ++i) t.height_nodes[fill[height[order[i]]]++] = i;
    }

    // ---- query order
    std::vector<int> vkey(V, 1 << 30), seq_node(nleaves_seen, 0);
    for (int i = 0; i < N; ++i) {
        if (leaf_seq[i] < 0) continue;
        seq_node[leaf_seq[i]] = i;
        for (int f : groups[order[i]].faces)
            for (int k = 0; k < 3; ++k) vkey[faces[3 * f + k]] = std::min(vkey[faces[3 * f + k]], leaf_seq[i]);
    }
    std::vector<int32_t> vorder(V);
    for (int v = 0; v < V; ++v) vorder[v] = v;
    std::stable_sort(vorder.begin(), vorder.end(), [&](int a, int b) { return vkey[a] < vkey[b]; });
    t.num_qblocks = (V + 127) / 128;
    t.qperm.assign((size_t)t.num_qblocks * 128, vorder[V - 1]);
    std::copy(vorder.begin(), vorder.end(), t.qperm.begin());
    // the vertices keyed to a leaf are contiguous in this order: rows[leaf] = (first position, count)
    t.rows.assign((size_t)N * 2, 0);
    for (int pos = 0; pos < V; ++pos) {
        const int nd = seq_node[vkey[vorder[pos]]];
        if (t.rows[(size_t)nd * 2 + 1]++ == 0) t.rows[(size_t)nd * 2] = pos;
    }
    for (int i = N - 1; i >= 0; --i) {               // inner nodes: the span of their leaves
        const int c0 = t.nodes[(size_t)i * 8 + 5], c1 = t.nodes[(size_t)i * 8 + 6];
        if (c0 < 0) continue;
        int lo = 1 << 30, hi = 0;
        for (int c : {c0, c1})
            if (t.rows[(size_t)c * 2 + 1] > 0) {
                lo = std::min(lo, t.rows[(size_t)c * 2]);
                hi = std::max(hi, t.rows[(size_t)c * 2] + t.rows[(size_t)c * 2 + 1]);
            }
        if (hi > 0) { t.rows[(size_t)i * 2] = lo; t.rows[(size_t)i * 2 + 1] = hi - lo; }
    }

    // ---- frontiers: sets of subtrees that together cover the mesh, one workgroup column each
    t.frontier_off.push_back(0);
    for (int target : {1, 2, 4, 8, 16, 32, 64}) {
        std::vector<int> fr{0};
        while ((int)fr.size() < target) {
            int best = -1;
            for (size_t i = 0; i < fr.size(); ++i)
                if (t.nodes[(size_t)fr[i] * 8 + 5] >= 0 &&
                    (best < 0 || t.nodes[(size_t)fr[i] * 8 + 7] > t.nodes[(size_t)fr[best] * 8 + 7])) best = (int)i;
            if (best < 0) break;
            const int n = fr[best];
            fr[best] = t.nodes[(size_t)n * 8 + 5];
            fr.push_back(t.nodes[(size_t)n * 8 + 6]);
        }
        std::sort(fr.begin(), fr.end());
        // launch order of the (subtree, query block) pairs: pairs whose queries live inside the subtree
        // walk it down to the leaves (long-running) and go first, far pairs (one cap) fill the tail
        std::vector<std::pair<long, int32_t>> cost;
        for (size_t si = 0; si < fr.size(); ++si) {
            const int lo = fr[si], hi = t.nodes[(size_t)fr[si] * 8 + 4];
            for (int qb = 0; qb < t.num_qblocks; ++qb) {
                long home = 0;
                for (int k = 0; k < 128; ++k) {
                    const int ln = seq_node[vkey[t.qperm[(size_t)qb * 128 + k]]];
                    home += (ln >= lo && ln < hi);
                }
                cost.emplace_back(-(home * 100000L + t.nodes[(size_t)fr[si] * 8 + 7]), (int32_t)((si << 16) | qb));
            }
        }
        std::stable_sort(cost.begin(), cost.end(), [](const std::pair<long, int32_t>& a, const std::pair<long, int32_t>& b) {
            return a.first < b.first;
        });
        for (auto& c : cost) t.launch_order.push_back(c.second);
        // the first 8 ancestors (root first) of every frontier node; preorder: a is an ancestor of n iff a < n < skip[a]
        for (size_t si = 0; si < fr.size(); ++si) {
            int count = 0;
            for (int a = 0; a < fr[si] && count < 8; ++a)
                if (t.nodes[(size_t)a * 8 + 4] > fr[si]) { t.ancestors.push_back(a); ++count; }
            for (; count < 8; ++count) t.ancestors.push_back(-1);
        }
        t.frontier_nodes.insert(t.frontier_nodes.end(), fr.begin(), fr.end());
        t.frontier_off.push_back((int)t.frontier_nodes.size());
    }
    return true;
}

// ---- host-only C entry points (tests and tools; no device needed) ---------------------------
extern "C" int tuch_cluster_tree_build(int V, int F, const int32_t* faces, int leaf_faces, tuch_cluster_tree** out)
{
    TUCH_REQUIRE(out && faces && V > 0 && F > 0, "tuch_cluster_tree_build: bad arguments");
    tuch_cluster_tree* t = new tuch_cluster_tree();
    if (!tuch_cluster_tree_build_impl(V, F, faces, leaf_faces, *t)) {
        delete t;
        *out = nullptr;
        tuch_set_error("tuch_cluster_tree_build: the mesh is not a closed, consistently oriented manifold");
        return TUCH_ERR_ARG;
    }
    *out = t;
    return TUCH_OK;
}

extern "C" void tuch_cluster_tree_free(tuch_cluster_tree* t) { delete t; }

extern "C" int tuch_cluster_tree_info(const tuch_cluster_tree* t, int* num_nodes, int* exact_len, int* stream_len,
                                      int* num_qblocks, int* num_frontiers, int* frontier_total)
{
    TUCH_REQUIRE(t, "tuch_cluster_tree_info: null tree");
    if (num_nodes) *num_nodes = t->num_nodes;
    if (exact_len) *exact_len = t->exact_len;
    if (stream_len) *stream_len = t->stream_len;
    if (num_qblocks) *num_qblocks = t->num_qblocks;
    if (num_frontiers) *num_frontiers = (int)t->frontier_off.size() - 1;
    if (frontier_total) *frontier_total = (int)t->frontier_nodes.size();
    return TUCH_OK;
}

extern "C" int tuch_cluster_tree_export(const tuch_cluster_tree* t, int32_t* nodes, int32_t* vidx, float* sign,
                                        int32_t* qperm, int32_t* frontier_off, int32_t* frontier_nodes,
                                        int32_t* launch_order, int32_t* rows, int32_t* face_leaf)
{
    TUCH_REQUIRE(t, "tuch_cluster_tree_export: null tree");
    if (nodes) memcpy(nodes, t->nodes.data(), t->nodes.size() * sizeof(int32_t));
    if (vidx) memcpy(vidx, t->vidx.data(), t->vidx.size() * sizeof(int32_t));
    if (sign) memcpy(sign, t->sign.data(), t->sign.size() * sizeof(float));
    if (qperm) memcpy(qperm, t->qperm.data(), t->qperm.size() * sizeof(int32_t));
    if (frontier_off) memcpy(frontier_off, t->frontier_off.data(), t->frontier_off.size() * sizeof(int32_t));
    if (frontier_nodes) memcpy(frontier_nodes, t->frontier_nodes.data(), t->frontier_nodes.size() * sizeof(int32_t));
    if (launch_order) memcpy(launch_order, t->launch_order.data(), t->launch_order.size() * sizeof(int32_t));
    if (rows) memcpy(rows, t->rows.data(), t->rows.size() * sizeof(int32_t));
    if (face_leaf) memcpy(face_leaf, t->face_leaf.data(), t->face_leaf.size() * sizeof(int32_t));
    return TUCH_OK;
}
