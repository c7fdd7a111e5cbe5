// Host-side preparation of the robot geometry: the meshlet partition the rasteriser walks, built from the welded link
// meshes (what MeshLoader.load hands pyrender in the reference, robotpose/simulation/render_utils.py:22-41).
//
// rope_partition_mesh   one link's triangles -> patches of <= max_tris triangles over <= max_verts vertices, grown over
//                       the triangle adjacency graph from Morton-ordered seeds (compact surface patches: small screen
//                       boxes, few tiles straddled); needs no GPU and no context
// rope_set_robot_mesh   all links: partition, meshlet headers / vertex copies / packed indices, then rope_set_robot —
//                       so that a host that is not Python needs nothing but the vertex and index arrays
//
// Any partition renders the same image (the depth test is a minimum, the sums are exact), so this is a performance
// structure only; the Python host uses the same routine (rope_s3d_amd/robot.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/rope_s3d.h"

void rope_set_error(rope_ctx *c, const std::string &msg);       // rope_abi.hip

namespace {

uint64_t spread10(uint64_t v)
{
    v &= 0x3FF;
    v = (v | (v << 16)) & 0x30000FF;
    v = (v | (v << 8)) & 0x300F00F;
    v = (v | (v << 4)) & 0x30C30C3;
    v = (v | (v << 2)) & 0x9249249;
    return v;
}

using Item = std::pair<double, int>;                            // (squared distance to the patch seed, triangle)
using MinHeap = std::priority_queue<Item, std::vector<Item>, std::greater<Item>>;

struct Partition {
    std::vector<std::vector<int>> clusters;
};

// Region growing: take the nearest (to the seed's centroid) unvisited neighbour that still fits the vertex budget;
// what is left on the frontier seeds the next patch, so patches tile the surface outward instead of leaving slivers.
// Growth stops `soft_margin` short of the vertex limit; small islands stranded between finished patches are then
// folded into the neighbouring patch they share most vertices with, up to the hard limits.
Partition grow(const float *V, int n_verts, const int32_t *F, int T, int hard_t, int hard_v, int soft_margin = 4, int absorb_below = 24)
{
    const int max_t = hard_t, max_v = std::max(hard_v - soft_margin, 3);
    std::vector<double> cent((size_t)T * 3);
    for (int t = 0; t < T; t++)
        for (int k = 0; k < 3; k++)
            cent[3 * (size_t)t + k] = (((double)V[3 * (size_t)F[3 * t] + k] + (double)V[3 * (size_t)F[3 * t + 1] + k]) + (double)V[3 * (size_t)F[3 * t + 2] + k]) / 3.0;
    // adjacency over shared edges
    struct Edge { int64_t key; int tri; };
    std::vector<Edge> edges((size_t)T * 3);
    for (int t = 0; t < T; t++)
        for (int e = 0; e < 3; e++) {
            const int64_t a = F[3 * t + e], b = F[3 * t + (e + 1) % 3];
            edges[(size_t)e * T + t] = {std::min(a, b) * ((int64_t)n_verts + 1) + std::max(a, b), t};
        }
    std::stable_sort(edges.begin(), edges.end(), [](const Edge &x, const Edge &y) { return x.key < y.key; });
    std::vector<std::vector<int>> nbr(T);
    for (size_t i = 0; i + 1 < edges.size(); i++)
        if (edges[i].key == edges[i + 1].key) {
            nbr[edges[i].tri].push_back(edges[i + 1].tri);
            nbr[edges[i + 1].tri].push_back(edges[i].tri);
        }
    // seeds in Morton order of the quantised centroids
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int t = 0; t < T; t++)
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], cent[3 * (size_t)t + k]);
            hi[k] = std::max(hi[k], cent[3 * (size_t)t + k]);
        }
    std::vector<std::pair<uint64_t, int>> seeds(T);
    for (int t = 0; t < T; t++) {
        uint64_t code = 0;
        for (int k = 0; k < 3; k++) {
            const double span = std::max(hi[k] - lo[k], 1e-9);
            const int64_t q = std::min<int64_t>((int64_t)((cent[3 * (size_t)t + k] - lo[k]) / span * 1023), 1023);
            code |= spread10((uint64_t)q) << k;
        }
        seeds[t] = {code, t};
    }
    std::stable_sort(seeds.begin(), seeds.end(), [](const auto &x, const auto &y) { return x.first < y.first; });

    Partition out;
    std::vector<char> visited(T, 0);
    std::vector<int> inheap_stamp(T, -1);
    std::vector<Item> carry;
    int sp = 0;
    for (;;) {
        int seed = -1;
        {
            MinHeap ch(std::greater<Item>(), std::move(carry));
            carry.clear();
            while (!ch.empty()) {
                const int t = ch.top().second;
                ch.pop();
                if (!visited[t]) { seed = t; break; }
            }
        }
        if (seed < 0) {
            while (sp < T && visited[seeds[sp].second]) sp++;
            if (sp >= T) break;
            seed = seeds[sp].second;
        }
        const int stamp = (int)out.clusters.size();
        const double *c0 = &cent[3 * (size_t)seed];
        std::vector<int> tris, verts;                             // verts: sorted unique vertex ids of the patch
        MinHeap heap;
        heap.push({0.0, seed});
        inheap_stamp[seed] = stamp;
        while (!heap.empty() && (int)tris.size() < max_t) {
            const int t = heap.top().second;
            heap.pop();
            if (visited[t]) continue;
            int fresh[3], nf = 0;
            for (int e = 0; e < 3; e++) {
                const int v = F[3 * t + e];
                bool dup = std::binary_search(verts.begin(), verts.end(), v);
                for (int k = 0; k < nf; k++) dup = dup || fresh[k] == v;
                if (!dup) fresh[nf++] = v;
            }
            if ((int)verts.size() + nf > max_v) continue;
            for (int k = 0; k < nf; k++) verts.insert(std::lower_bound(verts.begin(), verts.end(), fresh[k]), fresh[k]);
            tris.push_back(t);
            visited[t] = 1;
            for (int n : nbr[t])
                if (!visited[n] && inheap_stamp[n] != stamp) {
                    inheap_stamp[n] = stamp;
                    const double dx = cent[3 * (size_t)n] - c0[0], dy = cent[3 * (size_t)n + 1] - c0[1], dz = cent[3 * (size_t)n + 2] - c0[2];
                    heap.push({(dx * dx + dy * dy) + dz * dz, n});
                }
        }
        while (!heap.empty()) {
            if (!visited[heap.top().second]) carry.push_back(heap.top());
            heap.pop();
        }
        out.clusters.push_back(std::move(tris));
    }

    // fold small islands into a neighbour
    std::vector<int> owner(T);
    std::vector<std::vector<int>> vsets(out.clusters.size());
    for (size_t ci = 0; ci < out.clusters.size(); ci++) {
        for (int t : out.clusters[ci]) {
            owner[t] = (int)ci;
            for (int e = 0; e < 3; e++) vsets[ci].push_back(F[3 * t + e]);
        }
        std::sort(vsets[ci].begin(), vsets[ci].end());
        vsets[ci].erase(std::unique(vsets[ci].begin(), vsets[ci].end()), vsets[ci].end());
    }
    std::vector<char> alive(out.clusters.size(), 1);
    std::vector<int> by_size(out.clusters.size());
    for (size_t i = 0; i < by_size.size(); i++) by_size[i] = (int)i;
    std::stable_sort(by_size.begin(), by_size.end(), [&](int a, int b) { return out.clusters[a].size() < out.clusters[b].size(); });
    for (int ci : by_size) {
        if ((int)out.clusters[ci].size() >= absorb_below) break;
        std::vector<int> cands;
        for (int t : out.clusters[ci])
            for (int n : nbr[t])
                if (owner[n] != ci) cands.push_back(owner[n]);
        std::sort(cands.begin(), cands.end());
        cands.erase(std::unique(cands.begin(), cands.end()), cands.end());
        int best = -1;
        size_t best_shared = 0;
        for (int cj : cands) {
            if (!alive[cj] || (int)(out.clusters[cj].size() + out.clusters[ci].size()) > hard_t) continue;
            std::vector<int> common;
            std::set_intersection(vsets[ci].begin(), vsets[ci].end(), vsets[cj].begin(), vsets[cj].end(), std::back_inserter(common));
            if ((int)(vsets[ci].size() + vsets[cj].size() - common.size()) <= hard_v && common.size() > best_shared) {
                best = cj;
                best_shared = common.size();
            }
        }
        if (best >= 0) {
            for (int t : out.clusters[ci]) owner[t] = best;
            out.clusters[best].insert(out.clusters[best].end(), out.clusters[ci].begin(), out.clusters[ci].end());
            std::vector<int> merged;
            std::set_union(vsets[best].begin(), vsets[best].end(), vsets[ci].begin(), vsets[ci].end(), std::back_inserter(merged));
            vsets[best].swap(merged);
            alive[ci] = 0;
        }
    }
    Partition kept;
    for (size_t ci = 0; ci < out.clusters.size(); ci++)
        if (alive[ci]) kept.clusters.push_back(std::move(out.clusters[ci]));
    return kept;
}

bool mesh_ok(const float *verts, int n_verts, const int32_t *faces, int n_tris)
{
    if (!verts || !faces || n_verts < 3 || n_tris < 1) return false;
    for (size_t i = 0; i < (size_t)n_tris * 3; i++)
        if (faces[i] < 0 || faces[i] >= n_verts) return false;
    return true;
}

}  // namespace

extern "C" int rope_partition_mesh(const float *verts, int n_verts, const int32_t *faces, int n_tris, int max_tris, int max_verts,
                                   int32_t *tri_order, int32_t *meshlet_first)
{
    if (!tri_order || !meshlet_first || max_tris < 1 || max_tris > 128 || max_verts < 3 || max_verts > 64) return ROPE_E_ARG;
    if (!mesh_ok(verts, n_verts, faces, n_tris)) return ROPE_E_ARG;
    const Partition p = grow(verts, n_verts, faces, n_tris, max_tris, max_verts);
    int pos = 0, m = 0;
    for (const auto &c : p.clusters) {
        meshlet_first[m++] = pos;
        for (int t : c) tri_order[pos++] = t;
    }
    meshlet_first[m] = pos;
    return m;
}

extern "C" int rope_set_robot_mesh(rope_ctx *c, const float *verts, const int32_t *faces, const int32_t *vtx_off, const int32_t *tri_off,
                                   int n_links, const double *joint_fixed, const double *joint_axes)
{
    if (!c) return ROPE_E_ARG;
    auto fail = [&](const char *msg) { rope_set_error(c, msg); return (int)ROPE_E_ARG; };
    if (!verts || !faces || !vtx_off || !tri_off || !joint_fixed || !joint_axes) return fail("rope_set_robot_mesh: null pointer");
    if (n_links < 1 || n_links > ROPE_MAX_LINKS) return fail("rope_set_robot_mesh: n_links must be 1..6");
    std::vector<uint32_t> header, tris;
    std::vector<float> mverts;
    std::vector<int32_t> link_first{0};
    for (int l = 0; l < n_links; l++) {
        const int nv = vtx_off[l + 1] - vtx_off[l], nt = tri_off[l + 1] - tri_off[l];
        const float *V = verts + 3 * (size_t)vtx_off[l];
        const int32_t *F = faces + 3 * (size_t)tri_off[l];               // indices local to the link
        if (nv < 3 || nt < 1 || !mesh_ok(V, nv, F, nt)) return fail("rope_set_robot_mesh: a link's mesh is empty or indexes outside its vertices");
        const Partition p = grow(V, nv, F, nt, 128, 64);
        for (const auto &cl : p.clusters) {
            std::vector<int> uniq;
            for (int t : cl)
                for (int e = 0; e < 3; e++) uniq.push_back(F[3 * t + e]);
            std::sort(uniq.begin(), uniq.end());
            uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (int v : uniq)
                for (int k = 0; k < 3; k++) {
                    lo[k] = std::min(lo[k], (double)V[3 * (size_t)v + k]);
                    hi[k] = std::max(hi[k], (double)V[3 * (size_t)v + k]);
                }
            double ctr[3], r2 = 0.0;
            for (int k = 0; k < 3; k++) ctr[k] = (lo[k] + hi[k]) * 0.5;
            for (int v : uniq) {
                double d2 = 0.0;
                for (int k = 0; k < 3; k++) d2 += ((double)V[3 * (size_t)v + k] - ctr[k]) * ((double)V[3 * (size_t)v + k] - ctr[k]);
                r2 = std::max(r2, d2);
            }
            const float h4[4] = {(float)ctr[0], (float)ctr[1], (float)ctr[2], (float)(std::sqrt(r2) * (1.0 + 1e-5) + 1e-7)};
            uint32_t h[8];
            std::memcpy(h, h4, 16);
            h[4] = (uint32_t)(mverts.size() / 3);
            h[5] = (uint32_t)tris.size();
            h[6] = (uint32_t)uniq.size() | ((uint32_t)cl.size() << 16);
            h[7] = (uint32_t)l;
            header.insert(header.end(), h, h + 8);
            for (int v : uniq) mverts.insert(mverts.end(), V + 3 * (size_t)v, V + 3 * (size_t)v + 3);
            for (int t : cl) {
                uint32_t packed = 0;
                for (int e = 0; e < 3; e++)
                    packed |= (uint32_t)(std::lower_bound(uniq.begin(), uniq.end(), F[3 * t + e]) - uniq.begin()) << (8 * e);
                tris.push_back(packed);
            }
        }
        link_first.push_back((int32_t)(header.size() / 8));
    }
    return rope_set_robot(c, header.data(), (int)(header.size() / 8), mverts.data(), (int)(mverts.size() / 3), tris.data(), (int)tris.size(),
                          link_first.data(), n_links, joint_fixed, joint_axes);
}
