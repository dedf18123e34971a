// nv_graph_* / nv_nav_*: the HOST side-car of the navigation step behind the C ABI.
//
// What the reference does per nav step in Python dict/loop code, and what replaces it here:
//   models/graph_utils.py:47-96    FloydGraph (dict-of-dict all-pairs state, O(n^2) python relaxation per visit, recursive path)
//                                  -> dense double/int matrices, one tight relaxation loop, iterative path length
//   models/graph_utils.py:18-44,144-165   calculate_vp_rel_pos_fts / get_angle_fts / GraphMap.get_pos_fts (per node python)
//                                  -> nv_graph_pos_fts: all slots of a map in one call, written straight into the caller's
//                                     (pinned) float buffer
//   models/nav_model.py:174-190    which current-view candidate feeds which map slot (python loops over B x G x N strings)
//                                  -> nv_nav_match_tables on integer node ids
//   models/nav_model.py:216-223,234-242   candidate selection under the per-sample permutation + the inverse scatter of
//                                  the head's logits  -> nv_nav_perm_tables (the permutation itself stays torch.randperm on the
//                                     host: RNG parity with the reference)
// Nodes are small integers (the Python wrapper interns viewpoint-id strings once, when a node first appears).  Pure host
// code: no HIP calls, no allocation in the per-step entry points (the graph object grows geometrically when a node is added).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define NV_OK 0
#define NV_ERR_ARG (-1)

namespace {
const double kNoEdge = 95959595.0;   // the reference's "no edge" sentinel (graph_utils.py:49)
const double kMaxDist = 30.0, kMaxStep = 10.0;
}  // namespace

struct nv_graph {
    int n = 0, cap = 0;
    std::vector<double> dis;       // [cap, cap]
    std::vector<int> point;        // [cap, cap]; -1 = direct edge / none
    std::vector<uint8_t> visited;  // [cap]
    std::vector<double> pos;       // [cap, 3]
    std::vector<uint8_t> has_pos;

    void grow(int need) {
        if (need <= cap) return;
        int nc = cap ? cap : 16;
        while (nc < need) nc *= 2;
        std::vector<double> d((size_t)nc * nc, kNoEdge);
        std::vector<int> p((size_t)nc * nc, -1);
        for (int i = 0; i < n; ++i) {
            memcpy(&d[(size_t)i * nc], &dis[(size_t)i * cap], sizeof(double) * n);
            memcpy(&p[(size_t)i * nc], &point[(size_t)i * cap], sizeof(int) * n);
        }
        dis.swap(d);
        point.swap(p);
        visited.resize(nc, 0);
        pos.resize((size_t)nc * 3, 0.0);
        has_pos.resize(nc, 0);
        cap = nc;
    }
    double D(int i, int j) const { return i == j ? 0.0 : dis[(size_t)i * cap + j]; }
};

extern "C" {

nv_graph* nv_graph_create(void) { return new (std::nothrow) nv_graph(); }

void nv_graph_destroy(nv_graph* g) { delete g; }

// a new node; returns its id (0, 1, 2, ...)
int nv_graph_add_node(nv_graph* g) {
    if (!g) return NV_ERR_ARG;
    g->grow(g->n + 1);
    return g->n++;
}

int nv_graph_num_nodes(const nv_graph* g) { return g ? g->n : NV_ERR_ARG; }

int nv_graph_set_position(nv_graph* g, int node, const double* xyz) {
    if (!g || !xyz || node < 0 || node >= g->n) return NV_ERR_ARG;
    memcpy(&g->pos[(size_t)node * 3], xyz, 3 * sizeof(double));
    g->has_pos[node] = 1;
    return NV_OK;
}

// FloydGraph.add_edge (graph_utils.py:59-64): keep the shorter of the old and the new length, symmetric, direct
int nv_graph_add_edge(nv_graph* g, int x, int y, double dist) {
    if (!g || x < 0 || y < 0 || x >= g->n || y >= g->n) return NV_ERR_ARG;
    const size_t c = g->cap;
    if (dist < g->dis[x * c + y]) {
        g->dis[x * c + y] = g->dis[y * c + x] = dist;
        g->point[x * c + y] = g->point[y * c + x] = -1;
    }
    return NV_OK;
}

// FloydGraph.update(k) (graph_utils.py:66-75): relax every ordered pair through k in the reference's sweep order (x outer,
// y inner, the symmetric entry written along), then mark k visited
int nv_graph_update(nv_graph* g, int k) {
    if (!g || k < 0 || k >= g->n) return NV_ERR_ARG;
    const size_t c = g->cap;
    const int n = g->n;
    double* d = g->dis.data();
    int* p = g->point.data();
    for (int x = 0; x < n; ++x) {
        const double dxk = d[x * c + k];
        if (dxk >= kNoEdge) continue;                       // nothing can improve through an unreachable k
        for (int y = 0; y < n; ++y) {
            if (x == y) continue;
            const double via = dxk + d[k * c + y];
            if (via < d[x * c + y]) {
                d[x * c + y] = d[y * c + x] = via;
                p[x * c + y] = p[y * c + x] = k;
            }
        }
    }
    g->visited[k] = 1;
    return NV_OK;
}

int nv_graph_visited(const nv_graph* g, int k) { return (g && k >= 0 && k < g->n) ? g->visited[k] : 0; }

double nv_graph_distance(const nv_graph* g, int x, int y) {
    if (!g || x < 0 || y < 0 || x >= g->n || y >= g->n) return kNoEdge;
    return g->D(x, y);
}

// FloydGraph.path(x, y) (graph_utils.py:80-96): [v1, ..., y]; writes at most `cap` ids, returns the path length (>= 0)
int nv_graph_path(const nv_graph* g, int x, int y, int* out, int cap) {
    if (!g || x < 0 || y < 0 || x >= g->n || y >= g->n) return NV_ERR_ARG;
    if (x == y) return 0;
    // iterative expansion of the recursion path(x,k) + path(k,y) with an explicit stack of (from, to) segments
    std::vector<int> stack;
    stack.push_back(x);
    stack.push_back(y);
    int len = 0;
    const size_t c = g->cap;
    while (!stack.empty()) {
        const int b = stack.back(); stack.pop_back();
        const int a = stack.back(); stack.pop_back();
        if (a == b) continue;
        const int k = g->point[a * c + b];
        if (k < 0) {
            if (out && len < cap) out[len] = b;
            ++len;
            if (len > 4 * g->n + 8) return NV_ERR_ARG;      // corrupted predecessor table: never loop forever
        } else {
            stack.push_back(k); stack.push_back(b);         // second half first on the stack -> first half is expanded first
            stack.push_back(a); stack.push_back(k);
        }
    }
    return len;
}

// GraphMap.get_pos_fts (graph_utils.py:144-165) for all slots at once: ids[i] < 0 is the reference's `None` slot (zeros ->
// sin 0 = 0, cos 0 = 1).  out: [n, angle_feat_size + 3] fp32 = get_angle_fts(heading, elevation) repeated angle_feat_size/4
// times, then line distance / 30, shortest-path distance / 30, shortest-path steps / 10.  The reference computes the angles in
// fp64, stores them as fp32 and takes sin/cos of the fp32 values; so does this.
int nv_graph_pos_fts(const nv_graph* g, int cur, const int* ids, int n, double cur_heading, double cur_elevation,
                     int angle_feat_size, float* out) {
    if (!g || !ids || !out || cur < 0 || cur >= g->n || !g->has_pos[cur] || angle_feat_size < 4 || (angle_feat_size & 3)) return NV_ERR_ARG;
    const int rep = angle_feat_size / 4, w = angle_feat_size + 3;
    const double* a = &g->pos[(size_t)cur * 3];
    for (int i = 0; i < n; ++i) {
        float h = 0.f, e = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
        const int v = ids[i];
        if (v >= 0) {
            if (v >= g->n || !g->has_pos[v]) return NV_ERR_ARG;
            const double* b = &g->pos[(size_t)v * 3];
            const double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
            const double xy = fmax(sqrt(dx * dx + dy * dy), 1e-8), xyz = fmax(sqrt(dx * dx + dy * dy + dz * dz), 1e-8);
            double heading = asin(dx / xy);                 // the simulator's x-y axes are transposed (graph_utils.py:26-30)
            if (b[1] < a[1]) heading = M_PI - heading;
            heading -= cur_heading;
            const double elevation = asin(dz / xyz) - cur_elevation;
            h = (float)heading;
            e = (float)elevation;
            d0 = (float)(xyz / kMaxDist);
            d1 = (float)(g->D(cur, v) / kMaxDist);
            d2 = (float)(nv_graph_path(g, cur, v, nullptr, 0) / kMaxStep);
        }
        float* o = out + (size_t)i * w;
        // numpy evaluates sin/cos of the float32 angles in float32; here: double precision, rounded once (agrees with numpy's
        // float32 routines up to their own last-bit error)
        const float f4[4] = {(float)sin((double)h), (float)cos((double)h), (float)sin((double)e), (float)cos((double)e)};
        for (int r = 0; r < rep; ++r) memcpy(o + 4 * r, f4, sizeof(f4));
        o[angle_feat_size] = d0; o[angle_feat_size + 1] = d1; o[angle_feat_size + 2] = d2;
    }
    return NV_OK;
}

// nav_model.py:174-190 on integer ids.  Per sample b: map slots gmap_ids[b, 0..G) (-1 = padding or the stop slot 0), their
// visited flags, and the current panorama's candidate ids cand_ids[b, 0..Nv) (-1 = slot 0 / padding).  An unvisited map node
// (slot j > 0) that is one of the unvisited current candidates (view v > 0; the LAST such view wins, like the reference's dict)
// receives that view's embedding: src[b*G+j] = b*Nv+v and inv[b*Nv+v] = b*G+j; every other unvisited real slot gets token
// type 1.  Outputs must be sized [B*G], [B*Nv], [B*G]; src/inv are filled with -1 first, ttype with 0.
int nv_nav_match_tables(const int* gmap_ids, const uint8_t* gmap_visited, const int* cand_ids, int B, int G, int Nv, int* src,
                        int* inv, int* ttype) {
    if (!gmap_ids || !gmap_visited || !cand_ids || !src || !inv || !ttype || B < 0 || G < 0 || Nv < 0) return NV_ERR_ARG;
    for (long i = 0; i < (long)B * G; ++i) { src[i] = -1; ttype[i] = 0; }
    for (long i = 0; i < (long)B * Nv; ++i) inv[i] = -1;
    std::vector<int> slot_of;                                // node id -> view index of the candidate (per sample)
    std::vector<uint8_t> vis_node;
    for (int b = 0; b < B; ++b) {
        const int* gi = gmap_ids + (long)b * G;
        const uint8_t* gv = gmap_visited + (long)b * G;
        const int* ci = cand_ids + (long)b * Nv;
        int mx = -1;
        for (int j = 0; j < G; ++j) mx = gi[j] > mx ? gi[j] : mx;
        for (int v = 0; v < Nv; ++v) mx = ci[v] > mx ? ci[v] : mx;
        slot_of.assign(mx + 1, -1);
        vis_node.assign(mx + 1, 0);
        for (int j = 0; j < G; ++j) if (gi[j] >= 0 && gv[j]) vis_node[gi[j]] = 1;
        for (int v = 1; v < Nv; ++v) if (ci[v] >= 0 && !vis_node[ci[v]]) slot_of[ci[v]] = v;
        for (int j = 1; j < G; ++j) {
            const int node = gi[j];
            if (node < 0 || vis_node[node]) continue;
            const int v = slot_of[node];
            if (v >= 0) {
                src[(long)b * G + j] = b * Nv + v;
                inv[(long)b * Nv + v] = b * G + j;
            } else {
                ttype[(long)b * G + j] = 1;
            }
        }
    }
    return NV_OK;
}

// nav_model.py:216-223 and :234-242.  cand_mask [B, G] (valid & unvisited, slot 0 = stop included), perm = the per-sample
// torch.randperm results concatenated (sample b's has cand_num[b]-1 entries, starting at perm_off[b]).
//   sel      [n_sel = sum(cand_num-1)]  flat fuse rows in LM order: sample by sample, non-stop candidates permuted
//   inv_sel  [B*G]                      position of flat row (b, g) in sel, -1 if not selected
//   col      [B, G] (int64)             head output column that holds slot g's logit (0 for stop, 1 + inverse-perm rank)
// returns n_sel (>= 0)
int nv_nav_perm_tables(const uint8_t* cand_mask, const long* perm, const int* perm_off, int B, int G, int* sel, int* inv_sel,
                       long* col) {
    if (!cand_mask || !perm || !perm_off || !sel || !inv_sel || !col || B < 0 || G < 0) return NV_ERR_ARG;
    for (long i = 0; i < (long)B * G; ++i) { inv_sel[i] = -1; col[i] = 0; }
    int n_sel = 0;
    std::vector<int> slots;
    for (int b = 0; b < B; ++b) {
        slots.clear();
        for (int g = 0; g < G; ++g) if (cand_mask[(long)b * G + g]) slots.push_back(g);
        if (slots.empty()) return NV_ERR_ARG;                // the stop slot is always a candidate
        const int m = (int)slots.size() - 1;                 // non-stop candidates
        const long* p = perm + perm_off[b];
        col[(long)b * G + slots[0]] = 0;
        for (int r = 0; r < m; ++r) {
            const long q = p[r];                             // LM position r of this sample takes candidate q (of the non-stop list)
            if (q < 0 || q >= m) return NV_ERR_ARG;
            const int g = slots[1 + q];
            inv_sel[(long)b * G + g] = n_sel;
            sel[n_sel++] = b * G + g;
            col[(long)b * G + g] = 1 + r;                    // pred[b, 1 + r] is candidate q's logit  (inverse permutation)
        }
    }
    return n_sel;
}

}  // extern "C"
