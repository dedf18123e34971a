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
    std::vector<int> step_id;      // [cap]  GraphMap.node_step_ids (0 = never stood on)
    std::vector<int> pos_order;    // nodes in the order their position was FIRST set = the insertion order of the reference's
                                   // `node_positions` dict, which is the order mp3d_agent.py:316-321 lists the map slots in.  (Not the
                                   // node-id order: add_edge / a step-id write / an adopted pre-built graph intern ids earlier.)

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
        step_id.resize(nc, 0);
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
    if (!g->has_pos[node]) g->pos_order.push_back(node);
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

// GraphMap.node_step_ids[vp] = step (mp3d_agent.py:688): kept next to the node so that the collation below needs no Python dict
int nv_graph_set_step_id(nv_graph* g, int node, int step) {
    if (!g || node < 0 || node >= g->n) return NV_ERR_ARG;
    g->step_id[node] = step;
    return NV_OK;
}

// MP3DAgent.nav_gmap_variable + the pose half of nav_vp_variable (tasks/agents/mp3d_agent.py:264-371) for a whole batch in ONE call.
// Per sample b (map graphs[b], agent at node cur[b] with heading / elevation, episode start node start[b], the current panorama's
// candidate nodes cand_ids[cand_off[b] .. cand_off[b+1])):
//   map slots = [stop] + visited nodes + unvisited nodes (enc_full_graph) or [stop] + unvisited nodes, each group in the order the
//   nodes entered `node_positions` (the order their position was first set: nv_graph::pos_order -- equal to the node-id order only
//   while every node is interned by its position write);
//   G = the longest slot list of the batch; every output row is padded to G (zeros / -1 / false), rows are G apart:
//     gmap_ids [B,G] i32 (-1 = stop slot / padding)      gmap_step_ids [B,G] i64       gmap_visited [B,G] u8      gmap_masks [B,G] u8
//     gmap_pos_fts [B,G,afs+3] f32 (GraphMap.get_pos_fts of every slot)               gmap_lens [B] i32          no_vp_left [B] u8
//     pair_dists [B,G,G] f32 or NULL (graph.distance between the slots' nodes; the reference builds it, NaviLLM's model never reads it)
//     vp_pos_fts [B,Nv,2*(afs+3)] f32: [:, :afs+3] = pose of the start node, rows 1..K [afs+3:] = pose of candidate k (mp3d_agent.py:275-289)
//     vp_cand_ids [B,Nv] i32 (-1 at slot 0 and past the K candidates)
// Every output buffer must hold the row counts for G = Gcap.  Returns G (>= 1), or a negative status (Gcap too small: NV_ERR_ARG).
// Pure host code on the caller's (pinned) staging buffers; no allocation beyond two small scratch vectors.
int nv_nav_collate(nv_graph* const* graphs, int B, const int* cur, const int* start, const double* heading, const double* elevation,
                   const int* cand_ids, const int* cand_off, int Nv, int angle_feat_size, int enc_full_graph, int Gcap,
                   int* gmap_ids, long* gmap_step_ids, uint8_t* gmap_visited, uint8_t* gmap_masks, float* gmap_pos_fts, int* gmap_lens,
                   uint8_t* no_vp_left, float* pair_dists, float* vp_pos_fts, int* vp_cand_ids) {
    if (!graphs || B <= 0 || !cur || !start || !heading || !elevation || !cand_ids || !cand_off || Nv < 1 || angle_feat_size < 4 ||
        (angle_feat_size & 3) || Gcap < 1 || !gmap_ids || !gmap_step_ids || !gmap_visited || !gmap_masks || !gmap_pos_fts || !gmap_lens ||
        !no_vp_left || !vp_pos_fts || !vp_cand_ids)
        return NV_ERR_ARG;
    const int w = angle_feat_size + 3;
    int G = 1;
    for (int b = 0; b < B; ++b) {
        const nv_graph* g = graphs[b];
        if (!g || cur[b] < 0 || cur[b] >= g->n || start[b] < 0 || start[b] >= g->n) return NV_ERR_ARG;
        int len = 1;
        for (int v : g->pos_order) len += (enc_full_graph || !g->visited[v]) ? 1 : 0;
        gmap_lens[b] = len;
        G = len > G ? len : G;
        const int K = cand_off[b + 1] - cand_off[b];
        if (K < 0 || K > Nv - 1) return NV_ERR_ARG;
    }
    if (G > Gcap) return NV_ERR_ARG;
    memset(gmap_step_ids, 0, sizeof(long) * (size_t)B * G);
    memset(gmap_visited, 0, (size_t)B * G);
    memset(gmap_masks, 0, (size_t)B * G);
    memset(gmap_pos_fts, 0, sizeof(float) * (size_t)B * G * w);
    memset(vp_pos_fts, 0, sizeof(float) * (size_t)B * Nv * 2 * w);
    if (pair_dists) memset(pair_dists, 0, sizeof(float) * (size_t)B * G * G);
    std::vector<float> tmp;
    for (int b = 0; b < B; ++b) {
        const nv_graph* g = graphs[b];
        int* ids = gmap_ids + (size_t)b * G;
        for (int j = 0; j < G; ++j) ids[j] = -1;
        int j = 1, n_unvis = 0;
        if (enc_full_graph)
            for (int v : g->pos_order)
                if (g->visited[v]) { gmap_visited[(size_t)b * G + j] = 1; ids[j++] = v; }
        for (int v : g->pos_order)
            if (!g->visited[v]) { ids[j++] = v; ++n_unvis; }
        const int len = gmap_lens[b];
        if (j != len) return NV_ERR_ARG;
        no_vp_left[b] = n_unvis == 0;
        for (int k = 0; k < len; ++k) {
            gmap_masks[(size_t)b * G + k] = 1;
            gmap_step_ids[(size_t)b * G + k] = ids[k] >= 0 ? g->step_id[ids[k]] : 0;
        }
        int rc = nv_graph_pos_fts(g, cur[b], ids, len, heading[b], elevation[b], angle_feat_size, gmap_pos_fts + (size_t)b * G * w);
        if (rc != NV_OK) return rc;
        if (pair_dists) {
            float* pd = pair_dists + (size_t)b * G * G;
            for (int x = 1; x < len; ++x)
                for (int y = x + 1; y < len; ++y) pd[(size_t)x * G + y] = pd[(size_t)y * G + x] = (float)g->D(ids[x], ids[y]);
        }
        // nav_vp_variable: [stop] + K candidates; every row carries the start node's pose in its first half
        const int K = cand_off[b + 1] - cand_off[b];
        const int* cand = cand_ids + cand_off[b];
        int* vc = vp_cand_ids + (size_t)b * Nv;
        for (int v = 0; v < Nv; ++v) vc[v] = -1;
        float sf[64];
        if (w > 64) return NV_ERR_ARG;
        rc = nv_graph_pos_fts(g, cur[b], &start[b], 1, heading[b], elevation[b], angle_feat_size, sf);
        if (rc != NV_OK) return rc;
        tmp.resize((size_t)(K > 0 ? K : 1) * w);
        if (K > 0) {
            rc = nv_graph_pos_fts(g, cur[b], cand, K, heading[b], elevation[b], angle_feat_size, tmp.data());
            if (rc != NV_OK) return rc;
        }
        float* vp = vp_pos_fts + (size_t)b * Nv * 2 * w;
        for (int v = 0; v < Nv; ++v) memcpy(vp + (size_t)v * 2 * w, sf, sizeof(float) * w);
        for (int k = 0; k < K; ++k) {
            memcpy(vp + (size_t)(k + 1) * 2 * w + w, tmp.data() + (size_t)k * w, sizeof(float) * w);
            vc[k + 1] = cand[k];
        }
    }
    return G;
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
