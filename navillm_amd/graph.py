"""Host-side topological map (CPU side-car of the hot path): restatement of
models/graph_utils.py:18-165 (FloydGraph / GraphMap / relative pose features) on numpy arrays.

Same results as the reference's dict-of-dict implementation (pinned by tests/golden/g7_graph.npz),
but the all-pairs state is a dense matrix so `update` is a vectorised O(n^2) relaxation and
`get_pos_fts` evaluates all nodes at once (SURVEY.md §8f item 3 "vectorised graph side-car" starts here).
"""
import numpy as np

MAX_DIST = 30
MAX_STEP = 10
_INF = 95959595  # the reference's "no edge" sentinel (graph_utils.py:49)


def calculate_vp_rel_pos_fts(a, b, base_heading=0, base_elevation=0):
    """graph_utils.py:18-35, vectorised over b: a (3,), b (n,3) -> heading, elevation, dist (n,)"""
    a = np.asarray(a, dtype=np.float64)
    b = np.atleast_2d(np.asarray(b, dtype=np.float64))
    dx, dy, dz = b[:, 0] - a[0], b[:, 1] - a[1], b[:, 2] - a[2]
    xy = np.maximum(np.sqrt(dx ** 2 + dy ** 2), 1e-8)
    xyz = np.maximum(np.sqrt(dx ** 2 + dy ** 2 + dz ** 2), 1e-8)
    heading = np.arcsin(dx / xy)
    heading = np.where(b[:, 1] < a[1], np.pi - heading, heading) - base_heading
    elevation = np.arcsin(dz / xyz) - base_elevation
    return heading, elevation, xyz


def get_angle_fts(headings, elevations, angle_feat_size):
    """graph_utils.py:38-44"""
    f = np.vstack([np.sin(headings), np.cos(headings), np.sin(elevations), np.cos(elevations)]).transpose().astype(np.float32)
    r = angle_feat_size // 4
    return np.concatenate([f] * r, 1) if r > 1 else f


class FloydGraph:
    """graph_utils.py:47-96 on a growing dense matrix."""

    def __init__(self):
        self.idx = {}
        self.names = []
        self.dis = np.zeros((0, 0), dtype=np.float64)
        self.point = np.zeros((0, 0), dtype=np.int64)   # -1 = direct edge
        self._visited = set()

    def _id(self, x):
        if x not in self.idx:
            n = len(self.names)
            self.idx[x] = n
            self.names.append(x)
            d = np.full((n + 1, n + 1), float(_INF))
            p = np.full((n + 1, n + 1), -1, dtype=np.int64)
            d[:n, :n] = self.dis
            p[:n, :n] = self.point
            self.dis, self.point = d, p
        return self.idx[x]

    def distance(self, x, y):
        if x == y:
            return 0
        if x not in self.idx or y not in self.idx:
            return _INF
        return self.dis[self.idx[x], self.idx[y]]

    def add_edge(self, x, y, dis):
        i, j = self._id(x), self._id(y)
        if dis < self.dis[i, j]:
            self.dis[i, j] = self.dis[j, i] = dis
            self.point[i, j] = self.point[j, i] = -1

    def update(self, k):
        """relax every pair through k (graph_utils.py:66-75); the reference's sequential sweep and this
        one-shot relaxation agree because dis[x,k] and dis[k,y] are not themselves improved via k."""
        kk = self._id(k)
        via = self.dis[:, kk][:, None] + self.dis[kk, :][None, :]
        better = via < self.dis
        np.fill_diagonal(better, False)
        self.dis = np.where(better, via, self.dis)
        self.point = np.where(better, kk, self.point)
        self._visited.add(k)

    def visited(self, k):
        return k in self._visited

    def path(self, x, y):
        if x == y:
            return []
        i, j = self.idx[x], self.idx[y]
        k = self.point[i, j]
        if k < 0:
            return [y]
        return self.path(x, self.names[k]) + self.path(self.names[k], y)


class GraphMap:
    """graph_utils.py:99-165 (node embeddings kept as running sums of detached device tensors)."""

    def __init__(self, start_vp):
        self.start_vp = start_vp
        self.node_positions = {}
        self.graph = FloydGraph()
        self.node_embeds = {}
        self.node_step_ids = {}

    def update_graph(self, ob):
        self.node_positions[ob["viewpoint"]] = ob["position"]
        for cc in ob["candidate"]:
            self.node_positions[cc["viewpointId"]] = cc["position"]
            a, b = np.asarray(ob["position"], dtype=np.float64), np.asarray(cc["position"], dtype=np.float64)
            self.graph.add_edge(ob["viewpoint"], cc["viewpointId"], float(np.sqrt(((b - a) ** 2).sum())))
        self.graph.update(ob["viewpoint"])

    def update_node_embed(self, vp, embed, rewrite=False):
        if rewrite or vp not in self.node_embeds:
            self.node_embeds[vp] = [embed, 1, embed]          # running sum, count, cached mean
        else:
            e = self.node_embeds[vp]
            e[0] = e[0] + embed
            e[1] += 1
            e[2] = None

    def get_node_embed(self, vp):
        """sum / count (graph_utils.py:137-142).  The mean is cached until the node is updated again: the reference
        re-divides every node on every step, which on a device tensor is one tiny kernel launch per node per step
        (432 launches per 8-episode step here); the cached value is bit-identical."""
        e = self.node_embeds[vp]
        if e[2] is None:
            e[2] = e[0] / e[1]
        return e[2]

    def get_pos_fts(self, cur_vp, gmap_vpids, cur_heading, cur_elevation, angle_feat_size=4):
        n = len(gmap_vpids)
        ang = np.zeros((n, 2), dtype=np.float32)
        dist = np.zeros((n, 3), dtype=np.float32)
        real = [i for i, v in enumerate(gmap_vpids) if v is not None]
        if real:
            pos = np.stack([np.asarray(self.node_positions[gmap_vpids[i]], dtype=np.float64) for i in real])
            h, e, dd = calculate_vp_rel_pos_fts(self.node_positions[cur_vp], pos, cur_heading, cur_elevation)
            ang[real, 0], ang[real, 1] = h, e
            for r, i in enumerate(real):
                vp = gmap_vpids[i]
                dist[i] = [dd[r] / MAX_DIST, self.graph.distance(cur_vp, vp) / MAX_DIST,
                           len(self.graph.path(cur_vp, vp)) / MAX_STEP]
        return np.concatenate([get_angle_fts(ang[:, 0], ang[:, 1], angle_feat_size), dist], 1)
