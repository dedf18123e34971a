"""Host-side topological map (the CPU side-car of the hot path): the reference's `FloydGraph` / `GraphMap`
(models/graph_utils.py:47-165) and the per-step index tables of `forward_navigation` (models/nav_model.py:174-190,216-223,
234-242), served by the C++ side-car behind the C ABI (`nv_graph_*`, `nv_nav_*`: navillm_amd/csrc/graph_host.cpp,
include/navillm_hip.h).

Same Python surface as the reference classes (viewpoint-id strings in, numpy out), so the agent-side code that drives them
is unchanged; underneath, nodes are interned to small integers once and every per-step operation is one C call on dense
matrices: the O(n^2) relaxation per visit, all slots' 7-d pose features in one call, the candidate/map matching tables in one
call per batch.  Results are pinned by tests/golden/g7_graph.npz (generated from the reference) through this C ABI.
"""
import ctypes
import os

import numpy as np

from . import lib as _lib

MAX_DIST = 30
MAX_STEP = 10
_INF = 95959595  # the reference's "no edge" sentinel (graph_utils.py:49)
_I32P = ctypes.POINTER(ctypes.c_int)


def _L():
    return _lib.load()


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def calculate_vp_rel_pos_fts(a, b, base_heading=0, base_elevation=0):
    """graph_utils.py:18-35, vectorised over b: a (3,), b (n,3) -> heading, elevation, dist (n,)  (numpy helper for callers
    that hold raw positions; the map itself uses nv_graph_pos_fts)"""
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
    """graph_utils.py:47-96 over `nv_graph_*`."""

    def __init__(self):
        self._g = ctypes.c_void_p(_L().nv_graph_create())
        if not self._g:
            raise MemoryError("nv_graph_create failed")
        self.idx = {}
        self.names = []

    def __del__(self):
        g, self._g = getattr(self, "_g", None), None
        if g:
            try:
                _L().nv_graph_destroy(g)
            except Exception:
                pass

    def _id(self, x):
        i = self.idx.get(x)
        if i is None:
            i = _L().nv_graph_add_node(self._g)
            assert i == len(self.names)
            self.idx[x] = i
            self.names.append(x)
        return i

    def distance(self, x, y):
        if x == y:
            return 0
        if x not in self.idx or y not in self.idx:
            return _INF
        return _L().nv_graph_distance(self._g, self.idx[x], self.idx[y])

    def add_edge(self, x, y, dis):
        _lib.check(_L().nv_graph_add_edge(self._g, self._id(x), self._id(y), float(dis)), "nv_graph_add_edge")

    def update(self, k):
        _lib.check(_L().nv_graph_update(self._g, self._id(k)), "nv_graph_update")

    def visited(self, k):
        i = self.idx.get(k)
        return i is not None and bool(_L().nv_graph_visited(self._g, i))

    def path(self, x, y):
        if x == y:
            return []
        n = len(self.names)
        buf = (ctypes.c_int * (n + 1))()
        ln = _L().nv_graph_path(self._g, self.idx[x], self.idx[y], buf, n + 1)
        if ln < 0:
            raise _lib.NaviLLMHipError("nv_graph_path failed")
        return [self.names[buf[i]] for i in range(ln)]


class _Positions(dict):
    """`GraphMap.node_positions`: a dict whose writes are mirrored into the C graph"""

    def __init__(self, owner):
        super().__init__()
        self._owner = owner

    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        self._owner._push_position(k, v)


class _StepIds(dict):
    """`GraphMap.node_step_ids`: a dict whose writes are mirrored into the C graph (nv_nav_collate reads them there)"""

    def __init__(self, owner):
        super().__init__()
        self._owner = owner

    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        g = self._owner._graph
        _lib.check(_L().nv_graph_set_step_id(g._g, g._id(k), int(v)), "nv_graph_set_step_id")

    # dict's bulk writers do not go through __setitem__: mirror them too
    def update(self, *a, **k):
        for key, v in dict(*a, **k).items():
            self[key] = v

    def setdefault(self, key, default=None):
        if key not in self:
            self[key] = default
        return self[key]


class GraphMap:
    """graph_utils.py:99-165 (node embeddings kept as running sums of detached device tensors)."""

    def __init__(self, start_vp):
        self.start_vp = start_vp
        self._graph = FloydGraph()
        self._positions = _Positions(self)
        self.node_embeds = {}
        self._step_ids = _StepIds(self)

    @property
    def node_step_ids(self):
        return self._step_ids

    @node_step_ids.setter
    def node_step_ids(self, d):
        self._step_ids = _StepIds(self)
        for k, v in dict(d).items():
            self._step_ids[k] = v

    # the reference's attributes stay assignable (tests / callers replace them wholesale)
    @property
    def graph(self):
        return self._graph

    @graph.setter
    def graph(self, g):
        self._graph = g
        for k, v in self._positions.items():
            self._push_position(k, v)
        for k, v in list(self._step_ids.items()):
            self._step_ids[k] = v

    @property
    def node_positions(self):
        return self._positions

    @node_positions.setter
    def node_positions(self, d):
        self._positions = _Positions(self)
        for k, v in dict(d).items():
            self._positions[k] = v

    def _push_position(self, vp, pos):
        p = (ctypes.c_double * 3)(*[float(x) for x in pos])
        _lib.check(_L().nv_graph_set_position(self._graph._g, self._graph._id(vp), p), "nv_graph_set_position")

    def update_graph(self, ob):
        self.node_positions[ob["viewpoint"]] = ob["position"]
        a = np.asarray(ob["position"], dtype=np.float64)
        for cc in ob["candidate"]:
            self.node_positions[cc["viewpointId"]] = cc["position"]
            b = np.asarray(cc["position"], dtype=np.float64)
            self._graph.add_edge(ob["viewpoint"], cc["viewpointId"], float(np.sqrt(((b - a) ** 2).sum())))
        self._graph.update(ob["viewpoint"])

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

    def node_ids(self, vpids):
        """viewpoint ids -> int32 node ids (-1 for the reference's `None` slot)"""
        idx = self._graph.idx
        return np.fromiter((-1 if v is None else idx[v] for v in vpids), dtype=np.int32, count=len(vpids))

    def get_pos_fts(self, cur_vp, gmap_vpids, cur_heading, cur_elevation, angle_feat_size=4, out=None):
        """[len(gmap_vpids), angle_feat_size + 3] fp32, one C call; `out` may be a (pinned) buffer to fill in place"""
        ids = self.node_ids(gmap_vpids)
        if out is None:
            out = np.empty((len(gmap_vpids), angle_feat_size + 3), dtype=np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape == (len(gmap_vpids), angle_feat_size + 3)
        _lib.check(_L().nv_graph_pos_fts(self._graph._g, self._graph.idx[cur_vp], _ptr(ids), len(gmap_vpids), float(cur_heading),
                                         float(cur_elevation), int(angle_feat_size), _ptr(out)), "nv_graph_pos_fts")
        return out


# ------------------------------------------------------------------ per-step collation of the maps (the agent's tensor builders)
class NavCollator:
    """`MP3DAgent.nav_gmap_variable` + the pose half of `nav_vp_variable` (tasks/agents/mp3d_agent.py:264-371) for a whole batch in
    ONE C call (`nv_nav_collate`): map slots [stop] + visited + unvisited nodes per sample padded to the batch's longest list, step
    ids, visited / valid masks, the 7-d pose of every slot, the 14-d pose table of the panorama's views -- written straight into a
    pinned staging buffer and shipped to the device in ONE asynchronous copy.  A ring of staging buffers guarded by events lets the
    host run ahead of the GPU (training with teacher forcing never waits for a step's result).

    `collate()` -> dict of numpy views (host side, valid until the ring comes round) and, with `device`, the device tensors under the
    reference's batch keys: gmap_step_ids [B,G] i64, gmap_pos_fts [B,G,7] f32, gmap_visited_masks / gmap_masks [B,G] bool,
    vp_pos_fts [B,Nv,14] f32; plus gmap_ids / vp_cand_ids (int32 node ids for `match_tables`), gmap_lens, no_vp_left."""

    # staging buffers in the ring.  Round 5: 4 -> 32.  A slot is reused once the copy that read it has RUN, and that copy is ordered
    # behind everything queued before it -- under teacher forcing the whole deferred backward of the previous episode (~270 ms at 7B):
    # with 4 slots the host blocked on its 5th step of every episode until the GPU had drained (tools/episode_host_probe.py: step 4
    # took 278 ms of host time), then the GPU idled while the host caught up -- 2-4 % of every episode.  32 slots (a few MB of pinned
    # memory) let the host run five episodes ahead.  128: an accumulation window (`begin_episode(..., accumulate=8)`) records 48+ steps
    # while the previous window's batch still occupies the GPU.
    RING = 128

    def __init__(self, B, Nv, Gcap=128, angle_feat_size=4, enc_full_graph=True, pair_dists=False, pin=None):
        import torch
        self.B, self.Nv, self.Gcap, self.afs, self.full, self.pair = B, Nv, Gcap, angle_feat_size, bool(enc_full_graph), bool(pair_dists)
        w = angle_feat_size + 3
        # regions (capacity for G = Gcap), 16-byte aligned so that dtype views of the packed buffer are legal on both sides
        spec = [("gmap_step_ids", np.int64, B * Gcap), ("gmap_pos_fts", np.float32, B * Gcap * w), ("vp_pos_fts", np.float32, B * Nv * 2 * w),
                ("gmap_ids", np.int32, B * Gcap), ("vp_cand_ids", np.int32, B * Nv), ("gmap_lens", np.int32, B),
                ("gmap_visited", np.uint8, B * Gcap), ("gmap_masks", np.uint8, B * Gcap), ("no_vp_left", np.uint8, B)]
        if self.pair:
            spec.append(("pair_dists", np.float32, B * Gcap * Gcap))
        self.off, o = {}, 0
        for name, dt, n in spec:
            self.off[name] = (o, dt, n)
            o += (n * np.dtype(dt).itemsize + 15) // 16 * 16
        self.nbytes = o
        pin = torch.cuda.is_available() if pin is None else pin
        # (up to RING slots within ~8 MB of pinned memory, never fewer than 32: the [B, G, G] distance matrices make a slot 0.5 MB)
        self.RING = max(32, min(int(os.environ.get("NAVILLM_COLLATE_RING", self.RING)), (8 << 20) // max(o, 1)))
        self.ring = [torch.empty(o, dtype=torch.uint8, pin_memory=pin) for _ in range(self.RING)]
        self.events = [None] * self.RING
        self.k = 0
        self._w = w

    def _region(self, buf_np, name):
        o, dt, n = self.off[name]
        return buf_np[o:o + n * np.dtype(dt).itemsize].view(dt)

    def collate(self, gmaps, cur_vps, headings, elevations, cand_vpids, device=None):
        import torch
        B, Nv, w = self.B, self.Nv, self._w
        assert len(gmaps) == B
        slot = self.k % self.RING
        self.k += 1
        if self.events[slot] is not None:
            self.events[slot].synchronize()                  # the copy that last read this staging buffer has run
            self.events[slot] = None
        host = self.ring[slot]
        hnp = host.numpy()
        handles = (ctypes.c_void_p * B)(*[g._graph._g for g in gmaps])
        cur = np.fromiter((g._graph.idx[v] for g, v in zip(gmaps, cur_vps)), np.int32, B)
        start = np.fromiter((g._graph.idx[g.start_vp] for g in gmaps), np.int32, B)
        hd = np.asarray(headings, np.float64)
        el = np.asarray(elevations, np.float64)
        off = np.zeros(B + 1, np.int32)
        off[1:] = np.cumsum([len(c) for c in cand_vpids])
        cand = np.fromiter((g._graph.idx[v] for g, c in zip(gmaps, cand_vpids) for v in c), np.int32, int(off[-1]))
        R = {n: self._region(hnp, n) for n in self.off}
        G = _L().nv_nav_collate(handles, B, _ptr(cur), _ptr(start), _ptr(hd), _ptr(el), _ptr(cand), _ptr(off), Nv, self.afs, int(self.full),
                                self.Gcap, _ptr(R["gmap_ids"]), _ptr(R["gmap_step_ids"]), _ptr(R["gmap_visited"]), _ptr(R["gmap_masks"]),
                                _ptr(R["gmap_pos_fts"]), _ptr(R["gmap_lens"]), _ptr(R["no_vp_left"]),
                                _ptr(R["pair_dists"]) if self.pair else None, _ptr(R["vp_pos_fts"]), _ptr(R["vp_cand_ids"]))
        if G < 1:
            raise _lib.NaviLLMHipError(f"nv_nav_collate failed ({G}): a map with more than Gcap = {self.Gcap} slots, or an unknown node")
        shapes = {"gmap_step_ids": (B, G), "gmap_pos_fts": (B, G, w), "vp_pos_fts": (B, Nv, 2 * w), "gmap_ids": (B, G), "vp_cand_ids": (B, Nv),
                  "gmap_lens": (B,), "gmap_visited": (B, G), "gmap_masks": (B, G), "no_vp_left": (B,), "pair_dists": (B, G, G)}
        out = {"G": G, "host": {n: R[n][:int(np.prod(shapes[n]))].reshape(shapes[n]) for n in self.off}}
        if device is not None:
            dev = host.to(device, non_blocking=True)
            if dev.is_cuda:
                ev = torch.cuda.Event()
                ev.record()
                self.events[slot] = ev
            tdt = {np.int64: torch.int64, np.float32: torch.float32, np.int32: torch.int32, np.uint8: torch.uint8}

            def view(name):
                o, dt, _ = self.off[name]
                n = int(np.prod(shapes[name]))
                return dev[o:o + n * np.dtype(dt).itemsize].view(tdt[dt]).view(*shapes[name])
            out.update(gmap_step_ids=view("gmap_step_ids"), gmap_pos_fts=view("gmap_pos_fts"), vp_pos_fts=view("vp_pos_fts"),
                       gmap_visited_masks=view("gmap_visited").view(torch.bool), gmap_masks=view("gmap_masks").view(torch.bool))
            if self.pair:
                out["gmap_pair_dists"] = view("pair_dists")
        return out

    def vpids(self, gmaps, host):
        """the reference's `gmap_vpids` lists (viewpoint-id strings, None for the stop slot) of the last collate()"""
        ids, lens = host["gmap_ids"], host["gmap_lens"]
        return [[None] + [g._graph.names[i] for i in ids[b, 1:lens[b]]] for b, g in enumerate(gmaps)]


# ------------------------------------------------------------------ per-step index tables of forward_navigation
def match_tables(gmap_ids, gmap_visited, cand_ids):
    """nav_model.py:174-190 on integer ids ([B,G] int32 with -1 padding, [B,G] bool, [B,Nv] int32 with -1 at slot 0 / padding)
    -> src [B*G] int32, inv [B*Nv] int32, ttype [B*G] int32"""
    gmap_ids = np.ascontiguousarray(gmap_ids, dtype=np.int32)
    vis = np.ascontiguousarray(gmap_visited, dtype=np.uint8)
    cand_ids = np.ascontiguousarray(cand_ids, dtype=np.int32)
    B, G = gmap_ids.shape
    Nv = cand_ids.shape[1]
    src, inv, tt = np.empty(B * G, np.int32), np.empty(B * Nv, np.int32), np.empty(B * G, np.int32)
    _lib.check(_L().nv_nav_match_tables(_ptr(gmap_ids), _ptr(vis), _ptr(cand_ids), B, G, Nv, _ptr(src), _ptr(inv), _ptr(tt)),
               "nv_nav_match_tables")
    return src, inv, tt


def intern_vpids(gmap_vpids, vp_cand_vpids, G, Nv):
    """string viewpoint ids of one batch -> the integer tables `match_tables` takes (a per-call dictionary: ids only need to be
    consistent within a sample).  Callers that already hold node ids (navillm_amd/synthetic.py) skip this."""
    B = len(gmap_vpids)
    gi = np.full((B, G), -1, dtype=np.int32)
    ci = np.full((B, Nv), -1, dtype=np.int32)
    for b in range(B):
        table = {}
        for j, v in enumerate(gmap_vpids[b]):
            if v is not None:
                gi[b, j] = table.setdefault(v, len(table))
        for j, v in enumerate(vp_cand_vpids[b]):
            if j > 0 and v is not None:
                ci[b, j] = table.setdefault(v, len(table))
    return gi, ci


def perm_tables(cand_mask, perms):
    """nav_model.py:216-223,234-242: cand_mask [B,G] bool, perms = list of B int64 permutations (torch.randperm results)
    -> sel [n] int32 (fuse rows in LM order), inv_sel [B*G] int32, col [B,G] int64 (head column of every slot)"""
    cm = np.ascontiguousarray(cand_mask, dtype=np.uint8)
    B, G = cm.shape
    off = np.zeros(B + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(p) for p in perms])
    flat = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.int64) for p in perms]) if off[-1] else np.zeros(0, np.int64))
    sel, inv_sel, col = np.empty(max(int(off[-1]), 1), np.int32), np.empty(B * G, np.int32), np.empty((B, G), np.int64)
    n = _L().nv_nav_perm_tables(_ptr(cm), _ptr(flat), _ptr(off), B, G, _ptr(sel), _ptr(inv_sel), _ptr(col))
    if n < 0:
        raise _lib.NaviLLMHipError("nv_nav_perm_tables failed: invalid permutation / candidate mask")
    return sel[:n], inv_sel, col
