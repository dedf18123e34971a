"""Synthetic episode driver: reproduces the CALL PATTERN of `MP3DAgent.rollout`
(tasks/agents/mp3d_agent.py:593-964) on fabricated Matterport-shaped episodes (SURVEY.md §8d,
Appendix B).  No MatterSim, no datasets, no tokenizer files: views are N(0,1) features with the
simulator's 36 heading/elevation angles, the map is a random walk in 3-D, the instruction is 512
random in-vocab token ids wrapped in the real R2R prompt template.

Per nav step, in the reference's order:
  model('panorama') -> masked mean-pool + map update (detached) -> gmap/vp collation ->
  prompt -> model('navigation') -> CE_sum * w / B / accum -> backward (inside no_sync unless
  last step) -> teacher action -> history append -> move.
"""
import contextlib
import os
import math
import zlib

import numpy as np
import torch

from . import ops
from .graph import GraphMap, NavCollator, get_angle_fts
from .prompts import static_prefix, navigation_prompt, object_grounding_prompt, summarization_prompt, embodied_qa_prompt, qa3d_prompt


# --------------------------------------------------------------------------- stub tokenizer
class StubTokenizer:
    """whitespace/word-hash tokenizer for the fixed template text (roughly one token per word;
    the real Llama tokenizer gives ~1.3) with the five special tokens mapped to their real ids."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.special = {"<cand>": cfg.cand_token_id, "<hist>": cfg.hist_token_id, "<obj>": cfg.obj_token_id,
                        "<cls_1>": cfg.cls_token_ids[0], "<cls_2>": cfg.cls_token_ids[1]}
        self.bos, self.pad = 1, cfg.pad_token_id

    def _word(self, w):
        return 3 + zlib.crc32(w.encode()) % (self.cfg.base_vocab_size - 3)

    def encode(self, prompt, instr_ids=None, placeholder="{INSTR}"):
        ids = [self.bos]
        for w in prompt.split():
            if w == placeholder and instr_ids is not None:
                ids.extend(instr_ids)
            elif w in self.special:
                ids.append(self.special[w])
            else:
                ids.append(self._word(w))
        return ids

    def pad_left(self, seqs, max_length=1024, types=None):
        """left pad / left truncate (modified_lm.py:57,77-87); `types`: per-token segment ids (0 = prompt, 1 = answer) ->
        also returns token_type_ids (padding = 0), as the tokenizer does for a [prompt, answer] text pair"""
        seqs = [s[-max_length:] for s in seqs]              # truncation_side='left'
        S = max(len(s) for s in seqs)
        ids = torch.full((len(seqs), S), self.pad, dtype=torch.int64)
        am = torch.zeros((len(seqs), S), dtype=torch.int64)
        tt = torch.zeros((len(seqs), S), dtype=torch.int64)
        for i, s in enumerate(seqs):
            ids[i, S - len(s):] = torch.tensor(s)
            am[i, S - len(s):] = 1
            if types is not None:
                tt[i, S - len(s):] = torch.tensor(types[i][-max_length:])
        return (ids, am) if types is None else (ids, am, tt)

    def encode_pair(self, prompt, answer_ids, instr_ids=None, eos=2):
        """[prompt, answer + eos] as one sequence (nav_model.py:299-306, 370-377) -> ids, segment ids"""
        p = self.encode(prompt, instr_ids)
        a = list(answer_ids) + [eos]
        return p + a, [0] * len(p) + [1] * len(a)


# --------------------------------------------------------------------------- view geometry
def view_angle_features(angle_feat_size=4):
    """36 discretised views: 12 headings x 3 elevations (tasks/datasets/mp3d_envs.py:42-62) + [1,1,1] box."""
    h = np.array([(ix % 12) * math.radians(30) for ix in range(36)])
    e = np.array([(ix // 12 - 1) * math.radians(30) for ix in range(36)])
    return np.concatenate([get_angle_fts(h, e, angle_feat_size), np.ones((36, 3), np.float32)], 1)


class SyntheticEpisodes:
    """B lock-step episodes on one rank."""

    def __init__(self, cfg, batch_size, seed, instr_len=512, n_views=36, device=None, max_frontier=None, task="r2r", fast_maps=True):
        """max_frontier: cap on the number of known-but-unvisited nodes per map (long-horizon episodes: in a real
        Matterport graph most candidates of a late step are nodes seen before, so the frontier saturates; the action head is
        100-way (nav_model.py:82-85), i.e. stop + at most 99 unvisited candidates)."""
        self.cfg, self.B, self.N = cfg, batch_size, n_views
        self.seed = seed
        # fast_maps: the maps' node embeddings live in ONE device matrix per batch (running sums + counts, row = node id + 1) and the
        # visited flags in host sets, instead of one device tensor per node in `GraphMap.node_embeds` and one side-car call per
        # node: `update_maps` is then three index ops and `nav_inputs` one gather per step, for any map size (the per-node form
        # is what mp3d_agent.py:304-371 / graph_utils.py:119-142 do: ~1000 tiny device ops per 8-episode step at 100 map slots).
        # Same arithmetic -- sum in arrival order, divide by the count at read time -- hence bit-identical embeddings
        # (tests/test_round2_gpu.py::test_fast_maps_equal_the_per_node_maps).
        self.fast_maps = fast_maps and os.environ.get("NAVILLM_FAST_MAPS", "1") != "0"     # (env: A/B measurements)
        self.c_collate = os.environ.get("NAVILLM_C_COLLATE", "1") != "0"                   # round 4: the per-step collation in one C call
        self._collator = None
        self._next_x = None
        self._tok_cache = {}
        self.max_frontier = max_frontier
        self.task = task                  # which agent's prompts: r2r | reverie | soon | cvdn (tasks/agents/*.py)
        self.rng = np.random.RandomState(seed)
        self.tgen = torch.Generator().manual_seed(seed)
        self.device = device
        self.tok = StubTokenizer(cfg)
        self.instr_len = instr_len
        self.S_hist = []
        self.loc_fts = torch.from_numpy(view_angle_features(cfg.angle_feat_size)[:n_views])
        self.reset()

    def reset(self):
        B = self.B
        self.t = 0
        self._next_x = None
        self._tok_cache = {}
        self.instr = [self.rng.randint(3, self.cfg.base_vocab_size, size=self.instr_len).tolist() for _ in range(B)]
        self.pos = [{f"e{b}_n0": self.rng.randn(3) * 2.0} for b in range(B)]
        self.cur = [f"e{b}_n0" for b in range(B)]
        self.heading = [float(self.rng.rand() * 2 * math.pi) for _ in range(B)]
        self.gmaps = [GraphMap(self.cur[b]) for b in range(B)]
        self.history = [[] for _ in range(B)]
        self.hist_vis = [[] for _ in range(B)]
        self.n_nodes = [1] * B
        self.vis_nodes = [set() for _ in range(B)]             # viewpoints the agent has stood on (= FloydGraph.visited)
        self._store_rows = 0
        self.obs = [self._observe(b) for b in range(B)]
        for b in range(B):
            self._update_graph(b)

    def _update_graph(self, b):
        self.gmaps[b].update_graph(self.obs[b])
        self.vis_nodes[b].add(self.obs[b]["viewpoint"])

    def _visited(self, b, vp):
        return (vp in self.vis_nodes[b]) if self.fast_maps else self.gmaps[b].graph.visited(vp)

    def _store(self, rows_needed, d, device):
        """device store of the node embeddings: E_sum [B, R, d] fp32 running sums, E_cnt [B, R] counts; row 0 = the zero row of
        the stop slot / padding (count 1), node id k lives in row k + 1"""
        if self._store_rows < rows_needed:
            R = max(256, 1 << (rows_needed - 1).bit_length())
            E_sum = torch.zeros(self.B, R, d, dtype=torch.float32, device=device)
            E_cnt = torch.zeros(self.B, R, dtype=torch.float32, device=device)
            E_cnt[:, 0] = 1.0
            if self._store_rows:
                E_sum[:, :self._store_rows] = self.E_sum
                E_cnt[:, :self._store_rows] = self.E_cnt
            self.E_sum, self.E_cnt, self._store_rows = E_sum, E_cnt, R
        return self.E_sum, self.E_cnt

    def _observe(self, b):
        """new panorama at the current node: K candidates (mostly new frontier nodes, sometimes a known one)."""
        K = int(self.rng.randint(2, 9))
        cands = []
        known = [v for v in self.pos[b] if v != self.cur[b]]
        saturated = False
        if self.max_frontier is not None:
            frontier = [v for v in known if not self._visited(b, v)]
            saturated = len(frontier) >= self.max_frontier
            if saturated:
                known = frontier + [v for v in known if self._visited(b, v)][:2]     # mostly unvisited nodes: the walk can go on
        for j in range(K):
            if known and (saturated or self.rng.rand() < 0.25):
                vp = known[self.rng.randint(len(known))]
                if any(c["viewpointId"] == vp for c in cands):
                    continue
            else:
                vp = f"e{b}_n{self.n_nodes[b]}"
                self.n_nodes[b] += 1
                self.pos[b][vp] = self.pos[b][self.cur[b]] + self.rng.randn(3) * np.array([2.0, 2.0, 0.3])
            cands.append({"viewpointId": vp, "position": self.pos[b][vp], "pointId": j})
        return {"viewpoint": self.cur[b], "position": self.pos[b][self.cur[b]], "heading": self.heading[b],
                "elevation": 0.0, "candidate": cands}

    def prefetch_features(self):
        """draw the NEXT panorama's view features now (same generator, same order of panorama draws) -- called by the inference rollout
        right after a step's LM has been enqueued, so that this host work (2 ms at B = 8: a stand-in for the feature-DB lookup of
        tasks/feature_db.py, which does not depend on the model either) overlaps the GPU instead of sitting between the action's
        arrival and the next step's launches"""
        if self._next_x is None:
            self._next_x = torch.randn(self.B, self.N, self.cfg.image_feat_size, generator=self.tgen)

    # ---- tensor builders (mp3d_agent.py:143-212, 264-371)
    def panorama_inputs(self, objects=None, first12=False):
        """objects=(lo, hi): also O ~ U{lo..hi} object features per sample (panorama_feature_variable_object, mp3d_agent.py:143-212);
        first12: the 12-view variant of mp3d_agent.py:214-248 (all 36 views, the first 12 flagged navigable, no candidates)"""
        B, N, F = self.B, self.N, self.cfg.image_feat_size
        x, self._next_x = self._next_x, None
        if x is None:
            x = torch.randn(B, N, F, generator=self.tgen)
        nav = torch.zeros(B, N, dtype=torch.int64)
        cand_vpids = []
        for b, ob in enumerate(self.obs):
            if first12:
                nav[b, :min(12, N)] = 1
                cand_vpids.append([None] * N)
                continue
            k = len(ob["candidate"])
            nav[b, :k] = 1
            cand_vpids.append([c["viewpointId"] for c in ob["candidate"]])
        d = self.device
        out = {"view_img_fts": ops.h2d(x, d), "loc_fts": ops.h2d(self.loc_fts.unsqueeze(0).repeat(B, 1, 1), d),
               "nav_types": ops.h2d(nav, d), "view_lens": ops.h2d(torch.full((B,), N, dtype=torch.int64), d),
               "cand_vpids": cand_vpids}
        if objects is not None:
            lens = self.rng.randint(objects[0], objects[1] + 1, size=B)
            O = int(lens.max())
            of = torch.randn(B, O, self.cfg.obj_feat_size, generator=self.tgen)
            ang = self.rng.rand(B, O, 2) * np.array([2 * math.pi, 1.0]) - np.array([0.0, 0.5])
            olf = np.concatenate([get_angle_fts(ang[..., 0].reshape(-1), ang[..., 1].reshape(-1), self.cfg.angle_feat_size).reshape(B, O, -1),
                                  self.rng.rand(B, O, 3).astype(np.float32)], 2).astype(np.float32)
            for b in range(B):
                of[b, lens[b]:] = 0
                olf[b, lens[b]:] = 0
            out.update(obj_img_fts=ops.h2d(of, d), obj_lens=torch.from_numpy(lens.astype(np.int64)),
                       obj_loc_fts=ops.h2d(torch.from_numpy(olf), d))
        return out

    def update_maps(self, pano_embeds, pano_masks, cand_vpids):
        avg = ops.masked_mean_f32(pano_embeds.detach().contiguous(), pano_masks.to(torch.float32).contiguous())
        pe = pano_embeds.detach()
        if self.fast_maps:
            B, N, d = pe.shape
            need = 1 + max(len(g._graph.idx) for g in self.gmaps)
            E_sum, E_cnt = self._store(need, d, pe.device)
            R = self._store_rows
            cur_rows, add_rows, add_src = [], [], []
            for b, gmap in enumerate(self.gmaps):
                idx = gmap._graph.idx
                gmap.node_step_ids[self.cur[b]] = self.t + 1
                cur_rows.append(b * R + idx[self.cur[b]] + 1)
                vis = self.vis_nodes[b]
                for j, vp in enumerate(cand_vpids[b]):
                    if vp not in vis:
                        add_rows.append(b * R + idx[vp] + 1)
                        add_src.append(b * N + j)
            packed = ops.h2d(torch.tensor(cur_rows + add_rows + add_src, dtype=torch.int64), pe.device)
            nb, na = len(cur_rows), len(add_rows)
            cur_t, rows_t, src_t = packed[:nb], packed[nb:nb + na], packed[nb + na:]
            Ef, Cf = E_sum.view(B * R, d), E_cnt.view(B * R)
            Ef.index_copy_(0, cur_t, avg)                                        # visited node: rewrite with its own panorama mean
            Cf.index_fill_(0, cur_t, 1.0)
            if na:                                                               # frontier nodes: one more view pointing at them
                Ef.index_add_(0, rows_t, pe.reshape(B * N, d).index_select(0, src_t))    # (a node appears at most once per step:
                Cf.index_add_(0, rows_t, torch.ones(na, dtype=torch.float32, device=pe.device))   # the additions do not race)
            return
        for b, gmap in enumerate(self.gmaps):
            gmap.node_step_ids[self.cur[b]] = self.t + 1
            gmap.update_node_embed(self.cur[b], avg[b], rewrite=True)
            for j, vp in enumerate(cand_vpids[b]):
                if not gmap.graph.visited(vp):
                    gmap.update_node_embed(vp, pe[b, j])

    def nav_inputs(self, pano_embeds, pano_masks, cand_vpids):
        if self.fast_maps and self.c_collate:
            return self._nav_inputs_collated(pano_embeds, pano_masks, cand_vpids)
        return self._nav_inputs_python(pano_embeds, pano_masks, cand_vpids)

    def _nav_inputs_collated(self, pano_embeds, pano_masks, cand_vpids):
        """nav_gmap_variable + nav_vp_variable (mp3d_agent.py:264-371) through ONE side-car call (`NavCollator` -> nv_nav_collate):
        slot lists, step ids, masks, every pose feature -- in a pinned staging buffer, one H2D copy.  Same tensors as
        `_nav_inputs_python` (tests/test_host_cpu.py compares them on random maps)."""
        B, d, dev = self.B, self.cfg.hidden_size, self.device
        Nv = pano_embeds.shape[1] + 1
        col = self._collator
        if col is None or col.Nv != Nv:
            col = self._collator = NavCollator(B, Nv, Gcap=128, angle_feat_size=self.cfg.angle_feat_size)
        c = col.collate(self.gmaps, self.cur, self.heading, [0.0] * B, cand_vpids, device=dev)
        h, G = c["host"], c["G"]
        gids = h["gmap_ids"].copy()
        gmask = torch.from_numpy(h["gmap_masks"].astype(bool))
        gvis = torch.from_numpy(h["gmap_visited"].astype(bool))
        R = self._store_rows
        rows = (gids.astype(np.int64) + 1) + (np.arange(B, dtype=np.int64) * R)[:, None]
        rows_t = ops.h2d(torch.from_numpy(rows.reshape(-1)), dev)
        inv = (1.0 / self.E_cnt.view(B * R).index_select(0, rows_t))[:, None]
        gimg = (self.E_sum.view(B * R, d).index_select(0, rows_t) * inv).view(B, G, d)
        vp_img = torch.cat([torch.zeros_like(pano_embeds[:, :1]), pano_embeds], 1)
        pm = torch.cat([torch.ones_like(pano_masks[:, :1]), pano_masks], 1)
        return {"gmap_vpids": col.vpids(self.gmaps, h), "gmap_img_embeds": gimg, "gmap_step_ids": c["gmap_step_ids"],
                "gmap_pos_fts": c["gmap_pos_fts"], "gmap_visited_masks": c["gmap_visited_masks"], "gmap_masks": c["gmap_masks"],
                "vp_img_embeds": vp_img, "pano_masks": pm, "vp_pos_fts": c["vp_pos_fts"],
                "vp_nav_masks": torch.ones(B, Nv, dtype=torch.bool, device=dev),
                "vp_cand_vpids": [[None] + x for x in cand_vpids], "hist_vis": self.hist_vis, "history": self.history,
                "data_type": ["r2r"] * B, "instruction": ["{INSTR}"] * B,
                "_gmask_cpu": gmask, "_gvis_cpu": gvis, "_gmap_ids": gids, "_cand_ids": h["vp_cand_ids"].copy()}

    def _nav_inputs_python(self, pano_embeds, pano_masks, cand_vpids):
        B, d, dev = self.B, self.cfg.hidden_size, self.device
        vpids, vis, steps, embeds = [], [], [], []
        for b, gmap in enumerate(self.gmaps):
            visited = [k for k in gmap.node_positions if self._visited(b, k)]
            unvisited = [k for k in gmap.node_positions if not self._visited(b, k)]
            gv = [None] + visited + unvisited                                  # enc_full_graph (configs/multi.yaml:109)
            vpids.append(gv)
            vis.append([0] + [1] * len(visited) + [0] * len(unvisited))
            steps.append([gmap.node_step_ids.get(v, 0) for v in gv])
            if not self.fast_maps:
                e = [gmap.get_node_embed(v) for v in gv[1:]]
                embeds.append(torch.stack([torch.zeros_like(e[0])] + e, 0))
        G = max(len(v) for v in vpids)
        gmask_np = np.zeros((B, G), dtype=bool)
        gvis_np = np.zeros((B, G), dtype=bool)
        gstep_np = np.zeros((B, G), dtype=np.int64)
        gmask, gvis, gstep = torch.from_numpy(gmask_np), torch.from_numpy(gvis_np), torch.from_numpy(gstep_np)   # views, filled below
        gpos = torch.zeros(B, G, 7)
        gpos_np = gpos.numpy()
        gids = np.full((B, G), -1, dtype=np.int32)                            # integer node ids for the side-car's match tables
        gimg = None if self.fast_maps else torch.zeros(B, G, d, device=dev)
        for b, gmap in enumerate(self.gmaps):
            n = len(vpids[b])
            gmask_np[b, :n] = True
            gvis_np[b, :n] = vis[b]
            gstep_np[b, :n] = steps[b]
            gmap.get_pos_fts(self.cur[b], vpids[b], self.heading[b], 0.0, out=gpos_np[b, :n])   # all slots, one C call, in place
            gids[b, :n] = gmap.node_ids(vpids[b])
            if not self.fast_maps:
                gimg[b, :n] = embeds[b]
        if self.fast_maps:
            # slot -> store row: node id + 1; the stop slot (id -1) and the padding (-1) -> row 0 = zeros / count 1
            R = self._store_rows
            rows = (gids.astype(np.int64) + 1) + (np.arange(B, dtype=np.int64) * R)[:, None]
            rows_t = ops.h2d(torch.from_numpy(rows.reshape(-1)), dev)
            # sum * (1 / count): what torch's `tensor / python_int` (graph_utils.py:137-142) computes on the device -- a division by
            # a host scalar is carried out as a multiplication by its fp32 reciprocal -- so the means are bit-identical to it
            inv = (1.0 / self.E_cnt.view(B * R).index_select(0, rows_t))[:, None]
            gimg = (self.E_sum.view(B * R, d).index_select(0, rows_t) * inv).view(B, G, d)
        Nv = pano_embeds.shape[1] + 1
        vp_img = torch.cat([torch.zeros_like(pano_embeds[:, :1]), pano_embeds], 1)
        pm = torch.cat([torch.ones_like(pano_masks[:, :1]), pano_masks], 1)
        vp_pos = np.zeros((B, Nv, 14), dtype=np.float32)
        cids = np.full((B, Nv), -1, dtype=np.int32)
        for b, gmap in enumerate(self.gmaps):
            cids[b, 1:len(cand_vpids[b]) + 1] = gmap.node_ids(cand_vpids[b])
            cf = gmap.get_pos_fts(self.cur[b], cand_vpids[b], self.heading[b], 0.0)
            sf = gmap.get_pos_fts(self.cur[b], [gmap.start_vp], self.heading[b], 0.0)
            vp_pos[b, :, :7] = sf
            vp_pos[b, 1:len(cf) + 1, 7:] = cf
        return {"gmap_vpids": vpids, "gmap_img_embeds": gimg, "gmap_step_ids": ops.h2d(gstep, dev), "gmap_pos_fts": ops.h2d(gpos, dev),
                "gmap_visited_masks": ops.h2d(gvis, dev), "gmap_masks": ops.h2d(gmask, dev), "vp_img_embeds": vp_img,
                "pano_masks": pm, "vp_pos_fts": ops.h2d(torch.from_numpy(vp_pos), dev),
                "vp_nav_masks": torch.ones(B, Nv, dtype=torch.bool, device=dev),
                "vp_cand_vpids": [[None] + x for x in cand_vpids], "hist_vis": self.hist_vis, "history": self.history,
                "data_type": ["r2r"] * B, "instruction": ["{INSTR}"] * B,
                "_gmask_cpu": gmask, "_gvis_cpu": gvis, "_gmap_ids": gids, "_cand_ids": cids}

    def tokenise(self, nav, cls_token):
        cand_nums = (nav["_gmask_cpu"] & ~nav["_gvis_cpu"]).sum(-1)
        seqs = []
        for b in range(self.B):
            p = navigation_prompt(self.task, "{INSTR}", len(self.history[b]), int(cand_nums[b]), cls_token)
            # the static prefix (task sentence + 512 instruction tokens + history header) is encoded once per episode; per step only
            # the rest of the prompt goes through the tokenizer (the stub splits on whitespace and the cut sits on one, so the
            # concatenation equals encode(p): tests/test_host_cpu.py)
            head = static_prefix(p)
            key = (b, head)
            if self._tok_cache.get("key%d" % b) != key:
                self._tok_cache["key%d" % b] = key
                self._tok_cache[b] = self.tok.encode(head, self.instr[b])
            seqs.append(self._tok_cache[b] + self.tok.encode(p[len(head):], self.instr[b])[1:])
        ids, am = self.tok.pad_left(seqs)
        self.S_hist.append(ids.shape[1])
        # a caller that tokenises itself tells the model how many leading tokens of each prompt never change inside the episode (the
        # reference's agents hand over prompt STRINGS instead and the model cuts at "### History:" itself): what an AUTOMATIC episode
        # registers as its prefix (NavModel._auto_prefix_ids)
        nav["prefix_lens"] = [len(self._tok_cache[b]) for b in range(self.B)]
        return ids, am

    def prefix_ids(self):
        """token ids of each episode's static prompt prefix (task sentence + instruction + history header), for
        `NavModel.begin_episode`"""
        out = []
        for b in range(self.B):
            p = static_prefix(navigation_prompt(self.task, "{INSTR}", 0, 1, "<cls_1>"))
            out.append(self.tok.encode(p, self.instr[b]))
        return out

    def teacher_targets(self, nav, last):
        """a random unvisited current candidate's map slot (0 = stop on the last step)."""
        tg = []
        for b in range(self.B):
            if last:
                tg.append(0)
                continue
            gv = nav["gmap_vpids"][b]
            opts = [gv.index(c["viewpointId"]) for c in self.obs[b]["candidate"]
                    if not self._visited(b, c["viewpointId"])]
            tg.append(opts[self.rng.randint(len(opts))] if opts else 0)
        return torch.tensor(tg, dtype=torch.int64)

    def advance(self, nav, actions, fuse_embeds):
        for b in range(self.B):
            a = int(actions[b])
            if a == -100:
                continue
            self.history[b].append("<hist>")
            self.hist_vis[b].append(fuse_embeds[b, a])
            if a > 0:
                self.cur[b] = nav["gmap_vpids"][b][a]
                self.heading[b] = float(self.rng.rand() * 2 * math.pi)
        self.t += 1
        self.obs = [self._observe(b) for b in range(self.B)]
        for b in range(self.B):
            self._update_graph(b)


def nav_step(model, criterion, ep, train=True, last=False, loss_weight=1.0, accum=1, final=None, feedback=None, temperature=1.0):
    """One iteration of the rollout loop for all B episodes. Returns (loss tensor or None, logits).
    feedback: how the next viewpoint is chosen (mp3d_agent.py:759-772): 'teacher' (default when training), 'argmax' (default
    otherwise) or 'sample' = one draw from Categorical(softmax(logits / temperature)), as the REVERIE / SOON evaluation scripts
    run it (`--do_sample --temperature 0.01`).
    last: the episode's last step (mp3d_agent.py:661-676: every earlier step runs inside `no_sync`).
    final: this step's backward is the LAST one before the optimizer step (default: `last`); a `NavDataParallel` wrapper then
    exchanges the accumulated gradients from inside it, overlapped (navillm_amd/parallel.py)."""
    inner = model.module if hasattr(model, "module") else model
    final = last if final is None else final
    ctx = contextlib.nullcontext
    if train and hasattr(model, "no_sync"):
        if not last:
            ctx = model.no_sync
        elif final and hasattr(model, "final_backward"):
            ctx = model.final_backward
    with ctx():
        pin = ep.panorama_inputs()
        pano = model("panorama", pin)
        pe, pm = pano["pano_embeds"], pano["pano_masks"]
        ep.update_maps(pe, pm, pin["cand_vpids"])
        nav = ep.nav_inputs(pe, pm, pin["cand_vpids"])
        ids, am = ep.tokenise(nav, inner.lang_model.cls_token[0])
        nav["input_ids"], nav["attention_mask"] = ids, am
        out = model("navigation", nav)
        logits = out["fuse_logits"]
        targets = ep.teacher_targets(nav, last)
        loss = None
        if train:
            # (`logits` is a losses.DeferredLogits handle inside a teacher-forced prefix-reuse episode: same two lines, the work happens
            # in finish_episode())
            loss = criterion(logits, ops.h2d(targets, getattr(logits, "device", inner.device))) * loss_weight / ep.B / accum
            loss.backward()
        feedback = feedback or ("teacher" if train else "argmax")
        if not train and os.environ.get("NAVILLM_PREFETCH_FEATURES", "1") != "0":
            ep.prefetch_features()                             # host work under the GPU's LM forward, before the wait for the action
        if feedback == "teacher":
            actions = targets
        elif feedback == "sample":
            probs = torch.softmax(logits.detach() / temperature, 1).float()
            actions = torch.distributions.Categorical(probs).sample().cpu()
        elif feedback == "argmax":
            actions = logits.detach().float().argmax(1).cpu()
        else:
            raise NotImplementedError(feedback)
        ep.advance(nav, actions, out["fuse_embeds"])
    return loss, logits


def reference_rollout(model, criterion, ep, steps, feedback="teacher", train_ml=1.0, accum=1, temperature=1.0, follow_teacher=False):
    """`MP3DAgent.rollout`'s training branch, call for call (tasks/agents/mp3d_agent.py:660-778), with the synthetic driver standing
    in for MatterSim: per step `model('panorama')`, the map update, `model('navigation')`, `torch.softmax(nav_logits / T, 1)` (:732 --
    the logits must be real tensors), `cnt_loss += criterion(...) * train_ml / batch_size / gradient_accumulation_step`, `ml_loss +=
    cnt_loss.detach()`, `cnt_loss.backward()` at once (:750-757), the action from the teacher or one draw from `Categorical(nav_probs)`
    (:759-769), the history append (:774-778); every step but the last inside `model.no_sync()` when the model is a
    DistributedDataParallel (:661-676).  NOTHING else: no begin_episode / finish_episode -- what an unmodified agent does to the model.
    -> ml_loss (a tensor; the caller reads `.item()`, train.py:83)"""
    ml_loss, cnt_loss = 0., 0.
    is_ddp = isinstance(model, torch.nn.parallel.DistributedDataParallel)
    inner = model.module if hasattr(model, "module") else model
    for t in range(steps):
        last = t == steps - 1
        context = model.no_sync if (is_ddp and not last) else contextlib.nullcontext
        with context():
            pin = ep.panorama_inputs()
            pano = model("panorama", pin)
            pe, pm = pano["pano_embeds"], pano["pano_masks"]
            ep.update_maps(pe, pm, pin["cand_vpids"])
            nav = ep.nav_inputs(pe, pm, pin["cand_vpids"])
            ids, am = ep.tokenise(nav, inner.lang_model.cls_token[0])
            nav["input_ids"], nav["attention_mask"] = ids, am
            nav_outs = model("navigation", nav)
            nav_logits = nav_outs["fuse_logits"]
            nav_probs = torch.softmax(nav_logits / temperature, 1)
            nav_targets = ep.teacher_targets(nav, last)
            cnt_loss += criterion(nav_logits, ops.h2d(nav_targets, nav_logits.device)) * train_ml / ep.B / accum
            ml_loss += cnt_loss.detach()
            cnt_loss.backward()
            cnt_loss = 0.
            if feedback == "teacher":
                a_t = nav_targets
            elif feedback == "sample":
                a_t = torch.distributions.Categorical(nav_probs.float()).sample().detach().cpu()
                if follow_teacher:          # (tests: the logits are READ as a sampled rollout reads them, the trajectory stays comparable)
                    a_t = nav_targets
            else:
                raise NotImplementedError(feedback)
            ep.advance(nav, a_t, nav_outs["fuse_embeds"])
    return ml_loss


def reference_train_steps(model, optimizer, criterion, ep, meta_steps, steps, accum=1, stage="pretrain", on_rollout=None, fused_clip=False,
                          after_rollout=None):
    """`train_one_epoch` (train.py:60-91) around `MP3DAgent.train` (mp3d_agent.py:497-527), verbatim in what it does to the model and the
    optimizer: per meta-step one rollout (`stage == 'pretrain'` or every even step: teacher forcing, else DAgger sampling), `loss.item()`
    for the running metrics, and every `accum` meta-steps `torch.nn.utils.clip_grad_norm_(model.parameters(), 40.)`, `optimizer.step()`,
    `optimizer.zero_grad()`.  `ep.reset()` draws the next batch of episodes.  fused_clip: INTEGRATION.md's edit 3 (`optimizer.clip_grad_norm_(40.)`,
    the clip folded into FlatAdamW's update kernel) instead of torch's clip.  on_rollout / after_rollout(step): hooks around a rollout
    (tests and the bench wrap the SAME loop in explicit begin_episode / finish_episode calls for the comparison).
    -> the per-meta-step loss values"""
    losses = []
    for step in range(meta_steps):
        ep.reset()
        if on_rollout is not None:
            on_rollout(step)
        fb = "teacher" if (stage == "pretrain" or step % 2 == 0) else "sample"
        loss = reference_rollout(model, criterion, ep, steps, feedback=fb, accum=accum) * accum
        if after_rollout is not None:
            after_rollout(step)
        losses.append(loss.item())
        if (step + 1) % accum == 0:
            if fused_clip:
                optimizer.clip_grad_norm_(40.)
            else:
                torch.nn.utils.clip_grad_norm_(model.parameters(), 40.)
            optimizer.step()
            optimizer.zero_grad()
    return losses


# --------------------------------------------------------------------------- inference: two episode batches in flight
# A validation rollout is a strict chain per batch -- action t decides observation t+1 -- and per step the host builds inputs for
# ~16 ms (maps, prompts, index tables; the reference's agent does the same work in Python) before the GPU gets ~20 ms of decoder
# work: run alone, the two alternate and each idles half the time (tools/kv_trace.py: GPU busy 50 %).  Two INDEPENDENT batches
# hide each other's host phase: while the GPU runs batch A's step, the host prepares batch B's.  Nothing changes per batch
# (same kernels, same batch size per forward, same trajectories); each batch has its own K/V cache.
def nav_step_launch(model, ep, kv=None, feedback="argmax", temperature=1.0):
    """enqueue one no-grad navigation step of `ep` and return a handle; never waits for the GPU.  `kv`: this batch's KVCacheLM."""
    inner = model.module if hasattr(model, "module") else model
    if kv is not None:
        inner.kv = kv if kv is not False else None           # False: full recompute (no K/V reuse)
    with torch.no_grad():
        pin = ep.panorama_inputs()
        pano = model("panorama", pin)
        pe, pm = pano["pano_embeds"], pano["pano_masks"]
        ep.update_maps(pe, pm, pin["cand_vpids"])
        nav = ep.nav_inputs(pe, pm, pin["cand_vpids"])
        ids, am = ep.tokenise(nav, inner.lang_model.cls_token[0])
        nav["input_ids"], nav["attention_mask"] = ids, am
        out = model("navigation", nav)
        logits = out["fuse_logits"]
        ep.teacher_targets(nav, False)                       # keeps the episode's RNG stream identical to nav_step's
        if feedback == "argmax":
            act = logits.float().argmax(1)
        elif feedback == "sample":
            act = torch.distributions.Categorical(torch.softmax(logits / temperature, 1).float()).sample()
        else:
            raise NotImplementedError(feedback)
        ring = ep.__dict__.setdefault("_act_ring", [torch.empty(act.shape, dtype=act.dtype).pin_memory() for _ in range(2)])
        host = ring[ep.t & 1]                                 # pinned landing buffers, allocated once (two: a step's actions are read
        host.copy_(act, non_blocking=True)                    # after the next step of the OTHER batch was launched, never later)
        ev = torch.cuda.Event()
        ev.record()
    return {"nav": nav, "fuse_embeds": out["fuse_embeds"], "actions": host, "event": ev, "logits": logits}


def nav_step_finish(ep, h):
    """wait for the step's actions (only), then move the episodes on"""
    h["event"].synchronize()
    ep.advance(h["nav"], h["actions"], h["fuse_embeds"])


def rollout_interleaved(model, eps, kvs, steps, feedback="argmax", temperature=1.0):
    """`steps` navigation steps of every batch in `eps` (each with its K/V cache in `kvs`), software-pipelined: the host
    prepares batch i+1 while the GPU runs batch i.  Returns the logits of the last step per batch."""
    pending = [None] * len(eps)
    last = [None] * len(eps)
    for t in range(steps):
        for i, ep in enumerate(eps):
            if pending[i] is not None:
                nav_step_finish(ep, pending[i])
            pending[i] = nav_step_launch(model, ep, kvs[i], feedback, temperature)
            last[i] = pending[i]["logits"]
    for i, ep in enumerate(eps):
        nav_step_finish(ep, pending[i])
    return last


# --------------------------------------------------------------------------- the other training sub-tasks of a rollout
def _sync_ctx(model, sync):
    """sync: 'no_sync' (a non-last step), 'plain' (synced, but more backwards follow before the optimizer step) or 'final'"""
    if sync == "no_sync" and hasattr(model, "no_sync"):
        return model.no_sync
    if sync == "final" and hasattr(model, "final_backward"):
        return model.final_backward
    return contextlib.nullcontext


def og_step(model, criterion, ep, train=True, sync="plain", coef=1.0, accum=1, objects=(5, 40)):
    """Object-prediction sub-task of an episode's last step (SOON / REVERIE, mp3d_agent.py:788-842): panorama with objects ->
    model('object_grounding') -> CE_sum * obj_loss_coef / B / accum -> backward.  Returns (loss, obj_logits)."""
    with _sync_ctx(model, sync if train else "plain")():
        pin = ep.panorama_inputs(objects=objects)
        pano = model("panorama", pin)
        lens = pin["obj_lens"]
        seqs = [ep.tok.encode(object_grounding_prompt(ep.task, "{INSTR}", len(ep.history[b]), int(lens[b]) + 1, "<cls_1>"), ep.instr[b])
                for b in range(ep.B)]
        ids, am = ep.tok.pad_left(seqs)
        batch = {"obj_embeds": pano["obj_embeds"], "obj_masks": pano["obj_masks"], "obj_loc_fts": pano["obj_loc_fts"],
                 "hist_vis": ep.hist_vis, "history": ep.history, "input_ids": ids, "attention_mask": am}
        logits = model("object_grounding", batch)["obj_logits"]
        # teacher_object (mp3d_agent.py:458-472): the target object's index + 1, or ignore
        tg = torch.tensor([int(ep.rng.randint(1, int(lens[b]) + 1)) if ep.rng.rand() < 0.8 else -100 for b in range(ep.B)])
        loss = None
        if train:
            loss = criterion(logits, ops.h2d(tg, logits.device)) * coef / ep.B / accum
            loss.backward()
    return loss, logits


def lm_aux_step(model, ep, mode, train=True, sync="plain", coef=1.0, accum=1, answer_len=24, **gen):
    """The LM-loss sub-tasks of a rollout: 'summarization' (last step, mp3d_agent.py:871-909: the label is the episode's
    instruction) and 'embodied_qa' (fine-grained R2R on a non-last step, :845-868: a short answer, no history).  12-view
    panorama -> vp tokens -> model(mode, training=True) -> loss * gen_loss_coef / B / accum -> backward."""
    B = ep.B
    with _sync_ctx(model, sync if train else "plain")():
        pin = ep.panorama_inputs(first12=True)
        pano = model("panorama", pin)
        pe = pano["pano_embeds"]
        vp_img = torch.cat([torch.zeros_like(pe[:, :1]), pe], 1)
        nav_masks = torch.cat([torch.ones(B, 1, dtype=torch.bool), (pin["nav_types"] == 1).cpu()], 1)
        cn = int(nav_masks[0, 1:].sum())
        seqs, types = [], []
        for b in range(B):
            if mode == "summarization":
                prompt, hv, ans = summarization_prompt(ep.task, "", len(ep.history[b]), cn), ep.hist_vis, ep.instr[b]
            else:
                prompt, hv = embodied_qa_prompt("r2r", "where are we going with direction (1) ?", 0, cn), [[] for _ in range(B)]
                ans = ep.rng.randint(3, ep.cfg.base_vocab_size, size=answer_len).tolist()
            s_, t_ = ep.tok.encode_pair(prompt, ans)
            seqs.append(s_)
            types.append(t_)
        ids, am, tt = ep.tok.pad_left(seqs, types=types)
        batch = {"vp_img_embeds": vp_img, "vp_nav_masks": nav_masks, "hist_vis": hv, "data_type": [ep.task if mode == "summarization" else "fgr2r"] * B,
                 "input_ids": ids, "attention_mask": am, "token_type_ids": tt, "instruction": [""] * B, "answer": [""] * B}
        out = model(mode, batch, training=train, **gen)
        loss = None
        if train:
            loss = out["loss"] * coef / B / accum
            loss.backward()
    return loss, out


def qa_step(model, ep, train=True, sync="final", coef=1.0, accum=1, answer_len=8, n_rows=1, **gen):
    """ScanQA / LLaVA batch (tasks/agents/llava.py:19-42): one feature row per `<cand>`, question -> answer, loss * coef / accum."""
    B, F, cfg = ep.B, ep.cfg.image_feat_size, ep.cfg
    with _sync_ctx(model, sync if train else "plain")():
        feats = [ops.h2d(torch.randn(n_rows, F, generator=ep.tgen), ep.device) for _ in range(B)]
        seqs, types = [], []
        for b in range(B):
            q = " ".join("w%d" % int(w) for w in ep.rng.randint(0, 500, size=int(ep.rng.randint(6, 14))))
            prompt = qa3d_prompt(q) if n_rows == 1 else " ".join(["<cand>"] * n_rows) + " ### Question: " + q + " ### Answer: "
            if train:
                s_, t_ = ep.tok.encode_pair(prompt, ep.rng.randint(3, cfg.base_vocab_size, size=answer_len).tolist())
            else:                                                   # generation: the prompt only
                s_ = ep.tok.encode(prompt)
                t_ = [0] * len(s_)
            seqs.append(s_)
            types.append(t_)
        ids, am, tt = ep.tok.pad_left(seqs, types=types)
        out = model("3dqa", {"features": feats, "question": [""] * B, "input_ids": ids, "attention_mask": am, "token_type_ids": tt},
                    training=train, **gen)
        loss = None
        if train:
            loss = out.loss * coef / accum
            loss.backward()
    return loss, out


def mixed_task_episode(model, criterion, ep, steps, enable_og=None, enable_summarize=True, enable_fgr2r=True, accum=1, train=True,
                       prefix_reuse=False, teacher_forced=False):
    """One training meta-step of the multi-task mix (BASELINE config 3; rollout of tasks/agents/mp3d_agent.py:593-964 for the
    task `ep.task`): `steps` navigation steps, each with its backward inside `no_sync` except the last; fine-grained R2R
    (embodied_qa + LM loss) on the non-last steps of an R2R episode; on the last step the object-grounding sub-task (SOON /
    REVERIE) and the summarization sub-task (not CVDN), each with its own backward -- the last of them is the `final` one a
    data-parallel wrapper exchanges gradients from.  Returns the losses of the episode.
    prefix_reuse: the NAVIGATION steps run over the episode's cached prompt prefix (navillm_amd/episode.py); the sub-tasks have
    prompts of their own and go through the whole LM as before; the prefix's deferred backward then is the episode's last one."""
    if enable_og is None:
        enable_og = ep.task in ("soon", "reverie")
    enable_summarize = enable_summarize and ep.task != "cvdn"
    enable_fgr2r = enable_fgr2r and ep.task == "r2r"
    inner = model.module if hasattr(model, "module") else model
    prefix_reuse = prefix_reuse and train
    if prefix_reuse:
        # teacher_forced: the navigation steps' LM forward is deferred and batched into finish_episode() too (the sub-tasks' prompts
        # still go through the whole LM when they are called)
        inner.begin_episode(ep.prefix_ids(), teacher_forced=teacher_forced)
    losses = {"nav": [], "og": None, "sum": None, "fgr2r": []}
    for t in range(steps):
        last = t == steps - 1
        more = last and (enable_og or enable_summarize)
        l, _ = nav_step(model, criterion, ep, train=train, last=last, accum=accum, final=last and not more and not prefix_reuse)
        losses["nav"].append(l)
        if enable_fgr2r and not last and t % 2 == 0 and train:
            losses["fgr2r"].append(lm_aux_step(model, ep, "embodied_qa", sync="no_sync", accum=accum)[0])
        if last and enable_og and train:
            losses["og"] = og_step(model, criterion, ep, sync="plain" if (enable_summarize or prefix_reuse) else "final", accum=accum)[0]
        if last and enable_summarize and train:
            losses["sum"] = lm_aux_step(model, ep, "summarization", sync="plain" if prefix_reuse else "final", accum=accum)[0]
    if prefix_reuse:
        ctx = model.final_backward if hasattr(model, "final_backward") else contextlib.nullcontext
        with ctx():
            inner.finish_episode()
    return losses


def prefix_reuse_episode(model, criterion, ep, steps, accum=1, teacher_forced=False):
    """One training episode with the prompt's static prefix computed once (navillm_amd/episode.py): `begin_episode`, `steps`
    navigation steps whose per-step backward()s cover the suffix rows only (all inside the accumulation phase of a data-parallel
    wrapper), then the prefix's single backward -- the LAST backward before the optimizer step, from inside which a
    `NavDataParallel` wrapper exchanges the gradients, layer by layer."""
    inner = model.module if hasattr(model, "module") else model
    inner.begin_episode(ep.prefix_ids(), teacher_forced=teacher_forced)
    losses = []
    for t in range(steps):
        losses.append(nav_step(model, criterion, ep, train=True, last=(t == steps - 1), accum=accum, final=False)[0])
    ctx = model.final_backward if hasattr(model, "final_backward") else contextlib.nullcontext
    with ctx():
        inner.finish_episode()
    return losses


def accumulation_window(model, criterion, episodes, steps):
    """One accumulation window of teacher-forced prefix-reuse episodes (`begin_episode(..., accumulate=n)`; the reference's
    `--gradient_accumulation_step n`, train.py:68,86-89): `episodes` = n SyntheticEpisodes batches, `steps[e]` navigation steps each,
    every loss scaled / B / n.  The n-th `finish_episode()` runs the whole window; under a `NavDataParallel` wrapper it is the one inside
    `final_backward()`, so the exchange is launched from inside the window's batched backward."""
    inner = model.module if hasattr(model, "module") else model
    n = len(episodes)
    losses = []
    for e, ep in enumerate(episodes):
        inner.begin_episode(ep.prefix_ids(), teacher_forced=True, accumulate=n)
        for t in range(steps[e]):
            losses.append(nav_step(model, criterion, ep, train=True, last=(t == steps[e] - 1), accum=n, final=False)[0])
        last = e == n - 1
        ctx = model.final_backward if (last and hasattr(model, "final_backward")) else contextlib.nullcontext
        with ctx():
            inner.finish_episode()
    return losses
