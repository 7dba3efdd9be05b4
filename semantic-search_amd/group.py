"""`MapperGroup`: N `SemanticMapper`-compatible views over ONE batched `MapEngine`.

The reference builds one mapper per environment and one environment per OS process
(eval_objnav.py:144, train_objnav.py:85).  A caller that keeps its environments in one process
(threads, or a plain loop) gets the batched rate through this class without touching env.py:
every view has the reference's interface (`reset`, `update_map`, the public arrays, the
late-mutable attributes), and the `update_map` calls of a round are collected and stepped as one
engine batch.

    group = MapperGroup(n, num_semantic_classes=9, depth_hfov=59, depth_range=(0.5, 5.0), ...)
    mappers = group.mappers                    # hand mappers[i] to environment i as its `mapping`
    # a) one thread per environment: mappers[i].update_map(obs, aux) blocks until every expected
    #    view has called (or `round_timeout_s` passed, or someone called group.flush()), then all
    #    return their own results
    # b) a single-threaded loop: results = group.update_maps([obs_0, None, obs_2, ...])

A round is "the next call of every view": the result of a call depends on that view's own call
sequence only, so views may be in different phases of their episodes (one resets and maps its
first frame while the others step).  Views without a frame in a round are skipped by the engine
(`skip` of ssmap_frames): their maps, trail and hit/view overlay stay as they are.

All views share the engine's rule parameters (`entropy_threshold`, `hv_ratio`, ...,
`found_dist_m`, `filter_goal_prob`) and schema table: they are read from the first view of the
round and must be equal on the others.
"""
from __future__ import annotations

import threading
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import engine as _engine
from .mapper import SemanticMapper, _RULE_ATTRS

_IDLE_POSE = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])


class GroupMapper(SemanticMapper):
    """One view of a MapperGroup: the reference's interface for environment `index`."""

    def __init__(self, group: "MapperGroup", index: int, **kwargs):
        super().__init__(**kwargs)
        self._group = group
        self._e = index

    def _ensure_engine(self, h, w):   # the group owns the engine
        raise RuntimeError("a view of a MapperGroup is stepped by its group")

    def reset(self, map_bounds: np.ndarray):
        with self._group._lock:
            self._reset_state(map_bounds)
            self._engine = self._group._engine
            self._bound = False
            if self._engine is not None:
                self._group._bind(self)

    def _reset_state(self, map_bounds):
        if self.collect_nb_data:
            self.eps_nb_data = {"goal": [], "non_goal": [], "goal_id": None}   # mapper.py:265
        self.map_bounds = np.asarray(map_bounds)
        self._host_cache = {}
        self.prev_pos_px = None
        self.found_idx = None
        self.path_mask = np.zeros(_engine.map_size(self.map_bounds, self.pixel2meter), dtype=np.uint8)

    def update_map(self, obs, auxillary_angle=None):
        return self._group._submit(self._e, obs, auxillary_angle)


class MapperGroup:
    def __init__(self, n_envs: int, max_map_hw: Optional[Sequence[int]] = None,
                 round_timeout_s: Optional[float] = None, **mapper_kwargs):
        """`mapper_kwargs`: the reference constructor's keywords (mapper.py:49-72), the same for
        every view.  `max_map_hw`: largest map (rows, cols) any episode will need; by default the
        largest among the views' bounds when the engine is first built (a later `reset` with a
        larger map raises).  `round_timeout_s`: how long a blocked `update_map` waits for the other
        views before the round runs without them (None: until all have called or `flush`)."""
        if n_envs < 1:
            raise ValueError("n_envs must be positive")
        self.n_envs = int(n_envs)
        self.max_map_hw = None if max_map_hw is None else (int(max_map_hw[0]), int(max_map_hw[1]))
        self.round_timeout_s = round_timeout_s
        self._lock = threading.RLock()
        self._cond = threading.Condition(self._lock)
        self._engine: Optional[_engine.MapEngine] = None
        self._engine_hw = None
        self._schema_sent = False
        self._buf: Dict[str, torch.Tensor] = {}
        self._pending: Dict[int, tuple] = {}
        self._done: Dict[int, object] = {}
        self.expected = set(range(self.n_envs))   # views whose calls a blocked round waits for
        self.mappers: List[GroupMapper] = [GroupMapper(self, i, **mapper_kwargs) for i in range(self.n_envs)]

    # ------------------------------------------------------------------ engine plumbing
    def _bind(self, view: GroupMapper):
        eng = self._engine
        rows, cols = _engine.map_size(view.map_bounds, view.pixel2meter)
        if rows > eng.max_map_hw[0] or cols > eng.max_map_hw[1]:
            raise ValueError(f"map of {rows}x{cols} cells exceeds the group's max_map_hw {eng.max_map_hw}: "
                             "construct the MapperGroup with a larger max_map_hw")
        eng.reset_env(view._e, view.map_bounds)
        view._engine = eng
        view._bound = True

    def _ensure_engine(self, h: int, w: int):
        first = self.mappers[0]
        if self._engine is not None and self._engine_hw != (h, w):
            raise ValueError(f"frames of {h}x{w} after frames of {self._engine_hw}: one frame size per group")
        if self._engine is None:
            cap = self.max_map_hw
            if cap is None:
                sizes = [_engine.map_size(m.map_bounds, m.pixel2meter) for m in self.mappers if m.map_bounds is not None]
                cap = (max(s[0] for s in sizes), max(s[1] for s in sizes))
            eng = _engine.MapEngine(
                self.n_envs, (h, w), num_semantic_classes=first.num_semantic_classes, depth_hfov=first.depth_hfov,
                depth_range=first.depth_range, pixel2meter=first.pixel2meter, height_cutoff_m=first.height_cutoff_m,
                probablistic_mapping=first.probablistic_mapping, prob_aggregate=first.prob_aggregate,
                seg_model_n_classes=first.sem_model_n_classes, max_map_hw=cap, success_dist_m=first.found_dist_m,
                collect_nb_data=first.collect_nb_data, device=first._device)
            for goal, model in (first.nb_models or {}).items():
                eng.set_nb_model(int(goal), model)
            eng.set_egomap(first._ego_cfg["egomap_small_side_m"], first._ego_cfg["egomap_large_side_m"],
                           first.egomap_size, first._ego_cfg["agent_draw_radius_m"], first.aux_bin_number,
                           first._ego_cfg["aux_radius_m"], first._ego_cfg["aux_bin_m"], first.offset_for_cut)
            self._engine, self._engine_hw = eng, (h, w)
            self._schema_sent = False
            self._buf = {}
        eng = self._engine
        # every env of the engine needs maps before a step: a view that was never reset borrows the
        # bounds of one that was (its own reset replaces them)
        some = next(m.map_bounds for m in self.mappers if m.map_bounds is not None)
        for m in self.mappers:
            if not m._bound or m._engine is not eng:
                if m.map_bounds is None:
                    eng.reset_env(m._e, some)
                    m._engine = eng
                    m._bound = True   # placeholder maps until the view's own reset
                else:
                    self._bind(m)
        return eng

    def _send_schema(self, eng, view):
        if self._schema_sent or not view.sem_model_n_classes or view.map_seg_to_schema is None:
            return
        lut = np.asarray(view.map_seg_to_schema(np.arange(view.sem_model_n_classes)))
        pools = None
        if view.objnav_to_sem_pred is not None:
            pools = {g: list(view.objnav_to_sem_pred[g]) for g in range(3, 9) if g in view.objnav_to_sem_pred}
        eng.set_schema(lut, pools)
        self._schema_sent = True

    def _batch(self, name, like: torch.Tensor):
        buf = self._buf.get(name)
        if buf is None:
            buf = torch.zeros((self.n_envs,) + tuple(like.shape), dtype=like.dtype, device=like.device)
            self._buf[name] = buf
        return buf

    # ------------------------------------------------------------------ one round
    def _round(self, calls: Dict[int, tuple]) -> Dict[int, object]:
        """Steps the views in `calls` ({index: (obs, aux)}) as one engine batch; returns
        {index: (ego maps, is_found)} or {index: exception} for a view whose frame raised."""
        if not calls:
            return {}
        order = sorted(calls)
        lead = self.mappers[order[0]]
        for i in order:
            if self.mappers[i].map_bounds is None:
                raise RuntimeError(f"update_map of view {i} before its reset")
        h, w = calls[order[0]][0]["depth"].shape
        eng = self._ensure_engine(h, w)
        self._send_schema(eng, lead)
        lead._sync_params(eng)
        for i in order[1:]:
            for name in _RULE_ATTRS + ("filter_goal_prob", "found_dist_m"):
                if hasattr(lead, name) and getattr(self.mappers[i], name, None) != getattr(lead, name):
                    raise ValueError(f"views {order[0]} and {i} disagree on {name}: the views of a group share "
                                     "the engine's rule parameters")
        E = self.n_envs
        skip = np.ones(E, dtype=np.uint8)
        agent = np.tile(_IDLE_POSE, (E, 1))
        camera = np.tile(_IDLE_POSE, (E, 1))
        goals = np.full(E, 3, dtype=np.int32)
        names = None
        for i in order:
            obs = calls[i][0]
            tens, goal_id = self.mappers[i]._inputs(obs, eng.device)
            present = tuple(k for k, t in tens.items() if t is not None)
            if names is None:
                names = present
            elif present != names:
                raise ValueError(f"view {i} passes observation keys {present}, view {order[0]} {names}")
            for k in names:
                self._batch(k, tens[k])[i].copy_(tens[k])
            skip[i] = 0
            agent[i] = np.asarray(obs["agent_position_and_rotation"], dtype=np.float64)
            camera[i] = np.asarray(obs["camera_position_and_rotation"], dtype=np.float64)
            goals[i] = goal_id
        kwargs = {k: self._buf[k] for k in names if k != "depth"}
        eng.step(self._buf["depth"], agent, camera, goals, skip=skip if skip.any() else None, **kwargs)
        found, status, n_upd = eng.results()
        out: Dict[int, object] = {}
        paint = False
        for i in order:
            try:
                is_found, p = self.mappers[i]._after_step(eng, calls[i][0], self._buf["depth"][i], found[i],
                                                          status[i], n_upd[i], int(goals[i]))
                out[i] = is_found
                paint |= p
            except IndexError as exc:   # this view's frame only (mapper.py:338)
                out[i] = exc
        if paint:
            eng.path_paint()
        auxes = [calls[i][1] for i in order]
        aux = None
        if any(a is not None for a in auxes):
            if any(a is None for a in auxes):
                raise ValueError("some views pass an auxiliary angle vector and some None in one round")
            aux = np.zeros((E, len(np.asarray(auxes[0]))), dtype=np.float64)
            for i, a in zip(order, auxes):
                aux[i] = np.asarray(a, dtype=np.float64)
        small, large, unc_s, unc_l = eng.ego_maps(agent, goals, aux)
        small, large = small.cpu().numpy(), large.cpu().numpy()
        unc_s = None if unc_s is None else unc_s.cpu().numpy()
        unc_l = None if unc_l is None else unc_l.cpu().numpy()
        for i in order:
            if not isinstance(out[i], Exception):
                out[i] = ((small[i], large[i], None if unc_s is None else unc_s[i],
                           None if unc_l is None else unc_l[i]), out[i])
        return out

    # ------------------------------------------------------------------ single-threaded form
    def update_maps(self, obs_list, aux_list=None):
        """One round from a single thread: obs_list[i] is view i's `obs` or None (no frame for it).
        Returns a list with ((ego_small, ego_large, unc_small, unc_large), is_found) per view, None
        where there was no frame; raises the first IndexError a frame raised."""
        if len(obs_list) != self.n_envs:
            raise ValueError("one entry per view")
        calls = {i: (obs, None if aux_list is None else aux_list[i]) for i, obs in enumerate(obs_list) if obs is not None}
        with self._lock:
            res = self._round(calls)
        for i in sorted(res):
            if isinstance(res[i], Exception):
                raise res[i]
        return [res.get(i) for i in range(self.n_envs)]

    # ------------------------------------------------------------------ threaded form
    def _run_pending(self):
        calls, self._pending = self._pending, {}
        try:
            res = self._round(calls)
        except Exception as exc:   # the whole round failed: every caller of it hears about it
            res = {i: exc for i in calls}
        self._done.update(res)
        self._cond.notify_all()

    def _submit(self, index, obs, aux):
        with self._cond:
            if index in self._pending or index in self._done:
                raise RuntimeError(f"view {index} called update_map twice in one round")
            self._pending[index] = (obs, aux)
            if self.expected <= set(self._pending):
                self._run_pending()
            else:
                waited = self._cond.wait_for(lambda: index in self._done, timeout=self.round_timeout_s)
                if not waited and index in self._pending:
                    self._run_pending()   # the others did not come: the round runs with those who did
            res = self._done.pop(index)
        if isinstance(res, Exception):
            raise res
        return res

    def flush(self):
        """Runs the round with the views that are waiting now (from any thread)."""
        with self._cond:
            if self._pending:
                self._run_pending()

    def close(self):
        with self._lock:
            if self._engine is not None:
                self._engine.close()
                self._engine = None
            for m in self.mappers:
                m._engine = None
                m._bound = False
