"""Fused M2-Mixer training engines: the whole `shared_step` + backward + Adam of the reference's multi-loss
LightningModules as a fixed sequence of libm2mixer launches over flat fp32 parameter / gradient / Adam-state buffers.

  AVMnistEngine   AVMnistMixerMultiLoss   models/avmnist.py:236-312, :413-422   (two patch towers, CE, total x3)
  MMIMDBEngine    MMIMDBMixerMultiLoss    models/mmimdb.py:96-147               (two patch towers, BCE(pos_weight), plain sum)
  MimicEngine     MimicMixerMultiLoss     models/mimic.py:93-142                (static MLP + time tower, CE, no x3)

What changes relative to the module path (modules/mixer.py + torch autograd):
  * parameters are views into ONE flat buffer (names = the reference's state-dict keys, in the reference's creation
    order), gradients into ONE flat buffer -> no memset (Adam clears what it consumes), one RCCL all-reduce;
  * the two modality towers write their outputs straight into their parts of the fused (B, N1+N2, D) buffer
    (ConcatFusion costs nothing) and hand the token means to the heads kernel;
  * the three heads, their losses and gradients are one launch;
  * each tower's Adam update + operand re-pack follows its own weight gradients on its own HIP stream;
  * the step can be captured into a hipGraph (torch.cuda.CUDAGraph); the dropout step counter, the Adam
    step counter and the learning rate live in device memory so replays stay correct.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import os

import torch

from . import _lib as L
from . import config
from .runtime import (AdamPackPlan, BLOCK_FIELDS, BLOCK_KEYS, EmbedRuntime, MlpRuntime, TowerRuntime, block_param_shapes, heads_bce,
                      heads_ce, can_group, can_group_embeds, can_pack_all, embeds_forward, pack_all,
                      towers_backward, towers_forward, towers_forward_embeds_ok, towers_wgrad, wgrad_slot_groups)


def config_fused_update(n_params: int = 0) -> bool:
    """Adam and the operand re-pack as ONE launch (m2m_adam_pack_all) or as two (flat Adam, then m2m_pack_all).
    M2M_FUSED_UPDATE=1 / 0 forces either; default: one launch.
    Measured (round 4): with 32-column-group tiles the one-launch form read W2 in 128-byte row segments 12 KB apart (M2-Mixer-B:
    70-78 us against 60-65 us as two launches); with W2 in 8-row x 512-column tiles (2 KiB runs) and the moment streams
    non-temporal it takes 55-56 us and the step 0.4856 against 0.4912 ms (profiles/r04_ab_results.txt r5l-r5n); MM-IMDb
    (2.7 M parameters, latency-bound launches) 0.4602 against 0.4637 ms per step."""
    env = os.environ.get("M2M_FUSED_UPDATE")
    if env is not None:
        return env == "1"
    return n_params > 0


def _num_patch(c: dict) -> int:
    return (c["image_size"][0] // c["patch_size"]) * (c["image_size"][1] // c["patch_size"])


def _tower_shapes(s: "OrderedDict[str, tuple]", prefix: str, c: dict, N: int, embed: Optional[str]):
    D = c["hidden_dim"]
    if embed == "patch":
        s[prefix + "to_patch_embedding.0.weight"] = (D, c["in_channels"], c["patch_size"], c["patch_size"])
        s[prefix + "to_patch_embedding.0.bias"] = (D,)
    elif embed == "proj":
        s[prefix + "proj.weight"] = (c["proj_dim"], c["embedding_dim"])
        s[prefix + "proj.bias"] = (c["proj_dim"],)
    shapes = block_param_shapes(D, N, c["token_dim"], c["channel_dim"])
    for i in range(c["num_mixers"]):
        for f in BLOCK_FIELDS:
            s[f"{prefix}mixer_blocks.{i}.{BLOCK_KEYS[f]}"] = shapes[f]
    s[prefix + "layer_norm.weight"] = (D,)
    s[prefix + "layer_norm.bias"] = (D,)


def _head_shapes(s, names: Sequence[str], dims: Sequence[int], K: int):
    for n, d in zip(names, dims):
        key = "classifier_fusion.classifer." if n == "fusion" else f"classifier_{n}."     # sic: classification.py:87
        s[key + "weight"] = (K, d)
        s[key + "bias"] = (K,)


def two_tower_param_shapes(cfg: dict, mods: Tuple[str, str]) -> "OrderedDict[str, tuple]":
    """state-dict key -> shape in the reference's creation order (models/avmnist.py:181-191, models/mmimdb.py:35-45)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    a, b = mods
    na, nb = _num_patch(cfg[a]), _num_patch(cfg[b])
    _tower_shapes(s, f"{a}_mixer.", cfg[a], na, "patch")
    _tower_shapes(s, f"{b}_mixer.", cfg[b], nb, "patch")
    _tower_shapes(s, "fusion_mixer.", cfg["multimodal"], na + nb, None)
    _head_shapes(s, (a, b, "fusion"), (cfg[a]["hidden_dim"], cfg[b]["hidden_dim"], cfg["multimodal"]["hidden_dim"]),
                 cfg["num_classes"])
    return s


def avmnist_param_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    return two_tower_param_shapes(cfg, ("image", "audio"))


def mimic_param_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    """Creation order of models/mimic.py:39-49: time_mixer, static_extractor, fusion_mixer, three classifiers."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    t, st = cfg["time"], cfg["static"]
    _tower_shapes(s, "time_mixer.", t, t["num_patch"], "proj")
    for i in range(st["num_blocks"]):
        s[f"static_extractor.module_list.{3 * i}.weight"] = (st["hidden_dim"], st["input_dim"] if i == 0 else st["hidden_dim"])
        s[f"static_extractor.module_list.{3 * i}.bias"] = (st["hidden_dim"],)
    k = 3 * st["num_blocks"]
    s[f"static_extractor.module_list.{k}.weight"] = (st["output_dim"], st["hidden_dim"])
    s[f"static_extractor.module_list.{k}.bias"] = (st["output_dim"],)
    _tower_shapes(s, "fusion_mixer.", cfg["multimodal"], 1 + t["num_patch"], None)
    _head_shapes(s, ("static", "time", "fusion"), (st["output_dim"], t["hidden_dim"], cfg["multimodal"]["hidden_dim"]),
                 cfg["num_classes"])
    return s


def _is_layer_norm(k: str) -> bool:
    return ("layer_norm." in k) or k.endswith("token_mix.0.weight") or k.endswith("token_mix.0.bias") \
        or k.endswith("channel_mix.0.weight") or k.endswith("channel_mix.0.bias")


class _FlatEngine:
    """Flat parameter / gradient / Adam buffers, the optimizer, data-parallel hooks and hipGraph capture.
    Subclasses supply `shapes`, `_segment_of(key)`, `_build()`, `_forward(...)`, `_backward(...)`, `pack()`."""

    SEGMENTS: Tuple[str, ...] = ()

    def __init__(self, cfg: dict, batch_size: int, device, precision, lr, betas, eps, weight_decay, seed, init, share=None):
        """share: another engine of the same model; this one then works on ITS parameters, gradients, Adam state and
        step counters (its own activation buffers and packed operand copies, for another batch size) -- the second,
        smaller step that takes the ragged last batch of an epoch (data.run_epoch).  After a step on one of the two
        the other one's packed copies are stale: call its pack() before using it."""
        self.cfg, self.B = cfg, int(batch_size)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"{type(self).__name__} runs on the GPU only (MI355X); there is no CPU path")
        L.lib()  # fail loudly if the HIP library is absent
        self.prec = config.prec_id(precision)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.p_drop = float(cfg.get("dropout", 0.0))
        self.K = cfg["num_classes"]
        self.shapes = self._param_shapes(cfg)
        n = sum(int(torch.Size(s).numel()) for s in self.shapes.values())
        self.n_params = n
        self._fused_update = config_fused_update(n)          # (read once: the environment at construction decides)
        dev = self.device
        if share is not None:
            if share.n_params != n or list(share.shapes.items()) != list(self.shapes.items()) or share.device != dev:
                raise RuntimeError("share=: the other engine must hold the same model on the same device")
            self.flat_p, self.flat_g, self.flat_m, self.flat_v = share.flat_p, share.flat_g, share.flat_m, share.flat_v
        else:
            self.flat_p = torch.zeros(n, device=dev)
            self.flat_g = torch.zeros(n, device=dev)
            self.flat_m = torch.zeros(n, device=dev)
            self.flat_v = torch.zeros(n, device=dev)
        self.params: Dict[str, torch.Tensor] = OrderedDict()
        self.grads: Dict[str, torch.Tensor] = OrderedDict()
        self.exp_avg: Dict[str, torch.Tensor] = OrderedDict()        # Adam's first / second moments, same keys
        self.exp_avg_sq: Dict[str, torch.Tensor] = OrderedDict()
        bounds: Dict[str, Tuple[int, int]] = {}
        off = 0
        for k, shp in self.shapes.items():
            cnt = int(torch.Size(shp).numel())
            self.params[k] = self.flat_p[off:off + cnt].view(shp)
            self.grads[k] = self.flat_g[off:off + cnt].view(shp)
            self.exp_avg[k] = self.flat_m[off:off + cnt].view(shp)
            self.exp_avg_sq[k] = self.flat_v[off:off + cnt].view(shp)
            seg = self._segment_of(k)
            lo, hi = bounds.get(seg, (off, off))
            if hi != off:
                raise AssertionError(f"segment {seg} is not contiguous in the flat buffer")
            bounds[seg] = (lo, off + cnt)
            off += cnt
        # contiguous segments of the flat buffers, one per tower (the last one also holds the heads)
        self.segments = bounds
        if share is not None:
            self.adam_state, self.drop_step = share.adam_state, share.drop_step
        else:
            self.adam_state = torch.tensor([0.0, lr, 0.0, 0.0], device=dev)     # [step, lr, -, -]
            self.drop_step = torch.zeros(1, dtype=torch.int32, device=dev)       # device-side dropout step counter
        self.seed = (share.seed if share is not None else seed) & 0xFFFFFFFF
        if init and share is None:
            self.reset_parameters(seed)
        self.logits = torch.zeros(3, self.B, self.K, device=dev)
        self.losses = torch.zeros(4, device=dev)
        self._graph = None
        self._static = None
        self._adam_plans: Dict[tuple, AdamPackPlan] = {}
        # weight-gradient launch options (_setup_wgrad): towers whose second row group lands in a slot, and the special
        # index ranges of the flat Adam: (lo, n, slot or None, keep)
        self._slot_towers: List[TowerRuntime] = []
        self._ranges_add: list = []
        self._ranges_keep: list = []
        self._slots_folded = False
        # side streams for the paths whose towers cannot share a launch (wide towers, mixed shapes, the MIMIC static MLP): the
        # second modality runs beside the first.  The AV-MNIST step needs none of them: nine launches on the main stream.
        self.s_b = torch.cuda.Stream(device=dev)
        self.s_fus = torch.cuda.Stream(device=dev)
        self.s_emb = torch.cuda.Stream(device=dev)
        self.concurrent = os.environ.get("M2M_CONCURRENT", "1") != "0"      # 0: every launch on the main stream (A/B)
        self._build()
        self.pack()

    # ---- helpers for subclasses ----------------------------------------------------------------------------
    def _make_tower(self, prefix: str, c: dict, N: int, site: int) -> TowerRuntime:
        rt = TowerRuntime(c["hidden_dim"], N, c["token_dim"], c["channel_dim"], c["num_mixers"], True, self.p_drop,
                          self.prec, site)
        nb = c["num_mixers"]
        blocks = [{f: self.params[f"{prefix}mixer_blocks.{i}.{BLOCK_KEYS[f]}"] for f in BLOCK_FIELDS} for i in range(nb)]
        rt.bind_params(blocks, (self.params[prefix + "layer_norm.weight"], self.params[prefix + "layer_norm.bias"]))
        rt.bind_grad_tensors([{f: self.grads[f"{prefix}mixer_blocks.{i}.{BLOCK_KEYS[f]}"] for f in BLOCK_FIELDS}
                              for i in range(nb)],
                             (self.grads[prefix + "layer_norm.weight"], self.grads[prefix + "layer_norm.bias"]))
        rt.ensure_buffers(self.B)
        rt.ensure_workspace(self.B)
        return rt

    def _setup_wgrad(self, launches: Sequence[Sequence[TowerRuntime]]):
        """Options of the channel-mixing weight-gradient launch for a step that runs ONE backward per optimizer step:
          * overwrite: a tower whose gradient elements have a single owner writes them with "=" (no read of the old values)
            and the flat Adam leaves those ranges uncleared (-66 MB per step on M2-Mixer-B);
          * slot: in a grouped launch a tower that needs two row groups (the fusion tower: twice the rows of its neighbours)
            stores the second group's sums into a slot that the flat Adam adds -- plain stores instead of float atomics.
        M2M_WGRAD_OVERWRITE=0 / M2M_WGRAD_SLOT=0 switch them off (A/B)."""
        self._slot_towers, self._ranges_add, self._ranges_keep = [], [], []
        if os.environ.get("M2M_WGRAD_OVERWRITE", "1") == "0":
            return
        # `launches`: the towers of each weight-gradient launch of the step (a launch of several towers is the grouped form)
        cand = []                                    # (tower, takes a slot)
        for towers in launches:
            bits = 0
            if len(towers) > 1 and os.environ.get("M2M_WGRAD_SLOT", "1") != "0":
                have = [t.alloc_wslot(self.flat_g) for t in towers]
                bits = wgrad_slot_groups(towers, self.B) if all(have) else 0
            cand += [(t, bool(bits >> i & 1)) for i, t in enumerate(towers)]
        table = []                                   # all-or-nothing per tower (the kernel's range table holds MAX_GRAD_RANGES)
        cand.sort(key=lambda c: not c[1])            # slot towers first: their ranges are mandatory
        for t, slot in cand:
            if not slot:
                t.clear_wslot()
            rng = t.channel_grad_ranges(self.flat_g)
            fits = rng is not None and len(table) + len(rng) <= L.MAX_GRAD_RANGES
            if slot and not fits:
                raise RuntimeError("more slot ranges than m2m_adam_step_ranges takes")
            if not fits or (t.wgrad_groups(self.B) != 1 and not slot):
                continue
            t.set_wgrad_overwrite(True)
            views = t.wslot_views() if slot else [None] * len(rng)
            if slot:
                self._slot_towers.append(t)
            table += [(lo, n, v, 1) for (lo, n), v in zip(rng, views)]
        self._ranges_add = sorted(table, key=lambda r: r[0])
        self._ranges_keep = [(lo, n, None, k) for lo, n, _, k in self._ranges_add]

    def _head(self, name: str, pooled, d_pooled, weight: float, with_grad: bool) -> dict:
        key = "classifier_fusion.classifer." if name == "fusion" else f"classifier_{name}."
        P, Gr = self.params, self.grads
        return dict(pooled=pooled, w=P[key + "weight"], b=P[key + "bias"], g_w=Gr[key + "weight"], g_b=Gr[key + "bias"],
                    d_pooled=d_pooled if with_grad else None, weight=weight)

    def _streams(self, concurrent: Optional[bool] = None):
        main = torch.cuda.current_stream()
        conc = self.concurrent if concurrent is None else (concurrent and self.concurrent)
        return main, (self.s_b if conc else main), (self.s_fus if conc else main)

    # ---- parameters --------------------------------------------------------------------------------------
    def reset_parameters(self, seed: int = 42):
        """torch default init (Linear / Conv2d: kaiming-uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)); LayerNorm 1/0),
        drawn on the CPU generator under `seed` in creation order (run.py:32 seeds 42)."""
        gen = torch.Generator().manual_seed(seed)
        for k, shp in self.shapes.items():
            if _is_layer_norm(k):
                v = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
            else:
                wshape = self.shapes[k[:-4] + "weight"] if k.endswith("bias") else shp
                fan_in = int(torch.Size(wshape[1:]).numel())
                bound = 1.0 / fan_in ** 0.5
                v = (torch.rand(shp, generator=gen) * 2 - 1) * bound
            self.params[k].copy_(v)

    #: non-parameter entries a reference state_dict may carry for this model (buffers of its loss modules)
    EXTRA_STATE_KEYS: Tuple[str, ...] = ()

    def _consume_extra_state(self, key: str, value: torch.Tensor):
        raise KeyError(key)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        missing = [k for k in self.shapes if k not in sd]
        extra = [k for k in sd if k not in self.shapes and k not in self.EXTRA_STATE_KEYS]
        if missing or extra:
            raise KeyError(f"state dict mismatch: missing {missing[:4]}, unexpected {extra[:4]}")
        for k in self.shapes:
            self.params[k].copy_(sd[k].to(self.device, torch.float32).reshape(self.shapes[k]))
        for k in self.EXTRA_STATE_KEYS:
            if k in sd:
                self._consume_extra_state(k, sd[k])
        self.pack()

    def _extra_state(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict()

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        """The reference module's state_dict: parameters under its key names, in its order, then the loss-module buffers."""
        out = OrderedDict((k, v.detach().clone()) for k, v in self.params.items())
        out.update(self._extra_state())
        return out

    def optimizer_state_dict(self) -> dict:
        """torch.optim.Adam.state_dict() layout (what Lightning stores under `optimizer_states[0]`): per parameter index
        `step`, `exp_avg`, `exp_avg_sq`; one param group with this engine's hyper-parameters."""
        step = float(self.adam_state[0])
        state = {i: {"step": torch.tensor(step), "exp_avg": self.exp_avg[k].detach().cpu().clone(),
                     "exp_avg_sq": self.exp_avg_sq[k].detach().cpu().clone()} for i, k in enumerate(self.shapes)}
        group = {"lr": float(self.adam_state[1]), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "params": list(range(len(self.shapes)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, osd: dict):
        """torch.optim.Adam.state_dict() as the reference writes it (Lightning: `optimizer_states[0]`).  Parameters without an
        entry -- a state dict saved before the first step, or with modalities frozen (`requires_grad=False` parameters never get
        Adam state: models/avmnist.py:314-324) -- start from zero moments; the ONE step count the fused Adam keeps is taken
        from the entries that are present (they must agree)."""
        keys = list(self.shapes)
        state = osd.get("state", {})
        bad = [i for i in state if not isinstance(i, int) or i < 0 or i >= len(keys)]
        if bad:
            raise KeyError(f"optimizer state: entries {bad[:4]} do not index this model's {len(keys)} parameters")
        steps = {float(st["step"]) for st in state.values() if "step" in st}
        if len(steps) > 1:
            raise RuntimeError("optimizer state: the fused Adam keeps ONE step count for all parameters")
        for i, k in enumerate(keys):
            st = state.get(i)
            if st is None or "exp_avg" not in st:
                self.exp_avg[k].zero_()
                self.exp_avg_sq[k].zero_()
                continue
            self.exp_avg[k].copy_(st["exp_avg"].to(self.device, torch.float32).reshape(self.shapes[k]))
            self.exp_avg_sq[k].copy_(st["exp_avg_sq"].to(self.device, torch.float32).reshape(self.shapes[k]))
        self.adam_state[0] = steps.pop() if steps else 0.0
        if osd.get("param_groups"):
            self.set_lr(float(osd["param_groups"][0]["lr"]))

    def set_lr(self, lr: float):
        self.adam_state[1] = lr

    def _sibling_kwargs(self) -> dict:
        return {"fusion_loss_weight": self.fusion_loss_weight} if hasattr(self, "fusion_loss_weight") else {}

    def sibling(self, batch_size: int, trains: bool = True):
        """An engine for another batch size over THIS engine's parameters, gradients, Adam state and step counters
        (constructor argument share=): the step that takes an epoch's ragged last batch, or a validation engine.
        trains=False: the sibling only evaluates (it never writes a gradient): nothing of this engine changes.

        Both engines work on ONE gradient buffer.  A range one of them leaves uncleared ("keep": its own next backward
        overwrites it) must be overwritten by the other one's backward too, or that one would accumulate onto stale values --
        so the keep flags of both are narrowed to the ranges BOTH overwrite.  A captured graph holds its range table BY VALUE
        (m2m_adam_step_ranges copies it into the kernel arguments), so narrowing after capture() would leave the graph with
        the old flags: that is refused -- build training siblings before capture(), or capture again."""
        prec = {v: k for k, v in L.PREC_BY_NAME.items() if k in ("bf16", "fp32")}[self.prec]
        sib = type(self)(self.cfg, batch_size, device=self.device, precision=prec, lr=float(self.adam_state[1]),
                         betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, seed=self.seed, init=False,
                         share=self, **self._sibling_kwargs())
        if hasattr(self, "pos_weight"):                      # MM-IMDb: the LOADED criterion buffer, not cfg's
            sib.pos_weight.copy_(self.pos_weight)
        if not trains:
            sib._eval_only = True
            return sib
        mine, theirs = {r[0] for r in self._ranges_add}, {r[0] for r in sib._ranges_add}
        narrowed = [lo for lo, n, v, k in self._ranges_add if k and lo not in theirs]
        if narrowed and self._graph is not None:
            raise RuntimeError(f"sibling({batch_size}): this engine's captured graph keeps {len(narrowed)} gradient range(s) uncleared "
                               "that the sibling's backward would not overwrite; build training siblings before capture() "
                               "(or capture again afterwards)")
        for eng, other in ((self, theirs), (sib, mine)):
            eng._ranges_add = [(lo, n, v, int(k and lo in other)) for lo, n, v, k in eng._ranges_add]
            eng._ranges_keep = [(lo, n, None, k) for lo, n, _, k in eng._ranges_add]
        return sib

    # ---- optimizer -------------------------------------------------------------------------------------------
    def _adam(self, lo: int, hi: int, grad_scale: float, bump: bool, grad_bf16: Optional[torch.Tensor] = None, ranges=None):
        """Adam over flat elements [lo, hi); clears the gradients it consumes.  grad_bf16: a bf16 copy of the whole flat
        gradient (the compressed all-reduce result) to take the values from instead of flat_g."""
        n = hi - lo
        off = lo * 4
        tail = (self.adam_state.data_ptr(), self.betas[0], self.betas[1], self.eps, self.weight_decay, -abs(grad_scale), int(bump),
                L.stream_ptr())
        if grad_bf16 is not None and (grad_bf16.dtype != torch.bfloat16 or grad_bf16.numel() != self.n_params or not grad_bf16.is_cuda):
            raise RuntimeError("grad_bf16 must be a bf16 device copy of the whole flat gradient")
        if ranges:
            if lo != 0 or hi != self.n_params:
                raise RuntimeError("gradient ranges are indexed over the whole flat buffer")
            arr = (L.GradRange * len(ranges))()
            for i, (rlo, rn, add, keep) in enumerate(ranges):
                arr[i].lo, arr[i].n, arr[i].add, arr[i].keep = rlo, rn, L.ptr(add), int(keep)
            L.check(L.lib().m2m_adam_step_ranges(self.flat_p.data_ptr(), self.flat_g.data_ptr(), L.ptr(grad_bf16), self.flat_m.data_ptr(),
                                                 self.flat_v.data_ptr(), n, *tail[:-1], arr, len(ranges), tail[-1]), "adam_step_ranges")
            return
        if grad_bf16 is None:
            L.check(L.lib().m2m_adam_step(self.flat_p.data_ptr() + off, self.flat_g.data_ptr() + off, self.flat_m.data_ptr() + off,
                                          self.flat_v.data_ptr() + off, n, *tail), "adam_step")
        else:
            L.check(L.lib().m2m_adam_step_bf16(self.flat_p.data_ptr() + off, self.flat_g.data_ptr() + off,
                                               grad_bf16.data_ptr() + lo * 2, self.flat_m.data_ptr() + off,
                                               self.flat_v.data_ptr() + off, n, *tail), "adam_step_bf16")

    def _prologue(self):
        """Head of a training step: Adam step count += 1, dropout counter += 1, losses = 0 -- one tiny launch."""
        L.check(L.lib().m2m_step_prologue(self.adam_state.data_ptr(), self.drop_step.data_ptr(), self.losses.data_ptr(), 4,
                                          L.stream_ptr()), "step_prologue")

    def forward_backward(self, *batch):
        """forward (dropout on) -> multi-head loss -> backward; gradients are ADDED into flat_g, which must be
        zero on entry: it is cleared at construction and again by every optimizer_step (the Adam kernel clears
        each element it consumes), so no separate fill pass is needed.  (Channel-mixing weight gradients of the
        overwriting towers -- _setup_wgrad -- are written, not added: those ranges need no clearing.)  On return flat_g
        holds the complete gradient: a row group the weight-gradient launch left in a slot is folded in here."""
        self._check_trains()
        self._slots_folded = False
        self._forward(*batch, training=True, with_grad=True, prologue=True)
        self._backward(*batch[:-1])
        for t in self._slot_towers:
            t.wgrad_fold()
        self._slots_folded = True                       # (consumed by the optimizer_step that follows; fused_step resets it)

    def fused_step(self, *batch):
        """forward + backward + Adam + re-pack in one go (no gradient exchange: single-GPU training)."""
        self._check_trains()
        # (a forward_backward() without its optimizer_step() -- gradient inspection, a skipped step -- must not make THIS step's
        # Adam skip the slot of the two-group tower: the slot is folded by forward_backward only)
        self._slots_folded = False
        self._forward(*batch, training=True, with_grad=True, prologue=True)
        self._backward(*batch[:-1], fused_update=True)
        return self.losses

    def _check_trains(self):
        if getattr(self, "_eval_only", False):
            raise RuntimeError("this engine was built with sibling(..., trains=False): it only evaluates")

    def _adam_pack_modules(self):
        """(towers, embeds) whose parameters all live in the flat buffers and whose operand copies one m2m_adam_pack_all
        launch can rebuild -- None: the model needs the two-launch form (Adam, then pack())."""
        return None

    def _update(self, grad_scale: float = 1.0, grad_bf16: Optional[torch.Tensor] = None):
        """Adam over every parameter + operand re-pack.  One launch (m2m_adam_pack_all) where the model allows it: the
        re-pack then takes the updated weights from the workgroup that computed them instead of re-reading the masters."""
        mods = self._adam_pack_modules() if self._fused_update else None
        # the slot of a two-group tower: added inside Adam (fused step) unless forward_backward has folded it in already
        ranges = self._ranges_keep if (self._slots_folded or grad_bf16 is not None) else self._ranges_add
        self._slots_folded = False
        if mods is None:
            self._adam(0, self.n_params, grad_scale, False, grad_bf16, ranges)    # negative scale inside: clears the gradients
            self.pack()
            return
        key = (abs(float(grad_scale)), 0 if grad_bf16 is None else grad_bf16.data_ptr(), tuple((lo, n, 0 if v is None else v.data_ptr(), k) for lo, n, v, k in ranges))
        plan = self._adam_plans.get(key)
        if plan is None:
            if grad_bf16 is not None and (grad_bf16.dtype != torch.bfloat16 or grad_bf16.numel() != self.n_params or not grad_bf16.is_cuda):
                raise RuntimeError("grad_bf16 must be a bf16 device copy of the whole flat gradient")
            plan = AdamPackPlan(mods[0], mods[1], self.flat_p, self.flat_g, grad_bf16, self.flat_m, self.flat_v, self.adam_state,
                                self.betas, self.eps, self.weight_decay, grad_scale, ranges)
            self._adam_plans[key] = plan
        plan.run()

    def optimizer_step(self, grad_scale: float = 1.0, grad_bf16: Optional[torch.Tensor] = None):
        self._update(grad_scale, grad_bf16)                                    # (the step counters were advanced by _prologue)

    def train_step(self, *batch, grad_sync=None):
        """One optimisation step.  grad_sync: optional callable(flat_grad) doing the data-parallel
        all-reduce (parallel.GradSync); it returns the factor the summed gradient must be scaled by."""
        if grad_sync is None:
            return self.fused_step(*batch)
        self.forward_backward(*batch)
        if getattr(grad_sync, "pipelined", False):
            self._exchange_and_update_pipelined(grad_sync)
        else:
            self.optimizer_step(grad_sync(self.flat_g), getattr(grad_sync, "reduced_bf16", None))
        return self.losses

    # ---- exchange pipelined with the optimizer (parallel.PipelinedGradSync) ------------------------------------------------
    def _update_chunks(self):
        """[(lo, hi, [modules to re-pack]), ...]: a partition of the flat buffers into pieces whose Adam update and operand
        re-pack need nothing outside the piece -- the parameter segments (one per tower; the last also holds the heads).
        Engines without per-segment packers return the whole buffer as one chunk."""
        return [(0, self.n_params, None)]

    def _adam_chunk(self, lo: int, hi: int, scale: float):
        """Adam over flat elements [lo, hi) with the engine's kept ranges cut to the chunk (the slots were folded in by
        forward_backward: `keep` is all that is left of the range table)."""
        ranges = []
        for rlo, rn, _, keep in self._ranges_keep:
            a, b = max(rlo, lo), min(rlo + rn, hi)
            if a < b:
                ranges.append((a - lo, b - a, None, keep))
        n, off = hi - lo, lo * 4
        tail = (self.adam_state.data_ptr(), self.betas[0], self.betas[1], self.eps, self.weight_decay, -abs(scale), 0)
        if ranges:
            arr = (L.GradRange * len(ranges))()
            for i, (rlo, rn, add, keep) in enumerate(ranges):
                arr[i].lo, arr[i].n, arr[i].add, arr[i].keep = rlo, rn, None, int(keep)
            L.check(L.lib().m2m_adam_step_ranges(self.flat_p.data_ptr() + off, self.flat_g.data_ptr() + off, None, self.flat_m.data_ptr() + off,
                                                 self.flat_v.data_ptr() + off, n, *tail, arr, len(ranges), L.stream_ptr()), "adam_step_ranges")
        else:
            L.check(L.lib().m2m_adam_step(self.flat_p.data_ptr() + off, self.flat_g.data_ptr() + off, self.flat_m.data_ptr() + off,
                                          self.flat_v.data_ptr() + off, n, *tail, L.stream_ptr()), "adam_step")

    def _update_one_chunk(self, k: int, scale: float):
        lo, hi, mods = self._update_chunks()[k]
        self._adam_chunk(lo, hi, scale)
        if mods is None:
            self.pack()
        else:
            for m in mods:
                m.pack(force=True)

    def _exchange_and_update_pipelined(self, sync):
        """all-reduce chunk k + 1 .. on the communication stream while chunk k's Adam + re-pack run here."""
        if not self._slots_folded:
            raise RuntimeError("the pipelined update follows forward_backward (which folds the weight-gradient slots)")
        chunks = self._update_chunks()
        scale = sync.start(self.flat_g, [(lo, hi) for lo, hi, _ in chunks])
        for k in range(len(chunks)):
            sync.wait(k)
            self._update_one_chunk(k, scale)
        self._slots_folded = False

    # ---- hipGraph capture -----------------------------------------------------------------------------------
    def capture(self, *batch, grad_sync=None, steps: int = 1):
        """Capture train_step on static input buffers; returns a callable replay(*batch).
        With a grad_sync the step is captured as two graphs with the all-reduce between them.
        steps > 1 (single-GPU path only) captures that many consecutive training steps in ONE graph -- every replayed
        graph costs ~17 us of launch gap on this platform, which is 2 % of a step -- with one static input slot per step:
        replay(*batch_0, *batch_1, ...) or replay() to reuse what the slots hold; losses / logits / preds of step i land
        in self.losses_steps[i] etc. (replay() returns, and replay.losses is, that (steps, 4) buffer; the engine's own
        losses / logits / preds are left to the single-step paths).
        capture() does not train: the warm-up steps it needs are undone (parameters, Adam moments, step counters)."""
        if steps < 1 or (steps > 1 and grad_sync is not None):
            raise ValueError("steps > 1 is only supported without a gradient exchange")
        nb = len(batch)
        slots = [tuple(t.clone() for t in batch) for _ in range(steps)]
        if steps > 1 and os.environ.get("M2M_CAPTURE_SHARE_SLOTS", "0") == "1":
            slots = [slots[0]] * steps                  # diagnostic (scripts/spg_probe.sh): every step reads the SAME (cache-resident) batch
        self._static = slots[0] if steps == 1 else slots
        st = slots[0]
        # The warm-up below runs REAL training steps (lazy initialisation of the launches must happen outside the capture).
        # Everything they change is put back afterwards, so capture() leaves the model exactly as it found it.
        snap = [t.clone() for t in (self.flat_p, self.flat_m, self.flat_v, self.flat_g, self.adam_state, self.drop_step)]
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            scale = 1.0
            for _ in range(2):                       # warm-up: lazy inits (LDS attributes, allocations) happen here
                if grad_sync is None:
                    self.fused_step(*st)
                elif getattr(grad_sync, "pipelined", False):
                    self.forward_backward(*st)
                    self._exchange_and_update_pipelined(grad_sync)
                    scale = 1.0 / getattr(grad_sync, "world", 1)
                else:
                    self.forward_backward(*st)
                    scale = grad_sync(self.flat_g)
                    self.optimizer_step(scale, getattr(grad_sync, "reduced_bf16", None))
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for dst, src in zip((self.flat_p, self.flat_m, self.flat_v, self.flat_g, self.adam_state, self.drop_step), snap):
            dst.copy_(src)
        self.pack()
        torch.cuda.synchronize()
        # thread_local capture mode: a data-parallel process has other threads (the RCCL watchdog) that may touch the HIP
        # runtime while this thread captures; only this thread's calls belong to the graph
        g1 = torch.cuda.CUDAGraph()
        out_losses = self.losses
        if grad_sync is None:
            if steps > 1:
                # per-step output slots of THIS graph; the engine's own losses / logits / preds stay what they were (other
                # captured graphs and evaluate() write those)
                home = (self.losses, self.logits, self.preds)
                self.losses_steps = torch.zeros(steps, *self.losses.shape, device=self.device)
                self.logits_steps = torch.zeros(steps, *self.logits.shape, device=self.device)
                self.preds_steps = torch.zeros(steps, *self.preds.shape, dtype=self.preds.dtype, device=self.device)
                out_losses = self.losses_steps
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                for i in range(steps):
                    if steps > 1:
                        self.losses, self.logits, self.preds = self.losses_steps[i], self.logits_steps[i], self.preds_steps[i]
                    self.fused_step(*slots[i])
            if steps > 1:
                self.losses, self.logits, self.preds = home
            graphs = (g1,)
        elif getattr(grad_sync, "pipelined", False):
            # forward + backward | per chunk: [its all-reduce lands] its Adam + re-pack (one small graph per chunk)
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                self.forward_backward(*st)
            chunk_graphs = []
            for k in range(len(self._update_chunks())):
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk, capture_error_mode="thread_local"):
                    self._update_one_chunk(k, scale)
                chunk_graphs.append(gk)
            self._slots_folded = False
            graphs = (g1, *chunk_graphs)
        else:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                self.forward_backward(*st)
            with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                self.optimizer_step(scale, getattr(grad_sync, "reduced_bf16", None))   # (the buffer exists since the warm-up)
            graphs = (g1, g2)
        self._graph = graphs
        pipelined = grad_sync is not None and getattr(grad_sync, "pipelined", False)

        def replay(*new_batch):
            if new_batch and new_batch[0] is not None:
                if len(new_batch) != nb * steps:
                    raise ValueError(f"replay expects {nb * steps} tensors ({steps} step(s) x {nb}), got {len(new_batch)}")
                for i in range(steps):
                    for dst, src in zip(slots[i], new_batch[i * nb:(i + 1) * nb]):
                        dst.copy_(src, non_blocking=True)
            graphs[0].replay()
            if pipelined:
                grad_sync.start(self.flat_g, [(lo, hi) for lo, hi, _ in self._update_chunks()])
                for k in range(len(graphs) - 1):
                    grad_sync.wait(k)
                    graphs[1 + k].replay()
            elif grad_sync is not None:
                grad_sync(self.flat_g)
                graphs[1].replay()
            return out_losses                     # this graph's own buffer: (4,) or, for a multi-step graph, (steps, 4)

        replay.losses = out_losses
        return replay


class _TwoTowerEngine(_FlatEngine):
    """Two MLPMixer towers (patch embedding) -> ConcatFusion(dim=1) -> FusionMixer -> three heads."""

    MODS: Tuple[str, str] = ("image", "audio")

    def _param_shapes(self, cfg):
        return two_tower_param_shapes(cfg, self.MODS)

    def _segment_of(self, k: str) -> str:
        a, b = self.MODS
        return a if k.startswith(f"{a}_mixer.") else (b if k.startswith(f"{b}_mixer.") else "fusion")

    def _build(self):
        a, b = self.MODS
        cfg = self.cfg
        ca, cb, cm = cfg[a], cfg[b], cfg["multimodal"]
        self.D = ca["hidden_dim"]
        if not (cb["hidden_dim"] == self.D == cm["hidden_dim"]):
            raise RuntimeError("both towers and the fusion mixer must share hidden_dim (ConcatFusion on dim 1)")
        self.Na, self.Nb = _num_patch(ca), _num_patch(cb)
        self.Nf = self.Na + self.Nb
        self.t_a = self._make_tower(f"{a}_mixer.", ca, self.Na, 0)
        self.t_b = self._make_tower(f"{b}_mixer.", cb, self.Nb, 1024)
        self.t_fus = self._make_tower("fusion_mixer.", cm, self.Nf, 2048)

        def make_embed(prefix, c):
            e = EmbedRuntime(c["in_channels"], c["image_size"][0], c["image_size"][1], c["patch_size"], c["patch_size"],
                             self.D, self.prec)
            e.bind_params(self.params[prefix + "to_patch_embedding.0.weight"], self.params[prefix + "to_patch_embedding.0.bias"])
            e.bind_grads(self.grads[prefix + "to_patch_embedding.0.weight"], self.grads[prefix + "to_patch_embedding.0.bias"])
            return e

        self.e_a = make_embed(f"{a}_mixer.", ca)
        self.e_b = make_embed(f"{b}_mixer.", cb)
        B, D, dev = self.B, self.D, self.device
        f = lambda *s: torch.zeros(*s, device=dev)
        # embedding outputs; a long-K embedding (audio) is computed as k-split partial sums that the tower launch adds
        grouped = can_group_embeds(self.e_a, self.e_b) and can_group(self.t_a, self.t_b, self.B)
        self.x0_splits = (self.e_a.fwd_splits(), self.e_b.fwd_splits()) if grouped and not self.t_a.wide else (1, 1)
        self._x0_a, self._x0_b = f(self.x0_splits[0], B * self.Na, D), f(self.x0_splits[1], B * self.Nb, D)
        self.x0_a, self.x0_b = self._x0_a[0], self._x0_b[0]
        self.fused, self.fus_out = f(B, self.Nf, D), f(B, self.Nf, D)
        self.pool_a, self.pool_b, self.pool_fus = f(B, D), f(B, D), f(B, D)
        self.dpool_a, self.dpool_b, self.dpool_fus = f(B, D), f(B, D), f(B, D)
        self.d_fused = f(B, self.Nf, D)
        self.dx0_a, self.dx0_b = f(B * self.Na, D), f(B * self.Nb, D)
        self.preds = torch.zeros(self._preds_shape(), dtype=torch.int32, device=dev)
        # M2M_EARLY_FUSION_WGRAD=1 (wide towers, small batch): the fusion tower's weight gradients on a side stream beside the
        # two modality towers' backward launches instead of inside the merged launch at the end.  Measured on MM-IMDb at its
        # cfg batch: 0.555 ms against 0.502 ms merged -- the extra fork / join of the replayed graph and a single-tower launch
        # without its slot (80 steps) cost more than the shorter tail returns.  Off.
        self._early_fus_wgrad = (self.concurrent and self.t_fus.wide and B * self.Nf <= 8192 and
                                 os.environ.get("M2M_EARLY_FUSION_WGRAD", "0") == "1")
        self._setup_wgrad([[self.t_fus], [self.t_a, self.t_b]] if self._early_fus_wgrad else [[self.t_fus, self.t_a, self.t_b]])
        self._fused_heads = self._heads_are_ce() and self.t_fus.backward_heads_ok(B, 3, self.K)
        # Wide towers (MM-IMDb): their forward cannot pool inside the chain launch (a workgroup does not own whole samples) and
        # appends a token-mean launch; the heads kernel pools the tower outputs itself instead (m2m_head.tokens): two launches less.
        self._heads_pool = (self.t_a.wide and self.t_b.wide and self.t_fus.wide and os.environ.get("M2M_HEADS_POOL", "1") != "0")
        # the fusion tower's backward is always followed by the weight-gradient launch (_backward): the reduction of its
        # small-gradient slots rides there instead of being a launch between the two backward launches (M2M_DEFER_SMALL=0: A/B)
        self.t_fus.set_wgrad_reduces_small(not self._fused_heads and os.environ.get("M2M_DEFER_SMALL", "1") != "0")
        # the heads' weight gradients: per-workgroup slots added in a fixed order by the weight-gradient launch instead of float
        # atomics -- with them a bf16 step is bit-reproducible (M2M_HEAD_SLOTS=0: atomics)
        self._head_part, self._wgrad_heads = None, None
        if (self._heads_are_ce() and not self._fused_heads and os.environ.get("M2M_HEAD_SLOTS", "1") != "0"
                and self.K * D + self.K + 2 <= L.SPLIT_GPART):
            self._head_part = torch.zeros(3, int(L.lib().m2m_heads_part_tiles(B)), L.SPLIT_GPART, device=dev)
        # the two modality towers' backward launch: slots instead of 128-way contended float atomics for the small gradients,
        # their reduction in the weight-gradient launch as well (M2M_GROUP_SLOTS=0: atomics, A/B)
        if (os.environ.get("M2M_GROUP_SLOTS", "1") != "0" and self.t_a.has_small_slots() and self.t_b.has_small_slots()
                and can_group(self.t_a, self.t_b, self.B)):
            for t in (self.t_a, self.t_b):
                t.set_wgrad_group_slots(True)
                t.set_wgrad_reduces_small(True)
        # the embeddings' weight gradients in their single-owner form: the tower backward leaves d_x0^T as packed blocks
        self._embed_towers = []
        if os.environ.get("M2M_EMBED_FAST", "1") != "0" and self.t_a.enable_dx0_image(B) and self.t_b.enable_dx0_image(B):
            self._embed_towers = [self.t_a, self.t_b]
            # their weight gradients are then written ("="), and Adam leaves those ranges uncleared -- if the range table has room
            import ctypes as C
            ep = (C.POINTER(L.Embed) * 2)(C.pointer(self.e_a.desc), C.pointer(self.e_b.desc))
            tp = (C.POINTER(L.Tower) * 2)(C.pointer(self.t_a.desc), C.pointer(self.t_b.desc))
            fast = bool(L.lib().m2m_embeds_wgrad_form(ep, tp, 2, B))
            if fast and os.environ.get("M2M_WGRAD_OVERWRITE", "1") != "0" and len(self._ranges_add) + 2 <= L.MAX_GRAD_RANGES:
                for e, pre in ((self.e_a, f"{a}_mixer."), (self.e_b, f"{b}_mixer.")):
                    gw = self.grads[pre + "to_patch_embedding.0.weight"]
                    e.set_wgrad_overwrite(True)
                    self._ranges_add.append(((gw.data_ptr() - self.flat_g.data_ptr()) // 4, gw.numel(), None, 1))
                self._ranges_add.sort(key=lambda r: r[0])
                self._ranges_keep = [(lo, n, None, k) for lo, n, _, k in self._ranges_add]

        # Patch embeddings inside the two-tower forward launch (m2m_towers_forward_embeds: -1 launch at the head of the step).
        # The step head then splits: losses = 0 / Adam step += 1 ride in that launch, the dropout counter -- which that launch
        # READS -- holds "steps completed", every launch of a training step is given step = 1 and the merged weight-gradient
        # launch (the last reader is long done) advances it.  Observable counter values are those of the prologue form.
        # OFF by default (M2M_EMBED_FOLD=1 enables it).  Measured on M2-Mixer-B, batch 512, three interleaved repetitions in one
        # process: towers forward 103-105 us with the embeddings inside against 84-85 + 21-22 us as two launches, step
        # 0.5116-0.5163 against 0.5067-0.5121 ms -- the audio tower's 128 workgroups each stream the whole 800 KB embedding
        # weight + 200 KB of input alone (+19 us, not the 7.5 us the CU's 64 B/clk would allow) while the image tower's 128 CUs
        # wait; the launch of its own spreads that work over 256 + 128 workgroups (k-split).  Parity-green either way.
        self._embed_fold = (os.environ.get("M2M_EMBED_FOLD", "0") == "1" and grouped and self.concurrent and not self._early_fus_wgrad
                            and self.x0_splits is not None and self.e_a.prec == self.t_a.prec and self.e_a.D == self.t_a.D
                            and towers_forward_embeds_ok([self.t_a, self.t_b], [self.e_a, self.e_b], B)
                            and self.t_a.wgrad_form(B) == 0 and self.t_fus.wgrad_form(B) == 0)
        self._pending_bump = False

    def _preds_shape(self):
        return (3, self.B)

    def pack(self):
        """Rebuild the packed MFMA-operand copies from the fp32 masters: every tower and both embeddings in ONE launch
        on the main stream (~100 MB of traffic at the HBM roofline).  Replaced five launches forked over the side
        streams, whose fork and join edges cost more inside the replayed graph than the concurrency returned."""
        towers, embeds = [self.t_a, self.t_b, self.t_fus], [self.e_a, self.e_b]
        if can_pack_all(towers, embeds):
            pack_all(towers, embeds)
            return
        for m in towers + embeds:
            m.pack(force=True)

    def _loss_heads(self, heads, labels, zero_losses):
        raise NotImplementedError

    def _heads_are_ce(self) -> bool:
        """True: three cross-entropy heads (the form m2m_tower_backward_heads computes)."""
        return False

    # ---- one training step (enqueue only; no host synchronisation) ----------------------------------------
    def _forward(self, xa, xb, labels, training: bool, with_grad: bool, prologue: bool = False):
        B, D = self.B, self.D
        sd = self.drop_step if training else None
        fs = self.Nf * D
        b_part = self.fused.view(-1)[self.Na * D:]
        main, side, _ = self._streams()
        so = 0                                               # host step offset of this pass's launches (see _embed_fold)
        pa, pb, pf = (None, None, None) if self._heads_pool else (self.pool_a, self.pool_b, self.pool_fus)
        self._pending_bump = False
        if self._embed_fold and training and prologue:
            # both patch embeddings inside the two-tower launch; head of the step: losses = 0, Adam step += 1 in that launch,
            # the dropout counter advances in the weight-gradient launch of _backward (every launch in between gets step = 1)
            so, self._pending_bump = 1, True
            towers_forward([self.t_a, self.t_b],
                           [(self.x0_a, self.Na * D, self.fused, fs, pa), (self.x0_b, self.Nb * D, b_part, fs, pb)],
                           B, training, self.seed, so, sd, embeds=[self.e_a, self.e_b], inputs=[xa, xb],
                           head=(self.adam_state, self.losses))
        elif self.concurrent and can_group(self.t_a, self.t_b, self.B):
            # one launch for both patch embeddings, one for both towers (blockIdx.y = tower), all on the main stream: no
            # cross-queue fork / join in the graph (a join costs ~6 us even when its event fired long ago)
            sa, sb = self.x0_splits
            if can_group_embeds(self.e_a, self.e_b):
                # the step prologue (counters, losses = 0) rides in this launch, the first one of the step
                head = (self.adam_state, self.drop_step, self.losses) if prologue else None
                embeds_forward([self.e_a, self.e_b], [xa, xb], [self._x0_a, self._x0_b], B, [sa, sb], head)
            else:
                if prologue:
                    self._prologue()
                sa = sb = 1
                self.e_a.forward(xa, B, self.x0_a)
                self.e_b.forward(xb, B, self.x0_b)
            towers_forward([self.t_a, self.t_b],
                           [(self.x0_a, self.Na * D, self.fused, fs, pa, sa, B * self.Na * D),
                            (self.x0_b, self.Nb * D, b_part, fs, pb, sb, B * self.Nb * D)],
                           B, training, self.seed, 0, sd)
        else:
            if prologue:
                self._prologue()
            side.wait_stream(main)
            with torch.cuda.stream(side):                   # second tower beside the first
                self.e_b.forward(xb, B, self.x0_b)
                self.t_b.forward(self.x0_b, self.Nb * D, B, b_part, fs, pb, training, self.seed, 0, sd)
            self.e_a.forward(xa, B, self.x0_a)
            self.t_a.forward(self.x0_a, self.Na * D, B, self.fused, fs, pa, training, self.seed, 0, sd)
            main.wait_stream(side)
        self.t_fus.forward(self.fused, fs, B, self.fus_out, fs, pf, training, self.seed, so, sd)
        a, b = self.MODS
        hw = self.head_weights
        heads = [self._head(a, self.pool_a, self.dpool_a, hw[a], with_grad),
                 self._head(b, self.pool_b, self.dpool_b, hw[b], with_grad),
                 self._head("fusion", self.pool_fus, self.dpool_fus, hw["fusion"], with_grad)]
        if self._heads_pool:
            heads[0]["tokens"] = (self.fused, self.Na, fs)
            heads[1]["tokens"] = (b_part.data_ptr(), self.Nb, fs)
            heads[2]["tokens"] = (self.fus_out, self.Nf, fs)
        self._wgrad_heads = None
        if training and with_grad and self._head_part is not None:
            for i, h in enumerate(heads):
                h["g_part"] = self._head_part[i]
            self._wgrad_heads = heads                     # _backward's weight-gradient launch adds the slots to g_w / g_b
        self._heads = heads
        self._labels = labels
        if training and with_grad and self._fused_heads:
            return                                         # heads + losses ride in the fusion tower's backward launch (_backward)
        self._loss_heads(heads, labels, not training)      # a training step's _prologue already cleared the losses

    def _backward(self, xa, xb, fused_update: bool = False):
        """Backward of the whole model.  fused_update: also apply Adam + re-pack (single-GPU path; with a gradient
        all-reduce the update is a separate phase)."""
        B, D = self.B, self.D
        fs = self.Nf * D
        sd = self.drop_step
        a, b = self.MODS
        so = 1 if self._pending_bump else 0                 # (the forward of this step left the dropout counter un-advanced)
        if self._fused_heads:
            # the three heads + multi-head CE in the prologue of this launch (m2m_tower_backward_heads): one launch less
            self.t_fus.backward_heads(B, self._heads, 2, self._labels, self.K, (self.logits, self.losses, self.preds),
                                      self.d_fused, fs, self.seed, so, sd)
        else:
            self.t_fus.backward(B, None, 0, self.dpool_fus, self.d_fused, fs, self.seed, so, sd)
        d_b_part = self.d_fused.view(-1)[self.Na * D:]
        main, s_a, s_f = self._streams()
        wg_towers = [self.t_fus, self.t_a, self.t_b]
        if self._early_fus_wgrad:
            s_f.wait_stream(main)
            with torch.cuda.stream(s_f):
                self.t_fus.wgrad(B, self.seed, so, sd)
            wg_towers = [self.t_a, self.t_b]
        s_e = self.s_emb if self.concurrent else main
        # The two tower chains fill the chip (128 + 128 workgroups) side by side; then ONE launch computes the channel-mixing
        # weight gradients of all three towers (the hardware dispatcher balances their ~300 workgroups; three launches on
        # three queues of a replayed graph raced and sometimes serialised each other) AND the two patch-embedding
        # gradients, whose workgroups back-fill the CUs the tower workgroups leave idle: the whole step is one queue.
        # (Any side stream forks from and joins into `main` directly: a fork from a forked stream crashed hipGraph
        # capture on ROCm 7.2.)
        if self.concurrent and can_group(self.t_a, self.t_b, self.B):
            towers_backward([self.t_a, self.t_b],
                            [(self.d_fused, fs, self.dpool_a, self.dx0_a, self.Na * D), (d_b_part, fs, self.dpool_b, self.dx0_b, self.Nb * D)],
                            B, self.seed, so, sd)
        else:
            s_a.wait_stream(main)
            with torch.cuda.stream(s_a):
                self.t_a.backward(B, self.d_fused, fs, self.dpool_a, self.dx0_a, self.Na * D, self.seed, so, sd)
            self.t_b.backward(B, d_b_part, fs, self.dpool_b, self.dx0_b, self.Nb * D, self.seed, so, sd)
            main.wait_stream(s_a)
        bump = self.drop_step if self._pending_bump else None
        if can_group_embeds(self.e_a, self.e_b) and self.e_a.prec == self.t_a.prec and self.e_a.D == self.t_a.D:
            towers_wgrad(wg_towers, B, [self.e_a, self.e_b], [xa, xb], [self.dx0_a, self.dx0_b],
                         seed=self.seed, step=so, step_dev=sd, embed_towers=self._embed_towers, heads=self._wgrad_heads, K=self.K,
                         bump=bump)
        else:
            if bump is not None:
                raise RuntimeError("the embedding fold needs the merged weight-gradient launch (it advances the dropout counter)")
            s_e.wait_stream(main)
            with torch.cuda.stream(s_e):
                self.e_b.wgrad(xb, self.dx0_b, B)
                self.e_a.wgrad(xa, self.dx0_a, B)
            towers_wgrad(wg_towers, B, seed=self.seed, step=so, step_dev=sd, heads=self._wgrad_heads, K=self.K)
            main.wait_stream(s_e)
        self._pending_bump = False
        if self._early_fus_wgrad:
            main.wait_stream(s_f)
        if fused_update:
            self._update(1.0)

    def _adam_pack_modules(self):
        towers, embeds = [self.t_a, self.t_b, self.t_fus], [self.e_a, self.e_b]
        return (towers, embeds) if can_pack_all(towers, embeds) else None

    def _update_chunks(self):
        """One chunk per parameter segment, in flat order: modality a (its patch embedding + tower), modality b, fusion tower
        + the three heads (heads have no packed copies)."""
        a, b = self.MODS
        mods = {a: [self.t_a, self.e_a], b: [self.t_b, self.e_b], "fusion": [self.t_fus]}
        segs = sorted(self.segments.items(), key=lambda kv: kv[1][0])
        return [(lo, hi, mods[name]) for name, (lo, hi) in segs]

    @torch.no_grad()
    def evaluate(self, xa, xb, labels):
        """validation/test step: dropout off (the reference's shared_step under model.eval())."""
        self._forward(xa, xb, labels, False, False)
        a, b = self.MODS
        return {"logits": self.logits[2], f"{a}_logits": self.logits[0], f"{b}_logits": self.logits[1],
                f"loss_{a}": self.losses[0], f"loss_{b}": self.losses[1], "loss_fusion": self.losses[2],
                "loss": self.losses[3], "preds": self.preds[2], f"preds_{a}": self.preds[0], f"preds_{b}": self.preds[1]}


class AVMnistEngine(_TwoTowerEngine):
    """cfg: {'dropout', 'num_classes', 'image': {...}, 'audio': {...}, 'multimodal': {...}} with the keys of
    cfg/avmnist/avmnist_m2-mixer_*.yml (model.modalities.*)."""

    MODS = ("image", "audio")

    def __init__(self, cfg: dict, batch_size: int, device="cuda:0", precision: Optional[str] = None,
                 lr: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 fusion_loss_weight: float = 1.0 / 3, seed: int = 42, init: bool = True, share=None):
        w = fusion_loss_weight
        ow = (1 - w) / 2
        # loss = (w Lf + ow Li + ow La) * 3      (models/avmnist.py:289-290)
        self.head_weights = {"image": 3 * ow, "audio": 3 * ow, "fusion": 3 * w}
        self.fusion_loss_weight = w
        super().__init__(cfg, batch_size, device, precision, lr, betas, eps, weight_decay, seed, init, share)

    def _loss_heads(self, heads, labels, zero_losses):
        heads_ce(heads, labels, self.B, self.D, self.K, out=(self.logits, self.losses, self.preds), zero_losses=zero_losses)

    def _heads_are_ce(self) -> bool:
        return True

    # names kept from the first engine revision (bench.py, tests)
    t_img = property(lambda self: self.t_a)
    t_aud = property(lambda self: self.t_b)
    e_img = property(lambda self: self.e_a)
    e_aud = property(lambda self: self.e_b)
    s_aud = property(lambda self: self.s_b)
    Ni = property(lambda self: self.Na)
    x0_img = property(lambda self: self.x0_a)


class MMIMDBEngine(_TwoTowerEngine):
    """cfg: {'dropout', 'num_classes', 'pos_weight': [K], 'image': {...}, 'text': {...}, 'multimodal': {...}} with the
    keys of cfg/mmimdb/mmimdb_3loss.yml.  Loss = BCE_image + BCE_text + BCE_fusion (models/mmimdb.py:115-123),
    preds = sigmoid(logits) > 0.5 (:128-133); labels are (B, K) multi-hot floats."""

    MODS = ("image", "text")
    #: buffers of the three BCEWithLogitsLoss modules in the reference's state_dict (models/mmimdb.py:47-50)
    EXTRA_STATE_KEYS = ("image_criterion.pos_weight", "text_criterion.pos_weight", "fusion_criterion.pos_weight")

    def __init__(self, cfg: dict, batch_size: int, device="cuda:0", precision: Optional[str] = None,
                 lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 seed: int = 42, init: bool = True, share=None):
        self.head_weights = {"image": 1.0, "text": 1.0, "fusion": 1.0}
        super().__init__(cfg, batch_size, device, precision, lr, betas, eps, weight_decay, seed, init, share)
        self.pos_weight = torch.tensor(cfg["pos_weight"], dtype=torch.float32, device=self.device)
        if self.pos_weight.numel() != self.K:
            raise RuntimeError("pos_weight needs one entry per class")

    def _consume_extra_state(self, key, value):
        v = value.to(self.device, torch.float32).reshape(-1)
        if v.numel() != self.K:
            raise RuntimeError(f"{key}: expected {self.K} entries")
        if key == self.EXTRA_STATE_KEYS[0]:
            self.pos_weight.copy_(v)                 # one pos_weight serves the three heads (the reference builds them equal)
        elif not torch.equal(v, self.pos_weight):
            raise RuntimeError(f"{key} differs from image_criterion.pos_weight: the fused heads kernel takes one pos_weight")

    def _extra_state(self):
        return OrderedDict((k, self.pos_weight.detach().clone()) for k in self.EXTRA_STATE_KEYS)

    def _preds_shape(self):
        return (3, self.B, self.K)

    def _loss_heads(self, heads, labels, zero_losses):
        heads_bce(heads, labels, self.pos_weight, self.B, self.D, self.K, out=(self.logits, self.losses, self.preds),
                  zero_losses=zero_losses)


class MimicEngine(_FlatEngine):
    """cfg: {'dropout', 'num_classes', 'time': {...}, 'static': {...}, 'multimodal': {...}} with the keys of
    cfg/mimic/mimic_m2-mixer_H.yml.  static (B, 5) -> MLP -> one token; time (B, 24, 12) -> Linear proj -> MixerBlocks;
    fused = cat(static token, time tokens) (models/mimic.py:98-103); loss = w Lf + ow Ls + ow Lt (:115-121, no x3)."""

    def __init__(self, cfg: dict, batch_size: int, device="cuda:0", precision: Optional[str] = None,
                 lr: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 fusion_loss_weight: float = 1.0 / 3, seed: int = 42, init: bool = True, share=None):
        w = fusion_loss_weight
        ow = (1 - w) / 2
        self.head_weights = {"static": ow, "time": ow, "fusion": w}
        self.fusion_loss_weight = w
        super().__init__(cfg, batch_size, device, precision, lr, betas, eps, weight_decay, seed, init, share)

    def _param_shapes(self, cfg):
        return mimic_param_shapes(cfg)

    def _segment_of(self, k: str) -> str:
        return "time" if k.startswith("time_mixer.") else ("static" if k.startswith("static_extractor.") else "fusion")

    def _build(self):
        cfg = self.cfg
        ct, cs, cm = cfg["time"], cfg["static"], cfg["multimodal"]
        self.D = ct["hidden_dim"]
        if not (ct["proj_dim"] == self.D == cm["hidden_dim"] == cs["output_dim"]):
            raise RuntimeError("time hidden_dim / proj_dim, static output_dim and fusion hidden_dim must agree")
        self.Nt = ct["num_patch"]
        self.Nf = 1 + self.Nt
        # Side streams (static MLP beside the time tower; the three parameter segments' weight gradients + Adam side by side):
        # every fork / join edge of the replayed graph costs ~5 us, the launches they hide are 5-30 us at the cfg batch:
        # measured 0.230 ms on one stream against 0.255 ms on three (fwd only 0.257, bwd only 0.25-0.27); at batch 8192 the
        # launches are long enough: 2.445 ms on three streams against 2.492 on one.  M2M_MIMIC_STREAMS=fwd|bwd|both|none (A/B).
        mode = os.environ.get("M2M_MIMIC_STREAMS", "none" if self.B <= 1024 else "both")
        self._merged_tail = os.environ.get("M2M_MIMIC_MERGED_TAIL", "1") != "0"          # (A/B)
        self._conc_fwd, self._conc_bwd = mode in ("fwd", "both"), mode in ("bwd", "both")
        self._heads_pool = os.environ.get("M2M_HEADS_POOL", "1") != "0"       # (both towers are wide: N = 24 / 25)
        # the static MLP's two launches as extra workgroups of the time tower's token-mixing launches (MlpRuntime.forward_ride)
        self._mlp_ride = os.environ.get("M2M_MLP_RIDE", "1") != "0" and self.B <= 2048
        self.t_time = self._make_tower("time_mixer.", ct, self.Nt, 0)
        self.t_fus = self._make_tower("fusion_mixer.", cm, self.Nf, 2048)
        # (B, N, K) rows == a (B, 1, N, K) image cut into (1, K) patches
        self.e_time = EmbedRuntime(1, self.Nt, ct["embedding_dim"], 1, ct["embedding_dim"], self.D, self.prec)
        self.e_time.bind_params(self.params["time_mixer.proj.weight"], self.params["time_mixer.proj.bias"])
        self.e_time.bind_grads(self.grads["time_mixer.proj.weight"], self.grads["time_mixer.proj.bias"])
        nb = cs["num_blocks"]
        keys = [f"static_extractor.module_list.{3 * i}." for i in range(nb)] + [f"static_extractor.module_list.{3 * nb}."]
        self.mlp = MlpRuntime([cs["input_dim"]] + [cs["hidden_dim"]] * nb + [cs["output_dim"]], True, self.p_drop, 4096)
        self.mlp.bind([(self.params[k + "weight"], self.params[k + "bias"]) for k in keys],
                      [(self.grads[k + "weight"], self.grads[k + "bias"]) for k in keys], self.B)
        B, D, dev = self.B, self.D, self.device
        f = lambda *s: torch.zeros(*s, device=dev)
        self.x0_time = f(B * self.Nt, D)
        self.fused, self.fus_out = f(B, self.Nf, D), f(B, self.Nf, D)
        self.pool_static, self.pool_time, self.pool_fus = f(B, D), f(B, D), f(B, D)
        self.dpool_static, self.dpool_time, self.dpool_fus = f(B, D), f(B, D), f(B, D)
        self.d_fused = f(B, self.Nf, D)
        self.dx0_time = f(B * self.Nt, D)
        self.preds = torch.zeros(3, B, dtype=torch.int32, device=dev)

    def pack(self):
        towers, embeds = [self.t_time, self.t_fus], [self.e_time]
        if can_pack_all(towers, embeds):
            pack_all(towers, embeds)
            return
        for m in towers + embeds:
            m.pack(force=True)

    def _adam_pack_modules(self):
        towers, embeds = [self.t_time, self.t_fus], [self.e_time]
        return (towers, embeds) if can_pack_all(towers, embeds) else None

    def _forward(self, static, time, labels, training: bool, with_grad: bool, prologue: bool = False):
        B, D = self.B, self.D
        sd = self.drop_step if training else None
        fs = self.Nf * D
        time_part = self.fused.view(-1)[D:]                 # tokens 1..Nt of every sample
        main, side, _ = self._streams(self._conc_fwd)
        head = None
        if prologue and side is main:
            # one stream: the step prologue rides in the input projection's launch, which then goes first (it reads none of the
            # counters; the static MLP draws its dropout masks from the bumped one)
            head = (self.adam_state, self.drop_step, self.losses)
            self.e_time.forward(time, B, self.x0_time, step_head=head)
        elif prologue:
            self._prologue()
        ride = self._mlp_ride and side is main              # one stream: the MLP's workgroups ride in the time tower's token-mixing launch
        side.wait_stream(main)
        with torch.cuda.stream(side):                       # the static MLP beside the time tower: token 0 + its head's input
            (self.mlp.forward_ride if ride else self.mlp.forward)(static, B, self.fused, fs, self.pool_static, training, self.seed, 0, sd)
        if head is None:
            self.e_time.forward(time, B, self.x0_time)
        hp = self._heads_pool                               # the heads pool the two towers' outputs themselves (m2m_head.tokens)
        self.t_time.forward(self.x0_time, self.Nt * D, B, time_part, fs, None if hp else self.pool_time, training, self.seed, 0, sd)
        if ride:
            self.mlp.ride_flush()                           # (a launch of its own if no token launch took it)
        main.wait_stream(side)
        self.t_fus.forward(self.fused, fs, B, self.fus_out, fs, None if hp else self.pool_fus, training, self.seed, 0, sd)
        hw = self.head_weights
        heads = [self._head("static", self.pool_static, self.dpool_static, hw["static"], with_grad),
                 self._head("time", self.pool_time, self.dpool_time, hw["time"], with_grad),
                 self._head("fusion", self.pool_fus, self.dpool_fus, hw["fusion"], with_grad)]
        if hp:
            heads[1]["tokens"] = (time_part.data_ptr(), self.Nt, fs)
            heads[2]["tokens"] = (self.fus_out, self.Nf, fs)
        heads_ce(heads, labels, B, D, self.K, out=(self.logits, self.losses, self.preds), zero_losses=not training)

    def _backward(self, static, time, fused_update: bool = False):
        B, D = self.B, self.D
        fs = self.Nf * D
        sd = self.drop_step
        self.t_fus.backward(B, None, 0, self.dpool_fus, self.d_fused, fs, self.seed, 0, sd)
        d_time_part = self.d_fused.view(-1)[D:]
        if not (self._conc_bwd and self.concurrent) and self._merged_tail:
            # one stream (small batch): both towers' weight gradients in ONE launch, ONE Adam over the flat buffer, ONE re-pack
            # (seven launches less than the per-segment form the three-stream step needs)
            (self.mlp.backward_ride if self._mlp_ride else self.mlp.backward)(static, B, self.d_fused, fs, self.dpool_static)
            self.t_time.backward(B, d_time_part, fs, self.dpool_time, self.dx0_time, self.Nt * D, self.seed, 0, sd)
            if self._mlp_ride:
                self.mlp.ride_flush()
            if os.environ.get("M2M_MIMIC_EMBED_WGRAD_MERGED", "1") != "0":
                # the input projection's weight gradient in the towers' weight-gradient launch (one launch less)
                towers_wgrad([self.t_fus, self.t_time], B, embeds=[self.e_time], inputs=[time], d_x0s=[self.dx0_time],
                             seed=self.seed, step=0, step_dev=sd)
            else:
                towers_wgrad([self.t_fus, self.t_time], B, seed=self.seed, step=0, step_dev=sd)
                self.e_time.wgrad(time, self.dx0_time, B)
            if fused_update:
                self._update(1.0)                       # one launch for small models (config_fused_update), else Adam + pack_all
            return
        main, s_b, s_f = self._streams(self._conc_bwd)
        s_b.wait_stream(main)
        s_f.wait_stream(main)
        with torch.cuda.stream(s_b):                        # static MLP: gradient of token 0 + of its own head
            self.mlp.backward(static, B, self.d_fused, fs, self.dpool_static)
            if fused_update:
                self._adam(*self.segments["static"], 1.0, False)
        with torch.cuda.stream(s_f):
            self.t_fus.wgrad(B, self.seed, 0, sd)
            if fused_update:
                self._adam(*self.segments["fusion"], 1.0, False)
                self.t_fus.pack(force=True)
        self.t_time.backward(B, d_time_part, fs, self.dpool_time, self.dx0_time, self.Nt * D, self.seed, 0, sd)
        self.t_time.wgrad(B, self.seed, 0, sd)
        self.e_time.wgrad(time, self.dx0_time, B)
        if fused_update:
            self._adam(*self.segments["time"], 1.0, False)
            self.t_time.pack(force=True)
            self.e_time.pack(force=True)
        main.wait_stream(s_b)
        main.wait_stream(s_f)

    @torch.no_grad()
    def evaluate(self, static, time, labels):
        self._forward(static, time, labels, False, False)
        return {"logits": self.logits[2], "logits_static": self.logits[0], "logits_time": self.logits[1],
                "loss_static": self.losses[0], "loss_time": self.losses[1], "loss_fusion": self.losses[2],
                "loss": self.losses[3], "preds": self.preds[2]}
