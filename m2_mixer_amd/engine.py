"""Fused AV-MNIST M2-Mixer training engine: the whole `shared_step` + backward + Adam of
AVMnistMixerMultiLoss (reference: models/avmnist.py:236-312, :413-422) as a fixed sequence of
libm2mixer launches over flat fp32 parameter / gradient / Adam-state buffers.

What changes relative to the module path (modules/mixer.py + torch autograd):
  * parameters are views into ONE flat buffer (names = the reference's state-dict keys, creation
    order of models/avmnist.py:181-191), gradients into ONE flat buffer -> one memset, one RCCL
    all-reduce, one Adam launch;
  * the image / audio towers write their outputs straight into the halves of the fused (B, Ni+Na, D)
    buffer (ConcatFusion costs nothing) and hand the token means to the heads kernel;
  * the three heads, their cross-entropies and gradients are one launch;
  * the step can be captured into a hipGraph (torch.cuda.CUDAGraph); the dropout step counter, the Adam
    step counter and the learning rate live in device memory so replays stay correct.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import _lib as L
from . import config
from .runtime import BLOCK_FIELDS, BLOCK_KEYS, EmbedRuntime, TowerRuntime, block_param_shapes, heads_ce


def _num_patch(c: dict) -> int:
    return (c["image_size"][0] // c["patch_size"]) * (c["image_size"][1] // c["patch_size"])


def avmnist_param_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    """state-dict key -> shape in the reference's creation order (models/avmnist.py:181-191)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    ni, na = _num_patch(cfg["image"]), _num_patch(cfg["audio"])

    def tower(prefix, c, N, patch):
        D = c["hidden_dim"]
        if patch:
            s[prefix + "to_patch_embedding.0.weight"] = (D, c["in_channels"], c["patch_size"], c["patch_size"])
            s[prefix + "to_patch_embedding.0.bias"] = (D,)
        shapes = block_param_shapes(D, N, c["token_dim"], c["channel_dim"])
        for i in range(c["num_mixers"]):
            for f in BLOCK_FIELDS:
                s[f"{prefix}mixer_blocks.{i}.{BLOCK_KEYS[f]}"] = shapes[f]
        s[prefix + "layer_norm.weight"] = (D,)
        s[prefix + "layer_norm.bias"] = (D,)

    tower("image_mixer.", cfg["image"], ni, True)
    tower("audio_mixer.", cfg["audio"], na, True)
    tower("fusion_mixer.", cfg["multimodal"], ni + na, False)
    K = cfg["num_classes"]
    s["classifier_image.weight"] = (K, cfg["image"]["hidden_dim"])
    s["classifier_image.bias"] = (K,)
    s["classifier_audio.weight"] = (K, cfg["audio"]["hidden_dim"])
    s["classifier_audio.bias"] = (K,)
    s["classifier_fusion.classifer.weight"] = (K, cfg["multimodal"]["hidden_dim"])
    s["classifier_fusion.classifer.bias"] = (K,)
    return s


class AVMnistEngine:
    """cfg: {'dropout', 'num_classes', 'image': {...}, 'audio': {...}, 'multimodal': {...}} with the keys of
    cfg/avmnist/avmnist_m2-mixer_*.yml (model.modalities.*)."""

    def __init__(self, cfg: dict, batch_size: int, device="cuda:0", precision: Optional[str] = None,
                 lr: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 fusion_loss_weight: float = 1.0 / 3, seed: int = 42, init: bool = True):
        self.cfg, self.B = cfg, int(batch_size)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("AVMnistEngine runs on the GPU only (MI355X); there is no CPU path")
        L.lib()  # fail loudly if the HIP library is absent
        self.prec = config.prec_id(precision)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.p_drop = float(cfg.get("dropout", 0.0))
        ci, ca, cm = cfg["image"], cfg["audio"], cfg["multimodal"]
        self.D = ci["hidden_dim"]
        if not (ca["hidden_dim"] == self.D == cm["hidden_dim"]):
            raise RuntimeError("image / audio / fusion hidden_dim must agree (ConcatFusion on dim 1)")
        self.Ni, self.Na = _num_patch(ci), _num_patch(ca)
        self.Nf = self.Ni + self.Na
        self.K = cfg["num_classes"]
        w = fusion_loss_weight
        ow = (1 - w) / 2
        # loss = (w Lf + ow Li + ow La) * 3      (models/avmnist.py:289-290)
        self.head_weights = {"image": 3 * ow, "audio": 3 * ow, "fusion": 3 * w}

        # ---- flat parameter / gradient / Adam buffers ----
        self.shapes = avmnist_param_shapes(cfg)
        n = sum(int(torch.Size(s).numel()) for s in self.shapes.values())
        self.n_params = n
        dev = self.device
        self.flat_p = torch.zeros(n, device=dev)
        self.flat_g = torch.zeros(n, device=dev)
        self.flat_m = torch.zeros(n, device=dev)
        self.flat_v = torch.zeros(n, device=dev)
        self.params: Dict[str, torch.Tensor] = OrderedDict()
        self.grads: Dict[str, torch.Tensor] = OrderedDict()
        off = 0
        for k, shp in self.shapes.items():
            cnt = int(torch.Size(shp).numel())
            self.params[k] = self.flat_p[off:off + cnt].view(shp)
            self.grads[k] = self.flat_g[off:off + cnt].view(shp)
            off += cnt
        # contiguous segments of the flat buffers: [image tower | audio tower | fusion tower + heads]
        bounds, off = {}, 0
        for k, shp in self.shapes.items():
            seg = "image" if k.startswith("image_mixer.") else ("audio" if k.startswith("audio_mixer.") else "fusion")
            cnt = int(torch.Size(shp).numel())
            lo, hi = bounds.get(seg, (off, off))
            bounds[seg] = (min(lo, off), off + cnt)
            off += cnt
        self.segments = bounds
        assert bounds["image"][1] == bounds["audio"][0] and bounds["audio"][1] == bounds["fusion"][0] and bounds["fusion"][1] == n
        self.adam_state = torch.tensor([0.0, lr, 0.0, 0.0], device=dev)     # [step, lr, -, -]
        self.drop_step = torch.zeros(1, dtype=torch.int32, device=dev)       # device-side dropout step counter
        self.seed = seed & 0xFFFFFFFF
        if init:
            self.reset_parameters(seed)

        # ---- towers ----
        def make_tower(prefix, c, N, site):
            rt = TowerRuntime(self.D, N, c["token_dim"], c["channel_dim"], c["num_mixers"], True, self.p_drop,
                              self.prec, site)
            blocks = [{f: self.params[f"{prefix}mixer_blocks.{i}.{BLOCK_KEYS[f]}"] for f in BLOCK_FIELDS}
                      for i in range(c["num_mixers"])]
            rt.bind_params(blocks, (self.params[prefix + "layer_norm.weight"], self.params[prefix + "layer_norm.bias"]))
            rt.bind_grad_tensors([{f: self.grads[f"{prefix}mixer_blocks.{i}.{BLOCK_KEYS[f]}"] for f in BLOCK_FIELDS}
                                  for i in range(c["num_mixers"])],
                                 (self.grads[prefix + "layer_norm.weight"], self.grads[prefix + "layer_norm.bias"]))
            rt.ensure_buffers(self.B)
            return rt

        self.t_img = make_tower("image_mixer.", ci, self.Ni, 0)
        self.t_aud = make_tower("audio_mixer.", ca, self.Na, 1024)
        self.t_fus = make_tower("fusion_mixer.", cm, self.Nf, 2048)

        def make_embed(prefix, c):
            e = EmbedRuntime(c["in_channels"], c["image_size"][0], c["image_size"][1], c["patch_size"], c["patch_size"],
                             self.D, self.prec)
            e.bind_params(self.params[prefix + "to_patch_embedding.0.weight"], self.params[prefix + "to_patch_embedding.0.bias"])
            e.bind_grads(self.grads[prefix + "to_patch_embedding.0.weight"], self.grads[prefix + "to_patch_embedding.0.bias"])
            return e

        self.e_img = make_embed("image_mixer.", ci)
        self.e_aud = make_embed("audio_mixer.", ca)

        # ---- workspaces ----
        B, D = self.B, self.D
        f = lambda *s: torch.zeros(*s, device=dev)
        self.x0_img, self.x0_aud = f(B * self.Ni, D), f(B * self.Na, D)
        self.fused, self.fus_out = f(B, self.Nf, D), f(B, self.Nf, D)
        self.pool_img, self.pool_aud, self.pool_fus = f(B, D), f(B, D), f(B, D)
        self.dpool_img, self.dpool_aud, self.dpool_fus = f(B, D), f(B, D), f(B, D)
        self.d_fused = f(B, self.Nf, D)
        self.dx0_img, self.dx0_aud = f(B * self.Ni, D), f(B * self.Na, D)
        self.logits = f(3, B, self.K)
        self.losses = f(4)
        self.preds = torch.zeros(3, B, dtype=torch.int32, device=dev)
        self._graph = None
        self._static = None
        # the audio tower / the fusion weight gradients run beside the image tower on side streams
        self.s_aud = torch.cuda.Stream(device=dev)
        self.s_fus = torch.cuda.Stream(device=dev)
        self.concurrent = True
        self.pack()

    # ---- parameters --------------------------------------------------------------------------------------
    def reset_parameters(self, seed: int = 42):
        """torch default init (Linear / Conv2d: kaiming-uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)); LayerNorm 1/0),
        drawn on the CPU generator under `seed` in creation order (run.py:32 seeds 42)."""
        gen = torch.Generator().manual_seed(seed)
        for k, shp in self.shapes.items():
            is_ln = ("layer_norm." in k) or k.endswith("token_mix.0.weight") or k.endswith("token_mix.0.bias") \
                or k.endswith("channel_mix.0.weight") or k.endswith("channel_mix.0.bias")
            if is_ln:
                v = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
            else:
                wshape = self.shapes[k[:-4] + "weight"] if k.endswith("bias") else shp
                fan_in = int(torch.Size(wshape[1:]).numel())
                bound = 1.0 / fan_in ** 0.5
                v = (torch.rand(shp, generator=gen) * 2 - 1) * bound
            self.params[k].copy_(v)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        missing = [k for k in self.shapes if k not in sd]
        extra = [k for k in sd if k not in self.shapes]
        if missing or extra:
            raise KeyError(f"state dict mismatch: missing {missing[:4]}, unexpected {extra[:4]}")
        for k in self.shapes:
            self.params[k].copy_(sd[k].to(self.device, torch.float32).reshape(self.shapes[k]))
        self.pack()

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, v.detach().clone()) for k, v in self.params.items())

    def set_lr(self, lr: float):
        self.adam_state[1] = lr

    def pack(self):
        """Rebuild the packed MFMA-operand copies from the fp32 masters (three independent launches + two tiny ones:
        spread over the side streams)."""
        main = torch.cuda.current_stream()
        s_a = self.s_aud if self.concurrent else main
        s_f = self.s_fus if self.concurrent else main
        s_a.wait_stream(main)
        s_f.wait_stream(main)
        with torch.cuda.stream(s_a):
            self.t_aud.pack(force=True)
            self.e_aud.pack(force=True)
        with torch.cuda.stream(s_f):
            self.t_fus.pack(force=True)
        self.t_img.pack(force=True)
        self.e_img.pack(force=True)
        main.wait_stream(s_a)
        main.wait_stream(s_f)

    # ---- one training step (enqueue only; no host synchronisation) ----------------------------------------
    def _forward(self, image, audio, labels, training: bool, with_grad: bool):
        B, D = self.B, self.D
        sd = self.drop_step if training else None
        fs = self.Nf * D
        aud_half = self.fused.view(-1)[self.Ni * D:]
        main = torch.cuda.current_stream()
        side = self.s_aud if self.concurrent else main
        side.wait_stream(main)
        with torch.cuda.stream(side):                       # audio tower beside the image tower
            self.e_aud.forward(audio, B, self.x0_aud)
            self.t_aud.forward(self.x0_aud, self.Na * D, B, aud_half, fs, self.pool_aud, training, self.seed, 0, sd)
        self.e_img.forward(image, B, self.x0_img)
        self.t_img.forward(self.x0_img, self.Ni * D, B, self.fused, fs, self.pool_img, training, self.seed, 0, sd)
        main.wait_stream(side)
        self.t_fus.forward(self.fused, fs, B, self.fus_out, fs, self.pool_fus, training, self.seed, 0, sd)
        P, Gr = self.params, self.grads
        heads = []
        for name, pooled, dp, key in (("image", self.pool_img, self.dpool_img, "classifier_image."),
                                      ("audio", self.pool_aud, self.dpool_aud, "classifier_audio."),
                                      ("fusion", self.pool_fus, self.dpool_fus, "classifier_fusion.classifer.")):
            heads.append(dict(pooled=pooled, w=P[key + "weight"], b=P[key + "bias"], g_w=Gr[key + "weight"],
                              g_b=Gr[key + "bias"], d_pooled=dp if with_grad else None, weight=self.head_weights[name]))
        heads_ce(heads, labels, B, D, self.K, out=(self.logits, self.losses, self.preds))

    def _adam(self, lo: int, hi: int, grad_scale: float, bump: bool):
        n = hi - lo
        off = lo * 4
        L.check(L.lib().m2m_adam_step(self.flat_p.data_ptr() + off, self.flat_g.data_ptr() + off, self.flat_m.data_ptr() + off,
                                      self.flat_v.data_ptr() + off, n, self.adam_state.data_ptr(), self.betas[0], self.betas[1],
                                      self.eps, self.weight_decay, -abs(grad_scale), int(bump), L.stream_ptr()), "adam_step")

    def _backward(self, image, audio, fused_update: bool = False):
        """Backward of the whole model.  fused_update: also apply Adam + re-pack per tower as soon as that tower's
        gradients are complete (single-GPU path; with a gradient all-reduce the update is a separate phase)."""
        B, D = self.B, self.D
        fs = self.Nf * D
        sd = self.drop_step
        self.t_fus.backward(B, None, 0, self.dpool_fus, self.d_fused, fs, self.seed, 0, sd)
        d_aud_half = self.d_fused.view(-1)[self.Ni * D:]
        main = torch.cuda.current_stream()
        s_a = self.s_aud if self.concurrent else main
        s_f = self.s_fus if self.concurrent else main
        # The two tower backward chains fill the chip (128 + 128 workgroups): nothing else runs beside them.
        # All weight-gradient launches (three towers, two embeddings) follow, spread over three streams.
        s_a.wait_stream(main)
        with torch.cuda.stream(s_a):
            self.t_aud.backward(B, d_aud_half, fs, self.dpool_aud, self.dx0_aud, self.Na * D, self.seed, 0, sd)
            ev_aud = torch.cuda.Event()
            ev_aud.record()
        self.t_img.backward(B, self.d_fused, fs, self.dpool_img, self.dx0_img, self.Ni * D, self.seed, 0, sd)
        if self.concurrent:
            main.wait_event(ev_aud)                         # both chains done
        if fused_update:
            self._adam(0, 0, 1.0, True)                     # advance the Adam step counter once
        s_a.wait_stream(main)
        s_f.wait_stream(main)
        with torch.cuda.stream(s_a):
            self.t_aud.wgrad(B, self.seed, 0, sd)
            self.e_aud.wgrad(audio, self.dx0_aud, B)
            if fused_update:
                self._adam(*self.segments["audio"], 1.0, False)
                self.t_aud.pack(force=True)
                self.e_aud.pack(force=True)
        with torch.cuda.stream(s_f):
            self.t_fus.wgrad(B, self.seed, 0, sd)
            if fused_update:
                self._adam(*self.segments["fusion"], 1.0, False)
                self.t_fus.pack(force=True)
        self.t_img.wgrad(B, self.seed, 0, sd)
        self.e_img.wgrad(image, self.dx0_img, B)
        if fused_update:
            self._adam(*self.segments["image"], 1.0, False)
            self.t_img.pack(force=True)
            self.e_img.pack(force=True)
        main.wait_stream(s_a)
        main.wait_stream(s_f)
        if fused_update:
            L.check(L.lib().m2m_counter_add(self.drop_step.data_ptr(), 1, L.stream_ptr()), "counter_add")

    def forward_backward(self, image, audio, labels):
        """forward (dropout on) -> multi-head loss -> backward; gradients are ADDED into flat_g, which must be
        zero on entry: it is cleared at construction and again by every optimizer_step (the Adam kernel clears
        each element it consumes), so no separate fill pass is needed."""
        self._forward(image, audio, labels, True, True)
        self._backward(image, audio)

    def fused_step(self, image, audio, labels):
        """forward + backward + Adam + re-pack with the per-tower updates overlapped with the remaining
        weight-gradient work (no gradient exchange: single-GPU training)."""
        self._forward(image, audio, labels, True, True)
        self._backward(image, audio, fused_update=True)
        return self.losses

    def optimizer_step(self, grad_scale: float = 1.0):
        self._adam(0, self.n_params, grad_scale, True)                        # negative scale inside: clears the gradients
        L.check(L.lib().m2m_counter_add(self.drop_step.data_ptr(), 1, L.stream_ptr()), "counter_add")
        self.pack()

    def train_step(self, image, audio, labels, grad_sync=None):
        """One optimisation step.  grad_sync: optional callable(flat_grad) doing the data-parallel
        all-reduce (parallel.GradSync); it returns the factor the summed gradient must be scaled by."""
        if grad_sync is None:
            return self.fused_step(image, audio, labels)
        self.forward_backward(image, audio, labels)
        self.optimizer_step(grad_sync(self.flat_g))
        return self.losses

    @torch.no_grad()
    def evaluate(self, image, audio, labels):
        """validation/test step: dropout off (models/avmnist.py shared_step in eval mode)."""
        self._forward(image, audio, labels, False, False)
        return {"logits": self.logits[2], "image_logits": self.logits[0], "audio_logits": self.logits[1],
                "loss_image": self.losses[0], "loss_audio": self.losses[1], "loss_fusion": self.losses[2],
                "loss": self.losses[3], "preds": self.preds[2], "preds_image": self.preds[0], "preds_audio": self.preds[1]}

    # ---- hipGraph capture -----------------------------------------------------------------------------------
    def capture(self, image, audio, labels, grad_sync=None):
        """Capture train_step on static input buffers; returns a callable replay(image, audio, labels).
        With a grad_sync the step is captured as two graphs with the all-reduce between them."""
        self._static = (image.clone(), audio.clone(), labels.clone())
        si, sa, sl = self._static
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            scale = 1.0
            for _ in range(2):                       # warm-up: lazy inits (LDS attributes, allocations) happen here
                if grad_sync is None:
                    self.fused_step(si, sa, sl)
                else:
                    self.forward_backward(si, sa, sl)
                    scale = grad_sync(self.flat_g)
                    self.optimizer_step(scale)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        if grad_sync is None:
            with torch.cuda.graph(g1):
                self.fused_step(si, sa, sl)
            graphs = (g1,)
        else:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                self.forward_backward(si, sa, sl)
            with torch.cuda.graph(g2):
                self.optimizer_step(scale)
            graphs = (g1, g2)
        self._graph = graphs

        def replay(image=None, audio=None, labels=None):
            if image is not None:
                si.copy_(image, non_blocking=True)
                sa.copy_(audio, non_blocking=True)
                sl.copy_(labels, non_blocking=True)
            graphs[0].replay()
            if grad_sync is not None:
                grad_sync(self.flat_g)
                graphs[1].replay()
            return self.losses

        return replay
