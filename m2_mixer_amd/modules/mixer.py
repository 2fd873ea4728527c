"""Mixer building blocks with the reference's module protocol, backed by libm2mixer.so.

Mirrors the public surface of the reference's modules/mixer.py -- class names, constructor
signatures (extra kwargs swallowed, as `get_block_by_name(**cfg)` splats the whole cfg dict),
`.num_patch`, `forward(x) -> (B, N, D)` and, exactly, the state-dict keys so
the published checkpoints load.  The sub-modules below exist to own parameters under those keys;
their torch forward() is never used: `forward` hands the tensors to the HIP tower kernels
(csrc/tower_fwd.hip, tower_bwd.hip, tower_wgrad.hip) through a torch.autograd.Function.

There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import itertools
from typing import List, Optional

import torch
from torch import nn

from .. import _lib as L
from .. import config
from ..runtime import BLOCK_FIELDS, BLOCK_KEYS, EmbedRuntime, TowerRuntime

_site_counter = itertools.count(0)


def _train_repack(module: nn.Module) -> bool:
    """Re-pack the operand copies regardless of the parameters' version counters?  (see _HipTower._runtime)"""
    return module.training and torch.is_grad_enabled()


def _param_holder_ff(dim: int, hidden: int, dropout: float, out_dim: Optional[int] = None) -> nn.Module:
    return FeedForward(dim, hidden, dropout, out_dim)


class FeedForward(nn.Module):
    """Linear -> GELU(erf) -> Dropout -> Linear -> Dropout (reference: modules/mixer.py:9-22).
    Parameter container (keys net.0.*, net.3.*); computed inside the fused kernels."""

    def __init__(self, dim, hidden_dim, dropout=0., out_dim=None):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, out_dim or dim), nn.Dropout(dropout))

    def forward(self, x):  # pragma: no cover - never on the product path
        raise RuntimeError("FeedForward is fused into the MixerBlock HIP kernels; call the enclosing block/tower")


def _block_tensors(block: "MixerBlock") -> dict:
    sd = dict(block.named_parameters())
    return {f: sd[BLOCK_KEYS[f]] for f in BLOCK_FIELDS}


class _TowerFunction(torch.autograd.Function):
    """x (B, N, D) -> blocks (+ final LayerNorm) -> (B, N, D).  Re-entrant: every forward that will be differentiated
    gets its own set of saved activations (TowerRuntime.fresh_saved), kept in the autograd ctx.  Towers deeper than one
    m2m_tower holds are a chain of runtimes (owner._rts)."""

    @staticmethod
    def forward(ctx, x, owner, need_grad, *params):
        # NB: grad mode is switched off inside Function.forward, so the caller decides `need_grad`
        rts: List[TowerRuntime] = owner._rts
        B = x.shape[0]
        N, D = rts[0].N, rts[0].D
        dropping = owner.training and owner.dropout_p > 0
        seed = config.dropout_seed()
        step, step_dev = owner._step_args(x.device) if dropping else (0, None)
        p_drop = float(owner.dropout_p) if dropping else 0.0
        saved = []
        cur = x
        for rt in rts:
            rt.desc.p_drop = p_drop
            out = torch.empty(B, N, D, device=x.device, dtype=torch.float32)
            rt.ensure_workspace(B)
            saved.append(rt.fresh_saved(B) if need_grad else None)
            if dropping and not need_grad:
                # dropout without autograd (train() under no_grad, e.g. MC-dropout passes): the kernels still save their
                # activations -- into a scratch set, NOT into the set a pending differentiated forward left bound in the
                # descriptor (its backward re-binds its own set: use_saved)
                rt.fresh_saved(B)
            rt.forward(cur, N * D, B, out, N * D, None, need_grad or dropping, seed, step, step_dev)
            cur = out
        ctx.saved = saved if need_grad else None
        ctx.owner, ctx.B, ctx.seed, ctx.step, ctx.step_dev, ctx.p_drop = owner, B, seed, step, step_dev, p_drop
        ctx.nparams = len(params)
        return cur

    @staticmethod
    def backward(ctx, dout):
        owner = ctx.owner
        rts: List[TowerRuntime] = owner._rts
        if ctx.saved is None:
            raise RuntimeError("backward through a tower forward that ran without gradients enabled")
        B, N, D = ctx.B, rts[0].N, rts[0].D
        grads_per_chunk = []
        g = dout.contiguous()
        for rt, saved in zip(reversed(rts), reversed(ctx.saved)):
            rt.ensure_buffers(B)            # operand images sized for THIS batch (a forward at another batch size may have
            rt.ensure_workspace(B)          # re-allocated them since), then this forward's own activations
            rt.use_saved(saved)
            rt.desc.p_drop = ctx.p_drop
            flat = torch.zeros(rt.grad_numel(), device=dout.device, dtype=torch.float32)
            views = rt.bind_grads(flat)
            dx = torch.empty(B, N, D, device=dout.device, dtype=torch.float32)
            rt.backward(B, g, N * D, None, dx, N * D, ctx.seed, ctx.step, ctx.step_dev)
            rt.wgrad(B, ctx.seed, ctx.step, ctx.step_dev)
            grads_per_chunk.append(views)
            g = dx
        # parameter order of _run_tower: the blocks' fields chunk by chunk, then the final LayerNorm (in the last chunk's views)
        views = [v for chunk in reversed(grads_per_chunk) for v in chunk]
        return (g, None, None) + tuple(views)


class _EmbedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, owner, w, b):
        rt: EmbedRuntime = owner._ert
        B = inp.shape[0]
        x0 = torch.empty(B, rt.N, rt.D, device=inp.device, dtype=torch.float32)
        rt.forward(inp, B, x0)
        ctx.owner, ctx.B = owner, B
        ctx.save_for_backward(inp)
        return x0

    @staticmethod
    def backward(ctx, dx0):
        rt: EmbedRuntime = ctx.owner._ert
        (inp,) = ctx.saved_tensors
        g_w = torch.zeros_like(rt._keep["w"])
        g_b = torch.zeros_like(rt._keep["b"])
        rt.bind_grads(g_w, g_b)
        rt.wgrad(inp, dx0.contiguous(), ctx.B)
        return None, None, g_w, g_b


class _HipTower(nn.Module):
    """Shared machinery: a list of MixerBlocks (+ optional final LayerNorm) run as one HIP tower."""

    def _init_tower(self, hidden_dim, num_patch, token_dim, channel_dim, dropout, precision=None):
        self.hidden_dim, self.token_dim, self.channel_dim = hidden_dim, token_dim, channel_dim
        self.dropout_p = float(dropout)
        self.precision = precision
        self._rt: Optional[TowerRuntime] = None
        self._rts: Optional[List[TowerRuntime]] = None
        self._drop_step = 0
        self._site_base = 1024 * next(_site_counter)

    def _bump_step(self) -> int:
        self._drop_step += 1
        return self._drop_step

    def _step_args(self, device):
        """(step, step_dev) of the next dropout draw: a host integer, or -- config.set_device_dropout_step(True), for steps
        replayed from a hipGraph -- 0 and a device counter that a one-thread launch has just advanced."""
        if not config.device_dropout_step():
            return self._bump_step(), None
        ctr = getattr(self, "_drop_counter", None)
        if ctr is None or ctr.device != device:
            ctr = self._drop_counter = torch.full((1,), self._drop_step, dtype=torch.int32, device=device)
        L.check(L.lib().m2m_counter_add(ctr.data_ptr(), 1, L.stream_ptr()), "counter_add")
        self._drop_step += 1                 # (host mirror: exact in eager mode; replays advance only the device counter)
        return 0, ctr

    def _tower_blocks(self) -> List["MixerBlock"]:
        raise NotImplementedError

    def _final_ln(self) -> Optional[nn.LayerNorm]:
        return getattr(self, "layer_norm", None)

    def _runtime(self) -> TowerRuntime:
        """The tower's TowerRuntime(s).  One m2m_tower holds at most L.MAX_BLOCKS MixerBlocks; a deeper tower (the reference's
        sweeps go to 16 mixers, sweeps/avmnist_mixer.yaml:19-35) is a chain of them -- `_rts`: consecutive chunks of blocks,
        the final LayerNorm on the last chunk -- and forward / backward walk the chain.  `_rt` stays the first chunk."""
        all_blocks = [_block_tensors(b) for b in self._tower_blocks()]
        ln = self._final_ln()
        lnf = (ln.weight, ln.bias) if ln is not None else None
        prec = config.prec_id(self.precision)
        chunks = [all_blocks[i:i + L.MAX_BLOCKS] for i in range(0, len(all_blocks), L.MAX_BLOCKS)] or [[]]
        rts = getattr(self, "_rts", None)
        if rts is None or len(rts) != len(chunks) or rts[0].prec != prec:
            rts = []
            for ci, blocks in enumerate(chunks):
                last = ci == len(chunks) - 1
                rt = TowerRuntime(self.hidden_dim, self.num_patch, self.token_dim, self.channel_dim, len(blocks),
                                  ln is not None and last, self.dropout_p, prec, self._site_base + 4 * L.MAX_BLOCKS * ci)
                rt.bind_params(blocks, lnf if last else None)
                rts.append(rt)
            self._rts = rts
        else:
            for ci, (rt, blocks) in enumerate(zip(rts, chunks)):
                l = lnf if ci == len(chunks) - 1 else None
                if rt.params_changed(blocks, l):
                    rt.bind_params(blocks, l)
        # Version counters catch in-place edits by ordinary ops (optimizers' foreach kernels, load_state_dict, p.mul_()) -- but NOT
        # torch's fused optimizers (torch.optim.Adam(fused=True) updates the values and leaves `_version` alone: measured on
        # torch 2.10 / ROCm).  A module in training mode with gradients enabled is about to be stepped: re-pack unconditionally
        # there (one ~12 us launch per tower, which a training step pays anyway); evaluation keeps the version check.
        force = _train_repack(self)
        for rt in self._rts:
            rt.pack(force=force)
        self._rt = self._rts[0]
        return self._rt

    def _run_tower(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("m2_mixer_amd modules run on the GPU only (MI355X); there is no CPU path")
        if x.dim() != 3 or x.shape[1] != self.num_patch or x.shape[2] != self.hidden_dim:
            raise RuntimeError(f"expected (B, {self.num_patch}, {self.hidden_dim}) tokens, got {tuple(x.shape)}")
        self._runtime()
        blocks = [_block_tensors(b) for b in self._tower_blocks()]
        params = [bp[f] for bp in blocks for f in BLOCK_FIELDS]
        ln = self._final_ln()
        if ln is not None:
            params += [ln.weight, ln.bias]
        need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return _TowerFunction.apply(x.contiguous().float(), self, need_grad, *params)


class MixerBlock(_HipTower):
    """token-mixing MLP + channel-mixing MLP with residuals (reference: modules/mixer.py:25-47)."""

    def __init__(self, hidden_dim, num_patch, token_dim, channel_dim, dropout=0.):
        super().__init__()
        self.num_patch = num_patch
        # index 1 / 3 of token_mix are the reference's parameter-free Rearrange layers
        self.token_mix = nn.Sequential(nn.LayerNorm(hidden_dim), nn.Identity(),
                                       _param_holder_ff(num_patch, token_dim, dropout), nn.Identity())
        self.channel_mix = nn.Sequential(nn.LayerNorm(hidden_dim), _param_holder_ff(hidden_dim, channel_dim, dropout))
        self._init_tower(hidden_dim, num_patch, token_dim, channel_dim, dropout)

    def _tower_blocks(self):
        return [self]

    def forward(self, x):
        return self._run_tower(x)


def _make_blocks(n, hidden_dim, num_patch, token_dim, channel_dim, dropout) -> nn.ModuleList:
    return nn.ModuleList([MixerBlock(hidden_dim, num_patch, token_dim, channel_dim, dropout=dropout) for _ in range(n)])


class FusionMixer(_HipTower):
    """MixerBlocks + LayerNorm over already-embedded tokens (reference: modules/mixer.py:112-132)."""

    def __init__(self, hidden_dim, num_patches, num_mixers, token_dim, channel_dim, dropout=0., **kwargs):
        super().__init__()
        self.num_patch = num_patches
        self.mixer_blocks = _make_blocks(num_mixers, hidden_dim, num_patches, token_dim, channel_dim, dropout)
        self.layer_norm = nn.LayerNorm(hidden_dim)
        self._init_tower(hidden_dim, num_patches, token_dim, channel_dim, dropout, kwargs.get("precision"))

    def _tower_blocks(self):
        return list(self.mixer_blocks)

    def forward(self, x):
        return self._run_tower(x)


class MLPMixer(_HipTower):
    """Conv patch embedding -> MixerBlocks -> LayerNorm (reference: modules/mixer.py:135-162)."""

    def __init__(self, in_channels, hidden_dim, patch_size, image_size, num_mixers, token_dim, channel_dim,
                 dropout=0., **kwargs):
        super().__init__()
        if image_size[0] % patch_size or image_size[1] % patch_size:
            raise AssertionError('Image dimensions must be divisible by the patch size.')
        self.in_channels, self.patch_size, self.image_size = in_channels, patch_size, list(image_size)
        self.num_patch = (image_size[0] // patch_size) * (image_size[1] // patch_size)
        # index 1 is the reference's Rearrange('b c h w -> b (h w) c')
        self.to_patch_embedding = nn.Sequential(nn.Conv2d(in_channels, hidden_dim, patch_size, patch_size), nn.Identity())
        self.mixer_blocks = _make_blocks(num_mixers, hidden_dim, self.num_patch, token_dim, channel_dim, dropout)
        self.layer_norm = nn.LayerNorm(hidden_dim)
        self._init_tower(hidden_dim, self.num_patch, token_dim, channel_dim, dropout, kwargs.get("precision"))
        self._ert: Optional[EmbedRuntime] = None

    def _tower_blocks(self):
        return list(self.mixer_blocks)

    def _embed(self, x):
        conv = self.to_patch_embedding[0]
        prec = config.prec_id(self.precision)
        if self._ert is None or self._ert.prec != prec:
            self._ert = EmbedRuntime(self.in_channels, self.image_size[0], self.image_size[1], self.patch_size,
                                     self.patch_size, self.hidden_dim, prec)
            self._ert.bind_params(conv.weight, conv.bias)
        elif self._ert.params_changed(conv.weight, conv.bias):
            self._ert.bind_params(conv.weight, conv.bias)
        self._ert.pack(force=_train_repack(self))
        return _EmbedFunction.apply(x.contiguous().float(), self, conv.weight, conv.bias)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("m2_mixer_amd modules run on the GPU only (MI355X); there is no CPU path")
        if tuple(x.shape[1:]) != (self.in_channels, self.image_size[0], self.image_size[1]):
            raise RuntimeError(f"expected (B, {self.in_channels}, {self.image_size[0]}, {self.image_size[1]}), got {tuple(x.shape)}")
        return self._run_tower(self._embed(x))


class MLPMixerNoPatching(_HipTower):
    """Linear projection of given tokens -> MixerBlocks -> LayerNorm (reference: modules/mixer.py:165-186)."""

    def __init__(self, hidden_dim, num_patch, num_mixers, token_dim, channel_dim, embedding_dim, proj_dim,
                 dropout=0., **kwargs):
        super().__init__()
        if proj_dim != hidden_dim:
            raise ValueError("MLPMixerNoPatching needs proj_dim == hidden_dim (as in the reference's forward)")
        self.num_patch = num_patch
        self.embedding_dim = embedding_dim
        self.proj = nn.Linear(embedding_dim, proj_dim)
        self.mixer_blocks = _make_blocks(num_mixers, hidden_dim, num_patch, token_dim, channel_dim, dropout)
        self.layer_norm = nn.LayerNorm(hidden_dim)
        self._init_tower(hidden_dim, num_patch, token_dim, channel_dim, dropout, kwargs.get("precision"))
        self._ert: Optional[EmbedRuntime] = None

    def _tower_blocks(self):
        return list(self.mixer_blocks)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("m2_mixer_amd modules run on the GPU only (MI355X); there is no CPU path")
        prec = config.prec_id(self.precision)
        if self._ert is None or self._ert.prec != prec:
            # (B, N, K) rows == a (B, 1, N, K) image cut into (1, K) patches
            self._ert = EmbedRuntime(1, self.num_patch, self.embedding_dim, 1, self.embedding_dim, self.hidden_dim, prec)
            self._ert.bind_params(self.proj.weight, self.proj.bias)
        elif self._ert.params_changed(self.proj.weight, self.proj.bias):
            self._ert.bind_params(self.proj.weight, self.proj.bias)
        self._ert.pack(force=_train_repack(self))
        x0 = _EmbedFunction.apply(x.contiguous().float(), self, self.proj.weight, self.proj.bias)
        return self._run_tower(x0)
