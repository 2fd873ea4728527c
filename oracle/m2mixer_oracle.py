"""CPU oracle for the M2-Mixer hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain fp32/fp64 tensor arithmetic on the CPU, what the
reference (bezirganyan/m2-mixer) computes on the training hot path.  It is the
checker that tests/, __graft_entry__.smoke() and bench.py's baseline legs
(cpu_baseline; the same eager ops timed on the GPU as the "vs eager" comparator)
use; nothing under m2_mixer_amd/ imports it and the product path never routes
through it.

Parity status: PINNED.  tests/golden/*.npz were produced by importing the
reference's own `modules` package in the build container
(tests/golden/make_golden.py) and tests/test_oracle_golden.py checks every
function below against them.  The reference's own tests pin nothing numeric on
this path, so those fixtures are the pin.

Every function cites the reference lines it follows (paths relative to the
reference checkout).  Parameters are passed as dicts keyed by the reference's
state-dict key names so one seeded parameter set feeds reference, oracle and
HIP path alike.

Only basic tensor ops are used (matmul, mean, erf, exp ...) -- no torch.nn
layers -- so the arithmetic that has to be matched is spelled out.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]

LN_EPS = 1e-5  # torch.nn.LayerNorm default, modules/mixer.py:31,38,153


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = LN_EPS) -> Tensor:
    """nn.LayerNorm(hidden_dim) over the last axis, biased variance.
    modules/mixer.py:31 (token_mix.0), :38 (channel_mix.0), :153 (tower LN)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * weight + bias


def gelu(x: Tensor) -> Tensor:
    """nn.GELU() with approximate='none' (exact erf).  modules/mixer.py:15."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """nn.Linear: y = x W^T + b with W stored (out, in).  modules/mixer.py:14,17."""
    y = x @ weight.t()
    return y if bias is None else y + bias


def dropout(x: Tensor, p: float, mask: Optional[Tensor]) -> Tensor:
    """nn.Dropout(p) in training mode with an explicit keep-mask (1 keep, 0 drop);
    mask=None means eval / p == 0.  modules/mixer.py:16,18."""
    if mask is None or p == 0.0:
        return x
    return x * mask * (1.0 / (1.0 - p))


def feed_forward(x: Tensor, p: Params, prefix: str, drop_p: float = 0.0,
                 masks: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """FeedForward: Linear -> GELU -> Dropout -> Linear -> Dropout.
    modules/mixer.py:9-22; state-dict keys `<prefix>net.0.*`, `<prefix>net.3.*`."""
    h = gelu(linear(x, p[prefix + "net.0.weight"], p[prefix + "net.0.bias"]))
    h = dropout(h, drop_p, None if masks is None else masks[0])
    y = linear(h, p[prefix + "net.3.weight"], p[prefix + "net.3.bias"])
    return dropout(y, drop_p, None if masks is None else masks[1])


def mixer_block(x: Tensor, p: Params, prefix: str = "", drop_p: float = 0.0,
                masks: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """MixerBlock.forward, modules/mixer.py:25-47.

    x: (B, N, D).  token_mix = LN -> 'b n d -> b d n' -> FeedForward(N, T) ->
    'b d n -> b n d' (:30-35); channel_mix = LN -> FeedForward(D, C) (:37-40);
    both added residually (:43, :45).

    masks (training with dropout): dict with keep-masks
      tok_h (B, D, T), tok_o (B, D, N), ch_h (B, N, C), ch_o (B, N, D).
    """
    u = layer_norm(x, p[prefix + "token_mix.0.weight"], p[prefix + "token_mix.0.bias"])
    u = u.transpose(1, 2)                                            # b n d -> b d n
    tm = None if masks is None else (masks["tok_h"], masks["tok_o"])
    u = feed_forward(u, p, prefix + "token_mix.2.", drop_p, tm)
    x = x + u.transpose(1, 2)                                        # b d n -> b n d
    a = layer_norm(x, p[prefix + "channel_mix.0.weight"], p[prefix + "channel_mix.0.bias"])
    cm = None if masks is None else (masks["ch_h"], masks["ch_o"])
    x = x + feed_forward(a, p, prefix + "channel_mix.1.", drop_p, cm)
    return x


def patchify(img: Tensor, patch: int) -> Tensor:
    """(B, C, H, W) -> (B, N, C*p*p) with N = (H/p)*(W/p) in (h, w) raster order
    and the inner axis ordered (c, ph, pw): the unfold that makes
    Conv2d(C, D, p, stride=p) + Rearrange('b c h w -> b (h w) c') a plain
    Linear(C*p*p, D).  modules/mixer.py:143-146."""
    B, C, H, W = img.shape
    gh, gw = H // patch, W // patch
    x = img.reshape(B, C, gh, patch, gw, patch)
    x = x.permute(0, 2, 4, 1, 3, 5)                                  # b gh gw c ph pw
    return x.reshape(B, gh * gw, C * patch * patch)


def patch_embed(img: Tensor, weight: Tensor, bias: Tensor, patch: int) -> Tensor:
    """to_patch_embedding, modules/mixer.py:143-146. weight: (D, C, p, p)."""
    return linear(patchify(img, patch), weight.reshape(weight.shape[0], -1), bias)


def _blocks(x: Tensor, p: Params, prefix: str, num_mixers: int, drop_p: float,
            masks: Optional[List[Dict[str, Tensor]]]) -> Tensor:
    for i in range(num_mixers):                                      # mixer.py:128-129,158-159,182-183
        x = mixer_block(x, p, f"{prefix}mixer_blocks.{i}.", drop_p,
                        None if masks is None else masks[i])
    return layer_norm(x, p[prefix + "layer_norm.weight"], p[prefix + "layer_norm.bias"])


def mlp_mixer(img: Tensor, p: Params, prefix: str, patch: int, num_mixers: int,
              drop_p: float = 0.0, masks=None) -> Tensor:
    """MLPMixer.forward, modules/mixer.py:155-162."""
    x = patch_embed(img, p[prefix + "to_patch_embedding.0.weight"],
                    p[prefix + "to_patch_embedding.0.bias"], patch)
    return _blocks(x, p, prefix, num_mixers, drop_p, masks)


def fusion_mixer(x: Tensor, p: Params, prefix: str, num_mixers: int,
                 drop_p: float = 0.0, masks=None) -> Tensor:
    """FusionMixer.forward, modules/mixer.py:125-132."""
    return _blocks(x, p, prefix, num_mixers, drop_p, masks)


def mlp_mixer_no_patching(x: Tensor, p: Params, prefix: str, num_mixers: int,
                          drop_p: float = 0.0, masks=None) -> Tensor:
    """MLPMixerNoPatching.forward, modules/mixer.py:179-186."""
    x = linear(x, p[prefix + "proj.weight"], p[prefix + "proj.bias"])
    return _blocks(x, p, prefix, num_mixers, drop_p, masks)


def mlp(x: Tensor, p: Params, prefix: str, num_blocks: int, has_out: bool,
        drop_p: float = 0.0, masks: Optional[Sequence[Tensor]] = None) -> Tensor:
    """MLP.forward (Linear, ReLU, Dropout)*num_blocks [+ Linear], modules/mlp.py:4-27.
    module_list indices: block i -> Linear at 3*i; output Linear at 3*num_blocks."""
    for i in range(num_blocks):
        x = torch.relu(linear(x, p[f"{prefix}module_list.{3 * i}.weight"],
                              p[f"{prefix}module_list.{3 * i}.bias"]))
        x = dropout(x, drop_p, None if masks is None else masks[i])
    if has_out:
        k = 3 * num_blocks
        x = linear(x, p[f"{prefix}module_list.{k}.weight"], p[f"{prefix}module_list.{k}.bias"])
    return x


def concat_fusion(*xs: Tensor, dim: int = 1) -> Tensor:
    """ConcatFusion.__call__, modules/fusion.py:116-117."""
    return torch.cat(xs, dim=dim)


def standard_classifier(x: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    """StandardClassifier.forward: reshape(B,-1,D).mean(1) -> Linear.
    modules/classification.py:89-90 (attribute spelled `classifer`)."""
    return linear(x.reshape(x.shape[0], -1, x.shape[-1]).mean(dim=1), weight, bias)


def cross_entropy(logits: Tensor, labels: Tensor) -> Tensor:
    """nn.CrossEntropyLoss() (mean reduction), models/avmnist.py:192-194,276-278."""
    m = logits.max(dim=1, keepdim=True).values
    lse = (logits - m).exp().sum(dim=1).log() + m.squeeze(1)
    picked = logits.gather(1, labels.reshape(-1, 1).long()).squeeze(1)
    return (lse - picked).mean()


def bce_with_logits(logits: Tensor, targets: Tensor, pos_weight: Tensor) -> Tensor:
    """nn.BCEWithLogitsLoss(pos_weight=...), mean reduction. models/mmimdb.py:47-50."""
    # -[pw * y * log sigmoid(x) + (1-y) * log(1 - sigmoid(x))]
    log_sig = -torch.log1p(torch.exp(-logits.abs())) + torch.minimum(logits, torch.zeros_like(logits))
    log_one_minus = log_sig - logits
    return (-(pos_weight * targets * log_sig + (1 - targets) * log_one_minus)).mean()


# --------------------------------------------------------------------------
# task-level forward: AV-MNIST multi-head loss  (models/avmnist.py:236-312)
# --------------------------------------------------------------------------
def avmnist_forward(image: Tensor, audio: Tensor, labels: Tensor, p: Params, cfg: dict,
                    drop_p: float = 0.0, masks: Optional[dict] = None,
                    fusion_loss_weight: float = 1.0 / 3) -> Dict[str, Tensor]:
    """AVMnistMixerMultiLoss.shared_step without the freeze/mute/softadapt/gradblend
    branches (inactive in the S/M/B configs).  models/avmnist.py:259-298.

    cfg: {'image': {'patch_size','num_mixers'}, 'audio': {...}, 'multimodal': {'num_mixers'}}
    masks: {'image': [per block], 'audio': [...], 'fusion': [...]} or None.
    """
    g = (lambda k: None) if masks is None else masks.get
    img_tok = mlp_mixer(image, p, "image_mixer.", cfg["image"]["patch_size"],
                        cfg["image"]["num_mixers"], drop_p, g("image"))          # :259
    aud_tok = mlp_mixer(audio, p, "audio_mixer.", cfg["audio"]["patch_size"],
                        cfg["audio"]["num_mixers"], drop_p, g("audio"))          # :260
    fused = concat_fusion(img_tok, aud_tok, dim=1)                                # :263
    fus_tok = fusion_mixer(fused, p, "fusion_mixer.", cfg["multimodal"]["num_mixers"],
                           drop_p, g("fusion"))                                  # :264
    image_logits = linear(img_tok.mean(dim=1), p["classifier_image.weight"],
                          p["classifier_image.bias"])                            # :271
    audio_logits = linear(aud_tok.mean(dim=1), p["classifier_audio.weight"],
                          p["classifier_audio.bias"])                            # :272
    logits = standard_classifier(fus_tok, p["classifier_fusion.classifer.weight"],
                                 p["classifier_fusion.classifer.bias"])          # :273
    loss_image = cross_entropy(image_logits, labels)                             # :276
    loss_audio = cross_entropy(audio_logits, labels)                             # :277
    loss_fusion = cross_entropy(logits, labels)                                  # :278
    ow = (1 - fusion_loss_weight) / 2                                            # :289
    loss = (fusion_loss_weight * loss_fusion + ow * loss_image + ow * loss_audio) * 3  # :290
    return {
        "image_tokens": img_tok, "audio_tokens": aud_tok, "fusion_tokens": fus_tok,
        "image_logits": image_logits, "audio_logits": audio_logits, "logits": logits,
        "loss_image": loss_image, "loss_audio": loss_audio, "loss_fusion": loss_fusion,
        "loss": loss,
        "preds": torch.softmax(logits, dim=1).argmax(dim=1),                     # :296
        "preds_image": torch.softmax(image_logits, dim=1).argmax(dim=1),         # :297
        "preds_audio": torch.softmax(audio_logits, dim=1).argmax(dim=1),         # :298
    }


def mimic_forward(static: Tensor, time: Tensor, labels: Tensor, p: Params, cfg: dict,
                  drop_p: float = 0.0, masks: Optional[dict] = None,
                  fusion_loss_weight: float = 1.0 / 3) -> Dict[str, Tensor]:
    """MimicMixerMultiLoss.shared_step (no gradblend), models/mimic.py:93-142.
    Note: no '*3' on the weighted sum here (models/mimic.py:115-121)."""
    g = (lambda k: None) if masks is None else masks.get
    st = mlp(static, p, "static_extractor.", cfg["static"]["num_blocks"], True, drop_p, g("static"))   # :98
    tm = mlp_mixer_no_patching(time, p, "time_mixer.", cfg["time"]["num_mixers"], drop_p, g("time"))   # :99
    fused = concat_fusion(st.unsqueeze(1), tm, dim=1)                                                  # :102
    fus = fusion_mixer(fused, p, "fusion_mixer.", cfg["multimodal"]["num_mixers"], drop_p, g("fusion"))  # :103
    static_logits = linear(st, p["classifier_static.weight"], p["classifier_static.bias"])             # :106
    time_logits = linear(tm.mean(1), p["classifier_time.weight"], p["classifier_time.bias"])           # :107
    logits = standard_classifier(fus, p["classifier_fusion.classifer.weight"],
                                 p["classifier_fusion.classifer.bias"])                                # :108
    lf = cross_entropy(logits, labels)
    ls = cross_entropy(static_logits, labels)
    lt = cross_entropy(time_logits, labels)
    ow = (1 - fusion_loss_weight) / 2
    loss = fusion_loss_weight * lf + ow * ls + ow * lt
    return {"static_feat": st, "time_tokens": tm, "fusion_tokens": fus, "logits": logits,
            "logits_static": static_logits, "logits_time": time_logits,
            "loss_fusion": lf, "loss_static": ls, "loss_time": lt, "loss": loss,
            "preds": torch.softmax(logits, dim=1)}


def mmimdb_forward(image: Tensor, text: Tensor, labels: Tensor, p: Params, cfg: dict,
                   pos_weight: Tensor, drop_p: float = 0.0, masks: Optional[dict] = None) -> Dict[str, Tensor]:
    """MMIMDBMixerMultiLoss.shared_step, models/mmimdb.py:96-147 (sum of three BCE losses)."""
    g = (lambda k: None) if masks is None else masks.get
    it = mlp_mixer(image, p, "image_mixer.", cfg["image"]["patch_size"], cfg["image"]["num_mixers"], drop_p, g("image"))
    tt = mlp_mixer(text, p, "text_mixer.", cfg["text"]["patch_size"], cfg["text"]["num_mixers"], drop_p, g("text"))
    fused = concat_fusion(it, tt, dim=1)
    ft = fusion_mixer(fused, p, "fusion_mixer.", cfg["multimodal"]["num_mixers"], drop_p, g("fusion"))
    il = linear(it.mean(1), p["classifier_image.weight"], p["classifier_image.bias"])
    tl = linear(tt.mean(1), p["classifier_text.weight"], p["classifier_text.bias"])
    fl = standard_classifier(ft, p["classifier_fusion.classifer.weight"], p["classifier_fusion.classifer.bias"])
    y = labels.float()
    li, lt, lf = (bce_with_logits(il, y, pos_weight), bce_with_logits(tl, y, pos_weight),
                  bce_with_logits(fl, y, pos_weight))
    return {"image_logits": il, "text_logits": tl, "logits": fl, "loss_image": li, "loss_text": lt,
            "loss_fusion": lf, "loss": li + lt + lf, "preds": (torch.sigmoid(fl) > 0.5).long()}


# --------------------------------------------------------------------------
# optimizer: torch.optim.Adam as configured by models/avmnist.py:413-415
# --------------------------------------------------------------------------
def adam_step(param: Tensor, grad: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
              weight_decay: float = 0.0) -> Tuple[Tensor, Tensor, Tensor]:
    """One Adam update (no amsgrad), the default torch.optim.Adam algorithm:
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    p -= lr / (1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).   `step` is 1-based."""
    if weight_decay != 0.0:
        grad = grad + weight_decay * param
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return param - (lr / bc1) * m / denom, m, v


# --------------------------------------------------------------------------
# whole training step used by the parity tests and bench.py's cpu_baseline leg
# --------------------------------------------------------------------------
def avmnist_train_step(image, audio, labels, params: Params, cfg: dict, opt_state: dict,
                       lr: float, drop_p: float = 0.0, masks=None,
                       betas=(0.9, 0.999), eps=1e-8) -> Dict[str, Tensor]:
    """forward (avmnist_forward) + backward (autograd over the restated forward) +
    Adam on every parameter; mutates `params` and `opt_state` in place."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    out = avmnist_forward(image, audio, labels, leaves, cfg, drop_p, masks)
    out["loss"].backward()
    opt_state["step"] = opt_state.get("step", 0) + 1
    grads = {}
    for k, leaf in leaves.items():
        g = leaf.grad if leaf.grad is not None else torch.zeros_like(leaf)
        grads[k] = g
        m = opt_state.setdefault("m", {}).setdefault(k, torch.zeros_like(g))
        v = opt_state.setdefault("v", {}).setdefault(k, torch.zeros_like(g))
        newp, m2, v2 = adam_step(params[k], g, m, v, opt_state["step"], lr, betas[0], betas[1], eps)
        params[k] = newp.detach()
        opt_state["m"][k], opt_state["v"][k] = m2, v2
    out = {k: (v.detach() if isinstance(v, Tensor) else v) for k, v in out.items()}
    out["grads"] = grads
    return out


def avmnist_random_masks(cfg: dict, B: int, p: float, generator: Optional[torch.Generator] = None, device=None) -> dict:
    """Bernoulli keep-masks for every dropout site of the three towers (what nn.Dropout draws in the
    reference's train mode); used by bench.py's baseline legs so that the baseline step does the same work
    (device: where to draw them -- the eager-GPU baseline draws on the GPU, as nn.Dropout would)."""
    def n_patch(c):
        return (c["image_size"][0] // c["patch_size"]) * (c["image_size"][1] // c["patch_size"])

    def tower(c, N):
        D, T, C = c["hidden_dim"], c["token_dim"], c["channel_dim"]
        f = lambda *s: (torch.rand(*s, generator=generator, device=device) >= p).float()
        return [{"tok_h": f(B, D, T), "tok_o": f(B, D, N), "ch_h": f(B, N, C), "ch_o": f(B, N, D)}
                for _ in range(c["num_mixers"])]

    ni, na = n_patch(cfg["image"]), n_patch(cfg["audio"])
    return {"image": tower(cfg["image"], ni), "audio": tower(cfg["audio"], na),
            "fusion": tower(cfg["multimodal"], ni + na)}
