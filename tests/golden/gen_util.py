"""Seeded parameter / input generators shared by make_golden.py (which feeds the
reference), the oracle tests and the GPU parity tests.

Everything is generated from numpy's PCG64 (`default_rng(seed)`) in a fixed key
order, so the same tensors can be regenerated anywhere (no weight blobs in the
repo, nothing read from /root/reference at test time).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

# ---- model configs (values from the reference's cfg/*.yml) ----
AVMNIST = {
    # cfg/avmnist/avmnist_m2-mixer_S.yml:25-56
    "S": dict(dropout=0.1, num_classes=10,
              image=dict(in_channels=1, hidden_dim=32, patch_size=14, image_size=[28, 28], token_dim=16, channel_dim=256, num_mixers=2),
              audio=dict(in_channels=1, hidden_dim=32, patch_size=56, image_size=[112, 112], token_dim=16, channel_dim=256, num_mixers=2),
              multimodal=dict(hidden_dim=32, token_dim=16, channel_dim=256, num_mixers=1)),
    # cfg/avmnist/avmnist_m2-mixer_M.yml:25-59
    "M": dict(dropout=0.1, num_classes=10,
              image=dict(in_channels=1, hidden_dim=64, patch_size=14, image_size=[28, 28], token_dim=16, channel_dim=1024, num_mixers=2),
              audio=dict(in_channels=1, hidden_dim=64, patch_size=56, image_size=[112, 112], token_dim=16, channel_dim=1024, num_mixers=2),
              multimodal=dict(hidden_dim=64, token_dim=16, channel_dim=1024, num_mixers=1)),
    # cfg/avmnist/avmnist_m2-mixer_B.yml:24-56
    "B": dict(dropout=0.5, num_classes=10,
              image=dict(in_channels=1, hidden_dim=128, patch_size=14, image_size=[28, 28], token_dim=32, channel_dim=3072, num_mixers=4),
              audio=dict(in_channels=1, hidden_dim=128, patch_size=56, image_size=[112, 112], token_dim=32, channel_dim=3072, num_mixers=4),
              multimodal=dict(hidden_dim=128, token_dim=32, channel_dim=3078, num_mixers=2)),
}

# cfg/mimic/mimic_m2-mixer_H.yml:20-52
MIMIC_H = dict(dropout=0.3, num_classes=6,
               time=dict(embedding_dim=12, proj_dim=64, hidden_dim=64, num_patch=24, token_dim=16, channel_dim=64, num_mixers=1),
               static=dict(input_dim=5, hidden_dim=64, num_blocks=2, output_dim=64),
               multimodal=dict(hidden_dim=64, token_dim=8, channel_dim=64, num_mixers=1))

# cfg/mmimdb/mmimdb_3loss.yml:36-78
MMIMDB = dict(dropout=0.5, num_classes=23,
              image=dict(in_channels=3, hidden_dim=256, patch_size=32, image_size=[160, 256], token_dim=16, channel_dim=512, num_mixers=2),
              text=dict(in_channels=1, hidden_dim=256, patch_size=32, image_size=[160, 256], token_dim=16, channel_dim=512, num_mixers=2),
              multimodal=dict(hidden_dim=256, token_dim=16, channel_dim=512, num_mixers=2),
              pos_weight=[4.57642832, 7.38544978, 10.79846869, 13.23391421, 15.59020924, 18.62735849,
                          22.48861048, 25.21711367, 74.50943396, 31.31641554, 31.79549114, 32.90833333,
                          39.64859438, 56.90201729, 40.46106557, 58.24483776, 67.3890785, 84.92473118,
                          58.33087149, 62.68253968, 114.13294798, 141.54121864, 116.83431953])


def num_patch(c: dict) -> int:
    return (c["image_size"][0] // c["patch_size"]) * (c["image_size"][1] // c["patch_size"])


# ---- state-dict shapes ----
def block_shapes(prefix: str, D: int, N: int, T: int, C: int) -> "OrderedDict[str, tuple]":
    s = OrderedDict()
    s[prefix + "token_mix.0.weight"] = (D,)
    s[prefix + "token_mix.0.bias"] = (D,)
    s[prefix + "token_mix.2.net.0.weight"] = (T, N)
    s[prefix + "token_mix.2.net.0.bias"] = (T,)
    s[prefix + "token_mix.2.net.3.weight"] = (N, T)
    s[prefix + "token_mix.2.net.3.bias"] = (N,)
    s[prefix + "channel_mix.0.weight"] = (D,)
    s[prefix + "channel_mix.0.bias"] = (D,)
    s[prefix + "channel_mix.1.net.0.weight"] = (C, D)
    s[prefix + "channel_mix.1.net.0.bias"] = (C,)
    s[prefix + "channel_mix.1.net.3.weight"] = (D, C)
    s[prefix + "channel_mix.1.net.3.bias"] = (D,)
    return s


def tower_shapes(prefix: str, c: dict, N: int, kind: str) -> "OrderedDict[str, tuple]":
    s = OrderedDict()
    D = c["hidden_dim"]
    if kind == "patch":
        s[prefix + "to_patch_embedding.0.weight"] = (D, c["in_channels"], c["patch_size"], c["patch_size"])
        s[prefix + "to_patch_embedding.0.bias"] = (D,)
    elif kind == "proj":
        s[prefix + "proj.weight"] = (c["proj_dim"], c["embedding_dim"])
        s[prefix + "proj.bias"] = (c["proj_dim"],)
    for i in range(c["num_mixers"]):
        s.update(block_shapes(f"{prefix}mixer_blocks.{i}.", D, N, c["token_dim"], c["channel_dim"]))
    s[prefix + "layer_norm.weight"] = (D,)
    s[prefix + "layer_norm.bias"] = (D,)
    return s


def avmnist_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    """Creation order of models/avmnist.py:181-191."""
    s = OrderedDict()
    ni, na = num_patch(cfg["image"]), num_patch(cfg["audio"])
    s.update(tower_shapes("image_mixer.", cfg["image"], ni, "patch"))
    s.update(tower_shapes("audio_mixer.", cfg["audio"], na, "patch"))
    s.update(tower_shapes("fusion_mixer.", cfg["multimodal"], ni + na, "none"))
    K = cfg["num_classes"]
    s["classifier_image.weight"] = (K, cfg["image"]["hidden_dim"])
    s["classifier_image.bias"] = (K,)
    s["classifier_audio.weight"] = (K, cfg["audio"]["hidden_dim"])
    s["classifier_audio.bias"] = (K,)
    s["classifier_fusion.classifer.weight"] = (K, cfg["multimodal"]["hidden_dim"])
    s["classifier_fusion.classifer.bias"] = (K,)
    return s


def mimic_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    """Creation order of models/mimic.py:39-49."""
    s = OrderedDict()
    t, st = cfg["time"], cfg["static"]
    s.update(tower_shapes("time_mixer.", t, t["num_patch"], "proj"))
    for i in range(st["num_blocks"]):
        s[f"static_extractor.module_list.{3 * i}.weight"] = (st["hidden_dim"], st["input_dim"] if i == 0 else st["hidden_dim"])
        s[f"static_extractor.module_list.{3 * i}.bias"] = (st["hidden_dim"],)
    k = 3 * st["num_blocks"]
    s[f"static_extractor.module_list.{k}.weight"] = (st["output_dim"], st["hidden_dim"])
    s[f"static_extractor.module_list.{k}.bias"] = (st["output_dim"],)
    s.update(tower_shapes("fusion_mixer.", cfg["multimodal"], 1 + t["num_patch"], "none"))
    K = cfg["num_classes"]
    s["classifier_static.weight"] = (K, st["output_dim"]); s["classifier_static.bias"] = (K,)
    s["classifier_time.weight"] = (K, t["hidden_dim"]); s["classifier_time.bias"] = (K,)
    s["classifier_fusion.classifer.weight"] = (K, cfg["multimodal"]["hidden_dim"])
    s["classifier_fusion.classifer.bias"] = (K,)
    return s


def mmimdb_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    """Creation order of models/mmimdb.py:35-45."""
    s = OrderedDict()
    ni, nt = num_patch(cfg["image"]), num_patch(cfg["text"])
    s.update(tower_shapes("image_mixer.", cfg["image"], ni, "patch"))
    s.update(tower_shapes("text_mixer.", cfg["text"], nt, "patch"))
    s.update(tower_shapes("fusion_mixer.", cfg["multimodal"], ni + nt, "none"))
    K = cfg["num_classes"]
    s["classifier_image.weight"] = (K, cfg["image"]["hidden_dim"]); s["classifier_image.bias"] = (K,)
    s["classifier_text.weight"] = (K, cfg["text"]["hidden_dim"]); s["classifier_text.bias"] = (K,)
    s["classifier_fusion.classifer.weight"] = (K, cfg["multimodal"]["hidden_dim"])
    s["classifier_fusion.classifer.bias"] = (K,)
    return s


def _is_ln(key: str) -> bool:
    return (key.endswith("token_mix.0.weight") or key.endswith("token_mix.0.bias")
            or key.endswith("channel_mix.0.weight") or key.endswith("channel_mix.0.bias")
            or "layer_norm." in key)


def make_params(shapes: "OrderedDict[str, tuple]", seed: int, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """Linear/Conv: U(+-1/sqrt(fan_in)) (torch's default bound); LayerNorm gamma
    1 + 0.1 N(0,1), beta 0.1 N(0,1) so that affine terms are exercised."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for k, shp in shapes.items():
        if _is_ln(k):
            a = rng.standard_normal(shp) * 0.1 + (1.0 if k.endswith("weight") else 0.0)
        else:
            if k.endswith("bias"):
                wk = k[:-4] + "weight"
                fan_in = int(np.prod(shapes[wk][1:]))
            else:
                fan_in = int(np.prod(shp[1:]))
            bound = 1.0 / math.sqrt(fan_in)
            a = rng.uniform(-bound, bound, size=shp)
        out[k] = torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    return out


def avmnist_batch(B: int, seed: int, cfg: dict):
    """image ~ U[0,1) (B,1,28,28), audio ~ U[0,1) (B,1,112,112), label ~ randint(0,10).
    Shapes per datasets/avmnist.py:113-114."""
    rng = np.random.default_rng(seed)
    ih, iw = cfg["image"]["image_size"]
    ah, aw = cfg["audio"]["image_size"]
    image = torch.from_numpy(rng.random((B, cfg["image"]["in_channels"], ih, iw), dtype=np.float32))
    audio = torch.from_numpy(rng.random((B, cfg["audio"]["in_channels"], ah, aw), dtype=np.float32))
    label = torch.from_numpy(rng.integers(0, cfg["num_classes"], size=(B,), dtype=np.int64))
    return image, audio, label


def mimic_batch(B: int, seed: int, cfg: dict):
    rng = np.random.default_rng(seed)
    static = torch.from_numpy(rng.standard_normal((B, cfg["static"]["input_dim"])).astype(np.float32))
    time = torch.from_numpy(rng.standard_normal((B, cfg["time"]["num_patch"], cfg["time"]["embedding_dim"])).astype(np.float32))
    label = torch.from_numpy(rng.integers(0, cfg["num_classes"], size=(B,), dtype=np.int64))
    return static, time, label


def mmimdb_batch(B: int, seed: int, cfg: dict):
    rng = np.random.default_rng(seed)
    ih, iw = cfg["image"]["image_size"]
    image = torch.from_numpy(rng.standard_normal((B, cfg["image"]["in_channels"], ih, iw)).astype(np.float32))
    text = torch.from_numpy(rng.standard_normal((B, cfg["text"]["in_channels"], ih, iw)).astype(np.float32))
    label = torch.from_numpy((rng.random((B, cfg["num_classes"])) < 0.1).astype(np.float32))
    return image, text, label


# single-MixerBlock shapes (N, D, T, C) exercised by the block-level fixtures (SURVEY.md section 8c i)
BLOCK_CASES = [
    (4, 128, 32, 3072), (8, 128, 32, 3078), (4, 32, 16, 256), (8, 32, 16, 256), (4, 64, 16, 1024),
    (24, 64, 16, 64), (25, 64, 8, 64), (40, 256, 16, 512), (80, 256, 16, 512),
]


def block_case_tensors(case, B: int, seed: int):
    N, D, T, C = case
    p = make_params(block_shapes("", D, N, T, C), seed)
    rng = np.random.default_rng(seed + 7919)
    x = torch.from_numpy(rng.standard_normal((B, N, D)).astype(np.float32))
    dy = torch.from_numpy(rng.standard_normal((B, N, D)).astype(np.float32))
    return p, x, dy


def block_masks(case, B: int, seed: int, p_drop: float):
    """Bernoulli keep-masks for the four dropout sites of one block."""
    N, D, T, C = case
    rng = np.random.default_rng(seed + 104729)
    f = lambda *s: torch.from_numpy((rng.random(s) >= p_drop).astype(np.float32))
    return {"tok_h": f(B, D, T), "tok_o": f(B, D, N), "ch_h": f(B, N, C), "ch_o": f(B, N, D)}
