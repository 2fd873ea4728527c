#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's `modules` package (mixer / fusion / classification / mlp) is
imported from /root/reference; `models/*.py` cannot be imported here
(pytorch_lightning, omegaconf, wandb, torchmetrics are absent), so the task
modules are composed exactly as models/avmnist.py:181-191,259-291,
models/mimic.py:39-49,98-121 and models/mmimdb.py:35-45,96-123 compose them.
Only numbers are written out (inputs are regenerable from seeds, see
gen_util.py); no reference source or bytecode is copied.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
REF = os.environ.get("M2M_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import gen_util as G  # noqa: E402
import modules as R   # noqa: E402  (the reference package)

torch.set_num_threads(8)
torch.manual_seed(0)


def digest(t: torch.Tensor, full_limit: int = 4096) -> dict:
    t = t.detach().to(torch.float64).flatten()
    d = {"l2": np.float64(t.norm().item()), "sum": np.float64(t.sum().item()), "numel": np.int64(t.numel())}
    if t.numel() <= full_limit:
        d["full"] = t.to(torch.float32).numpy()
    else:
        idx = np.linspace(0, t.numel() - 1, 257).astype(np.int64)
        d["sample"] = t[idx].to(torch.float32).numpy()
    return d


def put(store: dict, name: str, t: torch.Tensor, full_limit: int = 4096):
    for k, v in digest(t, full_limit).items():
        store[f"{name}//{k}"] = v


# ---------------------------------------------------------------- (i) single MixerBlock
def golden_blocks():
    out = {}
    for ci, case in enumerate(G.BLOCK_CASES):
        N, D, T, C = case
        B = 2
        p, x, dy = G.block_case_tensors(case, B, seed=1000 + ci)
        blk = R.MixerBlock(D, N, T, C, dropout=0.0)
        missing = blk.load_state_dict(p, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        blk.train()
        xr = x.clone().requires_grad_(True)
        y = blk(xr)
        (y * dy).sum().backward()
        tag = f"case{ci}"
        out[f"{tag}//shape"] = np.array([B, N, D, T, C], dtype=np.int64)
        put(out, f"{tag}//y", y, full_limit=1 << 17)
        put(out, f"{tag}//dx", xr.grad, full_limit=1 << 17)
        for k, prm in blk.named_parameters():
            put(out, f"{tag}//grad//{k}", prm.grad)
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **out)
    print("blocks.npz", len(out))


# ---------------------------------------------------------------- dropout placement / scaling
def golden_dropout():
    """Train-mode block with p=0.5: capture the masks nn.Dropout drew (via hooks) so the
    oracle can be fed the same masks; pins where dropout sits and its 1/(1-p) scaling."""
    out = {}
    case = (4, 32, 16, 256)
    N, D, T, C = case
    B, pdrop = 3, 0.5
    p, x, dy = G.block_case_tensors(case, B, seed=4242)
    blk = R.MixerBlock(D, N, T, C, dropout=pdrop)
    blk.load_state_dict(p, strict=True)
    blk.train()
    masks = {}

    def hook(name):
        def f(mod, inp, outp):
            masks[name] = (outp != 0).to(torch.float32) if (inp[0] != 0).all() else None
            assert masks[name] is not None
        return f

    blk.token_mix[2].net[2].register_forward_hook(hook("tok_h"))
    blk.token_mix[2].net[4].register_forward_hook(hook("tok_o"))
    blk.channel_mix[1].net[2].register_forward_hook(hook("ch_h"))
    blk.channel_mix[1].net[4].register_forward_hook(hook("ch_o"))
    torch.manual_seed(77)
    xr = x.clone().requires_grad_(True)
    y = blk(xr)
    (y * dy).sum().backward()
    out["shape"] = np.array([B, N, D, T, C], dtype=np.int64)
    out["p"] = np.float64(pdrop)
    for k, m in masks.items():
        out[f"mask//{k}"] = m.numpy().astype(np.uint8)
    put(out, "y", y, full_limit=1 << 17)
    put(out, "dx", xr.grad, full_limit=1 << 17)
    for k, prm in blk.named_parameters():
        put(out, f"grad//{k}", prm.grad)
    np.savez_compressed(os.path.join(HERE, "dropout_block.npz"), **out)
    print("dropout_block.npz", len(out))


# ---------------------------------------------------------------- (ii)/(iii) AV-MNIST full step
class RefAVMnist(torch.nn.Module):
    """Composition of models/avmnist.py:181-191 (ctor order) and :259-291 (forward)."""

    def __init__(self, cfg, dropout):
        super().__init__()
        img = dict(cfg["image"], block_type="MLPMixer")
        aud = dict(cfg["audio"], block_type="MLPMixer")
        mm = dict(cfg["multimodal"], block_type="FusionMixer", fusion_function="ConcatFusion")
        cls = dict(num_classes=cfg["num_classes"], classifier="StandardClassifier",
                   input_shape=[16, 49, cfg["multimodal"]["hidden_dim"]], hidden_dims=[1024, 512, 256, 32])
        self.image_mixer = R.get_block_by_name(**img, dropout=dropout)
        self.audio_mixer = R.get_block_by_name(**aud, dropout=dropout)
        self.fusion_function = R.get_fusion_by_name(**mm)
        npatch = self.fusion_function.get_output_shape(self.image_mixer.num_patch, self.audio_mixer.num_patch, dim=1)
        self.fusion_mixer = R.get_block_by_name(**mm, num_patches=npatch, dropout=dropout)
        self.classifier_image = torch.nn.Linear(cfg["image"]["hidden_dim"], cfg["num_classes"])
        self.classifier_audio = torch.nn.Linear(cfg["audio"]["hidden_dim"], cfg["num_classes"])
        self.classifier_fusion = R.get_classifier_by_name(**cls)
        self.crit = torch.nn.CrossEntropyLoss()
        self.fusion_loss_weight = 1.0 / 3

    def forward(self, image, audio, labels):
        il = self.image_mixer(image)
        al = self.audio_mixer(audio)
        fused = self.fusion_function(il, al)
        lg = self.fusion_mixer(fused)
        al_t, il_t = al, il
        al = al.reshape(al.shape[0], -1, al.shape[-1])
        il = il.reshape(il.shape[0], -1, il.shape[-1])
        il = self.classifier_image(il.mean(dim=1))
        al = self.classifier_audio(al.mean(dim=1))
        lg_t = lg
        lg = self.classifier_fusion(lg)
        li, la, lf = self.crit(il, labels), self.crit(al, labels), self.crit(lg, labels)
        ow = (1 - self.fusion_loss_weight) / 2
        loss = (self.fusion_loss_weight * lf + ow * li + ow * la) * 3
        return dict(image_tokens=il_t, audio_tokens=al_t, fusion_tokens=lg_t, image_logits=il, audio_logits=al,
                    logits=lg, loss_image=li, loss_audio=la, loss_fusion=lf, loss=loss)


def golden_avmnist(size: str, B: int, seed: int):
    cfg = G.AVMNIST[size]
    out = {}
    model = RefAVMnist(cfg, dropout=0.0)
    shapes = G.avmnist_shapes(cfg)
    sd = model.state_dict()
    assert list(sd.keys()) == list(shapes.keys()), "state-dict key order differs from gen_util"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(shapes[k]), (k, sd[k].shape, shapes[k])
    out["n_params"] = np.int64(sum(v.numel() for v in sd.values()))
    params = G.make_params(shapes, seed)
    model.load_state_dict(params, strict=True)
    model.train()
    image, audio, labels = G.avmnist_batch(B, seed + 1, cfg)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    for step in range(2):
        opt.zero_grad()
        r = model(image, audio, labels)
        r["loss"].backward()
        tag = f"step{step}"
        for k in ("image_logits", "audio_logits", "logits", "loss_image", "loss_audio", "loss_fusion", "loss"):
            put(out, f"{tag}//{k}", r[k])
        for k in ("image_tokens", "audio_tokens", "fusion_tokens"):
            put(out, f"{tag}//{k}", r[k], full_limit=8192)
        out[f"{tag}//preds"] = torch.softmax(r["logits"], 1).argmax(1).numpy()
        out[f"{tag}//preds_image"] = torch.softmax(r["image_logits"], 1).argmax(1).numpy()
        out[f"{tag}//preds_audio"] = torch.softmax(r["audio_logits"], 1).argmax(1).numpy()
        if step == 0:
            for k, prm in model.named_parameters():
                put(out, f"grad//{k}", prm.grad, full_limit=512)
        opt.step()
    for k, prm in model.named_parameters():
        put(out, f"after2//{k}", prm.data, full_limit=512)
    np.savez_compressed(os.path.join(HERE, f"avmnist_{size}.npz"), **out)
    print(f"avmnist_{size}.npz", len(out), "params", int(out["n_params"]))


# ---------------------------------------------------------------- (iv) MIMIC-H and MM-IMDb
def golden_mimic():
    cfg = G.MIMIC_H
    out = {}
    t = dict(cfg["time"], block_type="MLPMixerNoPatching", in_channels=1)
    st = dict(cfg["static"], block_type="MLP", in_channels=1)
    mm = dict(cfg["multimodal"], block_type="FusionMixer", fusion_function="ConcatFusion")
    m = torch.nn.Module()
    m.time_mixer = R.get_block_by_name(**t, dropout=0.0)
    m.static_extractor = R.get_block_by_name(**st, dropout=0.0)
    fusion = R.get_fusion_by_name(**mm)
    npatch = fusion.get_output_shape(1, m.time_mixer.num_patch, dim=1)
    m.fusion_mixer = R.get_block_by_name(**mm, num_patches=npatch, dropout=0.0)
    m.classifier_static = torch.nn.Linear(cfg["static"]["output_dim"], cfg["num_classes"])
    m.classifier_time = torch.nn.Linear(cfg["time"]["hidden_dim"], cfg["num_classes"])
    m.classifier_fusion = R.get_classifier_by_name(num_classes=cfg["num_classes"], classifier="StandardClassifier",
                                                   input_shape=[16, 1024, 64])
    shapes = G.mimic_shapes(cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys()), (list(sd.keys()), list(shapes.keys()))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(shapes[k]), k
    out["n_params"] = np.int64(sum(v.numel() for v in sd.values()))
    m.load_state_dict(G.make_params(shapes, 31), strict=True)
    m.train()
    static, time, labels = G.mimic_batch(6, 32, cfg)
    sl = m.static_extractor(static)
    tm = m.time_mixer(time)
    fused = fusion(sl.unsqueeze(1), tm)
    lg = m.fusion_mixer(fused)
    static_logits = m.classifier_static(sl)
    time_logits = m.classifier_time(tm.mean(1))
    logits = m.classifier_fusion(lg)
    ce = torch.nn.CrossEntropyLoss()
    lf, ls, lt = ce(logits, labels), ce(static_logits, labels), ce(time_logits, labels)
    w = 1.0 / 3
    ow = (1 - w) / 2
    loss = w * lf + ow * ls + ow * lt
    loss.backward()
    for k, v in dict(logits=logits, logits_static=static_logits, logits_time=time_logits, loss=loss,
                     loss_fusion=lf, loss_static=ls, loss_time=lt, time_tokens=tm, fusion_tokens=lg).items():
        put(out, k, v, full_limit=16384)
    for k, prm in m.named_parameters():
        put(out, f"grad//{k}", prm.grad, full_limit=512)
    np.savez_compressed(os.path.join(HERE, "mimic_H.npz"), **out)
    print("mimic_H.npz", len(out), "params", int(out["n_params"]))


def golden_mmimdb():
    cfg = G.MMIMDB
    out = {}
    img = dict(cfg["image"], block_type="MLPMixer")
    txt = dict(cfg["text"], block_type="MLPMixer")
    mm = dict(cfg["multimodal"], block_type="FusionMixer", fusion_function="ConcatFusion",
              num_modality=1, proj_modality_dim=16, modality_dim=64)
    m = torch.nn.Module()
    m.image_mixer = R.get_block_by_name(**img, dropout=0.0)
    m.text_mixer = R.get_block_by_name(**txt, dropout=0.0)
    fusion = R.get_fusion_by_name(**mm)
    npatch = fusion.get_output_shape(m.image_mixer.num_patch, m.text_mixer.num_patch, dim=1)
    m.fusion_mixer = R.get_block_by_name(**mm, num_patches=npatch, dropout=0.0)
    m.classifier_image = torch.nn.Linear(cfg["image"]["hidden_dim"], cfg["num_classes"])
    m.classifier_text = torch.nn.Linear(cfg["text"]["hidden_dim"], cfg["num_classes"])
    m.classifier_fusion = R.get_classifier_by_name(num_classes=cfg["num_classes"], classifier="StandardClassifier",
                                                   input_shape=[16, 49, 256], hidden_dims=[1024, 512, 256, 32])
    shapes = G.mmimdb_shapes(cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(shapes[k]), k
    out["n_params"] = np.int64(sum(v.numel() for v in sd.values()))
    m.load_state_dict(G.make_params(shapes, 51), strict=True)
    m.train()
    image, text, labels = G.mmimdb_batch(3, 52, cfg)
    il = m.image_mixer(image)
    tl = m.text_mixer(text)
    lg = m.fusion_mixer(fusion(il, tl))
    il = m.classifier_image(il.reshape(il.shape[0], -1, il.shape[-1]).mean(dim=1))
    tl = m.classifier_text(tl.reshape(tl.shape[0], -1, tl.shape[-1]).mean(dim=1))
    lg = m.classifier_fusion(lg)
    crit = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor(cfg["pos_weight"]))
    li, lt, lf = crit(il, labels.float()), crit(tl, labels.float()), crit(lg, labels.float())
    loss = li + lt + lf
    loss.backward()
    for k, v in dict(image_logits=il, text_logits=tl, logits=lg, loss=loss, loss_image=li, loss_text=lt,
                     loss_fusion=lf).items():
        put(out, k, v)
    out["preds"] = (torch.sigmoid(lg) > 0.5).long().numpy()
    for k, prm in m.named_parameters():
        put(out, f"grad//{k}", prm.grad, full_limit=256)
    np.savez_compressed(os.path.join(HERE, "mmimdb.npz"), **out)
    print("mmimdb.npz", len(out), "params", int(out["n_params"]))


# ---------------------------------------------------------------- (v) fusion shape algebra
def golden_fusion_shapes():
    """The six shape tests of tests/modules/test_fusion.py that pass on the reference,
    evaluated on the reference and stored as data."""
    out = {}
    f = R.ConcatFusion(useless_arg=1)
    a, b = torch.rand(10, 20, 30), torch.rand(10, 20, 30)
    out["concat//call"] = np.array(f(a, b).shape)
    out["concat//shape"] = np.array(f.get_output_shape(a.shape, b.shape))
    out["concat//dim1"] = np.int64(f.get_output_shape(20, 20, dim=1))
    out["concat//dim0"] = np.int64(f.get_output_shape(20, 20, dim=0))
    for name in ("SumFusion", "MaxFusion", "MeanFusion"):
        g = getattr(R, name)(useless_arg=1)
        out[f"{name}//call"] = np.array(g(a, b).shape)
        out[f"{name}//shape"] = np.array(g.get_output_shape(a.shape, b.shape))
        out[f"{name}//dim1"] = np.int64(g.get_output_shape(20, 20, dim=1))
    d = R.ConcatDynaFusion(useless_arg=1)
    a4, b4 = torch.rand(10, 20, 20, 30), torch.rand(10, 20, 20, 30)
    out["dyna//call"] = np.array(d(a4, b4).shape)
    out["dyna//shape"] = np.array(d.get_output_shape(a4.shape, b4.shape))
    out["dyna//dim1"] = np.int64(d.get_output_shape(36, 36, dim=1))
    np.savez_compressed(os.path.join(HERE, "fusion_shapes.npz"), **out)
    print("fusion_shapes.npz", len(out))


if __name__ == "__main__":
    golden_blocks()
    golden_dropout()
    golden_avmnist("S", 8, 11)
    golden_avmnist("M", 4, 21)
    golden_avmnist("B", 8, 12)
    golden_mimic()
    golden_mmimdb()
    golden_fusion_shapes()
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz"))
    print("total fixture bytes", tot)
