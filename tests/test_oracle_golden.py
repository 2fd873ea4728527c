"""Pin the CPU oracle (oracle/m2mixer_oracle.py) against the golden vectors that
tests/golden/make_golden.py captured from the reference's own modules."""
import numpy as np
import pytest
import torch

import gen_util as G
from golden_util import check, load
from oracle import m2mixer_oracle as O

torch.set_num_threads(8)
ATOL = 2e-5   # fp32 CPU vs fp32 CPU, different op order only
RTOL = 2e-5


@pytest.mark.parametrize("ci", range(len(G.BLOCK_CASES)))
def test_block_forward_backward(ci):
    gold = load("blocks.npz")
    case = G.BLOCK_CASES[ci]
    B, N, D, T, C = [int(v) for v in gold[f"case{ci}//shape"]]
    assert (N, D, T, C) == case
    p, x, dy = G.block_case_tensors(case, B, seed=1000 + ci)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    y = O.mixer_block(xr, leaves)
    (y * dy).sum().backward()
    check(gold, f"case{ci}//y", y, ATOL, RTOL)
    check(gold, f"case{ci}//dx", xr.grad, 5 * ATOL, 5 * RTOL)
    for k, leaf in leaves.items():
        check(gold, f"case{ci}//grad//{k}", leaf.grad, 2e-4, 2e-4)


def test_block_dropout_masks():
    gold = load("dropout_block.npz")
    B, N, D, T, C = [int(v) for v in gold["shape"]]
    pd = float(gold["p"])
    p, x, dy = G.block_case_tensors((N, D, T, C), B, seed=4242)
    masks = {k: torch.from_numpy(gold[f"mask//{k}"].astype(np.float32)) for k in ("tok_h", "tok_o", "ch_h", "ch_o")}
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    y = O.mixer_block(xr, leaves, "", pd, masks)
    (y * dy).sum().backward()
    check(gold, "y", y, ATOL, RTOL)
    check(gold, "dx", xr.grad, 5 * ATOL, 5 * RTOL)
    for k, leaf in leaves.items():
        check(gold, f"grad//{k}", leaf.grad, 2e-4, 2e-4)


@pytest.mark.parametrize("size,B,seed", [("S", 8, 11), ("M", 4, 21), ("B", 8, 12)])
def test_avmnist_two_steps(size, B, seed):
    gold = load(f"avmnist_{size}.npz")
    cfg = G.AVMNIST[size]
    shapes = G.avmnist_shapes(cfg)
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(gold["n_params"])
    params = dict(G.make_params(shapes, seed))
    image, audio, labels = G.avmnist_batch(B, seed + 1, cfg)
    state = {}
    for step in range(2):
        r = O.avmnist_train_step(image, audio, labels, params, cfg, state, lr=1e-2)
        tag = f"step{step}"
        tol = 3e-5 if step == 0 else 2e-3     # step 1 goes through an Adam update with lr 1e-2
        for k in ("image_logits", "audio_logits", "logits", "loss_image", "loss_audio", "loss_fusion", "loss",
                  "image_tokens", "audio_tokens", "fusion_tokens"):
            check(gold, f"{tag}//{k}", r[k], tol, tol)
        if step == 0:
            for k in ("preds", "preds_image", "preds_audio"):
                assert np.array_equal(r[k].numpy(), gold[f"{tag}//{k}"])
            for k, g in r["grads"].items():
                check(gold, f"grad//{k}", g, 2e-5, 2e-4)
    for k, v in params.items():
        # The token-mix output bias adds the same constant to every channel of a token, which every
        # later LayerNorm removes: its true gradient is exactly 0, the computed one is rounding noise
        # (~1e-9) and Adam turns that noise into +-lr steps.  Ill-conditioned in the reference itself,
        # so it is excluded; every other parameter must match after two Adam steps.
        if k.endswith("token_mix.2.net.3.bias"):
            continue
        check(gold, f"after2//{k}", v, 2.5e-3, 0.0)


def test_mimic_forward_backward():
    gold = load("mimic_H.npz")
    cfg = G.MIMIC_H
    shapes = G.mimic_shapes(cfg)
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(gold["n_params"])
    leaves = {k: v.requires_grad_(True) for k, v in G.make_params(shapes, 31).items()}
    static, time, labels = G.mimic_batch(6, 32, cfg)
    r = O.mimic_forward(static, time, labels, leaves, cfg)
    r["loss"].backward()
    for k in ("logits", "logits_static", "logits_time", "loss", "loss_fusion", "loss_static", "loss_time",
              "time_tokens", "fusion_tokens"):
        check(gold, k, r[k], ATOL, RTOL)
    for k, leaf in leaves.items():
        check(gold, f"grad//{k}", leaf.grad, 2e-5, 2e-4)


def test_mmimdb_forward_backward():
    gold = load("mmimdb.npz")
    cfg = G.MMIMDB
    shapes = G.mmimdb_shapes(cfg)
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(gold["n_params"])
    leaves = {k: v.requires_grad_(True) for k, v in G.make_params(shapes, 51).items()}
    image, text, labels = G.mmimdb_batch(3, 52, cfg)
    r = O.mmimdb_forward(image, text, labels, leaves, cfg, torch.tensor(cfg["pos_weight"]))
    r["loss"].backward()
    for k in ("image_logits", "text_logits", "logits", "loss", "loss_image", "loss_text", "loss_fusion"):
        check(gold, k, r[k], 5e-5, 5e-5)
    assert np.array_equal(r["preds"].numpy(), gold["preds"])
    for k, leaf in leaves.items():
        check(gold, f"grad//{k}", leaf.grad, 5e-5, 5e-4)
