"""GPU parity tests (-m gpu) of the exact configuration bench.py times, and of the callers either side of the path.

  * the benchmark's kernel instantiations -- bf16 operands, M2-Mixer-B, dropout ON (p = 0.5: one-bit masks; p = 0.1:
    16-bit draws), the two-tower grouped launches, and at batch 512 the column-split launches -- against the CPU oracle
    fed the very keep-masks the kernels regenerate (m2m_dropout_mask);
  * Adam's moments and the updated parameters against the oracle's optimizer state;
  * MM-IMDb / MIMIC-H at the batch sizes of their configs and at a large batch;
  * the epoch loop (every sample, ragged last batch) and the checkpoint round trip against the oracle.
"""
import os

import numpy as np
import pytest
import torch

import gen_util as G
from conftest import observe
from oracle import m2mixer_oracle as O

pytestmark = pytest.mark.gpu

FP32_ATOL = 1e-3       # BASELINE.json north_star: logits within 1e-3 in fp32
BF16_LOGITS = 2e-2     # bf16 operands, fp32 accumulate: absolute on O(1) logits (<~ 2e-2; observed maxima: test summary)
BF16_GRAD_REL = 4e-2   # relative to the gradient tensor's max (observed maxima are printed in the test summary)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from m2_mixer_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)


def grads_cleared(eng):
    """After an optimizer step every gradient element Adam is responsible for clearing is zero: everything outside the ranges
    the engine leaves to the next backward's overwriting weight-gradient launch (engine._setup_wgrad)."""
    g = eng.flat_g.detach().clone()
    for lo, n, _, keep in eng._ranges_add:
        if keep:
            g[lo:lo + n] = 0
    return float(g.abs().max()) == 0.0


def abserr(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def engine_masks(eng, B):
    """The keep-masks of every dropout site of the three towers at the engine's current step, in the oracle's layout."""
    step, seed = int(eng.drop_step[0]), eng.seed
    out = {}
    for name, rt in ((eng.MODS[0], eng.t_a), (eng.MODS[1], eng.t_b), ("fusion", eng.t_fus)):
        blocks = []
        for b in range(rt.nblocks):
            m = {"tok_h": rt.dropout_mask(b, 0, B, seed, step).view(B, rt.D, rt.T),
                 "tok_o": rt.dropout_mask(b, 1, B, seed, step).view(B, rt.D, rt.N),
                 "ch_h": rt.dropout_mask(b, 2, B, seed, step).view(B, rt.N, rt.Cp)[:, :, :rt.C],
                 "ch_o": rt.dropout_mask(b, 3, B, seed, step).view(B, rt.N, rt.D)}
            blocks.append({k: v.float().cpu() for k, v in m.items()})
        out[name] = blocks
    return out


def p_effective(p):
    """The keep probability is quantised to 16 bits; the kernels scale by 1 / keep_q."""
    return 1 - round((1 - p) * 65536) / 65536


def assert_preds_match(pred, logits_gpu, logits_ref, tol):
    """The kernel's predictions are the argmax of its own logits, and equal the oracle's wherever the oracle's top-2
    margin exceeds what the bf16 tolerance could flip (random-init logits sit close together: ties within 2 x tol are
    not a class decision either implementation can be held to)."""
    assert torch.equal(pred.cpu().long(), logits_gpu.float().cpu().argmax(1))
    top2 = logits_ref.topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 2 * tol
    assert torch.equal(pred.cpu().long()[decided], logits_ref.argmax(1)[decided])


@pytest.mark.parametrize("p_drop,B", [(0.5, 40), (0.5, 13), (0.1, 40), (0.1, 13), (0.5, 512)])
def test_bench_instantiation_with_dropout_vs_oracle(p_drop, B, dev):
    """bf16, M2-Mixer-B, dropout on, through the launches the benchmark's graph holds (B = 512: exactly bench.py's
    configuration; B = 40 / 13: the same templates with ragged tiles).  The oracle gets the masks the kernels draw."""
    from m2_mixer_amd.engine import AVMnistEngine
    cfg = dict(G.AVMNIST["B"], dropout=p_drop)
    eng = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-2, init=False)
    shapes = G.avmnist_shapes(cfg)
    params = dict(G.make_params(shapes, 17))
    eng.load_state_dict(params)
    image, audio, labels = G.avmnist_batch(B, 18, cfg)
    eng.forward_backward(image.to(dev), audio.to(dev), labels.to(dev))
    torch.cuda.synchronize()
    masks = engine_masks(eng, B)
    keep = float(np.mean([m["ch_h"].mean() for m in masks["image"]]))
    assert abs(keep - (1 - p_drop)) < 0.02, keep
    ref = O.avmnist_train_step(image, audio, labels, dict(params), cfg, {}, lr=1e-2, drop_p=p_effective(p_drop), masks=masks)
    for i, k in enumerate(("image_logits", "audio_logits", "logits")):
        assert observe("bf16 logits (abs)", abserr(eng.logits[i], ref[k]), BF16_LOGITS) < BF16_LOGITS, k
        assert_preds_match(eng.preds[i], eng.logits[i], ref[k], BF16_LOGITS)
    for i, k in enumerate(("loss_image", "loss_audio", "loss_fusion", "loss")):
        assert abs(float(eng.losses[i]) - float(ref[k])) < 2e-2, k
    for k, g in ref["grads"].items():
        if k.endswith("token_mix.2.net.3.bias"):       # exactly-zero true gradient
            continue
        assert observe("bf16 gradients (rel to max)", relerr(eng.grads[k], g), BF16_GRAD_REL) < BF16_GRAD_REL, k


@pytest.mark.parametrize("p_drop,B", [(0.5, 200), (0.1, 77), (0.0, 130)])
def test_split_path_ragged_batches_vs_oracle(p_drop, B, dev, monkeypatch):
    """The column-split launches (csrc/split.h) forced on at batches that do not fill their tiles: B * N = 800 / 308 / 520
    token rows against 128-row chain workgroups and 16-row mix workgroups (ragged last tiles everywhere, an odd number of
    16-row tiles, a single sample group in the last mix workgroup), fusion tower with 97 column units over 8 splits.
    Same oracle comparison as the benchmark instantiation; and the fused one-launch path must agree with it closely."""
    from m2_mixer_amd.engine import AVMnistEngine
    cfg = dict(G.AVMNIST["B"], dropout=p_drop)
    shapes = G.avmnist_shapes(cfg)
    params = dict(G.make_params(shapes, 27))
    image, audio, labels = G.avmnist_batch(B, 28, cfg)
    gb = (image.to(dev), audio.to(dev), labels.to(dev))
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("M2M_SPLIT", mode)
        eng = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-2, init=False)
        eng.load_state_dict(params)
        eng.forward_backward(*gb)
        torch.cuda.synchronize()
        got[mode] = (eng.logits.clone(), eng.losses.clone(), eng.flat_g.clone())
        if mode == "1":
            split_eng = eng
    eng = split_eng
    masks = engine_masks(eng, B) if p_drop > 0 else None
    ref = O.avmnist_train_step(image, audio, labels, dict(params), cfg, {}, lr=1e-2, drop_p=p_effective(p_drop), masks=masks)
    logits, losses, flat_g = got["1"]
    for i, k in enumerate(("image_logits", "audio_logits", "logits")):
        assert observe("bf16 logits (abs)", abserr(logits[i], ref[k]), BF16_LOGITS) < BF16_LOGITS, k
    for i, k in enumerate(("loss_image", "loss_audio", "loss_fusion", "loss")):
        assert abs(float(losses[i]) - float(ref[k])) < 2e-2, k
    for k, g in ref["grads"].items():
        if k.endswith("token_mix.2.net.3.bias"):
            continue
        gv = eng.grads[k]
        o = (gv.data_ptr() - eng.flat_g.data_ptr()) // 4
        assert observe("bf16 gradients (rel to max)", relerr(flat_g[o:o + gv.numel()].view_as(gv), g), BF16_GRAD_REL) < BF16_GRAD_REL, k
    # split vs fused: same masks, same bf16 rounding points; they differ by the GELU table form and fp32 summation order
    assert abserr(got["1"][0], got["0"][0]) < 1e-2 and abserr(got["1"][1], got["0"][1]) < 1e-3
    # evaluation (dropout off, no saved activations: the carry stream is rewritten in place)
    monkeypatch.setenv("M2M_SPLIT", "1")
    out = eng.evaluate(*gb)
    torch.cuda.synchronize()
    ev = O.avmnist_forward(image, audio, labels, params, cfg)
    for k in ("logits", "image_logits", "audio_logits"):
        assert observe("bf16 logits (abs)", abserr(out[k], ev[k]), BF16_LOGITS) < BF16_LOGITS, k


@pytest.mark.parametrize("fused_update", ["0", "1"])
@pytest.mark.parametrize("size,B,seed", [("S", 8, 11), ("M", 4, 21), ("B", 8, 12)])
def test_adam_moments_and_parameters_vs_oracle(size, B, seed, fused_update, dev, monkeypatch):
    """Two optimisation steps, fp32 mode: Adam's first / second moments (linear / quadratic in the gradient, hence
    well-conditioned) against the oracle's optimizer state, and the parameters wherever the gradient is not ~0 (Adam's
    first step moves a parameter by lr * g / (|g| + eps): at |g| ~ eps that is noise, and at |g| < 5e-3 max|g| the fp32
    summation order of the kernels' partial sums already moves the update by more than the tolerance).  A no-op or mis-scaled update
    moves the parameters by up to lr = 1e-2 from the oracle's: the tolerance is 3 % of that (one tensor, the fusion
    tower's last channel_mix weight, sits at 1.9-2.3 % whatever the summation order of the kernels' partial sums).  Both forms of the update:
    the flat Adam launch followed by the re-pack, and the one-launch m2m_adam_pack_all (M2M_FUSED_UPDATE=1)."""
    from m2_mixer_amd.engine import AVMnistEngine
    monkeypatch.setenv("M2M_FUSED_UPDATE", fused_update)
    cfg = dict(G.AVMNIST[size], dropout=0.0)
    eng = AVMnistEngine(cfg, B, device=dev, precision="fp32", lr=1e-2, init=False)
    shapes = G.avmnist_shapes(cfg)
    params = dict(G.make_params(shapes, seed))
    eng.load_state_dict(params)
    image, audio, labels = G.avmnist_batch(B, seed + 1, cfg)
    gb = (image.to(dev), audio.to(dev), labels.to(dev))
    state, significant = {}, {k: torch.ones(s, dtype=torch.bool) for k, s in shapes.items()}
    for step in (1, 2):
        ref = O.avmnist_train_step(image, audio, labels, params, cfg, state, lr=1e-2)      # updates params / state in place
        eng.train_step(*gb)
        torch.cuda.synchronize()
        assert float(eng.adam_state[0]) == step and grads_cleared(eng)
        for k in shapes:
            if k.endswith("token_mix.2.net.3.bias"):
                continue
            g = ref["grads"][k]
            gmax = float(g.abs().max())
            assert abserr(eng.exp_avg[k], state["m"][k]) < 1e-3 * max(gmax, 1e-6), (step, k)
            assert abserr(eng.exp_avg_sq[k], state["v"][k]) < 2e-3 * max(gmax * gmax, 1e-12), (step, k)
            significant[k] &= g.abs() > max(1e-6, 5e-3 * gmax)
            sel = significant[k]
            if bool(sel.any()):
                err = (eng.params[k].cpu() - params[k])[sel].abs().max()
                assert float(err) < 3e-4, (step, k, float(err))
    assert sum(int(v.sum()) for v in significant.values()) > 0.3 * eng.n_params


@pytest.mark.parametrize("rowtiles", ["1", "0"])
def test_one_launch_update_is_bit_identical_to_adam_then_repack_bf16(rowtiles, dev, monkeypatch):
    """M2-Mixer-B, bf16: two training steps with the one-launch Adam + re-pack (m2m_adam_pack_all; M2M_AP_ROWTILES=1: W2 in 8-row x
    512-column tiles, the fusion tower's ragged last chunk included; 0: the 32-column-group tiles) against the flat Adam followed by
    m2m_pack_all: same arithmetic per element, so parameters, both moments, the cleared gradient and every packed operand copy the
    chain kernels read must agree BIT FOR BIT."""
    from m2_mixer_amd.engine import AVMnistEngine
    cfg, B = dict(G.AVMNIST["B"]), 16
    batch = tuple(t.to(dev) for t in G.avmnist_batch(B, 5, cfg))
    monkeypatch.setenv("M2M_AP_ROWTILES", rowtiles)
    engs = []
    for fused in ("0", "1"):
        monkeypatch.setenv("M2M_FUSED_UPDATE", fused)
        e = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-2, seed=3)
        if engs:
            e.load_state_dict(engs[0].state_dict())
            e.pack()
        engs.append(e)
    sep, fus = engs
    assert fus._fused_update and not sep._fused_update
    for _ in range(2):
        for e in engs:
            e.train_step(*batch)
    torch.cuda.synchronize()
    assert torch.equal(sep.flat_p, fus.flat_p) and torch.equal(sep.flat_m, fus.flat_m) and torch.equal(sep.flat_v, fus.flat_v)
    assert torch.equal(sep.flat_g, fus.flat_g)
    for ts, tf in zip((sep.t_a, sep.t_b, sep.t_fus), (fus.t_a, fus.t_b, fus.t_fus)):
        for i in range(ts.nblocks):
            for k, v in ts._keep[f"packed{i}"].items():
                if k == "w1tc" and ts.pack_all_skips_w1tc():
                    continue
                assert torch.equal(v, tf._keep[f"packed{i}"][k]), (i, k)
    for es, ef in zip((sep.e_a, sep.e_b), (fus.e_a, fus.e_b)):
        assert torch.equal(es._keep["wn"], ef._keep["wn"])


def test_update_from_a_bf16_gradient_copy_agrees_in_all_three_forms(dev, monkeypatch):
    """optimizer_step(scale, grad_bf16) -- the update after a bf16-compressed exchange (`--grad-compress bf16`): Adam takes the
    gradient VALUES from a bf16 copy of the flat gradient.  The flat Adam + re-pack, the one-launch form in row tiles and the
    one-launch form in column-group tiles must leave bit-identical parameters, moments and packed copies, and the result must be
    the fp32 update to within bf16 rounding of the gradient."""
    from m2_mixer_amd.engine import AVMnistEngine
    cfg, B = dict(G.AVMNIST["B"]), 16
    batch = tuple(t.to(dev) for t in G.avmnist_batch(B, 5, cfg))
    engs = []
    for fused, rowtiles in (("0", "1"), ("1", "1"), ("1", "0")):
        monkeypatch.setenv("M2M_FUSED_UPDATE", fused)
        e = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-2, seed=3)
        if engs:
            e.load_state_dict(engs[0].state_dict())
            e.pack()
        engs.append(e)
    ref = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-2, seed=3)
    ref.load_state_dict(engs[0].state_dict())
    ref.pack()
    for e, rowtiles in zip(engs, ("1", "1", "0")):
        monkeypatch.setenv("M2M_AP_ROWTILES", rowtiles)          # (the library reads it at every one-launch update)
        e.forward_backward(*batch)
        e.optimizer_step(0.5, e.flat_g.to(torch.bfloat16))
    ref.forward_backward(*batch)
    ref.optimizer_step(0.5)
    torch.cuda.synchronize()
    a = engs[0]
    for e in engs[1:]:
        assert torch.equal(a.flat_p, e.flat_p) and torch.equal(a.flat_m, e.flat_m) and torch.equal(a.flat_v, e.flat_v)
        for ta, te in zip((a.t_a, a.t_b, a.t_fus), (e.t_a, e.t_b, e.t_fus)):
            for i in range(ta.nblocks):
                for k, v in ta._keep[f"packed{i}"].items():
                    if k == "w1tc" and ta.pack_all_skips_w1tc():
                        continue
                    assert torch.equal(v, te._keep[f"packed{i}"][k]), (i, k)
    # against the fp32-gradient update: first moment m = 0.1 g, so bf16 rounding of g (2^-9 relative) shows there directly
    assert relerr(a.flat_m, ref.flat_m) < 2.0 ** -8


def _module_path_net(cfg, B, dev):
    import m2_mixer_amd as M
    from m2_mixer_amd import models as MD
    M.set_precision("bf16")
    npatch = lambda c: (c["image_size"][0] // c["patch_size"]) * (c["image_size"][1] // c["patch_size"])
    mods = {"image": dict(cfg["image"], block_type="MLPMixer"), "audio": dict(cfg["audio"], block_type="MLPMixer"),
            "multimodal": dict(cfg["multimodal"], block_type="FusionMixer", fusion_function="ConcatFusion"),
            "classification": dict(classifier="StandardClassifier", num_classes=cfg["num_classes"],
                                   input_shape=[B, npatch(cfg["image"]) + npatch(cfg["audio"]), cfg["multimodal"]["hidden_dim"]])}
    torch.manual_seed(42)
    net = MD.AVMnistMixerMultiLoss({"dropout": cfg["dropout"], "modalities": mods}, {"lr": 1e-2, "betas": (0.9, 0.999), "scheduler_patience": 2}).to(dev)
    net.train()
    return net


def test_module_path_step_replayed_as_one_graph(dev):
    """m2_mixer_amd.graphs.GraphedStep: the import-swap path's training step (shared_step -> backward -> Adam) captured into ONE
    hipGraph.  (a) dropout off: three replayed steps on three batches leave the parameters of three eager steps (to 1e-3: a tenth
    of one step's lr), and constructing the GraphedStep did not train; (b) dropout 0.5, lr = 0: two replays on the SAME batch give different losses
    -- the dropout step counter lives on the device and advances inside the graph (a host integer would be baked in)."""
    from m2_mixer_amd import config
    from m2_mixer_amd.graphs import GraphedStep
    B = 8
    try:
        cfg = dict(G.AVMNIST["S"], dropout=0.0)
        batches = []
        for i in range(3):
            image, audio, labels = G.avmnist_batch(B, 30 + i, cfg)
            batches.append({"image": image.to(dev), "audio": audio.to(dev), "label": labels.to(dev)})
        eager = _module_path_net(cfg, B, dev)
        graphed = _module_path_net(cfg, B, dev)
        graphed.load_state_dict(eager.state_dict())
        before = [p.detach().clone() for p in graphed.parameters()]
        opt_e = eager.configure_optimizers()["optimizer"]
        opt_g = graphed.configure_optimizers()["optimizer"]
        for g in opt_e.param_groups:
            g["capturable"] = True
        gs = GraphedStep(graphed, opt_g, batches[0], fused=False)      # (the same optimizer implementation on both sides)
        assert all(torch.equal(a, p) for a, p in zip(before, graphed.parameters()))          # construction did not train
        for b in batches:
            opt_e.zero_grad(set_to_none=True)
            eager.shared_step(b, mode="train")["loss"].backward()
            opt_e.step()
            gs(b)
        torch.cuda.synchronize()
        # (not bit for bit: the graphed optimizer reads its learning rate from a device tensor -- another rounding of
        # lr / bias_correction than the float path -- and the module path's embedding gradients use float atomics; a
        # parameter moves by up to lr = 1e-2 per step, the two runs must agree to a small fraction of that)
        worst = 0.0
        for (k, a), bpar in zip(eager.named_parameters(), graphed.parameters()):
            worst = max(worst, float((a.detach() - bpar.detach()).abs().max()))
        observe("graphed vs eager module-path step, max parameter difference after 3 steps (abs)", worst, 1e-3)
        assert worst < 1e-3
        # (c) torch's FUSED Adam updates the parameters without advancing their version counters: the modules must re-pack their
        #     operand copies anyway (training mode), or the towers would keep computing with the weights of the first pack
        config.set_device_dropout_step(False)
        net = _module_path_net(cfg, B, dev)
        opt = net.configure_optimizers()["optimizer"]
        for g in opt.param_groups:
            g["fused"], g["foreach"] = True, False
        for b in batches[:2]:
            opt.zero_grad(set_to_none=True)
            net.shared_step(b, mode="train")["loss"].backward()
            opt.step()
        net.shared_step(batches[2], mode="train")                 # the forward after the last update: its packed copies ...
        torch.cuda.synchronize()
        towers = [m for m in net.modules() if getattr(m, "_rts", None)]
        assert towers
        for m in towers:
            for rt in m._rts:
                for i in range(rt.nblocks):
                    have = {k: v.clone() for k, v in rt._keep[f"packed{i}"].items()}
                    rt.pack(force=True)                           # ... must be those of the CURRENT weights
                    torch.cuda.synchronize()
                    for k, v in rt._keep[f"packed{i}"].items():
                        assert torch.equal(v, have[k]), (type(m).__name__, i, k)
        # (b) the dropout stream advances inside the graph
        cfg = dict(G.AVMNIST["S"], dropout=0.5)
        net = _module_path_net(cfg, B, dev)
        opt = net.configure_optimizers()["optimizer"]
        gs = GraphedStep(net, opt, batches[0])
        gs.set_lr(0.0)
        l1 = float(gs(batches[0])["loss"].detach())
        l2 = float(gs(batches[0])["loss"].detach())
        assert l1 != l2 and abs(l1 - l2) < 0.5 * abs(l1)
    finally:
        config.set_device_dropout_step(False)


@pytest.mark.parametrize("task,B", [("mimic", 128), ("mmimdb", 32), ("mmimdb", 256)])
def test_wide_models_bf16_at_config_batches_vs_oracle(task, B, dev):
    """MIMIC-H at its cfg batch (128), MM-IMDb at its cfg batch (32 per GPU) and at 256 (the D = 256 token backward
    walks several column blocks per workgroup there): bf16 against autograd through the oracle."""
    from m2_mixer_amd.engine import MimicEngine, MMIMDBEngine
    if task == "mimic":
        cfg = dict(G.MIMIC_H, dropout=0.0)
        shapes, batch = G.mimic_shapes(cfg), G.mimic_batch(B, 61, cfg)
        eng = MimicEngine(cfg, B, device=dev, precision="bf16", lr=1e-3, init=False)
        fwd = lambda p: O.mimic_forward(*batch, p, cfg)
        names = ("logits_static", "logits_time", "logits")
    else:
        cfg = dict(G.MMIMDB, dropout=0.0)
        shapes, batch = G.mmimdb_shapes(cfg), G.mmimdb_batch(B, 62, cfg)
        eng = MMIMDBEngine(cfg, B, device=dev, precision="bf16", lr=1e-3, init=False)
        fwd = lambda p: O.mmimdb_forward(*batch, p, cfg, torch.tensor(cfg["pos_weight"]))
        names = ("image_logits", "text_logits", "logits")
    params = dict(G.make_params(shapes, 63))
    eng.load_state_dict(params)
    eng.forward_backward(*(t.to(dev) for t in batch))
    torch.cuda.synchronize()
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = fwd(leaves)
    ref["loss"].backward()
    for i, k in enumerate(names):
        assert observe("bf16 logits (abs)", abserr(eng.logits[i], ref[k]), BF16_LOGITS) < BF16_LOGITS * max(1.0, float(ref[k].detach().abs().max())), k
    assert abs(float(eng.losses[3]) - float(ref["loss"].detach())) < 2e-2 * max(1.0, abs(float(ref["loss"].detach())))
    for k, leaf in leaves.items():
        if k.endswith("token_mix.2.net.3.bias"):
            continue
        assert observe("bf16 gradients (rel to max)", relerr(eng.grads[k], leaf.grad), BF16_GRAD_REL) < BF16_GRAD_REL, k


def test_capture_leaves_the_model_untouched_and_graphs_keep_their_own_outputs(dev):
    """capture() needs warm-up steps; it must put back everything they changed.  A multi-step graph has its own
    per-step output buffers and must not redirect the single-step graph's."""
    from m2_mixer_amd.engine import AVMnistEngine
    cfg, B = dict(G.AVMNIST["S"]), 32
    batch = tuple(t.to(dev) for t in G.avmnist_batch(B, 70, cfg))
    eng = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-3, seed=9)
    before = [t.clone() for t in (eng.flat_p, eng.flat_m, eng.flat_v, eng.flat_g, eng.adam_state, eng.drop_step)]
    single = eng.capture(*batch)
    multi = eng.capture(*batch, steps=3)
    torch.cuda.synchronize()
    for a, b in zip(before, (eng.flat_p, eng.flat_m, eng.flat_v, eng.flat_g, eng.adam_state, eng.drop_step)):
        assert torch.equal(a, b)
    # same state, same counters: the eager step, the single-step graph and step 0 of the 3-step graph agree
    ref = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-3, seed=9)
    ref.train_step(*batch)
    out1 = single().clone()
    torch.cuda.synchronize()
    assert torch.allclose(out1, ref.losses, rtol=0, atol=2e-3)
    assert out1.shape == (4,) and single.losses is eng.losses
    outm = multi()
    torch.cuda.synchronize()
    assert outm.shape == (3, 4) and multi.losses is outm
    assert torch.equal(eng.losses, out1), "the multi-step graph must not write the single-step graph's loss buffer"
    assert float(eng.adam_state[0]) == 4.0 and int(eng.drop_step[0]) == 4


def _oracle_epoch(data, split, batch_size, params, cfg):
    """What the reference's validation loop accumulates (modules/train_test_module.py:95-113): step losses and hits."""
    image, audio, labels = (t.cpu() for t in data.splits[split])
    losses, hits, n = [], np.zeros(3), labels.shape[0]
    for lo in range(0, n, batch_size):
        sl = slice(lo, min(lo + batch_size, n))
        out = O.avmnist_forward(image[sl], audio[sl], labels[sl], params, cfg)
        losses.append([float(out[k]) for k in ("loss_image", "loss_audio", "loss_fusion", "loss")] + [sl.stop - sl.start])
        hits += [int((out[k] == labels[sl]).sum()) for k in ("preds_image", "preds_audio", "preds")]
    return np.array(losses), hits


def test_run_epoch_covers_every_sample_and_matches_the_oracle(dev, tmp_path):
    """f1 / f2: validation over a split whose size is not a multiple of the batch (the reference keeps the last partial
    batch, datasets/avmnist.py:180-190): accumulated losses and all three heads' hit counts against the oracle evaluated
    over the same split; then one training epoch with a ragged last batch advances Adam once per batch."""
    from test_host_cpu import _write_avmnist
    from m2_mixer_amd.data import ResidentAVMnist, prepare_tail_engine, run_epoch
    from m2_mixer_amd.engine import AVMnistEngine
    root = str(tmp_path / "avmnist")
    _write_avmnist(root, 300, 37, seed=5, learnable=True)           # train 275, val 25, test 37
    data = ResidentAVMnist(root, device=dev)
    cfg, B = dict(G.AVMNIST["S"], dropout=0.0), 16
    eng = AVMnistEngine(cfg, B, device=dev, precision="fp32", lr=1e-3, init=False)
    params = dict(G.make_params(G.avmnist_shapes(cfg), 77))
    eng.load_state_dict(params)
    for split in ("val", "train"):
        got = run_epoch(eng, data, split, B, train=False)
        ls, hits = _oracle_epoch(data, split, B, params, cfg)
        n = int(ls[:, 4].sum())
        assert got["samples"] == n == data.splits[split][2].shape[0] and got["steps"] == len(ls)
        assert n % B != 0, "the split must end in a partial batch for this test to mean anything"
        assert abs(got["loss"] - float((ls[:, 3] * ls[:, 4]).sum() / n)) < FP32_ATOL
        assert abs(got["loss_step_mean"] - float(ls[:, 3].mean())) < FP32_ATOL
        for j, k in enumerate(("loss_image", "loss_audio", "loss_fusion")):
            assert abs(got[k] - float(ls[:, j].mean())) < FP32_ATOL, k
        assert [got["hits_image"], got["hits_audio"], got["hits"]] == [int(h) for h in hits]
    # training: 275 = 17 x 16 + 3 -> 18 optimizer steps, the last one through the 3-sample sibling engine
    tail = prepare_tail_engine(eng, data, "train", B)               # the training sibling exists BEFORE the capture
    assert tail is not None and tail.B == 3
    replay = eng.capture(*next(iter(data.batches("train", B))))
    tr = run_epoch(eng, data, "train", B, train=True, replay=replay)
    torch.cuda.synchronize()
    assert tr["steps"] == 18 and tr["samples"] == 275 and float(eng.adam_state[0]) == 18.0
    state, p2 = {}, dict(params)
    image, audio, labels = (t.cpu() for t in data.splits["train"])
    for lo in range(0, 275, B):
        sl = slice(lo, min(lo + B, 275))
        O.avmnist_train_step(image[sl], audio[sl], labels[sl], p2, cfg, state, lr=1e-3)
    va, (ls, hits) = run_epoch(eng, data, "val", B, train=False), _oracle_epoch(data, "val", B, p2, cfg)
    assert abs(va["loss"] - float((ls[:, 3] * ls[:, 4]).sum() / ls[:, 4].sum())) < 5e-3


def test_embeddings_inside_the_tower_forward_launch_vs_oracle(dev, monkeypatch):
    """M2M_EMBED_FOLD=1 (off by default: measured no faster): the two patch embeddings computed by the tower workgroups
    themselves (m2m_towers_forward_embeds), the step head split between that launch (losses = 0, Adam step) and the merged
    weight-gradient launch (dropout counter, m2m_towers_wgrad_tail).  Same oracle comparison as the default path at a ragged
    batch, and the counters end where the prologue form leaves them."""
    from m2_mixer_amd.engine import AVMnistEngine
    monkeypatch.setenv("M2M_EMBED_FOLD", "1")
    cfg, B = dict(G.AVMNIST["B"], dropout=0.5), 40
    probe = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-2, init=False)
    assert probe._embed_fold, "M2M_EMBED_FOLD=1 did not select the folded launch"
    batch = tuple(t.to(dev) for t in G.avmnist_batch(B, 18, cfg))
    for k in range(1, 4):
        probe.train_step(*batch)
        torch.cuda.synchronize()
        assert float(probe.adam_state[0]) == float(k) and int(probe.drop_step[0]) == k
    del probe
    test_bench_instantiation_with_dropout_vs_oracle(0.5, B, dev)


@pytest.mark.parametrize("which,env", [("recomp", {"M2M_WGRAD_RECOMP": "1"}), ("fused_heads", {"M2M_FUSED_HEADS": "1"}),
                                       ("tickets", {"M2M_BWD_TICKETS": "1"})])
def test_opt_in_paths_vs_oracle(which, env, dev):
    """The kernels that are OFF by default (recompute-form weight gradients, heads in the fusion backward's prologue, the
    backward column loop's ticket counter) stay parity-green: each runs M2-Mixer-B / bf16 / B = 40 / dropout 0.5 against the
    oracle in a child process (the library reads these switches once per process); the child asserts the path was taken."""
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "opt_in_child.py")
    r = subprocess.run([sys.executable, child, which], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"{which}: {r.stdout[-2000:]}\n{r.stderr[-3000:]}"


def test_sibling_that_changes_the_kept_ranges_is_refused_after_capture(dev, monkeypatch):
    """a captured graph holds the Adam range table by value; a training sibling whose backward overwrites
    FEWER ranges than the captured optimizer leaves uncleared would accumulate onto stale gradients.  engine.sibling refuses
    that after capture(); an evaluating sibling (trains=False) changes nothing and cannot train."""
    from m2_mixer_amd.engine import AVMnistEngine
    cfg, B = dict(G.AVMNIST["S"], dropout=0.0), 16
    eng = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-3)
    if not any(k for *_, k in eng._ranges_add):
        pytest.skip("this configuration keeps no gradient range uncleared")
    image, audio, labels = (t.to(dev) for t in G.avmnist_batch(B, 3, cfg))
    before = list(eng._ranges_add)
    ev = eng.sibling(5, trains=False)                              # validation engine: nothing narrows
    assert eng._ranges_add == before
    with pytest.raises(RuntimeError, match="only evaluates"):
        ev.train_step(*(t[:5].contiguous() for t in (image, audio, labels)))
    same = eng.sibling(7)                                          # same overwrite set: fine before and after capture
    assert eng._ranges_add == before and same._ranges_add
    eng.capture(image, audio, labels)
    eng.sibling(9)
    monkeypatch.setenv("M2M_WGRAD_OVERWRITE", "0")                 # a sibling that overwrites nothing
    with pytest.raises(RuntimeError, match="before capture"):
        eng.sibling(11)
    assert eng._ranges_add == before                               # (and the refusal left this engine's table alone)


def _lightning_ckpt(path, state_dict, epoch=3):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"epoch": epoch, "global_step": 100, "pytorch-lightning_version": "1.8.6",
                "state_dict": {k: v.clone() for k, v in state_dict.items()}, "optimizer_states": [], "lr_schedulers": []}, path)
    return path


def test_checkpoint_to_predictions_round_trip_vs_oracle(dev, tmp_path):
    """f4 on the GPU: oracle-seeded weights written in Lightning's `.ckpt` layout under the reference's key names ->
    load_from_checkpoint -> to_engine -> evaluate: logits within 1e-3 (fp32), class predictions bit-exact against the
    oracle; test_preds.pt has the reference's keys and shapes (models/avmnist.py:382-398).  Then MM-IMDb, whose reference
    state_dict carries the loss modules' pos_weight buffers; and an engine's Adam state through optimizer_states."""
    from m2_mixer_amd import models as MD
    c = G.AVMNIST["S"]
    mods = {"image": dict(c["image"], block_type="MLPMixer"), "audio": dict(c["audio"], block_type="MLPMixer"),
            "multimodal": dict(c["multimodal"], block_type="FusionMixer", fusion_function="ConcatFusion"),
            "classification": dict(classifier="StandardClassifier", num_classes=10, input_shape=[16, 49, c["multimodal"]["hidden_dim"]])}
    cfg, ocfg = {"dropout": 0.1, "modalities": mods}, {"lr": 1e-2, "scheduler_patience": 2}
    params = G.make_params(G.avmnist_shapes(c), 81)
    ck = _lightning_ckpt(str(tmp_path / "version_0" / "checkpoints" / "epoch=3-step=100.ckpt"), params)
    net = MD.AVMnistMixerMultiLoss.load_from_checkpoint(ck, model_cfg=cfg, optimizer_cfg=dict(ocfg)).to(dev)
    B = 24
    image, audio, labels = G.avmnist_batch(2 * B, 82, c)
    eng = net.to_engine(B, precision="fp32")
    outs = []
    for lo in (0, B):
        sl = slice(lo, lo + B)
        res = eng.evaluate(image[sl].to(dev), audio[sl].to(dev), labels[sl].to(dev))
        torch.cuda.synchronize()
        ref = O.avmnist_forward(image[sl], audio[sl], labels[sl], params, c)
        for k in ("logits", "image_logits", "audio_logits"):
            assert abserr(res[k], ref[k]) < FP32_ATOL, k
        for k in ("preds", "preds_image", "preds_audio"):
            assert torch.equal(res[k].cpu().long(), ref[k]), k
        outs.append({"labels": labels[sl], **{k: res[k].clone() for k in net.TEST_PRED_KEYS if k != "labels"}})
    dump = torch.load(net.save_test_preds(outs))
    assert os.path.dirname(net.checkpoint_path) == str(tmp_path / "version_0" / "checkpoints")
    assert sorted(dump) == sorted(["preds", "preds_image", "preds_audio", "labels", "image_logits", "audio_logits", "logits"])
    assert dump["logits"].shape == (2 * B, 10) and dump["preds"].shape == (2 * B,)
    full = O.avmnist_forward(image, audio, labels, params, c)
    assert torch.equal(dump["preds"].long(), full["preds"]) and abserr(dump["logits"], full["logits"]) < FP32_ATOL
    # the module path (torch autograd over the HIP towers) agrees with the engine on the loaded weights
    net.eval()
    with torch.no_grad():
        mo = net.shared_step({"image": image[:B].to(dev), "audio": audio[:B].to(dev), "label": labels[:B].to(dev)}, mode="val")
    assert observe("bf16 logits (abs)", abserr(mo["logits"], full["logits"][:B]), BF16_LOGITS) < BF16_LOGITS        # module default precision is bf16
    # ---- an engine's training state through save_checkpoint(engine=...) and back
    eng.train_step(image[:B].to(dev), audio[:B].to(dev), labels[:B].to(dev))
    torch.cuda.synchronize()
    out = net.save_checkpoint(str(tmp_path / "resume" / "last.ckpt"), epoch=4, global_step=101, engine=eng)
    raw = torch.load(out, weights_only=True)
    assert len(raw["optimizer_states"]) == 1 and len(raw["optimizer_states"][0]["state"]) == len(eng.shapes)
    again = MD.AVMnistMixerMultiLoss.load_from_checkpoint(out, model_cfg=cfg, optimizer_cfg=dict(ocfg)).to(dev)
    eng2 = again.to_engine(B, precision="fp32")
    eng2.load_optimizer_state_dict(raw["optimizer_states"][0])
    assert torch.equal(eng2.flat_p, eng.flat_p) and torch.equal(eng2.flat_m, eng.flat_m) and torch.equal(eng2.flat_v, eng.flat_v)
    assert float(eng2.adam_state[0]) == 1.0
    # torch.optim.Adam accepts the same optimizer state (the layout Lightning stores)
    opt = again.configure_optimizers()["optimizer"]
    opt.load_state_dict(raw["optimizer_states"][0])
    # ---- MM-IMDb: the reference's state_dict ends with the three criteria's pos_weight buffers
    cm = G.MMIMDB
    mmods = {"image": dict(cm["image"], block_type="MLPMixer"), "text": dict(cm["text"], block_type="MLPMixer"),
             "multimodal": dict(cm["multimodal"], block_type="FusionMixer", fusion_function="ConcatFusion"),
             "classification": dict(classifier="StandardClassifier", num_classes=cm["num_classes"],
                                    input_shape=[16, 49, cm["multimodal"]["hidden_dim"]])}
    mcfg = {"dropout": 0.0, "modalities": mmods, "pos_weight": [1.0] * cm["num_classes"]}       # cfg value differs from the checkpoint's
    mparams = G.make_params(G.mmimdb_shapes(cm), 83)
    pw = torch.tensor(cm["pos_weight"], dtype=torch.float32)
    sd = dict(mparams, **{f"{n}_criterion.pos_weight": pw.clone() for n in ("image", "text", "fusion")})
    mck = _lightning_ckpt(str(tmp_path / "mm" / "last.ckpt"), sd)
    mnet = MD.MMIMDBMixerMultiLoss.load_from_checkpoint(mck, model_cfg=mcfg, optimizer_cfg={"lr": 1e-3}).to(dev)
    Bm = 4
    mi, mt, ml = G.mmimdb_batch(Bm, 84, cm)
    meng = mnet.to_engine(Bm, precision="fp32")
    assert torch.equal(meng.pos_weight.cpu(), pw), "the checkpoint's pos_weight, not the cfg's, must reach the heads kernel"
    mres = meng.evaluate(mi.to(dev), mt.to(dev), ml.to(dev))
    torch.cuda.synchronize()
    mref = O.mmimdb_forward(mi, mt, ml, mparams, cm, pw)
    assert abserr(mres["logits"], mref["logits"]) < FP32_ATOL and abs(float(mres["loss"]) - float(mref["loss"])) < FP32_ATOL
    assert torch.equal(mres["preds"].cpu().long(), mref["preds"])
    assert list(meng.state_dict().keys()) == list(sd.keys())


def test_module_path_is_reentrant(dev):
    """SURVEY section 8b asks for re-entrancy: a validation forward between a training forward and its backward, and
    gradient accumulation over two micro-batches (two forwards, then the two backwards), must give the gradients of the
    plain sequence.  Towers (fused and wide path) and the static MLP, fp32 mode, against autograd through the oracle."""
    import m2_mixer_amd as M
    from m2_mixer_amd import modules as MM
    M.set_precision("fp32")
    for case in ((4, 128, 32, 3072), (24, 64, 16, 64)):
        N, D, T, Cc = case
        p, x1, dy1 = G.block_case_tensors(case, 5, seed=91)
        _, x2, dy2 = G.block_case_tensors(case, 3, seed=92)              # another batch size
        blk = MM.MixerBlock(D, N, T, Cc, dropout=0.0).to(dev)
        blk.load_state_dict(p)
        blk.train()
        a = x1.to(dev).requires_grad_(True)
        b = x2.to(dev).requires_grad_(True)
        ya = blk(a)
        with torch.no_grad():
            blk(x2.to(dev))                                              # validation-style forward in between
            # ... and an MC-dropout pass (train mode, dropout ON, no autograd) at the SAME batch size as the pending
            # forward: the kernels save activations then too -- into a scratch set, not into `ya`'s
            blk.dropout_p = 0.5
            blk(x1.to(dev) * 3.0)
            blk.dropout_p = 0.0
        yb = blk(b)
        (yb * dy2.to(dev)).sum().backward()                              # backwards in the opposite order
        (ya * dy1.to(dev)).sum().backward()
        torch.cuda.synchronize()
        leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        r1, r2 = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
        ((O.mixer_block(r1, leaves) * dy1).sum() + (O.mixer_block(r2, leaves) * dy2).sum()).backward()
        assert abserr(a.grad, r1.grad) < FP32_ATOL and abserr(b.grad, r2.grad) < FP32_ATOL
        for k, prm in blk.named_parameters():
            assert relerr(prm.grad, leaves[k].grad) < 1e-3, (case, k)
    cs = G.MIMIC_H["static"]
    mlp = MM.MLP(cs["input_dim"], cs["hidden_dim"], cs["num_blocks"], cs["output_dim"], dropout=0.0).to(dev)
    xa, xb = torch.randn(7, cs["input_dim"], device=dev), torch.randn(4, cs["input_dim"], device=dev)
    ya = mlp(xa)
    yb = mlp(xb)
    yb.square().sum().backward()
    ya.square().sum().backward()
    torch.cuda.synchronize()
    ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mlp.state_dict().items()}
    (O.mlp(xa.cpu(), ref, "", cs["num_blocks"], True).square().sum() + O.mlp(xb.cpu(), ref, "", cs["num_blocks"], True).square().sum()).backward()
    for k, prm in mlp.named_parameters():
        assert relerr(prm.grad, ref[k].grad) < 1e-4, k


def test_towers_deeper_than_one_descriptor(dev):
    """The reference's sweeps go to 16 mixers per tower (sweeps/avmnist_mixer.yaml:19-35); one m2m_tower holds 8 blocks, so
    the module path chains descriptors.  11 blocks (8 + 3), fp32, dropout on (masks exported per chunk), against autograd
    through the oracle."""
    import m2_mixer_amd as M
    from m2_mixer_amd import modules as MM
    M.set_precision("fp32")
    cfg = dict(hidden_dim=32, token_dim=16, channel_dim=64, num_mixers=11)
    N, B = 8, 5
    tower = MM.get_block_by_name(block_type="FusionMixer", num_patches=N, dropout=0.0, **cfg).to(dev)
    shapes = G.tower_shapes("", cfg, N, "none")
    params = G.make_params(shapes, 95)
    tower.load_state_dict(params)
    assert len(tower.mixer_blocks) == 11
    x = torch.randn(B, N, cfg["hidden_dim"])
    xg = x.to(dev).requires_grad_(True)
    y = tower(xg)
    y.square().sum().backward()
    torch.cuda.synchronize()
    assert len(tower._rts) == 2 and [rt.nblocks for rt in tower._rts] == [8, 3]
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xr = x.clone().requires_grad_(True)
    yo = O.fusion_mixer(xr, leaves, "", 11)
    yo.square().sum().backward()
    assert abserr(y, yo) < FP32_ATOL and abserr(xg.grad, xr.grad) < 2e-3
    for k, prm in tower.named_parameters():
        if k.endswith("token_mix.2.net.3.bias"):
            continue
        assert relerr(prm.grad, leaves[k].grad) < 2e-3, k
