#!/usr/bin/env python3
"""Kernel bring-up report: runs every libm2mixer entry point on small cases and prints the error
against the CPU oracle (no asserts -- one run, maximum information).  GPU box only."""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import gen_util as G                      # noqa: E402
from oracle import m2mixer_oracle as O    # noqa: E402
import m2_mixer_amd as M                  # noqa: E402
from m2_mixer_amd import _lib as L        # noqa: E402
from m2_mixer_amd import modules as MM    # noqa: E402
from m2_mixer_amd.runtime import BLOCK_KEYS  # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()), float(b.abs().max())


def section(name):
    print(f"\n=== {name} ===", flush=True)


def run(fn):
    try:
        fn()
    except Exception:
        traceback.print_exc()
    torch.cuda.synchronize()


def t_gemm():
    section("gemm_probe (packed layouts, chained accumulators)")
    lib = L.lib()
    for prec, name in ((L.PREC_F32, "fp32"), (L.PREC_BF16, "bf16")):
        for (I, J, K, J2) in ((16, 32, 32, 16), (40, 70, 100, 24), (64, 96, 128, 128)):
            rng = np.random.default_rng(I + J)
            A = torch.from_numpy(rng.standard_normal((I, K)).astype(np.float32)).to(dev)
            Bm = torch.from_numpy(rng.standard_normal((J, K)).astype(np.float32)).to(dev)
            Bc = torch.from_numpy(rng.standard_normal((J2, J)).astype(np.float32)).to(dev)
            Cc = torch.zeros(I, J, device=dev)
            C2 = torch.zeros(I, J2, device=dev)
            ws = torch.zeros(L.packed_bytes(prec, I, K) + L.packed_bytes(prec, J, K) + L.packed_bytes(prec, J2, J) + 4096,
                             dtype=torch.uint8, device=dev)
            L.check(lib.m2m_gemm_probe(prec, A.data_ptr(), Bm.data_ptr(), I, J, K, Bc.data_ptr(), J2, Cc.data_ptr(),
                                       C2.data_ptr(), ws.data_ptr(), L.stream_ptr()), "gemm_probe")
            torch.cuda.synchronize()
            if prec == L.PREC_BF16:
                Ar, Br, Bcr = (t.bfloat16().double() for t in (A, Bm, Bc))
            else:
                Ar, Br, Bcr = A.double(), Bm.double(), Bc.double()
            ref = Ar @ Br.t()
            refc = ref.float().bfloat16().double() if prec == L.PREC_BF16 else ref
            ref2 = refc @ Bcr.t()
            e1, m1 = rel(Cc, ref)
            e2, m2 = rel(C2, ref2)
            print(f"  {name} I{I} J{J} K{K} J2{J2}: C err {e1:.3e} (max {m1:.2f})  C2 err {e2:.3e} (max {m2:.2f})")


def t_gelu():
    section("gelu probe")
    x = torch.linspace(-8, 8, 100001, device=dev)
    y = torch.empty_like(x)
    dy = torch.empty_like(x)
    L.check(L.lib().m2m_gelu_probe(x.data_ptr(), y.data_ptr(), dy.data_ptr(), x.numel(), L.stream_ptr()))
    xd = x.double().cpu().requires_grad_(True)
    ref = 0.5 * xd * (1 + torch.erf(xd / np.sqrt(2)))
    ref.sum().backward()
    print("  gelu err %.3e   gelu' err %.3e" % (rel(y, ref)[0], rel(dy, xd.grad)[0]))


def block_case(case, B, prec_name, p_drop=0.0, seed=0):
    N, D, T, Cc = case
    M.set_precision(prec_name)
    p, x, dy = G.block_case_tensors(case, B, seed=1000 + seed)
    blk = MM.MixerBlock(D, N, T, Cc, dropout=p_drop).to(dev)
    blk.load_state_dict(p)
    blk.train()
    xg = x.to(dev).requires_grad_(True)
    y = blk(xg)
    (y * dy.to(dev)).sum().backward()
    torch.cuda.synchronize()
    masks = None
    if p_drop > 0:
        rt = blk._rt
        st = blk._drop_step
        sd = M.config.dropout_seed()
        m0 = rt.dropout_mask(0, 0, B, sd, st).view(B, D, T)
        m1 = rt.dropout_mask(0, 1, B, sd, st).view(B, D, N)
        m2 = rt.dropout_mask(0, 2, B, sd, st).view(B, N, rt.Cp)[:, :, :Cc]
        m3 = rt.dropout_mask(0, 3, B, sd, st).view(B, N, D)
        masks = {k: v.float().cpu() for k, v in dict(tok_h=m0, tok_o=m1, ch_h=m2, ch_o=m3).items()}
        thr = round((1 - p_drop) * 65536)
        p_eff = 1 - thr / 65536
        print("   mask keep rates:", {k: round(float(v.mean()), 4) for k, v in masks.items()}, "p_eff", p_eff)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    yo = O.mixer_block(xr, leaves, "", p_drop, masks)
    (yo * dy).sum().backward()
    out = {"y": rel(y, yo), "dx": rel(xg.grad, xr.grad)}
    for k, prm in blk.named_parameters():
        out[k] = rel(prm.grad, leaves[k].grad)
    return out


def t_blocks():
    section("MixerBlock fwd/bwd vs oracle")
    for prec in ("fp32", "bf16"):
        for ci, case in enumerate(G.BLOCK_CASES):
            N, D, T, Cc = case
            if N > 8 or D > 128:
                continue
            for B in (2, 37):
                try:
                    r = block_case(case, B, prec, seed=ci)
                    worst = max(r.items(), key=lambda kv: kv[1][0] / (kv[1][1] + 1e-12))
                    print(f"  {prec} case{ci} {case} B{B}: y {r['y'][0]:.2e}/{r['y'][1]:.2f} dx {r['dx'][0]:.2e}/{r['dx'][1]:.2f} "
                          f"worst-rel {worst[0]} {worst[1][0]:.2e}/{worst[1][1]:.2e}")
                    if B == 2 and prec == "fp32":
                        for k, v in r.items():
                            print(f"       {k:40s} err {v[0]:.3e} ref max {v[1]:.3e}")
                except Exception:
                    traceback.print_exc()


def t_dropout():
    section("MixerBlock with dropout 0.5 (kernel masks exported to the oracle)")
    for prec in ("fp32", "bf16"):
        for case in ((4, 128, 32, 3072), (8, 32, 16, 256)):
            r = block_case(case, 5, prec, p_drop=0.5, seed=3)
            for k, v in r.items():
                print(f"   {prec} {case} {k:40s} err {v[0]:.3e} ref max {v[1]:.3e}")


def t_avmnist():
    section("AV-MNIST towers (module path) vs oracle: S and B, fp32 + bf16")
    for size, B in (("S", 8), ("B", 8), ("B", 70)):
        cfg = G.AVMNIST[size]
        shapes = G.avmnist_shapes(cfg)
        params = G.make_params(shapes, 11)
        image, audio, labels = G.avmnist_batch(B, 12, cfg)
        leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        ro = O.avmnist_forward(image, audio, labels, leaves, cfg)
        ro["loss"].backward()
        for prec in ("fp32", "bf16"):
            M.set_precision(prec)
            img = MM.MLPMixer(**cfg["image"], dropout=0.0).to(dev)
            aud = MM.MLPMixer(**cfg["audio"], dropout=0.0).to(dev)
            fus = MM.FusionMixer(**cfg["multimodal"], num_patches=img.num_patch + aud.num_patch, dropout=0.0).to(dev)
            img.load_state_dict({k[len("image_mixer."):]: v for k, v in params.items() if k.startswith("image_mixer.")})
            aud.load_state_dict({k[len("audio_mixer."):]: v for k, v in params.items() if k.startswith("audio_mixer.")})
            fus.load_state_dict({k[len("fusion_mixer."):]: v for k, v in params.items() if k.startswith("fusion_mixer.")})
            for m in (img, aud, fus):
                m.train()
            it = img(image.to(dev))
            at = aud(audio.to(dev))
            ft = fus(torch.cat([it, at], dim=1))
            print(f"  {size} B{B} {prec}: image_tokens {rel(it, ro['image_tokens'])}  audio_tokens {rel(at, ro['audio_tokens'])}  "
                  f"fusion_tokens {rel(ft, ro['fusion_tokens'])}")
            # backward through everything with the oracle's upstream gradients
            P = {k: v.to(dev) for k, v in params.items()}
            il = it.mean(1) @ P["classifier_image.weight"].t() + P["classifier_image.bias"]
            al = at.mean(1) @ P["classifier_audio.weight"].t() + P["classifier_audio.bias"]
            fl = ft.mean(1) @ P["classifier_fusion.classifer.weight"].t() + P["classifier_fusion.classifer.bias"]
            ce = torch.nn.functional.cross_entropy
            lab = labels.to(dev)
            loss = ce(il, lab) + ce(al, lab) + ce(fl, lab)
            loss.backward()
            print(f"     loss {float(loss):.6f} vs {float(ro['loss']):.6f}   logits err {rel(fl, ro['logits'])}")
            worst = []
            for pref, mod in (("image_mixer.", img), ("audio_mixer.", aud), ("fusion_mixer.", fus)):
                for k, prm in mod.named_parameters():
                    e, mx = rel(prm.grad, leaves[pref + k].grad)
                    worst.append((e / (mx + 1e-9), pref + k, e, mx))
            worst.sort(reverse=True)
            for w in worst[:6]:
                print(f"     grad {w[1]:55s} err {w[2]:.3e} ref max {w[3]:.3e}")


def t_heads_adam():
    section("heads_ce + adam")
    from m2_mixer_amd.runtime import heads_ce
    B, D, K = 70, 128, 10
    rng = np.random.default_rng(5)
    heads, refs = [], []
    labels = torch.from_numpy(rng.integers(0, K, size=(B,), dtype=np.int64))
    for h in range(3):
        pooled = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32))
        w = torch.from_numpy((rng.standard_normal((K, D)) * 0.1).astype(np.float32))
        b = torch.from_numpy((rng.standard_normal((K,)) * 0.1).astype(np.float32))
        refs.append((pooled.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)))
        heads.append(dict(pooled=pooled.to(dev), w=w.to(dev), b=b.to(dev), g_w=torch.zeros(K, D, device=dev),
                          g_b=torch.zeros(K, device=dev), d_pooled=torch.zeros(B, D, device=dev), weight=1.0 + 0.5 * h))
    logits, losses, preds = heads_ce(heads, labels.to(dev), B, D, K)
    torch.cuda.synchronize()
    tot = 0
    for h, (p, w, b) in enumerate(refs):
        lg = p @ w.t() + b
        l = O.cross_entropy(lg, labels)
        tot = tot + heads[h]["weight"] * l
        print(f"  head{h}: logits {rel(logits[h], lg)} loss {float(losses[h]):.6f} vs {float(l):.6f} "
              f"preds eq {bool((preds[h].cpu() == lg.argmax(1)).all())}")
    tot.backward()
    for h, (p, w, b) in enumerate(refs):
        print(f"  head{h}: d_pooled {rel(heads[h]['d_pooled'], p.grad)} g_w {rel(heads[h]['g_w'], w.grad)} g_b {rel(heads[h]['g_b'], b.grad)}")
    print(f"  total {float(losses[3]):.6f} vs {float(tot):.6f}")
    # adam
    n = 100003
    p0 = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
    g = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    pd, m, v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    state = torch.tensor([0.0, 1e-2, 0, 0], device=dev)
    for it in range(3):
        pt.grad = g * (it + 1)
        opt.step()
        gd = (g * (it + 1)).to(dev)
        L.check(L.lib().m2m_adam_step(pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, state.data_ptr(),
                                      0.9, 0.999, 1e-8, 0.0, 1.0, 1, L.stream_ptr()))
    torch.cuda.synchronize()
    print("  adam 3 steps err", rel(pd, pt.detach()))


def t_engine():
    section("AVMnistEngine (fused step) vs oracle train steps")
    from m2_mixer_amd.engine import AVMnistEngine
    for size, B, precs in (("S", 8, ("fp32", "bf16")), ("B", 8, ("fp32", "bf16"))):
        cfg = dict(G.AVMNIST[size])
        cfg0 = dict(cfg, dropout=0.0)
        shapes = G.avmnist_shapes(cfg)
        image, audio, labels = G.avmnist_batch(B, 12, cfg)
        for prec in precs:
            params = dict(G.make_params(shapes, 11))
            eng = AVMnistEngine(cfg0, B, precision=prec, lr=1e-2, init=False)
            assert list(eng.shapes.keys()) == list(shapes.keys())
            eng.load_state_dict(params)
            state = {}
            di, da, dl = image.to(dev), audio.to(dev), labels.to(dev)
            for step in range(2):
                ro = O.avmnist_train_step(image, audio, labels, params, cfg, state, lr=1e-2)
                eng.forward_backward(di, da, dl)
                torch.cuda.synchronize()
                gsnap = {k: v.clone() for k, v in eng.grads.items()}
                eng.optimizer_step()
                torch.cuda.synchronize()
                print(f"  {size} {prec} step{step}: loss {float(eng.losses[3]):.6f} vs {float(ro['loss']):.6f}  "
                      f"logits {rel(eng.logits[2], ro['logits'])} img {rel(eng.logits[0], ro['image_logits'])} "
                      f"preds eq {bool((eng.preds[2].cpu() == ro['preds']).all())}")
                if step == 0:
                    worst = sorted(((rel(gsnap[k], g)[0] / (float(g.abs().max()) + 1e-9), k, rel(gsnap[k], g)) for k, g in ro["grads"].items()
                                    if not k.endswith("token_mix.2.net.3.bias")), reverse=True)[:5]
                    for w in worst:
                        print(f"      grad {w[1]:55s} err {w[2][0]:.3e} ref max {w[2][1]:.3e}")
            worst = sorted(((rel(eng.params[k], v)[0], k) for k, v in params.items() if not k.endswith("token_mix.2.net.3.bias")), reverse=True)[:3]
            print("      params after 2 steps, worst abs err:", [(round(a, 6), k) for a, k in worst])
    # graph capture + dropout: losses must change between replays (fresh masks) and stay finite
    cfg = dict(G.AVMNIST["B"])
    B = 64
    eng = AVMnistEngine(cfg, B, precision="bf16", lr=1e-3)
    image, audio, labels = (t.to(dev) for t in G.avmnist_batch(B, 5, cfg))
    replay = eng.capture(image, audio, labels)
    ls = []
    for i in range(6):
        replay()
        torch.cuda.synchronize()
        ls.append([round(float(v), 4) for v in eng.losses])
    print("  graph replays (B, bf16, dropout 0.5) losses:", ls)
    print("  adam step counter", float(eng.adam_state[0]), "dropout counter", int(eng.drop_step[0]))


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), torch.__version__)
    t0 = time.time()
    which = sys.argv[1:] or ["gemm", "gelu", "blocks", "dropout", "avmnist", "heads"]
    table = dict(engine=t_engine, gemm=t_gemm, gelu=t_gelu, blocks=t_blocks, dropout=t_dropout, avmnist=t_avmnist, heads=t_heads_adam)
    for w in which:
        run(table[w])
    print("elapsed %.1fs" % (time.time() - t0))
