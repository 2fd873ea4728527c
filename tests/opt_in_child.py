"""Child process of tests/test_gpu_bench_path.py::test_opt_in_paths_vs_oracle: the library reads its opt-in switches
(M2M_WGRAD_RECOMP, M2M_FUSED_HEADS, M2M_BWD_TICKETS) ONCE per process, so each one is exercised in a process of its own --
AV-MNIST M2-Mixer-B, bf16, batch 40 (ragged tiles), dropout 0.5, against the CPU oracle fed the kernels' own masks; asserts
that the switched path really is the one that ran.  Usage: python tests/opt_in_child.py <recomp|fused_heads|tickets>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(which: str) -> int:
    import torch
    import test_gpu_bench_path as T
    from m2_mixer_amd.engine import AVMnistEngine
    import gen_util as G
    dev = torch.device("cuda:0")
    cfg, B = dict(G.AVMNIST["B"], dropout=0.5), 40
    probe = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-2, init=False)
    if which == "recomp":
        assert probe.t_a.wgrad_form(B) == 1 and probe.t_fus.wgrad_form(B) == 1, "M2M_WGRAD_RECOMP=1 did not select the recompute form"
    elif which == "fused_heads":
        assert probe._fused_heads, "M2M_FUSED_HEADS=1 did not select the heads-in-backward launch"
    elif which == "tickets":
        assert os.environ.get("M2M_BWD_TICKETS") == "1"
    else:
        raise SystemExit(f"unknown path {which}")
    del probe
    T.test_bench_instantiation_with_dropout_vs_oracle(0.5, B, dev)
    print(f"opt-in path {which}: parity with the oracle ok")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
