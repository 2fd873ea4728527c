"""Resident-in-HBM input pipeline and the epoch loop around the fused engines.

The reference feeds the model through a torch DataLoader with `num_workers: 2` (datasets/avmnist.py:163-190) and pulls
`loss.cpu().item()` to the host every step (modules/train_test_module.py:72-84).  At > 600 k samples/s neither survives:
the whole AV-MNIST training split is 55 000 x 53 KB = 2.9 GB in fp32, a rounding error of one MI355X's 288 GB, so it
is loaded ONCE into device memory in the reference's on-disk format and batches are views (train / val: `shuffle=False`
in the reference) or one device-side gather (test: `shuffle=True`); per-step losses and hit counts accumulate in device
memory and reach the host once per `log_interval_steps`, so the captured graph is never interrupted by a sync.
"""
from __future__ import annotations

import os
from typing import Dict, Iterator, Optional, Tuple

import numpy as np
import torch


class ResidentAVMnist:
    """`root/{image,audio}/{train,test}_data.npy`, `root/{train,test}_labels.npy` (datasets/avmnist.py:105-114):
    image (N, 784) or (N, 28, 28) -> (N, 1, 28, 28); audio (N, 112, 112) -> (N, 1, 112, 112); `astype(float32)` only
    (datasets/avmnist.py:17-21).  Splits as AVMnistDataModule.setup (:175-181): the first 55 000 training samples train,
    the rest validate; smaller sets (tests, subsets) keep the 11 : 1 proportion."""

    TRAIN_SPLIT = 55000

    def __init__(self, root_dir: str, device="cuda:0", rank: int = 0, world: int = 1):
        self.device = torch.device(device)
        self.rank, self.world = rank, world
        self.splits: Dict[str, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = {}
        tr = self._load(root_dir, "train")
        n = tr[2].shape[0]
        cut = self.TRAIN_SPLIT if n > self.TRAIN_SPLIT else (n * 11) // 12
        self.splits["train"] = tuple(t[:cut] for t in tr)
        self.splits["val"] = tuple(t[cut:] for t in tr)
        if os.path.exists(os.path.join(root_dir, "test_labels.npy")):
            self.splits["test"] = self._load(root_dir, "test")

    def _load(self, root: str, stage: str):
        image = np.load(os.path.join(root, "image", f"{stage}_data.npy"))
        audio = np.load(os.path.join(root, "audio", f"{stage}_data.npy"))
        labels = np.load(os.path.join(root, f"{stage}_labels.npy"))
        if image.shape[0] != audio.shape[0] or image.shape[0] != labels.shape[0]:
            raise ValueError(f"{stage}: image / audio / label counts differ")
        image = torch.from_numpy(image.astype(np.float32)).reshape(image.shape[0], 1, 28, 28)
        audio = torch.from_numpy(audio.astype(np.float32))[:, None, :, :]
        labels = torch.from_numpy(labels.astype(np.int64))
        return image.to(self.device), audio.to(self.device), labels.to(self.device)

    def num_samples(self, split: str) -> int:
        """Samples of `split` this rank sees.  As torch's DistributedSampler(drop_last=False) -- what Lightning puts in
        front of the reference's loaders under DDP (datasets/avmnist.py:180-190, run.py:69-70) -- EVERY rank gets
        ceil(n / world): the index list is padded with its own head until it divides evenly, then rank r takes
        r, r + world, ...  Equal counts are what keeps the ranks' collectives in step (an epoch is the same number of
        batches, of the same sizes, on every rank)."""
        n = self.splits[split][2].shape[0]
        return (n + self.world - 1) // self.world

    def num_batches(self, split: str, batch_size: int, drop_last: bool = False) -> int:
        """The reference's DataLoaders keep the ragged last batch (`drop_last` is left False, datasets/avmnist.py:180-190)."""
        n = self.num_samples(split)
        return n // batch_size if drop_last else (n + batch_size - 1) // batch_size

    def batches(self, split: str, batch_size: int, shuffle: bool = False, generator: Optional[torch.Generator] = None,
                drop_last: bool = False) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """Rank r takes samples r, r + world, ... of the padded index list (see num_samples: DistributedSampler semantics;
        under `shuffle` every rank must pass a generator in the same state).  The last batch may be smaller than batch_size
        (drop_last=False, as the reference)."""
        image, audio, labels = self.splits[split]
        n = labels.shape[0]
        if shuffle:
            order = torch.randperm(n, device=self.device, generator=generator)
        elif self.world > 1:
            order = torch.arange(n, device=self.device)
        else:
            order = None
        if order is not None and self.world > 1:
            pad = self.num_samples(split) * self.world - n
            if pad:
                order = torch.cat([order, order[:pad]])         # (n >= world: one wrap-around suffices)
            order = order[self.rank::self.world]
        mine = self.num_samples(split)
        for b in range(self.num_batches(split, batch_size, drop_last)):
            lo, hi = b * batch_size, min((b + 1) * batch_size, mine)
            if order is None:
                yield image[lo:hi], audio[lo:hi], labels[lo:hi]          # views: zero copies
            else:
                idx = order[lo:hi]
                yield image.index_select(0, idx), audio.index_select(0, idx), labels.index_select(0, idx)


class PlateauLR:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor=0.1, threshold=1e-4 rel) on the engine's device-side
    learning rate (models/avmnist.py:416-422: monitor `val_loss`, patience from `scheduler_patience`)."""

    def __init__(self, engine, lr: float, patience: int = 5, factor: float = 0.1, threshold: float = 1e-4, min_lr: float = 0.0):
        self.engine, self.lr, self.patience, self.factor, self.threshold, self.min_lr = engine, lr, patience, factor, threshold, min_lr
        self.best, self.bad = float("inf"), 0

    def step(self, metric: float) -> float:
        if metric < self.best * (1 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            self.lr = max(self.lr * self.factor, self.min_lr)
            self.engine.set_lr(self.lr)
            self.bad = 0
        return self.lr


def prepare_tail_engine(engine, data: "ResidentAVMnist", split: str, batch_size: int):
    """The training engine of `split`'s ragged last batch, built NOW (before engine.capture()) and kept on the engine for
    run_epoch; None when the split divides evenly."""
    rem = data.num_samples(split) % batch_size
    if rem == 0:
        return None
    tails = engine.__dict__.setdefault("_tail_engines", {})
    if rem not in tails:
        tails[rem] = engine.sibling(rem)
    return tails[rem]


def run_epoch(engine, data: ResidentAVMnist, split: str, batch_size: int, train: bool, log_interval_steps: int = 50,
              replay=None, log=None, tail_engine=None, grad_sync=None, epoch: int = 0, shuffle_seed: int = 0) -> Dict[str, float]:
    """One pass over EVERY sample of `split` (the reference's loaders keep the ragged last batch,
    datasets/avmnist.py:180-190).  train=True drives the captured step (`replay`, from engine.capture) or
    engine.train_step.  A last batch smaller than batch_size -- the captured graph and the engine's buffers have a static
    batch size -- goes through `tail_engine` (default: engine.sibling(remainder), same parameters / Adam state, built on
    first use and kept on the engine), eagerly; packed operand copies are re-synchronised around it.
    A TRAINING tail engine must exist before engine.capture() (engine.sibling narrows the gradient ranges the captured
    optimizer leaves uncleared and refuses to do so under a captured graph): prepare_tail_engine(engine, data, split, batch).
    Data parallel (world > 1): the test split is shuffled (as the reference's loader) with a generator seeded `shuffle_seed +
    epoch` on EVERY rank (torch's DistributedSampler does the same), so the ranks' strided shards partition one permutation.
    Pass the `grad_sync` the replay was captured with -- EVERY training step exchanges gradients,
    the eager ones (no replay, the ragged last batch) included, as Lightning's DDP does; the ranks see equally many samples
    (ResidentAVMnist.num_samples), hence the same sequence of collectives.  Returned metrics are this rank's (the reference
    logs without sync_dist).

    Per-step values are summed on the device and read back every `log_interval_steps` steps and at the end (cfg
    `log_interval_steps`, cfg/avmnist/*.yml:3): the four losses (modules/train_test_module.py:72-92: step losses), the three
    heads' hit counts (torchmetrics Accuracy on `preds`, :79-82, :105-110).  Returned:
      loss              sample-weighted mean of the step losses = what `self.log('val_loss', ..., on_epoch=True)` reduces to,
                        the quantity ReduceLROnPlateau / EarlyStopping monitor (run.py:61-67, models/avmnist.py:416-422)
      loss_step_mean    plain mean over steps = the wandb `val_loss` / `train_loss` number (train_test_module.py:92, :113)
      loss_image / loss_audio / loss_fusion   step means (models/avmnist.py:326-337)
      acc, acc_image, acc_audio, hits*, steps, samples"""
    dev = engine.device
    # [loss_a, loss_b, loss_fusion, loss] step sums | the same weighted by the step's batch size | hits a, b, fusion
    acc = torch.zeros(11, device=dev, dtype=torch.float64)
    seen, host = 0, np.zeros(11)
    if train and grad_sync is None and getattr(data, "world", 1) > 1:
        raise RuntimeError("run_epoch: world > 1 needs the gradient exchange (grad_sync=parallel.GradSync(...))")
    nb = data.num_batches(split, batch_size)
    shuffle, gen = split == "test", None
    if shuffle and getattr(data, "world", 1) > 1:
        gen = torch.Generator(device=data.device)
        gen.manual_seed(int(shuffle_seed) + int(epoch))
    for i, (image, audio, labels) in enumerate(data.batches(split, batch_size, shuffle=shuffle, generator=gen)):
        bs = labels.shape[0]
        eng = engine
        if bs != batch_size:                                       # the ragged last batch
            if tail_engine is None:
                tail_engine = getattr(engine, "_tail_engines", {}).get(bs)
            if tail_engine is None:
                tail_engine = engine.sibling(bs, trains=train)
                if train:                                          # (an evaluating sibling is not kept as the training one)
                    engine.__dict__.setdefault("_tail_engines", {})[bs] = tail_engine
            eng = tail_engine
            eng.pack()                                             # its packed copies missed every step since its last use
            image, audio, labels = image.contiguous(), audio.contiguous(), labels.contiguous()
        if train:
            if replay is not None and eng is engine:
                replay(image, audio, labels)
            else:
                eng.train_step(image, audio, labels, grad_sync=grad_sync)
        else:
            eng.evaluate(image, audio, labels)
        if eng is not engine and train:
            engine.pack()                                          # the tail step changed the weights
        acc[:4] += eng.losses                                      # device-side: no host round trip
        acc[4:8] += eng.losses * bs
        acc[8:11] += (eng.preds == labels.to(eng.preds.dtype)[None, :]).sum(dim=1)
        seen += bs
        if (i + 1) % log_interval_steps == 0 or i + 1 == nb:
            host = acc.cpu().numpy()                               # the only sync of the interval
            if log is not None:
                log({"split": split, "step": i + 1, "loss": host[3] / (i + 1), "acc": host[10] / seen})
    steps, n = max(nb, 1), max(seen, 1)
    a, b = getattr(engine, "MODS", ("image", "audio"))
    return {"loss": float(host[7]) / n, "loss_step_mean": float(host[3]) / steps,
            f"loss_{a}": float(host[0]) / steps, f"loss_{b}": float(host[1]) / steps, "loss_fusion": float(host[2]) / steps,
            "acc": float(host[10]) / n, f"acc_{a}": float(host[8]) / n, f"acc_{b}": float(host[9]) / n,
            "hits": int(round(host[10])), f"hits_{a}": int(round(host[8])), f"hits_{b}": int(round(host[9])),
            "steps": nb, "samples": seen}
