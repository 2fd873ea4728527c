"""The callers of the hot path: the reference's multi-loss task modules, Lightning-free.

    AVMnistMixerMultiLoss   models/avmnist.py:165-312, :413-422
    MimicMixerMultiLoss     models/mimic.py:24-142
    MMIMDBMixerMultiLoss    models/mmimdb.py:22-147

Same constructor contract (`model_cfg`, `optimizer_cfg` with the keys of cfg/*/*.yml -- plain dicts or anything with
`.get` / attribute access), same sub-module names (hence the same state-dict keys as the published checkpoints), same
`shared_step(batch, mode=...)` result dict, same `configure_optimizers()` (Adam + ReduceLROnPlateau on `val_loss`).
What is NOT here is Lightning itself (trainer hooks, logging, metrics): a `pl.LightningModule` subclass can inherit from
these and add them, or drive `to_engine()` -- the fused, hipGraph-captured step over the same weights -- from its loop.
The towers are `m2_mixer_amd.modules` (HIP kernels under torch autograd); heads and losses are the same few torch ops the
reference uses.  SoftAdapt / GradBlend weighting are outside the scope of this build and refused loudly.
"""
from __future__ import annotations

import os
import pickle
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import modules


class _LenientPickle:
    """`pickle_module` for torch.load that survives a Lightning `.ckpt` written by the reference's environment: its
    `hyper_parameters` / callback states pickle classes of packages this build does not need (omegaconf, pytorch_lightning).
    Unknown classes become inert placeholders; the tensors under `state_dict` load normally."""
    __name__ = "m2_mixer_amd.models._LenientPickle"

    class _Placeholder:
        def __init__(self, *a, **k):
            pass

        def __setstate__(self, state):
            pass

        def __call__(self, *a, **k):
            return self

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                return type(name, (_LenientPickle._Placeholder,), {"__module__": module})

    load = staticmethod(pickle.load)
    dumps = staticmethod(pickle.dumps)
    dump = staticmethod(pickle.dump)
    Pickler = pickle.Pickler


#: the Lightning release the reference pins (requirements.txt:8); a PEP 440 string, because Lightning's checkpoint migration
#: parses this field with packaging.version.Version
LIGHTNING_VERSION = "1.8.6"


def _load_checkpoint_file(path, map_location, trusted: bool):
    """torch.load of a `.ckpt`.  First with `weights_only=True` (tensors and plain containers only: nothing in the file can
    run code) -- enough for checkpoints written by save_checkpoint.  A Lightning checkpoint written by the reference's
    environment pickles omegaconf / pytorch_lightning classes next to the `state_dict`; reading those needs the full
    unpickler, which executes whatever the file says -- only with `trusted=True`, i.e. for files the caller vouches for."""
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except Exception as e:
        if not trusted:
            raise RuntimeError(
                f"{path}: not loadable with weights_only=True ({type(e).__name__}: {str(e)[:200]}).  A checkpoint that "
                "pickles foreign classes (Lightning hyper-parameters, omegaconf) needs load_from_checkpoint(..., trusted=True); "
                "unpickling runs code from the file, so pass it only for checkpoints from a source you trust") from e
    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_LenientPickle)


class _Cfg(dict):
    """dict with attribute access, recursively (stands in for omegaconf.DictConfig)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in dict(d or {}).items():
            self[k] = _Cfg(v) if isinstance(v, dict) else v

    __getattr__ = dict.__getitem__


def _plain(c) -> dict:
    return {k: (_plain(v) if isinstance(v, dict) else v) for k, v in dict(c).items()}


class _MultiLossModule(nn.Module):
    MODS: Tuple[str, str] = ("a", "b")

    def __init__(self, model_cfg, optimizer_cfg, **kwargs):
        super().__init__()
        self.model_cfg = _Cfg(model_cfg if isinstance(model_cfg, dict) else dict(model_cfg))
        self.optimizer_cfg = dict(optimizer_cfg)
        self.scheduler_patience = self.optimizer_cfg.pop("scheduler_patience", 5)       # models/avmnist.py:171
        for key in ("use_softadapt", "use_gradblend"):
            if self.model_cfg.get(key, False):
                raise NotImplementedError(f"{key}: loss re-weighting schemes are outside this build's scope")
        self.mute = self.model_cfg.get("mute", None)
        self.freeze_modalities_on_epoch = self.model_cfg.get("freeze_modalities_on_epoch", None)
        self.random_modality_muting_on_freeze = self.model_cfg.get("random_modality_muting_on_freeze", False)
        self.muting_probs = self.model_cfg.get("muting_probs", None)
        self.fusion_loss_weight = self.model_cfg.get("fusion_loss_weight", 1.0 / 3)
        self.modalities_freezed = False
        self.current_epoch = 0                       # a trainer sets it (Lightning property in the reference)
        self.dropout = self.model_cfg.get("dropout", 0.0)

    # ---- pieces shared by the three tasks ----------------------------------------------------------------
    def _fusion_and_heads(self, n_a: int, n_b: int, dim_a: int, dim_b: int):
        m = self.model_cfg.modalities
        a, b = self.MODS
        self.fusion_function = modules.get_fusion_by_name(**m.multimodal)
        num_patches = self.fusion_function.get_output_shape(n_a, n_b, dim=1)
        self.fusion_mixer = modules.get_block_by_name(**m.multimodal, num_patches=num_patches, dropout=self.dropout)
        K = m.classification.num_classes
        setattr(self, f"classifier_{a}", nn.Linear(dim_a, K))
        setattr(self, f"classifier_{b}", nn.Linear(dim_b, K))
        self.classifier_fusion = modules.get_classifier_by_name(**m.classification)

    def _maybe_freeze_and_mute(self, mode: Optional[str]):
        """Epoch-triggered freezing / random muting, models/avmnist.py:243-251 (train mode only)."""
        if mode != "train":
            return None
        if self.freeze_modalities_on_epoch is not None and self.current_epoch == self.freeze_modalities_on_epoch \
                and not self.modalities_freezed:
            self._freeze_modalities()
        if self.random_modality_muting_on_freeze and self.freeze_modalities_on_epoch is not None \
                and self.current_epoch >= self.freeze_modalities_on_epoch:
            names = list(self.MODS) + ["multimodal"]
            self.mute = np.random.choice(names, p=[self.muting_probs[n] for n in names])
        return self.mute

    def _freeze_modalities(self):
        """models/avmnist.py:314-324: the two towers and their heads stop training, the fusion part continues."""
        a, b = self.MODS
        for name in (f"{a}_mixer", f"{b}_mixer", f"classifier_{a}", f"classifier_{b}", "static_extractor", "time_mixer"):
            mod = getattr(self, name, None)
            if mod is not None:
                for p in mod.parameters():
                    p.requires_grad = False
        self.modalities_freezed = True

    # ---- checkpoint I/O ------------------------------------------------------------
    checkpoint_path: Optional[str] = None
    #: the per-step outputs the reference's test_epoch_end concatenates and dumps (models/avmnist.py:382-398)
    TEST_PRED_KEYS: Tuple[str, ...] = ()

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, hparams_file=None, strict: bool = True, **kwargs):
        """Build the module from `model_cfg` / `optimizer_cfg` (passed as the reference's run.py:48-50 does) and load the
        weights of a Lightning `.ckpt` -- a torch.save'd dict whose `state_dict` uses exactly the sub-module names of this
        class (models/avmnist.py:400-411 remembers the path for the test_preds.pt dump; so does this).
        trusted=True allows the full unpickler for checkpoints that carry foreign pickled classes (see _load_checkpoint_file)."""
        if "model_cfg" not in kwargs or "optimizer_cfg" not in kwargs:
            raise TypeError("load_from_checkpoint needs model_cfg= and optimizer_cfg= (the reference passes both, run.py:48-50)")
        ckpt = _load_checkpoint_file(checkpoint_path, map_location or "cpu", kwargs.pop("trusted", False))
        state = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
        model = cls(kwargs.pop("model_cfg"), kwargs.pop("optimizer_cfg"), **kwargs)
        model.load_state_dict(state, strict=strict)
        model.checkpoint_path = str(checkpoint_path)
        model.current_epoch = int(ckpt.get("epoch", 0)) if isinstance(ckpt, dict) else 0
        return model

    def save_checkpoint(self, path, epoch: Optional[int] = None, global_step: int = 0, engine=None) -> str:
        """A `.ckpt` in Lightning's top-level layout (`state_dict`, `epoch`, `global_step`, `pytorch-lightning_version`,
        `optimizer_states`, `lr_schedulers`): what load_from_checkpoint -- this one or a LightningModule's -- reads.
        engine: a fused engine trained over these weights; its parameters are written instead of the module's and its Adam
        state goes to `optimizer_states[0]` in torch.optim.Adam's state_dict layout (parameter index = position in
        `parameters()` order), so training can resume (engine.load_optimizer_state_dict)."""
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        ckpt = {"state_dict": sd, "epoch": self.current_epoch if epoch is None else int(epoch), "global_step": int(global_step),
                "pytorch-lightning_version": LIGHTNING_VERSION, "optimizer_states": [], "lr_schedulers": []}
        if engine is not None:
            for k, v in engine.state_dict().items():
                sd[k] = v.detach().cpu()
            ckpt["optimizer_states"] = [engine.optimizer_state_dict()]
        torch.save(ckpt, path)
        self.checkpoint_path = str(path)
        return str(path)

    def save_test_preds(self, outputs: Sequence[Dict[str, torch.Tensor]], save_dir: Optional[str] = None) -> str:
        """test_epoch_end's dump (models/avmnist.py:382-398): the listed shared_step outputs of every test batch,
        concatenated, as `test_preds.pt` next to the checkpoint."""
        if save_dir is None:
            if self.checkpoint_path is None:
                raise RuntimeError("save_test_preds: no checkpoint path to save next to; pass save_dir")
            save_dir = os.path.dirname(self.checkpoint_path)
        os.makedirs(save_dir or ".", exist_ok=True)
        if not self.TEST_PRED_KEYS:
            raise NotImplementedError(f"{type(self).__name__}: the reference dumps no test predictions for this task")
        out = {k: torch.cat([o[k].detach().cpu() for o in outputs]) for k in self.TEST_PRED_KEYS}
        path = os.path.join(save_dir, "test_preds.pt")
        torch.save(out, path)
        return path

    def configure_optimizers(self) -> Dict[str, Any]:
        """models/avmnist.py:413-422."""
        from torch.optim.lr_scheduler import ReduceLROnPlateau
        optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, self.parameters()), **self.optimizer_cfg)
        return {"optimizer": optimizer, "lr_scheduler": ReduceLROnPlateau(optimizer, patience=self.scheduler_patience),
                "monitor": "val_loss"}

    def _engine_cfg(self) -> dict:
        m = _plain(self.model_cfg.modalities)
        cfg = {k: v for k, v in m.items() if k != "classification"}
        cfg["dropout"] = self.dropout
        cfg["num_classes"] = m["classification"]["num_classes"]
        return cfg

    def _engine_kwargs(self) -> dict:
        oc = self.optimizer_cfg
        return dict(lr=oc.get("lr", oc.get("learning_rate", 1e-3)), betas=tuple(oc.get("betas", (0.9, 0.999))),
                    eps=oc.get("eps", 1e-8), weight_decay=oc.get("weight_decay", 0.0))

    def to_engine(self, batch_size: int, precision: Optional[str] = None):
        """The fused training engine (engine.py) over a copy of this module's weights."""
        eng = self._make_engine(self._engine_cfg(), batch_size, next(self.parameters()).device, precision)
        eng.load_state_dict(self.state_dict())
        return eng


class AVMnistMixerMultiLoss(_MultiLossModule):
    """batch = {'image': (B,1,28,28), 'audio': (B,1,112,112), 'label': (B,)}"""

    MODS = ("image", "audio")

    TEST_PRED_KEYS = ("preds", "preds_image", "preds_audio", "labels", "image_logits", "audio_logits", "logits")

    def __init__(self, model_cfg, optimizer_cfg, **kwargs):
        super().__init__(model_cfg, optimizer_cfg, **kwargs)
        m = self.model_cfg.modalities
        self.image_mixer = modules.get_block_by_name(**m.image, dropout=self.dropout)
        self.audio_mixer = modules.get_block_by_name(**m.audio, dropout=self.dropout)
        self._fusion_and_heads(self.image_mixer.num_patch, self.audio_mixer.num_patch, m.image.hidden_dim, m.audio.hidden_dim)
        self.image_criterion = self.audio_criterion = self.fusion_criterion = nn.CrossEntropyLoss()

    def shared_step(self, batch, **kwargs):
        image, audio, labels = batch["image"], batch["audio"], batch["label"]
        mode = kwargs.get("mode", None)
        mute = self._maybe_freeze_and_mute(mode)
        if mode == "train" and mute == "image":
            image = torch.zeros_like(image)
        elif mode == "train" and mute == "audio":
            audio = torch.zeros_like(audio)
        image_tok = self.image_mixer(image)
        audio_tok = self.audio_mixer(audio)
        fused = self.fusion_mixer(self.fusion_function(image_tok, audio_tok))
        image_logits = self.classifier_image(image_tok.mean(dim=1))
        audio_logits = self.classifier_audio(audio_tok.mean(dim=1))
        logits = self.classifier_fusion(fused)
        loss_image = self.image_criterion(image_logits, labels)
        loss_audio = self.audio_criterion(audio_logits, labels)
        loss_fusion = self.fusion_criterion(logits, labels)
        ow = (1 - self.fusion_loss_weight) / 2
        loss = (self.fusion_loss_weight * loss_fusion + ow * loss_image + ow * loss_audio) * 3     # models/avmnist.py:289-290
        if self.modalities_freezed and mode == "train":
            loss = loss_fusion
        return {"preds": logits.argmax(dim=1), "preds_image": image_logits.argmax(dim=1),
                "preds_audio": audio_logits.argmax(dim=1), "labels": labels, "loss": loss, "loss_image": loss_image,
                "loss_audio": loss_audio, "loss_fusion": loss_fusion, "image_logits": image_logits,
                "audio_logits": audio_logits, "logits": logits}

    def _make_engine(self, cfg, batch_size, device, precision):
        from .engine import AVMnistEngine
        return AVMnistEngine(cfg, batch_size, device=device, precision=precision,
                             fusion_loss_weight=self.fusion_loss_weight, init=False, **self._engine_kwargs())


class MMIMDBMixerMultiLoss(_MultiLossModule):
    """batch = {'image': (B,3,160,256), 'text': (B,1,160,256), 'label': (B,23) multi-hot}"""

    MODS = ("image", "text")

    TEST_PRED_KEYS = ("preds", "preds_image", "preds_text", "labels", "image_logits", "text_logits", "logits")   # models/mmimdb.py:194-209

    def __init__(self, model_cfg, optimizer_cfg, **kwargs):
        super().__init__(model_cfg, optimizer_cfg, **kwargs)
        m = self.model_cfg.modalities
        self.image_mixer = modules.get_block_by_name(**m.image, dropout=self.dropout)
        self.text_mixer = modules.get_block_by_name(**m.text, dropout=self.dropout)
        self._fusion_and_heads(self.image_mixer.num_patch, self.text_mixer.num_patch, m.image.hidden_dim, m.text.hidden_dim)
        # three BCEWithLogitsLoss modules under the reference's names (models/mmimdb.py:47-50): their persistent `pos_weight`
        # buffers are part of every reference state_dict (`image_criterion.pos_weight`, ...), so checkpoints round-trip
        pos_weight = torch.tensor(list(self.model_cfg.pos_weight), dtype=torch.float32)
        self.image_criterion = nn.BCEWithLogitsLoss(pos_weight=pos_weight.clone())
        self.text_criterion = nn.BCEWithLogitsLoss(pos_weight=pos_weight.clone())
        self.fusion_criterion = nn.BCEWithLogitsLoss(pos_weight=pos_weight.clone())

    def shared_step(self, batch, **kwargs):
        image, text, labels = batch["image"], batch["text"], batch["label"]
        mode = kwargs.get("mode", None)
        mute = self._maybe_freeze_and_mute(mode)
        if mode == "train" and mute == "image":
            image = torch.zeros_like(image)
        elif mode == "train" and mute == "text":
            text = torch.zeros_like(text)
        image_tok = self.image_mixer(image)
        text_tok = self.text_mixer(text)
        fused = self.fusion_mixer(self.fusion_function(image_tok, text_tok))
        image_logits = self.classifier_image(image_tok.mean(dim=1))
        text_logits = self.classifier_text(text_tok.mean(dim=1))
        logits = self.classifier_fusion(fused)
        y = labels.float()
        loss_image, loss_text = self.image_criterion(image_logits, y), self.text_criterion(text_logits, y)
        loss_fusion = self.fusion_criterion(logits, y)
        loss = loss_image + loss_text + loss_fusion                                              # models/mmimdb.py:115-123
        if self.modalities_freezed and mode == "train":
            loss = loss_fusion
        return {"preds": (logits > 0).long(), "preds_image": (image_logits > 0).long(), "preds_text": (text_logits > 0).long(),
                "labels": labels, "loss": loss, "loss_image": loss_image, "loss_text": loss_text, "loss_fusion": loss_fusion,
                "image_logits": image_logits, "text_logits": text_logits, "logits": logits}

    def _engine_cfg(self):
        cfg = super()._engine_cfg()
        pw = [c.pos_weight for c in (self.image_criterion, self.text_criterion, self.fusion_criterion)]
        if not (torch.equal(pw[0], pw[1]) and torch.equal(pw[0], pw[2])):
            raise RuntimeError("the fused engine takes one pos_weight for all three heads (models/mmimdb.py:47-50 builds them equal)")
        cfg["pos_weight"] = pw[0].detach().cpu().tolist()          # the loaded buffers, not the cfg: a checkpoint may carry its own
        return cfg

    def _make_engine(self, cfg, batch_size, device, precision):
        from .engine import MMIMDBEngine
        return MMIMDBEngine(cfg, batch_size, device=device, precision=precision, init=False, **self._engine_kwargs())


class MimicMixerMultiLoss(_MultiLossModule):
    """batch = (static (B,5), time (B,24,12), labels (B,))"""

    MODS = ("static", "time")

    def __init__(self, model_cfg, optimizer_cfg, **kwargs):
        super().__init__(model_cfg, optimizer_cfg, **kwargs)
        m = self.model_cfg.modalities
        self.time_mixer = modules.get_block_by_name(**m.time, dropout=self.dropout)          # creation order: models/mimic.py:39-40
        self.static_extractor = modules.get_block_by_name(**m.static, dropout=self.dropout)
        self._fusion_and_heads(1, self.time_mixer.num_patch, m.static.output_dim, m.time.hidden_dim)
        self.criterion = nn.CrossEntropyLoss()

    def shared_step(self, batch, mode="train", **kwargs):
        static, time, labels = batch
        static_feat = self.static_extractor(static)
        time_tok = self.time_mixer(time)
        fused = self.fusion_mixer(self.fusion_function(static_feat.unsqueeze(1), time_tok))
        logits_static = self.classifier_static(static_feat)
        logits_time = self.classifier_time(time_tok.mean(1))
        logits = self.classifier_fusion(fused)
        loss_fusion = self.criterion(logits, labels)
        loss_static = self.criterion(logits_static, labels)
        loss_time = self.criterion(logits_time, labels)
        ow = (1 - self.fusion_loss_weight) / 2
        loss = self.fusion_loss_weight * loss_fusion + ow * loss_static + ow * loss_time          # models/mimic.py:115-121 (no x3)
        return {"preds": torch.softmax(logits, dim=1), "preds_static": torch.softmax(logits_static, dim=1),
                "preds_time": torch.softmax(logits_time, dim=1), "labels": labels.long(), "loss": loss,
                "loss_fusion": loss_fusion, "loss_static": loss_static, "loss_time": loss_time, "logits": logits,
                "logits_static": logits_static, "logits_time": logits_time}

    def _make_engine(self, cfg, batch_size, device, precision):
        from .engine import MimicEngine
        return MimicEngine(cfg, batch_size, device=device, precision=precision,
                           fusion_loss_weight=self.fusion_loss_weight, init=False, **self._engine_kwargs())
