"""Training harness around the hot loop (reference: explainn/selene/__init__.py, `Trainer`).

Same constructor arguments, same step order (selene/__init__.py:283-295: train mode, forward, loss,
zero_grad, backward, optimiser step, loss.item()), same validation / early stopping / best-model
checkpoint contents and the same `train.txt`, `validation.txt`, `selene.log` outputs (which
train.py:240-241 re-reads).  The difference is what runs inside a step: when the criterion is one
of the two losses `get_loss` returns and the model is `explainn_amd.ExplaiNN`, the whole
forward + loss + backward is ONE C-ABI call (`StepEngine`), gradients land in a flat buffer, and
with torch.distributed initialised that buffer is all-reduced once per step.
"""
import contextlib
import copy
import logging
import os
import sys
from time import time

import numpy as np
import torch

from .architectures import ExplaiNN
from .engine import StepEngine
from . import parallel

logger = logging.getLogger("selene")


class Trainer(object):
    """Trains a model for `max_steps` mini-batches, validating every
    `report_stats_every_n_steps` and stopping early after `patience` steps without improvement."""

    def __init__(self, model, data_loaders, loss_criterion, metrics, optimizer, max_steps=128000,
                 patience=32000, report_stats_every_n_steps=1000, output_dir="./",
                 cpu_n_threads=1, use_cuda=False, checkpoint_resume=None, freeze_top_n_filters=0,
                 logging_verbosity=2):
        self.model = model
        self.data_loaders = data_loaders
        self.criterion = loss_criterion
        self.metrics = metrics
        self.optimizer = optimizer
        self.max_steps = max_steps
        self.patience = patience
        self.nth_step_report_stats = report_stats_every_n_steps
        torch.set_num_threads(cpu_n_threads)
        os.makedirs(output_dir, exist_ok=True)
        self.output_dir = output_dir
        self.freeze_top_n_filters = freeze_top_n_filters
        self.logger = selene_logger(self.output_dir, logging_verbosity)
        self.use_cuda = use_cuda
        if self.use_cuda:
            self.model.cuda()
            self.criterion.cuda()
            self.logger.debug("Set modules to use CUDA")
        self._data_iterators = {k: [] for k in self.data_loaders.keys()}
        self._engine = None
        self._grad_sync = None
        self._init_train()
        self._init_validate()
        if checkpoint_resume is not None:
            self._load_checkpoint(checkpoint_resume)

    # ---- bookkeeping (selene/__init__.py:191-221) ----
    def _init_train(self):
        self._start_step = 1
        self._train_logger = metrics_logger("train", self.output_dir)
        self.logger.info("Training metrics: loss")
        self._train_logger.log(10, "loss")
        self._time_per_step = []
        self._train_loss = []

    def _init_validate(self):
        self._min_loss = float("inf")
        self._best_step = 1
        self._validation_logger = metrics_logger("validation", self.output_dir)
        names = ["loss"] + [x for x in self.metrics.keys()]
        self.logger.info("Validation metrics: " + ", ".join(names))
        self._validation_logger.log(10, "\t".join(names))

    def _load_checkpoint(self, checkpoint_resume):
        checkpoint = _load_checkpoint_file(checkpoint_resume)
        self.model.load_state_dict(checkpoint["state_dict"])
        self._start_step = checkpoint["step"]
        self._min_loss = checkpoint["min_loss"]
        self._best_step = checkpoint["step"]
        self.optimizer.load_state_dict(checkpoint["optimizer"])
        if self.use_cuda:
            for state in self.optimizer.state.values():
                for k, v in state.items():
                    # the step counter stays on the host (torch's Adam requires it there unless
                    # capturable; the reference's blanket .cuda() predates that check)
                    if isinstance(v, torch.Tensor) and k != "step":
                        state[k] = v.cuda()
        self.logger.info("Resuming from checkpoint: step %s, min loss %s" % (
            self._start_step, self._min_loss))

    def _get_batch(self, which_data):
        """Next mini-batch; the iterator restarts when exhausted (selene/__init__.py:223-246)."""
        t0 = time()
        try:
            batch = next(self._data_iterators[which_data])
        except (StopIteration, TypeError):
            self._data_iterators[which_data] = iter(self.data_loaders[which_data])
            batch = next(self._data_iterators[which_data])
        self.logger.debug("[BATCH] Time to sample batch: %s s." % (time() - t0))
        return batch

    # ---- the loop (selene/__init__.py:248-308) ----
    def train_and_validate(self):
        hook = None
        if self.freeze_top_n_filters > 0 and not self._fused_step_available():
            n = self.freeze_top_n_filters

            def _mask(grad):
                keep = torch.ones_like(grad)
                keep[:n] = 0
                return grad * keep
            hook = self.model.linears[0].weight.register_hook(_mask)
        for step in range(self._start_step, self.max_steps + 1):
            self.step = step
            self.train()
            if self.step % self.nth_step_report_stats == 0:
                self.validate()
            if self.step >= self._best_step + self.patience:
                self.logger.info("Early stopping: stop training!")
                break
        if hook is not None:
            hook.remove()
        if self.use_cuda and isinstance(self.model, ExplaiNN) and self.model.validate_input:
            self._check_input_flags()
        self.logger.handlers.clear()
        self._train_logger.handlers.clear()
        self._validation_logger.handlers.clear()

    def _fused_step_available(self):
        return (isinstance(self.model, ExplaiNN) and self.use_cuda and
                type(self.criterion) in (torch.nn.BCEWithLogitsLoss, torch.nn.MSELoss) and
                getattr(self.criterion, "reduction", "mean") == "mean" and
                getattr(self.criterion, "weight", None) is None and
                getattr(self.criterion, "pos_weight", None) is None)

    def _fused_engine(self, batch):
        if self._engine is None:
            kind = "binary" if isinstance(self.criterion, torch.nn.BCEWithLogitsLoss) else "linear"
            self._engine = StepEngine(self.model, batch, loss=kind)
            if parallel.world() > 1:
                parallel.broadcast_parameters(self.model)
                self._grad_sync = parallel.GradAllReduce(self._engine.flat_grad,
                                                         split=self._engine.conv_grad_elements)
        return self._engine

    def train(self):
        """One optimisation step on one mini-batch; returns nothing, logs like the reference."""
        t_i = time()
        if not self.model.training:          # selene/__init__.py:283 (a 17-module walk when repeated)
            self.model.train()
        inputs, targets = self._get_batch("train")
        if self.use_cuda:
            inputs = inputs.cuda()
            targets = targets.cuda()
        if parallel.world() > 1:
            inputs, targets = parallel.shard_batch(inputs, targets)
        if self._fused_step_available():
            eng = self._fused_engine(inputs.shape[0])
            # parameters are re-pointed only outside the loop (train.py:318-324 re-assigns the filter
            # bank before training; the optimiser updates in place): rebuild the pointer table when
            # the first parameter moved, and every 64th step as a safety net
            w0 = self.model.linears[0].weight
            if self.step % 64 == 1 or getattr(self, "_w0_ptr", None) != w0.data_ptr():
                eng.refresh_params()
                self._w0_ptr = w0.data_ptr()
            if inputs.dtype != torch.uint8:           # base codes (loader.CodesLoader) go in as they are
                inputs = inputs.float()
            _, loss = eng.step(inputs.contiguous(), targets.float().contiguous(),
                               seed=int(torch.randint(0, 2 ** 62, (1,)).item()),
                               freeze_top_n_filters=self.freeze_top_n_filters)
            if self._grad_sync is not None:
                self._grad_sync()       # one all-reduce of the flat buffer
            eng.attach_grads()
            self.optimizer.step()
            # The reference reads loss.item() here, stalling the host on the device every step; the
            # values are only consumed when the running mean is reported, so the fused path keeps
            # them on the device until then (or until 256 have piled up) and the host runs ahead.
            loss_value = loss.clone()
            # the device-side input validation flag is sticky: reading it (a second device-to-host
            # round trip) on the first steps and then every 64th reports a non-one-hot input without
            # paying for the check on every step; train_and_validate() reads it once more at the end
            if self.model.validate_input and (self.step <= 3 or self.step % 64 == 0):
                self._check_input_flags()
        else:
            if parallel.world() > 1 and self.model.grad_sync is None:
                # the autograd path of a sharded run (weighted / custom criterion): same start and
                # the same averaged gradients as the fused path, one all-reduce per backward
                parallel.broadcast_parameters(self.model)
                self.model.grad_sync = parallel.average_gradients
            predictions = self.model(inputs)
            loss = self.criterion(predictions, targets)
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
            loss_value = loss.item()
        self._train_loss.append(loss_value)
        if len(self._train_loss) % 256 == 0:
            self._settle_train_loss()
        self._time_per_step.append(time() - t_i)
        if self.step > 0 and self.step % self.nth_step_report_stats == 0:
            self._settle_train_loss()
            self.logger.info("[STEP %d] average number of steps per second: %s" % (
                self.step, 1. / np.average(self._time_per_step)))
            self.logger.info("Training loss: %s" % np.average(self._train_loss))
            self._train_logger.log(10, np.average(self._train_loss))
            self._time_per_step = []
            self._train_loss = []

    def _settle_train_loss(self):
        """Turn the loss values still held as device tensors into floats (one transfer)."""
        pending = [i for i, v in enumerate(self._train_loss) if torch.is_tensor(v)]
        if pending:
            vals = torch.cat([self._train_loss[i].reshape(1) for i in pending]).tolist()
            for i, v in zip(pending, vals):
                self._train_loss[i] = v

    def _check_input_flags(self):
        if self.model.input_flags() & 1:
            raise ValueError("input is not one-hot (see explainn_amd.ExplaiNN.forward)")

    def _evaluate_on_data(self, which_data):
        """Average loss + all predictions/targets over a data set (selene/__init__.py:310-342)."""
        self.model.eval()
        batch_losses, all_predictions, all_targets = [], [], []
        cache = self.model.eval_cache() if isinstance(self.model, ExplaiNN) else contextlib.nullcontext()
        with cache:
            out = self._evaluate_batches(which_data, batch_losses, all_predictions, all_targets)
        if self.use_cuda and isinstance(self.model, ExplaiNN) and self.model.validate_input:
            self._check_input_flags()
        return out

    def _evaluate_batches(self, which_data, batch_losses, all_predictions, all_targets):
        for inputs, targets in iter(self.data_loaders[which_data]):
            # a device-resident loader (loader.CodesLoader) yields views of a staging ring: cloned
            # on the device and brought to the host ONCE after the loop; host batches are kept as is
            all_targets.append(targets.data.clone() if targets.is_cuda else targets.data)
            if self.use_cuda:
                inputs = inputs.cuda()
                targets = targets.cuda()
            with torch.no_grad():
                predictions = self.model(inputs)
                # kept on the device; the reference's per-batch .cpu()/.item() reads are done once,
                # after the loop
                batch_losses.append(self.criterion(predictions, targets).reshape(1))
                all_predictions.append(predictions.data)
        losses = torch.cat(batch_losses).tolist() if batch_losses else []
        preds = torch.cat(all_predictions).cpu().numpy() if all_predictions else np.zeros((0, 0))
        tgts = torch.cat(all_targets).cpu().numpy() if all_targets else np.zeros((0, 0))
        return (np.average(losses), preds, tgts)

    def validate(self):
        """Validation metrics on flattened predictions; best model -> best_model.pth.tar
        (selene/__init__.py:344-391)."""
        validation_loss, all_predictions, all_targets = self._evaluate_on_data("validation")
        self.logger.info("Validation loss: %s" % validation_loss)
        valid_scores = {}
        for metric in self.metrics:
            score = self.metrics[metric](all_targets.flatten(), all_predictions.flatten())
            valid_scores[metric] = score if isinstance(score, float) else score[0]
        for name, score in valid_scores.items():
            self.logger.info("Validation %s: %s" % (name, score))
        self._validation_metrics = valid_scores
        if validation_loss < self._min_loss:
            self._min_loss = validation_loss
            self._best_step = int(self.step)
            if parallel.rank() == 0:
                model = copy.deepcopy(self.model)
                self._save_checkpoint({
                    "step": self._best_step,
                    "arch": model.__class__.__name__,
                    "options": model._options,
                    "state_dict": model.state_dict(),
                    "min_loss": self._min_loss,
                    "optimizer": copy.deepcopy(self.optimizer.state_dict()),
                })
                self.logger.info("Updating `best_model.pth.tar`")
        self._validation_logger.log(
            10, "\t".join(map(str, [validation_loss] + list(valid_scores.values()))))

    def _save_checkpoint(self, state):
        self.logger.debug("[TRAIN] %s: Saving model state to file." % state["step"])
        torch.save(state, os.path.join(self.output_dir, "best_model.pth.tar"))


def _load_checkpoint_file(path):
    """Checkpoints hold tensors plus python/numpy scalars (`min_loss` is a numpy float in files the
    reference writes); allow exactly those under the safe unpickler."""
    import numpy.core.multiarray as _ma  # noqa: F401
    allow = [np.dtype, np.float64, np.float32]
    try:
        allow.append(np._core.multiarray.scalar)
    except AttributeError:
        allow.append(np.core.multiarray.scalar)
    try:
        allow.append(type(np.dtype(np.float64)))
    except Exception:
        pass
    with torch.serialization.safe_globals(allow):
        return torch.load(path, map_location="cpu", weights_only=True)


def selene_logger(logger_path="./", verbosity=2):
    """File + stdout logger named "selene" (selene/__init__.py:430-470); idempotent."""
    log = logging.getLogger("selene")
    if len(log.handlers):
        return log
    log.setLevel({0: logging.WARNING, 1: logging.INFO}.get(verbosity, logging.DEBUG))
    fh = logging.FileHandler(os.path.join(logger_path, "selene.log"))
    fh.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(message)s"))
    log.addHandler(fh)
    sh = logging.StreamHandler(sys.stdout)
    sh.setFormatter(logging.Formatter("%(asctime)s - %(message)s"))
    sh.setLevel(logging.INFO)
    log.addHandler(sh)
    return log


def metrics_logger(metric, logger_path="./", verbosity=2):
    """`<metric>.txt` value logger (selene/__init__.py:472-507); idempotent."""
    log = logging.getLogger(metric)
    if len(log.handlers):
        return log
    log.setLevel({0: logging.WARNING, 1: logging.INFO}.get(verbosity, logging.DEBUG))
    fh = logging.FileHandler(os.path.join(logger_path, "%s.txt" % metric))
    fh.setFormatter(logging.Formatter("%(message)s"))
    log.addHandler(fh)
    sh = logging.StreamHandler(sys.stdout)
    sh.setFormatter(logging.Formatter("%(asctime)s - %(message)s"))
    sh.setLevel(logging.INFO)
    log.addHandler(sh)
    return log


def freeze_top_n_filters(grad, freeze_top_n_filters=0, use_cuda=False):
    """Gradient hook of selene/__init__.py:509-515: zero the first n filters' gradient."""
    keep = torch.ones_like(grad)
    keep[:freeze_top_n_filters] = 0
    return grad * keep
