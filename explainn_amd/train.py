"""Training entry points (reference: explainn/train.py).

`_train` keeps the reference signature (train.py:304-307) and wiring: build `ExplaiNN`, loss,
metrics, Adam, optional transfer learning of the first filters, then `Trainer.train_and_validate`.
`main` is a thin argparse CLI with the reference's option names (the reference uses click +
click_option_group, which is not a dependency here).
"""
import argparse
import math
import os
import shutil
import time

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

from .architectures import ExplaiNN, get_loss, get_metrics, get_optimizer
from .selene import Trainer
from .sequence import one_hot_encode_many, rc_one_hot_encoding_many


def _get_seqs_labels_ids(tsv_file, debugging=False, reverse_complement=False):
    """train.py:266-284: headerless TSV `id <tab> sequence <tab> y0 [<tab> y1 ...]`."""
    import pandas as pd
    df = pd.read_csv(tsv_file, sep="\t", header=None)
    ids = df.pop(0).values
    seqs = one_hot_encode_many(df.pop(1).values)
    labels = df.values
    if reverse_complement:
        seqs = np.append(seqs, rc_one_hot_encoding_many(seqs), axis=0)
        labels = np.append(labels, labels, axis=0)
        ids = np.append(ids, ids, axis=0)
    if debugging:
        return seqs[:1000], labels[:1000], ids[:1000]
    return seqs, labels, ids


def _avoid_single_sample_batch(n, batch_size):
    """train.py:297-302: shrink the batch until the last batch is not a single sample
    (BatchNorm raises on one value per channel in train mode)."""
    while batch_size > 1 and n % batch_size == 1:
        batch_size -= 1
    return batch_size


class _BatchedTensorDataset(TensorDataset):
    """TensorDataset whose batches are cut with one index op per tensor (`__getitems__`, which
    torch's DataLoader prefers when present) instead of batch_size `__getitem__` calls plus a
    collate: 0.02 ms instead of 0.25 ms per batch of 100 -- at the reference's default shape the
    stock loader cost more host time than the whole training step takes on the GPU."""

    def __getitems__(self, indices):
        idx = torch.as_tensor(indices)
        return tuple(t[idx] for t in self.tensors)


def _already_batched(batch):
    return batch


def _get_data_loader(seqs, labels, batch_size=100, shuffle=False):
    """train.py:286-295: a torch DataLoader over (sequences, labels) -- same sampler, same batch
    order and contents as the reference's, cut batch-wise (see _BatchedTensorDataset)."""
    dataset = _BatchedTensorDataset(torch.Tensor(seqs), torch.Tensor(labels))
    return DataLoader(dataset, _avoid_single_sample_batch(len(dataset), batch_size),
                      shuffle=shuffle, collate_fn=_already_batched)


def _train(sequence_length, n_features, data_loaders, input_data, steps_per_epoch, cnn_units=100,
           kernel_size=19, lr=0.003, max_epochs=100, patience=10, cpu_threads=1, output_dir="./",
           filter_weights=[], freeze=False, checkpoint_resume=None):
    """train.py:304-342."""
    freeze_top_n_filters = 0
    exp_model = ExplaiNN(cnn_units, kernel_size, sequence_length, n_features)
    loss_criterion = get_loss(input_data=input_data)
    metrics = get_metrics(input_data=input_data)
    optimizer = get_optimizer(exp_model.parameters(), lr)
    if len(filter_weights) > 0:                      # transfer learning (train.py:316-324)
        w = exp_model.linears[0].weight.data
        for i in range(min(w.shape[0], len(filter_weights))):
            w[i] = torch.as_tensor(filter_weights[i], dtype=w.dtype)
            if freeze:
                freeze_top_n_filters += 1
    trainer = Trainer(
        exp_model, data_loaders, loss_criterion, metrics, optimizer,
        max_steps=steps_per_epoch * max_epochs, patience=steps_per_epoch * patience,
        report_stats_every_n_steps=steps_per_epoch, output_dir=output_dir,
        cpu_n_threads=cpu_threads, use_cuda=torch.cuda.is_available(),
        checkpoint_resume=checkpoint_resume, freeze_top_n_filters=freeze_top_n_filters)
    trainer.train_and_validate()
    return trainer


def main(argv=None):
    """train.py:151-264: load TSVs, pick the best of `--initialize` short runs, train."""
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("training_file"); ap.add_argument("validation_file")
    ap.add_argument("-b", "--batch-size", type=int, default=100)
    ap.add_argument("-c", "--cpu-threads", type=int, default=1)
    ap.add_argument("-d", "--debugging", action="store_true")
    ap.add_argument("-i", "--initialize", type=int, default=1)
    ap.add_argument("-o", "--output-dir", default="./")
    ap.add_argument("-r", "--rev-complement", action="store_true")
    ap.add_argument("-t", "--time-me", action="store_true")
    ap.add_argument("--cnn-units", type=int, default=100)
    ap.add_argument("--kernel-size", type=int, default=19)
    ap.add_argument("--lr", type=float, default=0.003)
    ap.add_argument("--max-epochs", type=int, default=100)
    ap.add_argument("--patience", type=int, default=10)
    args = ap.parse_args(argv)
    import pandas as pd
    start = time.time()
    os.makedirs(args.output_dir, exist_ok=True)
    s_tr, l_tr, _ = _get_seqs_labels_ids(args.training_file, args.debugging, args.rev_complement)
    s_va, l_va, _ = _get_seqs_labels_ids(args.validation_file, args.debugging, args.rev_complement)
    loaders = {"train": _get_data_loader(s_tr, l_tr, args.batch_size, shuffle=True),
               "validation": _get_data_loader(s_va, l_va, args.batch_size, shuffle=True)}
    L, T = s_tr[0].shape[1], l_tr[0].shape[0]
    input_data = "binary" if np.unique(l_tr[:, 0]).size == 2 else "linear"
    spe = math.ceil(len(loaders["train"].dataset) / loaders["train"].batch_size)
    best_loss, best_model = None, None
    for i in range(args.initialize):
        d = os.path.join(args.output_dir, "init.%d" % i)
        if not os.path.isdir(d):
            os.makedirs(d)
            _train(L, T, loaders, input_data, spe, args.cnn_units, args.kernel_size, args.lr, 5,
                   args.patience, args.cpu_threads, d)
        loss = pd.read_csv(os.path.join(d, "validation.txt"), sep="\t").loss.min()
        if best_model is None or loss < best_loss:
            best_loss, best_model = loss, os.path.join(d, "best_model.pth.tar")
    shutil.copy(best_model, args.output_dir)
    _train(L, T, loaders, input_data, spe, args.cnn_units, args.kernel_size, args.lr,
           args.max_epochs, args.patience, args.cpu_threads, args.output_dir,
           checkpoint_resume=best_model)
    if args.time_me:
        with open(os.path.join(args.output_dir, "time-train.py.txt"), "wt") as fh:
            fh.write("%.2f seconds" % (time.time() - start))


if __name__ == "__main__":
    main()
