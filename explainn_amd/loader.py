"""Input pipeline of the hot loop as base codes.

Replaces, for this path, `sequence.one_hot_encode_many` (sequence/__init__.py:4-28: a Python loop per
character producing a float64 (N,4,L) array), the reverse-complement augmentation of
train.py:275-278 (a second float64 copy of the data set) and the `DataLoader(TensorDataset(...))`
of train.py:286-295 (item-by-item batch assembly, pageable host memory):

  * `read_tsv_codes` / `read_fasta_codes`: file -> (N,L) uint8 base codes (0..3 = A,C,G,T, 4 =
    anything else) with ONE table lookup over the joined sequence text -- no per-sequence loop;
  * `CodesLoader`: batches of codes + targets cut with one index operation, staged in pinned
    host buffers and copied to the device asynchronously on a side stream one batch ahead;
    the reverse-complement half of an augmented data set is generated on the fly from the same
    bytes (`3 - code`, reversed), so nothing of size (2N,4,L) ever exists.

Order and contents of the batches are those of the reference's loader under the same torch RNG
state (the same `RandomSampler`/`BatchSampler` objects draw the indices), and the model consumes
codes bit-identically to the fp32 one-hot (tests/test_gpu_parity.py::test_base_codes_equal_onehot_path).
"""
import gzip

import numpy as np
import torch
from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler

from .sequence import _LUT


def _open(path, mode="rb"):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def codes_from_strings(seqs):
    """(N,L) uint8 codes of N equal-length sequences: the strings are joined once and mapped
    through a 256-entry table (sequence/__init__.py:19-26 semantics: ACGT in either case, anything
    else -> 4 = all-zero one-hot column)."""
    n = len(seqs)
    if n == 0:
        return np.zeros((0, 0), dtype=np.uint8)
    L = len(seqs[0])
    joined = "".join(seqs)
    if len(joined) != n * L:
        bad = next(i for i, s in enumerate(seqs) if len(s) != L)
        raise ValueError("sequence %d has length %d, expected %d" % (bad, len(seqs[bad]), L))
    raw = np.frombuffer(joined.encode("ascii", "replace"), dtype=np.uint8)
    return _LUT[raw].reshape(n, L)


def read_tsv_codes(tsv_file, debugging=False):
    """train.py:266-284 without the one-hot: headerless TSV `id <tab> sequence <tab> y0 [...]`
    -> (codes (N,L) uint8, labels (N,T) float32, ids).  debugging=True keeps the first 1000 rows:
    the reference's cut (train.py:280-282) comes AFTER its reverse-complement doubling, and item i
    of the doubled set is row i for i < N, so the first 1000 rows are all a debugging run can reach
    -- `CodesLoader(limit=1000)` applies the cut itself on the doubled range."""
    import pandas as pd
    df = pd.read_csv(tsv_file, sep="\t", header=None)
    if debugging:
        df = df.iloc[:1000]
    ids = df.pop(0).values
    codes = codes_from_strings(df.pop(1).tolist())
    labels = df.values.astype(np.float32)
    return codes, labels, ids


def read_fasta_codes(fasta_file):
    """predict.py:120-131 without Bio.SeqIO / the one-hot: FASTA (optionally gzipped, multi-line
    records) -> (codes (N,L) uint8, ids).  The file is split once on '>' ; each record's sequence
    lines are concatenated by deleting the newline bytes."""
    with _open(fasta_file, "rb") as fh:
        blob = fh.read()
    ids, seqs = [], []
    for rec in blob.split(b">")[1:]:
        head, _, body = rec.partition(b"\n")
        ids.append(head.split()[0].decode() if head.split() else "")
        seqs.append(body.translate(None, b"\r\n \t"))
    n = len(seqs)
    if n == 0:
        return np.zeros((0, 0), dtype=np.uint8), np.array(ids)
    L = len(seqs[0])
    joined = b"".join(seqs)
    if len(joined) != n * L:
        bad = next(i for i, s in enumerate(seqs) if len(s) != L)
        raise ValueError("record %d (%s) has length %d, expected %d" % (bad, ids[bad], len(seqs[bad]), L))
    return _LUT[np.frombuffer(joined, dtype=np.uint8)].reshape(n, L), np.array(ids)


def rc_codes_inplace(rows, flags):
    """Reverse-complement the rows of `rows` ((B,L) uint8, modified in place) where flags is set."""
    if flags.any():
        r = rows[flags][:, ::-1]
        rows[flags] = np.where(r < 4, 3 - r, r)
    return rows


class CodesLoader:
    """Mini-batches of (codes uint8 (B,L), targets float32 (B,T)) from host arrays.

    Stands where the reference has `DataLoader(TensorDataset(seqs, labels), batch_size, shuffle)`
    (train.py:286-295): `len()`, `.batch_size`, `.dataset` (anything with a length) and iteration
    are what `selene.Trainer` and `train.main` use.  reverse_complement=True makes the data set
    twice as long, item N+i being the reverse complement of item i with the same target
    (train.py:275-278), generated per batch.  With `device` set, batches arrive on that device:
    gathered into one of `depth` pinned staging buffers, copied on a side stream while the
    previous batch is being consumed, and handed over after a stream-wait (no host sync).
    `limit` truncates the LOGICAL data set after the reverse-complement doubling (train.py:280-282
    `--debugging`: `seqs[:1000]` of the doubled array).

    Device batches are views into a ring of `depth` staging slots: a yielded pair is valid until
    the next `next()` on the iterator (the slot is refilled two iterations later); a consumer
    that keeps batches must clone them."""

    def __init__(self, codes, labels, batch_size=100, shuffle=False, reverse_complement=False,
                 device=None, depth=3, limit=None):
        self.codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self.labels = np.ascontiguousarray(labels, dtype=np.float32)
        if self.labels.ndim == 1:
            self.labels = self.labels[:, None]
        if len(self.codes) != len(self.labels):
            raise ValueError("codes and labels differ in length")
        self.n_base = len(self.codes)
        self.rc = bool(reverse_complement)
        n = self.n_base * (2 if self.rc else 1)
        if limit is not None:
            n = min(n, int(limit))
        self.dataset = range(n)
        # train.py:297-302: never leave a last batch of one sample (BatchNorm raises on it)
        while batch_size > 1 and n % batch_size == 1:
            batch_size -= 1
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.device = torch.device(device) if device is not None else None
        self.depth = max(2, int(depth))
        self._slots = None
        self._stream = None

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def _sampler(self):
        # the same sampler objects torch's DataLoader would build: same draws from torch's RNG
        base = RandomSampler(self.dataset) if self.shuffle else SequentialSampler(self.dataset)
        return BatchSampler(base, self.batch_size, drop_last=False)

    def _gather(self, idx, out_codes, out_labels):
        idx = np.asarray(idx, dtype=np.int64)
        b = len(idx)
        if self.rc:
            flags = idx >= self.n_base
            src = np.where(flags, idx - self.n_base, idx)
        else:
            flags, src = None, idx
        np.take(self.codes, src, axis=0, out=out_codes[:b])
        np.take(self.labels, src, axis=0, out=out_labels[:b])
        if flags is not None:
            rc_codes_inplace(out_codes[:b], flags)
        return b

    def _host_iter(self):
        L, T = self.codes.shape[1], self.labels.shape[1]
        for idx in self._sampler():
            c = np.empty((len(idx), L), dtype=np.uint8)
            y = np.empty((len(idx), T), dtype=np.float32)
            self._gather(idx, c, y)
            yield torch.from_numpy(c), torch.from_numpy(y)

    def _make_slots(self):
        L, T, B = self.codes.shape[1], self.labels.shape[1], self.batch_size
        slots = []
        for _ in range(self.depth):
            hc = torch.empty((B, L), dtype=torch.uint8).pin_memory()
            hy = torch.empty((B, T), dtype=torch.float32).pin_memory()
            slots.append({"hc": hc, "hy": hy, "hc_np": hc.numpy(), "hy_np": hy.numpy(),
                          "dc": torch.empty((B, L), dtype=torch.uint8, device=self.device),
                          "dy": torch.empty((B, T), dtype=torch.float32, device=self.device),
                          "ready": torch.cuda.Event(), "free": torch.cuda.Event()})
        self._slots = slots
        self._stream = torch.cuda.Stream(device=self.device)

    def _stage(self, slot, idx):
        """Fill a slot's pinned buffers and enqueue its host-to-device copies on the side stream."""
        slot["free"].synchronize()             # the consumer of this slot's previous batch is done
        b = self._gather(idx, slot["hc_np"], slot["hy_np"])
        with torch.cuda.stream(self._stream):
            slot["dc"][:b].copy_(slot["hc"][:b], non_blocking=True)
            slot["dy"][:b].copy_(slot["hy"][:b], non_blocking=True)
            slot["ready"].record(self._stream)
        return b

    def _device_iter(self):
        if self._slots is None:
            self._make_slots()
        batches = iter(self._sampler())
        pending = []                            # (slot index, batch length), staged ahead
        k = 0
        # An earlier iterator may have been abandoned mid-epoch with a slot still being read by
        # work on the consumer's stream and no `free` event recorded for it: order every copy of
        # this iterator after whatever the consumer has enqueued so far.
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        for slot in self._slots:
            slot["free"] = torch.cuda.Event()    # an unrecorded event: synchronize() returns at once
        for _ in range(self.depth - 1):
            idx = next(batches, None)
            if idx is None:
                break
            pending.append((k % self.depth, self._stage(self._slots[k % self.depth], idx)))
            k += 1
        while pending:
            si, b = pending.pop(0)
            slot = self._slots[si]
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(slot["ready"])
            try:
                yield slot["dc"][:b], slot["dy"][:b]
            finally:
                # whatever the consumer enqueued on the current stream reads the slot: mark its end
                # (also when the generator is closed or collected instead of resumed)
                slot["free"].record(torch.cuda.current_stream(self.device))
            idx = next(batches, None)
            if idx is not None:
                pending.append((k % self.depth, self._stage(self._slots[k % self.depth], idx)))
                k += 1

    def __iter__(self):
        # torch's DataLoader draws one int64 from the global generator per iterator (its worker base
        # seed, drawn even without workers) before the sampler draws its own seed; the same draw
        # here keeps the shuffles of a run -- and everything seeded after them -- aligned with a
        # run over the reference's loader
        torch.empty((), dtype=torch.int64).random_()
        if self.device is not None and self.device.type == "cuda":
            return self._device_iter()
        return self._host_iter()
