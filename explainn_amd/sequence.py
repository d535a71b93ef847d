"""Input encoding for the hot path (reference: explainn/sequence/__init__.py).

Same semantics -- rows A,C,G,T, any other letter an all-zero column, reverse complement =
flip of both axes -- but table-driven/vectorised instead of a per-character Python loop.
"""
import numpy as np

_LUT = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    _LUT[ord(_c)] = _i
    _LUT[ord(_c.lower())] = _i
_COMP = bytes.maketrans(b"ACGTacgtNn", b"TGCAtgcaNn")


def encode_codes(seq):
    """Base codes 0..3 (4 = anything else) of one sequence, uint8 (L,)."""
    return _LUT[np.frombuffer(seq.encode("ascii", "replace"), dtype=np.uint8)]


def encode_codes_many(seqs):
    """(N, L) uint8 base codes of equal-length sequences: the input format of
    architectures.BaseCodes (L bytes per sequence instead of the 32*L of a float64 one-hot)."""
    out = np.empty((len(seqs), len(seqs[0]) if len(seqs) else 0), dtype=np.uint8)
    for i, s in enumerate(seqs):
        c = encode_codes(s)
        if len(c) != out.shape[1]:
            raise ValueError("sequence %d has length %d, expected %d" % (i, len(c), out.shape[1]))
        out[i] = c
    return out


def rc_codes(codes):
    """Reverse complement on base codes (what rc_one_hot_encoding does to the one-hot): reverse,
    A<->T, C<->G, N stays N."""
    codes = np.asarray(codes)
    r = codes[..., ::-1]
    return np.where(r < 4, 3 - r, r).astype(np.uint8)


def codes_to_one_hot(codes, dtype=np.float32):
    """(N, L) codes -> (N, 4, L) one-hot (N columns all zero)."""
    codes = np.asarray(codes)
    return (codes[:, None, :] == np.arange(4, dtype=np.uint8)[None, :, None]).astype(dtype)


def one_hot_encode(seq):
    """sequence/__init__.py:8-28 -> float64 (4, L)."""
    codes = encode_codes(seq)
    out = np.zeros((4, len(codes)), dtype=float)
    ok = codes < 4
    out[codes[ok], np.nonzero(ok)[0]] = 1.0
    return out


def one_hot_encode_many(seqs):
    """sequence/__init__.py:4-6 -> (N, 4, L)."""
    return np.array([one_hot_encode(s) for s in seqs])


def one_hot_decode(encoded_seq):
    """sequence/__init__.py:34-47: columns with exactly one 1 -> letter, anything else -> N."""
    enc = np.asarray(encoded_seq)
    idx = enc.argmax(axis=0)
    ok = (enc == 1).sum(axis=0) == 1
    letters = np.array(list("ACGT"))[idx]
    return "".join(np.where(ok, letters, "N"))


def one_hot_decode_many(seqs):
    return np.array([one_hot_decode(s) for s in seqs])


def rc_one_hot_encoding(encoded_seq):
    """sequence/__init__.py:59-61."""
    return encoded_seq[::-1, ::-1]


def rc_one_hot_encoding_many(arr):
    """sequence/__init__.py:49-57."""
    return np.array([rc_one_hot_encoding(e) for e in arr])


def rc(seq):
    """sequence/__init__.py:67-69 (Bio.Seq.reverse_complement for the DNA alphabet)."""
    return seq.translate(_COMP)[::-1]


def rc_many(arr):
    return np.array([rc(s) for s in arr])
