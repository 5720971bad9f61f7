"""CPU: the RLE checker (oracle/rle_ref.py) against hand-derived known answers of pycocotools' rleToString / rleEncode
(cocoapi common/maskApi.c -- the codec behind ytvis_eval.py:282-286), then the product's host codec (openvis_amd/rle.py)
against the checker."""
import numpy as np
import pytest

from oracle import rle_ref
from openvis_amd import rle

# value (count, or difference to the count two places back) -> characters, derived by hand from rleToString:
#   c = x & 31; x >>= 5; more = (c & 16) ? x != -1 : x != 0; c |= more << 5; emit c + 48
SINGLE = [(0, "0"), (5, "5"), (15, "?"),          # one group, bit 4 clear
          (16, "`0"), (31, "o0"),                 # bit 4 set on a non-negative value: a second group must follow (else it reads as negative)
          (32, "P1"), (1000, "Xo0"),              # two / three groups
          (-1, "O"), (-2, "N"), (-16, "@"),       # negative, one group (sign = bit 4)
          (-17, "_O"), (-33, "oN")]               # negative with continuation


@pytest.mark.parametrize("value,text", SINGLE)
def test_rle_to_string_known_answers_single_values(value, text):
    # a value in position 0..2 is stored as is: only non-negative ones can stand alone; negative ones arise as differences (position 3)
    if value >= 0:
        assert rle_ref.rle_to_string([value]) == text
        assert rle_ref.rle_from_string(text) == [value]
    base = [7, 40, 9]
    counts = base + [40 + value]                               # position 3 stores counts[3] - counts[1] = value
    assert rle_ref.rle_to_string(counts) == "7X19" + text     # 40 = 8 | 1 << 5 -> 'X' '1'
    assert rle_ref.rle_from_string("7X19" + text) == counts


def test_rle_to_string_known_answer_sequence():
    # [3, 5, 2, 10, 1, 40] -> 3, 5, 2, 10-5 = 5, 1-2 = -1, 40-10 = 30 (= 0x1e: bit 4 set, non-negative -> "n0")
    assert rle_ref.rle_to_string([3, 5, 2, 10, 1, 40]) == "3525On0"
    assert rle_ref.rle_from_string("3525On0") == [3, 5, 2, 10, 1, 40]


def test_rle_encode_known_answers():
    # column-major, zeros first: a mask that starts with a 1 gets a leading zero-length run
    assert rle_ref.rle_encode([[0, 1], [1, 1]]) == [1, 3]
    assert rle_ref.rle_encode([[1, 0], [0, 0]]) == [0, 1, 3]
    assert rle_ref.rle_encode([[0, 0, 0]]) == [3]
    assert rle_ref.rle_encode([[1], [1]]) == [0, 2]
    m = [[0, 1, 0], [0, 1, 1]]                                   # columns: 00 | 11 | 01
    assert rle_ref.rle_encode(m) == [2, 2, 1, 1]
    assert rle_ref.rle_decode([2, 2, 1, 1], 2, 3) == m


def test_product_host_codec_matches_the_checker():
    rng = np.random.default_rng(5)
    for h, w, p in [(1, 1, 0.5), (7, 5, 0.5), (64, 48, 0.1), (33, 130, 0.9), (50, 50, 0.0), (50, 50, 1.0), (200, 300, 0.5)]:
        mask = rng.random((h, w)) < p
        ref = rle_ref.rle_encode(mask.tolist())
        assert rle.mask_to_counts(mask) == ref
        s = rle_ref.rle_to_string(ref)
        assert rle.counts_to_string(ref).decode("ascii") == s
        assert rle.string_to_counts(s) == ref == rle_ref.rle_from_string(s)
        assert (rle.counts_to_mask(ref, h, w) == mask).all()
    # long runs and large negative / positive differences (blob masks at 1080p give runs of several thousand)
    counts = [0, 100000, 3, 5, 70000, 1, 2, 65536, 31, 32, 1 << 20]
    assert rle.counts_to_string(counts).decode("ascii") == rle_ref.rle_to_string(counts)
    assert rle.string_to_counts(rle_ref.rle_to_string(counts)) == counts
    for value, text in SINGLE:
        assert rle.counts_to_string([7, 40, 9, 40 + value]).decode("ascii") == "7X19" + text


def test_vectorised_checker_equals_the_loop_checker():
    """oracle/rle_ref.rle_encode_np (used at 720p, where the loop takes too long) against rle_encode, the restatement of maskApi.c rleEncode"""
    rng = np.random.default_rng(11)
    for h, w, p in [(1, 1, 0.0), (1, 1, 1.0), (7, 5, 0.5), (64, 48, 0.1), (33, 130, 0.9), (50, 50, 0.0), (50, 50, 1.0), (120, 77, 0.5), (3, 200, 0.3)]:
        mask = rng.random((h, w)) < p
        assert rle_ref.rle_encode_np(mask) == rle_ref.rle_encode(mask.tolist())
    yy, xx = np.mgrid[:90, :160]
    blob = (yy - 40) ** 2 + (xx - 70) ** 2 < 900
    assert rle_ref.rle_encode_np(blob) == rle_ref.rle_encode(blob.tolist())
