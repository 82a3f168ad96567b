"""The arithmetic of the reads' bit planes (ebwt2snp_amd/csrc/e2s_tail.hip: k_reads_pack2, the plane path of k_consensus_pair; DESIGN.md
section 3), restated with integers and checked against the bytes it stands for -- no GPU: the multiply that packs four 0/1 bytes into
four bits, a row's planes from two groups by a funnel shift, the letters back from the codes, and a row's mismatches against a consensus
string from the planes alone."""
import random

import numpy as np

A, C, G, T = 65, 67, 71, 84
LETTER = (A, C, T, G)                          # by code (ch >> 1) & 3


def pack(text):
    """two 32-bit planes per 32 characters; a character outside "ACGT" has code 0 (and would flag its read)"""
    n = (len(text) + 31) // 32
    lo, hi = [0] * (n + 2), [0] * (n + 2)
    for i, ch in enumerate(text):
        code = (ch >> 1) & 3 if ch in (A, C, G, T) else 0
        lo[i >> 5] |= (code & 1) << (i & 31)
        hi[i >> 5] |= (code >> 1) << (i & 31)
    return lo, hi


def alignbit(hi, lo, sh):                      # v_alignbit_b32
    return (((hi << 32) | lo) >> (sh & 31)) & 0xFFFFFFFF


def row_planes(lo, hi, base, have):
    g, sh = base >> 5, base & 31
    vm = 0xFFFFFFFF if have >= 32 else (1 << have) - 1
    return alignbit(lo[g + 1], lo[g], sh) & vm, alignbit(hi[g + 1], hi[g], sh) & vm


def test_four_flags_to_four_bits():
    """((c & 0x01010101) * 0x10204080) >> 28: byte i's low bit lands in bit i"""
    for bits in range(16):
        c = sum(((bits >> i) & 1) << (8 * i) for i in range(4))
        assert ((c * 0x10204080) & 0xFFFFFFFF) >> 28 == bits


def test_rows_from_planes_are_the_bytes():
    rnd = random.Random(3)
    text = [rnd.choice([A, C, G, T]) for _ in range(4000)]
    lo, hi = pack(text)
    for _ in range(20000):
        have = rnd.choice([0, 1, 5, 17, 30, 31, 32])
        base = rnd.randrange(0, len(text) - have + 1)
        rl, rh = row_planes(lo, hi, base, have)
        for i in range(32):
            code = ((rl >> i) & 1) | (((rh >> i) & 1) << 1)
            want = text[base + i] if i < have else 0
            assert (LETTER[code] if i < have else 0) == want and (i < have or code == 0)


def test_mismatches_from_planes():
    """popc((((row.lo ^ c.lo) | (row.hi ^ c.hi)) & have) | ~have) & columns) = the bytes that differ from the consensus (a letter in every
    column; behind the row's end its character is 0)"""
    rnd = random.Random(5)
    for _ in range(20000):
        kl = rnd.choice([1, 9, 24, 31, 32])
        have = rnd.randrange(0, kl + 1)
        row = [rnd.choice([A, C, G, T]) if i < have else 0 for i in range(32)]
        cons = [rnd.choice([A, C, G, T]) if (i >= have or rnd.random() < 0.2) else row[i] for i in range(32)]
        rl = sum((((ch >> 1) & 1) << i) for i, ch in enumerate(row) if ch)
        rh = sum((((ch >> 2) & 1) << i) for i, ch in enumerate(row) if ch)
        cl = sum((((ch >> 1) & 1) << i) for i, ch in enumerate(cons[:kl]))
        chh = sum((((ch >> 2) & 1) << i) for i, ch in enumerate(cons[:kl]))
        vm = 0xFFFFFFFF if have >= 32 else (1 << have) - 1
        colmask = 0xFFFFFFFF if kl >= 32 else (1 << kl) - 1
        d = bin(((((rl ^ cl) | (rh ^ chh)) & vm) | (~vm & 0xFFFFFFFF)) & colmask).count("1")
        assert d == sum(1 for i in range(kl) if row[i] != cons[i])


def test_pack_matches_numpy_view():
    """the planes of a text, group by group, against a vectorised restatement"""
    rnd = np.random.RandomState(11)
    t = np.frombuffer(b"ACGT", dtype=np.uint8)[rnd.randint(0, 4, 1000)]
    lo, hi = pack([int(x) for x in t])
    code = (t >> 1) & 3
    for g in range(len(t) // 32):
        seg = code[32 * g:32 * g + 32].astype(np.uint64)
        assert lo[g] == int(((seg & 1) << np.arange(32, dtype=np.uint64)).sum()) and hi[g] == int(((seg >> 1) << np.arange(32, dtype=np.uint64)).sum())
