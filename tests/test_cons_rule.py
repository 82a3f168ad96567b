"""The rule k_consensus_pair's counting form rests on (ebwt2snp_amd/csrc/e2s_tail.hip), checked against the
definition it replaces -- no GPU: `cons` (include.hpp:334-371) is sequential, the letter of a column changes when a count becomes STRICTLY
larger than the leader's; the kernel only counts and takes the winner from the four totals unless it says `redo` (then it runs the
sequential loop).  For every column of characters in {0, A, C, G, T} (0 = behind a read's end): not redo  =>  same letter."""
import itertools
import random

A, C, G, T = 65, 67, 71, 84


def b2i(ch):                                   # include.hpp:265-279 with N -> 0
    return {C: 1, G: 2, T: 3}.get(ch & 0xDF, 0)


def cons_sequential(col):
    cnt, letter, lead = [0, 0, 0, 0], A, 0
    for ch in col:
        b = b2i(ch)
        cnt[b] += 1
        if cnt[b] > cnt[lead]:
            letter, lead = ch, b
    return letter


def cons_from_totals(col):
    """what the kernel does in its counting form: (letter, redo)"""
    slot = lambda ch: (ch >> 1) & 3            # {0, A} 0, C 1, T 2, G 3
    tot = [0, 0, 0, 0]
    for ch in col:
        tot[slot(ch)] += 1
    m = max(tot)
    top = [s for s in range(4) if tot[s] == m]
    s0 = top[0]                                # the lowest slot that holds the largest total
    letter = (A, C, T, G)[s0]
    tie = len(top) > 1 and m != 0
    others = tot[1] + tot[2] + tot[3]
    never_lost = bool(col) and slot(col[0]) == 0 and max(tot[1:]) <= 1     # row 0 is class 0 and no other class occurs twice
    zero_col = 0 in col
    redo = bool(col) and (tie or (s0 == 0 and zero_col and others != 0 and not never_lost))
    return letter, redo


def check(col):
    letter, redo = cons_from_totals(col)
    assert redo or letter == cons_sequential(col), (col, letter, cons_sequential(col))
    return redo


def test_every_short_column():
    """all columns of up to 7 rows over {0, A, C, G, T}: 97 655 of them"""
    redone = total = 0
    for n in range(0, 8):
        for col in itertools.product((0, A, C, G, T), repeat=n):
            redone += check(list(col))
            total += 1
    assert total == sum(5 ** n for n in range(8)) and 0 < redone < total


def test_random_long_columns():
    """columns as lists of reads give them: one dominant letter, a few errors, some rows ended; up to 150 rows"""
    rnd = random.Random(7)
    redone = 0
    for _ in range(60000):
        n = rnd.choice([1, 2, 3, 5, 8, 13, 20, 20, 40, 150])
        main = rnd.choice([A, C, G, T])
        pe, pz = rnd.choice([0.0, 0.01, 0.05, 0.3]), rnd.choice([0.0, 0.0, 0.1, 0.5])
        col = [0 if rnd.random() < pz else (rnd.choice([A, C, G, T]) if rnd.random() < pe else main) for _ in range(n)]
        redone += check(col)
    assert redone < 20000                        # (the counting form decides most columns by itself)
