"""Synthetic two-individual read sets and their exact EGSA.

Input generator for tests and bench.py, not part of the hot path.  Everything is
torch so the same code builds the 2x10^5-entry test cases on the CPU and the
2x10^8-entry bench workload on the GPU (packed-k-mer LSD radix sort; torch is
plumbing here: memory, sort, gather).

EGSA convention (what egsa / BCR_LCP_GSA emit and include.hpp:120-199 reads):
one entry per suffix of every read including the terminator-only suffix, sorted
lexicographically with '$' < A < C < G < T and terminators of different reads
ordered by read index; lcp[i] = common prefix of suffixes i-1 and i not counting
terminators (lcp[0] = 0); bwt = the character before the suffix ('$' for suff 0);
text = read index, suff = offset of the suffix in the read.
"""
import numpy as np
import torch

_CH = 21          # characters per packed int64 word (3 bits each)
_ASCII = torch.tensor([65, 67, 71, 84], dtype=torch.uint8)  # A C G T


def make_reads(n_reads_per_indiv, read_len, coverage=20, snp_every=300, indel_every=5000,
               seed=1, device="cpu", truth=None, error_rate=0.0):
    """Two individuals from one random genome; individual 2 carries SNPs and 3-bp deletions.

    Returns (reads uint8 [2R, L] of ASCII codes, nr_reads1 = R).  Reads 0..R-1 come from
    individual 1, R..2R-1 from individual 2; no reverse complements, no 'N'.
    truth: an optional dict that receives the planted ground truth (genome of individual 1 as ASCII,
    0-based SNP positions on it, REF / ALT letters) -- what snp_vs_vcf (snp_vs_vcf.cpp:108-573) is fed with.
    error_rate: every base of every read is replaced by one of the three other letters with this probability (sequencing
    errors: they split clusters and add single-read alleles -- many more clusters reach the per-cluster pass's pre-filter).
    """
    R, L = int(n_reads_per_indiv), int(read_len)
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    G = max(L + 64, (R * L) // coverage)
    genome = torch.randint(0, 4, (G,), generator=g, dtype=torch.int64)
    ind2 = genome.clone()
    snp_pos = torch.arange(snp_every // 2, G, snp_every)
    if snp_pos.numel():
        ind2[snp_pos] = (genome[snp_pos] + 1 + torch.randint(0, 3, (snp_pos.numel(),), generator=g)) % 4
    keep = torch.ones(G, dtype=torch.bool)
    for p in range(indel_every // 2, G - 3, indel_every):
        keep[p:p + 3] = False
    if truth is not None:
        truth.update(genome=_ASCII[genome].numpy(), snp_pos=snp_pos.numpy(), ref=_ASCII[genome[snp_pos]].numpy(),
                     alt=_ASCII[ind2[snp_pos]].numpy(), deleted=(~keep).numpy())
    ind2 = ind2[keep]
    g2 = torch.Generator(device="cpu")
    g2.manual_seed(seed + 1)
    outs = []
    for indiv in (genome, ind2):
        pos = torch.randint(0, indiv.numel() - L + 1, (R,), generator=g2)
        indiv_d = indiv.to(device)
        idx = pos.to(device)[:, None] + torch.arange(L, device=device)[None, :]
        outs.append(indiv_d[idx])
    codes = torch.cat(outs, 0)                       # [2R, L] in 0..3
    if error_rate > 0:
        g3 = torch.Generator(device=codes.device)
        g3.manual_seed(seed + 2)
        hit = torch.rand(codes.shape, generator=g3, device=codes.device) < float(error_rate)
        shift = torch.randint(1, 4, codes.shape, generator=g3, device=codes.device)
        codes = torch.where(hit, (codes + shift) % 4, codes)
        del hit, shift
    reads = _ASCII.to(device)[codes]
    return reads, R


def _pack_positions(codes1, n_words):
    """codes1: [R2, L] int64 in 1..4.  Returns Pq [R2, L + 1 + _CH*(n_words-1)] where
    Pq[r, q] packs the 21 characters starting at q (zero beyond the read)."""
    R2, L = codes1.shape
    Q = L + 1 + _CH * (n_words - 1)
    pad = torch.zeros((R2, Q + _CH), dtype=torch.int64, device=codes1.device)
    pad[:, :L] = codes1
    acc = torch.zeros((R2, Q), dtype=torch.int64, device=codes1.device)
    for i in range(_CH):
        acc = acc * 8 + pad[:, i:i + Q]
    return acc


def _floor_log2(x):
    """x > 0 int64 -> index of the highest set bit."""
    h = torch.zeros_like(x)
    for s in (32, 16, 8, 4, 2, 1):
        m = (x >> s) > 0
        x = torch.where(m, x >> s, x)
        h = h + s * m.to(torch.int64)
    return h


def build_egsa(reads, device=None, lengths=None):
    """Exact EGSA of a read set.

    reads: uint8 [R2, L] ASCII (ACGT only).  Returns dict of torch tensors in SA order:
    text int64, suff int64, lcp int64, bwt uint8, each of n = R2*(L+1) entries.
    lengths (optional, int64 [R2], 0 <= lengths[r] <= L): reads of different lengths -- read r is reads[r, :lengths[r]],
    the rest of its row is ignored; n = sum(lengths + 1).
    """
    if device is None:
        device = reads.device
    reads = reads.to(device)
    R2, L = reads.shape
    lut = torch.zeros(256, dtype=torch.int64, device=device)
    for c, v in ((65, 1), (67, 2), (71, 3), (84, 4)):
        lut[c] = v
    codes1 = lut[reads.long()]
    if lengths is not None:
        lengths = torch.as_tensor(lengths, dtype=torch.int64, device=device)
        codes1 = torch.where(torch.arange(L, device=device)[None, :] < lengths[:, None], codes1, torch.zeros_like(codes1))
    W = (L + _CH - 1) // _CH
    Pq = _pack_positions(codes1, W)
    del codes1
    if lengths is None:
        n = R2 * (L + 1)
        perm = torch.arange(n, device=device)
    else:
        # the suffixes that exist, in (read, offset) order: flat indices r * (L + 1) + p with p <= lengths[r]
        ok = torch.arange(L + 1, device=device)[None, :] <= lengths[:, None]
        perm = torch.nonzero(ok.reshape(-1))[:, 0]
        n = int(perm.numel())
    # LSD radix over the packed words; stable sorts keep ties in (read, offset) order, which is
    # the terminator order: identical suffixes of different reads sort by read index.
    for k in reversed(range(W)):
        key = Pq[:, _CH * k:_CH * k + L + 1].reshape(-1)[perm]
        order = torch.sort(key, stable=True).indices
        perm = perm[order]
        del key, order
    text = perm // (L + 1)
    suff = perm % (L + 1)
    # lcp: leading equal words, then leading equal 3-bit groups of the first differing word
    lcp = torch.zeros(n, dtype=torch.int64, device=device)
    alive = torch.ones(n - 1, dtype=torch.bool, device=device)
    for k in range(W):
        key = Pq[:, _CH * k:_CH * k + L + 1].reshape(-1)[perm]
        a, b = key[:-1], key[1:]
        x = a ^ b
        diff = x != 0
        part = torch.full_like(x, _CH)
        if diff.any():
            h = _floor_log2(torch.where(diff, x, torch.ones_like(x)))
            part = torch.where(diff, (_CH - 1) - h // 3, part)
        lcp[1:] += torch.where(alive, part, torch.zeros_like(part))
        alive = alive & ~diff
        del key, a, b, x, diff, part
    rem = (L if lengths is None else lengths[text]) - suff   # suffix length without terminator
    cap = torch.minimum(rem[:-1], rem[1:])
    lcp[1:] = torch.minimum(lcp[1:], cap)
    prev = torch.clamp(suff - 1, min=0)
    bwt = reads[text, prev]
    bwt = torch.where(suff == 0, torch.full_like(bwt, 36), bwt)   # '$'
    return dict(text=text, suff=suff, lcp=lcp, bwt=bwt)


def egsa_bruteforce(reads_list):
    """Tiny pure-Python EGSA for checking build_egsa (test helper)."""
    suf = []
    for r, s in enumerate(reads_list):
        for p in range(len(s) + 1):
            suf.append((s[p:], r, p))
    order = {"$": 0, "A": 1, "C": 2, "G": 3, "T": 4}
    suf.sort(key=lambda t: ([order[c] for c in t[0]] + [0], t[1]))
    text = [t[1] for t in suf]
    suff = [t[2] for t in suf]
    lcp = [0]
    for i in range(1, len(suf)):
        a, b = suf[i - 1][0], suf[i][0]
        l = 0
        while l < len(a) and l < len(b) and a[l] == b[l]:
            l += 1
        lcp.append(l)
    bwt = [ord(reads_list[t][p - 1]) if p > 0 else 36 for t, p in zip(text, suff)]
    return dict(text=text, suff=suff, lcp=lcp, bwt=bwt)


def synth_case(n_reads_per_indiv, read_len, seed=1, device="cpu", coverage=20, tail_lcp0=True, error_rate=0.0):
    """reads + EGSA for one synthetic configuration.  tail_lcp0 forces the last LCP to 0 so the
    reference's EOF phantom (hazard H1) cannot extend a cluster past the end."""
    reads, n1 = make_reads(n_reads_per_indiv, read_len, coverage=coverage, seed=seed, device=device, error_rate=error_rate)
    e = build_egsa(reads, device=device)
    if tail_lcp0:
        e["lcp"][-1] = 0
    return reads, n1, e


def ragged_lengths(n_reads, read_len, seed, k_left=31):
    """lengths of a trimmed read set (what real FASTA input looks like): most reads between 40 % and 100 % of read_len, one in
    sixteen shorter than the left context (-L) altogether, the last read the shortest of all."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + 77)
    lo = max(2, (2 * read_len) // 5)
    ln = torch.randint(lo, read_len + 1, (n_reads,), generator=g)
    short = torch.rand(n_reads, generator=g) < (1.0 / 16.0)
    ln = torch.where(short, torch.randint(4, max(5, k_left), (n_reads,), generator=g), ln)
    ln[-1] = 3
    return ln


def synth_case_ragged(n_reads_per_indiv, read_len, seed=1, device="cpu", coverage=20, tail_lcp0=True):
    """as synth_case with reads of different lengths.  Returns (list of bytes, nr_reads1, EGSA dict, lengths int64 [2R])."""
    reads, n1 = make_reads(n_reads_per_indiv, read_len, coverage=coverage, seed=seed, device=device)
    ln = ragged_lengths(reads.shape[0], read_len, seed)
    e = build_egsa(reads, device=device, lengths=ln)
    if tail_lcp0:
        e["lcp"][-1] = 0
    rn = reads.cpu().numpy()
    return [rn[r, :int(ln[r])].tobytes() for r in range(rn.shape[0])], n1, e, ln


def egsa_to_numpy(e, x=1, y=4, z=1):
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}
    return dict(text=e["text"].cpu().numpy().astype(dt[y]), suff=e["suff"].cpu().numpy().astype(dt[z]),
                lcp=e["lcp"].cpu().numpy().astype(dt[x]), bwt=e["bwt"].cpu().numpy().astype(np.uint8))
