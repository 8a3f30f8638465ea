#!/usr/bin/env python3
"""Per-read and per-cluster known-answer tables for twelve hazard fixtures, derived WITHOUT either oracle.

The reference holds no vectors, so what pins the two restatements of detect.rs under
oracle/ is what this repository derives itself.  kat_survey was derived by hand in the survey.  This script is the
pencil-and-paper method of that derivation written down once, for fixtures too long to do by eye without slips:

  * it shares no code with oracle/detect_oracle.py or oracle/detect_oracle.cpp (nothing is imported from oracle/);
  * it does not walk the reference's loops: the two flanks are cut out of the FASTA as whole slices
    (detect.rs:116-131 as interval arithmetic: [pos-1, pos-1+L) or [pos+A-1-L, pos+A-1) reversed and complemented,
    positions outside the contig -> 'N'), the mismatch vectors ML / MR are compared position by position, and
    mm[k] is the CLOSED FORM of SURVEY.md section 3.3 item 4 -- |ML below k| + |MR from k on| -- not the
    reference's running sum (detect.rs:147-153); bp = lowest k of the minimum over k in [A, L-A-1];
  * one read (hz_contig_ends read 0) was also derived entirely by hand from detect.rs:105-187 and the FASTA text
    (DESIGN.md section 2 shows the work); the script must reproduce that row, see HAND_ROW below.

  * clusters are not found by walking the reference's two loops (detect.rs:203-227) either, but from their fixed point
    written as set comprehensions (derive_clusters below): compat(p, s) = every test of :214-223 as one predicate;
    because the evidence is sorted by (chr, pos), the two `break` tests are monotone and a seed's cluster is simply
    {p} + {s after p : compat(p, s)} -- reported or not, which is the reference's quirk (:225 pushes without looking at
    reported[s]); the seeds are the reads no EARLIER SEED is compatible with; pairing (detect.rs:79-87) is "a 3p line
    counts iff the nearest 5p line before it carries the same read number".

Output: tests/golden/hand_tables.json -- literal numbers and strings per read (input order of the paired reads):
swapped, mm[A..L-A-1], bp, lpos, rpos, junction, signature; and per fixture the printed clusters: seed, members in
sweep order, and the reads that survive remove_frag_id_duplicates (detect.rs:247-287: groups by fragment id, score
min(l,r)*10000+max(l,r), first maximum) -- from which tests/test_hand_tables.py writes the expected .sv text.
It checks BOTH restatements, the committed .sv files and the GPU against the committed tables.

Run:  python tests/golden/derive_tables.py        (rewrites hand_tables.json; the diff must be empty)
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = ["hz_tie_lowest_k", "hz_len_2a", "hz_contig_ends", "hz_ref_alphabet", "hz_dedup_min_evidence",
            # the clustering quirks (round 3): a read an earlier seed reported joins a later seed's cluster; reorientation
            # on one contig; coordinates that wrap below zero (printed and sorted as usize); unpaired lines; reads of
            # several lengths in one class; min_evidence 0 and 1 (singletons print)
            "hz_two_clusters", "hz_same_chr_swap", "hz_wrapped_coords", "hz_pairing", "hz_mixed_lengths",
            "hz_min_evidence_0", "hz_min_evidence_1"]

# bio 0.39 alphabets::dna::complement (Cargo.toml:14): identity except these letters, both cases
_A, _B = "AGCTYRWSKMDVHBN", "TCGARYWSMKHBDVN"
COMP = {ord(a): ord(b) for a, b in zip(_A + _A.lower(), _B + _B.lower())}

# derived by hand from the FASTA text and detect.rs:105-187; the script must agree
HAND_ROW = {"fixture": "hz_contig_ends", "read": 0, "swapped": False, "mm": [6, 5, 4, 3, 3, 2, 1, 0, 0, 1], "bp": 15,
            "lpos": 111, "rpos": 6, "junction": "GATGGATGAAGGCTT|CGTCGTAGTTAAAAGCTTTGAGTCC", "signature": "GAAGGCTT|CGTCGTAG"}


def rc(s: bytes) -> bytes:
    return bytes(COMP.get(c, c) for c in reversed(s))


def fasta(path):
    out, name = {}, None
    for line in open(path, "rb").read().split(b"\n"):
        if line.startswith(b">"):
            name = line[1:].split()[0]
            out[name] = b""
        elif name is not None:
            out[name] += line.strip()
    return out


def window(contig: bytes, start: int, L: int) -> bytes:
    """contig[start : start+L] with 'N' wherever the interval leaves the contig"""
    return bytes(contig[i] if 0 <= i < len(contig) else ord("N") for i in range(start, start + L))


def pairs(path):
    """(fragment id, sequence, (strand, chr, pos), (mstrand, mchr, mpos)) per 3p line that the pairing rule of
    detect.rs:79-87 accepts: the nearest 5p line before it carries the same read number (a 5p line without its 3p, a
    3p line after another read's 5p, or two 3p lines of one read after their 5p all follow from that)."""
    lines = [l for l in open(path, "rb").read().split(b"\n") if l]
    kind = [l[:2] for l in lines]
    num = [int(l.split(b":")[1]) for l in lines]
    assert all(k in (b"5p", b"3p") for k in kind)
    for i, l3 in enumerate(lines):
        before = [j for j in range(i) if kind[j] == b"5p"]
        if kind[i] != b"3p" or not before or num[before[-1]] != num[i]:
            continue
        h5, s5, c5, p5 = lines[before[-1]].split(b"\t")
        h3, s3, c3, p3 = l3.split(b"\t")
        _, _, frag, seq = h3.split(b":")[:4]
        yield frag, seq, (s5 == b"+", c5, int(p5)), (s3 == b"+", c3, int(p3))


def derive(name):
    A, F, E = (int(x) for x in open(os.path.join(HERE, name + ".args")).read().split())
    return derive_files(os.path.join(HERE, name + ".fa"), os.path.join(HERE, name + ".bowtie"), A, F, E)


def derive_files(fa_path, bowtie_path, A, F, E):
    """The tables of any (FASTA, Bowtie output, options) triple: the fixtures above, and the random cases of
    tests/test_hand_tables.py::test_independent_derivation_on_random_cases."""
    g = fasta(fa_path)
    rows = []
    for frag, seq, (st, ch, pos), (mst, mch, mpos) in pairs(bowtie_path):
        L = len(seq)
        swapped = ch > mch or (ch == mch and pos > mpos)                       # detect.rs:108
        if swapped:
            ch, mch, pos, mpos, st, mst, seq = mch, ch, mpos, pos, not mst, not st, rc(seq)
        left = window(g[ch], pos - 1, L) if st else rc(window(g[ch], pos + A - 1 - L, L))           # :116-122
        right = window(g[mch], mpos + A - 1 - L, L) if mst else rc(window(g[mch], mpos - 1, L))     # :124-131
        ML = [seq[k] != left[k] for k in range(L)]
        MR = [seq[k] != right[k] for k in range(L)]
        ks = [A] + list(range(A + 1, L - A))                                   # candidates: A itself (detect.rs:137-146), then :149
        mm = [sum(ML[:k]) + sum(MR[k:]) for k in ks]
        bp = ks[mm.index(min(mm))]                                             # lowest k of the minimum
        low = lambda c, m: c + 32 if m and 65 <= c <= 90 else c               # u8::to_ascii_lowercase
        junction = bytes(low(seq[k], ML[k]) for k in range(bp)) + b"|" + bytes(low(seq[k], MR[k]) for k in range(bp, L))
        signature = left[bp - 8:bp] + b"|" + right[bp:bp + 8]
        lpos = pos + bp - 1 if st else pos + A - bp                            # :172-181, usize arithmetic
        rpos = mpos + A - (L - bp) if mst else mpos + (L - bp)
        rows.append({"frag": frag.decode(), "swapped": swapped, "mm": mm, "bp": bp, "lpos": lpos % 2 ** 64, "rpos": rpos % 2 ** 64,
                     "strand": st, "mstrand": mst, "chr": ch.decode(), "mchr": mch.decode(),
                     "junction": junction.decode(), "signature": signature.decode(), "len": L})
    return {"args": [A, F, E], "reads": rows, "clusters": derive_clusters(rows, F, E)}


def derive_clusters(rows, F, E):
    """The printed clusters (detect.rs:193-241) as a fixed point, not as the reference's loops.
    order: the evidence sorted stably by (chr bytes, pos as usize).  For p before s in that order,
      compat(p, s) = same chr, pos[s] - pos[p] <= F, same mchr, |mpos[s] - mpos[p]| <= F as i64, same strands, same signature
    (:214-223; the first two are the loop's `break` tests, monotone in sorted order, so "until the first failure" and
    "everywhere" select the same reads).  A seed is a read that no earlier SEED is compatible with (:205, :226: only seeds
    report); a seed's cluster is itself and every later read compatible with it, whether an earlier seed already
    reported that read or not (:225).  Then the two min_evidence tests (:229, :231) around remove_frag_id_duplicates."""
    n = len(rows)
    order = sorted(range(n), key=lambda i: (rows[i]["chr"].encode(), rows[i]["lpos"]))            # stable; lpos is the wrapped usize
    i64 = lambda x: x - 2 ** 64 if x >= 2 ** 63 else x
    def compat(p, s):
        a, b = rows[order[p]], rows[order[s]]
        return (a["chr"] == b["chr"] and b["lpos"] - a["lpos"] <= F and a["mchr"] == b["mchr"] and
                abs(i64(b["rpos"]) - i64(a["rpos"])) <= F and a["strand"] == b["strand"] and a["mstrand"] == b["mstrand"] and
                a["signature"] == b["signature"])
    seeds = []
    for p in range(n):                                     # (a definition by recursion on p, evaluated in order)
        if not any(compat(q, p) for q in seeds):
            seeds.append(p)
    score = lambda m: min(rows[m]["bp"], rows[m]["len"] - rows[m]["bp"]) * 10000 + max(rows[m]["bp"], rows[m]["len"] - rows[m]["bp"])
    clusters = []
    for p in seeds:
        members = [order[p]] + [order[s] for s in range(p + 1, n) if compat(p, s)]
        if len(members) < E:
            continue
        kept = []
        for j in members:                                  # a group's representative stands at the slot of its first member
            group = [m for m in members if rows[m]["frag"] == rows[j]["frag"]]
            if group[0] != j:
                continue
            best = max(group, key=lambda m: (score(m), -group.index(m)))       # the first maximum (strict >, :280)
            kept.append(best)
        if len(kept) >= E:
            clusters.append({"seed": members[0], "members": members, "kept": kept})
    return clusters


def main():
    tables = {name: derive(name) for name in FIXTURES}
    r = tables[HAND_ROW["fixture"]]["reads"][HAND_ROW["read"]]
    for k in ("swapped", "bp", "lpos", "rpos", "junction", "signature"):
        assert r[k] == HAND_ROW[k], (k, r[k], HAND_ROW[k])
    assert r["mm"][:len(HAND_ROW["mm"])] == HAND_ROW["mm"]
    path = os.path.join(HERE, "hand_tables.json")
    json.dump(tables, open(path, "w"), indent=1, sort_keys=True)
    for name, t in tables.items():
        print("%-24s %d reads, %d clusters" % (name, len(t["reads"]), len(t["clusters"])))


if __name__ == "__main__":
    main()
