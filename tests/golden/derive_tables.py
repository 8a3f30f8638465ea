#!/usr/bin/env python3
"""Per-read known-answer tables for five hazard fixtures, derived WITHOUT either oracle.

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

Output: tests/golden/hand_tables.json -- literal numbers and strings per read (input order of the paired reads):
swapped, mm[A..L-A-1], bp, lpos, rpos, junction, signature; and for hz_dedup_min_evidence the surviving reads of
every printed cluster (detect.rs:247-287: groups by fragment id, score min(l,r)*10000+max(l,r), first maximum).
tests/test_hand_tables.py checks BOTH restatements and the GPU against the committed file.

Run:  python tests/golden/derive_tables.py        (rewrites hand_tables.json; the diff must be empty)
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = ["hz_tie_lowest_k", "hz_len_2a", "hz_contig_ends", "hz_ref_alphabet", "hz_dedup_min_evidence"]

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
    """(fragment id, sequence, (strand, chr, pos), (mstrand, mchr, mpos)) per 5p/3p pair that the pairing rule of
    detect.rs:79-87 accepts (these five fixtures have no unpaired lines)."""
    lines = [l for l in open(path, "rb").read().split(b"\n") if l]
    assert len(lines) % 2 == 0
    for l5, l3 in zip(lines[0::2], lines[1::2]):
        h5, s5, c5, p5 = l5.split(b"\t")
        h3, s3, c3, p3 = l3.split(b"\t")
        assert h5.startswith(b"5p:") and h3.startswith(b"3p:") and h5.split(b":")[1] == h3.split(b":")[1]
        _, _, frag, seq = h3.split(b":")[:4]
        yield frag, seq, (s5 == b"+", c5, int(p5)), (s3 == b"+", c3, int(p3))


def derive(name):
    g = fasta(os.path.join(HERE, name + ".fa"))
    A, F, E = (int(x) for x in open(os.path.join(HERE, name + ".args")).read().split())
    rows = []
    for frag, seq, (st, ch, pos), (mst, mch, mpos) in pairs(os.path.join(HERE, name + ".bowtie")):
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
    out = {"args": [A, F, E], "reads": rows}
    if name == "hz_dedup_min_evidence":
        out["clusters"] = dedup_clusters(rows, F, E)
    return out


def dedup_clusters(rows, F, E):
    """For this fixture every read of a junction shares (chr, strands, mchr, signature) and sits within max_frag_len, so
    a cluster is simply the reads of one junction in sorted order (asserted); what is derived here is
    remove_frag_id_duplicates (detect.rs:247-287) and the two min_evidence tests (:229, :231)."""
    order = sorted(range(len(rows)), key=lambda i: (rows[i]["chr"].encode(), rows[i]["lpos"]))     # stable, detect.rs:193-198
    klass = lambda r: (r["chr"], r["strand"], r["mchr"], r["mstrand"], r["signature"])
    clusters, seen = [], set()
    for i in order:
        if i in seen:
            continue
        members = [j for j in order if klass(rows[j]) == klass(rows[i])]
        assert all(abs(rows[j]["lpos"] - rows[i]["lpos"]) <= F and abs(rows[j]["rpos"] - rows[i]["rpos"]) <= F for j in members)
        seen.update(members)
        if len(members) < E:
            continue
        kept = []
        for j in members:                                  # representative at the slot of the group's first member
            group = [m for m in members if rows[m]["frag"] == rows[j]["frag"]]
            if group[0] != j:
                continue
            score = lambda m: min(rows[m]["bp"], rows[m]["len"] - rows[m]["bp"]) * 10000 + max(rows[m]["bp"], rows[m]["len"] - rows[m]["bp"])
            best = group[0]
            for m in group[1:]:
                if score(m) > score(best):                 # strict: the first maximum stays
                    best = m
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
        print("%-24s %d reads%s" % (name, len(t["reads"]), ", %d clusters" % len(t["clusters"]) if "clusters" in t else ""))


if __name__ == "__main__":
    main()
