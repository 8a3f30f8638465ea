"""CPU oracle (pure Python) for `breakfast detect` -- TEST INFRASTRUCTURE ONLY.

This is a line-by-line restatement of the reference's algorithm
(/root/reference/src/detect.rs:76-287) used to check the HIP path and the fast
C++ oracle (oracle/detect_oracle.cpp) on small cases.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the
product (breakfast_amd/, the C-ABI library, the CLI driver) never does.

PARITY UNPINNED by the reference: the reference ships no tests, fixtures or
golden vectors and cannot be built here (no cargo/rustc/bowtie/htslib).  This
restatement is pinned by (i) the hand-checked known-answer vector of SURVEY.md
section 8c (tests/golden/kat_survey.*), (ii) agreement with the independent
C++ restatement on every authored hazard fixture and on randomised inputs.

Third-party arithmetic restated here (not vendored under /root/reference):
  bio 0.39 `alphabets::dna::complement` / `revcomp` (used at detect.rs:112,
  121,130): identity except AGCTYRWSKMDVHBN -> TCGARYWSMKHBDVN and the same
  for lowercase.
  bio 0.39 `io::fasta::Reader` (detect.rs:48-56): id = first whitespace
  delimited word of the header, sequence lines appended with trailing
  whitespace trimmed, case preserved.

All `usize` arithmetic is 64-bit wrapping (the shipped binary is a release
build: README.md:19 `cargo install`, no overflow checks).
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Dict, Iterable, List, Tuple

U64 = (1 << 64) - 1

_COMP = bytearray(range(256))
for _a, _b in zip(b"AGCTYRWSKMDVHBN", b"TCGARYWSMKHBDVN"):
    _COMP[_a] = _b
    _COMP[_a + 32] = _b + 32
_COMP = bytes(_COMP)

SV_HEADER = (b"CHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\tCHROM\tSTRAND\tPOSITION\t"
             b"NEARBY FEATURES\tSUPPORTING READS\tSIGNATURE\tNOTES\n")


def complement(b: int) -> int:
    """bio::alphabets::dna::complement (detect.rs:121,130)."""
    return _COMP[b]


def revcomp(seq: bytes) -> bytes:
    """bio::alphabets::dna::revcomp (detect.rs:112)."""
    return bytes(_COMP[b] for b in reversed(seq))


def _i64(x: int) -> int:
    """`x as i64` for a wrapping usize (detect.rs:220)."""
    x &= U64
    return x - (1 << 64) if x >> 63 else x


def parse_fasta(text: bytes) -> Dict[bytes, bytes]:
    """bio::io::fasta::Reader + the HashMap insert loop (detect.rs:48-56).

    A repeated id overwrites the earlier record (HashMap::insert)."""
    genome: Dict[bytes, bytes] = {}
    name = None
    chunks: List[bytes] = []
    for raw in text.split(b"\n"):
        line = raw.rstrip(b" \t\r\n\x0b\x0c")
        if raw.startswith(b">"):
            if name is not None:
                genome[name] = b"".join(chunks)
            # header[1..].trim_end().splitn(2, char::is_whitespace).next()
            name = re.split(rb"[ \t\r\x0b\x0c]", line[1:], maxsplit=1)[0]
            chunks = []
        elif name is not None:
            chunks.append(line)
    if name is not None:
        genome[name] = b"".join(chunks)
    return genome


@dataclass
class Evidence:
    """struct Evidence (detect.rs:13-26).  `pos`/`mpos` hold the breakpoint
    coordinates (detect.rs:184-185), as wrapping u64."""
    chr: bytes
    pos: int
    strand: bool
    mchr: bytes
    mpos: int
    mstrand: bool
    sequence: bytes
    signature: bytes
    frag_id: bytes
    # extras for kernel-level parity checks (not in the reference struct)
    bp: int = 0
    swapped: bool = False
    read_num: int = 0
    full_len: int = 0


def _get(seq: bytes, idx: int) -> int:
    """`*chr_seq.get(idx).unwrap_or(&b'N')` with a wrapping usize index."""
    idx &= U64
    return seq[idx] if idx < len(seq) else 0x4E


def localise(genome: Dict[bytes, bytes], anchor_len: int, frag_id: bytes, seq: bytes,
             strand: bool, chr_: bytes, pos: int,
             mstrand: bool, mchr: bytes, mpos: int) -> Evidence:
    """One iteration of the consume loop body, detect.rs:92-187."""
    full_len = len(seq)
    swapped = False
    # detect.rs:108-113 reorient
    if chr_ > mchr or (chr_ == mchr and pos > mpos):
        chr_, mchr = mchr, chr_
        pos, mpos = mpos, pos
        strand, mstrand = (not mstrand), (not strand)
        seq = revcomp(seq)
        swapped = True

    chr_seq = genome[chr_]          # detect.rs:115 (panics if missing -> KeyError)
    if strand:                      # detect.rs:116-122
        left = bytes(_get(chr_seq, pos - 1 + k) for k in range(full_len))
    else:
        left = bytes(complement(_get(chr_seq, pos + anchor_len - 2 - k)) for k in range(full_len))
    mchr_seq = genome[mchr]         # detect.rs:124
    if mstrand:                     # detect.rs:125-131
        right = bytes(_get(mchr_seq, mpos + anchor_len - full_len - 1 + k) for k in range(full_len))
    else:
        right = bytes(complement(_get(mchr_seq, mpos + full_len - 2 - k)) for k in range(full_len))

    # detect.rs:135-153 split scan
    mism = [0] * full_len
    for k in range(anchor_len):
        mism[anchor_len] += int(seq[k] != left[k])
    for k in range(anchor_len, full_len):
        mism[anchor_len] += int(seq[k] != right[k])
    for k in range(anchor_len + 1, full_len - anchor_len):
        mism[k] = mism[k - 1] + int(seq[k - 1] != left[k - 1]) - int(seq[k - 1] != right[k - 1])
    bp = anchor_len
    least = mism[anchor_len]
    for k in range(anchor_len + 1, full_len - anchor_len):
        if mism[k] < least:
            bp, least = k, mism[k]

    # detect.rs:155-162 junction
    junction = bytearray(b" " * (full_len + 1))
    for k in range(bp):
        junction[k] = seq[k] if seq[k] == left[k] else _lower(seq[k])
    junction[bp] = 0x7C
    for k in range(bp, full_len):
        junction[k + 1] = seq[k] if seq[k] == right[k] else _lower(seq[k])

    # detect.rs:165-168 signature
    signature = bytes(left[bp - 8:bp]) + b"|" + bytes(right[bp:bp + 8])
    assert len(signature) == 17, "reference would panic (index out of range)"

    # detect.rs:172-181 breakpoint coordinates (wrapping usize)
    left_bp = (pos + bp - 1) & U64 if strand else (pos + anchor_len - bp) & U64
    right_bp = ((mpos + anchor_len - (full_len - bp)) & U64 if mstrand
                else (mpos + (full_len - bp)) & U64)
    return Evidence(chr_, left_bp, strand, mchr, right_bp, mstrand, bytes(junction),
                    signature, frag_id, bp=bp, swapped=swapped, full_len=full_len)


def _lower(b: int) -> int:
    """u8::to_ascii_lowercase (detect.rs:157,161)."""
    return b + 32 if 0x41 <= b <= 0x5A else b


def consume_lines(genome: Dict[bytes, bytes], lines: Iterable[bytes], anchor_len: int) -> List[Evidence]:
    """The bowtie-output consume loop, detect.rs:72-188."""
    evidence: List[Evidence] = []
    prev = b""
    prev_read_num = 0
    for line in lines:
        read_num = int(line.split(b":")[1])                 # detect.rs:79
        if line.startswith(b"5p:"):                          # detect.rs:81-84
            prev, prev_read_num = line, read_num
            continue
        elif line.startswith(b"3p:") and read_num != prev_read_num:   # detect.rs:85-87
            continue
        info = line.split(b":")
        frag_id = info[2]                                    # detect.rs:90
        seq = info[3]                                        # detect.rs:91
        cols = prev.split(b"\t")
        mcols = line.split(b"\t")
        strand = cols[1] == b"+"
        mstrand = mcols[1] == b"+"
        chr_, mchr = cols[2], mcols[2]
        pos, mpos = int(cols[3]), int(mcols[3])
        ev = localise(genome, anchor_len, frag_id, seq, strand, chr_, pos, mstrand, mchr, mpos)
        ev.read_num = read_num
        evidence.append(ev)
    return evidence


def remove_frag_id_duplicates(evidence: List[Evidence]) -> List[Evidence]:
    """detect.rs:247-287."""
    filtered: List[Evidence] = []
    redundant_with = [-1] * len(evidence)
    for a in range(len(evidence)):
        if redundant_with[a] >= 0:
            continue
        redundant_with[a] = a
        num_redundant = 1
        for b in range(a + 1, len(evidence)):
            if redundant_with[b] >= 0:
                continue
            if evidence[a].frag_id == evidence[b].frag_id:
                redundant_with[b] = a
                num_redundant += 1
        if num_redundant == 1:
            filtered.append(evidence[a])
            continue
        best, best_score = 0, 0
        for k in range(len(evidence)):
            if redundant_with[k] != a:
                continue
            seq = evidence[k].sequence
            left_len = seq.index(b"|")
            right_len = len(seq) - left_len - 1
            score = min(left_len, right_len) * 10000 + max(left_len, right_len)
            if score > best_score:
                best, best_score = k, score
        filtered.append(evidence[best])
    return filtered


def cluster_and_format(evidence: List[Evidence], max_frag_len: int, min_evidence: int
                       ) -> Tuple[bytes, List[Tuple[Evidence, List[Evidence]]]]:
    """Stable sort, greedy clustering, dedup and .sv printing, detect.rs:193-241."""
    # Python's sort is stable like Rust's sort_by; bytes compare lexicographically
    # byte-wise like Rust Strings (detect.rs:193-198).
    evidence = sorted(evidence, key=lambda e: (e.chr, e.pos))
    out = bytearray(SV_HEADER)                                # detect.rs:201
    clusters = []
    reported = [False] * len(evidence)
    for p in range(len(evidence)):
        if reported[p]:
            continue
        read = evidence[p]
        cluster = [read]
        for s in range(p + 1, len(evidence)):
            e = evidence[s]
            if e.chr != read.chr:
                break
            if ((e.pos - read.pos) & U64) > max_frag_len:
                break
            if e.mchr != read.mchr:
                continue
            if abs(_i64(e.mpos) - _i64(read.mpos)) > max_frag_len:
                continue
            if e.strand != read.strand:
                continue
            if e.mstrand != read.mstrand:
                continue
            if e.signature != read.signature:
                continue
            cluster.append(e)
            reported[s] = True
        if len(cluster) < min_evidence:
            continue
        cluster = remove_frag_id_duplicates(cluster)
        if len(cluster) < min_evidence:
            continue
        out += b"%s\t%s\t%d\t\t%s\t%s\t%d\t\t" % (
            read.chr, b"+" if read.strand else b"-", read.pos,
            read.mchr, b"+" if read.mstrand else b"-", read.mpos)
        out += b";".join(r.sequence for r in cluster)
        out += b"\t\t\n"
        clusters.append((read, cluster))
    return bytes(out), clusters


def detect(fasta_text: bytes, bowtie_lines: Iterable[bytes], anchor_len: int = 30,
           max_frag_len: int = 5000, min_evidence: int = 2) -> bytes:
    """stdout of `breakfast detect` given the FASTA text and bowtie's stdout lines."""
    genome = parse_fasta(fasta_text)
    ev = consume_lines(genome, bowtie_lines, anchor_len)
    return cluster_and_format(ev, max_frag_len, min_evidence)[0]
