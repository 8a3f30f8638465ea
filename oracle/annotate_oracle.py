"""CPU restatement of `breakfast annotate` (reference src/annotate.rs:26-93) -- TEST INFRASTRUCTURE ONLY.

Pure Python, line by line; nothing under breakfast_amd/ may import it.  The reference has no tests for
this tool: parity is unpinned by the reference and pinned by the hand-derived cases in
tests/test_text_tools.py.  `Fatal` = error! (common.rs:8-13), `Panic` = where the reference panics.
"""
from typing import List

from filter_oracle import Fatal, Panic, read_lines


def parse_u32(s: bytes) -> int:
    t = s[1:] if s[:1] == b"+" else s
    if not t or not t.isdigit() or int(t) >= 1 << 32:
        raise Panic("not a u32: %r" % s)
    return int(t)


def distance(pos: int, start: int, end: int) -> int:
    """annotate.rs:9-17"""
    if pos < start:
        return start - pos
    if pos > end:
        return pos - end
    return 0


def annotate(sv_path: str, bed_path: str) -> bytes:
    sv = read_lines(sv_path)
    out: List[bytes] = [(sv[0] if sv else b"") + b"\n"]                  # annotate.rs:33-34
    features = []
    for line in read_lines(bed_path):
        cols = line.split(b"\t")
        if len(cols) < 4:
            raise Fatal("Feature BED file has too few columns on this line")
        start = parse_u32(cols[1]) + 1                                   # 1-based first base, annotate.rs:46
        if start >= 1 << 32:
            raise Panic("u32 overflow")
        features.append((cols[0], start, parse_u32(cols[2]), cols[3]))
    for line in sv[1:]:
        if not line.startswith(b"chr"):
            continue
        cols = line.split(b"\t")
        if len(cols) < 11:
            raise Panic("cols[10]")
        pos_1, pos_2 = parse_u32(cols[2]), parse_u32(cols[6])
        def nearby(chrom, pos):
            near = [(distance(pos, s, e), name) for c, s, e, name in features if c == chrom]
            near = [x for x in near if x[0] <= 100000]
            near.sort(key=lambda x: x[0])                                # stable, annotate.rs:69
            return b", ".join(b"%s (%d)" % (name, d) for d, name in near)
        out.append(b"\t".join([cols[0], cols[1], b"%d" % pos_1, nearby(cols[0], pos_1), cols[4], cols[5], b"%d" % pos_2,
                               nearby(cols[4], pos_2), cols[8], cols[9], cols[10]]) + b"\n")
    return b"".join(out)
