"""CPU restatement of `breakfast filter` (reference src/filter.rs:39-156) -- TEST INFRASTRUCTURE ONLY.

Pure Python, line by line.  Nothing under breakfast_amd/ may import this module.  The reference ships no
tests or fixtures for this tool, so parity is unpinned by the reference; what pins
this restatement are the hand-derived cases in tests/test_text_tools.py.

Files are bytes; Rust's &str operations used by filter.rs are byte-wise on ASCII input (split, sort, slice).
`Fatal` = the reference's error! macro (stderr "ERROR: ..." + exit(-1), common.rs:8-13); `Panic` = where
the reference would panic (.unwrap() on a failed parse, slice index out of range).
"""
import os
from typing import Dict, List, Tuple


class Fatal(Exception):
    pass


class Panic(Exception):
    pass


def read_lines(path: str) -> List[bytes]:
    """FileReader::read_line until EOF (common.rs:49-56): one trailing '\\n' removed per line; a final
    line without newline still counts."""
    try:
        data = open(path, "rb").read()
    except OSError:
        raise Fatal("Cannot open file %s for reading." % path)          # common.rs:35-36
    if not data:
        return []
    lines = data.split(b"\n")
    if lines[-1] == b"":
        lines.pop()
    return lines


def parse_usize(s: bytes) -> int:
    t = s[1:] if s[:1] == b"+" else s
    if not t or not t.isdigit() or int(t) >= 1 << 64:
        raise Panic("not a usize: %r" % s)
    return int(t)


def read_rearrangement(cols: List[bytes]) -> Tuple:
    """filter.rs:139-146"""
    if len(cols) < 7 or not cols[1] or not cols[5]:
        raise Panic("rearrangement columns")
    return (cols[0], parse_usize(cols[2]), cols[1][:1], cols[4], parse_usize(cols[6]), cols[5][:1])


def deduplicate_reads(reads: List[bytes]) -> List[bytes]:
    """filter.rs:148-156: stable sort + dedup_by_key (first of a run stays) on the last 20 bytes, then sort +
    dedup on the first 20 bytes."""
    for r in reads:
        if len(r) < 20:
            raise Panic("read shorter than 20")                          # &r[len-20..len] underflows
    reads = sorted(reads, key=lambda r: r[-20:])
    out = []
    for r in reads:
        if not out or out[-1][-20:] != r[-20:]:
            out.append(r)
    reads = sorted(out)
    out = []
    for r in reads:
        if not out or out[-1][:20] != r[:20]:
            out.append(r)
    return out


def filter_cohort(sample_sheet: str, min_reads: int = 4, min_distance: int = 10, keep_duplicates: bool = False,
                  output_dir: str = "annotated", cwd: str = ".") -> Dict[str, bytes]:
    """-> {output path: bytes}.  `<sample>.sv` files are looked up relative to `cwd` like the reference
    looks them up relative to its working directory (filter.rs:86)."""
    merge = not keep_duplicates
    lines = read_lines(sample_sheet)
    header = lines[0] if lines else b""
    headers = header.split(b"\t")
    def col(name):
        return headers.index(name) if name in headers else 99
    patient_col, sample_col, type_col = col(b"PATIENT"), col(b"SAMPLE"), col(b"TYPE")
    if 99 in (patient_col, sample_col, type_col):
        raise Fatal("Required headers PATIENT, SAMPLE and TYPE were not all found on the first row of the sample sheet.")
    min_columns = max(patient_col, sample_col, type_col)
    samples = []
    for line in lines[1:]:
        if not line.strip():
            continue
        cols = line.split(b"\t")
        if len(cols) < min_columns:
            raise Fatal("Sample sheet line has too few columns")
        if len(cols) <= min_columns:
            raise Panic("index out of bounds, filter.rs:69-71")
        patient, sample, stype = cols[patient_col].strip(), cols[sample_col].strip(), cols[type_col].strip().lower()
        if not patient or not sample or not stype:
            raise Fatal("Missing PATIENT, SAMPLE or TYPE column on line")
        samples.append((sample, patient, stype in (b"wbc", b"gdna", b"germline")))

    rearrangements: Dict[Tuple, List[Tuple[bytes, int]]] = {}
    for name, _, _ in samples:
        sv = read_lines(os.path.join(cwd, name.decode() + ".sv"))
        for line in sv[1:]:
            cols = line.split(b"\t")
            if len(cols) < 9:
                raise Fatal("File %s.sv has a line with too few columns" % name.decode())
            r = read_rearrangement(cols)
            reads = cols[8].split(b";")
            if merge:
                reads = deduplicate_reads(reads)
            rearrangements.setdefault(r, []).append((name, len(reads)))

    outputs: Dict[str, bytes] = {}
    for name, patient, germline in samples:
        controls = {s for s, p, g in samples if p != patient or (not germline and g)}
        if not controls:
            raise Fatal("No negative controls found for sample %s." % name.decode())
        sv = read_lines(os.path.join(cwd, name.decode() + ".sv"))
        out = [(sv[0] if sv else b"") + b"\n"]
        for line in sv[1:]:
            cols = line.split(b"\t")
            r = read_rearrangement(cols)
            if r[0] == r[3] and r[2] == r[5] and abs(r[1] - r[4]) < min_distance:
                continue
            if len(cols) < 9:
                raise Panic("cols[8]")
            reads = cols[8].split(b";")
            if merge:
                reads = deduplicate_reads(reads)
            if len(reads) < min_reads:
                continue
            max_control = max([n for s, n in rearrangements[r] if s in controls], default=0)
            if len(reads) < 50 * max_control:
                continue
            out.append(b"\t".join(cols[:8] + [b";".join(reads)] + cols[9:]) + b"\n")
        outputs[os.path.join(output_dir, name.decode() + ".sv")] = b"".join(out)
    return outputs
