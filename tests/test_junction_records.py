"""Junction records (include/breakfast_hip.h, "Junction records"): the library's host decoder against
records packed here in plain Python from junction strings -- packed form, exception entries, text form,
breakpoint at either end, padding, malformed input.  No GPU involved."""
import numpy as np
import pytest

from breakfast_amd import engine

CODE = {ord("A"): 0, ord("C"): 1, ord("T"): 2, ord("G"): 3}


def pack(junction: bytes, form: str = "auto", swapped: bool = False) -> bytes:
    """Reference packer: the record the GPU writes for this string (form "auto"), or a given packed form.
    Header word: kind (1 dense, 2 sparse) | reverse-complemented << 2 | L << 3 | bp << 12 | n_exc << 21 | n_lower << 27;
    then per 8 bases {lo byte, hi byte[, lower-case byte]}, {position, byte} per base outside ACGT, lower-case
    positions as bytes (sparse), zero padding to 4 bytes.  Packed records hold reads of up to 256 bases."""
    bp = junction.index(b"|")
    seq = junction[:bp] + junction[bp + 1:]
    L = len(seq)
    if L > 256:
        return text_record(junction, swapped)
    units = (L + 7) // 8
    lo, hi, cs = [0] * units, [0] * units, [0] * units
    exc, lower = [], []
    for k, c in enumerate(seq):
        up = c - 32 if 97 <= c <= 122 else c
        if up in CODE:
            lo[k >> 3] |= (CODE[up] & 1) << (k & 7)
            hi[k >> 3] |= (CODE[up] >> 1) << (k & 7)
            if c != up:
                cs[k >> 3] |= 1 << (k & 7)
                lower.append(k)
        else:
            exc.append((k, c))
    fits = len(lower) <= 31 and len(exc) <= 63
    sparse = (fits and len(lower) < units) if form == "auto" else (form == "sparse" and fits)
    head = (2 if sparse else 1) | (4 if swapped else 0) | (L << 3) | (bp << 12) | (len(exc) << 21) | ((len(lower) if sparse else 0) << 27)
    out = np.array([head], "<u4").tobytes()
    for u in range(units):
        out += bytes([lo[u], hi[u]] + ([] if sparse else [cs[u]]))
    for k, c in exc:
        out += bytes([k, c])
    if sparse:
        out += bytes(lower)
    return out + b"\0" * (-len(out) % 4)


def text_record(junction: bytes, swapped: bool = False) -> bytes:
    rec = np.array([(4 if swapped else 0) | (junction.index(b"|") << 8), len(junction) - 1], "<u4").tobytes() + junction
    return rec + b"\0" * (-len(rec) % 4)


def random_junction(rng, L, n_exc=0, lower=0.05):
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L)
    seq = np.where(rng.random(L) < lower, seq | 0x20, seq).astype(np.uint8)
    for k in rng.choice(L, size=min(n_exc, L), replace=False):
        seq[k] = rng.choice(np.frombuffer(b"NnRYKM=", np.uint8))
    bp = int(rng.integers(0, L + 1))
    s = seq.tobytes()
    return s[:bp] + b"|" + s[bp:]


@pytest.mark.parametrize("L", [1, 7, 8, 31, 32, 33, 40, 64, 100, 150, 191, 192, 250, 256, 1000])
def test_packed_records_expand_to_the_string(L):
    rng = np.random.default_rng(L)
    for n_exc in (0, 1, 3, L // 8):
        for _ in range(20):
            j = random_junction(rng, L, n_exc)
            assert engine.junction_text(pack(j)) == j
            assert engine.junction_text(pack(j, "dense")) == j
            assert engine.junction_text(pack(j, "sparse")) == j
            assert engine.junction_text(text_record(j)) == j


def test_breakpoint_at_either_end_and_known_answer():
    assert engine.junction_text(pack(b"|ACGT")) == b"|ACGT"
    assert engine.junction_text(pack(b"ACGT|")) == b"ACGT|"
    # the survey's known-answer read 1
    j = b"AGTTATCGGTGCCG|AAcTGATGGA"
    rec = pack(j)                                   # 24 bases, one mismatch: sparse form, 4 + 3 * 2 + 1 bytes, padded
    assert len(rec) == 12 and engine.junction_text(rec) == j
    assert len(pack(j, "dense")) == 16              # 4 + 3 * 3, padded


def test_malformed_records_are_rejected():
    good = pack(b"ACGTACGTAC|GTNACGTACGT", "dense")
    sparse = pack(b"ACGTACgTAC|GTNACGTACGT", "sparse")
    assert engine.junction_text(good) == b"ACGTACGTAC|GTNACGTACGT" and engine.junction_text(sparse) == b"ACGTACgTAC|GTNACGTACGT"
    u4 = lambda *w: np.array(w, "<u4").tobytes()
    head = lambda kind, L, bp, nx=0, nl=0: u4(kind | (L << 3) | (bp << 12) | (nx << 21) | (nl << 27))
    for bad in (b"", good[:3], good[:8], good[:-4], sparse[:-4],
                head(2, 21, 10, 0, 1) + sparse[4:4 + 6] + bytes([99, 0]),         # lower-case position past the end
                head(1, 21, 10, 1, 1) + good[4:],                               # dense form with a count
                head(1, 4, 9) + b"\0" * 12,                                     # bp > len
                head(1, 4, 1, 1) + b"\0\0\0" + bytes([7, ord("N")]) + b"\0" * 3,   # exception past the end
                head(1, 300, 1) + b"\0" * 200,                                  # packed records end at 256 bases
                u4(3) + b"\0" * 12,                                             # unknown kind
                u4(0 | (1 << 8), 10) + b"A|CG",                                  # text shorter than len + 1
                u4(0 | (20 << 8), 10) + b"A|CGACGTACGT"):                        # text: bp > len
        with pytest.raises(engine.BreakfastError):
            engine.junction_text(bad)


def test_result_junction_texts_batch():
    rng = np.random.default_rng(7)
    js = [random_junction(rng, int(rng.integers(40, 300)), int(rng.integers(0, 4))) for _ in range(500)]
    recs = [pack(j) if i % 3 else text_record(j) for i, j in enumerate(js)]
    rd = np.zeros(len(js), engine.READ_DTYPE)
    off = 0
    for i, r in enumerate(recs):
        rd[i]["junction_off"] = off
        off += len(r)
    res = engine.Result(len(js), np.zeros(0, engine.CLUSTER_DTYPE), rd, np.frombuffer(b"".join(recs), np.uint8))
    text, toff = res.junction_texts()
    assert [text[int(toff[i]):int(toff[i + 1])].tobytes() for i in range(len(js))] == js
    assert res.junction(17) == js[17]
    # headers: length, breakpoint, reorientation flag, stored size
    buf = np.frombuffer(b"".join(recs), np.uint8)
    for i in (0, 1, 2, 250):
        L, bp, sw, size = engine.junction_header(buf, int(rd[i]["junction_off"]))
        assert (L, bp, sw, size) == (len(js[i]) - 1, js[i].index(b"|"), False, len(recs[i]))
    assert engine.junction_header(pack(js[3], swapped=True))[2] and engine.junction_header(text_record(js[3], True))[2]
    assert engine.junction_text(pack(js[3], swapped=True)) == js[3]


def test_records_found_through_cluster_offsets():
    """A one-GPU result arrives without the read table: a cluster's records are contiguous, every record says
    how long it is, cluster_junction_off says where each cluster starts (include/breakfast_hip.h, bf_result)."""
    rng = np.random.default_rng(11)
    js = [random_junction(rng, int(rng.integers(40, 300)), int(rng.integers(0, 4))) for _ in range(300)]
    recs = [pack(j) if i % 4 else text_record(j) for i, j in enumerate(js)]
    sizes = [0, 1, 2, 40, 7, 0, 250]                       # reads per cluster (empty clusters never occur; harmless)
    assert sum(sizes) == len(js)
    cl = np.zeros(len(sizes), engine.CLUSTER_DTYPE)
    cjo = np.zeros(len(sizes) + 1, np.uint64)
    k = 0
    for c, m in enumerate(sizes):
        cl[c]["read_begin"], cl[c]["read_end"] = k, k + m
        cjo[c + 1] = cjo[c] + sum(len(r) for r in recs[k:k + m])
        k += m
    jb = np.frombuffer(b"".join(recs), np.uint8)
    res = engine.Result(len(js), cl, np.zeros(0, engine.READ_DTYPE), jb, n_reads=len(js), cluster_junction_off=cjo)
    text, toff = res.junction_texts()
    assert [text[int(toff[i]):int(toff[i + 1])].tobytes() for i in range(len(js))] == js
    # offsets that do not meet the records are refused, not followed
    for c, delta in ((3, 4), (4, -4), (7, 8), (7, -4)):
        bad = cjo.copy()
        bad[c] = np.uint64(int(bad[c]) + delta)
        with pytest.raises(engine.BreakfastError):
            engine.Result(len(js), cl, np.zeros(0, engine.READ_DTYPE), jb, n_reads=len(js),
                          cluster_junction_off=bad).junction_texts()
    # neither table nor offsets
    with pytest.raises(engine.BreakfastError):
        lib = engine.load_library()
        r = engine.BfResult(0, 0, 5, None, None, jb.ctypes.data, jb.size, None)
        off = np.zeros(6, np.uint64)
        n = lib.bf_result_junction_text(engine.C.byref(r), None, 0, off.ctypes.data)
        if n < 0:
            raise engine.BreakfastError(int(n), "no way to the records")


def test_mutated_records_never_read_past_their_bytes():
    """Random byte flips, truncations and random bytes: the decoder answers with a length or BF_ERR_INVALID, and what it
    writes stays inside the buffer it was given (a guard band behind it must survive); a reference record (kind 3) is
    sized without its ctx and refused by the ctx-less expander."""
    import ctypes as C
    lib = engine.load_library()
    lib.bf_junction_record_bytes.restype = C.c_int64
    lib.bf_junction_record_bytes.argtypes = [C.c_void_p, C.c_uint64]
    rng = np.random.default_rng(77)
    guard = 64
    n_ok = n_bad = 0
    for trial in range(4000):
        L = int(rng.integers(1, 300))
        rec = bytearray(pack(random_junction(rng, L, n_exc=int(rng.integers(0, 4)))))
        how = trial % 4
        if how == 0:                                              # flip a few bytes
            for _ in range(int(rng.integers(1, 4))):
                rec[int(rng.integers(0, len(rec)))] = int(rng.integers(0, 256))
        elif how == 1:                                            # cut it short
            rec = rec[:int(rng.integers(0, len(rec)))]
        elif how == 2:                                            # noise
            rec = bytearray(rng.integers(0, 256, size=int(rng.integers(0, 80)), dtype=np.uint8).tobytes())
        # how == 3: left intact
        src = np.frombuffer(bytes(rec), np.uint8).copy() if rec else np.zeros(0, np.uint8)
        ptr = src.ctypes.data if src.size else None
        n = lib.bf_junction_text(ptr, src.size, None, 0)
        sz = lib.bf_junction_record_bytes(ptr, src.size)
        if n < 0:
            n_bad += 1
            assert n == -1 and (sz == -1 or (src.size >= 4 and src[0] & 3 == 3))   # BF_ERR_INVALID (kind 3: sized, not expanded)
            continue
        n_ok += 1
        assert 1 <= n <= 65536 and 0 < sz <= ((src.size + 3) & ~3)
        buf = np.full(n + guard, 0xEE, np.uint8)
        assert lib.bf_junction_text(ptr, src.size, buf.ctypes.data, n) == n
        assert np.all(buf[n:] == 0xEE) and bytes(buf[:n]).count(b"|") >= 1
        if how == 3:
            assert sz == len(rec)
    assert n_ok > 1000 and n_bad > 500

    # a reference record: header kind 3, two lower-case positions
    ref = np.array([3 | (2 << 4) | (5 << 16), 7], "<u4").tobytes() + bytes([1, 9, 0, 0])
    src = np.frombuffer(ref, np.uint8).copy()
    assert lib.bf_junction_record_bytes(src.ctypes.data, src.size) == 12
    assert lib.bf_junction_text(src.ctypes.data, src.size, None, 0) == -1
    assert lib.bf_junction_record_bytes(src.ctypes.data, 9) == -1
