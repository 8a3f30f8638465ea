"""The packed genome image (bf_genome_image_write / _open: host code, no GPU): the file is read back here with numpy from
the format's description (breakfast_amd/csrc/genome_image.hip) and every byte of the contigs is rebuilt from it."""
import os
import struct

import numpy as np
import pytest

from breakfast_amd import engine

HEADER = struct.Struct("<8sIIQqQQQQ10Q6Q")      # 192 bytes


def every_byte_genome(rng, n):
    """ACGT with soft-masked stretches, N runs of many lengths, single IUPAC letters in both cases, other bytes."""
    s = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
    for _ in range(n // 3000 + 3):
        a = int(rng.integers(0, n)); m = int(rng.integers(1, 700))
        s[a:a + m] |= 0x20
    for m in (1, 2, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, 9000):
        for c in b"Nn":
            if m < n // 4:
                a = int(rng.integers(0, n - m)); s[a:a + m] = c
    for _ in range(n // 500 + 5):
        a = int(rng.integers(0, n)); m = int(rng.integers(1, 6))
        s[a:a + m] = rng.choice(np.frombuffer(b"RYKMSWBDHVNrykmswbdhvn*-.Xx", np.uint8), min(m, n - a))
    for a in (0, 63, 64, n - 1, n - 64):
        if 0 <= a < n:
            s[a] = ord("N")
    return s


def case_genome(seed=3):
    rng = np.random.default_rng(seed)
    lens = [200000, 1, 63, 64, 65, 4096, 4097, 131072 + 5, 70000, (4 << 20) + 777]
    names = [b"c%02d" % i for i in range(len(lens))]
    seqs = [every_byte_genome(rng, n) for n in lens]
    seqs[-1][(4 << 20) - 100:(4 << 20) + 100] = ord("n")          # a run across the packer's chunk boundary
    seqs[-1][(4 << 20) - 300:(4 << 20) - 100] |= 0x20
    seqs.append(np.full(10000, ord("N"), np.uint8)); names.append(b"allN")
    seqs.append(np.frombuffer(b"ACGT" * 2500, np.uint8).copy()); names.append(b"plain")
    seqs.append(np.frombuffer(b"acgt" * 16, np.uint8).copy()); names.append(b"lower64")   # ends on a block edge: its run touches
    seqs.append(np.frombuffer(b"acgtNN", np.uint8).copy()); names.append(b"next")         # the next contig's first base
    return names, seqs


def read_image(path):
    raw = np.fromfile(path, np.uint8)
    f = HEADER.unpack(raw[:192].tobytes())
    magic, version, n_contigs, src_size, src_mtime, n_blocks, n_lower, n_other, names_bytes = f[:9]
    off = dict(zip(["lens", "name_off", "names", "blocks", "lower_start", "lower_len", "other_start", "other_len", "other_byte", "file_size"], f[9:19]))
    assert magic == b"BFGENOM1" and version == 1 and off["file_size"] == raw.size
    assert all(v % 64 == 0 for v in off.values())

    def col(name, dtype, n):
        return raw[off[name]:off[name] + n * np.dtype(dtype).itemsize].view(dtype)
    lens = col("lens", np.uint64, n_contigs)
    name_off = col("name_off", np.uint32, n_contigs + 1)
    nb = raw[off["names"]:off["names"] + names_bytes].tobytes()
    names = [nb[name_off[i]:name_off[i + 1]] for i in range(n_contigs)]
    blocks = col("blocks", np.uint64, 2 * n_blocks).reshape(-1, 2)
    img = dict(names=names, lens=lens, blocks=blocks, src=(src_size, src_mtime),
               lower=(col("lower_start", np.uint64, n_lower), col("lower_len", np.uint32, n_lower)),
               other=(col("other_start", np.uint64, n_other), col("other_len", np.uint32, n_other), col("other_byte", np.uint8, n_other)))
    return img


def rebuild(img):
    """every contig's bytes from the image"""
    n_blocks = img["blocks"].shape[0]
    lo = np.unpackbits(img["blocks"][:, 0].copy().view(np.uint8), bitorder="little").reshape(n_blocks, 64)
    hi = np.unpackbits(img["blocks"][:, 1].copy().view(np.uint8), bitorder="little").reshape(n_blocks, 64)
    code = (lo | (hi << 1)).reshape(-1)
    out = np.frombuffer(b"ACTG", np.uint8)[code].copy()
    ls, ll = img["lower"]
    for s, n in zip(ls.tolist(), ll.tolist()):
        out[s:s + n] |= 0x20
    os_, ol_, ob = img["other"]
    for s, n, b in zip(os_.tolist(), ol_.tolist(), ob.tolist()):
        out[s:s + n] = b
    seqs, blk = [], 0
    for n in img["lens"].tolist():
        seqs.append(out[64 * blk:64 * blk + n])
        blk += (n + 63) // 64
    assert blk == n_blocks
    return seqs


@pytest.mark.parametrize("threads", [1, 5])
def test_image_holds_every_byte(tmp_path, threads):
    names, seqs = case_genome()
    path = str(tmp_path / "g.bf2bit")
    engine.write_genome_image(path, names, seqs, src_size=12345, src_mtime_ns=987654321012345678, threads=threads)
    assert os.listdir(tmp_path) == ["g.bf2bit"]                       # written beside, renamed when complete
    img = read_image(path)
    assert img["names"] == names and img["lens"].tolist() == [s.size for s in seqs]
    assert img["src"] == (12345, 987654321012345678)
    for want, got, name in zip(seqs, rebuild(img), names):
        assert np.array_equal(want, got), name
    for start, length in (img["lower"], img["other"][:2]):           # ascending, no overlap, nothing empty
        end = start + length
        assert (length > 0).all() and (start[1:] >= end[:-1]).all()
    # runs are maximal: two neighbours that touch differ (lower-case runs never touch; other runs differ in their byte)
    ls, ll = img["lower"]
    assert (ls[1:] > (ls + ll)[:-1]).all()
    os_, ol_, ob = img["other"]
    touch = os_[1:] == (os_ + ol_)[:-1]
    assert (ob[1:][touch] != ob[:-1][touch]).all()
    # padding after a contig's last base is code 0 and in no run
    total = 64 * img["blocks"].shape[0]
    member = np.zeros(total + 1, np.int32)
    for s, n in list(zip(ls.tolist(), ll.tolist())) + list(zip(os_.tolist(), ol_.tolist())):
        member[s] += 1; member[s + n] -= 1
    member = np.cumsum(member)[:-1]
    assert member.max() <= 1
    blk = 0
    for n in img["lens"].tolist():
        nb = (n + 63) // 64
        pad = slice(64 * blk + n, 64 * (blk + nb))
        assert member[pad].sum() == 0
        blk += nb


def test_same_file_whatever_the_thread_count(tmp_path):
    names, seqs = case_genome(seed=8)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    engine.write_genome_image(a, names, seqs, threads=1)
    engine.write_genome_image(b, names, seqs, threads=7)
    assert open(a, "rb").read() == open(b, "rb").read()


def test_open_checks_the_file(tmp_path):
    names, seqs = case_genome(seed=5)
    path = str(tmp_path / "g.bf2bit")
    engine.write_genome_image(path, names, seqs, src_size=777, src_mtime_ns=42)
    img = engine.GenomeImage(path, 777, 42)
    assert img.names == names and img.lens == [s.size for s in seqs]
    n_blocks = sum((s.size + 63) // 64 for s in seqs)
    assert img.upload_bytes >= 16 * n_blocks
    img.close()
    engine.GenomeImage(path).close()                                  # (0, 0): the source is not checked
    for size, mtime in ((778, 42), (777, 43)):
        with pytest.raises(engine.BreakfastError) as e:
            engine.GenomeImage(path, size, mtime)
        assert e.value.code == -7 and "stale" in str(e.value)
    with pytest.raises(engine.BreakfastError) as e:
        engine.GenomeImage(str(tmp_path / "absent"))
    assert e.value.code == -7
    raw = bytearray(open(path, "rb").read())
    cases = {"short": raw[:100], "cut": raw[:-64], "magic": b"XXGENOM1" + raw[8:], "version": raw[:8] + struct.pack("<I", 9) + raw[12:]}
    f = list(HEADER.unpack(bytes(raw[:192])))
    g = list(f); g[12] = len(raw) - 64                                # off_blocks: the section would leave the file
    cases["section"] = HEADER.pack(*g) + raw[192:]
    g = list(f); g[5] = f[5] + 1                                      # n_blocks does not add up to the contigs
    cases["blocks"] = HEADER.pack(*g) + raw[192:]
    g = list(f); g[2] = 0
    cases["contigs"] = HEADER.pack(*g) + raw[192:]
    for name, data in cases.items():
        p = str(tmp_path / name)
        open(p, "wb").write(bytes(data))
        with pytest.raises(engine.BreakfastError) as e:
            engine.GenomeImage(p)
        assert e.value.code == -7, name


def test_write_rejects_bad_arguments(tmp_path):
    names, seqs = case_genome(seed=6)
    with pytest.raises(engine.BreakfastError) as e:
        engine.write_genome_image(str(tmp_path / "no_such_dir" / "g"), names, seqs)
    assert e.value.code == -7
    with pytest.raises(engine.BreakfastError):
        engine.write_genome_image(str(tmp_path / "g"), [], [])
    assert os.listdir(tmp_path) == []
