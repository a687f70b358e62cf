"""The FASTA reader's own gzip / DEFLATE decoder (csrc/gdca_inflate.cpp: the fast path of `read_fasta_alignment` on .gz files,
which is what the reference's test data are) against zlib, CPU only: every block type and copy path on crafted streams, the
reference's .gz fixtures, header variants, multi-member files, and random corruptions under AddressSanitizer + UBSan (the decoder
must never touch memory outside its buffers, and whenever it accepts an input its output must be zlib's).  tests/sanitize/
inflate_check.cpp is the driver."""
import gzip
import os
import shutil
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussdca.jl_amd", "csrc")
BUILD = os.path.join(ROOT, "tests", "_build")
REFDATA = os.path.join(ROOT, "tests", "golden", "reference")
LETTERS = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY-", dtype=np.uint8)


@pytest.fixture(scope="module")
def bins():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    os.makedirs(BUILD, exist_ok=True)
    src = [os.path.join(ROOT, "tests", "sanitize", "inflate_check.cpp"), os.path.join(CSRC, "gdca_inflate.cpp")]
    out = {"plain": os.path.join(BUILD, "inflate_check"), "asan": os.path.join(BUILD, "inflate_check_asan"), "tsan": os.path.join(BUILD, "inflate_check_tsan")}
    for kind, flags in (("plain", ["-O2"]),
                        ("asan", ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]),
                        ("tsan", ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=thread"])):
        r = subprocess.run(["g++", *flags, "-std=c++17", "-Wall", "-pthread", "-I" + CSRC, *src, "-o", out[kind], "-lz"], capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
    return out


def run(exe, *args, timeout=900):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=0")
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=timeout, env=env)
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    return r


def gz_member(payload: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flags=0, extra=b"", name=b"", comment=b"") -> bytes:
    """One gzip member built by hand (RFC 1952), so that header options and deflate strategies can be chosen."""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    body = co.compress(payload) + co.flush()
    hdr = b"\x1f\x8b\x08" + bytes([flags]) + struct.pack("<I", 12345) + b"\x00\x03"
    if flags & 4:
        hdr += struct.pack("<H", len(extra)) + extra
    if flags & 8:
        hdr += name + b"\x00"
    if flags & 16:
        hdr += comment + b"\x00"
    if flags & 2:
        hdr += struct.pack("<H", zlib.crc32(hdr) & 0xffff)
    return hdr + body + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload) & 0xffffffff)


def alignment_text(rng, N, M) -> bytes:
    centre = rng.integers(0, 20, size=N)
    out = []
    for k in range(M):
        s = centre.copy()
        mut = rng.random(N) < 0.15
        s[mut] = rng.integers(0, 21, size=int(mut.sum()))
        out.append(b">seq%d/1-%d\n" % (k, N) + LETTERS[s].tobytes() + b"\n")
    return b"".join(out)


def test_crc32_equals_zlib(bins):
    r = run(bins["asan"], "crc")
    assert r.returncode == 0 and "crc: equal" in r.stdout, r.stdout


def test_every_block_type_and_copy_path_equals_zlib(bins, tmp_path):
    rng = np.random.default_rng(3)
    text = alignment_text(rng, 180, 1500)
    noise = rng.integers(0, 256, size=1 << 20, dtype=np.uint8).tobytes()
    cases = {
        "text_l1": gz_member(text, 1), "text_l6": gz_member(text, 6), "text_l9": gz_member(text, 9),
        "stored": gz_member(text, 0),                                           # block type 0, several blocks (> 64 KB)
        "fixed": gz_member(text, 6, zlib.Z_FIXED),                              # block type 1
        "huffman_only": gz_member(text, 6, zlib.Z_HUFFMAN_ONLY),                # literals only: no distance code at all
        "rle": gz_member(text + b"-" * 5000, 6, zlib.Z_RLE),                    # distance 1 only: one distance code of one bit
        "runs": gz_member(b"A" * 100000 + b"AB" * 40000 + b"ABC" * 30000 + b"ABCDEFG" * 9000, 9),   # distances 1 .. 7, length 258
        "noise": gz_member(noise, 6),                                           # incompressible: long codes, sub-tables, stored blocks
        "far": gz_member(noise[:32768] + text[:3000] + noise[:32768], 9),       # matches at the far end of the 32 KB window
        "skewed": gz_member(bytes(rng.choice(np.arange(256, dtype=np.uint8), size=400000,
                                             p=np.r_[0.9, np.full(255, 0.1 / 255)])), 9),   # 1-bit and 15-bit codes side by side
        "empty": gz_member(b""), "one_byte": gz_member(b"x"),
        "all_header_fields": gz_member(text[:5000], 6, flags=2 | 4 | 8 | 16, extra=b"\x01\x02abcd", name=b"fam.fasta", comment=b"made by a test"),
        "two_members": gz_member(text[:70000], 6) + gz_member(text[70000:], 1, flags=8, name=b"second"),
        "python_gzip": gzip.compress(text, 6),
    }
    paths = []
    for name, blob in cases.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(blob)
        paths.append(str(p))
    paths += [os.path.join(REFDATA, "small.fasta.gz"), os.path.join(REFDATA, "large.fasta.gz")]
    for exe in (bins["plain"], bins["asan"]):
        r = run(exe, "files", *paths)
        print(r.stdout)
        assert r.returncode == 0, r.stdout
        assert "MISMATCH" not in r.stdout and "declines" not in r.stdout and "FAILS" not in r.stdout, r.stdout


def test_damaged_streams_are_left_to_zlib(bins, tmp_path):
    """Truncations, bit flips, random bytes, damaged headers, a truncated second member: never a crash or an out-of-bounds access
    (ASan), and no accepted input whose output differs from zlib's."""
    rng = np.random.default_rng(5)
    text = alignment_text(rng, 120, 400)
    for k, blob in enumerate((gz_member(text, 6), gz_member(text, 6, zlib.Z_FIXED), gz_member(text, 0), gz_member(text[:20000], 9) + gz_member(text[20000:], 1))):
        p = tmp_path / ("f%d.gz" % k)
        p.write_bytes(blob)
        r = run(bins["asan"], "fuzz", 100 + k, 1500, p)
        print(r.stdout.strip())
        assert r.returncode == 0 and "all equal to zlib: yes" in r.stdout, r.stdout + r.stderr[-2000:]
    # inputs that are valid gzip but bad trailers: the fast decoder must decline them (zlib then reports the error)
    good = gz_member(text, 6)
    bad_crc = good[:-8] + struct.pack("<I", (zlib.crc32(text) ^ 1) & 0xffffffff) + good[-4:]
    bad_len = good[:-4] + struct.pack("<I", len(text) + 1)
    for name, blob in (("bad_crc", bad_crc), ("bad_len", bad_len), ("garbage_after", good + b"\x00\x00\x00")):
        p = tmp_path / (name + ".gz")
        p.write_bytes(blob)
        r = run(bins["asan"], "files", p)
        assert r.returncode == 0 and "fast declines" in r.stdout, (name, r.stdout)


def test_one_member_on_several_threads(bins, tmp_path):
    """gdca_gunzip_parallel (speculative block starts, 16-bit symbols for the unknown windows, resolved afterwards) on files big enough
    for it: accepted and equal to zlib at 2, 5 and 16 threads for level 1 / 6 / 9 streams and a stream that mixes in stored and
    fixed-Huffman blocks; declined (never wrong) for two members; under ASan + UBSan and under TSan; and 25 random corruptions of a
    big file under ASan -- whatever the parallel decoder accepts must be zlib's output."""
    rng = np.random.default_rng(21)
    text = alignment_text(rng, 300, 40000)                      # 12.6 MB
    noise = rng.integers(0, 256, size=300000, dtype=np.uint8).tobytes()
    mixed = b"".join(text[k:k + 400000] + noise[:150000] for k in range(0, len(text), 400000))   # incompressible stretches: stored blocks
    cases = {"big_l1": gz_member(text, 1), "big_l6": gz_member(text, 6), "big_l9": gz_member(text, 9), "big_mixed": gz_member(mixed, 6),
             "big_two_members": gz_member(text[:len(text) // 2], 6) + gz_member(text[len(text) // 2:], 6)}
    paths = {}
    for name, blob in cases.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(blob)
        paths[name] = str(p)
    for kind in ("plain", "asan", "tsan"):
        # (under ThreadSanitizer the decoders run at a few MB/s: two files are enough for the thread structure)
        r = run(bins[kind], "files", *(paths.values() if kind != "tsan" else [paths["big_l6"], paths["big_mixed"]]))
        print(r.stdout)
        assert r.returncode == 0 and "MISMATCH" not in r.stdout and "FAILS" not in r.stdout, r.stdout + r.stderr[-3000:]
        blocks = r.stdout.split("\n/")          # (the parallel lines of a file are printed before its own line)
        for name in (("big_l1", "big_l6", "big_l9", "big_mixed") if kind != "tsan" else ("big_l6", "big_mixed")):
            mine = [b for b in r.stdout.split(".gz: zlib") if name in b.splitlines()[-1]]
            assert mine and mine[0].count("parallel x") >= 1 and "ok," in mine[0], (kind, name, r.stdout)
        assert len(blocks) >= 1
    r = run(bins["asan"], "fuzz", 77, 25, paths["big_l6"])
    print(r.stdout.strip())
    assert r.returncode == 0 and "all equal to zlib: yes" in r.stdout, r.stdout + r.stderr[-2000:]


def test_reader_gives_the_same_matrix_for_gz_and_plain(tmp_path):
    """Through the library (no GPU needed for the reader): a .gz file read with the project's decoder, with zlib only
    (GDCA_FASTA_ZLIB=1, a subprocess: the switch is read once) and the plain file give the same Z."""
    import sys

    rng = np.random.default_rng(11)
    text = alignment_text(rng, 97, 3000)
    (tmp_path / "a.fasta").write_bytes(text)
    (tmp_path / "a.fasta.gz").write_bytes(gzip.compress(text, 6))
    (tmp_path / "b.fasta.gz").write_bytes(gz_member(text[:len(text) // 2], 9) + gz_member(text[len(text) // 2:], 1))
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "from gaussdca.jl_amd import dcautils\n"
            "for f in sys.argv[1:]:\n"
            "    Z = dcautils.read_fasta_alignment(f, 0.9)\n"
            "    print(Z.shape, hashlib.sha256(Z.tobytes(order='F')).hexdigest())\n" % ROOT)
    files = [str(tmp_path / n) for n in ("a.fasta", "a.fasta.gz", "b.fasta.gz")]
    outs = []
    for env in ({}, {"GDCA_FASTA_ZLIB": "1"}):
        r = subprocess.run([sys.executable, "-c", code, *files], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        lines = r.stdout.strip().splitlines()
        assert len(lines) == 3 and lines[0] == lines[1] == lines[2], r.stdout
        outs.append(lines[0])
    assert outs[0] == outs[1] and outs[0].startswith("(97, 3000)")
    # a file big enough for the multi-threaded decoder (>= 4 MB compressed, several reader threads): the reader's default against
    # the serial decoder (GDCA_FASTA_SERIAL_INFLATE=1) and zlib, on the synthetic generator's text
    cli = os.path.join(ROOT, "gaussdca.jl_amd", "gdca_cli")
    big = str(tmp_path / "big.fasta.gz")
    subprocess.run([cli, "--synth", "300", "40000", "4660", big], check=True, stdout=subprocess.DEVNULL)
    assert os.path.getsize(big) > (4 << 20)
    seen = set()
    for env in ({"GDCA_FASTA_THREADS": "6"}, {"GDCA_FASTA_THREADS": "6", "GDCA_FASTA_SERIAL_INFLATE": "1"}, {"GDCA_FASTA_ZLIB": "1"}):
        r = subprocess.run([sys.executable, "-c", code, big], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, GDCA_INFLATE_TRACE="1", **env))
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("inflate-trace" in r.stderr) == (len(env) == 1 and "GDCA_FASTA_THREADS" in env), (env, r.stderr[-500:])   # the path really taken
        seen.add(r.stdout.strip())
    assert len(seen) == 1 and next(iter(seen)).startswith("(300, 40000)"), seen
