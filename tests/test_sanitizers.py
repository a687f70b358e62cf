"""CPU sanitizer runs of the product's HOST code: csrc/gdca_host.cpp (threaded FASTA(.gz) reader, duplicate
removal, ranking sort, writers) and csrc/gdca_cli.cpp (the batch mode's parser -> worker -> writer queues) built with
AddressSanitizer + UndefinedBehaviorSanitizer and, separately, ThreadSanitizer (`make -C gaussdca.jl_amd/csrc asan tsan`)
against tests/sanitize/gdca_stub.cpp, which stands in for the GPU entry points (test infrastructure: libgdca.so never contains
it).  Every scenario must finish with the expected exit code and without a sanitizer report."""
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussdca.jl_amd", "csrc")
BUILD = os.path.join(ROOT, "tests", "_build")
LETTERS = "ACDEFGHIKLMNPQRSTVWY-"


@pytest.fixture(scope="module")
def bins():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "-C", CSRC, "asan", "tsan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return {"asan": os.path.join(BUILD, "gdca_cli_asan"), "tsan": os.path.join(BUILD, "gdca_cli_tsan")}


def run(exe, *args, env=None, timeout=600):
    e = dict(os.environ)
    e.update({"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1",
              "TSAN_OPTIONS": "halt_on_error=0:second_deadlock_stack=1"})
    e.update(env or {})
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=timeout, env=e)
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    return r


def write_family(path, rng, N, M, gz=False, crlf=False, wrap=0, dups=0):
    seqs = ["".join(LETTERS[a] for a in rng.integers(0, 21, size=N)) for _ in range(M)]
    for d in range(dups):
        seqs.append(seqs[d])
    nl = "\r\n" if crlf else "\n"
    out = []
    for k, s in enumerate(seqs):
        out.append(">s%d some description" % k + nl)
        if wrap:
            out.extend(s[a:a + wrap] + nl for a in range(0, N, wrap))
        else:
            out.append(s + nl)
    data = "".join(out).encode()
    with (gzip.open(path, "wb") if gz else open(path, "wb")) as f:
        f.write(data)


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_single_family_paths(bins, tmp_path, san):
    exe = bins[san]
    rng = np.random.default_rng(1)
    fam = tmp_path / "fam.fasta.gz"
    assert run(exe, "--synth", 37, 240, "0xE003", fam).returncode == 0          # generator + gz writer
    out = tmp_path / "rank.txt"
    r = run(exe, "--remove_dups", "--min_separation", 3, fam, out)                 # reader, dedup, stub run, ranking, writer
    assert r.returncode == 0, r.stderr
    assert len(out.read_text().splitlines()) == (37 - 3) * (37 - 2) // 2
    assert run(exe, "--pseudocount", 0, fam).returncode == 1                      # the PosDefException path
    # wrapped lines, CRLF, duplicates, a trailing record without newline
    odd = tmp_path / "odd.fasta"
    write_family(odd, rng, 23, 50, crlf=True, wrap=7, dups=5)
    with open(odd, "ab") as f:
        f.write(b">last\n" + b"A" * 23)
    r = run(exe, "--remove_dups", odd, tmp_path / "odd.txt")
    assert r.returncode == 0 and "M = 51" in r.stderr, r.stderr                   # 50 + 5 duplicates removed + the last record
    # error paths: ragged record, empty file, insert columns that differ between records, a directory
    bad = tmp_path / "bad.fasta"
    bad.write_text(">a\nACDEF\n>b\nACD\n")
    assert run(exe, bad).returncode == 2
    (tmp_path / "empty.fasta").write_text("")
    assert run(exe, tmp_path / "empty.fasta").returncode == 2
    ins = tmp_path / "ins.fasta"
    ins.write_text(">a\nAC.dEF\n>b\nACg.EF\n>c\nA.CdEF\n")                      # third record: different match columns
    assert run(exe, ins).returncode == 2
    ins.write_text(">a\nAC.dEF\n>b\nACg.EF\n>c\nAC.wEF\n")
    assert run(exe, ins, tmp_path / "ins.txt").returncode == 0
    assert run(exe, tmp_path).returncode == 2


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_batch_queues(bins, tmp_path, san):
    """Parser threads -> bounded queue -> GPU workers (2 stub devices x 2 contexts in flight) -> writer threads, over a
    directory of 14 families of mixed sizes and encodings, one unreadable file and one non-FASTA file."""
    exe = bins[san]
    rng = np.random.default_rng(2)
    indir, outdir = tmp_path / "in", tmp_path / "out"
    indir.mkdir()
    for f in range(14):
        N, M = int(rng.integers(8, 60)), int(rng.integers(5, 400))
        write_family(indir / ("fam%02d.fasta%s" % (f, ".gz" if f % 3 == 0 else "")), rng, N, M, gz=f % 3 == 0, wrap=(0, 11)[f % 2])
    big = indir / "big.fasta"                                                    # > 1 MiB: the threaded reader path
    write_family(big, rng, 150, 9000)
    (indir / "broken.fasta").write_text(">a\nACDEF\n>b\nAC\n")
    (indir / "notes.txt").write_text("not an alignment")
    env = {"GDCA_STUB_DEVICES": "2", "GDCA_FASTA_THREADS": "4"}
    r = run(exe, "--batch", indir, "--out", outdir, "--parsers", 3, "--inflight", 2, env=env)
    assert r.returncode == 1 and "16 families" in r.stderr and "(1 failed)" in r.stderr, r.stderr[-2000:]
    assert len(os.listdir(outdir)) == 15
    # the same directory on one device, device list given explicitly: identical ranking files
    outdir2 = tmp_path / "out2"
    r = run(exe, "--batch", indir, "--out", outdir2, "--parsers", 2, "--inflight", 1,
            env=dict(env, GDCA_VISIBLE_DEVICES="1"))
    assert r.returncode == 1 and "gpu 1:" in r.stderr and "gpu 0:" not in r.stderr
    for nm in os.listdir(outdir):
        assert (outdir / nm).read_bytes() == (outdir2 / nm).read_bytes(), nm
    # ... and with the small families in merged batches of up to four (two sets of contexts taking turns beside the slots' pipeline)
    outdir3 = tmp_path / "out3"
    r = run(exe, "--batch", indir, "--out", outdir3, "--parsers", 3, "--inflight", 2, "--merge", 4, "--merge-blocks", 6, env=env)
    assert r.returncode == 1 and "16 families" in r.stderr and "(1 failed)" in r.stderr, r.stderr[-2000:]
    for nm in os.listdir(outdir):
        assert (outdir / nm).read_bytes() == (outdir3 / nm).read_bytes(), nm
    # parse-only (the host feed rate benchmark) and the start-up failure paths
    r = run(exe, "--batch", indir, "--parse-only", "--parsers", 4, env=env)
    assert r.returncode == 1 and "parse-only: 16 families" in r.stderr
    assert run(exe, "--batch", indir, "--out", outdir2, env=dict(env, GDCA_VISIBLE_DEVICES="0,7")).returncode == 2
    assert run(exe, "--batch", indir, "--out", outdir2, env=dict(env, GDCA_VISIBLE_DEVICES="0,0")).returncode == 2
    r = run(exe, "--batch", indir, "--out", outdir2, env=dict(env, GDCA_STUB_DEVICES="1", GDCA_STUB_FAIL_DEVICE="0"))
    assert r.returncode == 1 and "no GPU worker could start" in r.stderr        # parsers are released, nothing hangs
