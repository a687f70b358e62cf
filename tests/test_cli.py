"""gdca_cli (csrc/gdca_cli.cpp): the command-line driver over the C-ABI -- argument checks with the
reference's messages (src/GaussDCA.jl:49-65), the synthetic-FASTA writer, and on the GPU the reference's
golden cases and the directory-batch mode (one worker per GPU, parser threads in front)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gdca_testutil import CASES, compare_with_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "gaussdca.jl_amd", "gdca_cli")


def run(*args, **kw):
    return subprocess.run([CLI, *map(str, args)], capture_output=True, text=True, timeout=600, **kw)


def test_cli_is_built_and_prints_usage():
    assert os.path.exists(CLI), "gdca_cli missing: run __graft_entry__.build()"
    r = run("--help")
    assert r.returncode == 0 and "--batch DIR" in r.stdout


@pytest.mark.parametrize("args,msg", [
    (["--pseudocount", "1.5", "x.fasta"], "invalid pseudocount value: 1.5 (must be between 0 and 1)"),
    (["--theta", "1.5", "x.fasta"], "invalid theta value: 1.5 (must be either :auto, or a number between 0 and 1)"),
    (["--max_gap_fraction", "-1", "x.fasta"], "invalid max_gap_fraction value: -1 (must be between 0 and 1)"),
    (["--score", "foo", "x.fasta"], "invalid score value: foo (must be either :DI or :frob)"),
    (["--min_separation", "0", "x.fasta"], "invalid min_separation value: 0 (must be >= 1)"),
    (["/nonexistent/x.fasta"], "cannot open file /nonexistent/x.fasta"),
])
def test_argument_errors_mirror_the_reference(args, msg):
    r = run(*args)
    assert r.returncode == 2
    assert msg in r.stderr


def test_synth_writes_the_generator_family(tmp_path):
    from gaussdca.jl_amd import dcautils, synth

    path = tmp_path / "fam.fasta.gz"
    assert run("--synth", 45, 130, "0xE001", path).returncode == 0
    Z = dcautils.read_fasta_alignment(str(path), 1.0)
    assert np.array_equal(np.ascontiguousarray(Z.T), synth.synth_family(45, 130, 21, 0xE001))


def _read_rank(path):
    R = []
    with open(path) as f:
        for line in f:
            i, j, s = line.split()
            R.append((int(i), int(j), float(s)))
    return R


def _flags(kw):
    out = []
    for k, v in kw.items():
        if k == "remove_dups":
            out += ["--remove_dups"] if v else []
        else:
            out += ["--" + k, v]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("golden", [k for k in CASES if not k.startswith("large")])
def test_cli_reproduces_reference_goldens(golden, refdata, tmp_path):
    c = CASES[golden]
    out = tmp_path / "rank.txt"
    r = run(*_flags(c["kw"]), os.path.join(refdata, c["fasta"]), out)
    assert r.returncode == 0, r.stderr
    rep = compare_with_golden(_read_rank(out), os.path.join(refdata, golden))
    assert rep["keys_equal"] and rep["order_equal"], rep
    assert rep["max_rel"] <= 1e-6, rep                      # north_star bar for scores
    assert rep["string_mismatches"] <= 3, rep


@pytest.mark.gpu
def test_cli_batch_directory_equals_python_gdca(refdata, tmp_path):
    """Batch mode: every family's ranking file is byte-identical to printrank(gDCA(file)) from the Python mirror
    (same library, same device arithmetic: deterministic), whatever worker picked it up."""
    import shutil

    import gaussdca.jl_amd as g
    from gaussdca.jl_amd import synth

    indir, outdir = tmp_path / "in", tmp_path / "out"
    indir.mkdir()
    shutil.copy(os.path.join(refdata, "small.fasta.gz"), indir / "small.fasta.gz")
    for f, (N, M) in enumerate([(40, 600), (75, 1500), (33, 900), (120, 2500)]):
        synth.write_fasta(str(indir / f"fam{f}.fasta"), synth.synth_family(N, M, 21, 0xE000 + f))
    (indir / "notes.txt").write_text("not an alignment")
    r = run("--pseudocount", 0.2, "--score", "DI", "--batch", indir, "--out", outdir, "--parsers", 2)
    assert r.returncode == 0, r.stderr
    assert "5 families" in r.stderr
    names = sorted(os.listdir(outdir))
    assert names == ["fam0.rank.txt", "fam1.rank.txt", "fam2.rank.txt", "fam3.rank.txt", "small.rank.txt"]
    for nm in names:
        src = [p for p in os.listdir(indir) if p.startswith(nm.split(".")[0] + ".")][0]
        R = g.gDCA(str(indir / src), pseudocount=0.2, score="DI")
        want = tmp_path / "want.txt"
        g.printrank(str(want), R)
        assert (outdir / nm).read_text() == want.read_text(), nm


@pytest.mark.gpu
def test_cli_reports_not_positive_definite(tmp_path):
    from gaussdca.jl_amd import synth

    Z = synth.synth_family(30, 40, 21, 5)
    synth.write_fasta(str(tmp_path / "tiny.fasta"), Z)
    r = run("--pseudocount", 0, tmp_path / "tiny.fasta")    # pc = 0 on 40 sequences: C is singular
    assert r.returncode == 1
    assert "PosDefException" in r.stderr
