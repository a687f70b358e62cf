"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/gdca.h declares, and fails loudly (no CPU fallback) when no GPU is present.  No compute."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import host_mirrors as hm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gaussdca.jl_amd", "libgdca.so")


@pytest.fixture(scope="module")
def lib_path():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "gaussdca.jl_amd", "csrc"), "-j8"], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "gdca.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gdca_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(lib_path):
    names = _declared_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(lib_path)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gdca.h but not exported by libgdca.so"


def test_python_binding_covers_the_header(lib_path):
    from gaussdca.jl_amd import _lib

    assert sorted(_lib.SYMBOLS) == _declared_symbols()
    lib = _lib.load()
    assert lib.gdca_version() == 6 == _lib.ABI_VERSION
    assert lib.gdca_stats_bytes() == ctypes.sizeof(_lib.Stats) and lib.gdca_params_bytes() == ctypes.sizeof(_lib.Params)
    assert ctypes.sizeof(_lib.Stats) == 8 * 3 + 4 * 10 + 8 * 15 + 4 * 2  # ten int32, fifteen more doubles, two int32
    assert ctypes.sizeof(_lib.Params) == 24


def _c_layout(tmp_path):
    """sizeof/offsetof as gcc sees include/gdca.h (tests/abi/abi_layout.c)."""
    exe = tmp_path / "abi_layout"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi", "abi_layout.c"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    lay = {}
    for line in out.splitlines():
        st, f, off, size = line.split()
        lay.setdefault(st, []).append((f, int(off), int(size)))
    return lay


_JL_SIZES = {"Cdouble": 8, "Float64": 8, "UInt64": 8, "Int64": 8, "Int32": 4, "UInt32": 4, "Cint": 4}


def _julia_struct(name):
    """Field list of `struct name ... end` in julia/src/GaussDCAHip.jl with the offsets Julia's C-compatible
    layout gives it (natural alignment, declaration order)."""
    src = open(os.path.join(ROOT, "julia", "src", "GaussDCAHip.jl")).read()
    body = re.search(r"struct\s+" + name + r"\b(.*?)\n\s*end", src, flags=re.S).group(1)
    body = re.sub(r"#.*", "", body)
    fields, off = [], 0
    for fname, ftype in re.findall(r"(\w+)::(\w+)", body):
        size = _JL_SIZES[ftype]
        off = (off + size - 1) // size * size
        fields.append((fname, off, size))
        off += size
    total = (off + 7) // 8 * 8
    return fields, total


def test_struct_layouts_agree_across_c_ctypes_and_julia(tmp_path):
    """gdca_params / gdca_stats: the compiler's layout == the ctypes mirror == the Julia mirror, field for
    field; status codes as the Julia `check` and the Python `Context.check` decode them."""
    from gaussdca.jl_amd import _lib

    lay = _c_layout(tmp_path)
    for cname, ct, jl in (("gdca_params", _lib.Params, "GdcaParams"), ("gdca_stats", _lib.Stats, "GdcaStats")):
        rows = lay[cname]
        assert rows[0][0] == "." and rows[0][2] == ctypes.sizeof(ct)
        c_fields = rows[1:]
        py_fields = [(n, getattr(ct, n).offset, getattr(ct, n).size) for n, _ in ct._fields_]
        assert py_fields == c_fields, cname
        jl_fields, jl_total = _julia_struct(jl)
        assert jl_fields == c_fields, jl
        assert jl_total == rows[0][2]
    codes = {f: off for f, off, _ in lay["status"]}
    assert codes == dict(GDCA_OK=_lib.GDCA_OK, GDCA_EINVAL=_lib.GDCA_EINVAL, GDCA_ENOTPD=_lib.GDCA_ENOTPD,
                         GDCA_EHIP=_lib.GDCA_EHIP, GDCA_ENOMEM=_lib.GDCA_ENOMEM, GDCA_ENOCONV=_lib.GDCA_ENOCONV)
    jl = open(os.path.join(ROOT, "julia", "src", "GaussDCAHip.jl")).read()
    for code, exc in ((1, "ArgumentError"), (2, "PosDefException"), (4, "OutOfMemoryError"), (5, r"LinearAlgebra\.LAPACKException")):
        assert re.search(r"st == %d && throw\(%s" % (code, exc), jl), (code, exc)


def _header_prototypes():
    """name -> (return type, [argument types]) of every function include/gdca.h declares, C types normalised
    ('const int8_t *', 'gdca_ctx **', 'double', ...)."""
    src = open(os.path.join(ROOT, "include", "gdca.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[ \*]+)(gdca_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)

        def norm(t):
            t = re.sub(r"\s+", " ", t.strip())
            t = re.sub(r"\s*\*\s*", " *", t)         # 'T * const *' -> 'T *const *'
            t = t.replace("* *", "**").replace("*const *", "**").replace(" const *", " *")
            return t.strip()

        alist = []
        for a in [x for x in args.split(",") if x.strip() and x.strip() != "void"]:
            a = a.strip()
            a = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", a).strip() if not a.endswith("*") else a   # drop the name
            alist.append(norm(a))
        protos[name] = (norm(ret), alist)
    return protos


# C type (as _header_prototypes normalises it) -> the Julia ccall types that bind it correctly
_JL_OK = {
    "gdca_status": {"Cint"}, "int32_t": {"Int32", "Cint"}, "int64_t": {"Int64"}, "uint64_t": {"UInt64"},
    "double": {"Cdouble", "Float64"}, "const char *": {"Cstring", "Ptr{UInt8}"}, "void *": {"Ptr{Cvoid}"},
    "const void *": {"Ptr{Cvoid}"},
    "gdca_ctx *": {"Ptr{Cvoid}"}, "gdca_ctx **": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "gdca_dbuf *": {"Ptr{Cvoid}"}, "const gdca_dbuf *": {"Ptr{Cvoid}"}, "gdca_dbuf **": {"Ref{Ptr{Cvoid}}"},
    "const int8_t *": {"Ptr{Int8}", "Ptr{Cvoid}"}, "int8_t *": {"Ptr{Int8}", "Ptr{Cvoid}"},
    "const double *": {"Ptr{Float64}", "Ptr{Cdouble}", "Ptr{Cvoid}"},
    "double *": {"Ptr{Float64}", "Ptr{Cdouble}", "Ref{Cdouble}", "Ref{Float64}", "Ptr{Cvoid}"},
    "int32_t *": {"Ref{Int32}", "Ptr{Int32}", "Ptr{Cvoid}"}, "uint64_t *": {"Ref{UInt64}", "Ptr{UInt64}"},
    "const gdca_params *": {"Ref{GdcaParams}", "Ptr{GdcaParams}"}, "gdca_stats *": {"Ref{GdcaStats}", "Ptr{GdcaStats}"},
}


def _split_top(s):
    """split on commas that are not nested inside (), {} or []"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def test_julia_ccalls_match_the_header():
    """Every `ccall((:sym, libgdca), Ret, (ArgTypes...), args...)` of julia/src/GaussDCAHip.jl against the prototype of
    `sym` in include/gdca.h: the symbol exists, return type and every argument type bind the C type, and the number of
    values passed equals the number of declared arguments.  (The shim cannot be executed in this image -- no Julia --
    so its signatures are checked here; its structs are checked by the layout test above.)  Also: the exported surface
    is the reference's (src/GaussDCA.jl:3: gDCA, printrank) plus the DCAUtils-named operators of :28-39, and the package
    file names the reference's dependencies."""
    protos = _header_prototypes()
    assert len(protos) == len(_declared_symbols())
    jl = open(os.path.join(ROOT, "julia", "src", "GaussDCAHip.jl")).read()
    jl_nc = re.sub(r"#[^\n]*", "", jl)
    calls = 0
    for m in re.finditer(r"ccall\(\(:(gdca_[a-z0-9_]+), libgdca\),", jl_nc):
        sym = m.group(1)
        # the balanced argument list of this ccall
        i, depth, start = m.end(), 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(jl_nc[i], 0)
            i += 1
        parts = _split_top(jl_nc[start:i - 1])
        ret, argt = parts[0], parts[1]
        values = parts[2:]
        assert argt.startswith("(") and argt.endswith(")"), (sym, argt)
        types = _split_top(argt[1:-1])
        assert sym in protos, sym + " is not declared in include/gdca.h"
        c_ret, c_args = protos[sym]
        assert ret in _JL_OK[c_ret], (sym, "return", ret, c_ret)
        assert len(types) == len(c_args) == len(values), (sym, types, c_args, values)
        for k, (jt, ct) in enumerate(zip(types, c_args)):
            assert ct in _JL_OK, (sym, k, ct)
            assert jt in _JL_OK[ct], (sym, "argument %d" % k, jt, ct)
        calls += 1
    assert calls >= 20
    exported = re.search(r"^export (.*?)\n\n", jl, flags=re.S | re.M).group(1).replace("\n", " ")
    names = {x.strip() for x in exported.split(",")}
    assert {"gDCA", "printrank", "compute_weights", "compute_weighted_frequencies", "add_pseudocount", "compute_FN",
            "compute_DI_gauss"} <= names
    proj = open(os.path.join(ROOT, "julia", "Project.toml")).read()
    ref_deps = {"DCAUtils": "e41cd558-3099-4f6e-a65d-5336857e40aa", "LinearAlgebra": "37e2e46d-f89d-539d-b4ee-838fcccc9c8e",
                "GaussDCA": "bc09176e-aa47-5a77-a802-92a0219fe3db"}     # uuids of /root/reference/Project.toml:2,6-7
    for name, uuid in ref_deps.items():
        assert re.search(r'^%s = "%s"$' % (name, uuid), proj, flags=re.M), name


def test_no_cpu_fallback(lib_path):
    """Without a GPU every product entry point must raise; nothing may route through the oracle."""
    import gaussdca.jl_amd as g

    lib = g.load()
    if lib.gdca_device_count() > 0:
        pytest.skip("a HIP device is visible here")
    with pytest.raises(g.GdcaError):
        g.Context(0)
    with pytest.raises(g.GdcaError):
        g.compute_weights(np.ones((4, 5), dtype=np.int8), 21, 0.2)
    h = ctypes.c_void_p()
    assert lib.gdca_ctx_create(0, ctypes.byref(h)) != 0 and not h.value


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gaussdca.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "gdca_oracle" not in txt, f


def test_argument_checks_mirror_reference(tmp_path):
    """check_arguments: same order and conditions as src/GaussDCA.jl:49-65 (host side, no GPU)."""
    import gaussdca.jl_amd as g

    f = os.path.join(ROOT, "tests", "golden", "reference", "small.fasta.gz")
    assert g.check_arguments(f, 0.8, ":auto", 0.9, ":frob", 5)
    assert g.check_arguments(f, 0.2, 0.0, 0.8, "DI", 4)
    for args in [(f, 1.5, ":auto", 0.9, ":frob", 5), (f, 0.8, -0.1, 0.9, ":frob", 5),
                 (f, 0.8, ":bogus", 0.9, ":frob", 5), (f, 0.8, ":auto", 2.0, ":frob", 5),
                 (f, 0.8, ":auto", 0.9, ":plm", 5), (f, 0.8, ":auto", 0.9, ":frob", 0),
                 (str(tmp_path / "missing.fasta"), 0.8, ":auto", 0.9, ":frob", 5)]:
        with pytest.raises(g.ArgumentError):
            g.check_arguments(*args)


def test_host_fasta_dedup_ranking_match_oracle(refdata, tmp_path):
    """The host-side pieces around the hot path (FASTA reader, dedup, ranking, printrank): the native
    (libgdca.so, C++) forms, their pure-Python statements and the oracle agree."""
    import io

    import gaussdca.jl_amd as g
    from oracle import gdca_oracle as o

    for name, mgf in (("small.fasta.gz", 0.9), ("large.fasta.gz", 0.9), ("small.fasta.gz", 0.8),
                      ("large.fasta.gz", 0.84)):
        p = os.path.join(refdata, name)
        Z = g.read_fasta_alignment(p, mgf)           # (N, M), Fortran order, native reader
        Zo = o.read_fasta_alignment(p, mgf)          # (M, N), C order
        assert Z.flags.f_contiguous and Z.dtype == np.int8
        assert np.array_equal(Z.T, Zo)
        assert np.array_equal(hm.read_fasta_alignment_py(p, mgf), Z)
        Zu, idx = g.remove_duplicate_sequences(Z)
        Zou, idxo = o.remove_duplicate_sequences(Zo)
        assert np.array_equal(Zu.T, Zou) and np.array_equal(idx, idxo + 1)
        Zp, idxp = hm.remove_duplicate_sequences_py(Z)
        assert np.array_equal(Zp, Zu) and np.array_equal(idxp, idx)
    # 102 -> 97 sequences at 0.9 (five all-gap sequences), 94 after dedup
    Zl = g.read_fasta_alignment(os.path.join(refdata, "large.fasta.gz"), 0.9)
    assert Zl.shape == (400, 97) and g.remove_duplicate_sequences(Zl)[0].shape == (400, 94)

    # untested-by-the-reference corners, pinned to the restated rules: insert columns ('.' / lowercase) of the
    # first record are dropped, letters BJOUXZ and '*' map to 21, CRLF and wrapped lines, gap filter with '<='
    fa = tmp_path / "odd.fasta"
    fa.write_bytes(b">s1 first\r\nAC.dE-\r\nGH\r\n>s2\nBJxyOU\nXZ\n>s3\n--.a-A\n--\n\n>s4\nWY.tAC\nGT\n")
    for mgf in (0.9, 0.5, 0.49):
        Zn = g.read_fasta_alignment(str(fa), mgf)
        assert np.array_equal(Zn, hm.read_fasta_alignment_py(str(fa), mgf))
        assert np.array_equal(Zn.T, o.read_fasta_alignment(str(fa), mgf))
    Zn = g.read_fasta_alignment(str(fa), 0.9)
    assert Zn.shape[0] == 6 and Zn[:, 0].tolist() == [1, 2, 4, 21, 6, 7] and Zn[:, 1].tolist() == [21] * 6
    with pytest.raises(ValueError):
        g.read_fasta_alignment(str(tmp_path / "missing.fasta"), 0.9)
    bad = tmp_path / "ragged.fasta"
    bad.write_text(">a\nACD\n>b\nAC\n")
    with pytest.raises(ValueError):
        g.read_fasta_alignment(str(bad), 0.9)
    # "inconsistent inputs": a later record whose match columns (neither '.' nor lowercase) differ from the first
    # record's -- same length, so only the per-record column check catches it (DCAUtils errors here too)
    for body in ("AC.dE\nAC.DE\n", "AC.dE\nac.dE\n", "ACDE\nAC.E\n"):
        a, b = body.split("\n")[:2]
        inc = tmp_path / "inconsistent.fasta"
        inc.write_text(">a\n%s\n>b\n%s\n" % (a, b))
        for reader in (g.read_fasta_alignment, hm.read_fasta_alignment_py, o.read_fasta_alignment):
            with pytest.raises(ValueError):
                reader(str(inc), 0.9)
    ok = tmp_path / "consistent.fasta"
    ok.write_text(">a\nAC.dE\n>b\nWY.-K\n")   # '.' <-> lowercase may differ between records?  No: position 3 is a match
    with pytest.raises(ValueError):               # column in b ('-' is neither '.' nor lowercase) but not in a
        g.read_fasta_alignment(str(ok), 0.9)
    ok.write_text(">a\nAC.dE\n>b\nWYk.K\n")     # '.' and lowercase are interchangeable in insert columns
    assert g.read_fasta_alignment(str(ok), 0.9).T.tolist() == [[1, 2, 4], [19, 20, 9]]
    assert np.array_equal(g.read_fasta_alignment(str(ok), 0.9), hm.read_fasta_alignment_py(str(ok), 0.9))

    # duplicate removal with NON-ADJACENT duplicates, out of place and in place (Z_out aliasing Z): the in-place
    # compaction must not invalidate the keys of rows it has already kept
    Zd = np.asfortranarray(np.array([list(b"AABCBCD")], dtype=np.int8).repeat(4, axis=0))   # rows A A B C B C D
    Zu, idx = g.remove_duplicate_sequences(Zd)
    assert Zu.shape == (4, 4) and idx.tolist() == [1, 3, 4, 7] and Zu[0].tolist() == list(b"ABCD")
    lib = g.load()
    rng2 = np.random.default_rng(11)
    for trial in range(20):
        M0, N0 = int(rng2.integers(1, 200)), int(rng2.integers(1, 9))
        Zr = np.ascontiguousarray(rng2.integers(1, 4, size=(M0, N0)).astype(np.int8))      # many repeats
        want, keep = o.remove_duplicate_sequences(Zr)
        for alias in (False, True):
            src = Zr.copy()
            dst = src if alias else np.zeros_like(src)
            m = ctypes.c_int32()
            kidx = np.zeros(M0, dtype=np.int32)
            assert lib.gdca_remove_duplicates(src.ctypes.data, N0, M0, dst.ctypes.data, kidx.ctypes.data,
                                              ctypes.byref(m)) == 0
            assert m.value == want.shape[0] and np.array_equal(dst[:m.value], want), (trial, alias)
            assert np.array_equal(kidx[:m.value], keep + 1)

    rng = np.random.default_rng(0)
    S = rng.random((30, 30))
    S = S + S.T
    S[3, 20] = S[20, 3] = S[4, 25] = S[25, 4] = 0.123  # an exact tie: generation order must be kept
    for sep in (1, 4, 5, 29, 30, 31):
        assert g.compute_ranking(S, sep) == o.compute_ranking(S, sep) == hm.compute_ranking_py(S, sep)
    buf = io.StringIO()
    R = [(11, 35, 3.649475), (9, 46, -0.6752293), (1, 2, 1e-300), (3, 4, 123456789.0)]
    g.printrank(buf, R)
    assert buf.getvalue().startswith("11 35 3.649475e+00\n9 46 -6.752293e-01\n")
    out = tmp_path / "rank.txt"
    g.printrank(str(out), R)  # native writer: same bytes as the Python "%i %i %e"
    assert out.read_text() == buf.getvalue()


def test_ranking_order_is_julias_isless_reversed():
    """sort!(R, by = x -> x[3], rev = true) (src/GaussDCA.jl:96): stable, NaN first, 0.0 before -0.0, exact ties in
    generation order -- native radix sort, the numpy statement and the oracle agree, and the lazy Ranking behaves
    like the list of tuples the reference returns."""
    import gaussdca.jl_amd as g
    from oracle import gdca_oracle as o

    rng = np.random.default_rng(3)
    N = 60
    S = rng.standard_normal((N, N))
    S = S + S.T
    for (a, b, v) in [(3, 40, np.nan), (7, 50, -0.0), (8, 55, 0.0), (9, 30, np.inf), (10, 44, -np.inf), (11, 45, np.nan)]:
        S[a, b] = S[b, a] = v
    S[20:25, 40:48] = 0.5
    S[40:48, 20:25] = 0.5
    R = g.compute_ranking(S, 5)
    Rp = hm.compute_ranking_py(S, 5)
    Ro = o.compute_ranking(S, 5)
    same = lambda x, y: x == y or (x != x and y != y)  # noqa: E731
    assert len(R) == len(Rp) == len(Ro) == (N - 5) * (N - 4) // 2
    for a, b, c in zip(R, Rp, Ro):
        assert a[:2] == b[:2] == c[:2] and same(a[2], b[2]) and same(a[2], c[2])
    assert np.isnan(R[0][2]) and np.isnan(R[1][2]) and R[2][2] == np.inf and R[len(R) - 1][2] == -np.inf
    zeros = [t for t in range(len(R)) if R[t][2] == 0]
    assert [np.signbit(R[t][2]) for t in zeros] == [False, True]
    ties = [R[t][:2] for t in range(len(R)) if R[t][2] == 0.5]
    assert ties == sorted(ties) and len(ties) == 40
    # list-like behaviour
    assert isinstance(R[2:5], list) and R[2:5] == [R[2], R[3], R[4]] == Rp[2:5]      # (NaN != NaN: skip the first two)
    assert R[5] == tuple(R[5]) and len(list(iter(R))) == len(R)
    Rf = g.compute_ranking(np.where(np.isnan(S), 1.0, S), 5)
    assert Rf == list(Rf) and list(Rf) == hm.compute_ranking_py(np.where(np.isnan(S), 1.0, S), 5)
    assert g.compute_ranking(S, N) == [] and len(g.compute_ranking(S, N - 1)) == 1


def test_threaded_fasta_reader_on_a_large_file(tmp_path):
    """Files over 1 MB take the multi-threaded path of gdca_fasta_open (header scan per chunk, records in parallel,
    order-preserving compaction): wrapped lines, CRLF, blank lines, an insert column, filtered (gap-rich) records and
    a misaligned record in the middle, against the pure-Python statement of the same rules and the oracle."""
    import gaussdca.jl_amd as g
    from gaussdca.jl_amd import synth
    from oracle import gdca_oracle as o

    N, M = 150, 9000
    Z = synth.synth_family(N, M, 21, 77)
    letters = np.frombuffer(b"?ACDEFGHIKLMNPQRSTVWY-", dtype=np.uint8)
    rng = np.random.default_rng(5)
    gappy = set(rng.choice(M, size=300, replace=False).tolist())
    lines = []
    for k in range(M):
        row = Z[k].copy()
        if k in gappy:
            row[: int(0.95 * N)] = 21                       # 95 % gaps: dropped at max_gap_fraction 0.9
        seq = letters[row].tobytes()
        seq = seq[:70] + (b"a" if k % 2 else b".") + seq[70:]   # an insert column after position 70
        eol = b"\r\n" if k % 3 == 0 else b"\n"
        lines.append(b">seq%d some description" % k + eol)
        for a in range(0, len(seq), 60):                    # wrapped at 60 characters
            lines.append(seq[a:a + 60] + eol)
        if k % 7 == 0:
            lines.append(eol)
    path = tmp_path / "big.fasta"
    path.write_bytes(b"".join(lines))
    assert path.stat().st_size > (1 << 20)
    Zn = g.read_fasta_alignment(str(path), 0.9)
    keep = np.array([k not in gappy for k in range(M)])
    assert Zn.shape == (N, int(keep.sum()))
    Zexp = Z.copy()
    for k in gappy:
        Zexp[k, : int(0.95 * N)] = 21
    assert np.array_equal(Zn.T, Zexp[keep])
    assert np.array_equal(Zn.T, o.read_fasta_alignment(str(path), 0.9))
    assert np.array_equal(g.read_fasta_alignment(str(path), 1.0).T, Zexp)
    # a record of the wrong length deep inside the file: "inputs are not aligned"
    bad = tmp_path / "bad.fasta"
    cut = len(lines) // 2
    bad.write_bytes(b"".join(lines[:cut]) + b">short\nACDEF\n" + b"".join(lines[cut:]))
    with pytest.raises(ValueError):
        g.read_fasta_alignment(str(bad), 0.9)
