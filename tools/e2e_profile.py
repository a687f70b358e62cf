"""Where the wall time of gDCA(filename) goes at config C / D (FASTA file -> ranking): parse, copies, upload + hot path + download,
ranking.   python tools/e2e_profile.py [C|D] [reps] [gz]      (gz: the same family as a .fasta.gz, gDCA() only)"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gaussdca.jl_amd as g
from gaussdca.jl_amd import dcautils, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
as_gz = len(sys.argv) > 3 and sys.argv[3] == "gz"
N, M, seed = (500, 50000, 0xC500) if cfg == "C" else (1000, 100000, 0xD1000)
Zh = synth.synth_family(N, M, 21, seed)
letters = np.frombuffer(b"?ACDEFGHIKLMNPQRSTVWY-", dtype=np.uint8)
with tempfile.NamedTemporaryFile("wb", suffix=".fasta", delete=False) as f:
    path = f.name
    for k in range(Zh.shape[0]):
        f.write(b">s%d\n" % k)
        f.write(letters[Zh[k]].tobytes())
        f.write(b"\n")
ctx = g.Context(0)
if as_gz:
    import gzip
    import shutil

    gz = path + ".gz"
    with open(path, "rb") as fi, gzip.open(gz, "wb", compresslevel=6) as fo:
        shutil.copyfileobj(fi, fo)
    try:
        for r in range(reps):
            t0 = time.perf_counter()
            R = g.gDCA(gz, ctx=ctx)
            print("rep %d: gDCA(%s.fasta.gz, %d MB compressed) %.1f ms" % (r, cfg, os.path.getsize(gz) >> 20, (time.perf_counter() - t0) * 1e3), flush=True)
    finally:
        os.unlink(gz)
        os.unlink(path)
    sys.exit(0)
try:
    for r in range(reps):
        t0 = time.perf_counter()
        R = g.gDCA(path, ctx=ctx)
        t_all = time.perf_counter() - t0
        t0 = time.perf_counter()
        Z = dcautils.read_fasta_alignment(path, 0.9)
        t_parse = time.perf_counter() - t0
        t0 = time.perf_counter()
        q = int(Z.max())
        t_max = time.perf_counter() - t0
        t0 = time.perf_counter()
        S, st = ctx.run(np.asfortranarray(Z), q, 0.8, -1.0, 0)
        t_run = time.perf_counter() - t0
        t0 = time.perf_counter()
        R2 = dcautils.compute_ranking(S, 5)
        t_rank = time.perf_counter() - t0
        print("rep %d: gDCA %.1f ms | read_fasta_alignment %.1f (incl. the copy into numpy), max(Z) %.1f, gdca_run %.1f (device %.1f), "
              "compute_ranking %.1f" % (r, t_all * 1e3, t_parse * 1e3, t_max * 1e3, t_run * 1e3, st["ms_total"], t_rank * 1e3), flush=True)
finally:
    os.unlink(path)
