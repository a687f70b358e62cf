"""Phase times of the FASTA reader on one file of config C's size (GDCA_FASTA_TRACE=1), plain and gzip, for a ladder of reader
thread counts: where a single gDCA(filename) call's parse time goes.   python tools/parse_trace.py [N M]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (500, 50000)
code = ("import sys, time; sys.path.insert(0, %r)\n"
        "from gaussdca.jl_amd import dcautils\n"
        "for r in range(5):\n"
        "    t0 = time.perf_counter()\n"
        "    with dcautils.FastaAlignment(sys.argv[1], 0.9) as fa:\n"
        "        q = fa.q\n"
        "    print('open+close %%.2f ms' %% ((time.perf_counter() - t0) * 1e3), flush=True)\n" % ROOT)
with tempfile.TemporaryDirectory() as d:
    for ext in (".fasta", ".fasta.gz"):
        path = os.path.join(d, "fam" + ext)
        subprocess.run([os.path.join(ROOT, "gaussdca.jl_amd", "gdca_cli"), "--synth", str(N), str(M), str(0xC500), path], check=True, stdout=subprocess.DEVNULL)
        for threads in ("", "4", "8", "16"):
            env = dict(os.environ, GDCA_FASTA_TRACE="1")
            if threads:
                env["GDCA_FASTA_THREADS"] = threads
            r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, env=env)
            print("== %s, GDCA_FASTA_THREADS=%s" % (ext, threads or "(default)"))
            print("\n".join(r.stderr.strip().splitlines()[-2:] + r.stdout.strip().splitlines()[-2:]))
