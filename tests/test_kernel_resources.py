"""Compile-time guard on registers and scratch of the front-end and score kernels (no GPU: hipcc cross-compiles and reports
`-Rpass-analysis=kernel-resource-usage`).  Round 4 found the reweighting kernel `k_hamming<3, false>` carrying 105 spilled VGPRs
(408 bytes of scratch per lane) because a count pass that could never count anything in that form kept its 64 accumulators alive
across the refinement loop: 0.5 ms of a 23 ms family.  Nothing in the test suite could see that -- results were right.  This test
pins what the compiler reports today, so that the next such regression shows up when it is made: every kernel of these files runs
without scratch.
(`k_inverse.hip` takes six minutes to compile and is covered by `tools/kernel_resources.py` -> profiles/rNN_kernel_resources.txt.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussdca.jl_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-std=c++17", "-Wno-unused-function", "-Wno-pass-failed",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]

# (file, substring of the mangled kernel name) -> (max spilled VGPRs, max scratch bytes per lane); everything else: 0 / 0.
# Round 5: empty -- the bound form of the Hamming kernel, which carried 36 spilled VGPRs around an inlined refinement path (and
# stored them on every tile: 2.2 GB written per launch at config C), now only lists its candidates; a kernel of its own refines them.
ALLOWED = {}


@pytest.mark.parametrize("src", ["k_hamming.hip", "k_tally.hip", "k_score.hip", "k_theta.hip", "k_elementwise.hip", "k_rank.hip"])
def test_front_end_and_score_kernels_do_not_spill(src, tmp_path):
    if not os.path.exists(HIPCC) or shutil.which("c++filt") is None:
        pytest.skip("no hipcc")
    r = subprocess.run([HIPCC, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", str(tmp_path / "x.o")],
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    assert blocks, "no kernel-resource-usage remarks in the compiler's output"
    seen = []
    for b in blocks:
        name = b.split()[0]

        def field(label):
            m = re.search(label + r": (\d+)", b)
            assert m, (name, label)
            return int(m.group(1))

        spills, scratch, vgprs, occ = field("VGPRs Spill"), field(r"ScratchSize \[bytes/lane\]"), field("VGPRs"), field(r"Occupancy \[waves/SIMD\]")
        max_spill, max_scratch = next((v for (f, k), v in ALLOWED.items() if f == src and k in name), (0, 0))
        seen.append((name, vgprs, spills, scratch, occ))
        assert spills <= max_spill and scratch <= max_scratch, (name, "VGPRs", vgprs, "spilled", spills, "scratch B/lane", scratch)
    print("\n".join("%-70s VGPRs %3d spilled %3d scratch %3d B occupancy %d" % s for s in seen))


def test_sweep_kernels_keep_scratch_out_of_every_block_that_holds_an_mfma(tmp_path):
    """The four persistent sweep kernels (round 4: 144 scratch instructions in `k_sweep<true>`, 105 of them inside blocks of 32 MFMAs
    and more -- and a wrong-result incident that moved with the spill placement): since round 5 the cold item kinds are
    functions of their own and what is left in a kernel is the save of ONE register (the one holding spilled SGPRs) around the
    once-per-workgroup call of the chain worker.  Per basic block of the compiler's assembly: no scratch instruction in any block that
    holds an MFMA, at most eight in the whole kernel (saves around the calls of the out-of-line items)."""
    if not os.path.exists(HIPCC) or shutil.which("c++filt") is None:
        pytest.skip("no hipcc")
    asm = tmp_path / "k_inverse.s"
    r = subprocess.run([HIPCC, *FLAGS, "--cuda-device-only", "-S", os.path.join(CSRC, "k_inverse.hip"), "-o", str(asm)], capture_output=True,
                       text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    text = asm.read_text()
    funcs = re.split(r"\n(?=_Z[A-Za-z0-9_]+:)", text)
    seen = 0
    for f in funcs:
        m = re.match(r"(_Z\d+k_sweep(?:_merged)?ILb[01]EE[A-Za-z0-9_]*):", f)
        if not m or not re.search(r"^\s+s_endpgm", f, re.M):
            continue
        seen += 1
        body = f.split(".Lfunc_end")[0]
        blocks = re.split(r"\n\.LBB[0-9_]+:", body)
        total = len(re.findall(r"\bscratch_(?:load|store)", body))
        hot = sum(len(re.findall(r"\bscratch_(?:load|store)", b)) for b in blocks if re.search(r"\bv_mfma_", b))
        print("%-50s %4d basic blocks, %4d MFMAs, scratch instructions %d (in MFMA blocks: %d)"
              % (m.group(1), len(blocks), len(re.findall(r"\bv_mfma_", body)), total, hot))
        assert hot == 0 and total <= 8, (m.group(1), total, hot)
    assert seen == 4, seen
