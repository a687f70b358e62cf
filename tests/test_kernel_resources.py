"""Compile-time guard on registers and scratch of the front-end and score kernels (no GPU: hipcc cross-compiles and reports
`-Rpass-analysis=kernel-resource-usage`).  Round 4 found the reweighting kernel `k_hamming<3, false>` carrying 105 spilled VGPRs
(408 bytes of scratch per lane) because a count pass that could never count anything in that form kept its 64 accumulators alive
across the refinement loop: 0.5 ms of a 23 ms family.  Nothing in the test suite could see that -- results were right.  This test
pins what the compiler reports today, so that the next such regression shows up when it is made: every kernel of these files runs
without scratch.
(`k_inverse.hip`: 20 s since round 5's restructure; its assembly is made once per module and shared by the tests on it.)"""
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


@pytest.mark.parametrize("src", ["k_hamming.hip", "k_hamming_fp4.hip", "k_tally.hip", "k_score.hip", "k_theta.hip", "k_elementwise.hip", "k_rank.hip"])
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


def _asm(src, out, *defs):
    r = subprocess.run([HIPCC, *FLAGS, *defs, "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", str(out)], capture_output=True, text=True,
                       timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    return out.read_text()


@pytest.fixture(scope="module")
def inverse_asm(tmp_path_factory):
    """`hipcc -S` of k_inverse.hip as it ships (20 s), shared by the tests below"""
    if not os.path.exists(HIPCC) or shutil.which("c++filt") is None:
        pytest.skip("no hipcc")
    return _asm("k_inverse.hip", tmp_path_factory.mktemp("asm") / "k_inverse.s")


def _asm_loops():
    import importlib.util

    spec = importlib.util.spec_from_file_location("asm_loops", os.path.join(ROOT, "tools", "asm_loops.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_barrier_sits_inside_a_divergent_loop_of_the_sweep_kernels(inverse_asm, tmp_path):
    """The malformed-loop hazard of DESIGN 3.1b (VERDICT r05 weak #1b): with two thread-0-only blocks in front of one back edge hipcc
    turns the persistent loop of `k_sweep_merged` into an exec-masked one whose per-lane continue mask comes from the `tid == 0`
    compare (`s_andn2_b64 exec, exec, s[..]` in the loop's latch) -- lanes leave one by one a loop that holds 29 .. 47 workgroup
    barriers: wrong inverses in round 4, a hang on an empty list in round 5.  From the compiler's own loop annotations
    (tools/asm_loops.py): no loop of ANY function of k_inverse.hip that holds an `s_barrier` narrows exec at its own level -- and the
    kept reproducer, `-DGDCA_EXP_LATE_NEXT`, DOES trip the check in both merged kernels, so the test is known to see the shape."""
    al = _asm_loops()
    bad, seen = al.divergent_barrier_loops(inverse_asm)
    assert seen >= 100, seen  # (the annotations are there: 174 loops in round 6's file)
    assert not bad, bad
    # the four persistent loops themselves were looked at: depth 1, dozens of barriers, the tile item's MFMAs inside
    persistent = []
    for name, body in al.functions(inverse_asm):
        if re.match(r"_Z\d+k_sweep(?:_merged)?ILb[01]EE", name):
            persistent += [(name, h) for h, v in al.loops_of(body).items() if v["depth"] == 1 and v["barriers"] >= 20 and v["mfmas"] >= 1000]
    assert len(persistent) == 4, persistent
    late = _asm("k_inverse.hip", tmp_path / "late.s", "-DGDCA_EXP_LATE_NEXT")
    bad_late, _ = al.divergent_barrier_loops(late)
    tripped = {name for name, *_ in bad_late}
    assert any("k_sweep_mergedILb1" in n for n in tripped) and any("k_sweep_mergedILb0" in n for n in tripped), bad_late
    assert all(d == 1 or d == 2 for _, _, d, *_ in bad_late) and all(b >= 20 and m >= 1000 for _, _, _, b, m, _ in bad_late), bad_late


@pytest.mark.parametrize("src", ["k_hamming.hip", "k_hamming_fp4.hip", "k_tally.hip", "k_score.hip", "k_theta.hip", "k_elementwise.hip", "k_rank.hip"])
def test_no_barrier_sits_inside_a_divergent_loop_elsewhere(src, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    bad, _ = _asm_loops().divergent_barrier_loops(_asm(src, tmp_path / "x.s"))
    assert not bad, bad


def test_sweep_kernels_keep_scratch_out_of_every_block_that_holds_an_mfma(inverse_asm):
    """The four persistent sweep kernels (round 4: 144 scratch instructions in `k_sweep<true>`, 105 of them inside blocks of 32 MFMAs
    and more -- and a wrong-result incident that moved with the spill placement): since round 5 the cold item kinds are
    functions of their own and what is left in a kernel is the save of ONE register (the one holding spilled SGPRs) around the
    once-per-workgroup call of the chain worker.  Per basic block of the compiler's assembly: no scratch instruction in any block that
    holds an MFMA, at most eight in the whole kernel (saves around the calls of the out-of-line items)."""
    text = inverse_asm
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
