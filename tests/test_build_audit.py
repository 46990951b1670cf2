"""Audit of the compiled gfx950 code of csrc/winograd3w.hip, csrc/winograd3z.hip and csrc/winograd3_wgrad.hip (no GPU
needed: hipcc cross-compiles).

Those kernels keep 256 of their 400 accumulator registers under literal names (a0..a255) inside inline-asm statements, and
issue their MFMAs from inline asm.  hipcc neither knows that those registers are live between the statements nor pads
hazards around them, so three properties of the generated code are part of the kernel's correctness and are checked
here on every build of the test suite:
  1. the compiler never touches the accumulator half of the register file itself (it would only do so to spill vector
     registers, silently overwriting accumulators);
  2. no scratch memory, no register spills;
  3. no vector-ALU instruction writes an A / B operand register of an MFMA within the two instructions in front of it
     (the wait states hipcc would add for its own MFMAs; the operands are meant to come from LDS / buffer loads only).
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "monopsr_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


# source file -> (kernel name fragment, MFMA statements of its two unrolled K steps)
AUDITED = {"winograd3w.hip": ("wino3w_conv_kernel", 200, 256), "winograd3z.hip": ("wino3z_conv_kernel", 128, 256),
           "winograd3_wgrad.hip": ("wino3_wgrad_kernel", 64, 256)}


@pytest.fixture(scope="module", params=sorted(AUDITED))
def w3w_asm(request):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    src = request.param
    tmp = tempfile.mkdtemp(prefix="w3w_audit_")
    try:
        # the flags of csrc/Makefile's rule for these files
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-save-temps", "-c",
               os.path.join(CSRC, src), "-o", os.path.join(tmp, "audit.o")]
        subprocess.check_call(cmd, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        path = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")]
        assert len(path) == 1, os.listdir(tmp)
        with open(os.path.join(tmp, path[0])) as f:
            yield (f.read(),) + AUDITED[src]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _kernel_bodies(asm, frag):
    """name -> list of (instruction text, inside_inline_asm) for every kernel whose name holds `frag`."""
    bodies, name, inasm = {}, None, False
    for line in asm.splitlines():
        m = re.match(r"^(_ZN\S*" + frag + r"\S*):", line)
        if m:
            name, inasm = m.group(1), False
            bodies[name] = []
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end"):
            name = None
            continue
        if ";;#ASMSTART" in t:
            inasm = True
            continue
        if ";;#ASMEND" in t:
            inasm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        bodies[name].append((t.split(";")[0].strip(), inasm))
    return bodies


def test_makefile_builds_this_file_with_the_audited_flags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert re.search(r"winograd3w\.o winograd3z\.o winograd3_wgrad\.o:.*\n\t\$\(HIPCC\) \$\(FLAGS\) -fno-slp-vectorize -c", mk)


def test_compiler_leaves_the_accumulator_registers_alone(w3w_asm):
    asm, frag, n_mfma, n_reads = w3w_asm
    bodies = _kernel_bodies(asm, frag)
    assert len(bodies) >= 1, list(bodies)
    for name, body in bodies.items():
        own = [t for t, inasm in body if not inasm and re.search(r"v_accvgpr|\ba\[\d+:\d+\]|\ba\d+\b", t)]
        assert not own, "%s: compiler-generated accumulator-register traffic: %s" % (name, own[:5])
        reads = [t for t, inasm in body if inasm and t.startswith("v_accvgpr_read_b32")]
        mfmas = [t for t, inasm in body if inasm and t.startswith("v_mfma_f32_32x32x2_f32")]
        assert len(reads) == n_reads and len(mfmas) == n_mfma, (name, len(reads), len(mfmas))


def test_no_scratch_no_spills(w3w_asm):
    asm, frag = w3w_asm[0], w3w_asm[1]
    # metadata entries of the kernels (.amdgpu_metadata, amdhsa.kernels): one "  - .agpr_count: ..." block per kernel
    meta = asm[asm.index("amdhsa.kernels:"):]
    blocks = [b for b in re.split(r"\n  - ", meta) if re.search(r"\.name:\s+\S*" + frag, b)]
    assert len(blocks) >= 1
    for blk in blocks:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        assert re.search(r"\.private_segment_fixed_size:\s+0\b", blk), name
        assert re.search(r"\.vgpr_spill_count:\s+0\b", blk), name
        assert re.search(r"\.agpr_count:\s+256\b", blk), name


def test_no_vector_alu_write_of_an_mfma_operand_right_before_it(w3w_asm):
    asm, frag = w3w_asm[0], w3w_asm[1]
    for name, body in _kernel_bodies(asm, frag).items():
        for i, (t, inasm) in enumerate(body):
            if not t.startswith("v_mfma_f32_32x32x2_f32"):
                continue
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            srcs = [o for o in ops[1:3] if re.fullmatch(r"v\d+", o)]
            assert len(srcs) == 2, t
            for back in (1, 2):
                if i - back < 0:
                    continue
                p = body[i - back][0]
                if not p.startswith("v_") or p.startswith("v_mfma"):
                    continue  # loads, scalar ops, waits; an MFMA never writes another one's A / B operand here
                dst = p.split(None, 1)[1].split(",")[0].strip()
                written = set()
                m = re.fullmatch(r"v\[(\d+):(\d+)\]", dst)
                if m:
                    written = {"v%d" % r for r in range(int(m.group(1)), int(m.group(2)) + 1)}
                elif re.fullmatch(r"v\d+", dst):
                    written = {dst}
                assert not (written & set(srcs)), "%s: '%s' writes an operand of '%s' %d instruction(s) before it" % (
                    name, p, t, back)
