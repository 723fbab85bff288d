"""The strip-walk kernels of lunext.hip owe their speed to the compiler pairing their per-channel multiply-adds into
v_pk_fma_f32 (two FMAs per lane and issue).  That pairing is an optimisation the compiler may drop for reasons far from
the loop (a branch in the tail reduction once cost 43 of 118 packed instructions and 15 % of the kernel), and no
numerical test notices.  The kernels now spell the pairs out (fma2 on two-element vectors); this test compiles the source to gfx950 assembly (device code only, no GPU needed) and counts
the packed FMAs in the main loop of each instantiation."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# (mangled-name prefix, least number of v_pk_fma_f32 in the loop body, most scalar v_fma / v_fmac there)
EXPECT = [
    ("_Z24dwblock_bwd_strip_kernelILi18ELb0ELb0E", 120, 6),
    ("_Z24dwblock_bwd_strip_kernelILi18ELb1ELb0E", 120, 12),
    ("_Z24dwblock_bwd_strip_kernelILi18ELb0ELb1E", 120, 6),
    ("_Z23dwconv_fwd_strip_kernelILi18ELb0E", 56, 4),
    ("_Z23dwconv_fwd_strip_kernelILi18ELb1E", 56, 8),
]


def loop_body(asm, prefix):
    """lines of the innermost loop of the first function whose name starts with prefix"""
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().endswith(":") or
                 (l.startswith(prefix) and ": ; @" in l))
    body, inside = [], False
    for l in lines[start:]:
        if "s_endpgm" in l:
            break
        if "Inner Loop Header" in l:
            inside = True
            continue
        if inside:
            if re.match(r"^\.LBB\d+_\d+:\s*$", l):          # a label that is not "in Loop": the loop has ended
                break
            body.append(l)
    return body


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_strip_walk_loops_keep_their_packed_multiply_adds(tmp_path):
    out = tmp_path / "lunext.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-I", os.path.join(ROOT, "include"),
           "-S", "--cuda-device-only", "-o", str(out), os.path.join(ROOT, "unet_amd", "csrc", "lunext.hip")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    asm = out.read_text()
    for prefix, min_packed, max_scalar in EXPECT:
        body = loop_body(asm, prefix)
        assert body, f"no loop found in {prefix}"
        packed = sum("v_pk_fma_f32" in l for l in body)
        scalar = sum(bool(re.search(r"\bv_fmac?_f32", l)) for l in body)
        assert packed >= min_packed and scalar <= max_scalar, (
            f"{prefix}: {packed} v_pk_fma_f32 and {scalar} scalar FMAs in the main loop "
            f"(expected >= {min_packed} and <= {max_scalar}): the compiler no longer pairs the multiply-adds")
    # registers: the plain forward walk runs four waves per SIMD with 128 registers and three with 129
    for prefix, most in (("_Z23dwconv_fwd_strip_kernelILi18ELb0E", 128), ("_Z23dwconv_fwd_strip_kernelILi18ELb1E", 168),
                         ("_Z24dwblock_bwd_strip_kernelILi18ELb0ELb0E", 256), ("_Z24dwblock_bwd_strip_kernelILi18ELb1ELb0E", 256)):
        m = re.search(r"^%s.*?; NumVgprs: (\d+).*?; ScratchSize: (\d+)" % re.escape(prefix), asm, re.S | re.M)
        assert m, prefix
        assert int(m.group(1)) <= most and int(m.group(2)) == 0, f"{prefix}: {m.group(1)} registers, {m.group(2)} bytes of scratch"
