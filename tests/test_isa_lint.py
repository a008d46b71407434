"""ISA lint of the point-search kernel and of the micro-benchmark behind DESIGN.md's issue model (CPU test:
hipcc cross-compiles gfx950 to assembly here).

knn_mfma16.hip reads its A operands with inline-asm `ds_read_b128` and waits in a separate asm statement; the
compiler believes an asm output is ready at once, so a register copy scheduled between the two would read the
registers before the data lands (it happened once: DESIGN.md section 4.1).  Nothing at run time would tell on a
lucky box, so the emitted ISA is checked:
  * no instruction touches a ds_read_b128 destination between the load and its `s_waitcnt lgkmcnt(0)`;
  * the hot body of the default kernel holds exactly 8 MFMAs (two sub-tiles x four query groups), no scratch
    access, no long `s_nop` (the serial form's `s_nop 9` after every MFMA);
  * scripts/ubench/mfma_f16_valu.hip really issues 8 distinct MFMAs per loop iteration in every MFMA mode (round 1's
    version had them CSE'd into one, and a wrong conclusion was drawn from it).
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-math-errno", "-fno-honor-nans",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only"]

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _asm(src, tmp_path, extra=()):
    out = str(tmp_path / (os.path.basename(src) + ".s"))
    subprocess.check_call([HIPCC, *FLAGS, *extra, "-o", out, src], stderr=subprocess.DEVNULL)
    return open(out).read()


def _functions(text):
    """name -> list of instruction lines (comments and directives dropped)."""
    funcs, cur, name = {}, None, None
    for ln in text.splitlines():
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, cur = m.group(1), []
            funcs[name] = cur
            continue
        if cur is None:
            continue
        s = ln.split(";")[0].rstrip() if not ln.strip().startswith(";;#") else ln.strip()
        if s.strip().startswith(".") and not s.strip().startswith(".LBB"):
            if s.strip().startswith(".Lfunc_end"):
                cur = None
            continue
        if s.strip():
            cur.append(s.strip())
    return funcs


def _regs(operand_text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", operand_text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bv(\d+)\b", operand_text))
    return out


@pytest.fixture(scope="module")
def knn_asm(tmp_path_factory):
    return _functions(_asm(os.path.join(ROOT, "pcreg_amd", "csrc", "knn_mfma16.hip"), tmp_path_factory.mktemp("isa"),
                           ["-I" + os.path.join(ROOT, "pcreg_amd", "csrc")]))


def test_no_register_of_an_lds_load_is_touched_before_its_wait(knn_asm):
    checked = 0
    for name, ins in knn_asm.items():
        if "knn_candidates_f16" not in name:
            continue
        for i, s in enumerate(ins):
            # the inline-asm loads only: a load the COMPILER emits (ug_reduce_boxes' LDS reduction in the surplus workgroup)
            # is tracked by the compiler's own s_waitcnt lgkmcnt(N) bookkeeping
            if not s.startswith("ds_read_b128") or i == 0 or not ins[i - 1].startswith(";;#ASMSTART"):
                continue
            dst = _regs(s.split(",")[0])
            for j in range(i + 1, len(ins)):
                t = ins[j]
                if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                    break
                if t.startswith(";;#") or t.startswith(".LBB"):
                    continue
                assert not (dst & _regs(t.split(None, 1)[1] if " " in t else "")), f"{name}: `{t}` touches {s.split(',')[0]} before the wait"
                assert not t.startswith(("s_endpgm", "s_barrier")), f"{name}: {s} is never waited for"
            checked += 1
    assert checked >= 4


def test_no_mixed_precision_fma_in_the_f16_split(knn_asm):
    """v_fma_mixlo_f16 = `(_Float16)fma(a, b, c)` rounded once; next to a second use of the fp32 value (rounded twice)
    it breaks the error-free split at f16 rounding midpoints (knn_mfma16.hip, split2): 6 wrong second neighbours in
    50 000 queries with a passing certificate.  The sources make the values opaque; the ISA must show no such fold."""
    for name, ins in knn_asm.items():
        assert not any(s.startswith("v_fma_mix") for s in ins), f"{name} holds a mixed-precision FMA"


def _hot_body(ins):
    """The default kernel's hot body: from the header of the loop that holds the steady-state MFMAs to the first
    conditional scalar branch after it (the `any hit?` test)."""
    mf = [i for i, s in enumerate(ins) if s.startswith("v_mfma")]
    assert len(mf) >= 9
    first = mf[1]                                   # mf[0] is the tile prologue's product
    start = max(i for i in range(first) if ins[i].startswith(".LBB"))
    end = next(i for i in range(first, len(ins)) if ins[i].startswith("s_cbranch_scc"))
    return ins[start + 1:end]


def test_hot_body_of_the_default_kernel(knn_asm):
    name = next(n for n in knn_asm if "knn_candidates_f16_pipe_kernelILi4ELb0" in n)
    body = [s for s in _hot_body(knn_asm[name]) if not s.startswith(";;#")]
    assert sum(s.startswith("v_mfma_f32_32x32x16_f16") for s in body) == 8
    assert not any(s.startswith("scratch_") for s in body), "spill inside the hot body"
    for s in body:
        if s.startswith("s_nop"):
            assert int(s.split()[1]) <= 3, f"`{s}`: an MFMA result is waited for on the critical path"
    # selection: the first product of a pair 7 v_min3 + 1 v_min (VOP3 form), the second 8 v_min3 (the first's minimum is
    # its 17th value) + the pair's ONE v_cmp; nothing else on the VALU but address arithmetic
    assert sum(s.startswith("v_min3_f32") for s in body) == 60
    assert sum(s.startswith("v_min_f32") for s in body) == 4
    assert sum(s.startswith("v_cmp_lt_f32") for s in body) == 4
    other = [s for s in body if s.startswith("v_") and not s.startswith(("v_mfma", "v_min3_f32", "v_min_f32", "v_cmp_lt_f32"))]
    assert len(other) <= 4, other


def test_ubench_issues_eight_distinct_mfmas_per_iteration(tmp_path):
    funcs = _functions(_asm(os.path.join(ROOT, "scripts", "ubench", "mfma_f16_valu.hip"), tmp_path))
    seen = 0
    for mode in (0, 1, 2, 4, 5):
        name = next(n for n in funcs if n.startswith(f"_Z1kILi{mode}E"))
        ins = funcs[name]
        # the kernel has ONE loop (not unrolled further) and no MFMA outside it: the static count is the per-iteration count
        mf = [s for s in ins if s.startswith("v_mfma_f32_32x32x16_f16")]
        assert len(mf) == 8, f"mode {mode}: {len(mf)} MFMAs in the loop body"
        assert len({s.split()[1].rstrip(",") for s in mf}) >= 2 or mode == 1      # distinct accumulators (mode 1 is the serial form)
        seen += 1
    assert seen == 5


# ---- ransac.hip: inline-asm SCALAR loads (rs_score32_kernel's T32 rows, (round 2's lane refit kernel did the same with its records; it is gone)) ---------------
def _sregs(text):
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bs(\d+)\b", text))
    return out


def _walk_to_wait(ins, labels, start, dst, name):
    """Every path from instruction `start` to the first s_waitcnt that drains lgkmcnt: no instruction on the way may read or
    write the SGPRs in `dst`.  Returns the number of instructions visited."""
    seen, todo, visited = set(), [start], 0
    while todo:
        i = todo.pop()
        while i < len(ins):
            if i in seen:
                break
            seen.add(i)
            t = ins[i]
            if t.startswith(";;#") or t.startswith(".LBB"):
                i += 1
                continue
            visited += 1
            if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                break
            assert not t.startswith(("s_endpgm", "s_setpc", "s_swappc")), f"{name}: a scalar load into s{sorted(dst)[0]}.. is never waited for"
            ops = t.split(None, 1)[1] if " " in t else ""
            if not t.startswith("s_load_dword"):         # the other loads of the same asm statement name their own destinations
                assert not (dst & _sregs(ops)), f"{name}: `{t}` touches the destination of an in-flight scalar load (s{sorted(dst)[0]}..)"
            if t.startswith("s_branch"):
                i = labels[ops.strip()]
                continue
            if t.startswith("s_cbranch"):
                todo.append(labels[ops.strip()])
            i += 1
    return visited


def test_no_sgpr_of_an_asm_scalar_load_is_touched_before_its_wait(tmp_path):
    """rs_score32_kernel prefetches each hypothesis' row of T32 with inline-asm s_load_dwordx16 and waits in a separate asm
    statement.  The compiler believes an asm output
    is ready at once, so a phi copy, an SGPR spill or a re-coalescing between load and wait would read stale registers
    and give wrong inlier counts silently (ADVICE, round 2).  Walk the emitted ISA from every such load along EVERY path
    (branches followed, back edges included) to the wait that drains it."""
    flags = [f for f in FLAGS if f not in ("-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form=1")]
    out = str(tmp_path / "ransac.s")
    subprocess.check_call([HIPCC, *flags, "-I" + os.path.join(ROOT, "pcreg_amd", "csrc"), "-o", out,
                           os.path.join(ROOT, "pcreg_amd", "csrc", "ransac.hip")], stderr=subprocess.DEVNULL)
    funcs = _functions(open(out).read())
    checked = {}
    for name, ins in funcs.items():
        if "rs_score32_kernel" not in name:
            continue
        labels = {s.rstrip(":").split(":")[0]: i for i, s in enumerate(ins) if s.startswith(".LBB")}
        in_asm = False
        for i, s in enumerate(ins):
            if s.startswith(";;#ASMSTART"):
                in_asm = True
            elif s.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm and s.startswith("s_load_dword"):
                dst = _sregs(s.split(",")[0])
                assert dst, s
                # start after the END of this asm statement (its sibling loads are part of the same issue group)
                j = next(k for k in range(i, len(ins)) if ins[k].startswith(";;#ASMEND")) + 1
                assert _walk_to_wait(ins, labels, j, dst, name) > 0
                checked[name] = checked.get(name, 0) + 1
    assert len(checked) == 2 and all("rs_score32" in n for n in checked), checked            # both instantiations (EMIT = true / false)
    assert all(v >= 2 for v in checked.values()), checked
