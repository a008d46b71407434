"""The C-ABI library loads and exports every symbol include/pcreg.h declares; struct
layouts match; without a GPU every compute entry point fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pcreg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pcreg_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    from pcreg_amd import _lib
    L = _lib.lib()
    declared = _declared()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/pcreg.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared


def test_struct_layouts():
    from pcreg_amd import _lib
    assert C.sizeof(_lib.RansacOpts) == 40          # 2*i32, 2*f64, 2*i32, u64
    assert C.sizeof(_lib.MatchOpts) == 64
    assert C.sizeof(_lib.DevRansacResult) == 16 * 8 + 6 * 4
    assert _lib.lib().pcreg_version().decode().startswith("pcreg-hip")


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import pcreg_amd as pc
    from pcreg_amd._lib import PcregError, PCREG_E_NODEVICE
    p = np.random.default_rng(0).normal(size=(10, 3))
    for call in (lambda: pc.estimateTransform(p, p), lambda: pc.ransac(p, p, dict(minPtNum=3, iterNum=10, thDist=1, thInlrRatio=0.1, REFINE=True, VERBOSE=0)),
                 lambda: pc.knn2_points(p, p), lambda: pc.getMatches(p, p, dict(UNNORMALIZE=False, CHANGE_METRIC=False, Method="Exhaustive", MatchThreshold=10, MaxRatio=0.6, Metric="SSD", Unique=False, VERBOSE=0)),
                 lambda: pc.AlignPoints_KNN(p)):
        with pytest.raises(PcregError) as e:
            call()
        assert e.value.code == PCREG_E_NODEVICE


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pcreg_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in txt.replace("the oracle", "").replace("oracle's", "").replace("oracle C", "").replace("oracle standing", "").replace("(oracle", "") or f.endswith((".hip", ".hpp")), f
                assert "import oracle" not in txt and "from oracle" not in txt and "pcreg_oracle" not in txt, f


def test_comm_entry_points_fail_cleanly_without_a_device():
    """The multi-GPU entry points exist on every box; without a GPU pcreg_comm_init reports NODEVICE and the
    sharded calls refuse to run without a communicator (no crash, no hang)."""
    import torch
    from pcreg_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    L = _lib.lib()
    buf = (C.c_char * 128)()
    assert L.pcreg_comm_init(0, 1, C.byref(buf)) == _lib.PCREG_E_NODEVICE
    r, w = C.c_int(), C.c_int()
    assert L.pcreg_comm_rank(C.byref(r), C.byref(w)) == _lib.PCREG_E_ARG
    assert L.pcreg_comm_destroy() == 0


def test_default_library_carries_no_experiment_switches():
    """The experiment / debug environment switches (timing-only kernels, early returns, launch-shape overrides) exist only
    in `make EXPERIMENTS=1` builds: the library `__graft_entry__.build()` produces must not even contain their names
    (the objects depend on the flag string, so a default `make` after an experiments build relinks everything)."""
    from pcreg_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"PCREG_KNN_VARIANT", b"PCREG_DESC_STOP", b"PCREG_SAD_DRY", b"PCREG_KNN_F16_QG", b"PCREG_ALIGN_TIMES", b"PCREG_DESC_SUBDIV"):
        assert name not in blob, name.decode()


def test_default_library_reads_no_environment_variable():
    """ADVICE / VERDICT r3: the A/B switches the parity tests flip (exact search, exhaustive SAD, forced fallback, fused /
    fp64-only RANSAC) are pcreg_debug_set keys, not environment variables: the shipped library neither imports getenv nor
    contains any PCREG_* variable name, and an unknown key is an argument error."""
    import re
    import subprocess
    from pcreg_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    names = set(re.findall(rb"PCREG_(?:KNN|MATCH|RANSAC|ALIGN|SEG|SAD|DESC_S|DESC_X)[A-Z0-9_]*", blob))      # PCREG_METRIC_SAD etc. are enum names in messages
    assert not names, names
    nm = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert not re.search(r"\b(secure_)?getenv\b", nm), "libpcreg_hip.so imports getenv"
    L = _lib.lib()
    assert L.pcreg_debug_set(b"match_exact", 0) == _lib.PCREG_OK
    assert L.pcreg_debug_set(b"no_such_key", 1) == _lib.PCREG_E_ARG
    assert b"no_such_key" in L.pcreg_last_error()


def test_matlab_wrappers_only_use_gateway_commands_that_exist():
    """No MATLAB here: at least every pcreg_mex('<command>', ...) in matlab/*.m (and in INTEGRATION.md's snippets) must be a command
    the gateway dispatches, and every command of the gateway must be named in its header comment."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gw = open(os.path.join(root, "mex", "pcreg_mex.cpp")).read()
    have = set(re.findall(r'strcmp\(cmd, "([A-Za-z_]+)"\)', gw))
    assert len(have) >= 20
    used = {}
    for f in glob.glob(os.path.join(root, "matlab", "*.m")) + [os.path.join(root, "INTEGRATION.md")]:
        for c in re.findall(r"pcreg_mex\('([A-Za-z_]+)'", open(f).read()):
            used.setdefault(c, f)
    missing = {c: f for c, f in used.items() if c not in have}
    assert not missing, missing
    head = gw[:gw.index("#if __has_include")]
    undocumented = [c for c in have if "'" + c + "'" not in head]
    assert not undocumented, undocumented


def test_matlab_wrappers_pass_the_argument_counts_the_gateway_checks():
    """The other thing only MATLAB would notice: a wrapper handing the gateway a number of arguments its command refuses.  Every
    pcreg_mex('<command>', a, b, ...) call in matlab/*.m is parsed (nested brackets, '...' continuations) and its argument count held
    against the command's own `nrhs != N` / `nrhs < N` test in mex/pcreg_mex.cpp."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gw = open(os.path.join(root, "mex", "pcreg_mex.cpp")).read()
    rule = {}
    blocks = re.split(r'strcmp\(cmd, "', gw)[1:]
    for b in blocks:
        name = b[:b.index('"')]
        m = re.search(r"nrhs (!=|<) (\d+)", b)
        if m:
            rule[name] = (m.group(1), int(m.group(2)))
    checked = 0
    for f in glob.glob(os.path.join(root, "matlab", "*.m")):
        txt = re.sub(r"\.\.\.[^\n]*\n", " ", open(f).read())                  # join continued lines
        txt = "\n".join(ln.split("%")[0] if "'" not in ln.split("%")[0][-1:] else ln for ln in txt.split("\n"))
        for m in re.finditer(r"pcreg_mex\('([A-Za-z_]+)'", txt):
            i, depth, n_args, in_str = m.end(), 1, 1, False
            while depth > 0 and i < len(txt):
                ch = txt[i]
                if in_str:
                    in_str = ch != "'"
                elif ch == "'" and txt[i - 1] in "(,[ {=":
                    in_str = True
                elif ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                elif ch == "," and depth == 1:
                    n_args += 1
                i += 1
            assert depth == 0, (f, m.group(1))
            cmd = m.group(1)
            if cmd in rule:
                op, n = rule[cmd]
                ok = n_args == n if op == "!=" else n_args >= n
                assert ok, f"{os.path.basename(f)}: pcreg_mex('{cmd}', ...) passes {n_args} arguments, the gateway wants nrhs {'==' if op == '!=' else '>='} {n}"
                checked += 1
    assert checked >= 10, checked


def test_matlab_wrappers_ask_for_no_more_outputs_than_the_gateway_returns():
    """[a, b, c] = pcreg_mex('<command>', ...) in matlab/*.m against the highest plhs[k] the command's block of the gateway assigns."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gw = open(os.path.join(root, "mex", "pcreg_mex.cpp")).read()
    gives = {}
    for b in re.split(r'strcmp\(cmd, "', gw)[1:]:
        name = b[:b.index('"')]
        ks = [int(k) for k in re.findall(r"plhs\[(\d+)\]", b)] + [int(k) for k in re.findall(r"\bout\((\d+),", b)]
        gives[name] = (max(ks) + 1) if ks else 0
    checked = 0
    for f in glob.glob(os.path.join(root, "matlab", "*.m")):
        txt = re.sub(r"\.\.\.[^\n]*\n", " ", open(f).read())
        for m in re.finditer(r"(?:\[([^\]=]*)\]|([A-Za-z_][A-Za-z_0-9.]*))\s*=\s*pcreg_mex\('([A-Za-z_]+)'", txt):
            n_out = len([x for x in re.split(r"[,\s]+", m.group(1).strip()) if x]) if m.group(1) is not None else 1
            cmd = m.group(3)
            assert cmd in gives and n_out <= gives[cmd], f"{os.path.basename(f)}: {n_out} outputs asked of '{cmd}', the gateway returns {gives.get(cmd)}"
            checked += 1
    assert checked >= 10, checked
