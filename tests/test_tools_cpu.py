"""Every script kept under tools/ must at least parse (probes for options that no longer exist must not rot
silently): Python files compile, shell files pass `bash -n`; a script that sets an engine option by name must name one that exists."""
import os
import py_compile
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tools")


def _files(ext):
    out = []
    for d, _, fs in os.walk(TOOLS):
        out += [os.path.join(d, f) for f in fs if f.endswith(ext)]
    return sorted(out)


def test_python_tools_compile():
    for f in _files(".py"):
        py_compile.compile(f, doraise=True)


def test_shell_tools_parse():
    for f in _files(".sh"):
        r = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stderr)


def test_engine_options_named_by_tools_exist():
    src = open(os.path.join(ROOT, "csm-hf_amd", "csrc", "engine.hip")).read()
    src += open(os.path.join(ROOT, "csm-hf_amd", "csrc", "mimi.hip")).read()      # csm_mimi_set_option
    known = set(re.findall(r'!strcmp\(name, "([a-z0-9_]+)"\)', src))
    assert len(known) >= 20
    bad = []
    for f in _files(".py") + _files(".sh"):
        if os.sep + "ubench" + os.sep in f:
            continue
        txt = open(f).read()
        for m in re.finditer(r'set_option\("([a-z0-9_]+)"', txt):
            if m.group(1) not in known:
                bad.append((os.path.relpath(f, ROOT), m.group(1)))
        for m in re.finditer(r'--opt ([a-z0-9_]+)=', txt):
            if m.group(1) not in known:
                bad.append((os.path.relpath(f, ROOT), m.group(1)))
    assert not bad, bad
