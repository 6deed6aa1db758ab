"""`pip install .` parity with the reference's packaging (reference setup.py:20-49, install.sh:1-14: `pip install .` -> `import flash_attn_turing`).
Builds a wheel from the repo root in a temp dir (no network, no build isolation), unpacks it somewhere that is NOT the source tree and
imports the module from there in a fresh interpreter: library, host module and header must all come out of the wheel."""
import os
import subprocess
import sys
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wheel_builds_and_imports_outside_the_tree(tmp_path):
    wheels = tmp_path / "wheels"
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-m", "pip", "wheel", ROOT, "--no-build-isolation", "--no-deps", "--no-index", "-w", str(wheels)],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    whl = [f for f in os.listdir(wheels) if f.startswith("flash_attn_turing-") and f.endswith(".whl")]
    assert len(whl) == 1, os.listdir(wheels)
    assert "-any" not in whl[0], whl      # a platform wheel: it carries gfx950 code objects
    site = tmp_path / "site"
    with zipfile.ZipFile(wheels / whl[0]) as z:
        names = z.namelist()
        z.extractall(site)
    for need in ("flash_attn_turing/_C.so", "flash_attn_turing/libflash_attn_gfx950.so", "flash_attn_turing/include/flash_attn_gfx950.h",
                 "flash_attn_turing/__init__.py", "flash_attn_turing/capi.py", "flash_attn_turing/interface.py", "flash_attn_turing/sharding.py"):
        assert need in names, (need, names)
    code = ("import os, flash_attn_turing as F, flash_attn_turing.capi as c;"
            "assert F.abi_version() == 4, F.abi_version();"
            "assert os.path.dirname(F.LIBRARY_PATH) == os.path.dirname(F.__file__), F.LIBRARY_PATH;"
            "assert c.lib().fa_abi_version() == 4 and os.path.dirname(os.path.dirname(c.HEADER_PATH)) == os.path.dirname(F.__file__);"
            "assert set(c.declared_functions()) and all(hasattr(c.lib(), n) for n in c.declared_functions());"
            "print(F.__file__)")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = str(site)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.strip().startswith(str(site)), r.stdout      # imported from the unpacked wheel, not from the source tree
