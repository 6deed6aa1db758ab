"""`pip install .` for the MI355X drop-in of ssiu/flash-attention-turing (replaces the reference's CUDAExtension build, reference
setup.py:20-49 / install.sh:1-14).

The kernels are compiled by flash-attention-turing_amd/build.py (`hipcc --offload-arch=gfx950`, no GPU needed; the host module with
plain g++ against the installed torch); this file only runs it and ships the results INSIDE the package:

    flash_attn_turing/_C.so                        host module (rpath $ORIGIN)
    flash_attn_turing/libflash_attn_gfx950.so      HIP kernels + C ABI
    flash_attn_turing/include/flash_attn_gfx950.h  the C-ABI header (for non-Python hosts: `python -c "import flash_attn_turing.capi as c; print(c.HEADER_PATH)"`)

No network is needed: `pip install . --no-build-isolation` (torch must already be installed, as for the reference).
"""
import importlib.util
import os
import shutil

from setuptools import setup
from setuptools.command.build_py import build_py
from setuptools.command.install import install
from setuptools.dist import Distribution

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.join(ROOT, "flash-attention-turing_amd")


def _fa_build():
    spec = importlib.util.spec_from_file_location("fa_build", os.path.join(PKG_ROOT, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class BuildPyWithKernels(build_py):
    def run(self):
        fa = _fa_build()
        fa.build_all(force=False, torch_module=True, debug_variants=False)      # (the test builds of csrc/debug are not part of an installation)
        super().run()
        dst = os.path.join(self.build_lib, "flash_attn_turing")
        os.makedirs(os.path.join(dst, "include"), exist_ok=True)
        shutil.copy2(fa.LIB_PATH, os.path.join(dst, fa.LIB_NAME))
        shutil.copy2(fa.EXT_PATH, os.path.join(dst, "_C.so"))
        shutil.copy2(os.path.join(fa.INCLUDE, "flash_attn_gfx950.h"), os.path.join(dst, "include", "flash_attn_gfx950.h"))


class InstallIntoPlatlib(install):
    def finalize_options(self):      # everything (the .py files too) next to the binaries, not split over purelib / platlib
        super().finalize_options()
        self.install_lib = self.install_platlib


class BinaryDistribution(Distribution):
    def has_ext_modules(self):      # platform wheel: it carries gfx950 code objects and a CPython extension
        return True


setup(
    name="flash_attn_turing",
    version="0.1.0",
    description="MI355X (gfx950) fused attention behind the flash_attn_turing API",
    packages=["flash_attn_turing"],
    package_dir={"flash_attn_turing": os.path.join("flash-attention-turing_amd", "flash_attn_turing")},
    cmdclass={"build_py": BuildPyWithKernels, "install": InstallIntoPlatlib},
    distclass=BinaryDistribution,
    python_requires=">=3.9",
    zip_safe=False,
)
