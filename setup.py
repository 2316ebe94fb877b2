"""Build hook: compile libnessai_amd.so (gfx950) with the in-tree Makefile before the package data is
collected.  Metadata, entry points and package list live in pyproject.toml."""
import os
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildWithHip(build_py):
    def run(self):
        if os.environ.get("NESSAI_AMD_SKIP_HIP_BUILD") != "1":
            jobs = str(min(8, os.cpu_count() or 1))
            subprocess.run(["make", "-C", os.path.join(ROOT, "nessai_amd", "csrc"), "-j", jobs], check=True)
        super().run()


setup(cmdclass={"build_py": BuildWithHip})
