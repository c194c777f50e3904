"""Builds tests/abi_client (plain C99, gcc) against include/csi.h and the in-tree libcsi_hip.so; test infrastructure."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "abi_client.c")
EXE = os.path.join(ROOT, "tests", "abi_client")
LIBDIR = os.path.join(ROOT, "climaseaice.jl_amd")


def build(force=False):
    deps = [SRC, os.path.join(ROOT, "include", "csi.h"), os.path.join(LIBDIR, "libcsi_hip.so")]
    if not force and os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return EXE
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror=implicit-function-declaration", "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", SRC, "-o", EXE, "-L" + LIBDIR, "-lcsi_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return EXE


if __name__ == "__main__":
    print(build(force=True))
