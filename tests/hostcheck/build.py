"""Build tests/hostcheck/libhostcheck.so with g++ (test infrastructure only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libhostcheck.so")
SRC = os.path.join(HERE, "hostcheck.cpp")
HDR = os.path.join(HERE, "..", "..", "nmma_amd", "csrc", "em_math.h")


def build(force=False):
    if (not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(SRC), os.path.getmtime(HDR))):
        return LIB
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", SRC, "-o", LIB, "-lm"],
                   check=True)
    return LIB
