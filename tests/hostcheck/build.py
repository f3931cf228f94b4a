"""Build tests/hostcheck/libhostcheck.so with g++ (test infrastructure only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libhostcheck.so")
SRC = os.path.join(HERE, "hostcheck.cpp")
HDR = os.path.join(HERE, "..", "..", "nmma_amd", "csrc", "em_math.h")


GW_LIB = os.path.join(HERE, "libhostcheck_gw.so")
GW_SRC = os.path.join(HERE, "hostcheck_gw.cpp")
GW_HDR = os.path.join(HERE, "..", "..", "nmma_amd", "csrc", "gw_math.h")


def build_gw(force=False):
    """Host build of nmma_amd/csrc/gw_math.h (the gravitational-wave leg's scalar math)."""
    if (not force and os.path.exists(GW_LIB) and os.path.getmtime(GW_LIB) >= max(os.path.getmtime(GW_SRC), os.path.getmtime(GW_HDR))):
        return GW_LIB
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", GW_SRC, "-o", GW_LIB, "-lm"],
                   check=True)
    return GW_LIB


def build(force=False):
    if (not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(SRC), os.path.getmtime(HDR))):
        return LIB
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", SRC, "-o", LIB, "-lm"],
                   check=True)
    return LIB
