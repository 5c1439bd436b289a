#!/usr/bin/env python3
"""Builds the three COMPILED extension modules that carry the reference's names -- pointnet2_cuda, iou3d_cuda, roipool3d_cuda
(pointnet2/setup.py:7, iou3d/setup.py:7, roipool3d/setup.py:7) -- into 3d_adapt_auto_driving_amd/dropin_native/: pybind11 wrappers
(this directory) around the C ABI of ../../lib/libprcnn_hip.so.  Host code only, so plain g++ with torch's include / library
paths (what torch.utils.cpp_extension would pass); the modules find libprcnn_hip.so through an $ORIGIN rpath.  In-tree outputs:
they travel to the GPU box with the snapshot.   usage: python build.py [--force]"""
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(PKG, "dropin_native")
LIB = os.path.join(PKG, "lib")
MODULES = {"pointnet2_cuda": "pointnet2_api_hip.cpp", "iou3d_cuda": "iou3d_api_hip.cpp", "roipool3d_cuda": "roipool3d_api_hip.cpp"}


def target(name):
    return os.path.join(OUT, name + sysconfig.get_config_var("EXT_SUFFIX"))


def stale(name):
    t = target(name)
    if not os.path.exists(t):
        return True
    deps = [os.path.join(HERE, MODULES[name]), os.path.join(HERE, "binding_common.h"), os.path.join(os.path.dirname(PKG), "include", "prcnn_hip.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(t) for d in deps)


def build_one(name):
    import torch
    from torch.utils import cpp_extension as ce
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include"]
    libdirs = ce.library_paths()
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=" + name, "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + p for p in inc] + [os.path.join(HERE, MODULES[name]), "-o", target(name)]
    cmd += ["-L" + p for p in libdirs] + ["-L" + LIB, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", "-lprcnn_hip",
                                          "-Wl,-rpath,$ORIGIN/../lib"] + ["-Wl,-rpath," + p for p in libdirs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building %s failed:\n%s" % (name, r.stderr[-3000:]))
    return name


def main(force=False):
    os.makedirs(OUT, exist_ok=True)
    todo = [n for n in MODULES if force or stale(n)]
    if todo:
        with ThreadPoolExecutor(len(todo)) as pool:
            for n in pool.map(build_one, todo):
                print("built", os.path.relpath(target(n), os.path.dirname(PKG)))
    return [target(n) for n in MODULES]


if __name__ == "__main__":
    main("--force" in sys.argv)
