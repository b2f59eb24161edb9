#!/usr/bin/env python3
"""Build libsea_hip.so (hand-written HIP kernels for gfx950 + the host C++ pieces) in-tree.

    python robust-segmentation_amd/build_native.py [--force]

Plain hipcc, no cmake: every translation unit is compiled to an object in parallel and linked into
``robust-segmentation_amd/lib/libsea_hip.so``.  hipcc cross-compiles for gfx950 without a GPU.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libsea_hip.so")
ARCH = "gfx950"

# (source, extra flags).  -ffp-contract=off where results must match the reference's separate
# float32 multiply/add ops bit for bit.
SOURCES = [
    ("linf_kernels.hip", ["-ffp-contract=off"]),
    ("l2_kernels.hip", ["-ffp-contract=off"]),
    ("apgd_control.hip", ["-ffp-contract=off"]),
    ("loss_kernels.hip", []),
    ("loss_stream.hip", []),
    ("loss_split.hip", []),
    ("loss_upsampled.hip", []),
    ("stats_kernels.hip", []),
    ("dwconv_kernels.hip", []),
    ("upsample_kernels.hip", []),
    ("transpose_kernels.hip", ["-ffp-contract=off"]),
    ("wino_kernels.hip", []),
    ("ln_kernels.hip", []),
    ("stem_kernels.hip", []),
    ("fpn_fused.hip", []),
    ("probe_kernels.hip", []),
    ("attention.hip", []),
    ("attention_bf16.hip", []),
    ("gemm_split.hip", []),
    ("gemm_split_big.hip", ["-fno-slp-vectorize"]),
    ("gemm_split_pp.hip", ["-fno-slp-vectorize"]),   # (packed f32 VALU beside MFMAs costs issue time: MI355X guide)
    ("mlp_fused.hip", ["-fno-slp-vectorize"]),
    ("classifier.hip", []),
    ("greedy_host.cpp", ["-ffp-contract=off"]),
    ("api_misc.cpp", []),
]
HEADERS = ["sea_common.h", "loss_common.h", "gemm_split.h", "bilinear_map.h", os.path.join("..", "..", "include", "sea_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _digest() -> str:
    h = hashlib.sha256()
    for name, flags in SOURCES:
        h.update(open(os.path.join(CSRC, name), "rb").read())
        h.update(" ".join(flags).encode())
    for name in HEADERS:
        h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = True) -> str:
    stamp_file = os.path.join(LIBDIR, "libsea_hip.stamp")
    digest = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == digest:
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    common = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wno-unused-result"]

    hdr = hashlib.sha256()
    for name in HEADERS:
        hdr.update(open(os.path.join(CSRC, name), "rb").read())

    def compile_one(item):
        # one object per translation unit, rebuilt only when the unit, a header or its flags changed (the stamp of the
        # LIBRARY still covers everything: see _digest)
        name, flags = item
        if name == "api_misc.cpp":          # the only unit that carries the library stamp (sea_build_info)
            flags = flags + [f'-DSEA_BUILD_STAMP="{digest}"']
        obj = os.path.join(OBJDIR, os.path.splitext(name)[0] + ".o")
        h = hdr.copy()
        h.update(open(os.path.join(CSRC, name), "rb").read())
        h.update(" ".join(common + flags).encode())
        key, keyfile = h.hexdigest(), obj + ".key"
        if not force and os.path.exists(obj) and os.path.exists(keyfile) and open(keyfile).read() == key:
            return obj
        cmd = [hipcc, *common, *flags, "-c", os.path.join(CSRC, name), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(keyfile, "w") as f:
            f.write(key)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB + ".tmp"
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp, *objs]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)
    with open(stamp_file, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
