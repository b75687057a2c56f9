"""Build ``libbv2.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m bert_vits2_amd.build [--force]

The .so lands next to the sources (``bert-vits2_amd/csrc/libbv2.so``): git-ignored, but it travels to the
GPU box with the repo snapshot.  No JIT cache, no torch extension machinery: the library has a plain C ABI
(``include/bv2.h``) and is loaded with ctypes.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.realpath(__file__))     # bert_vits2_amd is a symlink to bert-vits2_amd: one directory, one stamp
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libbv2.so")
STAMP = os.path.join(CSRC, ".libbv2.stamp")

SOURCES = [
    "bv2_api.cpp", "bv2_model.cpp", "bv2_exec.cpp", "bv2_bert.cpp",
    "kernels/conv_mfma.hip", "kernels/conv_x6.hip", "kernels/respair_x6.hip", "kernels/resblock_fused.hip", "kernels/gen_bf16.hip", "kernels/resblock_cl_bf16.hip", "kernels/resblock_c16_bf16.hip", "kernels/respair_cl_bf16.hip", "kernels/enc_f16.hip", "kernels/layernorm.hip", "kernels/attention.hip", "kernels/misc.hip", "kernels/dds_fused.hip", "kernels/flow_boundary.hip", "kernels/bert.hip", "kernels/deberta_attn.hip",
]
EXPORTS = "libbv2.map"     # linker version script: export bv2_* only
HEADERS = [EXPORTS, "bv2_internal.h", "bv2_kernels.h", "kernels/spline.h", "kernels/cl_bf16.h", os.path.join(ROOT, "include", "bv2.h"),
           os.path.join(ROOT, "include", "bv2_testing.h"), os.path.join(ROOT, "include", "bv2_bert.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-x", "hip"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libbv2.so for gfx950)")


def _digest() -> str:
    h = hashlib.sha256()
    for rel in SOURCES + HEADERS:
        p = rel if os.path.isabs(rel) else os.path.join(CSRC, rel)
        with open(p, "rb") as f:
            # name relative to the repo root: the digest must not depend on where the tree is checked out (the GPU box
            # runs a copy under another path and must not rebuild what travelled with it)
            h.update(os.path.relpath(p, ROOT).encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _digest()


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    for rel in SOURCES:
        obj = os.path.join(CSRC, "build", rel.replace("/", "_") + ".o")
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, rel), "-o", obj]
        procs.append((rel, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for rel, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"[bv2 build] {rel} failed:\n{out}\n")
        elif verbose and out.strip():
            sys.stderr.write(f"[bv2 build] {rel}:\n{out}\n")
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={os.path.join(CSRC, EXPORTS)}", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    with open(STAMP, "w") as f:
        f.write(_digest())
    if verbose:
        sys.stderr.write(f"[bv2 build] built {LIB}\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
