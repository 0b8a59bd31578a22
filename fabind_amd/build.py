"""Builds libfabind_hip.so (gfx950) in-tree with hipcc.  `python -m fabind_amd.build` or `build()`.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libfabind_hip.so")
SOURCES = ["capi.hip", "gemm.hip", "graph.hip", "gcl.hip", "fused_edge.hip", "fused_edge_fwd2.hip", "fused_edge_fwd3.hip", "fused_edge_bwd3.hip", "fused_edge_bwd4.hip", "pair_fused.hip", "attn.hip", "inter_attn_rows.hip", "attn_mfma.hip", "norm.hip", "bwd.hip", "post_optim.hip", "node_chain.hip", "heads.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-I" + INCLUDE, "-I" + CSRC]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    objs = []
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "fused_common.h"), os.path.join(INCLUDE, "fabind_hip.h")]
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            # (rounds 1-5 skipped a missing source: the library then linked without its symbols and only _lib.load() noticed)
            raise RuntimeError("fabind_amd.build: source %s is missing" % sp)
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
