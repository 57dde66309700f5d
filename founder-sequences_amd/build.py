"""Builds the in-tree HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# translation units of the library: the path (kernels, geometry, phases, sharding, ABI) and the joiners / writers
SRCS = [os.path.join(HERE, "csrc", "fseq_api.hip"), os.path.join(HERE, "csrc", "fseq_api_join.hip"), os.path.join(HERE, "csrc", "fseq_reduced.hip"), os.path.join(HERE, "csrc", "fseq_kernelsets.hip"), os.path.join(HERE, "csrc", "fseq_kernelsets_stream.hip")]
SRC = SRCS[0]
import glob
DEPS = sorted(glob.glob(os.path.join(HERE, "csrc", "*"))) + [os.path.join(os.path.dirname(HERE), "include", "fseq.h"),
                                                             os.path.join(os.path.dirname(HERE), "include", "fseq_debug.h")]
OUT = os.path.join(HERE, "libfseq_hip.so")
OBJ_DIR = os.path.join(HERE, "build")


def hipcc():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    raise RuntimeError("hipcc not found")


FLAGS_FILE = os.path.join(OBJ_DIR, "hipcc_flags.txt")


def extra_flags():
    return os.environ.get("FSEQ_HIPCC_FLAGS", "").split()


def needs_build():
    """Out of date: a source is newer than the library, or the library was built with other FSEQ_HIPCC_FLAGS than the ones in
    force now (a diagnostic build -- cycle stamps -- must not be taken for the product: its library is newer than every source)."""
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    if any(os.path.getmtime(d) > t for d in DEPS):
        return True
    try:
        with open(FLAGS_FILE) as f:
            built_with = f.read().split()
    except OSError:
        built_with = []
    return built_with != extra_flags()


LAST_ACTION = {}        # target -> "compiled" | "reused": what the last build() / build_cli() / build_aux() call did


def build(force=False, verbose=False):
    if not force and not needs_build():
        LAST_ACTION.setdefault("libfseq_hip.so", "reused")       # (a later call in the same process finds it fresh: it stays "compiled")
        return OUT
    LAST_ACTION["libfseq_hip.so"] = "compiled"
    # FSEQ_HIPCC_FLAGS: extra flags for diagnostic builds (-DFSEQ_DP_STAMPS, -DFSEQ_DP_STATS)
    # roctx ranges per phase when the image has the library (rocprofv3 --marker-trace shows them)
    have_roctx = os.path.exists("/opt/rocm/lib/librocprofiler-sdk-roctx.so") and os.path.exists("/opt/rocm/include/rocprofiler-sdk-roctx/roctx.h")
    cflags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + extra_flags() + (["-DFSEQ_WITH_ROCTX"] if have_roctx else [])
    ldflags = ["-L/opt/rocm/lib", "-lrocprofiler-sdk-roctx", "-Wl,-rpath,/opt/rocm/lib"] if have_roctx else []
    os.makedirs(OBJ_DIR, exist_ok=True)
    # the translation units side by side (the kernel TU is ~50 s of hipcc, the joiners a few)
    procs = []
    objs = []
    for src in SRCS:
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc()] + cflags + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ldflags
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(FLAGS_FILE, "w") as f:
        f.write(" ".join(extra_flags()))
    return OUT


CLI_SRC = os.path.join(HERE, "host", "founder_sequences.cpp")
CLI_OUT = os.path.join(HERE, "bin", "founder_sequences")


def build_cli(force=False):
    """Host C++17 front end (same option surface as the reference CLI), linked against the C ABI."""
    build()
    deps = [CLI_SRC, OUT, os.path.join(HERE, "host", "fseq_shard_rccl.hpp"), os.path.join(os.path.dirname(HERE), "include", "fseq.h")]
    if not force and os.path.exists(CLI_OUT) and os.path.getmtime(CLI_OUT) >= max(os.path.getmtime(d) for d in deps):
        LAST_ACTION.setdefault("founder_sequences", "reused")
        return CLI_OUT
    LAST_ACTION["founder_sequences"] = "compiled"
    os.makedirs(os.path.dirname(CLI_OUT), exist_ok=True)
    # --gpus N shards one alignment over the node's GPUs: RCCL (ncclCommInitAll / ncclAllReduce) bound to the C ABI's
    # exchange callback in host/fseq_shard_rccl.hpp, so the front end links librccl and the HIP runtime itself
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", "/opt/rocm/include",
           CLI_SRC, "-o", CLI_OUT, "-L", HERE, "-lfseq_hip", "-Wl,-rpath,$ORIGIN/..", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
           "-lrccl", "-lamdhip64", "-lpthread"]
    subprocess.run(cmd, check=True)
    return CLI_OUT


AUX_TOOLS = ("remove_identity_columns", "insert_identity_columns", "match_founder_sequences")


def build_aux(force=False):
    """The reference's three auxiliary tools (host only, no GPU): same option surface, built with g++."""
    os.makedirs(os.path.dirname(CLI_OUT), exist_ok=True)
    outs = []
    for name in AUX_TOOLS:
        src = os.path.join(HERE, "host", name + ".cpp")
        out = os.path.join(HERE, "bin", name)
        deps = [src, os.path.join(HERE, "host", "aux_common.hpp")]
        if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
            subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", src, "-o", out], check=True)
            LAST_ACTION[name] = "compiled"
        else:
            LAST_ACTION.setdefault(name, "reused")
        outs.append(out)
    return outs


if __name__ == "__main__":
    build_aux(force="--force" in sys.argv)
    build_cli(force="--force" in sys.argv)
    print(build(force="--force" in sys.argv, verbose=True))
