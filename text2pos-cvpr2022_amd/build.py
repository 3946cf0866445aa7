"""Builds libt2p_hip.so (the gfx950 kernels + C ABI of include/t2p.h) in-tree with hipcc.

    python -m text2pos_amd.build            # or: python text2pos-cvpr2022_amd/build.py

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the source tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libt2p_hip.so")
SOURCES = ["api.hip", "sample_group.hip", "small_kernels.hip", "tg_gemm.hip", "tg_gemm_tn.hip", "tg_gemm_x3.hip", "train_gemm.hip", "ws_gemm.hip", "ga2.hip", "ws_sa.hip", "sa_rows.hip", "sa3.hip", "sa_points.hip", "lstm.hip", "train_ops.hip", "sim_topk.hip", "match.hip"]
HEADERS = [os.path.join(CSRC, "t2p_common.h"), os.path.join(HERE, "..", "include", "t2p.h")]
# -ffp-contract=off: the index-producing kernels (FPS, ball query, kNN) pin their fp32 distance arithmetic to the
# oracle's un-contracted form; fused multiply-adds are written explicitly (fmaf / MFMA) where they are wanted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name: str, defines, verbose: bool = False) -> str:
    """A/B build: same sources with extra -D flags into libt2p_hip_<name>.so (select it with T2P_LIB=<path>)."""
    global OBJ, LIB
    saved = (OBJ, LIB)
    OBJ, LIB = os.path.join(CSRC, "_obj_" + name), os.path.join(HERE, f"libt2p_hip_{name}.so")
    try:
        return build_hip(force=False, verbose=verbose, extra_flags=tuple("-D" + d for d in defines))
    finally:
        OBJ, LIB = saved


def build_hip(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([hipcc, *FLAGS, *extra_flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined", "-o", LIB, *objs])
    return LIB


def build_peaks() -> str:
    """profiles/microbench/libt2p_peaks.so: the two micro-kernels bench.py uses to state on-box peaks (dense f16 MFMA rate,
    float4 copy bandwidth).  Measurement code, kept out of the product library."""
    src = os.path.join(HERE, "..", "profiles", "microbench", "peaks.hip")
    lib = os.path.join(HERE, "..", "profiles", "microbench", "libt2p_peaks.so")
    if _stale(lib, [src]):
        r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", lib, src],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on peaks.hip:\n" + r.stderr)
    return os.path.abspath(lib)


def build_host_ext() -> str:
    """_t2p_host: the CPython helper of the drop-in entry point (csrc/host_ext.c: per-object float64 column sums of a whole
    encode_objects call in C, GIL released).  Plain C, built with the host compiler against this interpreter's headers."""
    import sysconfig
    src = os.path.join(CSRC, "host_ext.c")
    lib = os.path.join(HERE, "_t2p_host" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    if _stale(lib, [src]):
        cc = os.environ.get("CC") or "gcc"
        r = subprocess.run([cc, "-O3", "-std=gnu11", "-fPIC", "-shared", "-Wall", "-I" + sysconfig.get_paths()["include"], "-o", lib, src,
                            "-lpthread", "-lm"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building _t2p_host failed:\n" + r.stderr)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python build.py --variant NAME DEF1 DEF2 ...
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        print(build_hip(force="--force" in sys.argv, verbose=True))
        print(build_peaks())
        print(build_host_ext())
