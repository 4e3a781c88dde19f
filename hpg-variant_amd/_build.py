"""Build recipe of the gfx950 shared library (in-tree, so the .so travels with
the repo snapshot).  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libhpgv.so")
HOSTLIB = os.path.join(LIBDIR, "libhpgv_host.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-Wall", "-Wextra", "-Wno-unused-function"]
# per translation unit: the bgzip decoder's wave-per-block kernel branches on wave-uniform values only; this keeps its
# control flow as written (scalar branches) instead of structurizing it
UNIT_FLAGS = {"hpgv_inflate_capi.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions"]}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the engine cannot be built")
    return exe


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


ABLATION_LIB = os.path.join(LIBDIR, "ablation", "libhpgv.so")


def build_device_lib(force=False, verbose=False, ablation=False):
    """libhpgv.so from its translation units (csrc/*.hip), compiled side by side: the epistasis unit instantiates its
    scans per fold count and takes minutes, the rest seconds.  ablation=True: the same sources with -DHPGV_ABLATION -- every
    kernel form that lost an A/B comparison compiled in and selectable (hpgv.h "Options") -- into lib/ablation/libhpgv.so,
    for tools/ and for running the GPU suite over those forms (HPGV_LIB=<that file>)."""
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(ROOT, "include", "hpgv.h"))
    units = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    objdir = os.path.join(LIBDIR, "ablation", "obj") if ablation else os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + (["-DHPGV_ABLATION"] if ablation else [])
    LIB = ABLATION_LIB if ablation else globals()["LIB"]
    jobs = []
    objs = []
    for u in units:
        o = os.path.join(objdir, os.path.basename(u)[:-4] + ".o")
        objs.append(o)
        # a unit depends on the headers it includes (the epistasis unit does not see the other kernels' headers)
        own = [h for h in hdrs if os.path.basename(h) in open(u).read() or os.path.basename(h) in ("hpgv.h", "hpgv_internal.h")]
        deps = set(own)
        for h in list(own):
            txt = open(h).read()
            deps.update(x for x in hdrs if os.path.basename(x) in txt)
        if force or _stale(o, [u] + sorted(deps)):
            cmd = [_hipcc()] + flags + UNIT_FLAGS.get(os.path.basename(u), []) + ["-c", "-o", o, u]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    if jobs or force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_host_lib(force=False, verbose=False):
    hdir = os.path.join(HERE, "host")
    if not os.path.isdir(hdir):
        return None
    srcs = [os.path.join(hdir, f) for f in sorted(os.listdir(hdir))]
    srcs.append(os.path.join(ROOT, "include", "hpgv.h"))
    if not force and not _stale(HOSTLIB, srcs + [LIB]):
        return HOSTLIB
    csrcs = [s for s in srcs if s.endswith(".c")]
    cmd = ["gcc", "-O2", "-g", "-std=gnu99", "-fPIC", "-shared", "-fopenmp", "-Wall", "-Wextra",
           "-I", os.path.join(ROOT, "include"), "-I", hdir, "-o", HOSTLIB] + csrcs + \
          ["-L", LIBDIR, "-lhpgv", "-Wl,-rpath,$ORIGIN", "-lm", "-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOSTLIB


def build_all(force=False, verbose=False):
    build_device_lib(force, verbose)
    build_host_lib(force, verbose)
